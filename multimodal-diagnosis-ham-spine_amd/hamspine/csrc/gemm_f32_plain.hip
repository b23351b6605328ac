#include "gemm_launch.h"
namespace hs {
#define L(BM, BN, AK, BKD, V) \
    return launch_with_lds(gemm_f32_kernel<BM, BN, AK, BKD, V>, 2 * (BM + BN) * 32 * 4, 2 * (BM + BN) * 32 * 4, a, grid, s)
#define CFGS(AK, BKD)                                             \
    if (!vec) L(64, 64, AK, BKD, false);                          \
    switch (cfg) {                                                \
        case CFG_128x128: L(128, 128, AK, BKD, true);             \
        default: L(64, 64, AK, BKD, true);                        \
    }                                                             \
    break;
int launch_f32_plain(int cfg, int combo, bool vec, const GemmArgs& a, dim3 grid, hipStream_t s) {
    switch (combo) {
        case 0: CFGS(HS_A_KC, HS_B_KC)
        case 1: CFGS(HS_A_KC, HS_B_RC)
        case 2: CFGS(HS_A_RC, HS_B_RC)
    }
    set_error("launch_f32_plain: bad cfg/combo %d/%d", cfg, combo);
    return HS_ERR_ARG;
}
}  // namespace hs
