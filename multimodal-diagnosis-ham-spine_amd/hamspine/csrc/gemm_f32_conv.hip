#include "gemm_launch.h"
namespace hs {
#define L(BM, BN, AK, BKD, V) \
    return launch_with_lds(gemm_f32_kernel<BM, BN, AK, BKD, V>, 2 * (BM + BN) * 32 * 4, 2 * (BM + BN) * 32 * 4, a, grid, s)
#define CFGS(AK, BKD)                                 \
    switch (cfg) {                                    \
        case CFG_128x128: L(128, 128, AK, BKD, true); \
        default: L(64, 64, AK, BKD, true);            \
    }                                                 \
    break;
int launch_f32_conv(int cfg, int combo, const GemmArgs& a, dim3 grid, hipStream_t s) {
    switch (combo) {
        case 3: CFGS(HS_A_CONV, HS_B_KC)
        case 4: CFGS(HS_A_DGRAD, HS_B_WDGRAD)
        case 5: CFGS(HS_A_RC, HS_B_CONV)
    }
    set_error("launch_f32_conv: bad cfg/combo %d/%d", cfg, combo);
    return HS_ERR_ARG;
}
}  // namespace hs
