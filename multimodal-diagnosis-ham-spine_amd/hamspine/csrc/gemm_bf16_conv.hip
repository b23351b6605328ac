#include "gemm_launch.h"
namespace hs {
#define L(BM, BN, BK, AK, BKD) \
    return launch_with_lds(gemm_bf16_kernel<BM, BN, BK, AK, BKD, true>, a.lds_stages * (BM + BN) * BK * 2, 3 * (BM + BN) * BK * 2, a, grid, s)
#define CFGS(AK, BKD)                              \
    switch (cfg) {                                 \
        case CFG_128x128: L(128, 128, 64, AK, BKD); \
        case CFG_128x64: L(128, 64, 64, AK, BKD);  \
        case CFG_64x64: L(64, 64, 64, AK, BKD);    \
    }                                              \
    break;
int launch_bf16_conv(int cfg, int combo, const GemmArgs& a, dim3 grid, hipStream_t s) {
    if (a.bnb_partials) {     // data gradient + the following BatchNorm's backward sums
        if (combo == 4 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_bns_kernel<64, 64, HS_A_DGRAD, HS_B_WDGRAD>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 4 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_bns_kernel<128, 64, HS_A_DGRAD, HS_B_WDGRAD>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        set_error("launch_bf16_conv: no BatchNorm-sum variant for cfg/combo %d/%d", cfg, combo);
        return HS_ERR_ARG;
    }
    if (a.bnf_tickets) {      // forward convolution + the following BatchNorm's statistics, finished in the launch
        if (combo == 3 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_bnf_kernel<64, 64, 64, HS_A_CONV, HS_B_KC>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 3 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_bnf_kernel<128, 64, 64, HS_A_CONV, HS_B_KC>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        if (combo == 3 && cfg == CFG_STEM) return launch_with_lds(gemm_bf16_bnf_kernel<128, 64, 32, HS_A_CONV, HS_B_KC>, a.lds_stages * 192 * 32 * 2, 3 * 192 * 32 * 2, a, grid, s);
        set_error("launch_bf16_conv: no BatchNorm-finishing variant for cfg/combo %d/%d", cfg, combo);
        return HS_ERR_ARG;
    }
    if (a.persist > 0) {      // more tiles than resident workgroups: the persistent variant (no split-K, no row sums)
        if (combo == 3 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_persistent_kernel<64, 64, HS_A_CONV, HS_B_KC>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 3 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_persistent_kernel<128, 64, HS_A_CONV, HS_B_KC>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        if (combo == 4 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_persistent_kernel<64, 64, HS_A_DGRAD, HS_B_WDGRAD>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 4 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_persistent_kernel<128, 64, HS_A_DGRAD, HS_B_WDGRAD>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        set_error("launch_bf16_conv: no persistent variant for cfg/combo %d/%d", cfg, combo);
        return HS_ERR_ARG;
    }
    if (cfg == CFG_STEM && combo == 3) L(128, 64, 32, HS_A_CONV, HS_B_KC);
    switch (combo) {
        case 3: CFGS(HS_A_CONV, HS_B_KC)
        case 4: CFGS(HS_A_DGRAD, HS_B_WDGRAD)
        case 5: CFGS(HS_A_RC, HS_B_CONV)
    }
    set_error("launch_bf16_conv: bad cfg/combo %d/%d", cfg, combo);
    return HS_ERR_ARG;
}
int launch_bf16_grouped_conv(int combo, const GemmArgs* list, const int* first_wg, int n, int total_wgs, hipStream_t s) {
    if (combo != 5) {
        set_error("launch_bf16_grouped_conv: only the weight-gradient layout is grouped (combo %d)", combo);
        return HS_ERR_ARG;
    }
    auto kernel = gemm_bf16_grouped_kernel<HS_A_RC, HS_B_CONV>;
    constexpr int lds = 3 * 128 * 64 * 2;
    if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(256), lds, s, list, first_wg, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
}  // namespace hs
