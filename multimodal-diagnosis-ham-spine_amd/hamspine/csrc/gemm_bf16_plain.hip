#include "gemm_launch.h"
namespace hs {
#define L(BM, BN, BK, AK, BKD) \
    return launch_with_lds(gemm_bf16_kernel<BM, BN, BK, AK, BKD, true>, a.lds_stages * (BM + BN) * BK * 2, 3 * (BM + BN) * BK * 2, a, grid, s)
#define CFGS(AK, BKD)                              \
    switch (cfg) {                                 \
        case CFG_128x128: L(128, 128, 64, AK, BKD); \
        case CFG_128x64: L(128, 64, 64, AK, BKD);  \
        case CFG_64x64: L(64, 64, 64, AK, BKD);    \
        case CFG_128x128x32:                       \
            return launch_with_lds(gemm_bf16_kernel_w3<AK, BKD>, a.lds_stages * 256 * 32 * 2, HS_W3_RING * 256 * 32 * 2, a, grid, s); \
        case CFG_256x128x32:                       \
            return launch_with_lds(gemm_bf16_kernel<256, 128, 32, AK, BKD, true, 4>, a.lds_stages * 384 * 32 * 2, 3 * 384 * 32 * 2, a, grid, s, 512); \
        case CFG_256x128:                          \
            return launch_with_lds(gemm_bf16_kernel<256, 128, 64, AK, BKD, true, 4>, a.lds_stages * 384 * 64 * 2, 3 * 384 * 64 * 2, a, grid, s, 512); \
    }                                              \
    break;
int launch_bf16_plain(int cfg, int combo, const GemmArgs& a, dim3 grid, hipStream_t s) {
    if (a.bnb_partials) {     // data gradient + the following BatchNorm's backward sums
        if (combo == 1 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_bns_kernel<64, 64, HS_A_KC, HS_B_RC>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 1 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_bns_kernel<128, 64, HS_A_KC, HS_B_RC>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        set_error("launch_bf16_plain: no BatchNorm-sum variant for cfg/combo %d/%d", cfg, combo);
        return HS_ERR_ARG;
    }
    if (a.bnf_tickets) {      // 1x1 convolution (forward) + the following BatchNorm's statistics, finished in the launch
        if (combo == 0 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_bnf_kernel<64, 64, 64, HS_A_KC, HS_B_KC>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 0 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_bnf_kernel<128, 64, 64, HS_A_KC, HS_B_KC>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        set_error("launch_bf16_plain: no BatchNorm-finishing variant for cfg/combo %d/%d", cfg, combo);
        return HS_ERR_ARG;
    }
    if (a.persist > 0) {      // more tiles than resident workgroups: the persistent variant (no split-K, no row sums)
        if (combo == 0 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_persistent_kernel<64, 64, HS_A_KC, HS_B_KC>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 0 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_persistent_kernel<128, 64, HS_A_KC, HS_B_KC>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        if (combo == 1 && cfg == CFG_64x64) return launch_with_lds(gemm_bf16_persistent_kernel<64, 64, HS_A_KC, HS_B_RC>, a.lds_stages * 128 * 64 * 2, 3 * 128 * 64 * 2, a, grid, s);
        if (combo == 1 && cfg == CFG_128x64) return launch_with_lds(gemm_bf16_persistent_kernel<128, 64, HS_A_KC, HS_B_RC>, a.lds_stages * 192 * 64 * 2, 3 * 192 * 64 * 2, a, grid, s);
        set_error("launch_bf16_plain: no persistent variant for cfg/combo %d/%d", cfg, combo);
        return HS_ERR_ARG;
    }
    if (combo == 0 && a.rowsum[0]) {        // K-contiguous weight gradient (dY^T, X^T) that also produces the bias gradient
        switch (cfg) {
            case CFG_128x64:
                return launch_with_lds(gemm_bf16_kernel<128, 64, 64, HS_A_KC, HS_B_KC, true, 2, true>, a.lds_stages * 192 * 64 * 2,
                                       3 * 192 * 64 * 2, a, grid, s);
            case CFG_64x64:
                return launch_with_lds(gemm_bf16_kernel<64, 64, 64, HS_A_KC, HS_B_KC, true, 2, true>, a.lds_stages * 128 * 64 * 2,
                                       3 * 128 * 64 * 2, a, grid, s);
        }
        set_error("launch_bf16_plain: rowsum_a with K-contiguous operands needs the 128x64 or 64x64 tile (cfg %d)", cfg);
        return HS_ERR_ARG;
    }
    switch (combo) {
        case 0: CFGS(HS_A_KC, HS_B_KC)
        case 1: CFGS(HS_A_KC, HS_B_RC)
        case 2: CFGS(HS_A_RC, HS_B_RC)
    }
    set_error("launch_bf16_plain: bad cfg/combo %d/%d", cfg, combo);
    return HS_ERR_ARG;
}
int launch_bf16_grouped_plain(int combo, const GemmArgs* list, const int* first_wg, int n, int total_wgs, hipStream_t s) {
    if (combo != 2) {
        set_error("launch_bf16_grouped_plain: only the (row-contiguous, row-contiguous) layout is grouped (combo %d)", combo);
        return HS_ERR_ARG;
    }
    auto kernel = gemm_bf16_grouped_kernel<HS_A_RC, HS_B_RC>;
    constexpr int lds = 3 * 128 * 64 * 2;
    if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(256), lds, s, list, first_wg, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
int launch_bf16_grouped_big(int combo, const GemmArgs* list, const int* first_wg, int n, int total_wgs, hipStream_t s) {
    if (combo != 0) {
        set_error("launch_bf16_grouped_big: only K-contiguous operands (combo %d)", combo);
        return HS_ERR_ARG;
    }
    auto kernel = gemm_bf16_grouped_big_kernel<HS_A_KC, HS_B_KC>;
    constexpr int lds = 3 * 384 * 64 * 2;
    if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(512), lds, s, list, first_wg, n);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
int launch_bf16_grouped_big4(const GemmArgs* items, const int* first, int n, int total_wgs, hipStream_t s) {
    if (n < 1 || n > 4) {
        set_error("launch_bf16_grouped_big4: %d problems (1..4)", n);
        return HS_ERR_ARG;
    }
    GemmArgsPack4 pk;
    memset(&pk, 0, sizeof(pk));
    for (int i = 0; i < n; ++i) {
        pk.a[i] = items[i];
        pk.first[i] = first[i];
    }
    pk.first[n] = total_wgs;
    pk.n = n;
    auto kernel = gemm_bf16_grouped_big4_kernel<HS_A_KC, HS_B_KC>;
    constexpr int lds = 3 * 384 * 64 * 2;
    if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(512), lds, s, pk);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
}  // namespace hs
