// instantiations + launcher of the phase-pipelined GEMM body (gemm_p8.h)
#include "gemm_launch.h"
#include "gemm_p8.h"
namespace hs {
__global__ __launch_bounds__(512) void gemm_bf16_p8_256_diag_kernel(const GemmArgs a) {
    gemm_bf16_p8_body<256, 256, 2, 4, false, true, true>(a, blockIdx.x);
}
template <typename K>
static int launch_p8(K kernel, int lds, int threads, const GemmArgs& a, dim3 grid, hipStream_t s) {
    if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, a);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
int launch_bf16_p8_grouped(const GemmArgs* list, const int* first_wg, int n, int total_wgs, bool rs, hipStream_t s) {
    constexpr int lds = 128 * 1024;
    if (rs) {
        auto kernel = gemm_bf16_p8_grouped_kernel<true>;
        if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(512), lds, s, list, first_wg, n);
    } else {
        auto kernel = gemm_bf16_p8_grouped_kernel<false>;
        if (lds_attr_needed((const void*)kernel)) HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(512), lds, s, list, first_wg, n);
    }
    HS_LAUNCH_CHECK();
    return HS_OK;
}
int launch_bf16_p8(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
    const bool rs = a.rowsum[0] != nullptr;
    switch (cfg) {
        case CFG_P8_256:
            if (a.stamps && !rs) return launch_p8(gemm_bf16_p8_256_diag_kernel, 128 * 1024, 512, a, grid, s);
            return rs ? launch_p8(gemm_bf16_p8_256_kernel<true>, 128 * 1024, 512, a, grid, s)
                      : launch_p8(gemm_bf16_p8_256_kernel<false>, 128 * 1024, 512, a, grid, s);
        case CFG_P8_256x128:
            return rs ? launch_p8(gemm_bf16_p8_256x128_kernel<true>, 96 * 1024, 512, a, grid, s)
                      : launch_p8(gemm_bf16_p8_256x128_kernel<false>, 96 * 1024, 512, a, grid, s);
        case CFG_P8_128:
            if (!rs) return launch_p8(gemm_bf16_p8_128_kernel<false>, 64 * 1024, 256, a, grid, s);
    }
    set_error("launch_bf16_p8: bad cfg %d (rowsum %d)", cfg, (int)rs);
    return HS_ERR_ARG;
}
}  // namespace hs
