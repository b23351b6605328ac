// per-translation-unit launchers of the GEMM core (split so that `make -j` compiles them in parallel)
#pragma once
#include "gemm_core.h"
namespace hs {
enum { CFG_128x128 = 0, CFG_128x64 = 1, CFG_64x64 = 2, CFG_STEM = 3, CFG_256x128 = 4, CFG_128x128x32 = 5, CFG_256x128x32 = 6,
       CFG_P8_256 = 7, CFG_P8_256x128 = 8, CFG_P8_128 = 9, CFG_C3 = 10 };   // P8: the phase-pipelined body (gemm_p8.h), full tiles of plain nt GEMMs only
//   // STEM: 128x64 tile with BK = 32; 256x128: 8 waves, plain bf16 GEMMs only
// combos: 0 (KC,KC) 1 (KC,RC) 2 (RC,RC) 3 (CONV,KC) 4 (DGRAD,WDGRAD) 5 (RC,CONV)
int launch_bf16_plain(int cfg, int combo, const GemmArgs& a, dim3 grid, hipStream_t s);
int launch_bf16_conv(int cfg, int combo, const GemmArgs& a, dim3 grid, hipStream_t s);
int launch_bf16_p8(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s);
int launch_bf16_p8_grouped(const GemmArgs* list, const int* first_wg, int n, int total_wgs, bool rs, hipStream_t s);   // 256x256 p8 tiles
int launch_bf16_conv3(const GemmArgs& a, int fm, dim3 grid, hipStream_t s);   // C3: 3x3 stride-1 halo-patch kernel (conv3.hip)
int conv3_lds_bytes(int fm);
// one grid for n <= 64 problems of one operand layout, 64x64 tiles (gemm_bf16_grouped_kernel); tables in device memory
int launch_bf16_grouped_plain(int combo, const GemmArgs* list, const int* first_wg, int n, int total_wgs, hipStream_t s);
int launch_bf16_grouped_big(int combo, const GemmArgs* list, const int* first_wg, int n, int total_wgs, hipStream_t s);   // 256x128 tiles, nt, row sums
int launch_bf16_grouped_big4(const GemmArgs* items, const int* first, int n, int total_wgs, hipStream_t s);   // n <= 4, by value
int launch_bf16_grouped_conv(int combo, const GemmArgs* list, const int* first_wg, int n, int total_wgs, hipStream_t s);
int launch_f32_plain(int cfg, int combo, bool vec, const GemmArgs& a, dim3 grid, hipStream_t s);
int launch_f32_conv(int cfg, int combo, const GemmArgs& a, dim3 grid, hipStream_t s);

bool lds_attr_needed(const void* fn);   // true the first time a kernel pointer is seen (thread safe)

// lds: dynamic LDS of this launch; max_lds: the most this kernel is ever launched with (set once as its limit)
template <typename K>
inline int launch_with_lds(K kernel, int lds, int max_lds, const GemmArgs& a, dim3 grid, hipStream_t s, int threads = 256) {
    if (max_lds >= 48 * 1024 && lds_attr_needed((const void*)kernel)) {
        HS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, a);
    HS_LAUNCH_CHECK();
    return HS_OK;
}
}  // namespace hs
