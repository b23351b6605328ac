// Probe behind a claim the GEMM core relies on (csrc/gemm_core.h): a `buffer_load ... lds` (LDS-DMA) whose source offset is
// beyond the buffer resource's num_records writes ZEROS to its LDS slot (it does not skip the write).  Build:
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe.so tools/probe.hip ; run tools/glds_probe.py on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 out of range
    unsigned off = threadIdx.x < 32 ? threadIdx.x * 16 : 0x7ffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
extern "C" int run_probe(const unsigned* src, unsigned nbytes, unsigned* out) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, nbytes, out);
    return (int)hipDeviceSynchronize();
}
