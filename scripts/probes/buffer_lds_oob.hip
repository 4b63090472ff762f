// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out, int n_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s = (float*)smem;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = -7.0f;       // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, n_bytes, 0x00020000);
    unsigned voff = (threadIdx.x % 3 == 0) ? 0x7ffffff0u : (threadIdx.x % 3 == 1 ? threadIdx.x * 16 : (unsigned)(n_bytes - 8));  // OOB, valid, straddling end
    char* dst = smem + (threadIdx.x >> 6) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = s[i];
}
int main() {
    const int n = 4096; float *g, *o; float h[1024], src[n];
    for (int i = 0; i < n; ++i) src[i] = 1000.f + i;
    hipMalloc(&g, n * 4); hipMalloc(&o, 4096); hipMemcpy(g, src, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096, 0, g, o, n * 4);
    hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
    for (int t = 0; t < 9; ++t) printf("lane %d (%s): %g %g %g %g\n", t, t % 3 == 0 ? "OOB" : (t % 3 == 1 ? "valid" : "straddle"), h[t * 4], h[t * 4 + 1], h[t * 4 + 2], h[t * 4 + 3]);
    return 0;
}
