// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS[i] (b16) = i; every lane reads at byte address lane*8
// (4 consecutive b16 per lane) — print what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (unsigned short)i;
    __syncthreads();
    unsigned addr;
    const int l = threadIdx.x;
    if (mode == 0) addr = l * 8;                                  // consecutive 8-byte segments
    else addr = ((l & 3) * 128 + ((l >> 2) & 3) * 4 + (l >> 4) * 1024) * 2;   // row (l&3) [pitch 128 el], col block (l>>2)&3, group base
    unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)s;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_base + addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short* o; unsigned short h[256];
    hipMalloc(&o, 512);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, mode);
        hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
