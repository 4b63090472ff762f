// Probe: throughput of direct-to-LDS loads (buffer_load_dwordx4 ... lds) per CU, and whether it overlaps with MFMA work
// issued by the same waves.  Each wave streams 1 KiB per instruction from an L2-resident region into its own LDS ring.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int D, int MF, int DEPTH>
__global__ __launch_bounds__(512) void k(const char* src, float* out, int iters, unsigned region_bytes, int shared) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long ad = (unsigned long long)(src + (shared ? 0 : (size_t)blockIdx.x * region_bytes));
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)region_bytes, 0x00020000);
    char* ring = smem + wave * (DEPTH * D * 1024);
    f32x16 acc[2] = {(f32x16)(0.f), (f32x16)(0.f)};
    bf16x8 fa = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, 1u, 2u, 3u}), fb = fa;
    unsigned off = (unsigned)(((wave * 64 + lane) * 16 + (shared ? blockIdx.x * 40960 : 0)) % region_bytes);
    auto issue = [&](int slot) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + (slot * D + d) * 1024), 16, off, 0, 0, 0);
            off += 8192; if (off >= region_bytes) off -= region_bytes;
        }
    };
#pragma unroll
    for (int t = 0; t < DEPTH - 1; ++t) issue(t);
    int slot = DEPTH - 1;
    for (int it = 0; it < iters; ++it) {
        if (D > 0) issue(slot);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 1], 0, 0, 0);
        if (D > 0) {
            if (D * (DEPTH - 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (D * (DEPTH - 1) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (D * (DEPTH - 1) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (D * (DEPTH - 1) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (D * (DEPTH - 1) == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    s += reinterpret_cast<float*>(smem)[tid];
    if (s == 12345.678f) out[tid] = s;
}

template <int D, int MF, int DEPTH>
void run(int nw, const char* src, float* out, unsigned region, int shared = 0) {
    const int iters = 4000;
    const int lds = nw * DEPTH * (D ? D : 1) * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<D, MF, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<D, MF, DEPTH>), dim3(256), dim3(nw * 64), lds, 0, src, out, 10, region, shared);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<D, MF, DEPTH>), dim3(256), dim3(nw * 64), lds, 0, src, out, iters, region, shared);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us_it = ms * 1e3 / iters;
    const double bytes_cu = (double)nw * D * 1024;
    printf("%swaves=%d  DMA/wave/iter=%d (%5.1f KiB/CU/iter) MFMA/wave/iter=%2d depth=%d region=%4u KiB: %.3f us/iter  -> %6.1f GB/s/CU (%5.1f B/clk @2.1GHz), %6.2f TB/s chip; mfma-only time would be %.3f us\n",
           shared ? "[shared region] " : "", nw, D, bytes_cu / 1024, MF, DEPTH, region / 1024, us_it, bytes_cu / us_it / 1e3, bytes_cu / (us_it * 2100), bytes_cu * 256 / us_it / 1e6,
           (nw / 4.0) * MF * 32 / 2100.0);
}
int main() {
    char* src; float* out;
    hipMalloc(&src, 256u << 20); hipMemset(src, 1, 256u << 20); hipMalloc(&out, 1 << 16);
    // DMA only, L1/L2-hot 64 KiB region per block
    run<4, 0, 4>(4, src, out, 64u << 10);
    run<4, 0, 4>(8, src, out, 64u << 10);
    run<2, 0, 4>(8, src, out, 64u << 10);
    run<4, 0, 2>(4, src, out, 64u << 10);
    // larger region (L2 but not L1): 1 MiB per block
    run<4, 0, 4>(4, src, out, 1u << 20);
    run<4, 0, 4>(8, src, out, 1u << 20);
    // one 2 MiB region shared by all blocks: L2-resident, far larger than L1
    run<4, 0, 4>(4, src, out, 2u << 20, 1);
    run<4, 0, 4>(8, src, out, 2u << 20, 1);
    run<4, 8, 4>(8, src, out, 2u << 20, 1);
    run<4, 0, 4>(8, src, out, 16u << 20, 1);
    // MFMA only
    run<0, 8, 4>(4, src, out, 64u << 10);
    run<0, 16, 4>(4, src, out, 64u << 10);
    // both
    run<4, 8, 4>(4, src, out, 1u << 20);
    run<4, 16, 4>(4, src, out, 1u << 20);
    run<4, 8, 4>(8, src, out, 1u << 20);
    run<2, 8, 4>(8, src, out, 1u << 20);
    run<2, 16, 4>(8, src, out, 1u << 20);
    return 0;
}
