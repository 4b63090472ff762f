// Probe: what does ONE wave per SIMD pay per v_mfma_f32_32x32x16_bf16 when other instructions sit between the MFMAs?
// Eight independent accumulators (the consumer stream of conv3x3_pc_kernel), cycles by s_memtime inside the kernel, one block per CU.
//   MODE 0: bare MFMAs            MODE 1: + one satisfied s_waitcnt lgkmcnt(N) per MFMA      MODE 2: + 5 s_waitcnt per 8 MFMAs (the kernel's pattern)
//   MODE 3: + one v_xor per MFMA  MODE 4: + one ds_read_b128 per MFMA (6 per 8, counted waits)  MODE 5: mode 4 with a single lgkmcnt(0) per 8
// PARTNER: 0 = 256 threads (one wave per SIMD); 1 = 512 threads, waves 4-7 spin on an LDS word with s_sleep 1; 2 = waves 4-7 parked at s_sleep 127 loops
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k(unsigned long long* out, float* sink, int iters, int partner) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 64 * 1024 / 4; i += NT) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    if (wave >= 4) {
        if (partner == 1) {
            for (;;) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base + 65000u) : "memory");
                if (__builtin_amdgcn_readfirstlane((int)v) == 0x12345) break;
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
            for (;;) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base + 65000u) : "memory");
                if (__builtin_amdgcn_readfirstlane((int)v) == 0x12345) break;
                __builtin_amdgcn_s_sleep(127);
            }
        }
        return;
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
    u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u}, b = a;
    u32x4 f[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) f[i] = a;
    const unsigned ad = base + (unsigned)((lane & 31) * 128 + ((lane >> 5) << 4));
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (MODE == 1) { asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                if (MODE == 2 && (m == 0 || m == 1 || m == 2 || m == 4 || m == 6)) { asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                if (MODE == 4 && (m == 0 || m == 1 || m == 2 || m == 4 || m == 6)) {
                    if (m == 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); else if (m == 1) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                    else if (m == 2) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); else if (m == 4) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (MODE == 5 && m == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                const u32x4 aa = (MODE >= 4) ? f[m & 1] : a, bb = (MODE >= 4) ? f[2 + (m >> 1)] : b;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa), __builtin_bit_cast(bf16x8, bb), acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 3) { asm volatile("v_xor_b32 %0, 32, %0" : "+v"(a.z)); __builtin_amdgcn_sched_barrier(0); }
                if ((MODE == 4 || MODE == 5) && m < 3) {
                    const unsigned x = ad ^ (unsigned)(q << 5);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(f[2 * m]) : "v"(x) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f[2 * m + 1]) : "v"(x) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[tid] = s + __uint_as_float(f[0].x + f[1].x + f[2].x + f[3].x + f[4].x + f[5].x);
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
    if (NT == 512 && tid == 0) *reinterpret_cast<volatile unsigned*>(smem + 65000) = 0x12345u;      // release the partner waves
}

template <int MODE, int NT>
void run(const char* name, int partner, unsigned long long* out, float* sink) {
    const int iters = 500;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipMemset(out, 0, 256 * 4 * 8);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 65536, 0, out, sink, iters, partner);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int i = 0; i < 1024; ++i) if (h[i]) { sum += (double)h[i]; ++n; }
    printf("%-58s threads=%d partner=%d  %.1f clk per MFMA\n", name, NT, partner, sum / n / (iters * 32.0));
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 4 * 8); hipMalloc(&sink, 1 << 16);
    run<0, 256>("bare MFMAs", 0, out, sink);
    run<1, 256>("one satisfied s_waitcnt per MFMA", 0, out, sink);
    run<2, 256>("5 satisfied s_waitcnt per 8 MFMAs", 0, out, sink);
    run<3, 256>("one v_xor per MFMA", 0, out, sink);
    run<4, 256>("6 ds_read_b128 per 8 MFMAs, counted waits", 0, out, sink);
    run<5, 256>("6 ds_read_b128 per 8 MFMAs, one lgkmcnt(0) per 8", 0, out, sink);
    run<0, 512>("bare MFMAs, partner spinning (s_sleep 1)", 1, out, sink);
    run<0, 512>("bare MFMAs, partner parked (s_sleep 127)", 2, out, sink);
    run<4, 512>("6 reads per 8 MFMAs counted, partner spinning", 1, out, sink);
    run<4, 512>("6 reads per 8 MFMAs counted, partner parked", 2, out, sink);
    return 0;
}
