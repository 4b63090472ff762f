// Probe: steady-state rate of the "LDS fragment reads + MFMA" inner loop on gfx950, one block per CU, no global traffic.
// Per sub-step (kc) a wave reads MI A-fragments and NJ B-fragments (ds_read_b128, conflict-free swizzled tiles) and runs
// MI*NJ v_mfma_f32_32x32x16_bf16.  MODE 0: reads then MFMAs (wait for all); MODE 1: software-pipelined (reads of kc+1
// in flight under the MFMAs of kc); MODE 2: MFMAs only; MODE 3: reads only.  (The block size is a template argument: with a blanket
// __launch_bounds__(1024) the 128x64-per-wave variants were held to 128 registers and spilled their accumulators — 137 TFLOP/s "MFMA only".)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ u32x4 lds_read(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int MI, int NJ>
struct Frags { u32x4 a[MI], b[NJ]; };

template <int MI, int NJ>
__device__ __forceinline__ void issue_reads(Frags<MI, NJ>& f, unsigned abase, unsigned bbase, int kc, int lane) {
    const int row = lane & 31, hi = lane >> 5;
    const unsigned sw = ((((2 * kc) | hi) ^ ((row >> 1) & 7)) << 4) + row * 128;
#pragma unroll
    for (int i = 0; i < MI; ++i) f.a[i] = lds_read(abase + i * 32 * 128 + sw);
#pragma unroll
    for (int j = 0; j < NJ; ++j) f.b[j] = lds_read(bbase + j * 32 * 128 + sw);
}
template <int MI, int NJ>
__device__ __forceinline__ void mfmas(const Frags<MI, NJ>& f, f32x16 (&acc)[MI][NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b[j]), acc[i][j], 0, 0, 0);
}
template <int N, int MI, int NJ>
__device__ __forceinline__ void wait_lgkm(Frags<MI, NJ>& f) {      // ties the wait to the fragment registers
    if constexpr (MI == 2 && NJ == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0]), "+v"(f.b[1]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : "n"(N));
}

template <int MODE, int MI, int NJ, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters, int wm_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int wm = wave % wm_count, wn = wave / wm_count;
    // A region: 64 KiB (512 rows), B region: 64 KiB
    const unsigned abase = base + ((wm * MI * 32) & 511) * 128, bbase = base + 65536 + ((wn * NJ * 32) & 511) * 128;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x16)(0.f);
    Frags<MI, NJ> f0, f1;
    if (MODE == 1) issue_reads(f0, abase, bbase, 0, lane);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                issue_reads(f0, abase, bbase, kc, lane);
                wait_lgkm<0>(f0);
                mfmas(f0, acc);
            }
        } else if (MODE == 1) {
            issue_reads(f1, abase, bbase, 1, lane); wait_lgkm<MI + NJ>(f0); mfmas(f0, acc);
            issue_reads(f0, abase, bbase, 2, lane); wait_lgkm<MI + NJ>(f1); mfmas(f1, acc);
            issue_reads(f1, abase, bbase, 3, lane); wait_lgkm<MI + NJ>(f0); mfmas(f0, acc);
            issue_reads(f0, abase, bbase, 0, lane); wait_lgkm<MI + NJ>(f1); mfmas(f1, acc);
        } else if (MODE == 2) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
                for (int i = 0; i < MI; ++i) f0.a[i] = u32x4{(unsigned)it, 1u, 2u, (unsigned)lane};
#pragma unroll
                for (int j = 0; j < NJ; ++j) f0.b[j] = u32x4{(unsigned)it, 1u, 2u, (unsigned)lane};
                mfmas(f0, acc);
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) { issue_reads(f0, abase, bbase, kc, lane); wait_lgkm<0>(f0); asm volatile("" :: "v"(f0.a[0]), "v"(f0.b[0])); }
        }
    }
    if (MODE == 1) wait_lgkm<0>(f0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[tid] = s;
}

template <int MODE, int MI, int NJ, int NW>
void run(const char* name, int wm_count, float* out) {
    constexpr int nw = NW;
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, MI, NJ, NW * 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, MI, NJ, NW * 64>), dim3(256), dim3(nw * 64), 128 * 1024, 0, out, 10, wm_count);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, MI, NJ, NW * 64>), dim3(256), dim3(nw * 64), 128 * 1024, 0, out, iters, wm_count);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us_per_kc = ms * 1e3 / (iters * 4.0);
    const double flops = 256.0 * nw * iters * 4.0 * MI * NJ * 2.0 * 32 * 32 * 16;
    printf("%-34s waves=%2d tile/wave=%3dx%-3d  %.4f us/kc  %7.1f TFLOP/s (%.1f%% of 2500)\n", name, nw, MI * 32, NJ * 32, us_per_kc,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
}
int main() {
    float* out; hipMalloc(&out, 1 << 16);
    run<2, 2, 2, 8>("mfma only", 4, out);
    run<3, 2, 2, 8>("reads only", 4, out);
    run<0, 2, 2, 4>("read->wait->mfma", 2, out);
    run<0, 2, 2, 8>("read->wait->mfma", 4, out);
    run<0, 2, 2, 16>("read->wait->mfma", 4, out);
    run<1, 2, 2, 4>("pipelined", 2, out);
    run<1, 2, 2, 8>("pipelined", 4, out);
    run<1, 2, 2, 16>("pipelined", 4, out);
    run<2, 4, 2, 4>("mfma only", 2, out);
    run<2, 4, 2, 8>("mfma only", 2, out);
    run<0, 4, 2, 4>("read->wait->mfma", 2, out);
    run<0, 4, 2, 8>("read->wait->mfma", 2, out);
    run<1, 4, 2, 4>("pipelined", 2, out);
    run<1, 4, 2, 8>("pipelined", 2, out);
    return 0;
}
