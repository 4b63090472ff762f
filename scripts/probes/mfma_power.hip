// Probe: at the power cap, throughput IS energy per FLOP.  Sustained MFMA-only loops on RANDOM register operands (one block of 4 waves per CU,
// 8 x 16 accumulator registers per wave), in different issue orders / shapes; each variant runs ~1.5 s, TFLOP/s from wall time.
//   V0 kernel order (B changes every 2 MFMAs, A alternates)   V1 A stationary over 4 MFMAs   V2 both change every MFMA
//   V3 16x16x32 (16 accumulators of 4 registers)              V4 constant operands (lower bound on operand toggling)   V5 zero operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(const u32x4* rnd, float* sink, int iters) {
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];                                  // 4 A-type and 4 B-type fragments, reloaded never: register-resident random bf16
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = rnd[(blockIdx.x * 256 + tid) * 8 + i]; b[i] = rnd[(blockIdx.x * 256 + tid) * 8 + 4 + i]; }
    if (V == 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = u32x4{0, 0, 0, 0}; b[i] = u32x4{0, 0, 0, 0}; }
    }
    f32x16 acc[8];
    f32x4v acc4[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc4[i] = (f32x4v)(0.f);
#define MM(m, ai, bi) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ai]), __builtin_bit_cast(bf16x8, b[bi]), acc[m], 0, 0, 0)
    for (int it = 0; it < iters; ++it) {
        if (V == 0 || V == 5) {          // (i, j): (0,0) (1,0) (0,1) (1,1) (0,2) (1,2) (0,3) (1,3): A = a[i] alternates, B = b[j] every 2
            MM(0, 0, 0); MM(1, 1, 0); MM(2, 0, 1); MM(3, 1, 1); MM(4, 0, 2); MM(5, 1, 2); MM(6, 0, 3); MM(7, 1, 3);
            MM(0, 2, 0); MM(1, 3, 0); MM(2, 2, 1); MM(3, 3, 1); MM(4, 2, 2); MM(5, 3, 2); MM(6, 2, 3); MM(7, 3, 3);
        } else if (V == 1) {             // A stationary over four MFMAs
            MM(0, 0, 0); MM(2, 0, 1); MM(4, 0, 2); MM(6, 0, 3); MM(1, 1, 0); MM(3, 1, 1); MM(5, 1, 2); MM(7, 1, 3);
            MM(0, 2, 0); MM(2, 2, 1); MM(4, 2, 2); MM(6, 2, 3); MM(1, 3, 0); MM(3, 3, 1); MM(5, 3, 2); MM(7, 3, 3);
        } else if (V == 2) {             // both operands change every MFMA
            MM(0, 0, 0); MM(3, 1, 1); MM(4, 0, 2); MM(7, 1, 3); MM(1, 1, 0); MM(2, 0, 1); MM(5, 1, 2); MM(6, 0, 3);
            MM(0, 2, 0); MM(3, 3, 1); MM(4, 2, 2); MM(7, 3, 3); MM(1, 3, 0); MM(2, 2, 1); MM(5, 3, 2); MM(6, 2, 3);
        } else if (V == 3) {             // 16x16x32: same FLOPs as 16 of the above = 32 of these... (FLOPs counted by the host per variant)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc4[i * 4 + j], 0, 0, 0);
        } else {                          // V4: constant operands
            MM(0, 0, 0); MM(1, 0, 0); MM(2, 0, 0); MM(3, 0, 0); MM(4, 0, 0); MM(5, 0, 0); MM(6, 0, 0); MM(7, 0, 0);
            MM(0, 0, 0); MM(1, 0, 0); MM(2, 0, 0); MM(3, 0, 0); MM(4, 0, 0); MM(5, 0, 0); MM(6, 0, 0); MM(7, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc4[i].x + acc4[i].y + acc4[i].z + acc4[i].w;
    if (s == 12345.678f) sink[tid] = s;
}

template <int V>
void run(const char* name, const u32x4* rnd, float* sink) {
    const int iters = 400000;                 // x 16 MFMAs of 32 clk = 205 M clk ~ 0.1 s per launch
    const double flops_per_iter = (V == 3 ? 32.0 * 2 * 16 * 16 * 32 : 16.0 * 2 * 32 * 32 * 16) * 4 * 256;     // per iteration, all waves of the chip
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), 0, 0, rnd, sink, iters);      // settle the clocks
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), 0, 0, rnd, sink, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = flops_per_iter * iters * 10 / (ms * 1e-3) / 1e12;
    printf("%-64s %7.1f TFLOP/s  (%.1f %% of 2500; effective clock at full issue rate %.0f MHz)\n", name, tf, tf / 25.0, tf / 2500.0 * 2400.0);
    fflush(stdout);
}
int main() {
    const size_t n = 256 * 256 * 8;
    u32x4* h = (u32x4*)malloc(n * sizeof(u32x4));
    srand(1234);
    for (size_t i = 0; i < n; ++i) {
        unsigned w[4];
        for (int c = 0; c < 4; ++c) {
            unsigned v = 0;
            for (int half = 0; half < 2; ++half) {   // bf16: random sign, exponent 124..129 (|x| in 0.125 .. 8), random mantissa
                const unsigned sign = rand() & 1, ex = 124 + rand() % 6, man = rand() & 127;
                v |= ((sign << 15) | (ex << 7) | man) << (16 * half);
            }
            w[c] = v;
        }
        h[i] = u32x4{w[0], w[1], w[2], w[3]};
    }
    u32x4* rnd; float* sink;
    CHECK(hipMalloc(&rnd, n * sizeof(u32x4))); CHECK(hipMalloc(&sink, 1 << 16));
    CHECK(hipMemcpy(rnd, h, n * sizeof(u32x4), hipMemcpyHostToDevice));
    run<0>("32x32x16, kernel order (B every 2 MFMAs, A alternates)", rnd, sink);
    run<1>("32x32x16, A stationary over 4 MFMAs", rnd, sink);
    run<2>("32x32x16, both operands change every MFMA", rnd, sink);
    run<3>("16x16x32, 4 x 4 accumulators", rnd, sink);
    run<4>("32x32x16, constant operands", rnd, sink);
    run<5>("32x32x16, zero operands", rnd, sink);
    run<0>("32x32x16, kernel order (again)", rnd, sink);
    return 0;
}
