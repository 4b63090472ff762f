// Measurement entry (not on the hot path): the chip's SUSTAINED bf16 MFMA rate on random operands.
//
// v_mfma_f32_32x32x16_bf16 issues every 32 clk per SIMD (2.5 PFLOP/s at 2.4 GHz), but the chip clocks to its power budget and the power
// of the matrix pipe depends on how many operand bits toggle: a register-resident MFMA-only loop runs 2480 TFLOP/s on zero operands and
// ~1700 TFLOP/s (clock ~1.64 GHz) on random ones — profiles/r04_probe_mfma_power.txt.  bench.py runs this probe beside the step so that
// `roofline` can price the conv kernels against what the matrix pipe sustains on real data on THIS box, next to the nominal peak.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_probe_kernel(float* sink, int iters, unsigned seed, int zero) {
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // two bf16 per word: random sign and mantissa, exponent 124..129 (|x| in 0.125 .. 8)
            const unsigned h = mix32(seed ^ mix32(id * 32u + i * 4u + c));
            const unsigned lo = ((h & 1u) << 15) | ((124u + ((h >> 1) % 6u)) << 7) | ((h >> 4) & 127u);
            const unsigned hi = (((h >> 11) & 1u) << 15) | ((124u + ((h >> 12) % 6u)) << 7) | ((h >> 16) & 127u);
            w[c] = zero ? 0u : (lo | (hi << 16));
        }
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        if (i < 4) a[i] = v; else b[i - 4] = v;
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
#define MM(m, ai, bi) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ai]), __builtin_bit_cast(bf16x8, b[bi]), acc[m], 0, 0, 0)
    for (int it = 0; it < iters; ++it) {       // the issue order of the conv kernels: B operand every two MFMAs, A alternating
        MM(0, 0, 0); MM(1, 1, 0); MM(2, 0, 1); MM(3, 1, 1); MM(4, 0, 2); MM(5, 1, 2); MM(6, 0, 3); MM(7, 1, 3);
        MM(0, 2, 0); MM(1, 3, 0); MM(2, 2, 1); MM(3, 3, 1); MM(4, 2, 2); MM(5, 3, 2); MM(6, 2, 3); MM(7, 3, 3);
    }
#undef MM
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;          // keeps the accumulators alive; never true in practice
}

// Stand-in for a collective's copy kernel (bench.py's one-rank data-parallel trace): `blocks` workgroups of 512 threads stream `bytes`
// from src to dst with 16-byte accesses, grid-strided — the shape of an RCCL ring step (few workgroups, memory-bound).  What it measures
// is whether such a kernel, issued from inside the backward, GETS compute units while the persistent MFMA kernels hold theirs.
__global__ __launch_bounds__(512) void copy_probe_kernel(u32x4* dst, const u32x4* src, long long vecs) {
    for (long long i = (long long)blockIdx.x * 512 + threadIdx.x; i < vecs; i += (long long)gridDim.x * 512)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

// ---- compute units left to a concurrent collective.  The persistent kernels size their grids to the chip (one block per CU: 160 KiB of
// LDS, 512 threads x ~250 registers), so a kernel of another stream — RCCL's all-reduce issued from inside the backward — only finds
// a CU at a block boundary.  With n > 0 every persistent launcher (3x3 conv, 1x1 conv, both weight-gradient kernels) plans for 256 - n
// compute units instead.  Process-wide; the engine sets it when a process group is attached (DDPM_DP_RESERVED_CUS).
static std::atomic<int> g_reserved_cus{0};
int ddpm_reserved_cus() { return g_reserved_cus.load(std::memory_order_relaxed); }
extern "C" int ddpm_set_reserved_cus(int n) {
    if (n < 0 || n > 192) return DDPM_ERR_SHAPE;
    g_reserved_cus.store(n, std::memory_order_relaxed);
    return DDPM_OK;
}
extern "C" int ddpm_get_reserved_cus(void) { return ddpm_reserved_cus(); }

extern "C" int ddpm_copy_probe(void* dst, const void* src, long long bytes, int blocks, void* stream) {
    if (!dst || !src) return DDPM_ERR_NULL;
    if (bytes <= 0 || bytes % 16 || blocks <= 0 || blocks > 4096) return DDPM_ERR_SHAPE;
    if (!aligned16(dst) || !aligned16(src)) return DDPM_ERR_ALIGN;
    hipLaunchKernelGGL(copy_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (u32x4*)dst, (const u32x4*)src, bytes / 16);
    return check_launch();
}

// One launch of the MFMA-only loop on every CU (256 blocks x 4 waves: one wave per SIMD, 8 accumulator blocks each):
// 16 * iters MFMAs per wave = iters * 16 * 32768 * 1024 FLOP per launch.  sink: >= 256 floats.  zero_operands: the zero-data ceiling.
extern "C" int ddpm_mfma_probe(float* sink, int iters, int zero_operands, void* stream) {
    if (!sink) return DDPM_ERR_NULL;
    if (iters <= 0) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, sink, iters, 0x9E3779B9u, zero_operands);
    return check_launch();
}
