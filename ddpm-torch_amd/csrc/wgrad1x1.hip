// Weight gradient of the 1x1 convolutions (attention projections, skip connections — ddpm_torch/models/unet.py:21-34, :41-60; autograd of
// F.conv2d, ddpm_torch/modules.py:120-123):   dW[n][c] = sum_p dy[p][n] * x[p][c]   (+ db[n] = sum_p dy[p][n]),   bf16 in, fp32 out.
//
// The reduction runs over ALL pixels (K = B*H*W = 32768 .. 131072) while the output is tiny (<= 768 x 768): in the generic tile GEMM
// these launches put 4-24 output tiles x ~25 split-K slices on 256 CUs, every slice ending in a 64 KiB read-modify-write of fp32
// atomics behind an LDS-staged epilogue (104-240 TFLOP/s, 40-55 us for 4.3 GFLOP).  Here:
//   * one block = one 128 (c) x 128 (n) output tile x one contiguous pixel slice; tiles x slices = one block per CU, so a slice is
//     short (8-16 K-steps of 64 pixels) and the kernel is all pipeline: a ring of four 32 KiB stages, both operands DMA'd as stored
//     (pixel-major: k is the slow index) and read with the hardware transpose (ds_read_b64_tr_b16);
//   * MFMA rows = input channels c, columns = output channels n: a lane owns one n and runs of four consecutive c, i.e. 16 contiguous
//     bytes of dW[n][:] — the partial tile is STORED from registers into the slice's own slab copy (no atomics, no LDS staging);
//     ddpm_wgrad_reduce sums the copies in a fixed order (bit-deterministic);
//   * the bias gradient is the column sum of the dy fragments the kernel reads anyway (c-tile 0 only): no separate launch.
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Wg1Args {
    const bf16_t* dy; long long dy_ld; unsigned dy_extent;      // [P][dy_ld], N channels used
    const bf16_t* x; long long x_ld; unsigned x_extent;         // [P][x_ld], C channels used
    float* dw; long long slab_stride;                           // slab s: dw + s * slab_stride, layout [N][C]
    float* dbias; long long bias_stride;                        // slab s: dbias + s * bias_stride (null: no bias gradient)
    int P, C, N, tiles_c, tiles_n, ksteps, ksteps_per_split;
};

constexpr int RING = 4, OP_BYTES = 64 * 256, STAGE = 2 * OP_BYTES;     // stage = x tile [64 px][128 c] | dy tile [64 px][128 n]
constexpr int PER = 4;                                                  // LDS-DMA instructions per thread per stage

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__global__ __launch_bounds__(512, 2)
void wgrad1x1_kernel(Wg1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wn = wave >> 1;                   // wave tile: 64 c x 32 n
    const int ntiles = a.tiles_c * a.tiles_n;
    const int split = blockIdx.x / ntiles, tile = blockIdx.x - split * ntiles;      // tile fastest: the tiles of a slice share its pixels
    const int tn = tile / a.tiles_c, tc = tile - tn * a.tiles_c;
    const int ks0 = split * a.ksteps_per_split, nst = max(min(a.ksteps, ks0 + a.ksteps_per_split) - ks0, 0);

    auto rsrc_of = [&](const void* p, unsigned extent) {
        const unsigned long long ad = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rdy = rsrc_of(a.dy, a.dy_extent), rx = rsrc_of(a.x, a.x_extent);

    // DMA plan: vector v = tid + 512 i -> k-row (pixel) v >> 4 of the stage, PHYSICAL 16-byte chunk v & 15 of its 256-byte row; the logical
    // chunk is XOR-swizzled by (row & 3) << 2 on the source side so that the four k-rows one transpose read touches fall on four
    // different bank groups.  Channels beyond C / N and pixels beyond P: out-of-range offsets (zeros).
    unsigned xoff[2], yoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + 512 * i, row = v >> 4, lc = (v & 15) ^ ((row & 3) << 2);
        const int c = tc * 128 + lc * 8, n = tn * 128 + lc * 8;
        xoff[i] = c < a.C ? (unsigned)(((long long)row * a.x_ld + c) * 2) : 0x7ffffff0u;
        yoff[i] = n < a.N ? (unsigned)(((long long)row * a.dy_ld + n) * 2) : 0x7ffffff0u;
    }
    const unsigned x_step = (unsigned)(64 * a.x_ld * 2), y_step = (unsigned)(64 * a.dy_ld * 2);
    auto issue = [&](int st) {                                  // stage st of this block's slice into ring slot st % RING
        char* dst = smem + (st & (RING - 1)) * STAGE;
        const unsigned kx = (unsigned)(ks0 + st) * x_step, ky = (unsigned)(ks0 + st) * y_step;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + (wave * 64 + 512 * i) * 16), 16,
                                                     xoff[i] == 0x7ffffff0u ? xoff[i] : xoff[i] + kx, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)(dst + OP_BYTES + (wave * 64 + 512 * i) * 16), 16,
                                                     yoff[i] == 0x7ffffff0u ? yoff[i] : yoff[i] + ky, 0, 0, 0);
    };

    // fragment addressing (m-major image, k-rows of 256 bytes): this lane's transpose read of the 32-wide channel block at `rb`, K sub-step kc
    const int i16 = lane & 15;
    auto frag_off = [&](int rb, int kc) {
        const int mcol = rb + ((lane >> 4) & 1) * 16 + (i16 & 3) * 4;
        const int krow = kc * 16 + (lane >> 5) * 8 + (i16 >> 2);
        const int pch = (mcol >> 3) ^ ((krow & 3) << 2);
        return (unsigned)(krow * 256 + pch * 16 + (mcol & 7) * 2);
    };
    unsigned offA[2], offB;                                     // kc = 0; kc adds 16 rows = 4096 bytes (the swizzle key (krow & 3) is unchanged)
    offA[0] = frag_off(wc * 64, 0); offA[1] = frag_off(wc * 64 + 32, 0);
    offB = frag_off(wn * 32, 0) + OP_BYTES;
    const unsigned lds0 = (unsigned)(size_t)smem;

    f32x16 acc[2] = {(f32x16)(0.f), (f32x16)(0.f)};
    float bsum = 0.f;
    const bool want_bias = a.dbias != nullptr && tc == 0 && wc == 0;

#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
        if (t < nst) issue(t);

    for (int st = 0; st < nst; ++st) {
        const int newer = min(RING - 2, nst - 1 - st);
        if (newer >= 2) wait_vm<2 * PER>(); else if (newer == 1) wait_vm<PER>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                           // stage st has landed for every wave; every wave is done reading stage st - 1
        if (st + RING - 1 < nst) issue(st + RING - 1);          // ... whose slot this refills
        const unsigned base = lds0 + (st & (RING - 1)) * STAGE;
        // INLINE-ASM transpose reads (hipcc puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr builtin while LDS-DMA is pending: that
        // would drain the ring at every step).  All 24 reads of the step are issued, one wait, then the MFMAs; no register operands
        // on the wait ("+v" ties make the compiler copy the registers before the data is back), a scheduling barrier instead.
        uint2 fa[4][2][2], fb[4][2];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[kc][i][0]) : "v"(base + offA[i]), "n"(kc * 4096) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[kc][i][1]) : "v"(base + offA[i]), "n"(kc * 4096 + 1024) : "memory");
            }
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[kc][0]) : "v"(base + offB), "n"(kc * 4096) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[kc][1]) : "v"(base + offB), "n"(kc * 4096 + 1024) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const u32x4 vb = {fb[kc][0].x, fb[kc][0].y, fb[kc][1].x, fb[kc][1].y};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 va = {fa[kc][i][0].x, fa[kc][i][0].y, fa[kc][i][1].x, fa[kc][i][1].y};
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc[i], 0, 0, 0);
            }
            if (want_bias)                                      // column n = lane & 31 of the dy fragment: eight pixels of it
                bsum += __uint_as_float(vb.x << 16) + __uint_as_float(vb.x & 0xffff0000u) + __uint_as_float(vb.y << 16) + __uint_as_float(vb.y & 0xffff0000u)
                      + __uint_as_float(vb.z << 16) + __uint_as_float(vb.z & 0xffff0000u) + __uint_as_float(vb.w << 16) + __uint_as_float(vb.w & 0xffff0000u);
        }
    }

    // ---- the partial tile, straight from the accumulators: register r of block i, lane l = row c0 + i*32 + 8 (r >> 2) + 4 (l >> 5) + (r & 3),
    // column n0 + (l & 31) -> four consecutive c of dW[n][:]
    const int n = tn * 128 + wn * 32 + (lane & 31);
    float* slab = a.dw + (long long)split * a.slab_stride;
    if (n < a.N) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = tc * 128 + wc * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
                if (c < a.C) {
                    f32x4v o = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                    *reinterpret_cast<f32x4v*>(slab + (long long)n * a.C + c) = o;
                }
            }
    }
    if (want_bias) {
        bsum += __shfl_xor(bsum, 32, 64);                       // the two k halves of the fragment
        if (lane < 32 && n < a.N) a.dbias[(long long)split * a.bias_stride + n] = bsum;
    }
}

struct Plan { int tiles_c, tiles_n, ksteps, splits, ksteps_per_split; };
static bool make_plan(int P, int C, int N, Plan& p) {
    if (P <= 0 || C <= 0 || N <= 0 || C % 8 || N % 8) return false;
    p.tiles_c = (C + 127) / 128; p.tiles_n = (N + 127) / 128;
    p.ksteps = (P + 63) / 64;                                   // pixels beyond P: out-of-range offsets (zeros)
    const int tiles = p.tiles_c * p.tiles_n;
    // half the chip, like the 3x3 weight-gradient kernel: a launch on every CU makes the main stream's next kernel wait for a whole
    // block of this one (round 5, under the launch-plan step: 256 -> 9.81, 192 -> 9.66, 128 -> 9.65 / 9.73, 96 -> 9.70, 64 -> 9.86 ms per step)
    static const int cu_budget = getenv("DDPM_WGRAD1_CUS") ? atoi(getenv("DDPM_WGRAD1_CUS")) : 128;
    int splits = ddpm_cu_budget(cu_budget) / tiles;             // one block per CU
    if (splits < 1) splits = 1;
    const int max_splits = p.ksteps / 4 > 0 ? p.ksteps / 4 : 1; // a slice keeps >= 4 K-steps
    if (splits > max_splits) splits = max_splits;
    p.ksteps_per_split = (p.ksteps + splits - 1) / splits;
    p.splits = (p.ksteps + p.ksteps_per_split - 1) / p.ksteps_per_split;
    return true;
}

}  // namespace

// Slab copies ddpm_conv1x1_wgrad_nhwc writes for this geometry (0: geometry not covered — the caller keeps ddpm_conv2d_wgrad_nhwc).
extern "C" int ddpm_conv1x1_wgrad_splits(int P, int C, int N) {
    static const bool off = getenv("DDPM_NO_WGRAD1X1") != nullptr;
    // (round 5, under the launch-plan step: the 8 x 8 and 4 x 4 levels' 1x1 weight gradients — P = 8192 / 2048 — on this kernel too:
    //  16384 -> 9.41, 8192 -> 9.39, 2048 -> 9.34 ms per step; the generic kernel's 512-block grids held the main stream up more)
    static const int min_p = getenv("DDPM_WGRAD1_MIN_P") ? atoi(getenv("DDPM_WGRAD1_MIN_P")) : 2048;
    Plan p;
    if (off || P < min_p || !make_plan(P, C, N, p)) return 0;
    return p.splits;
}

// dW (and db) of a 1x1 / stride-1 convolution.  dy: [P][dy_ld] bf16 (N channels), x: [P][x_ld] bf16 (C channels), P = B*H*W pixels.
// Slice s STORES its partial dW ([N][C] fp32) at dw + s * slab_stride and its partial db at dbias + s * bias_stride (dbias may be
// NULL); `splits` must equal ddpm_conv1x1_wgrad_splits(P, C, N); ddpm_wgrad_reduce sums the copies.
extern "C" int ddpm_conv1x1_wgrad_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                                       float* dbias, long long bias_stride, int P, int C, int N, int splits, int dtype, void* stream) {
    if (!dy || !x || !dw) return DDPM_ERR_NULL;
    if (dtype != DDPM_BF16) return DDPM_ERR_DTYPE;
    Plan p;
    if (!make_plan(P, C, N, p) || splits != p.splits || slab_stride < (long long)N * C || (dbias && bias_stride < N) || dy_ld < N || x_ld < C) return DDPM_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(x) || !aligned16(dw) || dy_ld % 8 || x_ld % 8 || slab_stride % 4 || C % 4) return DDPM_ERR_ALIGN;
    const long long dyb = ((long long)(P - 1) * dy_ld + N) * 2, xb = ((long long)(P - 1) * x_ld + C) * 2;
    if (dyb > 0x7ffffff0ll || xb > 0x7ffffff0ll) return DDPM_ERR_SHAPE;
    Wg1Args a; memset(&a, 0, sizeof(a));
    a.dy = (const bf16_t*)dy; a.dy_ld = dy_ld; a.dy_extent = (unsigned)dyb;
    a.x = (const bf16_t*)x; a.x_ld = x_ld; a.x_extent = (unsigned)xb;
    a.dw = dw; a.slab_stride = slab_stride; a.dbias = dbias; a.bias_stride = bias_stride;
    a.P = P; a.C = C; a.N = N; a.tiles_c = p.tiles_c; a.tiles_n = p.tiles_n; a.ksteps = p.ksteps; a.ksteps_per_split = p.ksteps_per_split;
    constexpr int LDS = RING * STAGE;
    static DevOnce attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DDPM_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad1x1_kernel, dim3(p.tiles_c * p.tiles_n * p.splits), dim3(512), LDS, (hipStream_t)stream, a);
    return check_launch();
}
