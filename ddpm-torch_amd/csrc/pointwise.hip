// 1x1 convolutions of the UNet (skip connections, attention projections and their data gradients — ddpm_torch/models/unet.py:
// 21-34 `ResidualBlock.skip`, :41-60 `AttentionBlock.project_in / project_out`) as a PERSISTENT, STREAMING kernel (gfx950, bf16).
//
// These layers have K = C_in of 128..768: two to twelve K-steps per output tile.  In the generic tile GEMM every block is a latency
// chain — request the operands, wait, multiply briefly, stage the tile through LDS, write — and all blocks of a round walk through
// it in lockstep, so the chip alternates between "everyone reads" and "everyone writes" (measured: 2.1 TB/s on 101 MB, the K-loop
// is 2.4 us of a 10.6 us block).  Here one block per CU walks a flat sequence of (tile, K-step) stages through a ring that never
// drains: the operands of the next tile are in flight while the current one is multiplied and written.
//
//   * swapped operands: the MFMA's row operand is the WEIGHT tile, the column operand the PIXEL tile, so an accumulator lane owns
//     one pixel and runs of 4 consecutive output channels — it adds bias / residual, trades halves with its partner lane and stores
//     16-byte pieces of the NHWC row straight from registers.  No LDS staging, no epilogue barrier: the whole LDS is ring, and the stores drain under the next
//     tile's MFMAs.
//   * a block owns whole pixel tiles and loops over the channel tiles inside: the activation tile is fetched from HBM once (the
//     re-reads for further channel tiles hit the CU's own L2 slice), the weights (<= 1.2 MB) stay L2-resident.
//   * 8 waves; tile = TN channels x TM pixels: 256 x 128 (waves 4 x 2, 64 x 64 each — one fragment read per MFMA as in the
//     stationary-halo conv) when N is a multiple of 256, else 128 channels x 256 pixels (waves 2 x 4), or 128 x 128 when the layer
//     has too few pixels to give every CU a 256-pixel tile.
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr unsigned OOB = 0x7ffffff0u;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct PwArgs {
    const bf16_t* x; long long x_ld; unsigned x_extent;     // [M][x_ld], K = C_in contiguous
    const bf16_t* w; unsigned w_extent;                     // [N][K]
    bf16_t* out; long long out_ld;
    const float* bias;
    const bf16_t* res; long long res_ld;
    int accumulate;
    int ablate;                                             // timing experiments (DDPM_PW_ABLATE): 1 no stores, 2 no MFMA
    int M, N, K, tiles_m, tiles_n, ksteps;
    int spread;                                             // 1: blocks walk (pixel tile, channel tile) pairs (fewer pixel tiles than CUs)
};

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

template <int TM, int TN>
__global__ __launch_bounds__(512, 2)
void pw_conv_kernel(PwArgs a) {
    constexpr int WN = TN / 64, WM = 8 / WN;           // 8 waves as WN (64 channels each) x WM (pixels)
    constexpr int MJ = TM / (32 * WM);                 // 32-pixel accumulator blocks per wave (wave = 64 channels x 32 MJ pixels)
    constexpr int XV = TM / 64;                        // x vectors per thread per stage (TM rows x 8 chunks / 512)
    constexpr int WV = TN / 64;                        // w vectors per thread per stage
    constexpr int PER = XV + WV;                       // LDS-DMA instructions per thread per stage
    constexpr int RING = TM + TN == 256 ? 4 : 3;
    constexpr int XB = TM * 128, WB = TN * 128, STAGE = XB + WB;
    static_assert(TN == 128 || TN == 256, "channel tile");
    constexpr int BIAS_AT = RING * STAGE;              // 4 x 1 KiB: bias rows of the tiles in flight (fetched with a tile's first stage)
    static_assert(BIAS_AT + 4096 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;

    auto rsrc_of = [&](const void* p, unsigned extent) {
        const unsigned long long ad = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x, a.x_extent), rw = rsrc_of(a.w, a.w_extent);
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(a.bias ? (const void*)a.bias : (const void*)a.w, a.bias ? (unsigned)(a.N * 4) : 0u);

    // DMA plan (LDS rows of 128 bytes = 64 k; 16-byte chunk c of row r sits at physical chunk c ^ ((r >> 1) & 7))
    unsigned xoff[4], woff[4];            // (fixed bound: hipcc's host pass rejects a lambda capturing an array of template-dependent size)
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = tid + 512 * i, row = v >> 3, lc = (v & 7) ^ ((row >> 1) & 7);
        xoff[i] = (unsigned)(((long long)row * a.x_ld + lc * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int v = tid + 512 * i, row = v >> 3, lc = (v & 7) ^ ((row >> 1) & 7);
        woff[i] = (unsigned)(((long long)row * a.K + lc * 8) * 2);
    }
    const unsigned x_tile_bytes = (unsigned)((long long)TM * a.x_ld * 2), w_tile_bytes = (unsigned)((long long)TN * a.K * 2);

    // the block's stage sequence: pixel tiles mt = blockIdx.x, + gridDim.x, ...; for each, channel tiles 0 .. tiles_n-1; for each, K-steps.
    // `spread` (layers with fewer pixel tiles than CUs): the unit is one (pixel tile, channel tile) pair instead, vt = mt * tiles_n + nt
    const int units = a.spread ? a.tiles_m * a.tiles_n : a.tiles_m;
    const int my_units = units > (int)blockIdx.x ? (units - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = my_units * (a.spread ? 1 : a.tiles_n) * a.ksteps;
    const int mt0 = a.spread ? (int)blockIdx.x / a.tiles_n : (int)blockIdx.x, nt0 = a.spread ? (int)blockIdx.x - mt0 * a.tiles_n : 0;
    int i_vt = blockIdx.x, i_mt = mt0, i_nt = nt0, i_ks = 0, i_q = 0, i_tile = 0;        // issue cursor
    auto issue = [&]() {
        char* dst = smem + (i_q % RING) * STAGE;
        // the tile's bias row rides with its first stage (wave 0 only; issued BEFORE the stage's loads, so it has landed when the stage
        // has).  A vector load in the epilogue instead would have to wait for every LDS-DMA issued before it — loads retire in
        // order — and drain the ring at every tile.  Lanes >= TN / 4 are out of range and write zeros into the slot's padding.
        if (i_ks == 0) {
            if (wave == 0 && a.bias)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + BIAS_AT + (i_tile & 3) * 1024), 16,
                                                         lane < TN / 4 ? (unsigned)((i_nt * TN + lane * 4) * 4) : OOB, 0, 0, 0);
            ++i_tile;
        }
        const unsigned xb = (unsigned)i_mt * x_tile_bytes + (unsigned)(i_ks * 128);
        const unsigned wb = (unsigned)i_nt * w_tile_bytes + (unsigned)(i_ks * 128);
        // rows beyond M / N: the offset runs past the descriptor's extent (rows are laid out in increasing order) -> zeros
#pragma unroll
        for (int i = 0; i < XV; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + (wave * 64 + 512 * i) * 16), 16, xb + xoff[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + XB + (wave * 64 + 512 * i) * 16), 16, wb + woff[i], 0, 0, 0);
        ++i_q;
        if (++i_ks == a.ksteps) {
            i_ks = 0;
            if (a.spread) { i_vt += gridDim.x; i_mt = i_vt / a.tiles_n; i_nt = i_vt - i_mt * a.tiles_n; }
            else if (++i_nt == a.tiles_n) { i_nt = 0; i_mt += gridDim.x; }
        }
    };

    f32x16 acc[2][MJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x16)(0.f);

#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
        if (t < total) issue();

    const int sw = (lane >> 5) ^ ((lane >> 1) & 7);
    int c_vt = blockIdx.x, c_mt = mt0, c_nt = nt0, c_ks = 0, c_tile = 0;     // consume cursor
    constexpr int ST = 4 * MJ;                                 // 16-byte stores per lane in a tile's epilogue
    int st_window = 0;
    for (int q = 0; q < total; ++q) {
        // stage q has landed when at most the operations issued after it are outstanding: the newer stages and — loads and stores
        // retire in issue order — for the RING - 1 stages after a tile's epilogue also its ST stores (waiting those out would park the
        // block for the write burst of the whole chip at every tile).  Waiting for MORE to retire is always safe: the stores are left
        // out when the windows of two epilogues would overlap or the wave stored a partial channel range.
        const int newer = min(RING - 2, total - 1 - q);
        if (st_window > 0) {
            --st_window;
            if (newer >= 2) wait_vm<2 * PER + ST>(); else if (newer == 1) wait_vm<PER + ST>(); else wait_vm<ST>();
        } else {
            if (newer >= 2) wait_vm<2 * PER>(); else if (newer == 1) wait_vm<PER>(); else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();                          // ... for every wave; and every wave is done reading stage q-1
        // last K-step of a tile with a residual / "+=" epilogue: request those rows NOW, ahead of the next stage's LDS-DMA — loads
        // retire in order, so requested in the epilogue they would wait for every stage in flight (and the compiler's wait for them
        // would drain the ring); requested here they only wait for stages that are about to be consumed anyway
        const bool last = c_ks + 1 == a.ksteps;
        u32x2 rv[MJ][2][4], ov[MJ][2][4];
        const bool pre = last && (a.res || a.accumulate);
        if (pre) {
            // INLINE ASM loads: a C++ load here makes hipcc drain every pending LDS-DMA first (`s_waitcnt vmcnt(0)` in front of any
            // ordinary load while LDS-DMA is in flight).  The matching wait is issued by hand in front of the epilogue.
            const int n0 = c_nt * TN + wn * 64, p0 = c_mt * TM + wm * (32 * MJ);
#pragma unroll
            for (int j = 0; j < MJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int p = p0 + j * 32 + (lane & 31), n = n0 + i * 32 + 8 * g + 4 * (lane >> 5);
                        const bool ok = p < a.M && n < a.N;
                        rv[j][i][g] = (u32x2)(0u); ov[j][i][g] = (u32x2)(0u);
                        if (ok && a.res) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rv[j][i][g]) : "v"(a.res + (long long)p * a.res_ld + n) : "memory");
                        if (ok && a.accumulate) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ov[j][i][g]) : "v"(a.out + (long long)p * a.out_ld + n) : "memory");
                    }
        }
        const bool issued = i_q < total, bias_rides = issued && i_ks == 0 && wave == 0 && a.bias;
        if (issued) issue();                                   // into the slot of stage q-1
        const char* xs = smem + (q % RING) * STAGE;
        const char* ws = xs + XB;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            u32x4 fw[2], fx[MJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = *reinterpret_cast<const u32x4*>(ws + (wn * 64 + i * 32 + (lane & 31)) * 128 + (((2 * kc) ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < MJ; ++j) fx[j] = *reinterpret_cast<const u32x4*>(xs + (wm * (32 * MJ) + j * 32 + (lane & 31)) * 128 + (((2 * kc) ^ sw) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j)
                    if (!(a.ablate & 2))
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[i]), __builtin_bit_cast(bf16x8, fx[j]), acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of stage q are complete before it reaches the next barrier
        if (pre) {
            // the rows requested above: everything older than the stage just issued has to be back.  (No register operands on the wait:
            // "+v" ties make hipcc copy the registers in FRONT of the asm, i.e. read them while the loads are in flight; the scheduling
            // barrier keeps every use below the wait instead.)
            if (!issued) wait_vm<0>(); else if (bias_rides) wait_vm<PER + 1>(); else wait_vm<PER>();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (last) {
            // ---- tile done: out[p][n] = acc + bias[n] (+ residual) (+ out), straight from the accumulators.
            // accumulator (i, j), register r, lane l: channel n0 + i*32 + 8 (r >> 2) + 4 (l >> 5) + (r & 3), pixel p0 + j*32 + (l & 31)
            const int n0 = c_nt * TN + wn * 64, p0 = c_mt * TM + wm * (32 * MJ);
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int p = p0 + j * 32 + (lane & 31);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint2 pk[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + i * 32 + 8 * g + 4 * (lane >> 5);
                        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        if (a.bias) {
                            // inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of a C++ LDS read it cannot prove disjoint from the
                            // pending LDS-DMA — that would drain the ring here, at every tile
                            f32x4v bv;
                            const unsigned baddr = (unsigned)(size_t)(smem + BIAS_AT + (c_tile & 3) * 1024 + (n - c_nt * TN) * 4);
                            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(baddr) : "memory");
                            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                        }
                        if (a.res) {
                            const u32x2 r2 = rv[j][i][g];
                            v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
                            v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
                        }
                        if (a.accumulate) {
                            const u32x2 o2 = ov[j][i][g];
                            v[0] += __uint_as_float(o2.x << 16); v[1] += __uint_as_float(o2.x & 0xffff0000u);
                            v[2] += __uint_as_float(o2.y << 16); v[3] += __uint_as_float(o2.y & 0xffff0000u);
                        }
                        pk[g].x = pack_bf2(v[0], v[1]); pk[g].y = pack_bf2(v[2], v[3]);
                    }
                    // lanes l and l + 32 hold channels +0..3 / +4..7 of every 8-channel group of pixel l.  Trade: the lower lane takes
                    // both halves of the even groups, the upper lane both halves of the odd groups (v_permlane32_swap: upper half of the
                    // first register <-> lower half of the second) -> every lane stores 16 contiguous bytes, a lane pair 32.
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const u32x2 sx = __builtin_amdgcn_permlane32_swap(pk[2 * q2].x, pk[2 * q2 + 1].x, false, false);
                        const u32x2 sy = __builtin_amdgcn_permlane32_swap(pk[2 * q2].y, pk[2 * q2 + 1].y, false, false);
                        const int n = n0 + i * 32 + 16 * q2 + 8 * (lane >> 5);
                        if (p < a.M && n < a.N && !(a.ablate & 1)) {
                            u32x4 o; o.x = sx.x; o.y = sy.x; o.z = sx.y; o.w = sy.y;
                            *reinterpret_cast<u32x4*>(a.out + (long long)p * a.out_ld + n) = o;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x16)(0.f);
            // (full 64-channel range stored by this wave -> exactly ST stores; epilogues at least RING - 1 stages apart -> one window at a time)
            st_window = (a.ksteps >= RING - 1 && c_nt * TN + wn * 64 + 64 <= a.N && !(a.ablate & 1)) ? RING - 1 : 0;
            c_ks = 0; ++c_tile;
            if (a.spread) { c_vt += gridDim.x; c_mt = c_vt / a.tiles_n; c_nt = c_vt - c_mt * a.tiles_n; }
            else if (++c_nt == a.tiles_n) { c_nt = 0; c_mt += gridDim.x; }
        } else ++c_ks;
    }
}

}  // namespace

// Launcher behind ddpm_conv2d_nhwc (gemm.hip) for R = S = 1, stride 1, bf16 -> bf16 with bias / residual / accumulate epilogues.
// Returns -1 when the geometry is not covered (the caller keeps the generic kernel), a status code otherwise.  dry: decide only.
int ddpm_pointwise_launch(const void* x, long long x_ld, const void* w, void* y, long long y_ld, const float* bias, const void* residual,
                          long long res_ld, int accumulate, int M, int N, int K, int dry, void* stream) {
    static const bool off = getenv("DDPM_CONV_NO_POINTWISE") != nullptr;
    static const int min_m = getenv("DDPM_POINTWISE_MIN_M") ? atoi(getenv("DDPM_POINTWISE_MIN_M")) : 8192;      // below: too few tiles even as (pixel, channel) pairs (the 64x64-tile kernel with its split-K serves those)
    if (off || M < min_m || K % 64 || N % 8 || x_ld % 8 || y_ld % 8 || (residual && res_ld % 4)) return -1;
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (residual && (((uintptr_t)residual) & 7)) || (bias && !aligned16(bias))) return -1;
    if ((long long)M * x_ld * 2 > 0x7ffffff0ll || (long long)N * K * 2 > 0x7ffffff0ll) return -1;
    if (dry) return DDPM_OK;
    PwArgs a; memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x; a.x_ld = x_ld; a.x_extent = (unsigned)(((long long)(M - 1) * x_ld + K) * 2);
    a.w = (const bf16_t*)w; a.w_extent = (unsigned)((long long)N * K * 2);
    a.out = (bf16_t*)y; a.out_ld = y_ld; a.bias = bias; a.res = (const bf16_t*)residual; a.res_ld = res_ld; a.accumulate = accumulate;
    { const char* e = getenv("DDPM_PW_ABLATE"); a.ablate = e ? atoi(e) : 0; }
    a.M = M; a.N = N; a.K = K; a.ksteps = K / 64;
    hipStream_t st = (hipStream_t)stream;
    // Tile choice.  N >= 256 (the attention projections, the 2C-wide skips and their data gradients): 128 pixels x 256 channels — the
    // activation stage is fetched once for twice the channels, half as many stage hand-overs (barrier + DMA wait) per FLOP and one LDS
    // fragment read per MFMA instead of 1.5: the 16 x 16 layers went from 19.9 / 46.4 / 50.7 us to 10.7 / 19.6 / 23.9 us (256 -> 256,
    // 768 -> 256, 256 -> 768; isolated bursts, scripts/pw_ab.py), the step from 10.10 to 9.60 ms on the same box.  N = 384 takes two such
    // tiles with the second half empty (16.4 vs 28.3 us at 16 x 16).  Otherwise 128-channel tiles: 256 pixels when they give every CU
    // at least one, else 128 pixels.  Layers with fewer 128-pixel tiles than CUs (the 8 x 8 level, M = 8192) spread (pixel, channel)
    // tile pairs over the blocks: 11.9 / 9.1 us against 13.1 / 11.3 us on the generic tile kernel (512 -> 256, 256 -> 512).
    static const bool no_wide = getenv("DDPM_PW_NO_WIDE") != nullptr;        // A/B switch: the round-3 tiling
    const bool small_m = (M + 127) / 128 < 256;
    const bool wide = !no_wide && N >= 256;
    const bool big = (M + 255) / 256 >= 256;
    a.spread = small_m ? 1 : 0;
#define PW_LAUNCH(TM, TN, RINGV)                                                                                                  \
    do {                                                                                                                          \
        constexpr int LDS = RINGV * (TM * 128 + TN * 128) + 4096;                                                                 \
        static DevOnce attr_set;                                                                                                  \
        if (!attr_set) {                                                                                                          \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_conv_kernel<TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
                return DDPM_ERR_LAUNCH;                                                                                           \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        a.tiles_m = (M + TM - 1) / TM; a.tiles_n = (N + TN - 1) / TN;                                                             \
        const int units = a.spread ? a.tiles_m * a.tiles_n : a.tiles_m;                                                           \
        const int cus = ddpm_cu_budget(256);                                                                                      \
        const int grid = units < cus ? units : cus;                                                                               \
        hipLaunchKernelGGL((pw_conv_kernel<TM, TN>), dim3(grid), dim3(512), LDS, st, a);                                          \
    } while (0)
    if (wide) PW_LAUNCH(128, 256, 3); else if (big) PW_LAUNCH(256, 128, 3); else PW_LAUNCH(128, 128, 4);
#undef PW_LAUNCH
    return check_launch();
}
