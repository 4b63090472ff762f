// 3x3 / stride 1 / pad 1 convolution, bf16 — the hot conv of the UNet, forward and data gradient (ddpm_torch/modules.py:66-123 `Conv2d`
// inside ddpm_torch/models/unet.py:63-89 `ResidualBlock`, conv1 / conv2 at :76,79) — as a PERSISTENT stationary-halo kernel for images of 16x16 pixels and up.
//
// Same K-loop as conv3x3_halo_kernel (gemm.hip): a block owns one 16 x 16 output patch x 128 output channels and walks K as
// (64-channel chunk, tap); the 18 x 18 input halo of a chunk is DMA'd into LDS once and serves all nine taps, only the 128 x 64 weight
// tile streams per tap.  What changes is everything around the loop, which was 9.4 us of a 24.9 us block on the 128 -> 128 @ 32 x 32
// layers (prologue 4.9: nothing to multiply until the first halo and weight tile have crossed the chip; epilogue 4.5: the whole fp32
// tile staged through LDS, then written) and ran with every block of the launch in the same phase:
//   * one block per CU loops over its tiles; the step sequence (tile, chunk, tap) is flat: the halo of the NEXT tile's first chunk and
//     its first weight tiles are requested during the last taps of the current tile, so a tile's K-loop starts the moment the
//     previous one ends;
//   * the MFMA operands are swapped (rows = output channels, columns = pixels): an accumulator lane owns ONE pixel and runs of four
//     consecutive channels, adds bias / time-embedding bias / residual itself, trades halves with its partner lane (v_permlane32_swap)
//     and stores 16-byte pieces of the NHWC row straight from registers.  No LDS staging, no epilogue barrier — the LDS holds nothing
//     but the pipeline, and the stores drain under the next tile's MFMAs;
//   * the per-channel bias row and the per-image time-embedding row of a tile ride into LDS with its first weight tile: an ordinary
//     load in the epilogue would have to wait for every LDS-DMA issued before it (loads retire in order) and drain the ring.
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr unsigned OOB = 0x7ffffff0u;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct CsArgs {
    const bf16_t* x; long long x_ld; unsigned x_extent;
    const bf16_t* w; unsigned w_extent;                   // packed [N][9][C]: k = tap * C + c
    bf16_t* out; long long out_ld;
    const float* bias; const float* rowbias; long long rowbias_ld; unsigned rowbias_extent;
    const bf16_t* res; long long res_ld;
    int accumulate;
    int B, H, W, C, N, K;                                 // H, W: OUTPUT (= virtual input) image; ups = 1: x is stored at half that size (nearest 2x)
    int ups;
    int tiles_y, tiles_x, tiles_n, total_tiles, xcd;
    int flags;                                            // conv3x3_pc_kernel: bit 0 = consumers at s_setprio 1
    FastDiv d_tiles_n, d_tpi, d_tiles_x;
};

// Patch geometry.  16 x 16 patches (images of 16 x 16 and up): tile = 256 pixels x 128 channels, 8 waves as 2 (channels) x 4 (pixels) of
// 64 x 64.  8 x 8 patches (the 8 x 8 level, where 256-pixel tiles would leave three quarters of the CUs idle): tile = 64 pixels x 128
// channels, 8 waves as 4 x 2 of 32 x 32 — a K-step is then only four MFMAs per wave, so the ring is deeper (5 tiles in flight) to
// cover the same DMA latency with shorter steps.
template <int PATCH> struct Geo;
template <> struct Geo<16> {
    static constexpr int HWD = 18, HP = 324, LP = 4;           // halo width, halo pixels, log2(patch)
    static constexpr int NI = 6;                                // halo DMA parts: 5 x 512 vectors + 32
    static constexpr int RING = 4;                              // weight tiles in the ring (RING - 1 in flight)
    static constexpr int WM = 4, NIB = 2, NJ = 2;               // pixel waves; 32-channel / 32-pixel accumulator blocks per wave
    static constexpr int HALO_BYTES = 5 * 512 * 16 + 1024;      // 41,984: parts 0-4 by all waves, part 5 by wave 0
};
template <> struct Geo<8> {
    static constexpr int HWD = 10, HP = 100, LP = 3;
    static constexpr int NI = 2;                                // 800 vectors: two parts (the lanes past the halo fetch zeros into the tail)
    static constexpr int RING = 6;
    static constexpr int WM = 2, NIB = 1, NJ = 1;
    static constexpr int HALO_BYTES = 2 * 512 * 16;
};
constexpr int WT_BYTES = 128 * 128;
template <int PATCH> struct Lds {
    static constexpr int DUMP_AT = 2 * Geo<PATCH>::HALO_BYTES + Geo<PATCH>::RING * WT_BYTES;   // 1 KiB the other waves' (all out-of-range) part-5 lanes write zeros to
    static constexpr int ROWS_AT = DUMP_AT + 1024;                                               // 2 slots x (bias row 1 KiB | time-bias row 1 KiB)
    static constexpr int BYTES = ROWS_AT + 4096;
    static_assert(BYTES <= 160 * 1024, "LDS");
};

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ __forceinline__ int logical_tile(int id, int total, int xcd) {       // see xcd_logical_id in gemm.hip
    if (!xcd) return id;
    const int per = total >> 3;
    return id < (per << 3) ? (id & 7) * per + (id >> 3) : id;
}

struct TilePos { int img, py0, px0, tn; };

#ifdef C3_TRACE
// debug builds only (scripts/c3_trace.py): phase stamps of four consecutive steps of waves 0 and 4 (one SIMD's pair) of block 0.  The
// stamps are s_memtime reads into scalar registers with NO wait (the step's closing `s_waitcnt lgkmcnt(0)` covers them); steps
// C3_TRACE_FIRST .. +3 are kept and written out when the block is done — nothing is stored inside the loop
__device__ unsigned long long* g_c3_trace = nullptr;
#define C3_TR(id) asm volatile("s_memtime %0" : "=s"(tr_now[id]))
#ifndef C3_TRACE_FIRST
#define C3_TRACE_FIRST 5
#endif
#else
#define C3_TR(id)
#endif
#ifdef C3_TIMING
__device__ unsigned long long* g_c3_timing = nullptr;        // debug builds only (scripts/c3_timeline.py): [block][8] stamps
#define C3_STAMP(slot) do { if (g_c3_timing && threadIdx.x == 0) g_c3_timing[blockIdx.x * 8 + (slot)] = (slot) == 0 || (slot) == 7 ? wall_clock64() : clock64(); } while (0)
#else
#define C3_STAMP(slot)
#endif

template <int PATCH>
__global__ __launch_bounds__(512, 2)
void conv3x3_stream_kernel(CsArgs a) {
    typedef Geo<PATCH> G_;
    constexpr int HWD = G_::HWD, HP = G_::HP, LP = G_::LP, NI = G_::NI, RING = G_::RING, WM = G_::WM, NIB = G_::NIB, NJ = G_::NJ;
    constexpr int HALO_BYTES = G_::HALO_BYTES, DUMP_AT = Lds<PATCH>::DUMP_AT, ROWS_AT = Lds<PATCH>::ROWS_AT;
    C3_STAMP(0); C3_STAMP(1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wring = smem + 2 * HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;             // wave = 32 NIB channels x 32 NJ pixels

    auto rsrc_of = [&](const void* p, unsigned extent) {
        const unsigned long long ad = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x, a.x_extent), rw = rsrc_of(a.w, a.w_extent);
    const __amdgpu_buffer_rsrc_t rbias = rsrc_of(a.bias ? (const void*)a.bias : (const void*)a.w, a.bias ? (unsigned)(a.N * 4) : 0u);
    const __amdgpu_buffer_rsrc_t rrow = rsrc_of(a.rowbias ? (const void*)a.rowbias : (const void*)a.w, a.rowbias ? a.rowbias_extent : 0u);

    // "+ residual" or "+= out": one extra bf16 tensor added in the epilogue
    const bool extra = a.res || a.accumulate;
    const int G = gridDim.x;
    const int my_tiles = a.total_tiles > (int)blockIdx.x ? (a.total_tiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    if (my_tiles == 0) return;
    auto tile_pos = [&](int k) {
        const int lid = logical_tile(blockIdx.x + k * G, a.total_tiles, a.xcd);
        const int tmi = (int)fdiv((unsigned)lid, a.d_tiles_n);
        const int tpi = a.tiles_y * a.tiles_x;
        const int img = (int)fdiv((unsigned)tmi, a.d_tpi), pt = tmi - img * tpi;
        const int ty = (int)fdiv((unsigned)pt, a.d_tiles_x);
        TilePos t; t.img = img; t.py0 = ty * PATCH; t.px0 = (pt - ty * a.tiles_x) * PATCH; t.tn = lid - tmi * a.tiles_n;
        return t;
    };
    // halo DMA plan of a tile: vector v = tid + 512 i -> halo pixel hp = v >> 3 (row hy, column hx of the 18 x 18 halo), physical chunk
    // v & 7 = logical chunk ^ key, key = (hx >> 1) for 16-wide patches, (hx >> 1) ^ ((hy & 1) << 2) for 8-wide ones: the 16 pixels one
    // ds_read_b128 lane group touches (two half patch rows / four quarter rows) land on 16 distinct (bank half, chunk) pairs for every tap shift.  Pixels outside the image (and the 480 vectors past the halo in part 5) get an out-of-range offset.
    auto halo_plan = [&](const TilePos& t, unsigned (&ho)[NI]) {
        // (opaque thread id: otherwise the row / column / chunk of every part are kept in registers from the prologue on — eighteen
        // loop invariants for a plan that is rebuilt once per tile; at the register cap they were spilled and came back with a
        // `s_waitcnt vmcnt(0)` each, draining the DMA ring at every tile)
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = tid_ + 512 * i, hp = v >> 3;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int iy = t.py0 + hy - 1, ix = t.px0 + hx - 1;
            const bool ok = hp < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int lc = (v & 7) ^ (PATCH == 16 ? (hx >> 1) & 7 : ((hx >> 1) ^ ((hy & 1) << 2)) & 7);
            // nearest-neighbour 2x up-sampling folded into the gather: virtual pixel (iy, ix) is stored pixel (iy >> 1, ix >> 1)
            const int sy = iy >> a.ups, sx = ix >> a.ups;
            ho[i] = ok ? ((unsigned)((t.img * (a.H >> a.ups) + sy) * (a.W >> a.ups) + sx) * (unsigned)a.x_ld + (unsigned)(lc * 8)) * 2u : OOB;     // < 2^31: checked by the launcher
        }
    };
    // one part of a tile's plan (the main loop rebuilds the plan for the NEXT tile part by part, each in the step that requests it —
    // done in one piece at the head of a tile's last chunk, ~170 VALU instructions with integer multiplies, it held the younger wave of
    // every SIMD back by ~700 clk while the matrix pipe idled)
    auto halo_part_off = [&](const TilePos& t, int i) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        const int v = tid_ + 512 * i, hp = v >> 3;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = t.py0 + hy - 1, ix = t.px0 + hx - 1;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int lc = (v & 7) ^ (PATCH == 16 ? (hx >> 1) & 7 : ((hx >> 1) ^ ((hy & 1) << 2)) & 7);
        const int sy = iy >> a.ups, sx = ix >> a.ups;
        return ok ? ((unsigned)((t.img * (a.H >> a.ups) + sy) * (a.W >> a.ups) + sx) * (unsigned)a.x_ld + (unsigned)(lc * 8)) * 2u : OOB;
    };
    auto issue_halo_part = [&](unsigned ho, int i, int cc, char* dst) {
        const unsigned o = ho == OOB ? OOB : ho + (unsigned)(cc * 128);
        char* d = (PATCH == 8 || i < 5) ? dst + (wave * 64 + 512 * i) * 16 : (wave == 0 ? dst + 5 * 512 * 16 : smem + DUMP_AT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)d, 16, o, 0, 0, 0);
    };
    // weight DMA plan: vector v = tid + 512 i -> row n = v >> 3 of the 128-row tile, physical chunk v & 7 = logical ^ ((row >> 1) & 7)
    unsigned wrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + 512 * i, row = v >> 3, lc = (v & 7) ^ ((row >> 1) & 7);
        wrow[i] = (unsigned)(((long long)row * a.K + lc * 8) * 2);
    }
    const unsigned w_tile_bytes = (unsigned)((long long)128 * a.K * 2);
    const int nchunks = a.C >> 6;
    const int steps_per_tile = nchunks * 9, total = my_tiles * steps_per_tile;

    // ---- weight ring.  The tile requested in step (cc, tap) is the one RING - 1 = 3 steps ahead: tap + 3 of the same chunk, or tap - 6 of
    // the next chunk, or — in the last chunk — of chunk 0 of the NEXT tile.  With the tap loop unrolled all of that is static except the
    // chunk / tile switch, which keeps the per-step scalar work (the loop is issue-bound between its barriers) to a few instructions.
    auto issue_w = [&](int tn, int cc, int tap, int slot) {
        char* dst = wring + slot * WT_BYTES;
        const unsigned base = (unsigned)tn * w_tile_bytes + (unsigned)((tap * a.C + cc * 64) * 2);      // rows beyond N: past the extent -> zeros
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + (wave * 64 + 512 * i) * 16), 16, base + wrow[i], 0, 0, 0);
    };
    auto issue_w_half = [&](int tn, int cc, int tap, int slot, int i) {           // one of the two DMA instructions of issue_w
        char* dst = wring + slot * WT_BYTES;
        const unsigned base = (unsigned)tn * w_tile_bytes + (unsigned)((tap * a.C + cc * 64) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + (wave * 64 + 512 * i) * 16), 16, base + wrow[i], 0, 0, 0);
    };
    // first stage of a tile: its bias row (wave 0) and the time-embedding row of its image (wave 1) ride along — issued BEFORE the stage's
    // loads, so they have landed when the stage has.  Lanes >= 32 are out of range and write zeros into the slot's padding.
    auto issue_rows = [&](const TilePos& t, int parity) {
        char* slot = smem + ROWS_AT + parity * 2048;
        if (wave == 0 && a.bias)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (__attribute__((address_space(3))) void*)slot, 16,
                                                     lane < 32 ? (unsigned)((t.tn * 128 + lane * 4) * 4) : OOB, 0, 0, 0);
        if (wave == 1 && a.rowbias)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rrow, (__attribute__((address_space(3))) void*)(slot + 1024), 16,
                                                     lane < 32 ? (unsigned)(((long long)t.img * a.rowbias_ld + t.tn * 128 + lane * 4) * 4) : OOB, 0, 0, 0);
    };

    // this lane's pixels (columns of its two 32-pixel accumulator blocks) -> halo index of tap (0, 0)
    int hp0[NJ], pxl[NJ], pyl[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = wm * (32 * NJ) + j * 32 + (lane & 31);
        hp0[j] = (p >> LP) * HWD + (p & (PATCH - 1));
        pxl[j] = p & (PATCH - 1); pyl[j] = (p >> LP) & 1;
    }
    f32x16 acc[NIB][NJ];
#pragma unroll
    for (int i = 0; i < NIB; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x16)(0.f);

    // ---- prologue: halo of (tile 0, chunk 0), the tile's rows, then the first RING - 1 weight tiles; wait for the halo + tile 0 only
    TilePos cur = tile_pos(0);
    unsigned hoff[NI];
    halo_plan(cur, hoff);
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_halo_part(hoff[i], i, 0, halo);
    issue_rows(cur, 0);
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) issue_w(cur.tn, 0, t, t);
    wait_vm<2 * (RING - 2)>();
    __builtin_amdgcn_s_barrier();

    C3_STAMP(2);
    const int sw = (lane >> 5) ^ ((lane >> 1) & 7);
    const int hi = lane >> 5;
    // ---- fragment reads: INLINE-ASM ds_read_b128 into two register sets, software-pipelined by hand.  Left to hipcc the K sub-steps
    // came out as "read 4 fragments, wait for them, 4 MFMAs" with nothing in flight under the MFMAs (no registers left to hoist the
    // reads: 256-register cap, two waves per SIMD) — every sub-step paid the LDS latency, hidden only by the partner wave: 1800-2000
    // clk per step against 1024 clk of MFMA issue.  Now set (kc + 1) is requested before the MFMAs of set kc are issued, and the LAST
    // set of a step is multiplied at the top of the NEXT step, right after that step's first reads have been requested — its MFMAs
    // cover the one latency a barrier-separated step cannot prefetch across.  The LDS addresses are built per read (one or two
    // VALU operations: the VALU idles in this loop) instead of being kept in registers.
    // weight fragment (i, kc): row (wn * 32 NIB + 32 i + lane % 32) of the tile, 16-byte chunk (2 kc) ^ sw   -> wv[i] ^ (kc << 5) (+ slot)
    // halo fragment (j, kc) of tap (r, s): halo pixel hp0[j] + r * HWD + s, chunk ((2 kc) | hi) ^ key(j, r, s) -> (hv[j] + tap shift) | ((hi ^ key) << 4) ^ (kc << 5)
    const unsigned lds0 = (unsigned)(size_t)smem;
    unsigned wv[NIB], hv[NJ];
#pragma unroll
    for (int i = 0; i < NIB; ++i) wv[i] = lds0 + 2 * HALO_BYTES + (unsigned)((wn * (32 * NIB) + i * 32 + (lane & 31)) * 128 + (sw << 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) hv[j] = lds0 + (unsigned)(hp0[j] * 128);
    u32x4 fw[2][NIB], fx[2][NJ];                           // fragment sets: K sub-step kc uses set kc & 1
    bool pend = false;                                     // set 1 holds the previous step's last sub-step, not multiplied yet
#ifdef C3_TRACE
    int trace_step = 0;
    unsigned long long tr_now[16], tr_keep[4][16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) tr_keep[q][i] = 0;
#endif
    int slot = 0, hb = 0;                                  // ring slot of the step being consumed; halo buffer in use
    constexpr int ST = NIB * (32 * NJ / 16);               // 16-byte stores per lane in a tile's epilogue
    bool st8 = false;                                      // ... of the previous tile: may still be in flight
    for (int k = 0; k < my_tiles; ++k) {
        const bool more_tiles = k + 1 < my_tiles;
        TilePos nxt = cur;
        if (more_tiles) nxt = tile_pos(k + 1);
        for (int cc = 0; cc < nchunks; ++cc) {
            // (the last chunk prefetches the NEXT tile's halo: its plan replaces this tile's, part by part, in the steps that request them)
            const char* hcur = halo + hb * HALO_BYTES;
            char* hnxt = halo + (hb ^ 1) * HALO_BYTES;
            const bool same_tile = cc + 1 < nchunks;
            const bool prefetch = same_tile || more_tiles;  // is there a chunk after this one?
            const bool last_chunk = !same_tile;
            const int ncc = same_tile ? cc + 1 : 0;          // ... its chunk index and its tile's channel tile
            const int ntn = same_tile ? cur.tn : nxt.tn;
            u32x2 rv[NJ][NIB][4];                            // residual OR previous output (the launcher admits one of them)
            bool pre = false;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {              // fully unrolled: the halo-part index and the ring arithmetic are constants
                const int r = tap / 3, s = tap - 3 * r;
                const int shift = r * HWD + s;
                // per-step address parts: weight slot (wave-uniform), halo buffer + tap shift (wave-uniform), and the chunk key of this
                // lane's pixels under the tap's column (and, for 8-wide patches, row) shift
                const unsigned woff = (unsigned)(slot * WT_BYTES);
                const unsigned hoffs = (unsigned)(hb * HALO_BYTES + shift * 128);
                unsigned hk[NJ], wk[NIB];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    // (the empty asm makes the value opaque: hipcc otherwise computes these addresses for all nine taps and every sub-step
                    // ahead of the chunk loop and keeps ~60 registers of loop invariants alive — at the 256-register cap that meant spills,
                    // reloaded with `s_waitcnt vmcnt(0)` at the head of every chunk: the DMA ring drained every nine steps)
                    int px = pxl[j];
                    asm volatile("" : "+v"(px));
                    const int key = PATCH == 16 ? ((px + s) >> 1) & 7 : (((px + s) >> 1) ^ (((pyl[j] + r) & 1) << 2)) & 7;
                    hk[j] = hv[j] + hoffs + (unsigned)((hi ^ key) << 4);
                }
#pragma unroll
                for (int i = 0; i < NIB; ++i) { wk[i] = wv[i]; asm volatile("" : "+v"(wk[i])); }
                // One K sub-step: the MFMAs of set `cw/cx` INTERLEAVED one to one with the reads of the next set `nw/nx` — each read (and its
                // address arithmetic) issues in the shadow of the 32-clk MFMA in front of it, and the partner wave's MFMAs slot in between;
                // issued as "all reads, then all MFMAs" a wave's own instruction stream took ~1000 clk per step for 512 clk of MFMA work
                // (s_memtime trace, scripts/c3_trace.py) and the two waves of a SIMD left the pipe idle a quarter of the time.
                // Read order fw0, fx0, fx1, fw1 / MFMA order (0,0) (0,1) (1,0) (1,1).  mf = false: reads only; kc < 0: MFMAs only.
                auto sub_step = [&](bool mf, const u32x4 (&cw)[NIB], const u32x4 (&cx)[NJ], int kc, u32x4 (&nw)[NIB], u32x4 (&nx)[NJ]) {
                    auto rd_w = [&](int i) {
                        if (kc < 0) return;
                        const unsigned ad = (wk[i] ^ (unsigned)(kc << 5)) + woff;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(nw[i]) : "v"(ad) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    auto rd_x = [&](int j) {
                        if (kc < 0) return;
                        const unsigned ad = hk[j] ^ (unsigned)(kc << 5);
                        asm volatile("ds_read_b128 %0, %1" : "=v"(nx[j]) : "v"(ad) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    auto mm = [&](int i, int j) {
                        if (!mf) return;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cw[i]), __builtin_bit_cast(bf16x8, cx[j]), acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    if constexpr (NIB == 2 && NJ == 2) {
                        rd_w(0); mm(0, 0); rd_x(0); mm(0, 1); rd_x(1); mm(1, 0); rd_w(1); mm(1, 1);
                    } else {
                        static_assert(NIB == 1 && NJ == 1, "interleave pattern");
                        rd_w(0); mm(0, 0); rd_x(0);
                    }
                };
                // every read of the set about to be multiplied has returned (nothing newer is outstanding at this point); the scheduling
                // barrier keeps its MFMAs below the wait (no register operands on the wait: "+v" ties make hipcc copy the fragments)
#define C3_WAIT_SET() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
                // ---- sub-step 0 is requested first thing after the barrier, under the MFMAs of the previous step's last sub-step
                C3_TR(0);
                sub_step(tap > 0 || pend, fw[1], fx[1], 0, fw[0], fx[0]);
                C3_TR(1);
                // DMA requests, in program order: the weight tile RING - 1 steps ahead FIRST (its slot was read in the previous step; every
                // wave is past that barrier), a part of the next halo (next chunk of this tile, or chunk 0 of the next tile) LAST.  Requests
                // retire in order and the step's closing wait is for the weight tile of the next step (an L2 hit, requested three steps
                // back): with the halo part queued behind the weight tile of its step, a part that has to come from HBM has three steps to
                // arrive before a wait touches it instead of two.  Once the first weight tile of the next chunk has landed its whole halo
                // has too (all six parts are requested in taps 0-5, that weight tile in tap 6).  The requests are SPREAD over the step, one
                // behind each group of MFMAs: an LDS-DMA instruction takes 60-180 clk to issue, which the queued MFMAs cover.
                if (tap == 8 && last_chunk && extra) {
                    // last step of the tile with a residual / "+=" epilogue: request those rows NOW, ahead of this step's weight tile.
                    // INLINE ASM: a C++ load would make hipcc drain every pending LDS-DMA first; the wait is issued by hand below.
                    pre = true;
                    const int n0 = cur.tn * 128 + wn * (32 * NIB) + 4 * (lane >> 5);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int p = wm * (32 * NJ) + j * 32 + (lane & 31);
                        // one address per pixel, the 4-channel runs at constant byte offsets.  UNCONDITIONAL loads (a load under `if` makes the
                        // result a phi that the compiler fills right after the asm, before the data is back): waves whose 64 channels lie beyond
                        // N (N % 64 == 0) read the start of their row instead and are zeroed after the wait.
                        const bf16_t* base = (a.res ? a.res : a.out) + ((long long)(cur.img * a.H + cur.py0 + (p >> LP)) * a.W + cur.px0 + (p & (PATCH - 1))) * (a.res ? a.res_ld : a.out_ld)
                                             + (n0 < a.N ? n0 : 0);
#pragma unroll
                        for (int i = 0; i < NIB; ++i)
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(rv[j][i][g]) : "v"(base), "n"((i * 32 + 8 * g) * 2) : "memory");
                    }
                }
                if (tap == 9 - (RING - 1) && last_chunk && prefetch) issue_rows(nxt, (k + 1) & 1);     // rows of the next tile: ahead of its first weight tile
                const int wslot = slot + RING - 1 >= RING ? slot - 1 : slot + RING - 1;
                if (tap + RING - 1 < 9) issue_w_half(cur.tn, cc, tap + RING - 1, wslot, 0);
                else if (prefetch) issue_w_half(ntn, ncc, tap + RING - 1 - 9, wslot, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- sub-steps 1..3 requested under the MFMAs of the sub-step before; the last one stays in its registers for the next step
                C3_TR(2);
                C3_WAIT_SET();
                C3_TR(3);
                sub_step(true, fw[0], fx[0], 1, fw[1], fx[1]);
                C3_TR(4);
                if (tap + RING - 1 < 9) issue_w_half(cur.tn, cc, tap + RING - 1, wslot, 1);
                else if (prefetch) issue_w_half(ntn, ncc, tap + RING - 1 - 9, wslot, 1);
                __builtin_amdgcn_sched_barrier(0);
                C3_TR(5);
                C3_WAIT_SET();
                C3_TR(6);
                sub_step(true, fw[1], fx[1], 2, fw[0], fx[0]);
                C3_TR(7);
                if (tap < NI && prefetch) {
                    if (last_chunk) hoff[tap < NI ? tap : 0] = halo_part_off(nxt, tap);
                    issue_halo_part(hoff[tap < NI ? tap : 0], tap, ncc, hnxt);
                }
                __builtin_amdgcn_sched_barrier(0);
                C3_TR(8);
                C3_WAIT_SET();
                C3_TR(9);
                sub_step(true, fw[0], fx[0], 3, fw[1], fx[1]);
                C3_TR(10);
                if (tap == 8 && last_chunk) {                // the tile ends here: nothing is carried into the epilogue
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // every wave has read its last fragments of this halo buffer once it is past this barrier: the epilogue stages the
                    // output tile through the buffer (nothing is requested into it before the next tile's first step)
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    sub_step(true, fw[1], fx[1], -1, fw[0], fx[0]);
                    pend = false;
                } else pend = true;
#undef C3_WAIT_SET
                slot = slot + 1 == RING ? 0 : slot + 1;
                if (tap == 8 && last_chunk) break;          // the tile's last step: epilogue first, then this step's wait + barrier (below)
                // the next step's weight tile must have landed; the RING - 2 newer tiles (2 DMA instructions each) may stay in flight.  The
                // count ignores newer halo parts: waiting for more to retire is always safe.  Only the last chunk of the block's last tile
                // has fewer than two newer tiles (taps 6, 7).
                // (without a next chunk the tiles run out: after step `tap` only taps tap + 2 .. 8 are newer than the one waited for)
                if (!prefetch && 7 - tap < RING - 2) {
                    if (7 - tap >= 3) wait_vm<6>(); else if (7 - tap == 2) wait_vm<4>(); else if (7 - tap == 1) wait_vm<2>(); else wait_vm<0>();
                }
                else {
                    // exact count: besides the RING - 2 newer weight tiles, the halo parts requested in this step and in the previous one
                    // are newer than the tile waited for (a count that ignores them is safe but waits for a weight tile more than needed —
                    // one step less of latency cover in six steps out of nine); + the previous tile's stores, which sit between the tile
                    // waited for and the newer ones in the (in-order) queue
                    // (halo parts newer than the weight tile waited for: those of this step and of the RING - 2 steps before it, as far
                    //  as they lie in this chunk — the previous chunk's last steps request none)
                    int hp = 0;
                    if (prefetch) {
#pragma unroll
                        for (int b = 0; b <= RING - 2; ++b) hp += (tap - b >= 0 && tap - b < NI) ? 1 : 0;                 // (folds per unrolled tap)
                    }
                    static_assert(RING - 2 <= 4, "hp cases");
                    if (tap < RING - 2 && cc == 0 && st8) {
                        if (hp >= 3) wait_vm<2 * (RING - 2) + ST + 3>(); else if (hp == 2) wait_vm<2 * (RING - 2) + ST + 2>(); else if (hp == 1) wait_vm<2 * (RING - 2) + ST + 1>(); else wait_vm<2 * (RING - 2) + ST>();
                    } else {
                        if (hp >= 3) wait_vm<2 * (RING - 2) + 3>(); else if (hp == 2) wait_vm<2 * (RING - 2) + 2>(); else if (hp == 1) wait_vm<2 * (RING - 2) + 1>(); else wait_vm<2 * (RING - 2)>();
                    }
                }
                C3_TR(11);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                C3_TR(12);
                __builtin_amdgcn_s_barrier();
                C3_TR(13);
#ifdef C3_TRACE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (trace_step == C3_TRACE_FIRST + q) {
#pragma unroll
                        for (int i = 0; i < 14; ++i) tr_keep[q][i] = tr_now[i];
                    }
                ++trace_step;
#endif
            }
            if (last_chunk) {
                if (k == 0) C3_STAMP(3); else if (k == 1) C3_STAMP(5);
                // ---- tile done (the unrolled loop was left before the step's wait): out = acc + bias + time bias (+ residual | + out).
                // accumulator (i, j), register r, lane l: channel n0 + i*32 + 8 (r >> 2) + 4 (l >> 5) + (r & 3), pixel j*32 + (l & 31) of the wave
                const char* rows = smem + ROWS_AT + (k & 1) * 2048;
                // bias + time bias of this lane's 4-channel runs: INLINE-ASM LDS reads (hipcc puts `s_waitcnt vmcnt(0)` in front of a C++ LDS
                // read it cannot prove disjoint from the pending LDS-DMA: that would drain the ring at every tile), eight in flight per wait.
                // ... and the rows requested in the last step: everything older than the weight tile requested after them has to be back
                // (no register operands on the waits: "+v" ties make hipcc copy the registers in front of the asm; the scheduling barriers
                // keep every use below them instead)
                if (pre) {
                    if (prefetch) wait_vm<2>(); else wait_vm<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(cur.tn * 128 + wn * (32 * NIB) < a.N)) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int i = 0; i < NIB; ++i)
#pragma unroll
                                for (int g = 0; g < 4; ++g) rv[j][i][g] = (u32x2)(0u);
                    }
                }
                const int n0 = cur.tn * 128 + wn * (32 * NIB);
                // ---- stores.  In the accumulator layout a lane owns ONE pixel (16-byte pieces of its row): stored from there, every lane of a
                // store instruction is its own 16-byte write request to a different row — 4096 requests per tile and CU, 2.6 us per tile
                // (measured by compiling the stores out), a sixth of a two-tile block.  So each 32-channel block of the wave's tile
                // (32 NJ pixels x 64 bytes) takes a turn through LDS — the halo buffer this tile has finished with; wave-private slices, no
                // barrier — and leaves with FOUR consecutive lanes on the 64 contiguous bytes of a pixel: a quarter of the requests, all of
                // them 64-byte runs that pair up to full lines in L2.  (16-byte slot s of pixel row P sits at s ^ ((P >> 2) & 3): the eight
                // lanes of a ds_write_b128 group and the sixteen of a ds_read_b128 group each touch every bank once.)
                constexpr int STG_PX = 32 * NJ, NRR = STG_PX / 16;
                const unsigned stg = lds0 + (unsigned)(hb * HALO_BYTES + wave * (STG_PX * 64));
                static_assert(8 * STG_PX * 64 <= HALO_BYTES, "staging slices fit the halo buffer");
                // output address of read-back vector rr: pixel rr * 16 + lane / 4 of the wave's pixel block, channels n0 + 8 (lane % 4) ...
                // (built per store from one 64-bit base: four row pointers kept across the loop cost eight registers the kernel does not have)
                bf16_t* const obase = a.out + ((long long)(cur.img * a.H + cur.py0) * a.W + cur.px0) * a.out_ld + n0 + 8 * (lane & 3);
                auto orow = [&](int rr) {
                    int q = lane >> 2;
                    asm volatile("" : "+v"(q));
                    const int p = wm * STG_PX + rr * 16 + q;
                    return obase + (long long)(((p >> LP) * a.W + (p & (PATCH - 1))) * (int)a.out_ld);
                };
#pragma unroll
                for (int i = 0; i < NIB; ++i) {
                    f32x4v bsum[4], brow[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned ad = (unsigned)(size_t)(rows + (wn * (32 * NIB) + i * 32 + 8 * g + 4 * (lane >> 5)) * 4);
                        // (unconditional: a read under `if (a.bias)` makes the result a phi, and the compiler copies an asm output into
                        // its phi register right after the asm — before the data has arrived; absent rows are zeroed after the wait)
                        asm volatile("ds_read_b128 %0, %1" : "=v"(bsum[g]) : "v"(ad) : "memory");
                        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(brow[g]) : "v"(ad) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (!a.bias) bsum[g] = (f32x4v)(0.f);
                        if (!a.rowbias) brow[g] = (f32x4v)(0.f);
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        uint2 pk[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            v[0] += bsum[g].x + brow[g].x; v[1] += bsum[g].y + brow[g].y;
                            v[2] += bsum[g].z + brow[g].z; v[3] += bsum[g].w + brow[g].w;
                            if (extra) {
                                const u32x2 r2 = rv[j][i][g];
                                v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
                                v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
                            }
                            pk[g].x = pack_bf2(v[0], v[1]); pk[g].y = pack_bf2(v[2], v[3]);
                        }
                        // lanes l and l + 32 hold channels +0..3 / +4..7 of every 8-channel group of pixel l: the lower lane takes both halves of
                        // the even groups, the upper lane both halves of the odd groups -> every lane holds 16 contiguous bytes: slot 2 q2 + hi
                        // of pixel row j * 32 + l % 32
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            const u32x2 sx = __builtin_amdgcn_permlane32_swap(pk[2 * q2].x, pk[2 * q2 + 1].x, false, false);
                            const u32x2 sy = __builtin_amdgcn_permlane32_swap(pk[2 * q2].y, pk[2 * q2 + 1].y, false, false);
                            u32x4 o; o.x = sx.x; o.y = sy.x; o.z = sx.y; o.w = sy.y;
                            const int P = j * 32 + (lane & 31), sl = (2 * q2 + hi) ^ ((P >> 2) & 3);
                            const unsigned ad = stg + (unsigned)(P * 64 + (sl << 4));
                            asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(o) : "memory");
                        }
                    }
                    u32x4 back[NRR];
#pragma unroll
                    for (int rr = 0; rr < NRR; ++rr) {
                        const int Q = rr * 16 + (lane >> 2), sl = (lane & 3) ^ ((Q >> 2) & 3);
                        const unsigned ad = stg + (unsigned)(Q * 64 + (sl << 4));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(back[rr]) : "v"(ad) : "memory");     // (a wave's LDS operations execute in order)
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and the slice may be overwritten by the next block
                    __builtin_amdgcn_sched_barrier(0);
                    if (n0 < a.N) {
#pragma unroll
                        for (int rr = 0; rr < NRR; ++rr) {
#ifdef C3_NO_STORE
                            if (back[rr].x == 0x12345678u)
#endif
                            *reinterpret_cast<u32x4*>(orow(rr) + i * 32) = back[rr];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NIB; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x16)(0.f);
                if (k == 0) C3_STAMP(4); else if (k == 1) C3_STAMP(6);
                // the wait + barrier of the tile's last step.  Loads and stores retire in issue order, so the stores just issued (eight per
                // lane, none when the wave's 64 channels lie beyond N) count among the newer operations: waiting them out here would park the
                // block for the whole write burst of the chip.
                st8 = n0 < a.N;
                if (!more_tiles) wait_vm<0>(); else if (st8) wait_vm<2 * (RING - 2) + ST>(); else wait_vm<2 * (RING - 2)>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            hb ^= 1;
        }
        cur = nxt;
    }
    C3_STAMP(7);
#ifdef C3_TRACE
    if (g_c3_trace && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) g_c3_trace[((wave >> 2) * 64 + q) * 16 + i] = tr_keep[q][i];
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same tile, LDS image and K sequence with the waves SPECIALISED and NO per-step barrier (16 x 16 patches only).
//
// What the counters and the timelines said about the kernel above (profiles/r04_conv3x3_*): no bank conflicts, the LDS array busy < 50 %,
// yet a wave is parked in a wait or at the barrier 35 % of its life; every wave carries three LDS-DMA instructions per K-step (60-180 clk
// of issue each, in order with its MFMAs) and one fragment read per MFMA; and a first split into loader and consumer waves that kept the
// barrier ran its 1024 clk of MFMAs per step in 1600 clk with ALL memory traffic compiled out — the pipe drains into every barrier and
// refills behind it.  So here:
//   * waves 0-3 — one per SIMD — only multiply: each owns 64 channels x 128 pixels (2 x 4 accumulator blocks: SIX fragment reads feed
//     EIGHT MFMAs, 0.75 KiB of LDS traffic per MFMA instead of 1).  Their instruction stream is ds_read_b128 / v_mfma in one software
//     pipeline that runs across K-steps, chunks and (but for the epilogue) tiles: the fragments of sub-step q + 1 are requested under
//     the MFMAs of sub-step q, the addresses of the next step are built in the gaps of the current step's last sub-step;
//   * waves 4-7 — the second wave of every SIMD — are LOADERS: they own the whole DMA schedule (weight ring, next halo, bias rows) and
//     count their own vmcnt queue (a wave's counter sees only its own requests: the consumers' stores no longer sit in the ring's
//     in-order queue);
//   * the two sides meet through three COUNTERS IN LDS instead of s_barrier.  `ready`: a loader adds 1 when its share of the next step's
//     weight tile (and everything older: halo, rows) has landed — a consumer starts step s when every loader's ready >= s + 1; it reads
//     the counters with the fragments of step s - 1, so the check costs no latency.  `freed`: a consumer adds 1 when its last read of a
//     step's slot has been issued (LDS executes a wave's operations in order) — a loader refills the slot of step g - 1 when every
//     consumer's freed >= g.  `edone`: the consumers' own rendezvous before a tile's epilogue (which stages through the halo buffer the slowest
//     of them may still be reading).  `ready` and `freed` are FOUR counters each, one per signalling wave (a sum would let a wave that
//     runs ahead stand in for one that is behind), read together by one ds_read_b128.  Nobody waits for a wave that is not late: the
//     consumers drift apart by up to the depth of the ring and the pipe never drains;
//   * bias + time bias are the accumulators' INITIAL value (the rows ride into LDS with the tile's first stage, as above), so the
//     epilogue is convert, swap, stage, store.
constexpr int PC_CNT_AT = Lds<16>::BYTES;                  // counters (below) + a junk area the idle lanes of a signal write to
constexpr int PC_BYTES = PC_CNT_AT + 1024;
static_assert(PC_BYTES <= 160 * 1024, "LDS");
// ONE COUNTER PER WAVE, never a sum over waves: a loader that runs a step ahead of its siblings (it may: only `freed` holds it back)
// would otherwise make up for a sibling's missing signal — `ready >= 4 (s + 1)` held with one loader's quarter of the tile still in
// flight (seen as a rare wrong tile when the block's waves progress unevenly, e.g. next to another kernel).  ready[4] and freed[4] are
// 16 contiguous bytes each: one ds_read_b128 returns all four, the poller takes the minimum.
constexpr int PC_READY = 0, PC_FREED = 16, PC_EDONE = 32, PC_JUNK = 256;

// the smallest of four counters is >= target (wrap-safe signed distance)
__device__ __forceinline__ int pc_min_dist(const u32x4& v, unsigned target) {
    const int a = (int)(v.x - target), b = (int)(v.y - target), c = (int)(v.z - target), d = (int)(v.w - target);
    const int m = min(min(a, b), min(c, d));
    return __builtin_amdgcn_readfirstlane(m);
}
// ---- what a timed-out semaphore wait leaves behind (see pc_wait_all_ge)
__device__ unsigned g_pc_fault[4];
constexpr unsigned long long PC_WAIT_TICKS = 500000000ull;          // 5 s of the 100 MHz s_memrealtime clock
__device__ __forceinline__ bool pc_expired(unsigned long long& t_first) {
    const unsigned long long now = wall_clock64();
    if (t_first == 0) { t_first = now | 1ull; return false; }
    return now - t_first > PC_WAIT_TICKS;
}
__device__ __forceinline__ void pc_fault(unsigned addr, unsigned target, unsigned kind) {
    if ((threadIdx.x & 63) == 0) {
        g_pc_fault[0] = blockIdx.x; g_pc_fault[1] = addr; g_pc_fault[2] = target; g_pc_fault[3] = 1u + kind;
        __threadfence_system();
    }
    __builtin_trap();
}
__device__ __forceinline__ void pc_wait_all_ge(unsigned addr, unsigned target) {
#ifdef PC_ABL_NO_SYNC
    return;
#endif
    // bounded: a protocol error must end the launch with an error the host sees (trap), never hang the GPU.  The bound is WALL time
    // (s_memrealtime, 100 MHz): ~5 s, so that a profiler's thread trace, a debugger or a stalled co-tenant cannot turn a slow wait into
    // a fault; the clock is looked at every 4096 polls only.  pc_fault() leaves {block, counter address, target, 1 + kind} in
    // g_pc_fault before the trap (rocgdb / a core file name the wait that tripped; ddpm_conv3x3_pc_last_fault reads it where the
    // context survives).  DDPM_CONV_NO_PC=1 runs every call on conv3x3_stream_kernel<16>, which has no such waits.
    unsigned long long t_first = 0;
    for (unsigned spins = 0;; ++spins) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        if (pc_min_dist(v, target) >= 0) break;
        if ((spins & 4095u) == 4095u && pc_expired(t_first)) pc_fault(addr, target, 0u);
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void pc_wait_ge(unsigned addr, unsigned target) {          // one (summed) counter: the consumers' rendezvous
#ifdef PC_ABL_NO_SYNC
    return;
#endif
    unsigned long long t_first = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
        if ((spins & 4095u) == 4095u && pc_expired(t_first)) pc_fault(addr, target, 1u);
        __builtin_amdgcn_s_sleep(1);
    }
}
// +1 on a counter: one ds_add_u32 of the whole wave, lane 0 on the counter, every other lane on its own word of the junk area (no exec
// masking, no bank conflict); `lane_addr` is built once per wave
__device__ __forceinline__ void pc_signal(unsigned lane_addr) {
#ifdef PC_ABL_NO_SYNC
    return;
#endif
    const unsigned one = 1u;
    asm volatile("ds_add_u32 %0, %1" :: "v"(lane_addr), "v"(one) : "memory");
}

#ifdef C3_TIMING
// debug builds only (scripts/pc_timeline.py): consumer wave 0's phase stamps [block][8] as for the kernel above, then [block][64] clocks at the end of its first 64 K-steps
#define PC_STAMP(slot) do { if (g_c3_timing && threadIdx.x == 0) g_c3_timing[blockIdx.x * 8 + (slot)] = (slot) == 0 || (slot) == 7 ? wall_clock64() : clock64(); } while (0)
#ifdef PC_ABL_NO_STEPSTAMP
#define PC_STEP()
#else
#define PC_STEP() do { if (g_c3_timing && threadIdx.x == 0 && tstep < 64) g_c3_timing[2048 + blockIdx.x * 64 + tstep] = clock64(); ++tstep; } while (0)
#endif
#else
#define PC_STAMP(slot)
#define PC_STEP()
#endif
__global__ __launch_bounds__(512, 2)
void conv3x3_pc_kernel(CsArgs a) {
    typedef Geo<16> G_;
    constexpr int HWD = G_::HWD, HP = G_::HP, RING = G_::RING;
    constexpr int HALO_BYTES = G_::HALO_BYTES, DUMP_AT = Lds<16>::DUMP_AT, ROWS_AT = Lds<16>::ROWS_AT;
    constexpr int NP = 11;                                  // halo parts of 256 vectors: 10 x 256 + 32
    static_assert(RING == 4, "the vmcnt tables below are written for a ring of four");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wring = smem + 2 * HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int my_tiles = a.total_tiles > (int)blockIdx.x ? (a.total_tiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    if (my_tiles == 0) return;
    if (tid < 256) reinterpret_cast<unsigned*>(smem + PC_CNT_AT)[tid] = 0u;
    __syncthreads();                                        // the only barrier of the kernel
    auto tile_pos = [&](int k) {
        const int lid = logical_tile(blockIdx.x + k * G, a.total_tiles, a.xcd);
        const int tmi = (int)fdiv((unsigned)lid, a.d_tiles_n);
        const int tpi = a.tiles_y * a.tiles_x;
        const int img = (int)fdiv((unsigned)tmi, a.d_tpi), pt = tmi - img * tpi;
        const int ty = (int)fdiv((unsigned)pt, a.d_tiles_x);
        TilePos t; t.img = img; t.py0 = ty * 16; t.px0 = (pt - ty * a.tiles_x) * 16; t.tn = lid - tmi * a.tiles_n;
        return t;
    };
    const int nchunks = a.C >> 6;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned cnt0 = lds0 + PC_CNT_AT, junk = cnt0 + PC_JUNK + (unsigned)(lane * 4);

    if (wave >= 4) {
        // =============================================================== loaders
        const int lw = wave - 4, lid = lw * 64 + lane;
        auto rsrc_of = [&](const void* p, unsigned extent) {
            const unsigned long long ad = (unsigned long long)p;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                     __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x, a.x_extent), rw = rsrc_of(a.w, a.w_extent);
        const __amdgpu_buffer_rsrc_t rbias = rsrc_of(a.bias ? (const void*)a.bias : (const void*)a.w, a.bias ? (unsigned)(a.N * 4) : 0u);
        const __amdgpu_buffer_rsrc_t rrow = rsrc_of(a.rowbias ? (const void*)a.rowbias : (const void*)a.w, a.rowbias ? a.rowbias_extent : 0u);
        // halo part p of a tile: vector v = lid + 256 p -> halo pixel v >> 3, physical 16-byte chunk v & 7 = logical ^ key(hx) (the image the
        // consumers' reads expect is the one described at halo_plan above)
        auto halo_part_off = [&](const TilePos& t, int p) {
            int lid_ = lid;
            asm volatile("" : "+v"(lid_));
            const int v = lid_ + 256 * p, hp = v >> 3;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int iy = t.py0 + hy - 1, ix = t.px0 + hx - 1;
            const bool ok = hp < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int lc = (v & 7) ^ ((hx >> 1) & 7);
            const int sy = iy >> a.ups, sx = ix >> a.ups;
            return ok ? ((unsigned)((t.img * (a.H >> a.ups) + sy) * (a.W >> a.ups) + sx) * (unsigned)a.x_ld + (unsigned)(lc * 8)) * 2u : OOB;
        };
        auto issue_halo_part = [&](const TilePos& t, int p, int cc, char* dst) {
            const unsigned ho = halo_part_off(t, p);
            const unsigned o = ho == OOB ? OOB : ho + (unsigned)(cc * 128);
            // part 10 holds the last 32 vectors: loader 0 writes them (+ 32 zero vectors of padding), the others' lanes are all out of range
            char* d = p < 10 ? dst + (lw * 64 + 256 * p) * 16 : (lw == 0 ? dst + 2560 * 16 : smem + DUMP_AT);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)d, 16, o, 0, 0, 0);
        };
        // weight tile: vector v = lid + 256 i -> row v >> 3, physical chunk v & 7 = logical ^ ((row >> 1) & 7)
        unsigned wrow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = lid + 256 * i, row = v >> 3, lc = (v & 7) ^ ((row >> 1) & 7);
            wrow[i] = (unsigned)(((long long)row * a.K + lc * 8) * 2);
        }
        const unsigned w_tile_bytes = (unsigned)((long long)128 * a.K * 2);
        auto issue_w = [&](int tn, int cc, int tap, int slot) {
            char* dst = wring + slot * WT_BYTES;
            const unsigned base = (unsigned)tn * w_tile_bytes + (unsigned)((tap * a.C + cc * 64) * 2);      // rows beyond N: past the extent -> zeros
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + (lw * 64 + 256 * i) * 16), 16, base + wrow[i], 0, 0, 0);
        };
        // the tile's bias row (loader 0) and its image's time-embedding row (loader 1) — zeros when the launch has none: the consumers start
        // their accumulators from these rows unconditionally.  EVERY loader issues exactly one request here (the others an out-of-range
        // one into the dump area) so that the vmcnt tables below are the same for all four
        auto issue_rows = [&](const TilePos& t, int parity) {
            char* slot = smem + ROWS_AT + parity * 2048;
            if (lw == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (__attribute__((address_space(3))) void*)slot, 16,
                                                         (lane < 32 && a.bias) ? (unsigned)((t.tn * 128 + lane * 4) * 4) : OOB, 0, 0, 0);
            else if (lw == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrow, (__attribute__((address_space(3))) void*)(slot + 1024), 16,
                                                         (lane < 32 && a.rowbias) ? (unsigned)(((long long)t.img * a.rowbias_ld + t.tn * 128 + lane * 4) * 4) : OOB, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem + DUMP_AT), 16, OOB, 0, 0, 0);
        };
        TilePos cur = tile_pos(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) issue_halo_part(cur, p, 0, halo);
        issue_rows(cur, 0);
#pragma unroll
        for (int t = 0; t < RING - 1; ++t) issue_w(cur.tn, 0, t, t);
        const unsigned sig_ready = lane == 0 ? cnt0 + PC_READY + (unsigned)(lw * 4) : junk;
        wait_vm<8>();                                       // the halo, the rows and weight tile 0 are in; tiles 1 and 2 may be on their way
        pc_signal(sig_ready);
#ifdef PC_ABL_NO_LOADERS
        return;
#endif
        int slot = 0, hb = 0;
        unsigned gstep = 0;                                 // flat step index
        for (int k = 0; k < my_tiles; ++k) {
            const bool more_tiles = k + 1 < my_tiles;
            TilePos nxt = cur;
            if (more_tiles) nxt = tile_pos(k + 1);
            for (int cc = 0; cc < nchunks; ++cc) {
                char* hnxt = halo + (hb ^ 1) * HALO_BYTES;
                const bool same_tile = cc + 1 < nchunks;
                const bool prefetch = same_tile || more_tiles;
                const bool last_chunk = !same_tile;
                const int ncc = same_tile ? cc + 1 : 0;
                const int ntn = same_tile ? cur.tn : nxt.tn;
                const bool rows_here = last_chunk && prefetch;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    // every consumer is past step g - 1: its weight slot, the halo buffer of the chunk before this one and (when step g - 1
                    // closed a tile) the staging area of that tile's epilogue are free
                    pc_wait_all_ge(cnt0 + PC_FREED, gstep);
                    // requests of this step, in queue order: parts of the next halo (taps 0-5: 2 2 2 2 2 1), the next tile's rows (tap 6 of a
                    // tile's last chunk), then the weight tile three steps ahead
#ifndef PC_ABL_NO_HALO
                    if (prefetch && tap < 6) {
                        if (tap < 5) { issue_halo_part(same_tile ? cur : nxt, 2 * tap, ncc, hnxt); issue_halo_part(same_tile ? cur : nxt, 2 * tap + 1, ncc, hnxt); }
                        else issue_halo_part(same_tile ? cur : nxt, 10, ncc, hnxt);
                    }
#endif
                    if (tap == 6 && rows_here) issue_rows(nxt, (k + 1) & 1);
                    const int wslot = slot + RING - 1 >= RING ? slot - 1 : slot + RING - 1;
#ifndef PC_ABL_NO_W
                    if (tap + RING - 1 < 9) issue_w(cur.tn, cc, tap + RING - 1, wslot);
                    else if (prefetch) issue_w(ntn, ncc, tap + RING - 1 - 9, wslot);
#endif
                    slot = slot + 1 == RING ? 0 : slot + 1;
                    // The next step's weight tile (requested two steps back, LAST in its step) has to be in; everything requested in the
                    // previous step and in this one is newer and may stay in flight.  When the next step opens a chunk, that chunk's halo
                    // (requested in taps 0-5) and — for a new tile — its rows (tap 6, ahead of the weights) are older than the tile waited for.
                    if (prefetch) {
                        if (tap == 0) wait_vm<10>();
                        else if (tap < 5) wait_vm<12>();
                        else if (tap == 5) wait_vm<11>();
                        else if (tap == 6) { if (rows_here) wait_vm<10>(); else wait_vm<9>(); }
                        else if (tap == 7) { if (rows_here) wait_vm<9>(); else wait_vm<8>(); }
                        else wait_vm<8>();
                    } else {
                        if (tap < 6) wait_vm<8>(); else if (tap == 6) wait_vm<4>(); else wait_vm<0>();
                    }
                    pc_signal(sig_ready);
                    ++gstep;
                }
                hb ^= 1;
            }
            cur = nxt;
        }
        return;
    }

    // =================================================================== consumers
    constexpr int NIB = 2, NJ = 4;
    const int wn = wave >> 1, wm = wave & 1;              // 64 channels x 128 pixels (patch rows 8 wm .. 8 wm + 7)
    if (a.flags & 1) __builtin_amdgcn_s_setprio(1);
    // Fragment addresses.  Pixel block j of the wave = patch rows 8 wm + 2 j, 2 j + 1: lane l reads pixel (2 j + (l % 32) / 16, l % 16), so the
    // four blocks sit at a CONSTANT 2 x 18 x 128 bytes from each other, their chunk key ((px + s) >> 1) is the same, and ONE address
    // register + ds_read's immediate offset serves all four; likewise the two 32-row weight blocks (4096 bytes apart).  The sub-step
    // term (kc << 5, bits 5-6) is XORed into that one register: the offsets are multiples of 512 and do not touch those bits.
    f32x16 acc[NIB][NJ];
    const int sw = (lane >> 5) ^ ((lane >> 1) & 7);
    const int hi = lane >> 5;
    constexpr int XJ = 2 * HWD * 128, WI = 32 * 128;      // byte distance of pixel blocks / weight blocks
    const unsigned wv0 = lds0 + 2 * HALO_BYTES + (unsigned)((wn * 64 + (lane & 31)) * 128 + (sw << 4));
    unsigned kx[3];                                        // halo address of tap (0, s) in buffer 0, s = 0 .. 2 (without the tap's row / buffer offset)
    {
        const int p = wm * 128 + (lane & 31), px = p & 15;
        const unsigned hv0 = lds0 + (unsigned)(((p >> 4) * HWD + px) * 128);
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) kx[s_] = hv0 + (unsigned)(s_ * 128) + (unsigned)((hi ^ (((px + s_) >> 1) & 7)) << 4);
    }
    // accumulators start from bias + time bias of the tile's channels (lane: channels 8 g + 4 hi + 0..3 of each 32-channel block, any pixel)
    auto init_acc = [&](int parity) {
        const char* rows = smem + ROWS_AT + parity * 2048;
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            f32x4v bsum[4], brow[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned ad = (unsigned)(size_t)(rows + (wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5)) * 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(bsum[g]) : "v"(ad) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(brow[g]) : "v"(ad) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4v sum = bsum[g] + brow[g];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { acc[i][j][4 * g] = sum.x; acc[i][j][4 * g + 1] = sum.y; acc[i][j][4 * g + 2] = sum.z; acc[i][j][4 * g + 3] = sum.w; }
            }
        }
    };
    // LDS addresses of a step's fragments (without the sub-step term): halo pixel of tap (r, s) in buffer hb_, weight rows in slot slot_
    auto halo_addr = [&](int tap_, int hb_) {
        const int r = tap_ / 3, s_ = tap_ - 3 * r;
        unsigned v = kx[s_] + (unsigned)(hb_ * HALO_BYTES + r * HWD * 128);
        asm volatile("" : "+v"(v));
        return v;
    };
    auto w_addr = [&](int slot_) { unsigned v = wv0 + (unsigned)(slot_ * WT_BYTES); asm volatile("" : "+v"(v)); return v; };
    u32x4 fw[2][NIB], fx[2][NJ];                           // fragment sets: K sub-step kc uses set kc & 1
    unsigned hk, wk, hkn = 0, wkn = 0;                     // this step's / the next step's addresses
    u32x4 rcp = {0u, 0u, 0u, 0u};                          // the loaders' `ready` counters as read in the previous step's third sub-step (written by the asm read itself: no copies)
    const unsigned sig_freed = lane == 0 ? cnt0 + PC_FREED + (unsigned)(wave * 4) : junk;
    const unsigned sig_edone = lane == 0 ? cnt0 + PC_EDONE : junk;
    TilePos cur = tile_pos(0);
#ifdef C3_TIMING
    int tstep = 0;
#endif
    PC_STAMP(0); PC_STAMP(1);
    unsigned need = 1;                                      // `ready` value (of every loader) that opens the step about to start
    pc_wait_all_ge(cnt0 + PC_READY, need);                 // the loaders' prologue: halo 0, rows 0 and weight tile 0 are in LDS
    PC_STAMP(2);
    init_acc(0);
    int slot = 0, hb = 0;
    hk = halo_addr(0, 0);
    wk = w_addr(0);
    bool pend = false;
    for (int k = 0; k < my_tiles; ++k) {
        for (int cc = 0; cc < nchunks; ++cc) {
            const bool last_chunk = cc + 1 == nchunks;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const bool tile_end = tap == 8 && last_chunk;
                const int nslot = slot + 1 == RING ? 0 : slot + 1;
                // One K sub-step = the eight MFMAs of fragment set `cw/cx` + the six reads of the next set `nw/nx`, two per gap behind the first
                // three MFMAs.  NO full wait anywhere: LDS returns a wave's reads in order, so each MFMA waits (counted lgkmcnt) for exactly
                // the fragments it is the first to use — a batch of six 1-KiB reads takes ~250 clk to come back, more than the 256 clk of a
                // sub-step's MFMAs leave when all six must be in before the first multiply (measured: 390 clk per sub-step that way).
                // Queue on entry: the six reads of cw/cx (issue order w0 x0 w1 x1 x2 x3), then `pre` newer operations (the counter read /
                // signal of the previous sub-step).  post: 1 = read the `ready` counter behind the new reads, 2 = signal `sig` there.
                // mf = false: reads only (first sub-step of a tile); kc < 0: MFMAs only (last sub-step of a tile).
                auto sub_step = [&](bool mf, const u32x4 (&cw)[NIB], const u32x4 (&cx)[NJ], int kc, u32x4 (&nw)[NIB], u32x4 (&nx)[NJ], int pre, int post, unsigned sig, bool nextaddr) {
                    unsigned aw = 0, ax = 0;
                    if (kc >= 0) { aw = wk ^ (unsigned)(kc << 5); ax = hk ^ (unsigned)(kc << 5); }
                    const int rd = kc >= 0 ? 1 : 0;            // new reads are issued in this sub-step
                    auto rd_w = [&](int i) {
                        if (kc < 0) return;
#ifdef PC_ABL_NO_READS
                        return;
#endif
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nw[i]) : "v"(aw), "n"(i * WI) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    auto rd_x = [&](int j) {
                        if (kc < 0) return;
#ifdef PC_ABL_NO_READS
                        return;
#endif
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nx[j]) : "v"(ax), "n"(j * XJ) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    auto mm = [&](int i, int j) {
                        if (!mf) return;
#ifdef PC_ABL_NO_MFMA
                        return;
#endif
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cw[i]), __builtin_bit_cast(bf16x8, cx[j]), acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    // wait until at most `n` LDS operations of this wave are outstanding (n is a constant after inlining)
                    auto wl = [&](int n) {
                        if (!mf) return;
                        switch (n) {
                            case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
                            case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
                            case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
                            case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
                            case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
                            case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
                            case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
                            case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
                            case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
                            default: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const int po = post ? 1 : 0;
                    wl(4 + pre);                             // w0, x0 are in (w1 x1 x2 x3 and the `pre` operations may be on their way)
                    mm(0, 0); rd_w(0); rd_x(0);
                    wl(3 + pre + 2 * rd);                    // w1
                    mm(1, 0); rd_w(1); rd_x(1);
                    wl(2 + pre + 4 * rd);                    // x1
                    mm(0, 1); rd_x(2); rd_x(3);
                    if (post == 1) {
#ifndef PC_ABL_NO_SYNC
                        asm volatile("ds_read_b128 %0, %1" : "=v"(rcp) : "v"(cnt0 + PC_READY) : "memory");
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (post == 2) {
                        pc_signal(sig);                      // (behind this step's last reads in the wave's LDS queue)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    mm(1, 1);
                    if (nextaddr) {
                        hkn = tap < 8 ? halo_addr(tap + 1, hb) : halo_addr(0, hb ^ 1);
                        wkn = w_addr(nslot);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    wl(1 + pre + 6 * rd + po);               // x2
                    mm(0, 2); mm(1, 2);
                    wl(pre + 6 * rd + po);                   // x3
                    mm(0, 3); mm(1, 3);
                };
                // sub-step 1: the previous step's last fragments (behind them in the queue: that step's signal) | reads kc = 0.  The `ready`
                // counter read in the previous step's third sub-step is older than those fragments: valid once the first wait is through,
                // and checked BEFORE this step's first read is issued
#ifndef PC_ABL_NO_SYNC
                if (tap > 0 || pend) {
                    asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (pc_min_dist(rcp, need) < 0) pc_wait_all_ge(cnt0 + PC_READY, need);
                }
#endif
                sub_step(tap > 0 || pend, fw[1], fx[1], 0, fw[0], fx[0], 1, 0, 0u, false);
                sub_step(true, fw[0], fx[0], 1, fw[1], fx[1], 0, 0, 0u, false);
                sub_step(true, fw[1], fx[1], 2, fw[0], fx[0], 0, 1, 0u, false);
                // (a tile's last step hands its slot back AFTER the epilogue, with the staging area; here it signals the consumers' rendezvous)
                sub_step(true, fw[0], fx[0], 3, fw[1], fx[1], 1, 2, tile_end ? sig_edone : sig_freed, !tile_end);
                slot = nslot;
                need += 1;
                if (tile_end) {
                    sub_step(true, fw[1], fx[1], -1, fw[0], fx[0], 1, 0, 0u, false);
                    pend = false;
                    if (k == 0) PC_STAMP(3); else if (k == 1) PC_STAMP(5);
                    pc_wait_ge(cnt0 + PC_EDONE, 4u * (unsigned)(k + 1));
                    break;
                }
                pend = true;
                hk = hkn; wk = wkn;
                // the fragment sets live in registers across the chunk loop's back edge: everything requested is in before it is taken (the
                // compiler may place copies there, and it does not know about the reads in flight)
                if (tap == 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                PC_STEP();
            }
            if (last_chunk) {
                // ---- tile done: out = acc, staged through the finished halo buffer in wave-private slices of
                // 64 pixels x 64 bytes so that four consecutive lanes store the 64 contiguous bytes of a pixel (as above)
                const int n0 = cur.tn * 128 + wn * 64;
                const bool live = n0 < a.N;
                const unsigned stg = lds0 + (unsigned)(hb * HALO_BYTES + wave * (64 * 64));
                // pixel q16 * 16 + lane / 4 of the wave's 128 = patch row wm * 8 + q16, column lane / 4: a wave-uniform 64-bit base per patch row
                // (scalar arithmetic) plus ONE 32-bit lane offset for the whole tile — the stores take the `saddr + voffset` form and the
                // epilogue carries no per-store vector multiplies (they were a sixth of its 3500 clk on a wave that has the SIMD's VALU to itself
                // only every other cycle)
                const unsigned long long ob =
                    (unsigned long long)(a.out + ((long long)(cur.img * a.H + cur.py0 + wm * 8) * a.W + cur.px0) * a.out_ld + n0);
                const unsigned ob_lo = __builtin_amdgcn_readfirstlane((unsigned)ob), ob_hi = __builtin_amdgcn_readfirstlane((unsigned)(ob >> 32));
                const unsigned long long obase = ((unsigned long long)ob_hi << 32) | ob_lo;
                const unsigned rowstride = (unsigned)__builtin_amdgcn_readfirstlane((int)(a.W * (int)a.out_ld * 2));     // bytes per patch row
                const unsigned loff = (unsigned)(((lane >> 2) * (int)a.out_ld + 8 * (lane & 3)) * 2);
                typedef __attribute__((address_space(1))) u32x4 gvec_t;      // (an integer-made pointer is generic: name the global space, or the stores go out as flat_store)
                auto orow = [&](int q16, int col) { return (gvec_t*)(obase + (unsigned long long)q16 * rowstride + (unsigned)(col * 2)) + 0; };
#pragma unroll
                for (int i = 0; i < NIB; ++i) {
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jh + jj;
                            uint2 pk[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                                pk[g].x = pack_bf2(v[0], v[1]); pk[g].y = pack_bf2(v[2], v[3]);
                            }
#pragma unroll
                            for (int q2 = 0; q2 < 2; ++q2) {
                                const u32x2 sx = __builtin_amdgcn_permlane32_swap(pk[2 * q2].x, pk[2 * q2 + 1].x, false, false);
                                const u32x2 sy = __builtin_amdgcn_permlane32_swap(pk[2 * q2].y, pk[2 * q2 + 1].y, false, false);
                                u32x4 o; o.x = sx.x; o.y = sy.x; o.z = sx.y; o.w = sy.y;
                                const int P = jj * 32 + (lane & 31), sl = (2 * q2 + hi) ^ ((P >> 2) & 3);
                                const unsigned ad = stg + (unsigned)(P * 64 + (sl << 4));
                                asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(o) : "memory");
                            }
                        }
                        u32x4 back[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int Q = rr * 16 + (lane >> 2), sl = (lane & 3) ^ ((Q >> 2) & 3);
                            const unsigned ad = stg + (unsigned)(Q * 64 + (sl << 4));
                            asm volatile("ds_read_b128 %0, %1" : "=v"(back[rr]) : "v"(ad) : "memory");     // (a wave's LDS operations execute in order)
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        if (live) {
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) *(gvec_t*)((__attribute__((address_space(1))) char*)orow(jh * 4 + rr, i * 32) + loff) = back[rr];
                        }
                    }
                }
                // the staging reads are done (waited for above): slot, halo buffer and staging area go back to the loaders
                pc_signal(sig_freed);
                if (k + 1 < my_tiles) {
                    cur = tile_pos(k + 1);
                    pc_wait_all_ge(cnt0 + PC_READY, need);     // the next tile's first weight tile, its halo and its rows are in
                    init_acc((k + 1) & 1);
                    hk = halo_addr(0, hb ^ 1);
                    wk = w_addr(slot);
                }
                PC_STEP();
                if (k == 0) PC_STAMP(4); else if (k == 1) PC_STAMP(6);
            }
            hb ^= 1;
        }
    }
    PC_STAMP(7);
}


}  // namespace

#ifdef C3_TRACE
extern "C" int ddpm_debug_set_c3_trace(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_c3_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif
#ifdef C3_TIMING
extern "C" int ddpm_debug_set_c3_timing(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_c3_timing), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

// Launcher behind ddpm_conv2d_nhwc (gemm.hip) for 3x3 / stride 1 / pad 1, bf16 -> bf16 (H, W = output image; upsample: x stored at H/2 x W/2): 16 x 16 patches for images of 16 x 16 and up with
// >= 16384 pixels, 8 x 8 patches for 8 x 8-divisible images with >= 4096 pixels.  -1: geometry / epilogue not covered (the caller keeps
// its other kernels), else a status code.  dry: decide only, and return the patch edge (16 / 8) that would run.
// Diagnostic: the record a timed-out semaphore wait of conv3x3_pc_kernel left before it trapped — out4 (HOST memory) receives
// {block, LDS address of the counter, value waited for, 0 = none | 1 = a loader / consumer counter | 2 = the consumers' rendezvous}.
extern "C" int ddpm_conv3x3_pc_last_fault(unsigned* out4) {
    if (!out4) return DDPM_ERR_NULL;
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_pc_fault), 4 * sizeof(unsigned)) == hipSuccess ? DDPM_OK : DDPM_ERR_LAUNCH;
}

int ddpm_conv3x3_stream_launch(const void* x, long long x_ld, const void* w, void* y, long long y_ld, const float* bias, const float* rowbias,
                               long long rowbias_ld, const void* residual, long long res_ld, int accumulate, int B, int H, int W, int C, int N,
                               int upsample, int xcd, int dry, void* stream) {
    static const bool off = getenv("DDPM_CONV_NO_STREAM3") != nullptr, off8 = getenv("DDPM_CONV_NO_STREAM3_8") != nullptr;
    const long long M = (long long)B * H * W;
    const int patch = (H % 16 == 0 && W % 16 == 0 && M >= 16384) ? 16 : ((H % 8 == 0 && W % 8 == 0 && M >= 4096 && !off8) ? 8 : 0);
    if (off || !patch || (residual && accumulate) || C % 64 || N % 64 || x_ld % 8 || y_ld % 8 || (residual && res_ld % 4) || (rowbias && rowbias_ld % 4)) return -1;
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (residual && (((uintptr_t)residual) & 7)) || (bias && !aligned16(bias)) || (rowbias && !aligned16(rowbias))) return -1;
    if (upsample && ((H | W) & 1)) return -1;
    const long long xbytes = ((long long)B * (H >> upsample) * (W >> upsample) * x_ld - (x_ld - C)) * 2, wbytes = (long long)N * 9 * C * 2;
    const long long rbbytes = rowbias ? ((long long)(B - 1) * rowbias_ld + N) * 4 : 0;
    if (xbytes > 0x7ffffff0ll || wbytes > 0x7ffffff0ll || rbbytes > 0x7ffffff0ll) return -1;
    // 16 x 16 patches: the wave-specialised kernel (loaders + consumers) unless DDPM_CONV_NO_PC is set; DDPM_C3_PC_FLAGS: bit 0 = consumers at priority 1
#ifdef PC_DEFAULT_OFF
    static const bool use_pc = false;                       // (scripts/build_variant.sh prev: the round-3 kernel for same-process A/Bs)
#else
    static const bool use_pc = getenv("DDPM_CONV_NO_PC") == nullptr;
#endif
    static const int pc_flags = getenv("DDPM_C3_PC_FLAGS") ? atoi(getenv("DDPM_C3_PC_FLAGS")) : 0;
    // (a residual / "+=" epilogue stays on the kernel above: its rows have to be requested a K-step ahead, into registers the
    //  wave-specialised kernel does not have — acc 128 + fragments 48 — and requested late they cost 5-12 %; the two kernels give
    //  bit-identical results, so the choice is invisible)
    if (dry) return patch == 16 && use_pc && !residual && !accumulate ? 17 : patch;
    CsArgs a; memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x; a.x_ld = x_ld; a.x_extent = (unsigned)xbytes;
    a.w = (const bf16_t*)w; a.w_extent = (unsigned)wbytes;
    a.out = (bf16_t*)y; a.out_ld = y_ld;
    a.bias = bias; a.rowbias = rowbias; a.rowbias_ld = rowbias_ld; a.rowbias_extent = (unsigned)rbbytes;
    a.res = (const bf16_t*)residual; a.res_ld = res_ld; a.accumulate = accumulate;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.K = 9 * C; a.ups = upsample ? 1 : 0;
    a.tiles_y = H / patch; a.tiles_x = W / patch; a.tiles_n = (N + 127) / 128;
    a.total_tiles = B * a.tiles_y * a.tiles_x * a.tiles_n;
    a.xcd = xcd;
    a.d_tiles_n = make_fastdiv((unsigned)a.tiles_n); a.d_tpi = make_fastdiv((unsigned)(a.tiles_y * a.tiles_x)); a.d_tiles_x = make_fastdiv((unsigned)a.tiles_x);
    static const int max_grid = getenv("DDPM_C3_GRID") ? atoi(getenv("DDPM_C3_GRID")) : 256;      // (timing experiments: > 256 = fewer tiles per block)
    const int cus = max_grid == 256 ? ddpm_cu_budget(256) : max_grid;                             // (256 - DDPM_DP_RESERVED_CUS when a collective runs beside the step)
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
#define C3_LAUNCH(PATCHV)                                                                                                              \
    do {                                                                                                                               \
        static DevOnce attr_set;                                                                                                  \
        if (!attr_set) {                                                                                                               \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream_kernel<PATCHV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    Lds<PATCHV>::BYTES) != hipSuccess) return DDPM_ERR_LAUNCH;                                         \
            attr_set = true;                                                                                                           \
        }                                                                                                                              \
        hipLaunchKernelGGL(conv3x3_stream_kernel<PATCHV>, dim3(grid), dim3(512), Lds<PATCHV>::BYTES, (hipStream_t)stream, a);           \
    } while (0)
    if (patch == 16 && use_pc && !residual && !accumulate) {
        a.flags = pc_flags;
        static DevOnce pc_attr_set;
        if (!pc_attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PC_BYTES) != hipSuccess) return DDPM_ERR_LAUNCH;
            pc_attr_set = true;
        }
        hipLaunchKernelGGL(conv3x3_pc_kernel, dim3(grid), dim3(512), PC_BYTES, (hipStream_t)stream, a);
    }
    else if (patch == 16) C3_LAUNCH(16); else C3_LAUNCH(8);
#undef C3_LAUNCH
    return check_launch();
}
