// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions (bf16) with STATIONARY PATCHES — the wgrad counterpart of
// conv3x3_halo_kernel (gfx950).  Reference semantics: convolution_backward w.r.t. the weight (and the bias) of F.conv2d at
// ddpm_torch/modules.py:120-123, sites models/unet.py:76,79,127,141,201:
//     dw[n][r][s][c] += sum_{b,y,x} dy[b,y,x,n] * x[b, y+r-1, x+s-1, c]          db[n] += sum_{b,y,x} dy[b,y,x,n]
//
// Why not the generic transposed-operand GEMM (gemm_kernel<T, true, true>): as a GEMM this product is M = Cout (128..512),
// N = 9*Cin, K = B*H*W pixels — a tiny output with a huge reduction.  The generic kernel needs 128x128 output tiles to
// amortise its operand traffic (one CU moves <= ~60 B/clk from L2 into LDS), which leaves 9..72 tiles for 256 CUs; the
// reduction is therefore cut into 14..56 slices and every slice ends in 16 K fp32 atomics — measured 25-30 % of the kernel —
// while the im2col operand is fetched 9x (once per tap).
//
// Here a block owns a SMALL output tile, 64 out-channels x 32 in-channels x all 9 taps (72 KiB of fp32), and walks the pixels
// in stages of one patch (16x16, or several smaller images):
//   * per stage the dy patch (256 px x 64 n) and the x HALO of the patch ((16+2)^2 px x 32 c) are DMA'd to LDS once and serve
//     all nine taps — tap (r,s) reads the same halo image shifted by r*(PW+2)+s pixels.  L2 -> LDS bytes per FLOP are ~1/4 of the
//     generic kernel's although the tile is 8x smaller;
//   * both operands are pixel-major in memory (k = pixel is the slow index), so fragments come from ds_read_b64_tr_b16;
//   * the 8 waves are 2 (n halves) x 4 (k groups): the waves of a k group take every 4th 16-pixel k-step of the stage and keep
//     their own 9 accumulators (32x32 fp32 each); they are summed through LDS once, at the end of the block's pixel range;
//   * with small tiles there are 32..256 output tiles per layer: the pixel range is cut into FEW slices (2..16), so the
//     cross-block reduction is small — each slice STORES its partial tile into its own slab copy and ddpm_wgrad_reduce sums
//     the copies in a fixed order (bit-deterministic gradients, no atomics), or adds with atomics when asked to;
//   * the bias gradient (column sums of dy) falls out of the dy fragments the waves read anyway: the blocks of in-channel
//     tile 0 accumulate it — the separate column-sum launch over dy disappears.
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr unsigned OOB = 0x7ffffff0u;
constexpr int TN = 64, TC = 32;                 // output tile: out-channels x in-channels (x 9 taps)
constexpr int DY_ROW = TN * 2, X_ROW = TC * 2;  // LDS row bytes: one pixel of the dy tile / of the halo

struct Wg3Args {
    const void* dy; long long dy_ld; unsigned dy_extent;
    const void* x; long long x_ld; unsigned x_extent;
    int ups;                                               // 1: x is stored at H/2 x W/2 (nearest 2x up-sampling folded into the halo gather)
    float* dw; long long slab_stride; float* dbias; long long bias_stride;
    int B, H, W, C, N, Nreal;
    int PH, NB, lPP;                 // stage geometry (PW is a template parameter): NB * PH * PW == stage pixels
    int tiles_y, tiles_x;            // patches per image
    int stages, stages_per_split;
    int tiles_n, tiles_c;
    int atomic;                      // 1: atomicAdd into dw / dbias; 0: plain stores into slab copy `split`
    FastDiv d_tiles, d_tiles_c, d_tpi, d_tiles_x, d_halo_img, d_halo_w;
};

__device__ __forceinline__ int xcd_logical(int id, int total) {
    const int per = total >> 3;
    return id < (per << 3) ? (id & 7) * per + (id >> 3) : id;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef WG_TIMING
// debug builds only (scripts/wg_timeline.py): [block][wave][16] clock64 stamps (low 32 bits, scalar registers) of ONE mid-kernel stage of
// the 16 x 16-patch kernel — taken right behind waits that leave nothing outstanding, so reading the clock does not disturb the LGKM
// bookkeeping — plus loop begin / end and the stage count
__device__ unsigned long long* g_wg_timing = nullptr;
#define WG_T(k) do { if (tl) tq[k] = (unsigned)clock64(); } while (0)
#else
#define WG_T(k)
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// hardware transpose read: the 16-lane group's lanes 4r..4r+3 address 4 k-rows x (4 x 8 bytes); lane c receives, for its
// column c of the 16 columns, the 4 k values (see read_frag<T, true> in gemm.hip)
__device__ __forceinline__ uint2 tr_read(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}

// STAGE_PX: pixels per stage (256: one 16x16 patch; 128: 2 images of 8x8 / 8 images of 4x4).  PW: patch width (16 / 8 / 4).
// RING: LDS stages in flight + 1.  NI_DY / NI_X: DMA instructions per thread per stage for the dy tile / the halo.
// Wave w = (k group w >> 1, out-channel half w & 1).  16-wide patches: k group g owns patch rows 4g .. 4g+3 of the stage;
// smaller images: k-steps g, g + 4, ...
template <int STAGE_PX, int PW, int RING>
__global__ __launch_bounds__(512, 2)
void wgrad3x3_kernel(Wg3Args a) {
    constexpr int KSTEPS = STAGE_PX / 16;                  // 16-pixel k-steps per stage
    constexpr int KPW = KSTEPS / 4;                        // k-steps per wave per stage
    constexpr int HW = PW + 2;                             // halo width (pixels)
    constexpr int NI_DY = STAGE_PX * 8 / 512;              // dy tile: STAGE_PX rows x 8 chunks of 16 B
    constexpr int HALO_MAX = STAGE_PX == 256 ? 324 : (PW == 8 ? 200 : 288);     // NB * (PH+2) * (PW+2)
    constexpr int NI_X = (HALO_MAX * 4 + 511) / 512;       // halo: HP rows x 4 chunks of 16 B
    constexpr int DY_BYTES = STAGE_PX * DY_ROW;
    // halo region: whole DMA instructions (512 vectors = 128 rows each) — except for the 16x16 patch, where 3 x 56 KiB would not
    // fit: there the region ends after the last wave that still holds valid rows (336 rows) and the waves of the third
    // instruction that lie completely beyond it (their lanes fetch nothing: out-of-range offsets) write their zeros to a 1 KiB dump
    constexpr int X_BYTES = STAGE_PX == 256 ? 336 * X_ROW : NI_X * 128 * X_ROW;
    constexpr int STAGE_BYTES = DY_BYTES + X_BYTES;
    constexpr int DUMP_OFF = RING * STAGE_BYTES;
    constexpr int PER = NI_DY + NI_X;                      // DMA instructions per wave per stage (for the counted waits)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wk = wave >> 1;

    const int ntiles = a.tiles_n * a.tiles_c;
    const int lid = xcd_logical(blockIdx.x, gridDim.x);
    const int split = (int)fdiv((unsigned)lid, a.d_tiles), tile = lid - split * ntiles;
    const int tn = (int)fdiv((unsigned)tile, a.d_tiles_c), tc = tile - tn * a.tiles_c;
    const int st_begin = split * a.stages_per_split, st_end = min(a.stages, st_begin + a.stages_per_split);
    const int nst = max(st_end - st_begin, 0);
    const int PP = 1 << a.lPP, HH = a.PH + 2, HP = a.NB * HH * HW, tpi = a.tiles_y * a.tiles_x;

    auto rsrc_of = [&](const void* p, unsigned extent) {
        const unsigned long long ad = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rdy = rsrc_of(a.dy, a.dy_extent), rx = rsrc_of(a.x, a.x_extent);

    // ---- DMA plans.  dy tile: vector v = tid + 512 i -> stage pixel p = v >> 3, PHYSICAL chunk v & 7 of its 128-byte row; the
    // logical chunk is XOR-swizzled by ((p >> 1) & 1) << 2 on the source side so that the four k-rows one transpose read touches
    // (p, p+1, p+2, p+3) fall on four different 32-byte bank groups.  Halo: v -> halo pixel v >> 2, chunk v & 3 (64-byte rows:
    // four consecutive rows already cover the 64 banks exactly once).
    // (vector i of a thread is 64 stage pixels / 128 halo pixels after vector i-1: the plans keep one base per thread)
    unsigned dy_rel0, dy_delta; int dy_il0, dy_il_delta;
    {
        const int p = tid >> 3;
        const int lc = (tid & 7) ^ (((p >> 1) & 1) << 2);
        const int il = p >> a.lPP, q = p & (PP - 1), py = q / PW, px = q - py * PW;
        const int n = tn * TN + lc * 8;
        dy_il0 = n < a.N ? il : (1 << 20);                                     // channel block outside dy: always out of range
        dy_rel0 = (unsigned)((((long long)(il * a.H + py) * a.W + px) * a.dy_ld + n) * 2);
        // 64 stage pixels further: whole images (small images) or 64 / PW patch rows of the same image
        dy_il_delta = 64 >> a.lPP;
        dy_delta = (unsigned)((dy_il_delta ? (long long)dy_il_delta * a.H * a.W : (long long)(64 / PW) * a.W) * a.dy_ld * 2);
    }
    unsigned x_pos[NI_X], x_rel[NI_X];                                         // hy << 16 | hx << 8 | il (halo coordinates; image row = py0 + hy - 1) ; 0xffffffff: no pixel
    const unsigned x_c2 = (unsigned)((tc * TC + (tid & 3) * 8) * 2);
#pragma unroll
    for (int i = 0; i < NI_X; ++i) {
        const int hp = (tid >> 2) + 128 * i;
        const int il = (int)fdiv((unsigned)hp, a.d_halo_img), rem = hp - il * (HH * HW);
        const int hy = (int)fdiv((unsigned)rem, a.d_halo_w), hx = rem - hy * HW;
        x_pos[i] = hp < HP ? ((unsigned)hy << 16 | (unsigned)hx << 8 | (unsigned)il) : 0xffffffffu;
        // byte offset of this halo pixel RELATIVE to the patch origin's (img0, py0, px0 are multiples of the patch size: even, so the
        // nearest-2x gather's shifts distribute): everything a stage adds to it is wave-uniform — no 64-bit address product per vector
        // and stage.  Wraps modulo 2^32 for the rows above / left of the image, which are masked anyway.
        const int dyi = (hy - 1) >> a.ups, dxi = (hx - 1) >> a.ups;
        x_rel[i] = (unsigned)(((il * (a.H >> a.ups) + dyi) * (a.W >> a.ups) + dxi) * (int)a.x_ld * 2) + x_c2;
    }
    auto issue_stage = [&](int st, int slot) {
        const int grp = (int)fdiv((unsigned)st, a.d_tpi), pt = st - grp * tpi;
        const int ty = (int)fdiv((unsigned)pt, a.d_tiles_x), tx = pt - ty * a.tiles_x;
        const int img0 = grp * a.NB, py0 = ty * a.PH, px0 = tx * PW;
        char* dst = smem + slot * STAGE_BYTES;
        const unsigned base = (unsigned)((((long long)(img0 * a.H + py0) * a.W + px0) * a.dy_ld) * 2);
        const unsigned base_x = (unsigned)((((long long)(img0 * (a.H >> a.ups) + (py0 >> a.ups)) * (a.W >> a.ups) + (px0 >> a.ups)) * a.x_ld) * 2);   // wave-uniform
#pragma unroll
        for (int i = 0; i < NI_DY; ++i) {
            const unsigned o = (img0 + dy_il0 + i * dy_il_delta < a.B) ? base + dy_rel0 + i * dy_delta : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)(dst + (wave * 64 + 512 * i) * 16), 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI_X; ++i) {
            const unsigned xp = x_pos[i];
            const int gi = img0 + (int)(xp & 0xff), iy = py0 + (int)(xp >> 16) - 1, ix = px0 + (int)((xp >> 8) & 0xff) - 1;
            const bool ok = xp != 0xffffffffu && gi < a.B && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned o = ok ? base_x + x_rel[i] : OOB;
            const int rel = (wave * 64 + 512 * i) * 16;
            char* to = (rel + 1024 <= X_BYTES) ? dst + DY_BYTES + rel : smem + DUMP_OFF;       // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)to, 16, o, 0, 0, 0);
        }
    };
    auto wait_inflight = [&](int stages_newer) {      // stages issued after the one that must have landed may stay in flight
        if constexpr (RING >= 4) { if (stages_newer >= 2) { wait_vm<2 * PER>(); return; } }
        if (stages_newer >= 1) wait_vm<PER>(); else wait_vm<0>();
    };

    // ---- per-lane fragment addressing (constant over the whole kernel)
    const int i16 = lane & 15, q4 = i16 >> 2, h = lane >> 5;
    const int mcol = ((lane >> 4) & 1) * 16 + (i16 & 3) * 4;              // 0..31: the 4-element m chunk this lane ADDRESSES
    // dy fragment (32 n of this wave's half x 16 px): row = kstep*16 + 8h + q4 (+4), chunk swizzled by the row's bit 1
    const int a_m = wn * 32 + mcol;
    const int a_row = 8 * h + q4;
    const int a_off = a_row * DY_ROW + (((a_m >> 3) ^ (((a_row >> 1) & 1) << 2)) << 4) + (a_m & 7) * 2;     // + kstep * 16 * DY_ROW ; hi: + 4 * DY_ROW
    // halo fragment (32 c x 16 px of tap (r,s)): pixel k = 8h + q4 (+4) of the k-step -> (patch row k / PW, column k % PW)
    const int k_lo = 8 * h + q4, k_hi = k_lo + 4;
    const int b_lo = ((k_lo / PW) * HW + (k_lo % PW)) * X_ROW + (mcol >> 3) * 16 + (mcol & 7) * 2;
    const int b_hi = ((k_hi / PW) * HW + (k_hi % PW)) * X_ROW + (mcol >> 3) * 16 + (mcol & 7) * 2;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);
    float bsum = 0.f;                                    // column sum of dy for n = wn*32 + (lane & 31), this lane's k half
    const bool want_bias = a.dbias != nullptr && tc == 0;

    // The transpose reads are INLINE ASM on purpose: hipcc treats the ds_read_tr builtin as a possible alias of every pending
    // LDS-DMA and puts `s_waitcnt vmcnt(0)` in front of it, which drains the whole stage ring on every iteration (measured: the
    // first version of this kernel spent 4.5 us per stage against 1.1 us of MFMA work).  With asm reads the LGKM counter is
    // ours to manage: a group of reads is issued one MFMA group ahead of its use and retired by `s_waitcnt lgkmcnt(0)` tied to
    // the destination registers (in/out operands), so no MFMA can be scheduled above the wait that covers its operands.
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct Frag { uint2 lo, hi; };
    auto frag4 = [](const Frag& f) { return __builtin_bit_cast(bf16x8, u32x4{f.lo.x, f.lo.y, f.hi.x, f.hi.y}); };
    auto bias_add = [&](const Frag& f) {
        bsum += __uint_as_float(f.lo.x << 16) + __uint_as_float(f.lo.x & 0xffff0000u) + __uint_as_float(f.lo.y << 16) + __uint_as_float(f.lo.y & 0xffff0000u)
              + __uint_as_float(f.hi.x << 16) + __uint_as_float(f.hi.x & 0xffff0000u) + __uint_as_float(f.hi.y << 16) + __uint_as_float(f.hi.y & 0xffff0000u);
    };
#define WAIT3(F) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0].lo), "+v"(F[0].hi), "+v"(F[1].lo), "+v"(F[1].hi), "+v"(F[2].lo), "+v"(F[2].hi) :: "memory")

    if constexpr (PW == 16) {
        // ---- 16-wide patches: a k-step is one patch row.  The wave owns patch rows 4 wk .. 4 wk + 3 and walks the SIX halo rows they
        // touch: halo row hr feeds tap row r of patch row hr - r, so its three shifted fragments (s = 0, 1, 2) are read once and used
        // by up to nine MFMAs — 18 + 4 fragment reads per stage instead of 36 + 4.  Order of the halo rows inside a stage:
        //     4 | 5 + 0 | 1 | 2 | -- barrier -- | 3
        // every group of fragment reads is issued under the MFMAs of the group before it, and the LAST group (nine MFMAs) runs after
        // the barrier that releases the next stage, covering that stage's first reads and the DMA issue for the slot just freed.
        auto bases = [&](int slot, unsigned& ab, unsigned& bl, unsigned& bh) {
            const unsigned stage0 = lds0 + slot * STAGE_BYTES;
            ab = stage0 + (4 * wk) * 16 * DY_ROW + a_off;
            bl = stage0 + DY_BYTES + (4 * wk) * HW * X_ROW + b_lo;
            bh = stage0 + DY_BYTES + (4 * wk) * HW * X_ROW + b_hi;
        };
#define READ_ROW(F, hr) do { _Pragma("unroll") for (int sft = 0; sft < 3; ++sft) { TR_READ(F[sft].lo, blo0, ((hr) * HW + sft) * X_ROW); TR_READ(F[sft].hi, bhi0, ((hr) * HW + sft) * X_ROW); } } while (0)
#define READ_A(FA) do { _Pragma("unroll") for (int j = 0; j < 4; ++j) { TR_READ(FA[j].lo, abase, j * 16 * DY_ROW); TR_READ(FA[j].hi, abase, j * 16 * DY_ROW + 4 * DY_ROW); } } while (0)
#define WAIT_A(FA) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA[0].lo), "+v"(FA[0].hi), "+v"(FA[1].lo), "+v"(FA[1].hi), "+v"(FA[2].lo), "+v"(FA[2].hi), "+v"(FA[3].lo), "+v"(FA[3].hi) :: "memory")
#define MMA_ROW(FA, F, hr) do { _Pragma("unroll") for (int r = 0; r < 3; ++r) { const int j = (hr) - r; if (j >= 0 && j < 4) { _Pragma("unroll") for (int sft = 0; sft < 3; ++sft) \
            acc[r * 3 + sft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(FA[j]), frag4(F[sft]), acc[r * 3 + sft], 0, 0, 0); } } } while (0)
#pragma unroll
        for (int t = 0; t < RING; ++t)
            if (t < nst) issue_stage(st_begin + t, t);
        if (nst >= 3) wait_vm<2 * PER>(); else if (nst == 2) wait_vm<PER>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        unsigned abase, blo0, bhi0;
        Frag fa[4], f4[3], f50[6], f1[3], f2[3], f3[3];
        if (nst > 0) {
            bases(0, abase, blo0, bhi0);
            READ_A(fa); READ_ROW(f4, 4);
            WAIT_A(fa); WAIT3(f4);
        }
#ifdef WG_TIMING
        unsigned tq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) tq[q] = 0;
        const unsigned loop_t0 = (unsigned)clock64();
#endif
        for (int it = 0; it < nst; ++it) {
#ifdef WG_TIMING
            const bool tl = g_wg_timing != nullptr && it == nst / 2;
#endif
            WG_T(0);
            if (want_bias) { bias_add(fa[0]); bias_add(fa[1]); bias_add(fa[2]); bias_add(fa[3]); }
            Frag* f5 = f50; Frag* f0 = f50 + 3;
            READ_ROW(f5, 5); READ_ROW(f0, 0);
            MMA_ROW(fa, f4, 4);
            WG_T(1);
            WAIT3(f5); WAIT3(f0);
            WG_T(2);
            READ_ROW(f1, 1);
            MMA_ROW(fa, f5, 5); MMA_ROW(fa, f0, 0);
            WG_T(3);
            WAIT3(f1);
            WG_T(4);
            READ_ROW(f2, 2);
            MMA_ROW(fa, f1, 1);
            WG_T(5);
            WAIT3(f2);
            WG_T(6);
            READ_ROW(f3, 3);
            MMA_ROW(fa, f2, 2);
            WG_T(7);
            WAIT3(f3);
            WG_T(8);
            // every LDS read of this stage has returned: after the barrier its slot may be refilled.  Stage it+1 must have landed
            // (one newer stage, it+2, may stay in flight).
            const bool more = it + 1 < nst;
            // (waves 0-3 feed stage it + 2 HERE, see below: its slot was freed by the barrier that ended iteration it - 1)
            if (wave < 4 && it >= 1 && it + 2 < nst) issue_stage(st_begin + it + 2, (it + 2) % RING);
            if (more) { if (it + 2 < nst) wait_vm<PER>(); else wait_vm<0>(); }
            WG_T(9);
            __builtin_amdgcn_s_barrier();
            WG_T(10);
            Frag na[4], n4[3];
            if (more) {
                bases((it + 1) % RING, abase, blo0, bhi0);
                READ_A(na); READ_ROW(n4, 4);
            }
            // Feeding the ring.  scripts/wg_timeline.py (round 4): an LDS-DMA instruction blocks its wave for ~120 clk, a stage's seven for
            // 800-1000 clk; with all eight waves issuing here, together, that was a fifth of a 4600-clk stage with the matrix pipe idle.  The
            // two waves of a SIMD do not multiply side by side either: the older one (waves 0-3) gets the pipe for its four MFMA groups, then
            // idles ~950 clk at the stage barrier while its partner (waves 4-7) runs its own.  So each wave now issues where it would idle:
            // waves 4-7 HERE, right behind the barrier, while waves 0-3 multiply; waves 0-3 at the END of their iteration (above), while
            // waves 4-7 multiply — stage s is fed during iteration s - 2 by both, and every wave still waits for its own seven.
            // (Measured worse: two instructions at a time behind the MFMA groups, +3.5 % time; the whole schedule on waves 4-7 — a wave gets
            // ~1 instruction per 330 clk once ~14 are in flight — 7100 clk per stage.)
            if (wave >= 4 && it + RING < nst) issue_stage(st_begin + it + RING, it % RING);
            __builtin_amdgcn_sched_barrier(0);
            MMA_ROW(fa, f3, 3);
            WG_T(11);
            WG_T(12);
            if (more) {
                WAIT_A(na); WAIT3(n4);
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[j] = na[j];
#pragma unroll
                for (int sft = 0; sft < 3; ++sft) f4[sft] = n4[sft];
            }
            WG_T(15);
        }
#ifdef WG_TIMING
        if (g_wg_timing && lane == 0) {
            unsigned long long* o = g_wg_timing + ((long long)blockIdx.x * 8 + wave) * 16;
            tq[13] = loop_t0; tq[14] = (unsigned)clock64();
#pragma unroll
            for (int q = 0; q < 16; ++q) o[q] = tq[q];
            o[14] |= (unsigned long long)nst << 32;
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#undef READ_ROW
#undef READ_A
#undef WAIT_A
#undef MMA_ROW
    } else {
    // ---- small images.  Prologue: RING-1 stages in flight, wait for the first
#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
        if (t < nst) issue_stage(st_begin + t, t);
    wait_inflight(min(RING - 2, nst - 1));
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < nst; ++it) {
        const int slot = it % RING;
        // the stage RING-1 ahead goes into the slot read in iteration it-1 (every wave is past the barrier that ended it)
        if (it + RING - 1 < nst) issue_stage(st_begin + it + RING - 1, (it + RING - 1) % RING);
        const unsigned stage0 = lds0 + slot * STAGE_BYTES;
        {
            // a k-step covers 16 / PW patch rows; per k-step one dy fragment and nine halo fragments, read in groups of three
            // (one tap row) one group ahead of the MFMAs that use them
            Frag fa[KPW], fb[2][3];
            unsigned blo[KPW], bhi[KPW];
#pragma unroll
            for (int u = 0; u < KPW; ++u) {
                const int ks = wk + 4 * u;
                const unsigned pa = stage0 + ks * 16 * DY_ROW + a_off;
                TR_READ(fa[u].lo, pa, 0); TR_READ(fa[u].hi, pa, 4 * DY_ROW);
                const int p0 = ks * 16, il = p0 >> a.lPP, pr0 = (p0 & (PP - 1)) / PW;
                blo[u] = stage0 + DY_BYTES + ((il * HH + pr0) * HW) * X_ROW + b_lo;
                bhi[u] = stage0 + DY_BYTES + ((il * HH + pr0) * HW) * X_ROW + b_hi;
            }
#pragma unroll
            for (int sft = 0; sft < 3; ++sft) { TR_READ(fb[0][sft].lo, blo[0], sft * X_ROW); TR_READ(fb[0][sft].hi, bhi[0], sft * X_ROW); }
            if constexpr (KPW == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0].lo), "+v"(fa[0].hi), "+v"(fa[1].lo), "+v"(fa[1].hi) :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0].lo), "+v"(fa[0].hi) :: "memory");
            WAIT3(fb[0]);
            if (want_bias) {
#pragma unroll
                for (int u = 0; u < KPW; ++u) bias_add(fa[u]);
            }
#pragma unroll
            for (int g = 0; g < 3 * KPW; ++g) {                            // group g = (k-step u, tap row r)
                const int u = g / 3, r = g - 3 * u;
                auto& cur = fb[g & 1];
                auto& nxt = fb[(g + 1) & 1];
                if (g + 1 < 3 * KPW) {
                    const int u2 = (g + 1) / 3, r2 = (g + 1) - 3 * u2;
#pragma unroll
                    for (int sft = 0; sft < 3; ++sft) {
                        TR_READ(nxt[sft].lo, blo[u2], (r2 * HW + sft) * X_ROW);
                        TR_READ(nxt[sft].hi, bhi[u2], (r2 * HW + sft) * X_ROW);
                    }
                }
#pragma unroll
                for (int sft = 0; sft < 3; ++sft)
                    acc[r * 3 + sft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(fa[u]), frag4(cur[sft]), acc[r * 3 + sft], 0, 0, 0);
                if (g + 1 < 3 * KPW) WAIT3(nxt);
            }
        }
        // stage it+1 must have landed before anyone reads it; newer stages may stay in flight across the barrier
        wait_inflight(min(RING - 2, max(nst - 2 - it, 0)));
        __builtin_amdgcn_s_barrier();
    }
    }
#undef TR_READ
#undef WAIT3

    // ---- sum the four k groups through LDS (the ring is idle now) as a two-level tree, then hand the finished tile to ALL eight
    // waves for the write: [n][tap][c] rows of 32 floats leave as 16-byte vectors (one 128-byte line per 8 lanes).
    float* red = reinterpret_cast<float*>(smem);           // [slot][wn][9 taps][4][64 lanes][4] : one slot = 72 KiB
    constexpr int SLOT_F = 2 * 9 * 16 * 64;
    float* rbias = red + 2 * SLOT_F;                       // [4 k groups][wn][64 lanes]
    auto put = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<f32x4*>(red + slot * SLOT_F + ((wn * 9 + t) * 4 + r4) * 256 + lane * 4) =
                    f32x4{acc[t][4 * r4], acc[t][4 * r4 + 1], acc[t][4 * r4 + 2], acc[t][4 * r4 + 3]};
    };
    auto take = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(red + slot * SLOT_F + ((wn * 9 + t) * 4 + r4) * 256 + lane * 4);
                acc[t][4 * r4] += v.x; acc[t][4 * r4 + 1] += v.y; acc[t][4 * r4 + 2] += v.z; acc[t][4 * r4 + 3] += v.w;
            }
    };
    if (want_bias) rbias[(wk * 2 + wn) * 64 + lane] = bsum;
    if (wk >= 2) put(wk - 2);                              // groups 2, 3 -> slots 0, 1
    __syncthreads();
    if (wk < 2) take(wk);                                  // group 0 += group 2, group 1 += group 3
    __syncthreads();
    if (wk == 1) put(0);
    __syncthreads();
    // group 0 finishes the sum and lays the tile out as the output wants it: tile[n (64)][tap (9)][c (32)] fp32 in slot 1
    float* otile = red + SLOT_F;
    if (wk == 0) {
        take(0);
        const int cc = lane & 31, nb = wn * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) otile[((nb + (r & 3) + 8 * (r >> 2)) * 9 + t) * TC + cc] = acc[t][r];
    }
    __syncthreads();
    float* out = a.dw + (a.atomic ? 0 : (long long)split * a.slab_stride);
    for (int v = tid; v < TN * 9 * (TC / 4); v += 512) {   // 4608 vectors of 4 floats: row = (n, tap), 8 vectors per row
        const int row = v >> 3, c4 = (v & 7) * 4;
        const int nl = row / 9, t = row - nl * 9, n = tn * TN + nl;
        if (n >= a.Nreal) continue;
        const f32x4 val = *reinterpret_cast<const f32x4*>(otile + row * TC + c4);
        float* o = out + ((long long)n * 9 + t) * a.C + tc * TC + c4;
        if (a.atomic) { atomicAdd(o, val.x); atomicAdd(o + 1, val.y); atomicAdd(o + 2, val.z); atomicAdd(o + 3, val.w); }
        else *reinterpret_cast<f32x4*>(o) = val;
    }
    if (want_bias && tid < TN) {
        const int wnn = tid >> 5, l = tid & 31, n = tn * TN + tid;
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) tot += rbias[(g * 2 + wnn) * 64 + l] + rbias[(g * 2 + wnn) * 64 + l + 32];     // fixed order
        if (n < a.Nreal) {
            if (a.atomic) atomicAdd(a.dbias + n, tot);
            else a.dbias[(long long)split * a.bias_stride + n] = tot;
        }
    }
}


// =====================================================================================================================================
// Wave-specialised form for the 16 x 16 patches (round 6).  The kernel above multiplies on eight waves that also issue the stage's
// LDS-DMA and meet at a block barrier every stage: scripts/wg_timeline.py priced a stage at 4200-4600 clk for 2304 clk of MFMA — 800-1000
// clk of DMA issue with the matrix pipe idle and ~950 clk of the older wave of every SIMD waiting at the barrier.  Here, as in
// conv3x3_pc_kernel:
//   * waves 0-3 (one per SIMD) are CONSUMERS: wave (wn, wc) owns 32 out-channels x 32 in-channels x 9 taps of a 64 x 64 x 9 output tile
//     (nine 32 x 32 accumulators, as before) and walks ALL sixteen patch rows of a stage — no k split inside the block, no final sum
//     over wave groups; per patch row one dy fragment and ONE new halo row (three shifted fragments, each used by three patch rows):
//     8 transpose reads feed 9 MFMAs, issued one patch row ahead;
//   * waves 4-7 are LOADERS: they own the whole DMA schedule (dy tile 32 KiB + two 21-KiB channel-half planes of the halo per stage,
//     ring of two stages) and their own vmcnt queue;
//   * no s_barrier in the loop: ready[4] (one per loader: stages completely landed) and freed[4] (one per consumer: stages completely
//     read) are monotone counters in LDS, each with a single writer; a check is one ds_read_b128.
// A 64-wide in-channel tile halves the L2 -> LDS bytes per FLOP (74 KiB per 4608 clk of MFMA against 53 KiB per 2304).
// Covers C % 64 == 0 and N % 64 == 0 on images with 16-divisible sides (every 3 x 3 conv of the 32 x 32 / 16 x 16 levels but in_conv / out_conv).
constexpr int WS_TC = 64;
constexpr int WS_DY_BYTES = 256 * DY_ROW;                  // 256 px x 64 n
constexpr int WS_XP_BYTES = 336 * X_ROW;                   // one channel-half plane of the halo: 324 rows (+ 12 of padding) x 32 c
constexpr int WS_STAGE = WS_DY_BYTES + 2 * WS_XP_BYTES;    // 75776
constexpr int WS_CNT_AT = 2 * WS_STAGE;                    // ready[4] | freed[4] | junk words for the signalling waves' idle lanes
constexpr int WS_BYTES = WS_CNT_AT + 1024;
static_assert(WS_BYTES <= 160 * 1024 && TN * 9 * WS_TC * 4 + 1024 <= WS_CNT_AT, "LDS");
constexpr unsigned long long WS_WAIT_TICKS = 500000000ull; // 5 s of the 100 MHz clock: a protocol error traps, it never hangs the GPU
__device__ unsigned g_ws_fault[4];

__device__ __forceinline__ void ws_wait_all_ge(unsigned addr, unsigned target) {
    unsigned long long t_first = 0;
    for (unsigned spins = 0;; ++spins) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        const int m = min(min((int)(v.x - target), (int)(v.y - target)), min((int)(v.z - target), (int)(v.w - target)));
        if (__builtin_amdgcn_readfirstlane(m) >= 0) break;
        if ((spins & 4095u) == 4095u) {
            const unsigned long long now = wall_clock64();
            if (t_first == 0) t_first = now | 1ull;
            else if (now - t_first > WS_WAIT_TICKS) {
                if ((threadIdx.x & 63) == 0) { g_ws_fault[0] = blockIdx.x; g_ws_fault[1] = addr; g_ws_fault[2] = target; g_ws_fault[3] = 1; __threadfence_system(); }
                __builtin_trap();
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// a wave publishes its counter: lane 0 writes the counter, every other lane its own junk word (no exec masking)
__device__ __forceinline__ void ws_publish(unsigned lane_addr, unsigned value) {
    asm volatile("ds_write_b32 %0, %1" :: "v"(lane_addr), "v"(value) : "memory");
}

__global__ __launch_bounds__(512, 2)
void wgrad3x3_ws_kernel(Wg3Args a) {
    constexpr int PW = 16, HW = 18;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = a.tiles_n * a.tiles_c;
    const int lid = xcd_logical(blockIdx.x, gridDim.x);
    const int split = (int)fdiv((unsigned)lid, a.d_tiles), tile = lid - split * ntiles;
    const int tn = (int)fdiv((unsigned)tile, a.d_tiles_c), tc = tile - tn * a.tiles_c;
    const int st_begin = split * a.stages_per_split, st_end = min(a.stages, st_begin + a.stages_per_split);
    const int nst = max(st_end - st_begin, 0);
    const int tpi = a.tiles_y * a.tiles_x;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned cnt0 = lds0 + WS_CNT_AT;
    if (tid < 64) reinterpret_cast<unsigned*>(smem + WS_CNT_AT)[tid] = 0u;
    __syncthreads();

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);
    float bsum = 0.f;
#ifdef WG_TIMING
    // debug builds only (scripts/wg_ws_timeline.py): per wave {loop begin, loop end, clocks in semaphore waits, clocks in vmcnt waits (loaders), stages, end of kernel}
    unsigned long long wclk = 0, vclk = 0, tw0 = 0;
    const unsigned long long t_begin = clock64();
#define WS_T0() do { tw0 = clock64(); } while (0)
#define WS_T1(acc_) do { acc_ += clock64() - tw0; } while (0)
#else
#define WS_T0()
#define WS_T1(acc_)
#endif
    const int wn = wave & 1, wc = (wave >> 1) & 1;
    const bool want_bias = a.dbias != nullptr && tc == 0;

    // ---- DMA plan (every wave builds it: the first two stages are fetched by all eight waves, see below)
    auto rsrc_of = [&](const void* p, unsigned extent) {
        const unsigned long long ad = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rdy = rsrc_of(a.dy, a.dy_extent), rx = rsrc_of(a.x, a.x_extent);
    // dy tile: instruction d (0..31) fills the 1 KiB at d * 1024 = stage pixels d * 8 + lane / 8 (patch row d / 2, columns (d & 1) * 8 ...), physical
    // chunk lane % 8; the logical chunk is XOR-swizzled by bit 1 of the pixel (= bit 4 of the lane), as in the kernel above.
    // halo: instruction g (0..41) -> channel-half plane g / 21, rows (g % 21) * 16 + lane / 4 of the 18 x 18 halo.
    const unsigned dy_lane = (unsigned)((((long long)(lane >> 3)) * a.dy_ld + tn * TN + (((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 3)) * 2);
    const unsigned x_c2 = (unsigned)((tc * WS_TC + (lane & 3) * 8) * 2);
    auto issue = [&](int s, int d0, int dn, int g0, int gstep) {          // dy instructions d0 .. d0 + dn - 1, halo instructions g0, g0 + gstep, ...
        const int st = st_begin + s, slot = s & 1;
        const int img = (int)fdiv((unsigned)st, a.d_tpi), pt = st - img * tpi;
        const int ty = (int)fdiv((unsigned)pt, a.d_tiles_x), tx = pt - ty * a.tiles_x;
        const int py0 = ty * 16, px0 = tx * PW;
        char* dst = smem + slot * WS_STAGE;
        const unsigned base = (unsigned)((((long long)(img * a.H + py0) * a.W + px0) * a.dy_ld) * 2);
        for (int d = d0; d < d0 + dn; ++d) {
            const unsigned so = base + (unsigned)((((d >> 1) * a.W + (d & 1) * 8) * (int)a.dy_ld) * 2);      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, dy_lane, so, 0, 0);
        }
        const unsigned base_x = (unsigned)((((long long)(img * (a.H >> a.ups) + (py0 >> a.ups)) * (a.W >> a.ups) + (px0 >> a.ups)) * a.x_ld) * 2);
        for (int g = g0; g < 42; g += gstep) {
            const int pl = g >= 21 ? 1 : 0, q = g - pl * 21;
            const int hp = q * 16 + (lane >> 2);
            const int hy = (hp * 3641) >> 16, hx = hp - hy * HW;                // hp / 18 for hp < 336
            const int iy = py0 + hy - 1, ix = px0 + hx - 1;
            const bool ok = hp < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int dyi = (hy - 1) >> a.ups, dxi = (hx - 1) >> a.ups;
            unsigned o = ok ? base_x + (unsigned)((dyi * (a.W >> a.ups) + dxi) * (int)a.x_ld * 2) + x_c2 + (unsigned)(pl * 64) : OOB;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + WS_DY_BYTES + pl * WS_XP_BYTES + q * 1024), 16, o, 0, 0, 0);
        }
    };
    // ---- the first two stages are fetched by ALL eight waves (the consumers have nothing to multiply yet; an LDS-DMA instruction holds its
    // wave ~200 clk, so a loader needs ~3800 clk for its 18-19 of a stage) and published by a block barrier
    if (nst > 0) issue(0, wave * 4, 4, wave, 8);
    if (nst > 1) issue(1, wave * 4, 4, wave, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave >= 4) {
        // ---------------------------------------------------------------- loaders
        const int lw = wave - 4;
        const unsigned sig_ready = lane == 0 ? cnt0 + (unsigned)(lw * 4) : cnt0 + 64u + (unsigned)(lane * 4);
        for (int s = 2; s < nst; ++s) {
            WS_T0();
            ws_wait_all_ge(cnt0 + 16u, (unsigned)(s - 1));                      // every consumer is through stage s - 2: its slot is free
            WS_T1(wclk);
            issue(s, lw * 8, 8, lw, 4);
            WS_T0();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WS_T1(vclk);
            ws_publish(sig_ready, (unsigned)(s + 1));
        }
    } else {
        // ---------------------------------------------------------------- consumers
        const int i16 = lane & 15, q4 = i16 >> 2, h = lane >> 5;
        const int mcol = ((lane >> 4) & 1) * 16 + (i16 & 3) * 4;
        const int a_m = wn * 32 + mcol, a_row = 8 * h + q4;
        const int a_off = a_row * DY_ROW + (((a_m >> 3) ^ (((a_row >> 1) & 1) << 2)) << 4) + (a_m & 7) * 2;
        const int k_lo = 8 * h + q4, k_hi = k_lo + 4;                      // patch columns of this lane's two k groups (a k-step is one patch row)
        const int b_lo = k_lo * X_ROW + (mcol >> 3) * 16 + (mcol & 7) * 2 + wc * WS_XP_BYTES;
        const int b_hi = k_hi * X_ROW + (mcol >> 3) * 16 + (mcol & 7) * 2 + wc * WS_XP_BYTES;
        const unsigned sig_freed = lane == 0 ? cnt0 + 16u + (unsigned)(wave * 4) : cnt0 + 64u + 256u + (unsigned)(lane * 4);
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        struct Frag { uint2 lo, hi; };
        auto frag4 = [](const Frag& f) { return __builtin_bit_cast(bf16x8, u32x4{f.lo.x, f.lo.y, f.hi.x, f.hi.y}); };
        auto bias_add = [&](const Frag& f) {
            bsum += __uint_as_float(f.lo.x << 16) + __uint_as_float(f.lo.x & 0xffff0000u) + __uint_as_float(f.lo.y << 16) + __uint_as_float(f.lo.y & 0xffff0000u)
                  + __uint_as_float(f.hi.x << 16) + __uint_as_float(f.hi.x & 0xffff0000u) + __uint_as_float(f.hi.y << 16) + __uint_as_float(f.hi.y & 0xffff0000u);
        };
#define WS_READ_A(F, j) do { TR_READ(F.lo, abase, (j) * 16 * DY_ROW); TR_READ(F.hi, abase, (j) * 16 * DY_ROW + 4 * DY_ROW); } while (0)
#define WS_READ_H(F, hr) do { _Pragma("unroll") for (int sft = 0; sft < 3; ++sft) { TR_READ(F[sft].lo, blo0, ((hr) * HW + sft) * X_ROW); TR_READ(F[sft].hi, bhi0, ((hr) * HW + sft) * X_ROW); } } while (0)
#define WS_WAIT_AH(FA, F) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA.lo), "+v"(FA.hi), "+v"(F[0].lo), "+v"(F[0].hi), "+v"(F[1].lo), "+v"(F[1].hi), "+v"(F[2].lo), "+v"(F[2].hi) :: "memory")
#define WS_WAIT_H(F) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0].lo), "+v"(F[0].hi), "+v"(F[1].lo), "+v"(F[1].hi), "+v"(F[2].lo), "+v"(F[2].hi) :: "memory")
        for (int s = 0; s < nst; ++s) {
            const unsigned stage0 = lds0 + (unsigned)((s & 1) * WS_STAGE);
            const unsigned abase = stage0 + (unsigned)a_off, blo0 = stage0 + (unsigned)(WS_DY_BYTES + b_lo), bhi0 = stage0 + (unsigned)(WS_DY_BYTES + b_hi);
            WS_T0();
            if (s >= 2) ws_wait_all_ge(cnt0, (unsigned)(s + 1));                // every loader's share of this stage has landed (stages 0, 1: the barrier above)
            WS_T1(wclk);
            Frag fa[2], fh[4][3];
            WS_READ_A(fa[0], 0); WS_READ_H(fh[0], 0);
            WS_READ_H(fh[1], 1); WS_READ_H(fh[2], 2);
            WS_WAIT_AH(fa[0], fh[0]); WS_WAIT_H(fh[1]); WS_WAIT_H(fh[2]);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // patch row j pairs with halo rows j, j + 1, j + 2 (tap rows 0, 1, 2); the reads of row j + 1 go out under its nine MFMAs
                if (j + 1 < 16) WS_READ_A(fa[(j + 1) & 1], j + 1);
                if (j + 3 < 18) WS_READ_H(fh[(j + 3) & 3], j + 3);
                if (want_bias && wc == 0) bias_add(fa[j & 1]);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int sft = 0; sft < 3; ++sft)
                        acc[r * 3 + sft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(fa[j & 1]), frag4(fh[(j + r) & 3][sft]), acc[r * 3 + sft], 0, 0, 0);
                if (j + 3 < 18) { if (j + 1 < 16) WS_WAIT_AH(fa[(j + 1) & 1], fh[(j + 3) & 3]); else WS_WAIT_H(fh[(j + 3) & 3]); }
                else if (j + 1 < 16) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[(j + 1) & 1].lo), "+v"(fa[(j + 1) & 1].hi) :: "memory");
            }
            ws_publish(sig_freed, (unsigned)(s + 1));                            // (behind this stage's last reads, all of them retired above)
        }
#undef TR_READ
#undef WS_READ_A
#undef WS_READ_H
#undef WS_WAIT_AH
#undef WS_WAIT_H
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef WG_TIMING
    const unsigned long long t_loop_end = clock64();
#endif
    // ---- consumers store their accumulators straight from registers: for a fixed (n, tap) the 32 lanes of a half wave hold 32 consecutive
    // in-channels = one 128-byte line of the packed gradient (the kernel above stages the tile through LDS for 16-byte stores: two block
    // barriers and 147 KiB of LDS traffic, 10 000 clk per block here)
    float* out = a.dw + (a.atomic ? 0 : (long long)split * a.slab_stride);
    if (wave < 4) {
        const int cc = tc * WS_TC + wc * 32 + (lane & 31), nb = tn * TN + wn * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nb + (r & 3) + 8 * (r >> 2);
            if (n < a.Nreal) {
                float* o = out + (long long)n * 9 * a.C + cc;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (a.atomic) atomicAdd(o + t * a.C, acc[t][r]); else o[t * a.C] = acc[t][r];
                }
            }
        }
        if (want_bias && wc == 0) {
            // the two k halves of a column sum sit in lanes l and l + 32
            const float tot = bsum + __shfl_xor(bsum, 32, 64);
            const int n = tn * TN + wn * 32 + (lane & 31);
            if (lane < 32 && n < a.Nreal) {
                if (a.atomic) atomicAdd(a.dbias + n, tot);
                else a.dbias[(long long)split * a.bias_stride + n] = tot;
            }
        }
    }
#ifdef WG_TIMING
    if (g_wg_timing && lane == 0) {
        unsigned long long* o = g_wg_timing + ((long long)blockIdx.x * 8 + wave) * 16;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[0] = t_begin; o[1] = t_loop_end; o[2] = wclk; o[3] = vclk; o[4] = (unsigned long long)nst; o[5] = clock64();
    }
#endif
#undef WS_T0
#undef WS_T1
}

struct Plan { int ok, stage_px, PW, PH, NB, stages, splits, per, tiles_n, tiles_c, ws; };

static Plan make_plan(int B, int H, int W, int C, int N, int want_splits) {
    Plan p; memset(&p, 0, sizeof(p));
    if (C % TC || N % 8 || H <= 0 || W <= 0) return p;
    // the wave-specialised kernel (64-wide in-channel tiles): DDPM_WGRAD3_NO_WS=1 keeps every geometry on the kernel above
    static const bool no_ws = getenv("DDPM_WGRAD3_NO_WS") != nullptr;
    p.ws = !no_ws && W >= 16 && W % 16 == 0 && H % 16 == 0 && C % WS_TC == 0 && N % TN == 0;
    if (W >= 16) { if (W % 16 || H % 16) return p; p.stage_px = 256; p.PW = 16; p.PH = 16; p.NB = 1; }
    else if (W == 8 && H == 8) { p.stage_px = 128; p.PW = 8; p.PH = 8; p.NB = 2; }
    else if (W == 4 && H == 4) { p.stage_px = 128; p.PW = 4; p.PH = 4; p.NB = 8; }
    else return p;
    const int tpi = (H / p.PH) * (W / p.PW);
    p.stages = ((B + p.NB - 1) / p.NB) * tpi;
    p.tiles_n = (N + TN - 1) / TN; p.tiles_c = C / (p.ws ? WS_TC : TC);
    const int tiles = p.tiles_n * p.tiles_c;
    int splits = want_splits;
    if (splits <= 0) {
        // One block per CU (the stage ring fills the LDS, 512 threads x up to 256 registers fill the register file): a launch that
        // takes all 256 CUs leaves NOTHING for the main stream — its next kernels, however small, wait for a whole block of this one
        // (31-93 us; seen as a bimodal 3 / 30+ us duration of the GroupNorm finishing kernels on the CelebA-HQ step).  Half the chip
        // is the measured optimum: tiles x slices ~ 128 (same-box A/B of the step: 256 -> 10.31, 160 -> 9.95, 128 -> 10.08 / 9.81,
        // 96 -> 10.3, 64 -> 11.0 ms; CelebA-HQ B = 2: 12.36 -> 11.97 ms), which also halves the slab copies ddpm_wgrad_reduce sums.
        // Keep >= 4 stages per slice so that the block's prologue / final reduction amortise.
        static const int cu_budget = getenv("DDPM_WGRAD3_CUS") ? atoi(getenv("DDPM_WGRAD3_CUS")) : 128;
        static const int cu_budget_small = getenv("DDPM_WGRAD3_CUS_SMALL") ? atoi(getenv("DDPM_WGRAD3_CUS_SMALL")) : cu_budget;   // the 8 x 8 / 4 x 4 levels
        splits = ddpm_cu_budget(W <= 8 ? cu_budget_small : cu_budget) / tiles;
        const int cap = p.stages >= 4 ? p.stages / 4 : 1;
        if (splits > cap) splits = cap;
        if (splits < 1) splits = 1;
    }
    if (splits > p.stages) splits = p.stages;
    p.per = (p.stages + splits - 1) / splits;
    p.splits = (p.stages + p.per - 1) / p.per;
    p.ok = 1;
    return p;
}

}  // namespace

// diagnostic (include/ddpm_hip_debug.h): the record a timed-out semaphore wait of wgrad3x3_ws_kernel left before it trapped —
// out4 (HOST memory) receives {block, LDS address of the counter bank, value waited for, 0 = none | 1 = expired}
extern "C" int ddpm_wgrad3x3_ws_last_fault(unsigned* out4) {
    if (!out4) return DDPM_ERR_NULL;
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_ws_fault), 4 * sizeof(unsigned)) == hipSuccess ? DDPM_OK : DDPM_ERR_LAUNCH;
}

// instrumentation (include/ddpm_hip_debug.h): which kernel serves this geometry — 14 wgrad3x3_ws_kernel, 6 wgrad3x3_kernel, -1 not covered
extern "C" int ddpm_conv3x3_wgrad_variant(int B, int H, int W, int C, int N) {
    const Plan p = make_plan(B, H, W, C, N, 0);
    return !p.ok ? -1 : (p.ws ? 14 : 6);
}

// number of slab copies ddpm_conv3x3_wgrad_nhwc writes for this geometry (0: geometry not covered by the patch kernel)
extern "C" int ddpm_conv3x3_wgrad_splits(int B, int H, int W, int C, int N, int splits) {
    const Plan p = make_plan(B, H, W, C, N, splits);
    return p.ok ? p.splits : 0;
}

// dw (packed [Nreal][3][3][C]) and optionally dbias of a 3x3 / stride 1 / pad 1 convolution, bf16 activations.
//   slab_stride == 0: fp32 atomics into dw / dbias (zero-filled or holding earlier contributions);
//   slab_stride  > 0: slice s stores its partial at dw + s*slab_stride and dbias + s*bias_stride; every one of the
//                     ddpm_conv3x3_wgrad_splits(...) copies is written completely (ddpm_wgrad_reduce then sums them).
// Returns DDPM_ERR_SHAPE for geometries the patch kernel does not cover (the caller uses ddpm_conv2d_wgrad_nhwc + ddpm_colsum).
static int wgrad3x3_launch(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                           float* dbias, long long bias_stride, int B, int H, int W, int C, int N, int Nreal, int splits,
                           int dtype, int ups, void* stream) {
    if (!dy || !x || !dw) return DDPM_ERR_NULL;
    if (ups && ((H | W) & 1)) return DDPM_ERR_SHAPE;
    if (dtype != DDPM_BF16) return DDPM_ERR_DTYPE;
    if (B <= 0 || Nreal <= 0 || Nreal > N) return DDPM_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(x) || dy_ld % 8 || x_ld % 8) return DDPM_ERR_ALIGN;
    const Plan p = make_plan(B, H, W, C, N, splits);
    if (!p.ok) return DDPM_ERR_SHAPE;
    if (slab_stride < 0 || (slab_stride > 0 && slab_stride < (long long)Nreal * 9 * C)) return DDPM_ERR_SHAPE;
    if (slab_stride > 0 && dbias && bias_stride < Nreal) return DDPM_ERR_SHAPE;
    const long long dyb = ((long long)B * H * W * dy_ld - (dy_ld - N)) * 2, xb = ((long long)B * (H >> ups) * (W >> ups) * x_ld - (x_ld - C)) * 2;
    if (dyb > 0x7ffffff0ll || xb > 0x7ffffff0ll) return DDPM_ERR_SHAPE;
    Wg3Args a; memset(&a, 0, sizeof(a));
    a.dy = dy; a.dy_ld = dy_ld; a.dy_extent = (unsigned)dyb; a.x = x; a.x_ld = x_ld; a.x_extent = (unsigned)xb;
    a.dw = dw; a.slab_stride = slab_stride; a.dbias = dbias; a.bias_stride = bias_stride;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.Nreal = Nreal; a.ups = ups ? 1 : 0;
    a.PH = p.PH; a.NB = p.NB;
    int lpp = 0; while ((1 << lpp) < p.PH * p.PW) ++lpp;
    a.lPP = lpp;
    a.tiles_y = H / p.PH; a.tiles_x = W / p.PW;
    a.stages = p.stages; a.stages_per_split = p.per; a.tiles_n = p.tiles_n; a.tiles_c = p.tiles_c;
    a.atomic = slab_stride == 0;
    a.d_tiles = make_fastdiv((unsigned)(p.tiles_n * p.tiles_c)); a.d_tiles_c = make_fastdiv((unsigned)p.tiles_c);
    a.d_tpi = make_fastdiv((unsigned)(a.tiles_y * a.tiles_x)); a.d_tiles_x = make_fastdiv((unsigned)a.tiles_x);
    a.d_halo_img = make_fastdiv((unsigned)((p.PH + 2) * (p.PW + 2))); a.d_halo_w = make_fastdiv((unsigned)(p.PW + 2));
    const dim3 grid(p.tiles_n * p.tiles_c * p.splits);
    hipStream_t st = (hipStream_t)stream;
#define WG_LAUNCH(SPX, PWV, RINGV)                                                                                                \
    do {                                                                                                                          \
        constexpr int NIX = ((SPX == 256 ? 324 : (PWV == 8 ? 200 : 288)) * 4 + 511) / 512;                                       \
        constexpr int XB = SPX == 256 ? 336 * 64 : NIX * 128 * 64;                                                                \
        constexpr int RAW = RINGV * (SPX * 128 + XB) + (SPX == 256 ? 1024 : 0);         /* ring (+ dump): <= 160 KiB */          \
        constexpr int LDS = RAW > 147 * 1024 ? RAW : 147 * 1024;                        /* the final k-group sum needs 2 x 72 KiB + 2 KiB */ \
        static_assert(LDS <= 160 * 1024, "LDS budget");                                                                          \
        static DevOnce attr_set;                                                                                             \
        if (!attr_set) {                                                                                                          \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel<SPX, PWV, RINGV>),                            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DDPM_ERR_LAUNCH;      \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((wgrad3x3_kernel<SPX, PWV, RINGV>), grid, dim3(512), LDS, st, a);                                      \
    } while (0)
    if (p.ws) {
        static DevOnce ws_attr;
        if (!ws_attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WS_BYTES) != hipSuccess) return DDPM_ERR_LAUNCH;
            ws_attr = true;
        }
        hipLaunchKernelGGL(wgrad3x3_ws_kernel, grid, dim3(512), WS_BYTES, st, a);
    }
    else if (p.stage_px == 256) WG_LAUNCH(256, 16, 3);            // 3 x (32 KiB dy + 21 KiB halo) + 1 KiB = 160 KiB
    else if (p.PW == 8) WG_LAUNCH(128, 8, 4);
    else WG_LAUNCH(128, 4, 4);
#undef WG_LAUNCH
    return check_launch();
}

extern "C" int ddpm_conv3x3_wgrad_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                                       float* dbias, long long bias_stride, int B, int H, int W, int C, int N, int Nreal, int splits,
                                       int dtype, void* stream) {
    return wgrad3x3_launch(dy, dy_ld, x, x_ld, dw, slab_stride, dbias, bias_stride, B, H, W, C, N, Nreal, splits, dtype, 0, stream);
}

// The same for the conv of an Upsample block (nearest 2x, then 3x3 / s1 / p1): H, W are the OUTPUT image (dy's), x is stored at H/2 x W/2.
extern "C" int ddpm_conv3x3_wgrad_up_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                                          float* dbias, long long bias_stride, int B, int H, int W, int C, int N, int Nreal, int splits,
                                          int dtype, void* stream) {
    return wgrad3x3_launch(dy, dy_ld, x, x_ld, dw, slab_stride, dbias, bias_stride, B, H, W, C, N, Nreal, splits, dtype, 1, stream);
}

#ifdef WG_TIMING
extern "C" int ddpm_debug_set_wg_timing(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_timing), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif
