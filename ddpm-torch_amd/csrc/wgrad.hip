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
    int dy_il[NI_DY]; unsigned dy_rel[NI_DY];
#pragma unroll
    for (int i = 0; i < NI_DY; ++i) {
        const int v = tid + 512 * i, p = v >> 3;
        const int lc = (v & 7) ^ (((p >> 1) & 1) << 2);
        const int il = p >> a.lPP, q = p & (PP - 1), py = q / PW, px = q - py * PW;
        const int n = tn * TN + lc * 8;
        dy_il[i] = n < a.N ? il : (1 << 20);                                   // channel block outside dy: always out of range
        dy_rel[i] = (unsigned)((((long long)(il * a.H + py) * a.W + px) * a.dy_ld + n) * 2);
    }
    int x_il[NI_X], x_hy[NI_X], x_hx[NI_X]; unsigned x_c2[NI_X];
#pragma unroll
    for (int i = 0; i < NI_X; ++i) {
        const int v = tid + 512 * i, hp = v >> 2;
        const int il = (int)fdiv((unsigned)hp, a.d_halo_img), rem = hp - il * (HH * HW);
        const int hy = (int)fdiv((unsigned)rem, a.d_halo_w), hx = rem - hy * HW;
        x_il[i] = hp < HP ? il : (1 << 20);
        x_hy[i] = hy - 1; x_hx[i] = hx - 1;
        x_c2[i] = (unsigned)((tc * TC + (v & 3) * 8) * 2);
    }
    auto issue_stage = [&](int st, int slot) {
        const int grp = (int)fdiv((unsigned)st, a.d_tpi), pt = st - grp * tpi;
        const int ty = (int)fdiv((unsigned)pt, a.d_tiles_x), tx = pt - ty * a.tiles_x;
        const int img0 = grp * a.NB, py0 = ty * a.PH, px0 = tx * PW;
        char* dst = smem + slot * STAGE_BYTES;
        const unsigned base = (unsigned)((((long long)(img0 * a.H + py0) * a.W + px0) * a.dy_ld) * 2);
#pragma unroll
        for (int i = 0; i < NI_DY; ++i) {
            const unsigned o = (img0 + dy_il[i] < a.B) ? base + dy_rel[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)(dst + (wave * 64 + 512 * i) * 16), 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI_X; ++i) {
            const int gi = img0 + x_il[i], iy = py0 + x_hy[i], ix = px0 + x_hx[i];
            const bool ok = gi < a.B && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned o = ok ? (unsigned)((((long long)(gi * a.H + iy) * a.W + ix) * a.x_ld) * 2) + x_c2[i] : OOB;
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

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);
    float bsum = 0.f;                                    // column sum of dy for n = wn*32 + (lane & 31), this lane's k half
    const bool want_bias = a.dbias != nullptr && tc == 0;

    // ---- prologue: RING-1 stages in flight, wait for the first
#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
        if (t < nst) issue_stage(st_begin + t, t);
    wait_inflight(min(RING - 2, nst - 1));
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < nst; ++it) {
        const int slot = it % RING;
        // the stage RING-1 ahead goes into the slot read in iteration it-1 (every wave is past the barrier that ended it)
        if (it + RING - 1 < nst) issue_stage(st_begin + it + RING - 1, (it + RING - 1) % RING);
        const char* dyt = smem + slot * STAGE_BYTES;
        const char* xt = dyt + DY_BYTES;
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            const int ks = wk + 4 * u;                                   // this wave's k-steps: wk, wk+4, ...
            // A: dy fragment
            const char* pa = dyt + ks * 16 * DY_ROW + a_off;
            const uint2 alo = tr_read(pa), ahi = tr_read(pa + 4 * DY_ROW);
            const u32x4 fa = u32x4{alo.x, alo.y, ahi.x, ahi.y};
            if (want_bias) {
                bsum += __uint_as_float(fa.x << 16) + __uint_as_float(fa.x & 0xffff0000u) + __uint_as_float(fa.y << 16) + __uint_as_float(fa.y & 0xffff0000u)
                      + __uint_as_float(fa.z << 16) + __uint_as_float(fa.z & 0xffff0000u) + __uint_as_float(fa.w << 16) + __uint_as_float(fa.w & 0xffff0000u);
            }
            // halo base of the k-step: image il, first patch row pr0 (a k-step covers 16 / PW patch rows)
            const int p0 = ks * 16, il = p0 >> a.lPP, pr0 = (p0 & (PP - 1)) / PW;
            const char* pb = xt + ((il * HH + pr0) * HW) * X_ROW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, s = t - 3 * r;
                const char* pt = pb + (r * HW + s) * X_ROW;
                const uint2 blo = tr_read(pt + b_lo), bhi = tr_read(pt + b_hi);
                const u32x4 fb = u32x4{blo.x, blo.y, bhi.x, bhi.y};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[t], 0, 0, 0);
            }
        }
        // stage it+1 must have landed before anyone reads it; newer stages may stay in flight across the barrier
        wait_inflight(min(RING - 2, max(nst - 2 - it, 0)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- sum the four k groups through LDS (the ring is idle now): groups 1..3 hand their accumulators to group 0, one at a time
    float* red = reinterpret_cast<float*>(smem);           // [wn][9 taps][16 regs][64 lanes] = 72 KiB
    float* rbias = red + 2 * 9 * 16 * 64;                  // [wn][64 lanes]
    for (int g = 1; g < 4; ++g) {
        if (wk == g) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    *reinterpret_cast<f32x4*>(red + ((wn * 9 + t) * 4 + r4) * 256 + lane * 4) =
                        f32x4{acc[t][4 * r4], acc[t][4 * r4 + 1], acc[t][4 * r4 + 2], acc[t][4 * r4 + 3]};
            if (want_bias) rbias[wn * 64 + lane] = bsum;
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((wn * 9 + t) * 4 + r4) * 256 + lane * 4);
                    acc[t][4 * r4] += v.x; acc[t][4 * r4 + 1] += v.y; acc[t][4 * r4 + 2] += v.z; acc[t][4 * r4 + 3] += v.w;
                }
            if (want_bias) bsum += rbias[wn * 64 + lane];
        }
        __syncthreads();
    }
    if (wk != 0) return;

    // ---- output: packed gradient dw[n][tap][c]; accumulator (reg r, lane l) = row (r&3) + 8*(r>>2) + 4*(l>>5), column l & 31
    float* out = a.dw + (a.atomic ? 0 : (long long)split * a.slab_stride);
    const int c = tc * TC + (lane & 31);
    const int n0 = tn * TN + wn * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (r & 3) + 8 * (r >> 2);
            if (n < a.Nreal) {
                float* o = out + ((long long)n * 9 + t) * a.C + c;
                if (a.atomic) atomicAdd(o, acc[t][r]); else *o = acc[t][r];
            }
        }
    if (want_bias) {
        const float tot = bsum + __shfl_xor(bsum, 32, 64);           // the two k halves of the fragment
        const int n = tn * TN + wn * 32 + (lane & 31);
        if (lane < 32 && n < a.Nreal) {
            if (a.atomic) atomicAdd(a.dbias + n, tot);
            else a.dbias[(long long)split * a.bias_stride + n] = tot;
        }
    }
}

struct Plan { int ok, stage_px, PW, PH, NB, stages, splits, per, tiles_n, tiles_c; };

static Plan make_plan(int B, int H, int W, int C, int N, int want_splits) {
    Plan p; memset(&p, 0, sizeof(p));
    if (C % TC || N % 8 || H <= 0 || W <= 0) return p;
    if (W >= 16) { if (W % 16 || H % 16) return p; p.stage_px = 256; p.PW = 16; p.PH = 16; p.NB = 1; }
    else if (W == 8 && H == 8) { p.stage_px = 128; p.PW = 8; p.PH = 8; p.NB = 2; }
    else if (W == 4 && H == 4) { p.stage_px = 128; p.PW = 4; p.PH = 4; p.NB = 8; }
    else return p;
    const int tpi = (H / p.PH) * (W / p.PW);
    p.stages = ((B + p.NB - 1) / p.NB) * tpi;
    p.tiles_n = (N + TN - 1) / TN; p.tiles_c = C / TC;
    const int tiles = p.tiles_n * p.tiles_c;
    int splits = want_splits;
    if (splits <= 0) {
        // ~2 blocks per CU (512 blocks); keep >= 4 stages per slice so that the block's prologue / final reduction amortise
        splits = (512 + tiles - 1) / tiles;
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
extern "C" int ddpm_conv3x3_wgrad_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                                       float* dbias, long long bias_stride, int B, int H, int W, int C, int N, int Nreal, int splits,
                                       int dtype, void* stream) {
    if (!dy || !x || !dw) return DDPM_ERR_NULL;
    if (dtype != DDPM_BF16) return DDPM_ERR_DTYPE;
    if (B <= 0 || Nreal <= 0 || Nreal > N) return DDPM_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(x) || dy_ld % 8 || x_ld % 8) return DDPM_ERR_ALIGN;
    const Plan p = make_plan(B, H, W, C, N, splits);
    if (!p.ok) return DDPM_ERR_SHAPE;
    if (slab_stride < 0 || (slab_stride > 0 && slab_stride < (long long)Nreal * 9 * C)) return DDPM_ERR_SHAPE;
    if (slab_stride > 0 && dbias && bias_stride < Nreal) return DDPM_ERR_SHAPE;
    const long long dyb = ((long long)B * H * W * dy_ld - (dy_ld - N)) * 2, xb = ((long long)B * H * W * x_ld - (x_ld - C)) * 2;
    if (dyb > 0x7ffffff0ll || xb > 0x7ffffff0ll) return DDPM_ERR_SHAPE;
    Wg3Args a; memset(&a, 0, sizeof(a));
    a.dy = dy; a.dy_ld = dy_ld; a.dy_extent = (unsigned)dyb; a.x = x; a.x_ld = x_ld; a.x_extent = (unsigned)xb;
    a.dw = dw; a.slab_stride = slab_stride; a.dbias = dbias; a.bias_stride = bias_stride;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.Nreal = Nreal;
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
        constexpr int LDS = RINGV * (SPX * 128 + XB) + 1024 > 75 * 1024 ? RINGV * (SPX * 128 + XB) + 1024 : 75 * 1024;          \
        static bool attr_set = false;                                                                                             \
        if (!attr_set) {                                                                                                          \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel<SPX, PWV, RINGV>),                            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DDPM_ERR_LAUNCH;      \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((wgrad3x3_kernel<SPX, PWV, RINGV>), grid, dim3(512), LDS, st, a);                                      \
    } while (0)
    if (p.stage_px == 256) WG_LAUNCH(256, 16, 3);            // 3 x (32 KiB dy + 21 KiB halo) + 1 KiB = 160 KiB
    else if (p.PW == 8) WG_LAUNCH(128, 8, 4);
    else WG_LAUNCH(128, 4, 4);
#undef WG_LAUNCH
    return check_launch();
}
