// The two 3x3 / stride 1 / pad 1 convolutions at the EDGES of the UNet (bf16, gfx950): in_conv (3 image channels, stored as 8, -> hid:
// ddpm_torch/models/unet.py:127 of tqch/ddpm-torch) and out_conv (hid -> 3: unet.py:141, the last module of `out_conv`).  As GEMMs they
// are degenerate — K = 27 or N = 3 — and on the 128 x 128-tile kernels they ran 25 / 46 us (padded to a tile that is 97 % zeros) where
// ONE pass over the B x 32 x 32 x 128 tensor is 8 us: 112 us of every 3.2-ms sampling step.
//
//   few-out (out_conv): one 16 x 16 output patch per block.  The 18 x 18 halo of a 64-channel chunk is DMA'd into LDS once and serves all
//     nine taps; ALL weights (9 taps x C x 16 padded outputs, 36 KiB at C = 128) sit in LDS for the block's life.  v_mfma_f32_16x16x32_bf16
//     with rows = output channels (16, the 3 real ones in lanes 0-15) and columns = 16 pixels of a patch row; a wave owns two patch rows.
//     78 KiB of LDS: two blocks per CU, one hides the other's DMA wait.  Result written as NCHW fp32 (+ bias) straight from the lanes.
//   few-in (in_conv): K = (tap, 8 channels) = 72, padded to 96 = three 16x16x32 steps whose four 8-channel K groups are four TAPS of the
//     same pixel column — a lane's B fragment is the 16-byte pixel (y + r, x + s) of the 5-KiB halo.  Output-write-bound: the 16 x N tile
//     of a patch row goes through a padded LDS stage so that every lane stores 16 contiguous bytes of an NHWC row.
#include "common.h"
#include <string.h>

namespace {

constexpr unsigned OOB = 0x7ffffff0u;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct EdgeArgs {
    const bf16_t* x; long long x_ld; unsigned x_extent;
    const bf16_t* w;                  // packed [N][9][C]
    void* out; long long out_ld;      // few-out: fp32 NCHW; few-in: bf16 NHWC, pixel pitch out_ld
    const float* bias;
    int B, H, W, C, N;
    int tiles_y, tiles_x;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned extent) {
    const unsigned long long ad = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)extent), 0x00020000);
}
__device__ __forceinline__ u32x4 lds_rd16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// ------------------------------------------------------------------------------------------------ few output channels
constexpr int FO_HALO = 324 * 128 + 1024;        // 18 x 18 pixels x 64 channels (+ the tail the sixth DMA part of wave 0 zero-fills)

__global__ __launch_bounds__(512, 2)
void conv3x3_few_out_kernel(EdgeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;                             // [324 px][8 chunks of 16 B], chunk XOR-swizzled by (hx >> 1) & 7
    char* wl = smem + FO_HALO;                     // [9 taps][C / 8][16 n] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpi = a.tiles_y * a.tiles_x;
    const int img = blockIdx.x / tpi, pt = blockIdx.x - img * tpi;
    const int py0 = (pt / a.tiles_x) * 16, px0 = (pt % a.tiles_x) * 16;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x, a.x_extent);
    const int c8 = a.C >> 3;                       // 16-byte channel groups per pixel
    // ---- weights: vector v = (tap, channel group, n) -> LDS slot v; rows n >= N are zeros
    const int nwv = 9 * c8 * 16;
    for (int v = tid; v < nwv; v += 512) {
        const int n = v & 15, q = v >> 4, tap = q / c8, cg = q - tap * c8;
        u32x4 val = zero16();
        if (n < a.N) val = ldg16(a.w + ((long long)n * 9 + tap) * a.C + cg * 8);
        *reinterpret_cast<u32x4*>(wl + v * 16) = val;
    }
    // ---- halo plan: LDS slot v = tid + 512 i -> halo pixel v >> 3, physical chunk v & 7
    unsigned hoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int v = tid + 512 * i, hp = v >> 3, hy = hp / 18, hx = hp - hy * 18;
        const int c = (v & 7) ^ ((hx >> 1) & 7);
        const int y = py0 + hy - 1, x = px0 + hx - 1;
        const bool ok = hp < 324 && y >= 0 && y < a.H && x >= 0 && x < a.W;
        hoff[i] = ok ? (unsigned)((((long long)(img * a.H + y) * a.W + x) * a.x_ld + c * 8) * 2) : OOB;
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int px = lane & 15, g = lane >> 4;
    f32x4v acc[2] = {(f32x4v)(0.f), (f32x4v)(0.f)};
    const int nchunks = a.C >> 6;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch) __syncthreads();                   // everyone is done with the previous chunk's halo
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i == 5 && wave != 0) break;        // part 5 = 32 vectors: wave 0 alone (its upper lanes write zeros into the tail)
            unsigned o = hoff[i] == OOB ? OOB : hoff[i] + (unsigned)(ch * 128);
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(halo + (i * 512 + wave * 64) * 16), 16, o, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            const int hx = px + s, key = (hx >> 1) & 7;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const unsigned wa = lds0 + FO_HALO + (unsigned)(((tap * c8 + ch * 8 + kc * 4 + g) * 16 + px) * 16);
                const u32x4 fa = lds_rd16(wa);
                u32x4 fb[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int hp = (2 * wave + q + r) * 18 + hx;
                    fb[q] = lds_rd16(lds0 + (unsigned)(hp * 128 + (((kc * 4 + g) ^ key) << 4)));
                }
                lds_wait();
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[q]), acc[q], 0, 0, 0);
            }
        }
    }
    // ---- result: lane (px, g) holds channels n = 4 g + r of pixel (row 2 wave + q, column px) -> NCHW fp32
    float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int n = 4 * g + rr;
        if (n < a.N) {
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                out[(((long long)img * a.N + n) * a.H + py0 + 2 * wave + q) * a.W + px0 + px] = acc[q][rr] + bv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ few input channels (C = 8)
constexpr int FI_HALO = 324 * 16;                 // 18 x 18 pixels x 16 bytes
constexpr int FI_STAGE_ROW = 128 * 2 + 16;        // one pixel's N <= 128 outputs, padded: consecutive pixels 4 banks apart

__global__ __launch_bounds__(512, 2)
void conv3x3_few_in_kernel(EdgeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;                             // [324 px] x 16 B
    char* wl = smem + FI_HALO;                     // [3 steps][4 taps][N] x 16 B (taps >= 9: zeros)
    char* stage = wl + 12 * a.N * 16;              // [8 waves][16 px][FI_STAGE_ROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpi = a.tiles_y * a.tiles_x;
    const int img = blockIdx.x / tpi, pt = blockIdx.x - img * tpi;
    const int py0 = (pt / a.tiles_x) * 16, px0 = (pt % a.tiles_x) * 16;
    for (int v = tid; v < 12 * a.N; v += 512) {
        const int n = v % a.N, tap = v / a.N;
        *reinterpret_cast<u32x4*>(wl + v * 16) = tap < 9 ? ldg16(a.w + ((long long)n * 9 + tap) * 8) : zero16();
    }
    if (tid < 324) {
        const int hy = tid / 18, hx = tid - hy * 18;
        const int y = py0 + hy - 1, x = px0 + hx - 1;
        const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
        *reinterpret_cast<u32x4*>(halo + tid * 16) = ok ? ldg16(a.x + ((long long)(img * a.H + y) * a.W + x) * a.x_ld) : zero16();
    }
    __syncthreads();
    const int px = lane & 15, g = lane >> 4;
    const int nt_count = a.N >> 4;
    char* st = stage + wave * (16 * FI_STAGE_ROW);
    bf16_t* outp = reinterpret_cast<bf16_t*>(a.out);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int y = 2 * wave + q;
        u32x4 fb[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int tap = min(4 * m + g, 8), r = tap / 3, s = tap - r * 3;      // (taps 9-11 multiply zero weights: any pixel)
            fb[m] = *reinterpret_cast<const u32x4*>(halo + ((y + r) * 18 + px + s) * 16);
        }
        for (int nt = 0; nt < nt_count; ++nt) {
            f32x4v acc = (f32x4v)(0.f);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const u32x4 fa = *reinterpret_cast<const u32x4*>(wl + ((m * 4 + g) * a.N + nt * 16 + px) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[m]), acc, 0, 0, 0);
            }
            const int n0 = nt * 16 + 4 * g;                                     // this lane: channels n0 .. n0 + 3 of pixel px
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            if (a.bias) { const f32x4v bv = *reinterpret_cast<const f32x4v*>(a.bias + n0); b0 = bv[0]; b1 = bv[1]; b2 = bv[2]; b3 = bv[3]; }
            uint2 pk; pk.x = pack_bf2(acc[0] + b0, acc[1] + b1); pk.y = pack_bf2(acc[2] + b2, acc[3] + b3);
            *reinterpret_cast<uint2*>(st + px * FI_STAGE_ROW + n0 * 2) = pk;
        }
        // the wave's 16 x N tile -> NHWC rows, 16 bytes per lane (a wave only reads what it wrote: no barrier)
        const int vpr = a.N >> 3;                                              // 16-byte vectors per pixel
        for (int v = lane; v < 16 * vpr; v += 64) {
            const int p = v / vpr, c = v - p * vpr;
            const u32x4 val = *reinterpret_cast<const u32x4*>(st + p * FI_STAGE_ROW + c * 16);
            stg16(outp + ((long long)(img * a.H + py0 + y) * a.W + px0 + p) * a.out_ld + c * 8, val);
        }
    }
}

}  // namespace

// Launchers for ddpm_conv2d_nhwc (gemm.hip).  -1: geometry not covered (the caller falls through to the tile kernels).
int ddpm_edgeconv_few_out_launch(const void* x, long long x_ld, const void* w, void* y, const float* bias, int B, int H, int W, int C, int N,
                                 int dry, void* stream) {
    static const bool off = getenv("DDPM_CONV_NO_EDGE") != nullptr;
    if (off || N < 1 || N > 16 || C % 64 || C > 512 || H % 16 || W % 16 || x_ld % 8 || !aligned16(x) || !aligned16(w)) return -1;
    const long long xbytes = ((long long)B * H * W * x_ld - (x_ld - C)) * 2;
    if (xbytes > 0x7ffffff0ll) return -1;
    const size_t lds = (size_t)FO_HALO + (size_t)9 * (C / 8) * 256;
    if (lds > 160 * 1024) return -1;
    if (dry) return 0;
    EdgeArgs a; memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x; a.x_ld = x_ld; a.x_extent = (unsigned)xbytes; a.w = (const bf16_t*)w; a.out = y; a.bias = bias;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.tiles_y = H / 16; a.tiles_x = W / 16;
    static DevOnce attr;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_few_out_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DDPM_ERR_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(conv3x3_few_out_kernel, dim3(B * a.tiles_y * a.tiles_x), dim3(512), lds, (hipStream_t)stream, a);
    return check_launch();
}

int ddpm_edgeconv_few_in_launch(const void* x, long long x_ld, const void* w, void* y, long long y_ld, const float* bias, int B, int H, int W,
                                int C, int N, int dry, void* stream) {
    static const bool off = getenv("DDPM_CONV_NO_EDGE") != nullptr;
    if (off || C != 8 || N % 16 || N < 16 || N > 128 || H % 16 || W % 16 || x_ld % 8 || y_ld % 8 || !aligned16(x) || !aligned16(w) || !aligned16(y) ||
        (bias && !aligned16(bias))) return -1;
    if (dry) return 0;
    EdgeArgs a; memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x; a.x_ld = x_ld; a.w = (const bf16_t*)w; a.out = y; a.out_ld = y_ld; a.bias = bias;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.tiles_y = H / 16; a.tiles_x = W / 16;
    const size_t lds = (size_t)FI_HALO + (size_t)12 * N * 16 + (size_t)8 * 16 * FI_STAGE_ROW;
    hipLaunchKernelGGL(conv3x3_few_in_kernel, dim3(B * a.tiles_y * a.tiles_x), dim3(512), lds, (hipStream_t)stream, a);
    return check_launch();
}
