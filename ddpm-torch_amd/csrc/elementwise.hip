// HBM-bound helpers of the DDPM hot path (gfx950): timestep embedding, layout/pack kernels, the fused
// diffusion algebra (q_sample, eps-MSE, ancestral / DDIM step), softmax rows for attention, reductions.
// Every kernel is a grid-stride or one-row-per-wave stream with coalesced accesses; no device allocation,
// no synchronisation — all entry points enqueue on the caller's stream and return.
#include "common.h"
#include <stdlib.h>

static inline int grid_for(long long n, int block = 256, int cap = 256 * 16) {
    long long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ timestep embedding (functions.py:10-26)
// freqs[half] = exp(-i*ln(1e4)/(half-1)) is a host-built fp32 table (same values as the reference's torch.exp);
// the fp32 product t*f is formed first (like torch.outer), then sin / cos.  t*f reaches ~1e3 rad.
__global__ void temb_kernel(const long long* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
        const int b = i / dim, j = i - b * dim;
        float v = 0.f;                                 // odd dim: last column is the zero pad
        if (j < 2 * half) {
            const float arg = __fmul_rn((float)t[b], freqs[j < half ? j : j - half]);
            v = j < half ? sinf(arg) : cosf(arg);
        }
        out[i] = v;
    }
}
extern "C" int ddpm_timestep_embedding(const long long* t, const float* freqs, float* out, int B, int dim, void* stream) {
    if (!t || !out || !freqs) return DDPM_ERR_NULL;
    if (B <= 0 || dim < 4) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(temb_kernel, dim3(grid_for((long long)B * dim)), dim3(256), 0, (hipStream_t)stream, t, freqs, out, B, dim);
    return check_launch();
}

// ------------------------------------------------------------------ NCHW fp32 -> NHWC (channel-padded) T
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int HW, int Cp) {
    const long long n = (long long)B * HW * Cp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const long long bp = i / Cp;
        const int p = (int)(bp % HW);
        const long long b = bp / HW;
        Elem<T>::st(y + i, c < C ? x[(b * C + c) * HW + p] : 0.f);
    }
}
extern "C" int ddpm_nchw_to_nhwc(const float* x, void* y, int B, int C, int HW, int Cp, int dtype, void* stream) {
    if (!x || !y) return DDPM_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0 || Cp < C) return DDPM_ERR_SHAPE;
    const int g = grid_for((long long)B * HW * Cp);
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, B, C, HW, Cp);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (float*)y, B, C, HW, Cp);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// ------------------------------------------------------------------ weight packing
// master fp32 [N][C][R][S]  ->  fwd  [N][R][S][Cp]          (k = (r,s,c), c contiguous; zero channel pad)
//                           ->  dgrad [C][R][S][Np] with taps flipped: wd[c][r][s][n] = w[n][c][R-1-r][S-1-s]
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int N, int C, int R, int S, int Cp, int Np) {
    const int RS = R * S;
    if (wf) {
        const long long n = (long long)N * RS * Cp;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % Cp); const long long r1 = i / Cp;
            const int tap = (int)(r1 % RS); const int nn = (int)(r1 / RS);
            Elem<T>::st(wf + i, c < C ? w[((long long)nn * C + c) * RS + tap] : 0.f);
        }
    }
    if (wd) {
        const long long n = (long long)C * RS * Np;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const int nn = (int)(i % Np); const long long r1 = i / Np;
            const int tap = (int)(r1 % RS); const int c = (int)(r1 / RS);
            Elem<T>::st(wd + i, nn < N ? w[((long long)nn * C + c) * RS + (RS - 1 - tap)] : 0.f);
        }
    }
}
extern "C" int ddpm_pack_weight(const float* w, void* wf, void* wd, int N, int C, int R, int S, int Cp, int Np, int dtype, void* stream) {
    if (!w || (!wf && !wd)) return DDPM_ERR_NULL;
    if (Cp < C || Np < N) return DDPM_ERR_SHAPE;
    const long long big = (long long)(Cp > C ? Cp : C) * R * S * (Np > N ? Np : N);
    const int g = grid_for(big);
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wf, (bf16_t*)wd, N, C, R, S, Cp, Np);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (float*)wf, (float*)wd, N, C, R, S, Cp, Np);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// all layers in ONE launch: descs[i] = {w, wf, wd, N, C, R | flags, Cp, Np} (8 x int64; R == S; flag 0x100: wd receives the
// 4x4 stride-2 effective dgrad kernel of an upsample conv, see below); grid = (blocks, n_tensors).
// A block walks 32 (n) x 32 (c) x R*R tiles of one tensor through LDS: the master weights are read in runs of 32*R*R contiguous
// floats per output channel, and both derived layouts — [n][tap][c] with c fastest, [c][tap'][n] with n fastest — are written
// in 64-byte runs.  (The first version gathered with a 36-byte stride between lanes: 520 us for the 35.7 M parameters of the
// CIFAR UNet, once per training step; this one moves the same 290 MB in a fifth of that.)
constexpr int PK_T = 32;
// two adjacent elements of a derived layout leave as ONE store (bf16: 4 bytes per lane instead of 2; the padded extents Cp / Np are
// multiples of 8, so pairs never straddle a row end)
template <typename T> __device__ __forceinline__ void pk_store2(T* p, float a, float b);
template <> __device__ __forceinline__ void pk_store2<bf16_t>(bf16_t* p, float a, float b) { *reinterpret_cast<unsigned*>(p) = pack_bf2(a, b); }
template <> __device__ __forceinline__ void pk_store2<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }

// RS is a template argument: every index split below is a division by a constant (the first version divided by the runtime R*S in
// 64-bit arithmetic for every element — the kernel was bound by integer VALU work, 170 us for 286 MB)
template <typename T, int RS>
__device__ __forceinline__ void pack_tensor(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int N, int C, int Cp, int Np, bool up_dgrad,
                                            float* tile) {
    constexpr int pitch = PK_T * RS + 1;                               // odd pitch: the n-fastest reads below are conflict-free
    const int tiles_c = (Cp + PK_T - 1) / PK_T, tiles_n = (Np + PK_T - 1) / PK_T;
    const int tid = threadIdx.x;
    const bool vec_ok = ((long long)C * RS) % 4 == 0 && (((unsigned long long)w) & 15) == 0;   // every row of every tile starts on a 16-byte boundary (c0 is a multiple of 32)
    for (int t = blockIdx.x; t < tiles_c * tiles_n; t += gridDim.x) {
        const int n0 = (t / tiles_c) * PK_T, c0 = (t % tiles_c) * PK_T;
        const int cw = min(PK_T, C - c0);                              // real channels in this tile (<= 0: pure padding)
        __syncthreads();                                               // the previous tile has been consumed
        // Full tiles of 16-byte-aligned rows (every tile of the 3x3 / 1x1 layers but the channel tails): the 32 x 32 RS floats arrive as
        // 16-byte vectors, ALL of a thread's requests in flight before the first is parked in LDS (scalar loads, one or two in flight per
        // thread, left the launch at 1.9 TB/s: 153 us of every training step for 286 MB).
        constexpr int V = PK_T * RS / 4, KV = PK_T * V / 256;          // vectors per row; vectors per thread (9 / 1 / 4 for RS = 9 / 1 / 4)
        static_assert(PK_T * V % 256 == 0, "whole vectors per thread");
        if (cw == PK_T && vec_ok) {
            float4 xv[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int i = tid + 256 * k, nl = i / V, v = i - nl * V;
                xv[k] = n0 + nl < N ? *reinterpret_cast<const float4*>(w + ((long long)(n0 + nl) * C + c0) * RS + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int i = tid + 256 * k, nl = i / V, v = i - nl * V;
                float* t4 = tile + nl * pitch + 4 * v;
                t4[0] = xv[k].x; t4[1] = xv[k].y; t4[2] = xv[k].z; t4[3] = xv[k].w;
            }
        } else {
            for (int i = tid; i < PK_T * PK_T * RS; i += 256) {
                const int nl = i / (PK_T * RS), rem = i - nl * (PK_T * RS);
                const bool ok = n0 + nl < N && rem < cw * RS;
                tile[nl * pitch + rem] = ok ? w[((long long)(n0 + nl) * C + c0) * RS + rem] : 0.f;
            }
        }
        __syncthreads();
        constexpr bool BF = sizeof(T) == 2;                            // bf16 layouts are padded to multiples of 8: 16-byte stores of 8 elements
        if (wf) {                                                      // wf[n][tap][c], c fastest
            if constexpr (BF) {
                for (int i = tid; i < PK_T * RS * (PK_T / 8); i += 256) {
                    const int cl = (i % (PK_T / 8)) * 8, r1 = i / (PK_T / 8), tap = r1 % RS, nl = r1 / RS;
                    if (n0 + nl < N && c0 + cl < Cp) {
                        const float* t8 = tile + nl * pitch + cl * RS + tap;
                        u32x4 o;
                        o.x = pack_bf2(t8[0], t8[RS]); o.y = pack_bf2(t8[2 * RS], t8[3 * RS]); o.z = pack_bf2(t8[4 * RS], t8[5 * RS]); o.w = pack_bf2(t8[6 * RS], t8[7 * RS]);
                        *reinterpret_cast<u32x4*>(wf + ((long long)(n0 + nl) * RS + tap) * Cp + c0 + cl) = o;
                    }
                }
            } else {
                for (int i = tid; i < PK_T * RS * (PK_T / 2); i += 256) {
                    const int cl = (i % (PK_T / 2)) * 2, r1 = i / (PK_T / 2), tap = r1 % RS, nl = r1 / RS;
                    if (n0 + nl < N && c0 + cl < Cp)
                        pk_store2<T>(wf + ((long long)(n0 + nl) * RS + tap) * Cp + c0 + cl, tile[nl * pitch + cl * RS + tap], tile[nl * pitch + (cl + 1) * RS + tap]);
                }
            }
        }
        if constexpr (RS == 9) {
            if (wd && up_dgrad) {
                // Gradient of (nearest-2x upsample -> 3x3 / pad 1 conv) w.r.t. its LOW-resolution input, as ONE strided conv over dy:
                //   dx[i][j] = sum_{a,b in {0,1}} sum_{r,s} dy[2i+a+r-1][2j+b+s-1] * D[r][s]        (D = flipped 3x3 dgrad kernel)
                //            = sum_{P,Q in 0..3} dy[2i+P-1][2j+Q-1] * E[P][Q],   E[P][Q] = sum_{a+r=P} sum_{b+s=Q} D[r][s]
                // i.e. a 4x4 / stride 2 / pad 1 convolution: 16 taps on a quarter of the pixels instead of 9 taps at full resolution
                // plus a 2x2 reduction pass.  E is formed in fp32 from the master weights and rounded once.  Layout [C][4][4][Np].
                for (int i = tid; i < PK_T * 16 * (PK_T / 2); i += 256) {
                    const int nl = (i % (PK_T / 2)) * 2, r1 = i / (PK_T / 2), tap = r1 % 16, cl = r1 / 16;
                    if (c0 + cl >= C || n0 + nl >= Np) continue;
                    const int P = tap >> 2, Q = tap & 3;
                    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b2 = 0; b2 < 2; ++b2) {
                            const int r = P - a, s2 = Q - b2;
                            if (r >= 0 && r < 3 && s2 >= 0 && s2 < 3) {
                                acc0 += tile[nl * pitch + cl * 9 + (8 - (r * 3 + s2))];
                                acc1 += tile[(nl + 1) * pitch + cl * 9 + (8 - (r * 3 + s2))];
                            }
                        }
                    pk_store2<T>(wd + ((long long)(c0 + cl) * 16 + tap) * Np + n0 + nl, acc0, acc1);
                }
                continue;
            }
        }
        if (wd) {                                                      // wd[c][tap][n] with flipped taps, n fastest
            if constexpr (BF) {
                for (int i = tid; i < PK_T * RS * (PK_T / 8); i += 256) {
                    const int nl = (i % (PK_T / 8)) * 8, r1 = i / (PK_T / 8), tap = r1 % RS, cl = r1 / RS;
                    if (c0 + cl < C && n0 + nl < Np) {
                        const float* t8 = tile + nl * pitch + cl * RS + (RS - 1 - tap);
                        u32x4 o;
                        o.x = pack_bf2(t8[0], t8[pitch]); o.y = pack_bf2(t8[2 * pitch], t8[3 * pitch]); o.z = pack_bf2(t8[4 * pitch], t8[5 * pitch]);
                        o.w = pack_bf2(t8[6 * pitch], t8[7 * pitch]);
                        *reinterpret_cast<u32x4*>(wd + ((long long)(c0 + cl) * RS + tap) * Np + n0 + nl) = o;
                    }
                }
            } else {
                for (int i = tid; i < PK_T * RS * (PK_T / 2); i += 256) {
                    const int nl = (i % (PK_T / 2)) * 2, r1 = i / (PK_T / 2), tap = r1 % RS, cl = r1 / RS;
                    if (c0 + cl < C && n0 + nl < Np)
                        pk_store2<T>(wd + ((long long)(c0 + cl) * RS + tap) * Np + n0 + nl, tile[nl * pitch + cl * RS + (RS - 1 - tap)],
                                     tile[(nl + 1) * pitch + cl * RS + (RS - 1 - tap)]);
                }
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256)
void pack_weight_multi_kernel(const long long* __restrict__ descs) {
    const long long* d = descs + 8 * (long long)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(d[0]);
    T* wf = reinterpret_cast<T*>(d[1]);
    T* wd = reinterpret_cast<T*>(d[2]);
    const int N = (int)d[3], C = (int)d[4], R = (int)d[5] & 0xff, Cp = (int)d[6], Np = (int)d[7];
    const bool up_dgrad = ((int)d[5] & 0x100) != 0;
    __shared__ float tile[PK_T * (PK_T * 9 + 1)];
    if (R == 3) pack_tensor<T, 9>(w, wf, wd, N, C, Cp, Np, up_dgrad, tile);
    else if (R == 1) pack_tensor<T, 1>(w, wf, wd, N, C, Cp, Np, false, tile);
    else if (R == 2) pack_tensor<T, 4>(w, wf, wd, N, C, Cp, Np, false, tile);
}
extern "C" int ddpm_pack_weight_multi(const long long* descs, int n_tensors, int dtype, void* stream) {
    if (!descs) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;                 // (kernel sizes: R <= 3 — the LDS tile holds 32 x 32 x 9 floats; the caller only has 1x1 and 3x3)
    static const int pk_blocks = getenv("DDPM_PACK_BLOCKS") ? atoi(getenv("DDPM_PACK_BLOCKS")) : 256;    // (blocks per tensor: 64 left the 128-tile tensors two serial load -> store rounds per block)
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<bf16_t>, dim3(pk_blocks, n_tensors), dim3(256), 0, (hipStream_t)stream, descs);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3(pk_blocks, n_tensors), dim3(256), 0, (hipStream_t)stream, descs);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// packed conv weight gradients [N][R*S][C] -> parameter layout [N][C][R*S], all layers in one launch.
// descs[i] = {src offset (floats) in gpack, dst offset in gflat, N, C, R*S}; grid = (blocks, n_tensors).
// sumsq (optional): one slot per block that receives the block's sum of squares of everything it wrote (added up in a fixed order by
// sumsq_finish_kernel: optim.hip) — was: a bank of 64 fp32 accumulators fed by atomics — the global gradient norm of
// nn.utils.clip_grad_norm_ (ddpm_torch/utils/train.py:159) falls out of this pass instead of costing another read of all gradients.
__global__ __launch_bounds__(256) void wgrad_unpack_kernel(const float* __restrict__ gpack, float* __restrict__ gflat, const long long* __restrict__ descs, float scale,
                                                            float* __restrict__ sumsq) {
    __shared__ float sh[4];
    const long long* d = descs + 5 * (long long)blockIdx.y;
    const float* src = gpack + d[0];
    float* dst = gflat + d[1];
    const unsigned C = (unsigned)d[3], RS = (unsigned)d[4];
    const unsigned total = (unsigned)(d[2] * C * RS);                       // < 2^31 per tensor (checked by the host: the flat buffer is indexed with ints elsewhere too)
    float acc = 0.f;
    if (RS == 1) {                                                          // plain copies (everything but the conv weights)
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const float v = src[i] * scale;
            dst[i] = v; acc += v * v;
        }
    } else if (RS == 9 && C % 64 == 0) {
        // 3x3 layers: a WAVE moves one (n, 64 channels) slab — nine coalesced 256-byte row reads, all in flight together, transposed through a
        // wave-private 2.3-KiB LDS slab (odd stride: conflict-free), nine coalesced 256-byte writes of the contiguous [64 c][9 taps] run.
        // (The gather form below reads 28-byte runs from nine rows per wave instruction: 93 us of every step for 286 MB.)
        __shared__ float slab[4][64 * 9];
        const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ctiles = C / 64, items = (unsigned)d[2] * ctiles;
        float* mine = slab[wave];
        for (unsigned it = blockIdx.x * 4 + wave; it < items; it += gridDim.x * 4) {
            const unsigned n = it / ctiles, c0 = (it - n * ctiles) * 64;
            float v[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v[tap] = src[((unsigned long long)n * 9 + tap) * C + c0 + lane] * scale;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) { mine[lane * 9 + tap] = v[tap]; acc += v[tap] * v[tap]; }
            __builtin_amdgcn_wave_barrier();
            float* out = dst + ((unsigned long long)n * C + c0) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) out[k * 64 + lane] = mine[k * 64 + lane];
            __builtin_amdgcn_wave_barrier();                                  // the slab is reused by the wave's next item
        }
    } else {
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const unsigned nc = i / RS, tap = i - nc * RS;                  // (32-bit divisions by a block-uniform divisor)
            const unsigned n = nc / C, c = nc - n * C;
            const float v = src[((unsigned long long)n * RS + tap) * C + c] * scale;
            dst[i] = v; acc += v * v;
        }
    }
    if (sumsq) {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) sumsq[blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);      // this block's slot (fixed-order finish: optim.hip)
    }
}
extern "C" int ddpm_wgrad_unpack(const float* gpack, float* gflat, const long long* descs, int n_tensors, float scale, void* stream) {
    if (!gpack || !gflat || !descs) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(64, n_tensors), dim3(256), 0, (hipStream_t)stream, gpack, gflat, descs, scale, (float*)nullptr);
    return check_launch();
}
extern "C" int ddpm_wgrad_unpack_sumsq(const float* gpack, float* gflat, const long long* descs, int n_tensors, float scale, float* total_sq,
                                       long long total_sq_floats, void* stream) {
    if (!gpack || !gflat || !descs || !total_sq) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;
    if (total_sq_floats < 64 + 64ll * n_tensors) return DDPM_ERR_SHAPE;         // == ddpm_mt_sumsq_slots(n_tensors), optim.hip
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(64, n_tensors), dim3(256), 0, (hipStream_t)stream, gpack, gflat, descs, scale, total_sq + 64);
    const int rc = check_launch();
    return rc ? rc : ddpm_sumsq_finish_launch(total_sq + 64, 64 * n_tensors, total_sq, stream);
}

// sum of the split-K slab copies of the packed weight gradients, many tensors in one launch (fixed order: deterministic).
// table[i] = {src (device address of copy 0), dst (device address), length (floats), copies, stride between copies (floats)}
// The copies are summed in index order (0, 1, 2, ...: the order never depends on the launch geometry), but EIGHT of them are requested
// before the first is consumed: with one load in flight per thread the kernel was a chain of `copies` (up to 32) dependent memory
// latencies — 52 us per launch for 72 MB, 1.4 TB/s.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const long long* __restrict__ table) {
    const long long* d = table + 5 * (long long)blockIdx.y;
    const float* src = reinterpret_cast<const float*>(d[0]);
    float* dst = reinterpret_cast<float*>(d[1]);
    const long long len = d[2], stride = d[4];
    const int copies = (int)d[3];
    const long long nv = len >> 2;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
        const f32x4* col = reinterpret_cast<const f32x4*>(src) + v;            // copy c of this vector: col[c * stride / 4] (stride % 4 == 0)
        const long long sv = stride >> 2;
        f32x4 a = col[0];
        int c = 1;
        for (; c + 8 <= copies; c += 8) {
            f32x4 b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = __builtin_nontemporal_load(col + (long long)(c + k) * sv);
#pragma unroll
            for (int k = 0; k < 8; ++k) a += b[k];
        }
        if (c + 4 <= copies) {
            f32x4 b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] = __builtin_nontemporal_load(col + (long long)(c + k) * sv);
#pragma unroll
            for (int k = 0; k < 4; ++k) a += b[k];
            c += 4;
        }
        for (; c < copies; ++c) a += __builtin_nontemporal_load(col + (long long)c * sv);
        reinterpret_cast<f32x4*>(dst)[v] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (len & 3)) {
        const long long i = (nv << 2) + threadIdx.x;
        float a = src[i];
        for (int c = 1; c < copies; ++c) a += src[(long long)c * stride + i];
        dst[i] = a;
    }
}
extern "C" int ddpm_wgrad_reduce(const long long* table, int n_tensors, void* stream) {
    if (!table) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(128, n_tensors), dim3(256), 0, (hipStream_t)stream, table);
    return check_launch();
}

// dst_i = a_i (+ b_i) for many small fp32 tensors in one launch (the concatenated time-bias projection: every
// ResidualBlock.fc weight into its rows of one matrix, fc.bias + conv1.bias into one vector — both are added at the same
// place, unet.py:85-86).  table[i] = {a, b (or 0), dst, numel} as int64; grid = (blocks, n_tensors).
__global__ void mt_gather_kernel(const long long* __restrict__ table) {
    const long long* d = table + 4 * (long long)blockIdx.y;
    const float* a = reinterpret_cast<const float*>(d[0]);
    const float* b = reinterpret_cast<const float*>(d[1]);
    float* dst = reinterpret_cast<float*>(d[2]);
    const long long n = d[3];
    // 16-byte vectors when the three addresses allow it (they do for every fc.weight block: 10 MB moved at the head of each step — with
    // dword accesses and 8 blocks per tensor this launch took 30 us, 0.7 TB/s)
    if (((d[0] | d[1] | d[2]) & 15) == 0) {
        const long long nv = n >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
            f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
            if (b) v += reinterpret_cast<const f32x4*>(b)[i];
            reinterpret_cast<f32x4*>(dst)[i] = v;
        }
        for (long long i = (nv << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
            dst[i] = b ? a[i] + b[i] : a[i];
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = b ? a[i] + b[i] : a[i];
}
extern "C" int ddpm_mt_gather_f32(const long long* table, int n_tensors, void* stream) {
    if (!table) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;
    hipLaunchKernelGGL(mt_gather_kernel, dim3(32, n_tensors), dim3(256), 0, (hipStream_t)stream, table);
    return check_launch();
}

// ------------------------------------------------------------------ diffusion algebra (fp32, per-sample coefficients gathered by t)
// q_sample: x_t = a[t]*x0 + b[t]*noise   (diffusion.py:92-97); separate roundings like the reference's op chain
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const long long* __restrict__ t,
                                const float* __restrict__ ca, const float* __restrict__ cb, float* __restrict__ xt, int B, int n, int T) {
    const long long tot = (long long)B * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const long long tt = t[b];
        // a timestep outside the table (torch.gather raises in the reference, diffusion.py:83) cannot raise here: poison the sample
        xt[i] = (unsigned long long)tt < (unsigned long long)T ? __fadd_rn(__fmul_rn(ca[tt], x0[i]), __fmul_rn(cb[tt], noise[i])) : __builtin_nanf("");
    }
}
extern "C" int ddpm_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ab, const float* sqrt_1mab,
                             float* xt, int B, int n, int T, void* stream) {
    if (!x0 || !noise || !t || !sqrt_ab || !sqrt_1mab || !xt) return DDPM_ERR_NULL;
    if (B <= 0 || n <= 0 || T <= 0) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for((long long)B * n)), dim3(256), 0, (hipStream_t)stream, x0, noise, t, sqrt_ab, sqrt_1mab, xt, B, n, T);
    return check_launch();
}

// per-sample mean of (target - pred)^2 (diffusion.py:239, functions.py:99-101): one block per sample; 1024 threads, 16-byte loads, four
// pairs in flight per thread (256 threads with one scalar load in flight took 326 us per 256 x 256 sample — B = 2 on the CelebA-HQ step —
// and a split over several blocks would need scratch memory the ABI does not hand over)
__global__ __launch_bounds__(1024) void mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ loss, int n) {
    __shared__ float sh[16];
    const int b = blockIdx.x;
    const float* p = pred + (long long)b * n;
    const float* t = target + (long long)b * n;
    float acc = 0.f;
    const bool al = ((reinterpret_cast<unsigned long long>(p) | reinterpret_cast<unsigned long long>(t)) & 15) == 0;
    const int nv = al ? n >> 2 : 0;
    int i = threadIdx.x;
    for (; i + 3 * (int)blockDim.x < nv; i += 4 * blockDim.x) {
        f32x4 a[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = reinterpret_cast<const f32x4*>(t)[i + u * blockDim.x]; c[u] = reinterpret_cast<const f32x4*>(p)[i + u * blockDim.x]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const f32x4 d = a[u] - c[u]; acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
    }
    for (; i < nv; i += blockDim.x) { const f32x4 d = reinterpret_cast<const f32x4*>(t)[i] - reinterpret_cast<const f32x4*>(p)[i]; acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
    for (int j = (nv << 2) + threadIdx.x; j < n; j += blockDim.x) { const float d = t[j] - p[j]; acc += d * d; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sh[w];
        loss[b] = tot / (float)n;
    }
}
__global__ void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ gloss,
                               float* __restrict__ gpred, int B, int n) {
    const long long tot = (long long)B * n;
    const float inv = 2.0f / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        gpred[i] = (pred[i] - target[i]) * inv * gloss[b];
    }
}
extern "C" int ddpm_mse_fwd(const float* pred, const float* target, float* loss, int B, int n, void* stream) {
    if (!pred || !target || !loss) return DDPM_ERR_NULL;
    if (B <= 0 || n <= 0) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(B), dim3(n >= 16384 ? 1024 : 256), 0, (hipStream_t)stream, pred, target, loss, n);
    return check_launch();
}
extern "C" int ddpm_mse_bwd(const float* pred, const float* target, const float* gloss, float* gpred, int B, int n, void* stream) {
    if (!pred || !target || !gloss || !gpred) return DDPM_ERR_NULL;
    if (B <= 0 || n <= 0) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for((long long)B * n)), dim3(256), 0, (hipStream_t)stream, pred, target, gloss, gpred, B, n);
    return check_launch();
}

// ------------------------------------------------------------------ variational-bound term L_t in bits per dimension (loss_type = "kl",
// GaussianDiffusion.calc_all_bpd): /root/reference/ddpm_torch/diffusion.py:203-215 `_loss_term_bpd` with functions.py:30-36 `normal_kl`,
// :39-46 `approx_std_normal_cdf`, :49-65 `discretized_gaussian_loglik`.  Per sample b (one block): the model mean follows from the
// network output as in p_mean_var (diffusion.py:107-138), then
//   t[b] > 0 : mean over the elements of KL(q(x_{t-1} | x_t, x_0) || p(x_{t-1} | x_t)) / ln 2       (both variances are table entries)
//   t[b] = 0 : mean of -log(Phi((x_0 - mu + 1/255) / sigma) - Phi((x_0 - mu - 1/255) / sigma)) / ln 2, tanh approximation of Phi, open
//              tails beyond |x_0| = 0.999, the difference floored as the reference floors it (clamp(d - 1e-12, 0) + 1e-12).
// The backward (d loss_b / d model_out, training: clip_denoised = False) differentiates exactly those expressions.
struct VlbTables { const float *recip, *recip_m1, *coef1, *coef2, *post_logvar, *model_logvar; };
struct VlbCoef { float recip, recip_m1, c1, c2, lv1, lv2; };
__device__ __forceinline__ float vlb_cdf(float z) {
    return 0.5f * (1.0f + tanhf(0.7978845608028654f * (z + 0.044715f * z * z * z)));
}
__device__ __forceinline__ float vlb_cdf_grad(float z) {            // d vlb_cdf / dz
    const float th = tanhf(0.7978845608028654f * (z + 0.044715f * z * z * z));
    return 0.5f * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * z * z);
}
// model mean and x_0 estimate from the network output (mean_type 0 eps | 1 x_0 | 2 mean)
__device__ __forceinline__ void vlb_model_mean(const VlbCoef& k, int mean_type, int clip, float xt, float o, float& mean, float& x0p) {
    if (mean_type == 0) x0p = __fsub_rn(__fmul_rn(k.recip, xt), __fmul_rn(k.recip_m1, o));
    else if (mean_type == 1) x0p = o;
    else x0p = __fsub_rn(__fdiv_rn(o, k.c1), __fmul_rn(__fdiv_rn(k.c2, k.c1), xt));
    if (clip) x0p = x0p != x0p ? x0p : fminf(fmaxf(x0p, -1.f), 1.f);
    mean = mean_type == 2 ? o : __fadd_rn(__fmul_rn(k.c1, x0p), __fmul_rn(k.c2, xt));
}
__global__ __launch_bounds__(256) void vlb_terms_kernel(const float* __restrict__ x0, const float* __restrict__ xt, const float* __restrict__ out,
                                                        const long long* __restrict__ t, VlbTables tb, float* __restrict__ loss,
                                                        float* __restrict__ pred_x0, int n, int mean_type, int clip, int T) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const long long tt = t[b];
    const long long base = (long long)b * n;
    if ((unsigned long long)tt >= (unsigned long long)T) {               // index outside the tables: poison instead of reading out of bounds
        if (threadIdx.x == 0) loss[b] = __builtin_nanf("");
        if (pred_x0) for (int i = threadIdx.x; i < n; i += blockDim.x) pred_x0[base + i] = __builtin_nanf("");
        return;
    }
    const VlbCoef k{tb.recip[tt], tb.recip_m1[tt], tb.coef1[tt], tb.coef2[tt], tb.post_logvar[tt], tb.model_logvar[tt]};
    const float dlv = k.lv1 - k.lv2, e_dlv = expf(dlv), inv_var2 = expf(-k.lv2), inv_std = expf(-0.5f * k.lv2);
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = x0[base + i], xti = xt[base + i];
        float mean, x0p;
        vlb_model_mean(k, mean_type, clip, xti, out[base + i], mean, x0p);
        if (pred_x0) pred_x0[base + i] = x0p;
        float term;
        if (tt > 0) {
            const float m1 = __fadd_rn(__fmul_rn(k.c1, x), __fmul_rn(k.c2, xti)), d = m1 - mean;
            term = 0.5f * ((-1.0f - dlv) + d * d * inv_var2 + e_dlv);
        } else {
            const float xc = x - mean;
            const float cu = x > 0.999f ? 1.0f : vlb_cdf(inv_std * (xc + (1.0f / 255.0f)));
            const float cl = x < -0.999f ? 0.0f : vlb_cdf(inv_std * (xc - (1.0f / 255.0f)));
            term = -logf(fmaxf(cu - cl - 1e-12f, 0.f) + 1e-12f);
        }
        acc += term;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[b] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)n / 0.6931471805599453f;
}
__global__ void vlb_terms_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ xt, const float* __restrict__ out,
                                     const long long* __restrict__ t, VlbTables tb, const float* __restrict__ gloss, float* __restrict__ gout,
                                     int B, int n, int mean_type, int T) {
    const long long tot = (long long)B * n;
    const float inv = 1.0f / ((float)n * 0.6931471805599453f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const long long tt = t[b];
        if ((unsigned long long)tt >= (unsigned long long)T) { gout[i] = __builtin_nanf(""); continue; }
        const VlbCoef k{tb.recip[tt], tb.recip_m1[tt], tb.coef1[tt], tb.coef2[tt], tb.post_logvar[tt], tb.model_logvar[tt]};
        const float x = x0[i], xti = xt[i];
        float mean, x0p;
        vlb_model_mean(k, mean_type, 0, xti, out[i], mean, x0p);
        float dterm_dmean;
        if (tt > 0) {
            const float m1 = __fadd_rn(__fmul_rn(k.c1, x), __fmul_rn(k.c2, xti));
            dterm_dmean = -(m1 - mean) * expf(-k.lv2);
        } else {
            const float inv_std = expf(-0.5f * k.lv2), xc = x - mean;
            const float zu = inv_std * (xc + (1.0f / 255.0f)), zl = inv_std * (xc - (1.0f / 255.0f));
            const float cu = x > 0.999f ? 1.0f : vlb_cdf(zu), cl = x < -0.999f ? 0.0f : vlb_cdf(zl);
            const float p = cu - cl - 1e-12f;
            // d(cu - cl) / d mean = -inv_std * (cdf'(zu) - cdf'(zl)) (a constant tail contributes nothing); the floor cuts the gradient
            const float dp = -inv_std * ((x > 0.999f ? 0.f : vlb_cdf_grad(zu)) - (x < -0.999f ? 0.f : vlb_cdf_grad(zl)));
            dterm_dmean = p > 0.f ? -dp / (p + 1e-12f) : 0.f;
        }
        const float dmean_dout = mean_type == 0 ? -k.c1 * k.recip_m1 : (mean_type == 1 ? k.c1 : 1.0f);
        gout[i] = gloss[b] * inv * dterm_dmean * dmean_dout;
    }
}
extern "C" int ddpm_vlb_terms(const float* x_0, const float* x_t, const float* model_out, const long long* t,
                              const float* sqrt_recip_ab, const float* sqrt_recip_m1_ab, const float* post_coef1, const float* post_coef2,
                              const float* post_logvar, const float* model_logvar, float* loss, float* pred_x0,
                              int B, int n, int mean_type, int clip, int T, void* stream) {
    if (!x_0 || !x_t || !model_out || !t || !sqrt_recip_ab || !sqrt_recip_m1_ab || !post_coef1 || !post_coef2 || !post_logvar || !model_logvar || !loss)
        return DDPM_ERR_NULL;
    if (B <= 0 || n <= 0 || T <= 0 || mean_type < 0 || mean_type > 2) return DDPM_ERR_SHAPE;
    VlbTables tb{sqrt_recip_ab, sqrt_recip_m1_ab, post_coef1, post_coef2, post_logvar, model_logvar};
    hipLaunchKernelGGL(vlb_terms_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x_0, x_t, model_out, t, tb, loss, pred_x0, n, mean_type, clip, T);
    return check_launch();
}
extern "C" int ddpm_vlb_terms_bwd(const float* x_0, const float* x_t, const float* model_out, const long long* t,
                                  const float* sqrt_recip_ab, const float* sqrt_recip_m1_ab, const float* post_coef1, const float* post_coef2,
                                  const float* post_logvar, const float* model_logvar, const float* gloss, float* gout,
                                  int B, int n, int mean_type, int T, void* stream) {
    if (!x_0 || !x_t || !model_out || !t || !sqrt_recip_ab || !sqrt_recip_m1_ab || !post_coef1 || !post_coef2 || !post_logvar || !model_logvar || !gloss || !gout)
        return DDPM_ERR_NULL;
    if (B <= 0 || n <= 0 || T <= 0 || mean_type < 0 || mean_type > 2) return DDPM_ERR_SHAPE;
    VlbTables tb{sqrt_recip_ab, sqrt_recip_m1_ab, post_coef1, post_coef2, post_logvar, model_logvar};
    hipLaunchKernelGGL(vlb_terms_bwd_kernel, dim3(grid_for((long long)B * n)), dim3(256), 0, (hipStream_t)stream, x_0, x_t, model_out, t, tb, gloss, gout,
                       B, n, mean_type, T);
    return check_launch();
}

// out = sum_i x[i] * w[i] by ONE block in a fixed order (bit-reproducible): the batch mean of the per-sample losses
// (utils/train.py:151, `loss.mean()`), with w = d(mean)/d(loss_b) = 1/B — the same vector the backward kernel consumes.
__global__ void weighted_sum_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int n) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += x[i] * w[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
extern "C" int ddpm_weighted_sum_f32(const float* x, const float* w, float* out, int n, void* stream) {
    if (!x || !w || !out) return DDPM_ERR_NULL;
    if (n <= 0) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, w, out, n);
    return check_launch();
}

// C[m][n] = sum_k A[k][m] * B[k][n]  (fp32; A is [K][lda], B is [K][ldb], C is [M][ldc]) for SHORT reductions — K = the batch (128):
// the weight gradients of the time-embedding path, dW = d(out)^T in: the 22 ResidualBlock.fc projections (per all-reduce chunk), UNet.embed's
// two Linears (ddpm_torch/models/unet.py:77,122-126 through autograd).  The generic GEMM spends ~34 us on each of these (both operands
// k-strided, four K-steps of work behind its whole prologue); here a block owns a 64 x 64 tile of C, stages 32 rows of A and B at a time
// in LDS and every thread accumulates a 4 x 4 patch: the products are microseconds.  Plain stores, fixed order: bit-deterministic.
__global__ __launch_bounds__(256) void atb_f32_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ Bm, long long ldb,
                                                      float* __restrict__ C, long long ldc, int M, int N, int K) {
    __shared__ float sa[32][64 + 4], sb[32][64 + 4];
    const int tid = threadIdx.x, tm = tid >> 4, tn = tid & 15;            // 16 x 16 threads, 4 x 4 outputs each
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        for (int e = tid; e < 32 * 64; e += 256) {                         // coalesced along m / n
            const int kk = e >> 6, c = e & 63, k = k0 + kk;
            sa[kk][c] = (k < K && m0 + c < M) ? A[(long long)k * lda + m0 + c] : 0.f;
            sb[kk][c] = (k < K && n0 + c < N) ? Bm[(long long)k * ldb + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sa[kk][tm * 4 + i]; b[i] = sb[kk][tn * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + tm * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn * 4 + j;
            if (n < N) C[(long long)m * ldc + n] = acc[i][j];
        }
    }
}
// K <= 128 with 16-byte-aligned rows (every product of the training step): the WHOLE reduction of a 32 x 64 tile is fetched in one go — all
// of a thread's 16-byte loads are in flight together, one memory latency per block instead of one per 32 rows — and twice the blocks
// (the kernel above spent 30 us on 67 MFLOP: four load -> barrier -> multiply rounds on 64 CUs).  Same k-ascending fmaf chain per output:
// bit-identical results.
__global__ __launch_bounds__(256) void atb_f32_short_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ Bm, long long ldb,
                                                            float* __restrict__ C, long long ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float sa[128][32], sb[128][64];
    const int tid = threadIdx.x, tm = tid >> 4, tn = tid & 15;            // 16 x 16 threads, 2 x 4 outputs each
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 64;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 va[4], vb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                           // A tile: K rows x 8 vectors
        const int e = tid + 256 * i, kk = e >> 3, c = (e & 7) * 4;
        va[i] = (kk < K && m0 + c < M) ? *reinterpret_cast<const f32x4*>(A + (long long)kk * lda + m0 + c) : zero;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                           // B tile: K rows x 16 vectors
        const int e = tid + 256 * i, kk = e >> 4, c = (e & 15) * 4;
        vb[i] = (kk < K && n0 + c < N) ? *reinterpret_cast<const f32x4*>(Bm + (long long)kk * ldb + n0 + c) : zero;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; *reinterpret_cast<f32x4*>(&sa[e >> 3][(e & 7) * 4]) = va[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int e = tid + 256 * i; *reinterpret_cast<f32x4*>(&sb[e >> 4][(e & 15) * 4]) = vb[i]; }
    __syncthreads();
    float acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < K; ++kk) {
        const float a0 = sa[kk][tm * 2], a1 = sa[kk][tm * 2 + 1];
        const f32x4 b = *reinterpret_cast<const f32x4*>(&sb[kk][tn * 4]);
        acc[0][0] = fmaf(a0, b.x, acc[0][0]); acc[0][1] = fmaf(a0, b.y, acc[0][1]); acc[0][2] = fmaf(a0, b.z, acc[0][2]); acc[0][3] = fmaf(a0, b.w, acc[0][3]);
        acc[1][0] = fmaf(a1, b.x, acc[1][0]); acc[1][1] = fmaf(a1, b.y, acc[1][1]); acc[1][2] = fmaf(a1, b.z, acc[1][2]); acc[1][3] = fmaf(a1, b.w, acc[1][3]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + tm * 2 + i, n = n0 + tn * 4;
        if (m < M && n < N) *reinterpret_cast<f32x4*>(C + (long long)m * ldc + n) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    }
}
extern "C" int ddpm_atb_f32(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M, int N, int K, void* stream) {
    if (!a || !b || !c) return DDPM_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || lda < M || ldb < N || ldc < N) return DDPM_ERR_SHAPE;
    static const bool no_short = getenv("DDPM_ATB_NO_SHORT") != nullptr;
    const bool vec = ((M | N | (int)(lda & 3) | (int)(ldb & 3) | (int)(ldc & 3)) & 3) == 0 && aligned16(a) && aligned16(b) && aligned16(c);
    if (K <= 128 && vec && !no_short)
        hipLaunchKernelGGL(atb_f32_short_kernel, dim3((N + 63) / 64, (M + 31) / 32), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, c, ldc, M, N, K);
    else
        hipLaunchKernelGGL(atb_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, c, ldc, M, N, K);
    return check_launch();
}

// One fused sampling step for model_mean_type in {eps, x_0, mean} with a fixed variance table
// (diffusion.py:107-158): pred_x0 -> clamp -> posterior mean -> + 1[t>0]*exp(0.5*logvar)*z.
// tab = 7 fp32 tables of length T, concatenated: recip, recip_m1, coef1, coef2, logvar, (unused), (unused)
struct StepTables { const float *recip, *recip_m1, *coef1, *coef2, *logvar; };
__global__ void p_step_kernel(const float* __restrict__ x_t, const float* __restrict__ out, const float* __restrict__ z,
                              const long long* __restrict__ t, StepTables tb, float* __restrict__ x_prev, float* __restrict__ pred_x0,
                              int B, int n, int mean_type, int clip, int T) {
    const long long tot = (long long)B * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const long long tt = t[b];
        if ((unsigned long long)tt >= (unsigned long long)T) {          // index outside the tables: poison instead of reading out of bounds
            x_prev[i] = __builtin_nanf("");
            if (pred_x0) pred_x0[i] = __builtin_nanf("");
            continue;
        }
        const float xt = x_t[i], o = out[i];
        float x0, mean;
        if (mean_type == 0) {            // eps (diffusion.py:145-148)
            x0 = __fsub_rn(__fmul_rn(tb.recip[tt], xt), __fmul_rn(tb.recip_m1[tt], o));
        } else if (mean_type == 1) {     // x_0
            x0 = o;
        } else {                         // mean (diffusion.py:140-143)
            const float c1 = tb.coef1[tt], c2 = tb.coef2[tt];
            x0 = __fsub_rn(__fdiv_rn(o, c1), __fmul_rn(__fdiv_rn(c2, c1), xt));
        }
        if (clip) x0 = x0 != x0 ? x0 : fminf(fmaxf(x0, -1.f), 1.f);      // torch.clamp propagates NaN (a diverged model must stay visible)
        if (mean_type == 2) mean = o;
        else mean = __fadd_rn(__fmul_rn(tb.coef1[tt], x0), __fmul_rn(tb.coef2[tt], xt));
        const float mask = tt > 0 ? 1.f : 0.f;
        const float sd = expf(__fmul_rn(0.5f, tb.logvar[tt]));
        x_prev[i] = __fadd_rn(mean, __fmul_rn(__fmul_rn(mask, sd), z[i]));
        if (pred_x0) pred_x0[i] = x0;
    }
}
extern "C" int ddpm_p_sample_step(const float* x_t, const float* model_out, const float* z, const long long* t,
                                  const float* sqrt_recip_ab, const float* sqrt_recip_m1_ab, const float* post_coef1, const float* post_coef2,
                                  const float* logvar, float* x_prev, float* pred_x0, int B, int n, int mean_type, int clip, int T, void* stream) {
    if (!x_t || !model_out || !z || !t || !x_prev || !sqrt_recip_ab || !sqrt_recip_m1_ab || !post_coef1 || !post_coef2 || !logvar) return DDPM_ERR_NULL;
    if (B <= 0 || n <= 0 || T <= 0 || mean_type < 0 || mean_type > 2) return DDPM_ERR_SHAPE;
    StepTables tb{sqrt_recip_ab, sqrt_recip_m1_ab, post_coef1, post_coef2, logvar};
    hipLaunchKernelGGL(p_step_kernel, dim3(grid_for((long long)B * n)), dim3(256), 0, (hipStream_t)stream, x_t, model_out, z, t, tb, x_prev, pred_x0, B, n, mean_type, clip, T);
    return check_launch();
}

// t[b] = map[t_idx[b]] (DDIM sub-sequence, ddim.py:101) and in-place decrement for graph-captured loops
__global__ void gather_i64_kernel(const long long* __restrict__ idx, const long long* __restrict__ map, long long* __restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) out[i] = map[idx[i]];
}
__global__ void add_i64_kernel(long long* __restrict__ t, int B, long long delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) t[i] += delta;
}
// out[r][:] = table[idx[r]][:] (fp32 rows of row_len floats, a multiple of 4): the sampler's per-step time biases out of the table
// precomputed for every timestep (all samples of a sampling step share one t; the table replaces the embedding MLP's six launches).
// An index outside [0, table_rows) poisons the row with NaN, as ddpm_p_sample_step does for t.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx, float* __restrict__ out,
                                                          int row_len, int table_rows) {
    const long long t = idx[blockIdx.x];
    const bool ok = t >= 0 && t < table_rows;
    const f32x4* src = reinterpret_cast<const f32x4*>(table + (ok ? t : 0) * row_len);
    f32x4* dst = reinterpret_cast<f32x4*>(out + (long long)blockIdx.x * row_len);
    const f32x4 poison = (f32x4)(__builtin_nanf(""));
    for (int i = threadIdx.x; i < row_len / 4; i += 256) dst[i] = ok ? src[i] : poison;
}
extern "C" int ddpm_gather_rows_f32(const float* table, const long long* idx, float* out, int rows, int row_len, int table_rows, void* stream) {
    if (!table || !idx || !out) return DDPM_ERR_NULL;
    if (rows <= 0 || row_len <= 0 || row_len % 4 || table_rows <= 0) return DDPM_ERR_SHAPE;
    if (!aligned16(table) || !aligned16(out)) return DDPM_ERR_ALIGN;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, table, idx, out, row_len, table_rows);
    return check_launch();
}
extern "C" int ddpm_gather_i64(const long long* idx, const long long* map, long long* out, int B, void* stream) {
    if (!idx || !map || !out) return DDPM_ERR_NULL;
    hipLaunchKernelGGL(gather_i64_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, map, out, B);
    return check_launch();
}
extern "C" int ddpm_add_i64(long long* t, int B, long long delta, void* stream) {
    if (!t) return DDPM_ERR_NULL;
    hipLaunchKernelGGL(add_i64_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, B, delta);
    return check_launch();
}

// ------------------------------------------------------------------ SiLU on the (tiny, fp32) time-embedding path
__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = siluf_(x[i]);
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long long n, int accumulate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float g = dy[i] * silu_gradf_(x[i]);
        dx[i] = accumulate ? dx[i] + g : g;
    }
}
extern "C" int ddpm_silu_fwd(const float* x, float* y, long long n, void* stream) {
    if (!x || !y) return DDPM_ERR_NULL;
    hipLaunchKernelGGL(silu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch();
}
extern "C" int ddpm_silu_bwd(const float* x, const float* dy, float* dx, long long n, int accumulate, void* stream) {
    if (!x || !dy || !dx) return DDPM_ERR_NULL;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, accumulate);
    return check_launch();
}

// ------------------------------------------------------------------ column sums of an NHWC gradient (bias / time-bias grads)
// per_sample[b][c] += sum_p dy[b][p][c]   and/or   total[c] += sum_{b,p} dy   (both fp32 atomics into zero-initialised
// buffers; grid = (pixel slabs, B)).  Every block ends with one atomic per channel on the SAME C addresses of `total`, and
// same-address atomics serialise at ~27 ns each: with ~1024 small blocks the kernel took 28 us whatever the tensor size.
// Hence few (~256) large blocks of 1024 threads: 4x fewer atomics per address, still 16 waves per CU for the streaming.
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ dy, long long ld, float* __restrict__ per_sample, long long ps_ld,
                              float* __restrict__ total, int HW, int C, int pix_per_slab) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh[8192];
    const int b = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    const int p0 = blockIdx.x * pix_per_slab, p1 = min(HW, p0 + pix_per_slab);
    // wide tensors (the [B][sum Cout] time-bias gradient: 4992 columns): blockIdx.z walks groups of blockDim.x channel vectors in ONE launch
    // (it used to be one launch per 256 vectors — five 5-us launches in a row on the tail of the backward)
    const int c_base = blockIdx.z * blockDim.x * VEC;
    dy += c_base;
    if (per_sample) per_sample += c_base;
    if (total) total += c_base;
    C = min(C - c_base, (int)blockDim.x * VEC);
    const bool live = cx * VEC < C;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    const T* base = dy + (long long)b * HW * ld + cx * VEC;
    if (live) {
#pragma unroll 4
        for (int p = p0 + py; p < p1; p += blockDim.y) {
            float f[VEC];
            Elem<T>::unpack(ldg16(base + (long long)p * ld), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += f[j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) sh[py * C + cx * VEC + j] = acc[j];          // [PY][C] scratch, PY*C <= 8192
    }
    __syncthreads();
    for (int c = t; c < C; c += nt) {
        float a = 0.f;
        for (int r = 0; r < (int)blockDim.y; ++r) a += sh[r * C + c];
        if (per_sample) atomicAdd(per_sample + (long long)b * ps_ld + c, a);
        if (total) atomicAdd(total + c, a);
    }
}
extern "C" int ddpm_colsum(const void* dy, long long ld, float* per_sample, long long ps_ld, float* total, int B, int HW, int C, int dtype, void* stream) {
    if (!dy || (!per_sample && !total)) return DDPM_ERR_NULL;
    const int vec = dtype == DDPM_BF16 ? 8 : 4;
    if (C % vec || ld % vec || C / vec > 256 * 64) return DDPM_ERR_SHAPE;
    if (!aligned16(dy)) return DDPM_ERR_ALIGN;
    const int Z = (C / vec + 255) / 256;               // channel groups of <= 256 vectors (blockIdx.z)
    const int cv = Z > 1 ? 256 : C / vec;
    int py = 1024 / cv; if (py > HW) py = HW; if (py < 1) py = 1;
    int S = (256 + B - 1) / B;                        // ~256 blocks in total, >= 4 pixel iterations each
    const int maxS = (HW + 4 * py - 1) / (4 * py);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    const int pps = (HW + S - 1) / S;
    S = (HW + pps - 1) / pps;
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(S, B, Z), dim3(cv, py), 0, (hipStream_t)stream, (const bf16_t*)dy, ld, per_sample, ps_ld, total, HW, C, pps);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(colsum_kernel<float>, dim3(S, B, Z), dim3(cv, py), 0, (hipStream_t)stream, (const float*)dy, ld, per_sample, ps_ld, total, HW, C, pps);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// ------------------------------------------------------------------ 2x resampling without a convolution (resample_with_conv=False:
// /root/reference/ddpm_torch/models/unet.py:169 nn.AvgPool2d(2), :196 nn.Upsample(2, "nearest") alone) and the two backwards, one kernel:
//   up = 0: y[b,h,w,c] (+)= scale * sum of the 2x2 block of x at (2h.., 2w..)   — AvgPool2d(2) forward (scale 1/4); Upsample backward (scale 1)
//   up = 1: y[b,2h+a,2w+bb,c] (+)= scale * x[b,h,w,c]                            — Upsample forward (scale 1); AvgPool2d(2) backward (scale 1/4)
// H, W = the SMALL grid; both tensors NHWC with pixel pitches (they live inside the pitched concat buffers of the decoder).
template <typename T>
__global__ void resample2x_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y, long long y_ld, int B, int H, int W, int C, int up,
                                  float scale, int accumulate) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long n = (long long)B * H * W * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        long long r = i / cv;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const long long b = r / H;
        const long long small = (b * H + h) * W + w, big = ((b * 2 * H + 2 * h) * 2 * W) + 2 * w;
        if (!up) {
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    float f[VEC];
                    Elem<T>::unpack(ldg16(x + (big + (long long)a * 2 * W + bb) * x_ld + c), f);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] += f[j];
                }
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] *= scale;
            T* o = y + small * y_ld + c;
            if (accumulate) {
                float f[VEC];
                Elem<T>::unpack(ldg16(o), f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += f[j];
            }
            stg16(o, Elem<T>::pack(acc));
        } else {
            float v[VEC];
            Elem<T>::unpack(ldg16(x + small * x_ld + c), v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] *= scale;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    T* o = y + (big + (long long)a * 2 * W + bb) * y_ld + c;
                    float out[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) out[j] = v[j];
                    if (accumulate) {
                        float f[VEC];
                        Elem<T>::unpack(ldg16(o), f);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) out[j] += f[j];
                    }
                    stg16(o, Elem<T>::pack(out));
                }
        }
    }
}
extern "C" int ddpm_resample2x_nhwc(const void* x, long long x_ld, void* y, long long y_ld, int B, int H, int W, int C, int up, float scale,
                                    int accumulate, int dtype, void* stream) {
    if (!x || !y) return DDPM_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (up != 0 && up != 1)) return DDPM_ERR_SHAPE;
    const int vec = dtype == DDPM_BF16 ? 8 : 4;
    if (dtype != DDPM_BF16 && dtype != DDPM_F32) return DDPM_ERR_DTYPE;
    if (C % vec || x_ld % vec || y_ld % vec || x_ld < C || y_ld < C) return DDPM_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(y)) return DDPM_ERR_ALIGN;
    const int g = grid_for((long long)B * H * W * (C / vec));
    if (dtype == DDPM_BF16)
        hipLaunchKernelGGL(resample2x_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_ld, (bf16_t*)y, y_ld, B, H, W, C, up, scale, accumulate);
    else
        hipLaunchKernelGGL(resample2x_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_ld, (float*)y, y_ld, B, H, W, C, up, scale, accumulate);
    return check_launch();
}

// ------------------------------------------------------------------ backward of nearest-2x upsample: dx[y][x] = sum of the 2x2 block
template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dyu, T* __restrict__ dx, int B, int H, int W, int C, long long dx_ld, int accumulate) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long n = (long long)B * H * W * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        long long r = i / cv;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); const long long b = r / H;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                float f[VEC];
                Elem<T>::unpack(ldg16(dyu + (((b * 2 * H + 2 * y + dy) * 2 * W) + 2 * x + dxx) * C + c), f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += f[j];
            }
        T* o = dx + ((b * H + y) * W + x) * dx_ld + c;
        if (accumulate) {
            float f[VEC];
            Elem<T>::unpack(ldg16(o), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += f[j];
        }
        stg16(o, Elem<T>::pack(acc));
    }
}
extern "C" int ddpm_upsample2x_bwd(const void* dy_up, void* dx, long long dx_ld, int B, int H, int W, int C, int accumulate, int dtype, void* stream) {
    if (!dy_up || !dx) return DDPM_ERR_NULL;
    const int vec = dtype == DDPM_BF16 ? 8 : 4;
    if (C % vec || dx_ld % vec) return DDPM_ERR_SHAPE;
    const int g = grid_for((long long)B * H * W * (C / vec));
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy_up, (bf16_t*)dx, B, H, W, C, dx_ld, accumulate);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy_up, (float*)dx, B, H, W, C, dx_ld, accumulate);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// ------------------------------------------------------------------ y (+)= x over NHWC slices with pitches (gradient fan-in)
template <typename T>
__global__ void add_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y, long long y_ld, long long rows, int C, int accumulate) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long n = rows * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC; const long long r = i / cv;
        float a[VEC];
        Elem<T>::unpack(ldg16(x + r * x_ld + c), a);
        if (accumulate) {
            float b[VEC];
            Elem<T>::unpack(ldg16(y + r * y_ld + c), b);
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[j] += b[j];
        }
        stg16(y + r * y_ld + c, Elem<T>::pack(a));
    }
}
extern "C" int ddpm_add_rows(const void* x, long long x_ld, void* y, long long y_ld, long long rows, int C, int accumulate, int dtype, void* stream) {
    if (!x || !y) return DDPM_ERR_NULL;
    const int vec = dtype == DDPM_BF16 ? 8 : 4;
    if (C % vec || x_ld % vec || y_ld % vec) return DDPM_ERR_SHAPE;
    const int g = grid_for(rows * (C / vec));
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_ld, (bf16_t*)y, y_ld, rows, C, accumulate);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_ld, (float*)y, y_ld, rows, C, accumulate);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// ------------------------------------------------------------------ attention softmax rows (unet.py:47-49) — one wave per row
// fwd: P = softmax(S) with S fp32 logits already scaled by 1/sqrt(C);  bwd: dS = P * (dP - sum(dP*P))
template <typename T>
__global__ void softmax_fwd_kernel(const float* __restrict__ s, T* __restrict__ p, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * L;
    float m = -INFINITY;
    for (int i = lane; i < L; i += 64) m = fmaxf(m, sr[i]);
    m = wave_max(m);
    float sum = 0.f;
    for (int i = lane; i < L; i += 64) sum += __expf(sr[i] - m);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < L; i += 64) Elem<T>::st(p + row * L + i, __expf(sr[i] - m) * inv);
}
template <typename T>
__global__ void softmax_bwd_kernel(const T* __restrict__ p, const float* __restrict__ dp, T* __restrict__ ds, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int i = lane; i < L; i += 64) dot += Elem<T>::ld(p + row * L + i) * dp[row * L + i];
    dot = wave_sum(dot);
    for (int i = lane; i < L; i += 64) {
        const float pv = Elem<T>::ld(p + row * L + i);
        Elem<T>::st(ds + row * L + i, pv * (dp[row * L + i] - dot));
    }
}
extern "C" int ddpm_softmax_fwd(const float* s, void* p, long long rows, int L, int dtype, void* stream) {
    if (!s || !p) return DDPM_ERR_NULL;
    const int g = (int)((rows + 3) / 4);
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, rows, L);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(softmax_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, s, (float*)p, rows, L);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}
extern "C" int ddpm_softmax_bwd(const void* p, const float* dp, void* ds, long long rows, int L, int dtype, void* stream) {
    if (!p || !dp || !ds) return DDPM_ERR_NULL;
    const int g = (int)((rows + 3) / 4);
    if (dtype == DDPM_BF16) hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, dp, (bf16_t*)ds, rows, L);
    else if (dtype == DDPM_F32) hipLaunchKernelGGL(softmax_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)p, dp, (float*)ds, rows, L);
    else return DDPM_ERR_DTYPE;
    return check_launch();
}

// ------------------------------------------------------------------ test hook: the dropout keep-mask the GN kernels regenerate
__global__ void dropout_mask_kernel(float* __restrict__ mask, long long n, unsigned long long seed, unsigned thresh16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mask[i] = dropout_keep(seed, (unsigned long long)i, thresh16) ? 1.f : 0.f;
}
extern "C" int ddpm_dropout_mask(float* mask, long long n, float p, unsigned long long seed, void* stream) {
    if (!mask) return DDPM_ERR_NULL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, seed, dropout_thresh16(p));
    return check_launch();
}
