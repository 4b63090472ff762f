// Fused GroupNorm(32, eps) [+ SiLU] [+ dropout] forward / backward over NHWC activations (gfx950).
// Reference semantics: nn.GroupNorm(32, C, eps=1e-6) -> SiLU -> Dropout (ddpm_torch/models/unet.py:18-20,85-87).
//
// Algorithmic traffic: one read of x + one write of y per element (backward: x, dy read, dx written).  Three families of kernels:
//   staged (gn_lds_fwd / gn_lds_bwd / gn_lds_bwd8): ONE launch, a block owns (sample, chunk of groups) = all pixels x <= 64 KiB of
//       channels; the x slice is DMA'd into LDS, statistics are taken while it lands, the result is produced from LDS.  512 threads
//       for the 16-64 KiB slices, 256 for the <= 8 KiB ones (the 8x8 / 4x4 levels).  Every tensor of the CIFAR / CelebA-64 nets at
//       B = 128 goes through here; they are VALU-bound (exp, rcp, the dropout hash), not memory-bound — see DESIGN.md;
//   streaming (gn_stats + gn_stats_finalize + gn_apply; gn_bwd_reduce + gn_bwd_finalize + gn_bwd_apply): slices that do not fit
//       (the 256 x 256 ... 64 x 64 tensors of CelebA-HQ at B = 2): per-slab partial moments /
//       channel sums, finished once per sample in a fixed order, then one streaming pass; the second read of x is an L2 / Infinity
//       Cache hit at these sizes;
//   register-resident (gn_reg_fwd / gn_reg_bwd): the first single-launch form, kept as the fallback for geometries the staged plan
//       declines.
// SiLU and the dropout mask are recomputed in the backward (one hash word per two channels).
#include "common.h"
#include <stdlib.h>

extern "C" int ddpm_colsum(const void* dy, long long ld, float* per_sample, long long ps_ld, float* total, int B, int HW, int C, int dtype, void* stream);

constexpr int GN_THREADS = 256;
constexpr int GN_MAXC = 2048;

template <int I> struct GnIC { static constexpr int v = I; };
template <int N, typename F>
__device__ __forceinline__ void static_for_gn(F&& f) {
    if constexpr (N > 0) { static_for_gn<N - 1>(f); f(GnIC<N - 1>{}); }
}
// s_waitcnt vmcnt(N + extra): `extra` plain loads were issued AFTER the DMA instructions being counted (runtime 0 or NV)
template <int N>
__device__ __forceinline__ void gn_wait_vm(int extra) {
    if (extra == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    else if (extra == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 1) : "memory");
    else if (extra == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 2) : "memory");
    else if (extra == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 8) : "memory");
}

struct GnShape {
    int B, HW, C, G, cpg;       // cpg = C / G
    long long x_ld, y_ld;       // pixel pitches (elements)
    int S;                      // pixel slabs per sample
    int pix_per_slab;
};

// Pivot of group g of sample b: the group's first stored element.  The streaming kernels accumulate sum(x - k) and
// sum((x - k)^2) instead of raw moments: with k within a few standard deviations of the mean the single-pass variance keeps
// its accuracy when |mean| >> std (raw fp32 moments lose (mean/std)^2 * 1e-7 of relative precision).
template <typename T>
__device__ __forceinline__ float gn_pivot(const T* __restrict__ x, long long x_ld, int HW, int b, int g, int cpg) {
    return Elem<T>::ld(x + (long long)b * HW * x_ld + g * cpg);
}

// thread (cx, py): channel-vector cx fixed, pixels py, py+PY, ...  (blockDim = (CV, PY))
template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ x, GnShape s, float* __restrict__ partial /*[B][S][G][2]*/) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh_sum[GN_MAXC], sh_sq[GN_MAXC];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int cx = threadIdx.x, py = threadIdx.y, PY = blockDim.y;
    const int p0 = slab * s.pix_per_slab, p1 = min(s.HW, p0 + s.pix_per_slab);
    float sum[VEC], sq[VEC], piv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sum[j] = sq[j] = 0.f; piv[j] = gn_pivot(x, s.x_ld, s.HW, b, (cx * VEC + j) / s.cpg, s.cpg); }
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
#pragma unroll 4
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float d = f[j] - piv[j]; sum[j] += d; sq[j] += d * d; }
    }
    // ordered (deterministic) cross-thread reduction: row py of the [PY][C] scratch, then a fixed-order column sum
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sh_sum[py * s.C + cx * VEC + j] = sum[j]; sh_sq[py * s.C + cx * VEC + j] = sq[j]; }
    __syncthreads();
    for (int c = threadIdx.y * blockDim.x + threadIdx.x; c < s.C; c += blockDim.x * blockDim.y) {
        float a = 0.f, q = 0.f;
        for (int r = 0; r < PY; ++r) { a += sh_sum[r * s.C + c]; q += sh_sq[r * s.C + c]; }
        sh_sum[c] = a; sh_sq[c] = q;
    }
    __syncthreads();
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < s.G) {
        float a = 0.f, q = 0.f;
        for (int c = t * s.cpg; c < (t + 1) * s.cpg; ++c) { a += sh_sum[c]; q += sh_sq[c]; }
        float* o = partial + (((long long)b * s.S + slab) * s.G + t) * 2;
        o[0] = a; o[1] = q;
    }
}

struct GnApply {
    const float* gamma; const float* beta;
    float eps; int silu;
    float drop_p; unsigned thresh16; unsigned long long seed;   // dropout after SiLU (unet.py:87); p = 0 disables
    const unsigned long long* seed_dev;                         // optional device word added to `seed` (hipGraph replays: the per-step part of the seed lives in memory)
    float* stats;              // [B][G][2] (mean, rstd) saved for backward, or null
};

__device__ __forceinline__ unsigned long long gn_seed(const GnApply& a) {
    return (a.drop_p > 0.f && a.seed_dev) ? a.seed + a.seed_dev[0] : a.seed;
}

template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, GnShape s, const float* __restrict__ partial, GnApply a) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh_a[GN_MAXC], sh_b[GN_MAXC];
    __shared__ float sh_mean[64], sh_rstd[64];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    // `partial` = the FINISHED statistics [B][G][2] (mean, rstd) written by gn_stats_finalize_kernel (every block used to re-sum
    // the S per-slab partials of its sample here, serially, before touching a pixel: at 256 x 256 with S = 256 that was most of the kernel)
    if (t < s.G) { sh_mean[t] = partial[((long long)b * s.G + t) * 2]; sh_rstd[t] = partial[((long long)b * s.G + t) * 2 + 1]; }
    __syncthreads();
    for (int c = t; c < s.C; c += nt) {
        const int g = c / s.cpg;
        const float sc = sh_rstd[g] * a.gamma[c];
        sh_a[c] = sc; sh_b[c] = a.beta[c] - sh_mean[g] * sc;
    }
    __syncthreads();
    const int cx = threadIdx.x, py = threadIdx.y, PY = blockDim.y;
    const int p0 = slab * s.pix_per_slab, p1 = min(s.HW, p0 + s.pix_per_slab);
    float ca[VEC], cb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { ca[j] = sh_a[cx * VEC + j]; cb[j] = sh_b[cx * VEC + j]; }
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
    T* yb = y + ((long long)b * s.HW) * s.y_ld + cx * VEC;
    const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned long long seed = gn_seed(a);
#pragma unroll 4
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float z = f[j] * ca[j] + cb[j];
            if (a.silu) z = siluf_(z);
            if (a.drop_p > 0.f) {
                unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + cx * VEC + j;
                z = dropout_keep(seed, idx, a.thresh16) ? z * keep_scale : 0.f;
            }
            f[j] = z;
        }
        stg16(yb + (long long)p * s.y_ld, Elem<T>::pack(f));
    }
}

// Statistics of the streaming path, finished ONCE per sample: (mean, rstd) of every group from the S per-slab partial moments, summed in
// a fixed order (lane l takes slabs l, l + 64, ...; the 64 lane sums meet in a fixed butterfly): deterministic like the partials.
// final[b][g] = (mean, rstd); `stats` (optional) receives a copy for the backward.
// Neither finishing kernel uses LDS (shuffles do): they are 64-thread blocks that fit wherever a wave slot is free.  (They were
// suspected of queueing for LDS behind the side stream's weight-gradient blocks — 3 us alone, 27 us average on the CelebA-HQ step;
// the real cause was that those blocks take a CU's whole register file: see the block budget in wgrad.hip.)
__device__ __forceinline__ double gn_wave_sum(double v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename T>
__global__ __launch_bounds__(64) void gn_stats_finalize_kernel(const T* __restrict__ x, GnShape s, const float* __restrict__ partial /*[B][S][G][2]*/,
                                                                float eps, float* __restrict__ fin /*[B][G][2]*/, float* __restrict__ stats) {
    const int g = blockIdx.x, b = blockIdx.y, l = threadIdx.x;
    double sum = 0.0, sq = 0.0;
#pragma unroll 4
    for (int i = l; i < s.S; i += 64) {
        const float2 o = *reinterpret_cast<const float2*>(partial + (((long long)b * s.S + i) * s.G + g) * 2);
        sum += o.x; sq += o.y;
    }
    sum = gn_wave_sum(sum); sq = gn_wave_sum(sq);
    if (l == 0) {
        const double n = (double)s.HW * s.cpg;
        const double dmean = sum / n;                       // moments of (x - pivot): see gn_pivot
        double var = sq / n - dmean * dmean;
        if (var < 0.0) var = 0.0;
        const float mean = (float)((double)gn_pivot(x, s.x_ld, s.HW, b, g, s.cpg) + dmean);
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        fin[((long long)b * s.G + g) * 2] = mean; fin[((long long)b * s.G + g) * 2 + 1] = rstd;
        if (stats) { stats[((long long)b * s.G + g) * 2] = mean; stats[((long long)b * s.G + g) * 2 + 1] = rstd; }
    }
}

// Backward twin: per-(sample, channel) sums A1 = sum dz*xhat, A2 = sum dz over the S slabs, in a fixed order; also the dgamma / dbeta
// contribution of the sample (atomics, one per (b, c)).  grid = (ceil(C / CL), B), ONE wave = CL channels (8 CL contiguous bytes per
// slab) x 64 / CL slab lanes, eight loads in flight per thread; the slab lanes meet in a butterfly over the upper lane bits.
// CL = 16 while a lane gets <= 16 slabs (fewest, widest blocks: B = 128 means S = 24 and C / 4 x 128 single-wave blocks otherwise),
// 4 for the long reductions of the big-image / small-batch tensors (S = 512).
template <int CL>
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(GnShape s, const float* __restrict__ partial /*[B][S][C][2]*/, float* __restrict__ fin /*[B][C][2]*/,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int KL = 64 / CL;
    const int b = blockIdx.y, t = threadIdx.x, cl = t % CL, k = t / CL, c = blockIdx.x * CL + cl;
    float a1 = 0.f, a2 = 0.f;
    if (c < s.C) {
        const float2* col = reinterpret_cast<const float2*>(partial) + (long long)b * s.S * s.C + c;
        int i = k;
        for (; i + 7 * KL < s.S; i += 8 * KL) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(long long)(i + KL * u) * s.C];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a1 += v[u].x; a2 += v[u].y; }
        }
        for (; i < s.S; i += KL) { const float2 v = col[(long long)i * s.C]; a1 += v.x; a2 += v.y; }
    }
#pragma unroll
    for (int off = CL; off < 64; off <<= 1) { a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64); }
    if (t < CL && c < s.C) {
        fin[((long long)b * s.C + c) * 2] = a1; fin[((long long)b * s.C + c) * 2 + 1] = a2;
        if (dgamma) atomicAdd(dgamma + c, a1);
        if (dbeta) atomicAdd(dbeta + c, a2);
    }
}

// ---- per-channel / per-group constants of a thread's vector column, fetched with as few VMEM instructions as possible: an
// instruction occupies the CU's address path for 16 clk whatever its width, and with 16 resident waves the 24-32 scalar loads
// these constants used to cost (one per element for gamma, beta, mean, rstd) took longer than the data loads themselves.
// group of channel c: a shift for the power-of-two group widths (everything but the 384-channel tensors)
__device__ __forceinline__ int gn_gidx(int c, int cpg) { return (cpg & (cpg - 1)) == 0 ? c >> (__ffs(cpg) - 1) : c / cpg; }
typedef float gn_f32x4 __attribute__((ext_vector_type(4)));
typedef float gn_f32x2 __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ void gn_ld_channels(const float* __restrict__ p, bool ok, float (&o)[VEC]) {
#pragma unroll
    for (int q = 0; q < VEC; q += 4) {
        gn_f32x4 v = (gn_f32x4)(0.f);
        if (ok) v = *reinterpret_cast<const gn_f32x4*>(p + q);
        o[q] = v.x; o[q + 1] = v.y; o[q + 2] = v.z; o[q + 3] = v.w;
    }
}
// one load per GROUP the vector touches (1 when cpg is a multiple of the vector width, VEC / cpg when it divides it)
template <int VEC, typename L>
__device__ __forceinline__ void gn_per_group(int c_first, int cpg, bool ok, L&& at_group_start) {
    const bool p2 = (cpg & (cpg - 1)) == 0;
    const int sh = __ffs(cpg) - 1;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int c = c_first + e;
        const bool first = e == 0 || (p2 ? (c & (cpg - 1)) == 0 : c % cpg == 0);
        at_group_start(e, first && ok, p2 ? c >> sh : c / cpg);
    }
}
template <int VEC>
__device__ __forceinline__ void gn_group_stats(const float* __restrict__ stats, long long b_G, int c_first, int cpg, bool ok, float (&mean)[VEC], float (&rstd)[VEC]) {
    gn_f32x2 cur; cur.x = 0.f; cur.y = 1.f;
    gn_per_group<VEC>(c_first, cpg, ok, [&](int e, bool load, int g) {
        if (load) cur = *reinterpret_cast<const gn_f32x2*>(stats + (b_G + g) * 2);
        mean[e] = cur.x; rstd[e] = cur.y;
    });
}

// ---- single-launch kernels: one block owns (sample b, a chunk of GPB groups) = all HW pixels x seg_ch = GPB*cpg channels,
// and keeps that slice (<= 64 KiB) IN REGISTERS: thread (j, prow) holds the 16-byte vector column j of pixels prow,
// prow + R, ... (NV of them, all requested before the first is used).  Statistics and group coefficients come from an
// ordered LDS reduction of per-thread partials (deterministic), the output is produced from the registers: exactly one
// HBM read of every input and one write of the output, one launch, no workspace.
struct GnFused {
    int GPB, seg_ch, seg_vecs;     // groups per block, channels / 16-byte vectors per pixel segment
    int nt;                        // threads per block (256 or 512: a block holds up to nt * 16 vectors = 64 / 128 KiB)
    int nta;                       // active threads: largest multiple of seg_vecs <= nt (a thread keeps one vector column)
    int rows_per_iter;             // R = nta / seg_vecs pixels per sweep of the block
    int nv;                        // vectors per thread = ceil(HW / R)
    int xcd_remap;                 // LDS kernels: keep the channel chunks of a sample on one XCD (see gn_block_slice)
};

// Ordered block reduction of per-thread channel partials: part[e] belongs to channel j*VEC + e of the segment.  On return
// sh_ch[c] (c < seg_ch) holds the sum over all threads; sh_row is scratch of rows_per_iter * seg_ch floats.
template <int VEC>
__device__ __forceinline__ void gn_block_channel_sum(const float (&part)[VEC], const GnFused& f, bool active, int j, int prow, int tid,
                                                     float* sh_row, float* sh_ch) {
    __syncthreads();                                  // previous users of the scratch are done
    if (active) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh_row[prow * f.seg_ch + j * VEC + e] = part[e];
    }
    __syncthreads();
    for (int c = tid; c < f.seg_ch; c += f.nt) {
        float acc = 0.f;
        for (int r = 0; r < f.rows_per_iter; ++r) acc += sh_row[r * f.seg_ch + c];
        sh_ch[c] = acc;
    }
    __syncthreads();
}

template <typename T, int NV, int NT>
__global__ __launch_bounds__(NT)
void gn_reg_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, GnShape s, GnFused f, GnApply a) {
    constexpr int VEC = Elem<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float gsh[];
    float* sh_row = gsh;                                              // [rows_per_iter][seg_ch]
    float* sh_ch = sh_row + f.rows_per_iter * f.seg_ch;               // [seg_ch]
    float* sh_mean = sh_ch + f.seg_ch;                                // [GPB]
    float* sh_rstd = sh_mean + 32;                                    // [GPB]
    const int b = blockIdx.y, c0 = blockIdx.x * f.seg_ch, tid = threadIdx.x;
    const bool active = tid < f.nta;
    const int j = tid % f.seg_vecs, prow = tid / f.seg_vecs;
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + c0 + j * VEC;
    u32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        v[i] = (active && p < s.HW) ? ldg16(xb + (long long)p * s.x_ld) : zero16();
    }
    const float n = (float)s.HW * (float)s.cpg;
    // mean
    float part[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float fv[VEC];
        Elem<T>::unpack(v[i], fv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) part[e] += fv[e];              // padded pixels hold zeros
    }
    gn_block_channel_sum<VEC>(part, f, active, j, prow, tid, sh_row, sh_ch);
    if (tid < f.GPB) {
        float acc = 0.f;
        for (int c = tid * s.cpg; c < (tid + 1) * s.cpg; ++c) acc += sh_ch[c];
        sh_mean[tid] = acc / n;
    }
    __syncthreads();
    float mean[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) mean[e] = sh_mean[gn_gidx(j * VEC + e, s.cpg)];
    // centred variance
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        if (p < s.HW) {
            float fv[VEC];
            Elem<T>::unpack(v[i], fv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = fv[e] - mean[e]; part[e] += d * d; }
        }
    }
    gn_block_channel_sum<VEC>(part, f, active, j, prow, tid, sh_row, sh_ch);
    if (tid < f.GPB) {
        float acc = 0.f;
        for (int c = tid * s.cpg; c < (tid + 1) * s.cpg; ++c) acc += sh_ch[c];
        const float rstd = 1.0f / sqrtf(acc / n + a.eps);
        sh_rstd[tid] = rstd;
        if (a.stats) {
            const long long gi = (long long)b * s.G + blockIdx.x * f.GPB + tid;
            a.stats[gi * 2] = sh_mean[tid]; a.stats[gi * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    if (!active) return;
    float ca[VEC], cb[VEC];
    gn_ld_channels<VEC>(a.gamma + c0 + j * VEC, true, ca);
    gn_ld_channels<VEC>(a.beta + c0 + j * VEC, true, cb);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        ca[e] *= sh_rstd[gn_gidx(j * VEC + e, s.cpg)];
        cb[e] -= mean[e] * ca[e];
    }
    const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned long long seed = gn_seed(a);
    T* yb = y + ((long long)b * s.HW) * s.y_ld + c0 + j * VEC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        if (p >= s.HW) break;
        float fv[VEC];
        Elem<T>::unpack(v[i], fv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float z = fv[e] * ca[e] + cb[e];
            if (a.silu) z = siluf_(z);
            if (a.drop_p > 0.f) {
                const unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + c0 + j * VEC + e;
                z = dropout_keep(seed, idx, a.thresh16) ? z * keep_scale : 0.f;
            }
            fv[e] = z;
        }
        stg16(yb + (long long)p * s.y_ld, Elem<T>::pack(fv));
    }
}

// backward twin: x and dy slices in registers; pass 1 per-channel sums A1 = sum dz*xhat, A2 = sum dz -> dgamma / dbeta
// (atomics across samples) and the group coefficients; pass 2 dx = rstd * (dz*gamma - xhat*c1 - c2) from the registers.
template <typename T, int NV, int NT>
__global__ __launch_bounds__(NT)
void gn_reg_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, GnShape s, GnFused f, long long dy_ld,
                       long long dx_ld, const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta, GnApply a,
                       int accumulate, const T* __restrict__ addp, long long add_ld, float* __restrict__ dx_colsum, long long colsum_ld) {
    constexpr int VEC = Elem<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float gsh[];
    float* sh_row = gsh;
    float* sh_ch = sh_row + f.rows_per_iter * f.seg_ch;
    float* sh_c1 = sh_ch + f.seg_ch;
    float* sh_c2 = sh_c1 + 32;
    const int b = blockIdx.y, c0 = blockIdx.x * f.seg_ch, tid = threadIdx.x;
    const bool active = tid < f.nta;
    const int j = tid % f.seg_vecs, prow = tid / f.seg_vecs;
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + c0 + j * VEC;
    const T* db = dy + ((long long)b * s.HW) * dy_ld + c0 + j * VEC;
    u32x4 vx[NV], vd[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        const bool ok = active && p < s.HW;
        vx[i] = ok ? ldg16(xb + (long long)p * s.x_ld) : zero16();
        vd[i] = ok ? ldg16(db + (long long)p * dy_ld) : zero16();
    }
    float mean[VEC], rstd[VEC], gm[VEC], bt[VEC];
    gn_group_stats<VEC>(stats, (long long)b * s.G, c0 + j * VEC, s.cpg, active, mean, rstd);
    gn_ld_channels<VEC>(a.gamma + c0 + j * VEC, active, gm);
    gn_ld_channels<VEC>(a.beta + c0 + j * VEC, active, bt);
    const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned long long seed = gn_seed(a);
    auto dz_of = [&](int p, int e, float xh, float d) -> float {
        float dz = d;
        if (a.drop_p > 0.f) {
            const unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + c0 + j * VEC + e;
            dz = dropout_keep(seed, idx, a.thresh16) ? dz * keep_scale : 0.f;
        }
        if (a.silu) dz *= silu_gradf_(gm[e] * xh + bt[e]);
        return dz;
    };
    float a1[VEC], a2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a1[e] = a2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        if (p < s.HW) {
            float fx[VEC], fd[VEC];
            Elem<T>::unpack(vx[i], fx); Elem<T>::unpack(vd[i], fd);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = (fx[e] - mean[e]) * rstd[e];
                const float dz = dz_of(p, e, xh, fd[e]);
                a1[e] += dz * xh; a2[e] += dz;
            }
        }
    }
    const float inv_n = 1.0f / ((float)s.HW * s.cpg);
    gn_block_channel_sum<VEC>(a1, f, active, j, prow, tid, sh_row, sh_ch);
    for (int c = tid; c < f.seg_ch; c += NT) {
        if (dgamma) atomicAdd(dgamma + c0 + c, sh_ch[c]);
        sh_ch[c] *= a.gamma[c0 + c];
    }
    __syncthreads();
    if (tid < f.GPB) {
        float acc = 0.f;
        for (int c = tid * s.cpg; c < (tid + 1) * s.cpg; ++c) acc += sh_ch[c];
        sh_c1[tid] = acc * inv_n;
    }
    gn_block_channel_sum<VEC>(a2, f, active, j, prow, tid, sh_row, sh_ch);
    for (int c = tid; c < f.seg_ch; c += NT) {
        if (dbeta) atomicAdd(dbeta + c0 + c, sh_ch[c]);
        sh_ch[c] *= a.gamma[c0 + c];
    }
    __syncthreads();
    if (tid < f.GPB) {
        float acc = 0.f;
        for (int c = tid * s.cpg; c < (tid + 1) * s.cpg; ++c) acc += sh_ch[c];
        sh_c2[tid] = acc * inv_n;
    }
    __syncthreads();
    if (!active && !dx_colsum) return;
    float c1[VEC], c2[VEC], cs[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const int g = active ? gn_gidx(j * VEC + e, s.cpg) : 0; c1[e] = sh_c1[g]; c2[e] = sh_c2[g]; cs[e] = 0.f; }
    T* ob = dx + ((long long)b * s.HW) * dx_ld + c0 + j * VEC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        if (p >= s.HW || !active) break;
        float fx[VEC], fd[VEC], o[VEC], ad[VEC];
        Elem<T>::unpack(vx[i], fx); Elem<T>::unpack(vd[i], fd);
        if (accumulate) Elem<T>::unpack(ldg16(ob + (long long)p * dx_ld), o);
        if (addp) Elem<T>::unpack(ldg16(addp + ((long long)b * s.HW + p) * add_ld + c0 + j * VEC), ad);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xh = (fx[e] - mean[e]) * rstd[e];
            const float dz = dz_of(p, e, xh, fd[e]);
            float r = rstd[e] * (dz * gm[e] - xh * c1[e] - c2[e]);
            if (addp) r += ad[e];
            o[e] = accumulate ? o[e] + r : r;
        }
        const u32x4 packed = Elem<T>::pack(o);
        if (dx_colsum) {                                       // sums of the values as STORED (what a column sum over dx would read)
            float q[VEC];
            Elem<T>::unpack(packed, q);
#pragma unroll
            for (int e = 0; e < VEC; ++e) cs[e] += q[e];
        }
        stg16(ob + (long long)p * dx_ld, packed);
    }
    // per-(sample, channel) sums of dx — the time-bias gradient of the block (ddpm_torch/models/unet.py:86) — from the values this block
    // has just produced: the 4^2 / 8^2 levels used to pay a separate column-sum launch (8.5 us, 14 per step) for a few hundred KiB
    if (dx_colsum) {
        gn_block_channel_sum<VEC>(cs, f, active, j, prow, tid, sh_row, sh_ch);
        for (int c = tid; c < f.seg_ch; c += NT) dx_colsum[(long long)b * colsum_ld + c0 + c] = sh_ch[c];      // one owner per (b, c): plain store
    }
}

// ---- single-launch kernels for the big slices: x staged in LDS, one HBM read of every input
// Same slicing as the register kernels — a block owns (sample b, a chunk of GPB groups) over all HW pixels, up to 64 KiB —
// but the x slice is DMA'd straight into LDS (buffer_load ... lds: no VGPRs are spent on it) and, in the backward, only dy
// (<= 8 vectors per thread) sits in registers: ~100 VGPRs and ~66 KiB of LDS, i.e. TWO blocks per CU, so one block's
// load phase overlaps the other's arithmetic.  Thread t owns the 16-byte vectors v = t, t + 512, ... (vector column
// j = t % seg_vecs of pixels prow, prow + R, ...) in every pass, so it only ever reads LDS bytes its own wave fetched: a
// counted vmcnt per vector orders them, pass 1 starts on the first vectors while the rest of the slice is still in flight,
// and no barrier is needed between the passes.
// Block reduction of per-thread channel partials for these kernels: butterfly over the lanes of a wave that hold the same
// vector column (needs seg_vecs | 64), then a fixed-order sum over the 8 waves — 1 KiB of scratch instead of 16 KiB.
// Wave stage for power-of-two seg_vecs: a butterfly that HALVES the payload at every step — the lane whose bit `OFF` is set keeps
// the upper half of its values and sends the lower half to its partner (and vice versa), so 16 values cross the wave in
// 8 + 4 + 2 + 1 exchanges instead of 4 x 16 (the exchanges go through the LDS crossbar; with 16 resident waves they were the
// most expensive part of the statistics).  Each lane ends with the wave totals of the values base .. base + CNT - 1.
template <int CNT, int OFF, typename W>
__device__ __forceinline__ void gn_lane_reduce(float (&v)[CNT], int sv, int lane, int base, W&& write) {
    if constexpr (OFF == 0) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) write(base + k, v[k]);
    } else {
        if (OFF < sv) {
#pragma unroll
            for (int k = 0; k < CNT; ++k) write(base + k, v[k]);
            return;
        }
        if constexpr (CNT > 1) {
            const bool bit = (lane & OFF) != 0;
            float o[CNT / 2];
#pragma unroll
            for (int k = 0; k < CNT / 2; ++k) {
                const float send = bit ? v[k] : v[k + CNT / 2], keep = bit ? v[k + CNT / 2] : v[k];
                o[k] = keep + __shfl_xor(send, OFF, 64);
            }
            gn_lane_reduce<CNT / 2, OFF / 2>(o, sv, lane, base + (bit ? CNT / 2 : 0), write);
        } else {
            v[0] += __shfl_xor(v[0], OFF, 64);
            gn_lane_reduce<1, OFF / 2>(v, sv, lane, base, write);
        }
    }
}

// The same wave stage without the LDS crossbar (gfx950): lanes l and l ^ 32 / l ^ 16 trade HALF their payload with v_permlane32_swap /
// v_permlane16_swap — one swap + one add retire two values, for both partners at once — and the in-row steps (lanes of a 16-lane row
// that hold the same vector column: l, l + sv, l + 2 sv, ...) are v_add_f32 with a row_ror DPP operand.  16 values: 12 swaps + 12 adds,
// then 4 x log2(16 / sv) DPP adds — against 15 ds_bpermute round trips with two selects each.  Lane l ends with the wave totals of
// the values base .. base + CNT' - 1 for its column (base = bit 5 of the lane selects the upper half, bit 4 the upper quarter).
typedef unsigned gn_u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float gn_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CNT, typename W>
__device__ __forceinline__ void gn_lane_reduce_hw(float (&v)[CNT], int sv, int lane, W&& write) {
    static_assert(CNT % 4 == 0, "two halvings");
    float h[CNT / 2];
#pragma unroll
    for (int k = 0; k < CNT / 2; ++k) {
        const gn_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + CNT / 2]), false, false);
        h[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    int base = (lane & 32) ? CNT / 2 : 0;
    if (sv > 16) {                                   // 32 vector columns: rows 0-1 / 2-3 hold different columns, nothing else to add
#pragma unroll
        for (int k = 0; k < CNT / 2; ++k) write(base + k, h[k]);
        return;
    }
    float q[CNT / 4];
#pragma unroll
    for (int k = 0; k < CNT / 4; ++k) {
        const gn_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[k]), __float_as_uint(h[k + CNT / 4]), false, false);
        q[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    base += (lane & 16) ? CNT / 4 : 0;
    if (sv <= 8) {
#pragma unroll
        for (int k = 0; k < CNT / 4; ++k) q[k] = gn_dpp_add<0x128>(q[k]);          // row_ror:8
    }
    if (sv <= 4) {
#pragma unroll
        for (int k = 0; k < CNT / 4; ++k) q[k] = gn_dpp_add<0x124>(q[k]);          // row_ror:4
    }
    if (sv <= 2) {
#pragma unroll
        for (int k = 0; k < CNT / 4; ++k) q[k] = gn_dpp_add<0x122>(q[k]);          // row_ror:2
    }
    if (sv <= 1) {
#pragma unroll
        for (int k = 0; k < CNT / 4; ++k) q[k] = gn_dpp_add<0x121>(q[k]);          // row_ror:1
    }
#pragma unroll
    for (int k = 0; k < CNT / 4; ++k) write(base + k, q[k]);
}

// NARR partial vectors at once (same barriers): sh_row holds NARR x [NW waves][seg_ch], sh_ch NARR x [seg_ch]
// sum over the cpg consecutive lanes of a channel group (cpg a power of two <= 64, groups aligned to it)
__device__ __forceinline__ float gn_group_lanes_sum(float v, int cpg) {
    for (int off = 1; off < cpg; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// first stage alone: per-wave totals into sh_row (no barrier after it)
template <int VEC, int NARR, int NW = 8>
__device__ __forceinline__ void gn_wave_partials(float (&v)[NARR * VEC], const GnFused& f, int j, int tid, float* sh_row) {
    const int lane = tid & 63, wave = tid >> 6;
    if ((f.seg_vecs & (f.seg_vecs - 1)) == 0) {
        if constexpr ((NARR * VEC) % 4 == 0) {
            gn_lane_reduce_hw<NARR * VEC>(v, f.seg_vecs, lane, [&](int gi, float val) {
                const int m = gi / VEC, e = gi - m * VEC;        // VEC is a compile-time power of two
                sh_row[(m * NW + wave) * f.seg_ch + j * VEC + e] = val;      // lanes holding copies of a total store the same value
            });
        } else {
        gn_lane_reduce<NARR * VEC, 32>(v, f.seg_vecs, lane, 0, [&](int gi, float val) {
            const int m = gi / VEC, e = gi - m * VEC;        // VEC is a compile-time power of two
            sh_row[(m * NW + wave) * f.seg_ch + j * VEC + e] = val;      // lanes holding copies of a total store the same value
        });
        }
    } else {
        // lanes l, l + seg_vecs, l + 2 seg_vecs, ... hold the same vector column (3 / 6 / 12 vectors per pixel of the 384-channel
        // tensors): strided tree, after which lanes < seg_vecs hold their column's wave total
        int top = f.seg_vecs;
        while (top * 2 < 64) top *= 2;
        for (int off = top; off >= f.seg_vecs; off >>= 1) {
            const bool in = lane + off < 64;
#pragma unroll
            for (int e = 0; e < NARR * VEC; ++e) { const float o = __shfl_down(v[e], off, 64); v[e] += in ? o : 0.f; }
        }
        if (lane < f.seg_vecs) {
#pragma unroll
            for (int gi = 0; gi < NARR * VEC; ++gi) sh_row[((gi / VEC) * NW + wave) * f.seg_ch + j * VEC + (gi % VEC)] = v[gi];
        }
    }
}
template <int VEC, int NARR, int NW = 8>
__device__ __forceinline__ void gn_block_sum_w(float (&v)[NARR * VEC], const GnFused& f, int j, int tid, float* sh_row, float* sh_ch) {
    __syncthreads();                                  // previous users of the scratch are done
    gn_wave_partials<VEC, NARR, NW>(v, f, j, tid, sh_row);
    __syncthreads();
    for (int c = tid; c < NARR * f.seg_ch; c += NW * 64) {
        const int m = c / f.seg_ch, cc = c - m * f.seg_ch;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += sh_row[(m * NW + w) * f.seg_ch + cc];
        sh_ch[c] = acc;
    }
    __syncthreads();
}
template <int VEC, int NW = 8>
__device__ __forceinline__ void gn_block_channel_sum_w(const float (&part)[VEC], const GnFused& f, bool active, int j, int prow, int tid,
                                                       float* sh_row, float* sh_ch) {
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = active ? part[e] : 0.f;
    gn_block_sum_w<VEC, 1, NW>(v, f, j, tid, sh_row, sh_ch);
}
template <int VEC, int NW = 8>
__device__ __forceinline__ void gn_block_channel_sum2_w(const float (&pa)[VEC], const float (&pb)[VEC], const GnFused& f, bool active, int j, int tid,
                                                        float* sh_row, float* sh_ch) {
    float v[2 * VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { v[e] = active ? pa[e] : 0.f; v[VEC + e] = active ? pb[e] : 0.f; }
    gn_block_sum_w<VEC, 2, NW>(v, f, j, tid, sh_row, sh_ch);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gn_slice_rsrc(const void* base, long long bytes) {
    const unsigned long long ad = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

#ifdef GN_TIMING
__device__ unsigned long long* g_gn_timing = nullptr;          // debug builds only (scripts/gn_timeline.sh): [block][8] stamps
#define GN_STAMP(slot) do { if (g_gn_timing && threadIdx.x == 0) g_gn_timing[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = (slot) == 0 || (slot) == 7 ? wall_clock64() : clock64(); } while (0)
extern "C" int ddpm_debug_set_gn_timing(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_gn_timing), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#else
#define GN_STAMP(slot)
#endif

// Block -> (sample, channel chunk) for the LDS kernels.  The chunks of one sample split every 128-byte line of its rows between
// them; consecutive workgroup ids go round-robin over the 8 XCDs (one L2 each), so with the plain (chunk, sample) grid the parts
// of a line are fetched by different L2s.  Remapped, the chunks of a sample are ids 8 apart: same XCD, dispatched back to back —
// the line comes from HBM once and the partners hit in L2.
__device__ __forceinline__ void gn_block_slice(const GnFused& f, int& b, int& chunk) {
    const int gx = gridDim.x, gy = gridDim.y;
    if ((gy & 7) == 0 && f.xcd_remap) {
        const int L = blockIdx.y * gx + blockIdx.x, k = L >> 3;
        chunk = k % gx;
        b = (k / gx) * 8 + (L & 7);
    } else { b = blockIdx.y; chunk = blockIdx.x; }
}

// ---- the eight-vectors-per-thread backward (the 32 x 32 tensors), kept in its first form: dz in 32 registers, per-channel constants
// in registers, predicated rows.  The restructured kernel below needs ~130 registers at NV = 8 (50-62 spilled, each reload a
// vmcnt(0)); the variant that parks dy / dz in LDS instead (1024 threads, one block per CU) exposes its reduction phases — nothing
// else is resident to run meanwhile — and measured 40.6 / 76.2 / 195.8 us against this kernel's 37.3 / 69.6 / 145.9 us on the
// 32 x 32 x {128, 256, 384} tensors at B = 128 (scripts/gn_bench.py).  Both passes are VALU-bound here (exp, rcp, the dropout
// hash: ~930 clk per 8-channel vector and wave), which is why hiding the load phase under pass 1 changed nothing.
//   backward pass 1: A1[c] = sum dz*xhat, A2[c] = sum dz -> dgamma / dbeta atomics, group coefficients c1, c2
//            pass 2: dx = rstd * (dz*gamma - xhat*c1 - c2) (+= when accumulate), optionally the per-(sample, channel) sums of dx
//                    (the time-bias gradient of the block, ddpm_torch/models/unet.py:86: no separate column-sum launch).
template <typename T, int NV>
__global__ __launch_bounds__(512, 4)
void gn_lds_bwd8_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, GnShape s, GnFused f, long long dy_ld,
                       long long dx_ld, const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta, GnApply a,
                       int accumulate, float* __restrict__ dx_colsum, long long colsum_ld, const T* __restrict__ addp, long long add_ld) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    char* xs = lsm;                                               // [NV][512] vectors of x
    float* sh_row = reinterpret_cast<float*>(lsm + NV * 512 * 16);
    float* sh_ch = sh_row + 16 * f.seg_ch;
    float* sh_c1 = sh_ch + 2 * f.seg_ch;
    float* sh_c2 = sh_c1 + 32;
    float* sh_gam = sh_c2 + 32;                                  // gamma of the block's channels: the finalising threads index it by channel
    GN_STAMP(0); GN_STAMP(1);
    int b, chunk;
    gn_block_slice(f, b, chunk);
    const int c0 = chunk * f.seg_ch, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool active = tid < f.nta;
    const int j = tid % f.seg_vecs, prow = tid / f.seg_vecs;
    const __amdgpu_buffer_rsrc_t rx = gn_slice_rsrc(x + ((long long)b * s.HW) * s.x_ld + c0, ((long long)(s.HW - 1) * s.x_ld + f.seg_ch) * ES);
    // per-channel constants first: ordinary loads issued BEFORE the DMA can be waited for with a counted vmcnt
    float mean[VEC], rstd[VEC], gm[VEC], bt[VEC];
    gn_group_stats<VEC>(stats, (long long)b * s.G, c0 + j * VEC, s.cpg, active, mean, rstd);
    gn_ld_channels<VEC>(a.gamma + c0 + j * VEC, active, gm);
    gn_ld_channels<VEC>(a.beta + c0 + j * VEC, active, bt);
    const float gam_c = tid < f.seg_ch ? a.gamma[c0 + tid] : 0.f;      // (seg_ch <= 512) requested before the DMA, parked in LDS after pass 1
    const T* db = dy + ((long long)b * s.HW) * dy_ld + c0 + j * VEC;
    u32x4 vd[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {                     // issue order x_0, dy_0, x_1, dy_1, ...: pair i is complete at vmcnt(2 (NV-1-i))
        const int p = prow + i * f.rows_per_iter;
        const bool ok = active && p < s.HW;
        const unsigned ox = ok ? (unsigned)(((long long)p * s.x_ld + j * VEC) * ES) : 0x7ffffff0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + (i * 512 + wave * 64) * 16), 16, ox, 0, 0, 0);
        vd[i] = ok ? ldg16(db + (long long)p * dy_ld) : zero16();
    }
    const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned h0 = dropout_h0(gn_seed(a));
    const unsigned idx0 = (unsigned)(((unsigned)b * (unsigned)s.HW) * (unsigned)s.C + (unsigned)(c0 + j * VEC));   // < 2^32: checked by the host
    // dz = dy * dropout mask * silu'(z) for one vector: one hash word per pair of channels
    auto dz_vec = [&](int p, const float (&xh)[VEC], float (&d)[VEC]) {
        if (a.drop_p > 0.f) {
            const unsigned q0 = (idx0 + (unsigned)p * (unsigned)s.C) >> 1;
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const unsigned w = dropout_word32(h0, q0 + (e >> 1));
                d[e] = (w & 0xffffu) >= a.thresh16 ? d[e] * keep_scale : 0.f;
                d[e + 1] = (w >> 16) >= a.thresh16 ? d[e + 1] * keep_scale : 0.f;
            }
        }
        if (a.silu) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] *= silu_grad_fast_(gm[e] * xh[e] + bt[e]);
        }
    };
    float a1[VEC], a2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a1[e] = a2[e] = 0.f;
    const u32x4* myx = reinterpret_cast<const u32x4*>(xs) + tid;
    static_for_gn<NV>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        gn_wait_vm<2 * (NV - 1 - i)>(0);
        const int p = prow + i * f.rows_per_iter;
        if (active && p < s.HW) {
            float fx[VEC], fd[VEC], xh[VEC];
            Elem<T>::unpack(myx[i * 512], fx); Elem<T>::unpack(vd[i], fd);
#pragma unroll
            for (int e = 0; e < VEC; ++e) xh[e] = (fx[e] - mean[e]) * rstd[e];
            dz_vec(p, xh, fd);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { a1[e] += fd[e] * xh[e]; a2[e] += fd[e]; }
            vd[i] = Elem<T>::pack(fd);                         // pass 2 reuses dz (stored at the tensor dtype, as an autograd graph would) instead of redoing the mask and silu'
        }
    });
    GN_STAMP(2);
    if (tid < f.seg_ch) sh_gam[tid] = gam_c;                      // published by the first barrier of the reduction
    const float inv_n = 1.0f / ((float)s.HW * s.cpg);
    gn_block_channel_sum2_w<VEC>(a1, a2, f, active, j, tid, sh_row, sh_ch);        // sh_ch = [A1 | A2]
    for (int c = tid; c < 2 * f.seg_ch; c += 512) {
        const bool second = c >= f.seg_ch;
        const int cc = second ? c - f.seg_ch : c;
        float* dst = second ? dbeta : dgamma;
        if (dst) atomicAdd(dst + c0 + cc, sh_ch[c]);
        sh_ch[c] *= sh_gam[cc];
    }
    __syncthreads();
    if (tid < 2 * f.GPB) {
        const int second = tid >= f.GPB, g = tid - second * f.GPB;
        float acc = 0.f;
        for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) acc += sh_ch[second * f.seg_ch + c];
        (second ? sh_c2 : sh_c1)[g] = acc * inv_n;
    }
    __syncthreads();
    GN_STAMP(3);
    float c1[VEC], c2[VEC], cs[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const int g = active ? gn_gidx(j * VEC + e, s.cpg) : 0; c1[e] = sh_c1[g]; c2[e] = sh_c2[g]; cs[e] = 0.f; }
    T* ob = dx + ((long long)b * s.HW) * dx_ld + c0 + j * VEC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        if (active && p < s.HW) {
            float fx[VEC], fd[VEC], o[VEC];
            Elem<T>::unpack(myx[i * 512], fx); Elem<T>::unpack(vd[i], fd);
            float ad[VEC];
            if (accumulate) Elem<T>::unpack(ldg16(ob + (long long)p * dx_ld), o);
            if (addp) Elem<T>::unpack(ldg16(addp + ((long long)b * s.HW + p) * add_ld + c0 + j * VEC), ad);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = (fx[e] - mean[e]) * rstd[e];
                float r = rstd[e] * (fd[e] * gm[e] - xh * c1[e] - c2[e]);          // fd holds dz here
                if (addp) r += ad[e];
                o[e] = accumulate ? o[e] + r : r;
            }
            const u32x4 packed = Elem<T>::pack(o);
            if (dx_colsum) {                                   // sums of the values as STORED (what a column sum over dx would read)
                float q[VEC];
                Elem<T>::unpack(packed, q);
#pragma unroll
                for (int e = 0; e < VEC; ++e) cs[e] += q[e];
            }
            stg16(ob + (long long)p * dx_ld, packed);
        }
    }
    GN_STAMP(4);
    if (dx_colsum) {
        gn_block_channel_sum_w<VEC>(cs, f, active, j, prow, tid, sh_row, sh_ch);
        for (int c = tid; c < f.seg_ch; c += 512) dx_colsum[(long long)b * colsum_ld + c0 + c] = sh_ch[c];     // one owner per (b, c): plain store
    }
    GN_STAMP(5);
#ifdef GN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GN_STAMP(6); GN_STAMP(7);
}

// ---- LDS accesses of the staged kernels while their DMA is in flight go through inline asm.  hipcc cannot tell which LDS bytes an
// outstanding `buffer_load ... lds` will write, so it puts `s_waitcnt vmcnt(0)` in front of EVERY ds_read / ds_write it generates
// itself after the DMA was issued (disassembly of the first version: the counted waits below were each followed by a vmcnt(0), and
// the 16 spilled bytes of the backward kernel were reloaded with another one) — the whole slice had to land before the first
// vector was touched, and the VALU-heavy first pass (exp, rcp and the dropout hash: ~25 instructions per element) ran after the load
// phase instead of under it.  The asm forms carry no such dependency; a wave only reads LDS bytes it fetched itself, after its own
// counted wait.
__device__ __forceinline__ u32x4 gn_lds_rd16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void gn_lds_wr4(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// the VEC per-channel constants of a thread's vector column from a [seg_ch] float array in LDS
template <int VEC>
__device__ __forceinline__ void gn_lds_rd_consts(unsigned addr, float (&o)[VEC]) {
#pragma unroll
    for (int q = 0; q < VEC; q += 4) {
        const u32x4 v = gn_lds_rd16(addr + q * 4);
        o[q] = __uint_as_float(v.x); o[q + 1] = __uint_as_float(v.y); o[q + 2] = __uint_as_float(v.z); o[q + 3] = __uint_as_float(v.w);
    }
}
// Global accesses of these kernels are buffer instructions with 32-bit offsets into the block's slice: a lane without a pixel gets an
// out-of-range offset (reads return 0, writes are dropped) instead of a branch.  Branch-free matters beyond the saved address
// arithmetic: hipcc's counted waits assume the FEWEST younger VMEM operations over all paths, so loads skipped under `if` turn every
// later wait into vmcnt(0).
constexpr unsigned GN_OOB = 0x7ffffff0u;
__device__ __forceinline__ u32x4 gn_buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void gn_buf_st16(__amdgpu_buffer_rsrc_t r, unsigned off, const u32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
}
// barrier that publishes asm LDS writes without draining the DMA (__syncthreads() would wait for vmcnt(0))
__device__ __forceinline__ void gn_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

//   backward pass 1: A1[c] = sum dz*xhat, A2[c] = sum dz -> dgamma / dbeta atomics, group coefficients c1, c2
//            pass 2: dx = rstd * (dz*gamma - xhat*c1 - c2) (+= when accumulate), optionally the per-(sample, channel) sums of dx
//                    (the time-bias gradient of the block, ddpm_torch/models/unet.py:86: no separate column-sum launch).
// Per-channel constants live in LDS ([8][seg_ch], built by the first seg_ch threads from ONE load each of gamma, beta and the
// group's statistics while the slice is in flight) and a thread fetches the eight of its vector column when a pass needs them:
//   pass 1  z = x*ca + cb (ca = rstd*gamma, cb = beta - mean*ca), sums of dz*(x - mean) [scaled by rstd once per channel] and dz;
//   pass 2  dx = dz*ca + xhat*k2 + k3 (xhat = x*rs + ms, k2 = -rstd*c1, k3 = -rstd*c2) folded to dz*ca + x*q2 + q3 — two fmas per element.
// (NV <= 4; eight vectors per thread run gn_lds_bwd8_kernel above.)  NT = 512 for the 16 - 64 KiB slices, 256 for the small ones
// (<= 8 KiB: the 8 x 8 and 4 x 4 tensors).
// P2 (round 6): cpg is a power of two >= VEC / 2, so the lower and the upper half of a thread's vector column lie in ONE group each:
// mean and the pass-2 coefficients q2, q3 are two scalars per thread instead of VEC-wide arrays — with them the eight-vector form
// (the 32 x 32 tensors) fits 128 registers without spilling and replaces gn_lds_bwd8_kernel for bf16.
// FL >= 0: SiLU (bit 0) and dropout (bit 1) are compile-time facts — straight-line passes (with run-time flags hipcc keeps the unpacked
// operands of a vector alive across the flag branches: 33 spilled registers at NV = 8); FL < 0: read from the arguments.
// FL >= 0: SiLU (bit 0) and dropout (bit 1) are compile-time facts, so that a vector's four element pairs are ONE basic block and their
// dependent chains (exp -> rcp -> fma ...) interleave; FL < 0: read from the arguments (a branch per pair: 64 per pass).
template <typename T, int NV, int NT, bool P2, int FL = -1>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 2)
void gn_lds_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, GnShape s, GnFused f, long long dy_ld,
                       long long dx_ld, const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta, GnApply a,
                       int accumulate, float* __restrict__ dx_colsum, long long colsum_ld, const T* __restrict__ addp, long long add_ld) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T);
    const bool do_silu = FL < 0 ? a.silu != 0 : (FL & 1) != 0, do_drop = FL < 0 ? a.drop_p > 0.f : (FL & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    constexpr int NW = NT / 64;
    char* xs = lsm;                                               // [NV][NT] vectors of x
    float* sh_row = reinterpret_cast<float*>(lsm + NV * NT * 16);
    float* sh_ch = sh_row + 2 * NW * f.seg_ch;
    float* sh_c1 = sh_ch + 2 * f.seg_ch;
    float* sh_c2 = sh_c1 + 32;
    float* sh_k = sh_c2 + 32;                                    // [8][seg_ch]: rs, ms, mean, ca, cb, gamma, q2, q3
    enum { K_RS, K_MS, K_MEAN, K_CA, K_CB, K_GAM, K_K2, K_K3 };
    GN_STAMP(0); GN_STAMP(1);
    int b, chunk;
    gn_block_slice(f, b, chunk);
    const int c0 = chunk * f.seg_ch, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool active = tid < f.nta;
    const int j = tid % f.seg_vecs, prow = tid / f.seg_vecs;
    const unsigned lds0 = (unsigned)(size_t)lsm, k0 = (unsigned)(size_t)sh_k;
    const unsigned kcol = k0 + (unsigned)(j * VEC * 4), kstride = (unsigned)(f.seg_ch * 4);
    const __amdgpu_buffer_rsrc_t rx = gn_slice_rsrc(x + ((long long)b * s.HW) * s.x_ld + c0, ((long long)(s.HW - 1) * s.x_ld + f.seg_ch) * ES);
    // per-channel inputs first: ordinary loads issued BEFORE the DMA complete before it (in-order return), so a counted wait covers them
    // (all threads load, channel index clamped: no branch around VMEM)
    const int kc = c0 + min(tid, f.seg_ch - 1);
    const float in_g = a.gamma[kc], in_b = a.beta[kc];
    const gn_f32x2 in_st = *reinterpret_cast<const gn_f32x2*>(stats + ((long long)b * s.G + gn_gidx(kc, s.cpg)) * 2);
    const __amdgpu_buffer_rsrc_t rdy = gn_slice_rsrc(dy + ((long long)b * s.HW) * dy_ld + c0, ((long long)(s.HW - 1) * dy_ld + f.seg_ch) * ES);
    const __amdgpu_buffer_rsrc_t rdx = gn_slice_rsrc(dx + ((long long)b * s.HW) * dx_ld + c0, ((long long)(s.HW - 1) * dx_ld + f.seg_ch) * ES);
    const __amdgpu_buffer_rsrc_t rad = gn_slice_rsrc(addp ? addp + ((long long)b * s.HW) * add_ld + c0 : x, addp ? ((long long)(s.HW - 1) * add_ld + f.seg_ch) * ES : 0);
    const unsigned h0 = dropout_h0(gn_seed(a));            // (its load, if any, also goes out before the DMA)
    // slice offsets of pixel row prow + i * R: base + i * step (the slices are < 2^31 bytes: checked by the host)
    const unsigned x_o = (unsigned)((prow * s.x_ld + j * VEC) * ES), x_st = (unsigned)(f.rows_per_iter * s.x_ld * ES);
    const unsigned dy_o = (unsigned)((prow * dy_ld + j * VEC) * ES), dy_st = (unsigned)(f.rows_per_iter * dy_ld * ES);
    const unsigned dx_o = (unsigned)((prow * dx_ld + j * VEC) * ES), dx_st = (unsigned)(f.rows_per_iter * dx_ld * ES);
    const unsigned ad_o = (unsigned)((prow * add_ld + j * VEC) * ES), ad_st = (unsigned)(f.rows_per_iter * add_ld * ES);
    auto row_ok = [&](int i) { return active && prow + i * f.rows_per_iter < s.HW; };
    u32x4 vd[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {                     // issue order x_0, dy_0, x_1, dy_1, ...: pair i is complete at vmcnt(2 (NV-1-i))
        const bool ok = row_ok(i);
        unsigned ox = ok ? x_o + i * x_st : GN_OOB, od = ok ? dy_o + i * dy_st : GN_OOB;
        asm volatile("" : "+v"(ox), "+v"(od));          // one DMA + one load per row on every path (hipcc otherwise splits the select into two predicated DMA instructions)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + (i * NT + wave * 64) * 16), 16, ox, 0, 0, 0);
        vd[i] = gn_buf_ld16(rdy, od);
    }
    if (tid < f.seg_ch) {
        const float rs = in_st.y, mean = in_st.x, ca = rs * in_g;
        const unsigned ad = k0 + (unsigned)(tid * 4);
        gn_lds_wr4(ad + K_RS * kstride, rs); gn_lds_wr4(ad + K_MS * kstride, -mean * rs); gn_lds_wr4(ad + K_MEAN * kstride, mean);
        gn_lds_wr4(ad + K_CA * kstride, ca); gn_lds_wr4(ad + K_CB * kstride, in_b - mean * ca); gn_lds_wr4(ad + K_GAM * kstride, in_g);
    }
    gn_lds_barrier();
    constexpr int NM = P2 ? 2 : VEC;                   // distinct means / pass-2 coefficients of a vector column
    constexpr int MS = VEC / NM;                       // element e uses entry e / MS
    float ca[VEC], cb[VEC], mean[NM];
    gn_lds_rd_consts<VEC>(kcol + K_CA * kstride, ca);
    gn_lds_rd_consts<VEC>(kcol + K_CB * kstride, cb);
    if constexpr (P2) {
        float m8[VEC];
        gn_lds_rd_consts<VEC>(kcol + K_MEAN * kstride, m8);
        mean[0] = m8[0]; mean[1] = m8[VEC / 2];
    } else {
        gn_lds_rd_consts<VEC>(kcol + K_MEAN * kstride, mean);
    }
    const float keep_scale = do_drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned idx0 = (unsigned)(((unsigned)b * (unsigned)s.HW) * (unsigned)s.C + (unsigned)(c0 + j * VEC));   // < 2^32: checked by the host
    float a1[VEC], a2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a1[e] = a2[e] = 0.f;
    const unsigned myx_ad = lds0 + (unsigned)(tid * 16);
    static_for_gn<NV>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        // (the wait takes the previous vector's last sums as operands: in straight-line code hipcc otherwise retires ALL waits and LDS
        //  reads first, unpacks every vector and spills the lot — 100 registers at NV = 8)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (NV - 1 - i)), "v"(a1[VEC - 1]), "v"(a2[VEC - 1]) : "memory");
        const int p = prow + i * f.rows_per_iter;
        if constexpr (VEC == 8) {
            // bf16: element PAIRS in 2-wide float vectors from the unpack on, so that the arithmetic is v_pk_fma / v_pk_mul / v_pk_add on
            // register pairs that never have to be assembled (hipcc's own SLP pass packed the scalar form too, but paid for it with
            // ~33 v_mov per vector: 252 instructions per vector, now ~150).  silu'(z) = s + s (z - z s), s = 1 / (1 + 2^(-z log2 e)).
            const u32x4 xv = gn_lds_rd16(myx_ad + i * (NT * 16)), dv = vd[i];
            const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
            unsigned q0 = (idx0 + (unsigned)p * (unsigned)s.C) >> 1;
            asm volatile("" : "+v"(q0));                // (opaque: ordered behind this vector's wait)
            unsigned dzw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gn_f32x2 x2, d2, g2;
                x2.x = __uint_as_float(xw[k] << 16); x2.y = __uint_as_float(xw[k] & 0xffff0000u);
                d2.x = __uint_as_float(dw[k] << 16); d2.y = __uint_as_float(dw[k] & 0xffff0000u);
                gn_f32x2 ca2, cb2, mn2;
                ca2.x = ca[2 * k]; ca2.y = ca[2 * k + 1]; cb2.x = cb[2 * k]; cb2.y = cb[2 * k + 1];
                mn2.x = mean[(2 * k) / MS]; mn2.y = mean[(2 * k + 1) / MS];
                g2.x = g2.y = keep_scale;
                if (do_silu) {
                    const gn_f32x2 z = x2 * ca2 + cb2, nz = z * (-1.4426950408889634f);
                    gn_f32x2 t;
                    t.x = __builtin_amdgcn_exp2f(nz.x); t.y = __builtin_amdgcn_exp2f(nz.y);
                    t = t + 1.0f;
                    gn_f32x2 sg;
                    sg.x = __builtin_amdgcn_rcpf(t.x); sg.y = __builtin_amdgcn_rcpf(t.y);
                    const gn_f32x2 w = z - z * sg;
                    g2 = (sg + sg * w) * keep_scale;
                }
                if (do_drop) {
                    const unsigned wd = dropout_word32(h0, q0 + k);
                    g2.x = (wd & 0xffffu) >= a.thresh16 ? g2.x : 0.f;
                    g2.y = (wd >> 16) >= a.thresh16 ? g2.y : 0.f;
                }
                const gn_f32x2 dz = d2 * g2;
                gn_f32x2 s1, s2;
                s1.x = a1[2 * k]; s1.y = a1[2 * k + 1]; s2.x = a2[2 * k]; s2.y = a2[2 * k + 1];
                s1 = s1 + dz * (x2 - mn2); s2 = s2 + dz;
                a1[2 * k] = s1.x; a1[2 * k + 1] = s1.y; a2[2 * k] = s2.x; a2[2 * k + 1] = s2.y;
                dzw[k] = pack_bf2(dz.x, dz.y);
            }
            // pass 2 reuses dz (stored at the tensor dtype, as an autograd graph would) instead of redoing the mask and silu'
            u32x4 dzp; dzp.x = dzw[0]; dzp.y = dzw[1]; dzp.z = dzw[2]; dzp.w = dzw[3];
            vd[i] = dzp;
        } else
        {                                              // rows without a pixel hold zeros in both operands: dz = 0, nothing is added
            float fx[VEC], fd[VEC];
            Elem<T>::unpack(gn_lds_rd16(myx_ad + i * (NT * 16)), fx); Elem<T>::unpack(vd[i], fd);
            // dz = dy * dropout mask * silu'(z): one hash word per pair of channels
            unsigned q0 = (idx0 + (unsigned)p * (unsigned)s.C) >> 1;
            asm volatile("" : "+v"(q0));                // (opaque, ordered behind this vector's wait: hipcc otherwise computes the hash words of all eight vectors up front and spills them)
#pragma unroll
            for (int h = 0; h < VEC; h += 4) {
                if (do_drop) {
#pragma unroll
                    for (int e = h; e < h + 4; e += 2) {
                        const unsigned w = dropout_word32(h0, q0 + (e >> 1));
                        fd[e] = (w & 0xffffu) >= a.thresh16 ? fd[e] * keep_scale : 0.f;
                        fd[e + 1] = (w >> 16) >= a.thresh16 ? fd[e + 1] * keep_scale : 0.f;
                    }
                }
                if (do_silu) {
#pragma unroll
                    for (int e = h; e < h + 4; ++e) fd[e] *= silu_grad_fast_(fx[e] * ca[e] + cb[e]);
                }
#pragma unroll
                for (int e = h; e < h + 4; ++e) { a1[e] += fd[e] * (fx[e] - mean[e / MS]); a2[e] += fd[e]; }
            }
            // pass 2 reuses dz (stored at the tensor dtype, as an autograd graph would) instead of redoing the mask and silu'
            vd[i] = Elem<T>::pack(fd);
        }
        if constexpr (NV > 4 && VEC == 8) {
            // everything this vector computed passes through one ordered statement: the next vector's wait and LDS read (volatile asm too)
            // cannot be scheduled above it, nor can this vector's arithmetic sink below it
            asm volatile("" : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(a1[4]), "+v"(a1[5]), "+v"(a1[6]), "+v"(a1[7]),
                              "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]), "+v"(vd[i]));
        }
        __builtin_amdgcn_sched_barrier(0);             // one vector at a time: without the branches of the predicated version hipcc hoists the hash words of all eight vectors (70 spilled registers)
    });
    GN_STAMP(2);
    // `add` (the residual gradient that joins dx) is requested now and consumed in pass 2: its latency hides behind the reduction
    u32x4 adn = gn_buf_ld16(rad, addp && row_ok(0) ? ad_o : GN_OOB);
    const float inv_n = 1.0f / ((float)s.HW * s.cpg);
    const bool fast = P2 || ((s.cpg & (s.cpg - 1)) == 0 && s.cpg <= 64);
    if (fast) {
        // two barriers, as in the forward: thread c < seg_ch adds the waves' totals of its channel, emits the parameter gradients, the
        // cpg lanes of a group add up their gamma-weighted sums by shuffles and every lane derives its channel's pass-2 constants
        float v2[2 * VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { v2[e] = a1[e]; v2[VEC + e] = a2[e]; }       // (rows without a pixel contributed zeros)
        __builtin_amdgcn_s_setprio(3);                 // (the serial chain below at raised priority: see gn_lds_fwd_kernel)
        gn_wave_partials<VEC, 2, NW>(v2, f, j, tid, sh_row);
        // (barriers that wait for LDS only: __syncthreads() is s_waitcnt vmcnt(0) first, i.e. here it would wait for the `add` vector just
        //  requested, below for the round trips of the dgamma / dbeta atomics — 128 per address, chip-wide at the same moment — and in the
        //  column sums for every store of pass 2: 4.0 -> 2.x us of a 19-us block, scripts/gn_timeline.sh)
        gn_lds_barrier();
        if (tid < f.seg_ch) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { t1 += sh_row[w * f.seg_ch + tid]; t2 += sh_row[(NW + w) * f.seg_ch + tid]; }
            const float rs = sh_k[K_RS * f.seg_ch + tid], gam = sh_k[K_GAM * f.seg_ch + tid];
            t1 *= rs;                                                   // A1 = rstd * sum dz (x - mean)
            if (dgamma) atomicAdd(dgamma + c0 + tid, t1);
            if (dbeta) atomicAdd(dbeta + c0 + tid, t2);
            const float c1 = gn_group_lanes_sum(t1 * gam, s.cpg) * inv_n, c2 = gn_group_lanes_sum(t2 * gam, s.cpg) * inv_n;
            // dx = dz*ca + xhat*k2 + k3 with xhat = x*rs + ms, k2 = -rstd*c1, k3 = -rstd*c2  ->  dx = dz*ca + x*q2 + q3 (two fmas per element)
            const float k2 = -rs * c1, k3 = -rs * c2;
            sh_k[K_K2 * f.seg_ch + tid] = rs * k2;
            sh_k[K_K3 * f.seg_ch + tid] = sh_k[K_MS * f.seg_ch + tid] * k2 + k3;
        }
        gn_lds_barrier();
        __builtin_amdgcn_s_setprio(0);
    } else {
    gn_block_channel_sum2_w<VEC, NW>(a1, a2, f, active, j, tid, sh_row, sh_ch);        // sh_ch = [sum dz (x - mean) | sum dz]
        for (int c = tid; c < 2 * f.seg_ch; c += NT) {
            const bool second = c >= f.seg_ch;
            const int cc = second ? c - f.seg_ch : c;
            const float v = second ? sh_ch[c] : sh_ch[c] * sh_k[K_RS * f.seg_ch + cc];      // A1 = rstd * sum dz (x - mean)
            float* dst = second ? dbeta : dgamma;
            if (dst) atomicAdd(dst + c0 + cc, v);
            sh_ch[c] = v * sh_k[K_GAM * f.seg_ch + cc];
        }
        __syncthreads();
        if (tid < 2 * f.GPB) {
            const int second = tid >= f.GPB, g = tid - second * f.GPB;
            float acc = 0.f;
            for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) acc += sh_ch[second * f.seg_ch + c];
            (second ? sh_c2 : sh_c1)[g] = acc * inv_n;
        }
        __syncthreads();
        if (tid < f.seg_ch) {
            // dx = dz*ca + xhat*k2 + k3 with xhat = x*rs + ms, k2 = -rstd*c1, k3 = -rstd*c2  ->  dx = dz*ca + x*q2 + q3 (two fmas per element)
            const int g = gn_gidx(tid, s.cpg);
            const float rs = sh_k[K_RS * f.seg_ch + tid], k2 = -rs * sh_c1[g], k3 = -rs * sh_c2[g];
            sh_k[K_K2 * f.seg_ch + tid] = rs * k2;
            sh_k[K_K3 * f.seg_ch + tid] = sh_k[K_MS * f.seg_ch + tid] * k2 + k3;
        }
        __syncthreads();
    }
    GN_STAMP(3);
    float q2[NM], q3[NM], cs[VEC];
#pragma unroll
    for (int e = 0; e < NM; ++e) {
        const int c = j * VEC + e * MS;
        q2[e] = sh_k[K_K2 * f.seg_ch + c]; q3[e] = sh_k[K_K3 * f.seg_ch + c];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) cs[e] = 0.f;
    const u32x4* myx = reinterpret_cast<const u32x4*>(xs) + tid;
    // (opaque copies: hipcc otherwise computes the eight store / load offsets of this pass before pass 1 and parks them in scratch)
    unsigned dx_o2 = dx_o, ad_o2 = ad_o;
    int prow2 = prow;
    asm volatile("" : "+v"(dx_o2), "+v"(ad_o2), "+v"(prow2));
    auto row_ok2 = [&](int i) { return active && prow2 + i * f.rows_per_iter < s.HW; };
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool ok = row_ok2(i);
        const u32x4 adc = adn;
        if (i + 1 < NV) adn = gn_buf_ld16(rad, addp && row_ok2(i + 1) ? ad_o2 + (i + 1) * ad_st : GN_OOB);
        if constexpr (VEC == 8) {
            // bf16: pairs again — dx = dz*ca + x*q2 + q3 is two v_pk_fma per pair
            const u32x4 xv = myx[i * NT], dv = vd[i];
            const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w}, aw[4] = {adc.x, adc.y, adc.z, adc.w};
            u32x4 ov = zero16();
            if (accumulate) ov = gn_buf_ld16(rdx, ok ? dx_o2 + i * dx_st : GN_OOB);
            const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w};
            unsigned pw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gn_f32x2 x2, d2, c2, k2, k3;
                x2.x = __uint_as_float(xw[k] << 16); x2.y = __uint_as_float(xw[k] & 0xffff0000u);
                d2.x = __uint_as_float(dw[k] << 16); d2.y = __uint_as_float(dw[k] & 0xffff0000u);
                c2.x = ca[2 * k]; c2.y = ca[2 * k + 1];
                k2.x = q2[(2 * k) / MS]; k2.y = q2[(2 * k + 1) / MS]; k3.x = q3[(2 * k) / MS]; k3.y = q3[(2 * k + 1) / MS];
                gn_f32x2 r = x2 * k2 + (d2 * c2 + k3);                             // d2 holds dz here
                if (addp) { gn_f32x2 a2v; a2v.x = __uint_as_float(aw[k] << 16); a2v.y = __uint_as_float(aw[k] & 0xffff0000u); r = r + a2v; }
                if (accumulate) { gn_f32x2 o2; o2.x = __uint_as_float(ow[k] << 16); o2.y = __uint_as_float(ow[k] & 0xffff0000u); r = o2 + r; }
                pw[k] = pack_bf2(r.x, r.y);
            }
            u32x4 packed; packed.x = pw[0]; packed.y = pw[1]; packed.z = pw[2]; packed.w = pw[3];
            if (dx_colsum) {                                   // sums of the values as STORED (what a column sum over dx would read)
                if (!ok) packed = zero16();
                const unsigned sw[4] = {packed.x, packed.y, packed.z, packed.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gn_f32x2 q, c;
                    q.x = __uint_as_float(sw[k] << 16); q.y = __uint_as_float(sw[k] & 0xffff0000u);
                    c.x = cs[2 * k]; c.y = cs[2 * k + 1];
                    c = c + q;
                    cs[2 * k] = c.x; cs[2 * k + 1] = c.y;
                }
            }
            gn_buf_st16(rdx, ok ? dx_o2 + i * dx_st : GN_OOB, packed);
        } else {
            float fx[VEC], fd[VEC], o[VEC];
            Elem<T>::unpack(myx[i * NT], fx); Elem<T>::unpack(vd[i], fd);
            float ad[VEC];
            if (accumulate) Elem<T>::unpack(gn_buf_ld16(rdx, ok ? dx_o2 + i * dx_st : GN_OOB), o);
            if (addp) Elem<T>::unpack(adc, ad);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float r = fd[e] * ca[e] + q3[e / MS];                              // fd holds dz here
                r = fx[e] * q2[e / MS] + r;
                if (addp) r += ad[e];
                o[e] = accumulate ? o[e] + r : r;
            }
            u32x4 packed = Elem<T>::pack(o);
            if (dx_colsum) {                                   // sums of the values as STORED (what a column sum over dx would read)
                if (!ok) packed = zero16();
                float q[VEC];
                Elem<T>::unpack(packed, q);
#pragma unroll
                for (int e = 0; e < VEC; ++e) cs[e] += q[e];
            }
            gn_buf_st16(rdx, ok ? dx_o2 + i * dx_st : GN_OOB, packed);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    GN_STAMP(4);
    if (dx_colsum) {
        // (the scratch's earlier readers finished before the barrier that published the pass-2 coefficients; no wait for the stores of pass 2)
        gn_wave_partials<VEC, 1, NW>(cs, f, j, tid, sh_row);
        gn_lds_barrier();
        for (int c = tid; c < f.seg_ch; c += NT) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) acc += sh_row[w * f.seg_ch + c];
            dx_colsum[(long long)b * colsum_ld + c0 + c] = acc;      // one owner per (b, c): plain store
        }
    }
    GN_STAMP(5);
#ifdef GN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GN_STAMP(6); GN_STAMP(7);
}

// forward twin: x slice in LDS; pivot-shifted moments accumulated while the slice lands (one pass), then
// y = drop(silu(x * a_c + b_c)) streamed out.
// FL as in gn_lds_bwd_kernel: >= 0 bakes SiLU (bit 0) / dropout (bit 1) in, < 0 reads them from the arguments.
template <typename T, int NV, int NT, int FL = -1>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 2)
void gn_lds_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, GnShape s, GnFused f, GnApply a) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T), NW = NT / 64;
    const bool do_silu = FL < 0 ? a.silu != 0 : (FL & 1) != 0, do_drop = FL < 0 ? a.drop_p > 0.f : (FL & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    char* xs = lsm;
    float* sh_row = reinterpret_cast<float*>(lsm + NV * NT * 16);
    float* sh_ch = sh_row + 2 * NW * f.seg_ch;
    float* sh_mean = sh_ch + 2 * f.seg_ch;
    float* sh_rstd = sh_mean + 32;
    float* sh_k = sh_rstd + 32;                                  // [8][seg_ch]: gamma, beta, pivot of the channel's group
    enum { K_GAM, K_BET, K_PIV, K_CA, K_CB };
    GN_STAMP(0); GN_STAMP(1);
    int b, chunk;
    gn_block_slice(f, b, chunk);
    const int c0 = chunk * f.seg_ch, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool active = tid < f.nta;
    const int j = tid % f.seg_vecs, prow = tid / f.seg_vecs;
    const unsigned lds0 = (unsigned)(size_t)lsm, k0 = (unsigned)(size_t)sh_k;
    const unsigned kcol = k0 + (unsigned)(j * VEC * 4), kstride = (unsigned)(f.seg_ch * 4);
    const __amdgpu_buffer_rsrc_t rx = gn_slice_rsrc(x + ((long long)b * s.HW) * s.x_ld + c0, ((long long)(s.HW - 1) * s.x_ld + f.seg_ch) * ES);
    // per-channel inputs: all threads load (channel index clamped: no branch around VMEM), the first seg_ch publish; the pivot stays
    // raw until after the DMA is out — its conversion would otherwise wait for the load right here
    const int kc = c0 + min(tid, f.seg_ch - 1);
    const float in_g = a.gamma[kc], in_b = a.beta[kc];
    T in_praw = x[(long long)b * s.HW * s.x_ld + gn_gidx(kc, s.cpg) * s.cpg];
    const unsigned x_o = (unsigned)((prow * s.x_ld + j * VEC) * ES), x_st = (unsigned)(f.rows_per_iter * s.x_ld * ES);
    auto row_ok = [&](int i) { return active && prow + i * f.rows_per_iter < s.HW; };
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        unsigned ox = row_ok(i) ? x_o + i * x_st : GN_OOB;
        asm volatile("" : "+v"(ox));                     // exactly one DMA per row on every path (hipcc otherwise splits the select into two predicated instructions)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + (i * NT + wave * 64) * 16), 16, ox, 0, 0, 0);
    }
    if (tid < f.seg_ch) {
        const unsigned ad = k0 + (unsigned)(tid * 4);
        gn_lds_wr4(ad + K_GAM * kstride, in_g); gn_lds_wr4(ad + K_BET * kstride, in_b); gn_lds_wr4(ad + K_PIV * kstride, Elem<T>::ld(&in_praw));
    }
    gn_lds_barrier();
    float piv[VEC];
    gn_lds_rd_consts<VEC>(kcol + K_PIV * kstride, piv);
    const float n = (float)s.HW * (float)s.cpg;
    // single pass over the slice while it lands: moments of (x - pivot), pivot = the group's first element (gn_pivot) — accurate
    // when |mean| >> std, and the statistics cost nothing beyond the load phase
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
    const unsigned myx_ad = lds0 + (unsigned)(tid * 16);
    static_for_gn<NV>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        gn_wait_vm<NV - 1 - i>(0);
        const int p = prow + i * f.rows_per_iter;
        if (active && p < s.HW) {
            float fv[VEC];
            Elem<T>::unpack(gn_lds_rd16(myx_ad + i * (NT * 16)), fv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = fv[e] - piv[e]; s1[e] += d; s2[e] += d * d; }
        }
    });
    GN_STAMP(2);
#ifdef GN_TIMING
    __syncthreads();          // timing builds: separate 'waiting for the slowest wave' from the reduction proper
#endif
    GN_STAMP(3);
    float ca[VEC], cb[VEC];
    const bool fast = (s.cpg & (s.cpg - 1)) == 0 && s.cpg <= 64;
    if (fast) {
        // Finish in TWO barriers: per-wave totals -> (barrier) -> thread c < seg_ch adds the waves' totals of its channel, the cpg lanes of
        // a group add up by shuffles, every lane of the group derives (mean, rstd) and publishes its channel's scale / shift -> (barrier)
        // -> everyone reads its eight.  (Was: channel sums, group statistics and coefficients in separate steps of 32-512 active threads
        // with a barrier after each: 2.0 of the 16 x 16 forward's 7.2 us per block.)
        float v2[2 * VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { v2[e] = active ? s1[e] : 0.f; v2[VEC + e] = active ? s2[e] : 0.f; }
        // The reduction is a serial chain on a few threads.  A block that reaches it while its CU partner streams out its result competes
        // with sixteen VALU-bound waves under oldest-first arbitration: 5.2 us instead of 1.4 for the quarter of the blocks that end last
        // (scripts/gn_timeline.sh, "last-ending quarter") — so the chain runs at raised priority.
        __builtin_amdgcn_s_setprio(3);
        gn_wave_partials<VEC, 2, NW>(v2, f, j, tid, sh_row);
        __syncthreads();
        if (tid < f.seg_ch) {
            float acc = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { acc += sh_row[w * f.seg_ch + tid]; sq += sh_row[(NW + w) * f.seg_ch + tid]; }
            acc = gn_group_lanes_sum(acc, s.cpg); sq = gn_group_lanes_sum(sq, s.cpg);
            // fp32 is enough here: the moments are of (x - pivot) with the pivot inside the data, so E[d^2] - E[d]^2 does not cancel
            const float inv_n = 1.0f / n, dmean = acc * inv_n;
            const float var = fmaxf(sq * inv_n - dmean * dmean, 0.f);
            const float mean_g = sh_k[K_PIV * f.seg_ch + tid] + dmean;
            float rstd = __builtin_amdgcn_rsqf(var + a.eps);
            rstd = rstd * (1.5f - 0.5f * (var + a.eps) * rstd * rstd);          // one Newton step on the hardware estimate
            const float cag = rstd * sh_k[K_GAM * f.seg_ch + tid];
            sh_k[K_CA * f.seg_ch + tid] = cag;
            sh_k[K_CB * f.seg_ch + tid] = sh_k[K_BET * f.seg_ch + tid] - mean_g * cag;
            if (a.stats && (tid & (s.cpg - 1)) == 0) {
                const long long gi = (long long)b * s.G + chunk * f.GPB + gn_gidx(tid, s.cpg);
                a.stats[gi * 2] = mean_g; a.stats[gi * 2 + 1] = rstd;
            }
        }
        // (LDS-only barrier: __syncthreads() would wait for the ack of the statistics store just issued — 1.5 us for the median block, 5 us
        //  for the quarter of the blocks that get here while the others are storing: scripts/gn_timeline.sh, "last-ending quarter")
        gn_lds_barrier();
        __builtin_amdgcn_s_setprio(0);
        GN_STAMP(4);
        if (!active) return;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ca[e] = sh_k[K_CA * f.seg_ch + j * VEC + e]; cb[e] = sh_k[K_CB * f.seg_ch + j * VEC + e]; }
    } else {
    gn_block_channel_sum2_w<VEC, NW>(s1, s2, f, active, j, tid, sh_row, sh_ch);
    if (tid < f.GPB) {
        float acc = 0.f, sq = 0.f;
        for (int c = tid * s.cpg; c < (tid + 1) * s.cpg; ++c) { acc += sh_ch[c]; sq += sh_ch[f.seg_ch + c]; }
        // fp32 is enough here: the moments are of (x - pivot) with the pivot inside the data, so E[d^2] - E[d]^2 does not cancel
        const float inv_n = 1.0f / n, dmean = acc * inv_n;
        const float var = fmaxf(sq * inv_n - dmean * dmean, 0.f);
        const float mean_g = sh_k[K_PIV * f.seg_ch + tid * s.cpg] + dmean;
        float rstd = __builtin_amdgcn_rsqf(var + a.eps);
        rstd = rstd * (1.5f - 0.5f * (var + a.eps) * rstd * rstd);          // one Newton step on the hardware estimate
        sh_mean[tid] = mean_g; sh_rstd[tid] = rstd;
        if (a.stats) {
            const long long gi = (long long)b * s.G + chunk * f.GPB + tid;
            a.stats[gi * 2] = mean_g; a.stats[gi * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    GN_STAMP(4);
    if (!active) return;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int g = gn_gidx(j * VEC + e, s.cpg);
        ca[e] = sh_rstd[g] * sh_k[K_GAM * f.seg_ch + j * VEC + e];
        cb[e] = sh_k[K_BET * f.seg_ch + j * VEC + e] - sh_mean[g] * ca[e];
    }
    }
    const float keep_scale = do_drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned h0 = dropout_h0(gn_seed(a));
    const unsigned idx0 = (unsigned)(((unsigned)b * (unsigned)s.HW) * (unsigned)s.C + (unsigned)(c0 + j * VEC));
    const u32x4* myx = reinterpret_cast<const u32x4*>(xs) + tid;
    const __amdgpu_buffer_rsrc_t ry = gn_slice_rsrc(y + ((long long)b * s.HW) * s.y_ld + c0, ((long long)(s.HW - 1) * s.y_ld + f.seg_ch) * ES);
    const unsigned y_o = (unsigned)((prow * s.y_ld + j * VEC) * ES), y_st = (unsigned)(f.rows_per_iter * s.y_ld * ES);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = prow + i * f.rows_per_iter;
        if constexpr (VEC == 8) {
            // bf16: element pairs in 2-wide float vectors (v_pk_fma / v_pk_mul / v_pk_add): silu(z) = z / (1 + 2^(-z log2 e))
            const u32x4 xv = myx[i * NT];
            const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
            const unsigned q0 = (idx0 + (unsigned)p * (unsigned)s.C) >> 1;
            unsigned yw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gn_f32x2 x2, ca2, cb2;
                x2.x = __uint_as_float(xw[k] << 16); x2.y = __uint_as_float(xw[k] & 0xffff0000u);
                ca2.x = ca[2 * k]; ca2.y = ca[2 * k + 1]; cb2.x = cb[2 * k]; cb2.y = cb[2 * k + 1];
                gn_f32x2 z = x2 * ca2 + cb2;
                if (do_silu) {
                    const gn_f32x2 nz = z * (-1.4426950408889634f);
                    gn_f32x2 t;
                    t.x = __builtin_amdgcn_exp2f(nz.x); t.y = __builtin_amdgcn_exp2f(nz.y);
                    t = t + 1.0f;
                    gn_f32x2 sg;
                    sg.x = __builtin_amdgcn_rcpf(t.x); sg.y = __builtin_amdgcn_rcpf(t.y);
                    z = z * sg;
                }
                if (do_drop) {
                    const unsigned wd = dropout_word32(h0, q0 + k);
                    z = z * keep_scale;
                    z.x = (wd & 0xffffu) >= a.thresh16 ? z.x : 0.f;
                    z.y = (wd >> 16) >= a.thresh16 ? z.y : 0.f;
                }
                yw[k] = pack_bf2(z.x, z.y);
            }
            u32x4 yv; yv.x = yw[0]; yv.y = yw[1]; yv.z = yw[2]; yv.w = yw[3];
            gn_buf_st16(ry, row_ok(i) ? y_o + i * y_st : GN_OOB, yv);
            continue;
        }
        float fv[VEC];
        Elem<T>::unpack(myx[i * NT], fv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float z = fv[e] * ca[e] + cb[e];
            if (do_silu) z = silu_fast_(z);
            fv[e] = z;
        }
        if (do_drop) {
            const unsigned q0 = (idx0 + (unsigned)p * (unsigned)s.C) >> 1;
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const unsigned w = dropout_word32(h0, q0 + (e >> 1));
                fv[e] = (w & 0xffffu) >= a.thresh16 ? fv[e] * keep_scale : 0.f;
                fv[e + 1] = (w >> 16) >= a.thresh16 ? fv[e + 1] * keep_scale : 0.f;
            }
        }
        gn_buf_st16(ry, row_ok(i) ? y_o + i * y_st : GN_OOB, Elem<T>::pack(fv));
    }
    GN_STAMP(5);
#ifdef GN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GN_STAMP(6); GN_STAMP(7);
}

constexpr int GN_FUSED_SLICE = 64 * 1024;     // LDS bytes of activations per block: two blocks per CU

// picks the group chunk; false when no chunk of this geometry fits (falls back to the two-launch path)
static const long long gn_fused_max_bytes = getenv("DDPM_GN_FUSED_MAX_KB") ? atoll(getenv("DDPM_GN_FUSED_MAX_KB")) * 1024 : GN_FUSED_SLICE;
// (128 KiB slices = 512-thread blocks work and make the 32^2 x 128-channel forward 10 % faster in isolation, but the training
//  step measured 0.25 ms slower with them — A/B in one session, scripts/step_jitter.py — so the default stays at 64 KiB.)

// LDS-staged kernels: largest group chunk whose x slice fits 64 KiB with <= 8 vectors per thread; prefers >= 512 blocks
// (two per CU) as long as a pixel segment stays >= 128 bytes
// slice | sh_row (2 arrays x waves x seg_ch) | sh_ch | group scalars | [8][seg_ch] constants
static size_t gn_lds_bytes(int nvt, int nt, int seg_ch) {
    return (size_t)nvt * nt * 16 + ((size_t)2 * (nt / 64) * seg_ch + 2 * seg_ch + 64 + 8 * seg_ch) * sizeof(float);
}
static bool gn_lds_plan(const GnShape& s, int esize, GnFused& f, size_t& lds_bytes) {
    const int vec = 16 / esize;
    int best = 0;
    for (int gpb = 1; gpb <= s.G && gpb <= 32; gpb <<= 1) {
        if (s.G % gpb) continue;
        const int seg_ch = gpb * s.cpg;
        if (seg_ch % vec) continue;
        if ((long long)s.HW * seg_ch * esize > 64 * 1024 || seg_ch / vec > 512) break;
        best = gpb;
    }
    if (!best) return false;
    while (best > 1 && (long long)s.B * (s.G / best) < 512 && (best / 2) * s.cpg * esize >= 128 && ((best / 2) * s.cpg) % vec == 0) best >>= 1;
    f.GPB = best; f.seg_ch = best * s.cpg; f.seg_vecs = f.seg_ch / vec;
    if (f.seg_ch * esize < 48) return false;
    f.nt = 512;
    f.nta = (512 / f.seg_vecs) * f.seg_vecs; f.rows_per_iter = f.nta / f.seg_vecs;
    f.nv = (s.HW + f.rows_per_iter - 1) / f.rows_per_iter;
    if (f.nv > 8) return false;
    // Vector columns that do not divide a wave (384 channels: 3 / 12 vectors per pixel) are fine: the wave reduction is a strided
    // tree (gn_block_channel_sum_w).  One vector per thread = a slice of <= 8 KiB: those run the 256-thread form (callers).
    if (f.nv < 2 || f.seg_vecs > 32) return false;
    const int nvt = f.nv <= 1 ? 1 : f.nv <= 2 ? 2 : f.nv <= 4 ? 4 : 8;
    lds_bytes = gn_lds_bytes(nvt, 512, f.seg_ch);
    static const bool no_remap = getenv("DDPM_GN_NO_XCD_REMAP") != nullptr;
    f.xcd_remap = no_remap ? 0 : 1;
    return lds_bytes <= 160 * 1024 && (long long)s.B * s.HW * s.C < (1ll << 32);
}

static bool gn_fused_plan(const GnShape& s, int esize, GnFused& f, size_t& lds_bytes) {
    const int vec = 16 / esize;
    int best = 0;
    for (int gpb = 1; gpb <= s.G && gpb <= 32; gpb <<= 1) {
        if (s.G % gpb) continue;
        const int seg_ch = gpb * s.cpg;
        if (seg_ch % vec) continue;
        if ((long long)s.HW * seg_ch * esize > gn_fused_max_bytes) break;       // 512 threads x 16 vectors at most
        if (seg_ch / vec > 256) break;
        best = gpb;
    }
    if (!best) return false;
    // prefer >= 512 blocks (two per CU) as long as a pixel segment stays >= 128 bytes
    while (best > 1 && (long long)s.B * (s.G / best) < 512 && (best / 2) * s.cpg * esize >= 128 && ((best / 2) * s.cpg) % vec == 0) best >>= 1;
    f.GPB = best; f.seg_ch = best * s.cpg; f.seg_vecs = f.seg_ch / vec;
    // a block reads seg_ch * esize contiguous bytes per pixel: below a full 128-byte line two blocks (usually on different
    // XCDs) fetch every line twice and the two streaming launches win (measured: 32^2 x 128 ch bf16 in 64-byte segments 26
    // vs 25 us, 384 ch 106 vs 57 us; at >= 128 B the single launch wins: 16^2 x 256 ch 12.7 vs 16.8 us, 8^2 6.9 vs 11.4 us)
    if (f.seg_ch * esize < 128) return false;
    f.nt = (long long)s.HW * f.seg_ch * esize > GN_FUSED_SLICE ? 512 : 256;
    f.nta = (f.nt / f.seg_vecs) * f.seg_vecs; f.rows_per_iter = f.nta / f.seg_vecs;
    f.nv = (s.HW + f.rows_per_iter - 1) / f.rows_per_iter;
    if (f.nv > 16) return false;
    lds_bytes = ((size_t)f.rows_per_iter * f.seg_ch + f.seg_ch + 64) * sizeof(float);
    return true;
}

// ---- backward
// dz = dy * mask/(1-p) * silu'(z), z = gamma*xhat + beta.  Per (b, c): A1 = sum dz*xhat, A2 = sum dz.
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, GnShape s, long long dy_ld,
                                     const float* __restrict__ stats, GnApply a, float* __restrict__ partial /*[B][S][C][2]*/) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh1[GN_MAXC], sh2[GN_MAXC];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int cx = threadIdx.x, py = threadIdx.y, PY = blockDim.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    const int p0 = slab * s.pix_per_slab, p1 = min(s.HW, p0 + s.pix_per_slab);
    float mean[VEC], rstd[VEC], gm[VEC], bt[VEC], a1[VEC], a2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = cx * VEC + j, g = c / s.cpg;
        mean[j] = stats[((long long)b * s.G + g) * 2]; rstd[j] = stats[((long long)b * s.G + g) * 2 + 1];
        gm[j] = a.gamma[c]; bt[j] = a.beta[c]; a1[j] = a2[j] = 0.f;
    }
    const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned long long seed = gn_seed(a);
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
    const T* db = dy + ((long long)b * s.HW) * dy_ld + cx * VEC;
#pragma unroll 4
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC], d[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
        Elem<T>::unpack(ldg16(db + (long long)p * dy_ld), d);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (f[j] - mean[j]) * rstd[j];
            float dz = d[j];
            if (a.drop_p > 0.f) {
                unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + cx * VEC + j;
                dz = dropout_keep(seed, idx, a.thresh16) ? dz * keep_scale : 0.f;
            }
            if (a.silu) dz *= silu_gradf_(gm[j] * xh + bt[j]);
            a1[j] += dz * xh; a2[j] += dz;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sh1[py * s.C + cx * VEC + j] = a1[j]; sh2[py * s.C + cx * VEC + j] = a2[j]; }
    __syncthreads();
    for (int c = t; c < s.C; c += nt) {
        float u = 0.f, w = 0.f;
        for (int r = 0; r < PY; ++r) { u += sh1[r * s.C + c]; w += sh2[r * s.C + c]; }
        sh1[c] = u; sh2[c] = w;
    }
    __syncthreads();
    float* o = partial + (((long long)b * s.S + slab) * s.C) * 2;
    for (int c = t; c < s.C; c += nt) { o[2 * c] = sh1[c]; o[2 * c + 1] = sh2[c]; }
}

// dx = rstd * (dz*gamma - xhat*c1 - c2)   [+= when accumulate]
template <typename T>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, GnShape s, long long dy_ld,
                                    long long dx_ld, const float* __restrict__ stats, const float* __restrict__ partial,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, GnApply a, int accumulate,
                                    const T* __restrict__ addp, long long add_ld) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh1[GN_MAXC], sh2[GN_MAXC];
    __shared__ float sh_c1[64], sh_c2[64];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int cx = threadIdx.x, py = threadIdx.y, PY = blockDim.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    const int p0 = slab * s.pix_per_slab, p1 = min(s.HW, p0 + s.pix_per_slab);
    // group coefficients of this sample from its FINISHED channel sums [B][C][2] (gn_bwd_finalize_kernel; every block used to re-sum
    // the S per-slab partials here first — serial, S up to 256 — and block 0 issued the dgamma / dbeta atomics)
    for (int c = t; c < s.C; c += nt) {
        const float gmc = a.gamma[c];
        sh1[c] = partial[((long long)b * s.C + c) * 2] * gmc; sh2[c] = partial[((long long)b * s.C + c) * 2 + 1] * gmc;
    }
    __syncthreads();
    if (t < s.G) {
        float d1 = 0.f, d2 = 0.f;
        for (int c = t * s.cpg; c < (t + 1) * s.cpg; ++c) { d1 += sh1[c]; d2 += sh2[c]; }
        const float inv_n = 1.0f / ((float)s.HW * s.cpg);
        sh_c1[t] = d1 * inv_n;       // mean(dz*gamma*xhat) over the group
        sh_c2[t] = d2 * inv_n;       // mean(dz*gamma)
    }
    __syncthreads();
    float mean[VEC], rstd[VEC], gm[VEC], bt[VEC], c1[VEC], c2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = cx * VEC + j, g = c / s.cpg;
        mean[j] = stats[((long long)b * s.G + g) * 2]; rstd[j] = stats[((long long)b * s.G + g) * 2 + 1];
        c1[j] = sh_c1[g]; c2[j] = sh_c2[g];
        gm[j] = a.gamma[c]; bt[j] = a.beta[c];
    }
    const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const unsigned long long seed = gn_seed(a);
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
    const T* db = dy + ((long long)b * s.HW) * dy_ld + cx * VEC;
    T* ob = dx + ((long long)b * s.HW) * dx_ld + cx * VEC;
#pragma unroll 2
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC], d[VEC], o[VEC], ad[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
        Elem<T>::unpack(ldg16(db + (long long)p * dy_ld), d);
        if (accumulate) Elem<T>::unpack(ldg16(ob + (long long)p * dx_ld), o);
        if (addp) Elem<T>::unpack(ldg16(addp + ((long long)b * s.HW + p) * add_ld + cx * VEC), ad);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (f[j] - mean[j]) * rstd[j];
            float dz = d[j];
            if (a.drop_p > 0.f) {
                unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + cx * VEC + j;
                dz = dropout_keep(seed, idx, a.thresh16) ? dz * keep_scale : 0.f;
            }
            if (a.silu) dz *= silu_gradf_(gm[j] * xh + bt[j]);
            float r = rstd[j] * (dz * gm[j] - xh * c1[j] - c2[j]);
            if (addp) r += ad[j];
            o[j] = accumulate ? o[j] + r : r;
        }
        stg16(ob + (long long)p * dx_ld, Elem<T>::pack(o));
    }
}


// ---------------------------------------------------------------------------------------------- host side
static int gn_geometry(int B, int HW, int C, int G, long long x_ld, long long y_ld, int esize, GnShape& s, dim3& block, dim3& grid) {
    const int vec = 16 / esize;
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || C % G || C % vec || C > GN_MAXC) return DDPM_ERR_SHAPE;
    if (x_ld % vec || y_ld % vec) return DDPM_ERR_ALIGN;
    const int cv = C / vec;
    if (cv > GN_THREADS) return DDPM_ERR_SHAPE;
    int py = GN_THREADS / cv; if (py < 1) py = 1;
    if (py > HW) py = HW;
    block = dim3(cv, py);
    // enough slabs to give every CU a dozen blocks (256 threads each: ~8 resident per CU and a queue behind them), but at least ~4
    // pixel-iterations of work each.  (Was: 1024 blocks in all, <= 256 slabs — two blocks per CU on the 256 x 256 tensors at B = 2,
    // every thread with one load in flight: 67 us for 67 MB.)
    int S = (3072 + B - 1) / B;
    int maxS = (HW + 4 * py - 1) / (4 * py);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > 512) S = 512;                 // (the per-sample sums over S are finished by small, latency-bound kernels: ~28 us at S = 1024)
    int pps = (HW + S - 1) / S;
    S = (HW + pps - 1) / pps;
    s.B = B; s.HW = HW; s.C = C; s.G = G; s.cpg = C / G; s.x_ld = x_ld; s.y_ld = y_ld; s.S = S; s.pix_per_slab = pps;
    grid = dim3(S, B);
    return DDPM_OK;
}

static GnApply make_apply(const float* gamma, const float* beta, float eps, int silu, float drop_p, unsigned long long seed, float* stats,
                          const unsigned long long* seed_dev = nullptr) {
    GnApply a; a.seed_dev = seed_dev; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu; a.drop_p = drop_p; a.seed = seed; a.stats = stats;
    a.thresh16 = dropout_thresh16(drop_p);
    return a;
}

// floats of scratch the forward ([B][S][G][2]) and backward ([B][S][C][2] + [B][G][2]) launches need
extern "C" long long ddpm_gn_workspace_floats(int B, int HW, int C, int G, int dtype) {
    GnShape s; dim3 block, grid;
    if (gn_geometry(B, HW, C, G, C, C, dtype == DDPM_BF16 ? 2 : 4, s, block, grid)) return -1;
    return (long long)B * s.S * C * 2 + (long long)B * C * 2 + (long long)B * G * 2;     // per-slab partials | finished channel sums | finished statistics
}

extern "C" int ddpm_groupnorm_silu_fwd(const void* x, long long x_ld, void* y, long long y_ld, const float* gamma, const float* beta,
                                       float* stats, float* workspace, int B, int HW, int C, int G, float eps, int silu,
                                       float drop_p, unsigned long long seed, const unsigned long long* seed_dev, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !workspace) return DDPM_ERR_NULL;
    if (!aligned16(x) || !aligned16(y)) return DDPM_ERR_ALIGN;
    GnShape s; dim3 block, grid;
    const int es = dtype == DDPM_BF16 ? 2 : 4;
    if (dtype != DDPM_BF16 && dtype != DDPM_F32) return DDPM_ERR_DTYPE;
    int rc = gn_geometry(B, HW, C, G, x_ld, y_ld, es, s, block, grid);
    if (rc) return rc;
    GnApply a = make_apply(gamma, beta, eps, silu, drop_p, seed, stats, seed_dev);
    hipStream_t st = (hipStream_t)stream;
    GnFused f; size_t lds = 0;
    static const bool no_fused = getenv("DDPM_GN_NO_FUSED") != nullptr;
    static const bool no_lds = getenv("DDPM_GN_NO_LDS_FWD") != nullptr;
    GnFused fl; size_t lds_l = 0;
    const bool reg_ok = !no_fused && gn_fused_plan(s, es, f, lds);
    // small slices (<= 2 vectors per thread of a 256-thread block): the staged kernel in its 256-thread form
    static const bool small_reg = getenv("DDPM_GN_SMALL_REG") != nullptr;
    const bool small = reg_ok && f.nv <= 2 && f.nt == 256 && f.seg_vecs <= 32 && !small_reg && !no_lds;
    if (small) { fl = f; fl.xcd_remap = 0; lds_l = gn_lds_bytes(f.nv <= 1 ? 1 : 2, 256, f.seg_ch); }
    if (small || (!no_lds && !(reg_ok && f.nv <= 2) && gn_lds_plan(s, es, fl, lds_l))) {        // x staged in LDS: single launch, 1 read + 1 write of HBM
        const dim3 fgrid(G / fl.GPB, B);
#define GN_LF(T, NV, NT, FLV) do { static DevOnce attr; if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_lds_fwd_kernel<T, NV, NT, FLV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DDPM_ERR_LAUNCH; attr = true; } \
        hipLaunchKernelGGL((gn_lds_fwd_kernel<T, NV, NT, FLV>), fgrid, dim3(NT), lds_l, st, (const T*)x, (T*)y, s, fl, a); } while (0)
#define GN_LF_NV(T, FLV) do { if (small) { if (fl.nv <= 1) GN_LF(T, 1, 256, FLV); else GN_LF(T, 2, 256, FLV); } else if (fl.nv <= 2) GN_LF(T, 2, 512, FLV); else if (fl.nv <= 4) GN_LF(T, 4, 512, FLV); else GN_LF(T, 8, 512, FLV); } while (0)
        if (dtype == DDPM_BF16) {
            const int flg = (silu ? 1 : 0) | (drop_p > 0.f ? 2 : 0);       // the three combinations the UNet uses are baked in
            if (flg == 1) GN_LF_NV(bf16_t, 1); else if (flg == 3) GN_LF_NV(bf16_t, 3); else if (flg == 0) GN_LF_NV(bf16_t, 0); else GN_LF_NV(bf16_t, -1);
        } else GN_LF_NV(float, -1);
#undef GN_LF_NV
#undef GN_LF
        return check_launch();
    }
    if (reg_ok) {      // register-resident single launch (1 read + 1 write of HBM)
        const dim3 fgrid(G / f.GPB, B);
#define GN_FWD(T, NV) do { if (f.nt == 512) hipLaunchKernelGGL((gn_reg_fwd_kernel<T, NV, 512>), fgrid, dim3(512), lds, st, (const T*)x, (T*)y, s, f, a); \
                           else hipLaunchKernelGGL((gn_reg_fwd_kernel<T, NV, 256>), fgrid, dim3(256), lds, st, (const T*)x, (T*)y, s, f, a); } while (0)
#define GN_FWD_NV(T) do { if (f.nv <= 1) GN_FWD(T, 1); else if (f.nv <= 2) GN_FWD(T, 2); else if (f.nv <= 4) GN_FWD(T, 4); else if (f.nv <= 8) GN_FWD(T, 8); else GN_FWD(T, 16); } while (0)
        if (dtype == DDPM_BF16) GN_FWD_NV(bf16_t); else GN_FWD_NV(float);
#undef GN_FWD_NV
#undef GN_FWD
        return check_launch();
    }
    float* fin = workspace + (long long)B * s.S * C * 2 + (long long)B * C * 2;        // [B][G][2]
    if (dtype == DDPM_BF16) {
        hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, s, workspace);
        hipLaunchKernelGGL(gn_stats_finalize_kernel<bf16_t>, dim3(G, B), dim3(64), 0, st, (const bf16_t*)x, s, workspace, eps, fin, stats);
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, s, fin, a);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<float>, grid, block, 0, st, (const float*)x, s, workspace);
        hipLaunchKernelGGL(gn_stats_finalize_kernel<float>, dim3(G, B), dim3(64), 0, st, (const float*)x, s, workspace, eps, fin, stats);
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, block, 0, st, (const float*)x, (float*)y, s, fin, a);
    }
    return check_launch();
}

extern "C" int ddpm_groupnorm_silu_bwd(const void* x, long long x_ld, const void* dy, long long dy_ld, void* dx, long long dx_ld,
                                       const float* gamma, const float* beta, const float* stats, float* dgamma, float* dbeta,
                                       float* workspace, int B, int HW, int C, int G, int silu, float drop_p, unsigned long long seed,
                                       const unsigned long long* seed_dev, int accumulate, float* dx_colsum, long long colsum_ld,
                                       const void* add, long long add_ld, int dtype, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || !stats || !workspace) return DDPM_ERR_NULL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return DDPM_ERR_ALIGN;
    if (dtype != DDPM_BF16 && dtype != DDPM_F32) return DDPM_ERR_DTYPE;
    GnShape s; dim3 block, grid;
    const int es = dtype == DDPM_BF16 ? 2 : 4;
    int rc = gn_geometry(B, HW, C, G, x_ld, x_ld, es, s, block, grid);
    if (rc) return rc;
    if (dy_ld % (16 / es) || dx_ld % (16 / es) || (add && (add_ld % (16 / es) || !aligned16(add)))) return DDPM_ERR_ALIGN;
    GnApply a = make_apply(gamma, beta, 0.f, silu, drop_p, seed, nullptr, seed_dev);
    hipStream_t st = (hipStream_t)stream;
    GnFused f; size_t lds = 0;
    static const bool no_fused = getenv("DDPM_GN_NO_FUSED") != nullptr || getenv("DDPM_GN_NO_FUSED_BWD") != nullptr;
    // register-resident single launch (x, dy read once, dx written once) — for the small slices only: with x AND dy held in
    // registers the kernel needs > 128 VGPRs from 8 vectors per thread on and then cannot share a CU with the weight-gradient
    // blocks of the side stream; measured end to end (6 alternating pairs) the two streaming launches are 0.13 ms per step
    // faster on the 16^2 / 32^2 tensors, while the 4^2 / 8^2 tensors (<= 2 vectors per thread) keep the single launch.
    static const bool no_lds = getenv("DDPM_GN_NO_LDS_BWD") != nullptr;
    // 1) slices too big for two register vectors per thread: (x, dy) staged in LDS, one launch, one HBM read of each input
    GnFused fl; size_t lds_l = 0;
    const bool small = !no_fused && gn_fused_plan(s, es, f, lds) && f.nv <= 2;
    static const bool small_reg = getenv("DDPM_GN_SMALL_REG") != nullptr;
    const bool small_lds = small && f.nt == 256 && f.seg_vecs <= 32 && !small_reg && !no_lds;
    if (small_lds) { fl = f; fl.xcd_remap = 0; lds_l = gn_lds_bytes(f.nv <= 1 ? 1 : 2, 256, f.seg_ch); }
    if (small_lds || (!small && !no_lds && gn_lds_plan(s, es, fl, lds_l))) {
        const dim3 fgrid(G / fl.GPB, B);
#define GN_LDS(K, NT, ...) do { static DevOnce attr; if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&K<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DDPM_ERR_LAUNCH; attr = true; } \
        hipLaunchKernelGGL((K<__VA_ARGS__>), fgrid, dim3(NT), lds_l, st, (const T_*)x, (const T_*)dy, (T_*)dx, s, fl, dy_ld, dx_ld, stats, dgamma, dbeta, a, accumulate, dx_colsum, colsum_ld, (const T_*)add, add_ld); } while (0)
#define GN_LDS_NV(P2V, K8, ...) do { if (small_lds) { if (fl.nv <= 1) GN_LDS(gn_lds_bwd_kernel, 256, T_, 1, 256, P2V); else GN_LDS(gn_lds_bwd_kernel, 256, T_, 2, 256, P2V); } \
                         else if (fl.nv <= 2) GN_LDS(gn_lds_bwd_kernel, 512, T_, 2, 512, P2V); else if (fl.nv <= 4) GN_LDS(gn_lds_bwd_kernel, 512, T_, 4, 512, P2V); else GN_LDS(K8, 512, T_, 8, ##__VA_ARGS__); } while (0)
        // P2: every group is a power-of-two run of channels that the halves of a 16-byte vector column do not straddle
        const bool p2 = (s.cpg & (s.cpg - 1)) == 0 && s.cpg >= (16 / es) / 2 && s.cpg <= 64 && (fl.seg_vecs & (fl.seg_vecs - 1)) == 0;
        if (dtype == DDPM_BF16) {
            typedef bf16_t T_;
            if (p2) {
                // the three flag combinations the UNet uses, baked in (norm1 / out: SiLU; norm2 in training: SiLU + dropout; attention: neither)
#define GN_LDS_FL(FLV) do { if (small_lds) { if (fl.nv <= 1) GN_LDS(gn_lds_bwd_kernel, 256, T_, 1, 256, true, FLV); else GN_LDS(gn_lds_bwd_kernel, 256, T_, 2, 256, true, FLV); } \
                         else if (fl.nv <= 2) GN_LDS(gn_lds_bwd_kernel, 512, T_, 2, 512, true, FLV); else if (fl.nv <= 4) GN_LDS(gn_lds_bwd_kernel, 512, T_, 4, 512, true, FLV); \
                         else GN_LDS(gn_lds_bwd_kernel, 512, T_, 8, 512, true, FLV); } while (0)
                const int flg = (silu ? 1 : 0) | (drop_p > 0.f ? 2 : 0);
                if (flg == 1) GN_LDS_FL(1); else if (flg == 3) GN_LDS_FL(3); else if (flg == 0) GN_LDS_FL(0); else GN_LDS_FL(-1);
#undef GN_LDS_FL
            } else GN_LDS_NV(false, gn_lds_bwd_kernel, 512, false);
        } else {
            typedef float T_;
            if (p2) GN_LDS_NV(true, gn_lds_bwd8_kernel); else GN_LDS_NV(false, gn_lds_bwd8_kernel);
        }
#undef GN_LDS_NV
#undef GN_LDS
        return check_launch();
    }
    // every other path leaves the per-sample column sums of dx to the stand-alone kernel
    struct ColsumAfter {
        const void* dx; long long dx_ld; float* out; long long ld; int B, HW, C, dtype; void* stream;
        int run() const { return out ? ddpm_colsum(dx, dx_ld, out, ld, nullptr, B, HW, C, dtype, stream) : DDPM_OK; }
    } colsum_after{dx, dx_ld, dx_colsum, colsum_ld, B, HW, C, dtype, stream};
    if (small) {
        const dim3 fgrid(G / f.GPB, B);
#define GN_BWD(T, NV) do { if (f.nt == 512) hipLaunchKernelGGL((gn_reg_bwd_kernel<T, NV, 512>), fgrid, dim3(512), lds, st, (const T*)x, (const T*)dy, (T*)dx, s, f, dy_ld, dx_ld, stats, dgamma, dbeta, a, accumulate, (const T*)add, add_ld, dx_colsum, colsum_ld); \
                           else hipLaunchKernelGGL((gn_reg_bwd_kernel<T, NV, 256>), fgrid, dim3(256), lds, st, (const T*)x, (const T*)dy, (T*)dx, s, f, dy_ld, dx_ld, stats, dgamma, dbeta, a, accumulate, (const T*)add, add_ld, dx_colsum, colsum_ld); } while (0)
#define GN_BWD_NV(T) do { if (f.nv <= 1) GN_BWD(T, 1); else if (f.nv <= 2) GN_BWD(T, 2); else if (f.nv <= 4) GN_BWD(T, 4); else if (f.nv <= 8) GN_BWD(T, 8); else GN_BWD(T, 16); } while (0)
        if (dtype == DDPM_BF16) GN_BWD_NV(bf16_t); else GN_BWD_NV(float);
#undef GN_BWD_NV
#undef GN_BWD
        return check_launch();                                  // (the kernel wrote the per-sample column sums itself)
    }
    float* partial = workspace;                                   // [B][S][C][2]
    float* fin2 = workspace + (long long)B * s.S * C * 2;         // [B][C][2]
    if (dtype == DDPM_BF16)
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, s, dy_ld, stats, a, partial);
    else
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)dy, s, dy_ld, stats, a, partial);
    if (s.S <= 64) hipLaunchKernelGGL(gn_bwd_finalize_kernel<16>, dim3((C + 15) / 16, B), dim3(64), 0, st, s, partial, fin2, dgamma, dbeta);
    else if (s.S <= 192) hipLaunchKernelGGL(gn_bwd_finalize_kernel<8>, dim3((C + 7) / 8, B), dim3(64), 0, st, s, partial, fin2, dgamma, dbeta);
    else hipLaunchKernelGGL(gn_bwd_finalize_kernel<4>, dim3((C + 3) / 4, B), dim3(64), 0, st, s, partial, fin2, dgamma, dbeta);
    if (dtype == DDPM_BF16)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, s, dy_ld, dx_ld, stats, fin2, dgamma, dbeta, a, accumulate, (const bf16_t*)add, add_ld);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)dy, (float*)dx, s, dy_ld, dx_ld, stats, fin2, dgamma, dbeta, a, accumulate, (const float*)add, add_ld);
    const int rc3 = check_launch();
    return rc3 ? rc3 : colsum_after.run();
}
