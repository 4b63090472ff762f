// Fused GroupNorm(32, eps) [+ SiLU] [+ dropout] forward / backward over NHWC activations (gfx950).
// Reference semantics: nn.GroupNorm(32, C, eps=1e-6) -> SiLU -> Dropout (ddpm_torch/models/unet.py:18-20,85-87).
//
// HBM-bound: algorithmic traffic is one read of x + one write of y per element.  Two launches:
//   gn_stats : grid (S, B) — each block reduces a slab of pixels of one sample to per-group partial
//              (sum, sum of squares) with 16-byte coalesced loads and a wavefront/LDS reduction;
//   gn_apply : grid (S, B) — combines the S partials (fp64), builds per-channel scale/shift in LDS and
//              streams y = silu(x*a_c + b_c) [* keep/(1-p)]; the second read of x is an L2/MALL hit at
//              the CIFAR sizes (<= 64 MB per tensor vs 256 MB Infinity Cache).
// Backward mirrors it: gn_bwd_reduce (per-slab per-channel sums of dz and dz*xhat), gn_bwd_apply (per-sample group
// coefficients from those sums + dgamma/dbeta atomics by the slab-0 blocks, then dx).  SiLU and the dropout mask are
// recomputed.
#include "common.h"

constexpr int GN_THREADS = 256;
constexpr int GN_MAXC = 2048;

struct GnShape {
    int B, HW, C, G, cpg;       // cpg = C / G
    long long x_ld, y_ld;       // pixel pitches (elements)
    int S;                      // pixel slabs per sample
    int pix_per_slab;
};

// thread (cx, py): channel-vector cx fixed, pixels py, py+PY, ...  (blockDim = (CV, PY))
template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ x, GnShape s, float* __restrict__ partial /*[B][S][G][2]*/) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh_sum[GN_MAXC], sh_sq[GN_MAXC];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int cx = threadIdx.x, py = threadIdx.y, PY = blockDim.y;
    const int p0 = slab * s.pix_per_slab, p1 = min(s.HW, p0 + s.pix_per_slab);
    float sum[VEC], sq[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sum[j] = sq[j] = 0.f;
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { sum[j] += f[j]; sq[j] += f[j] * f[j]; }
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
    float drop_p; unsigned thresh24; unsigned long long seed;   // dropout after SiLU (unet.py:87); p = 0 disables
    float* stats;              // [B][G][2] (mean, rstd) saved for backward, or null
};

template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, GnShape s, const float* __restrict__ partial, GnApply a) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh_a[GN_MAXC], sh_b[GN_MAXC];
    __shared__ float sh_mean[64], sh_rstd[64];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    if (t < s.G) {
        double sum = 0.0, sq = 0.0;
        for (int i = 0; i < s.S; ++i) {
            const float* o = partial + (((long long)b * s.S + i) * s.G + t) * 2;
            sum += o[0]; sq += o[1];
        }
        const double n = (double)s.HW * s.cpg;
        const double mean = sum / n;
        double var = sq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        sh_mean[t] = (float)mean; sh_rstd[t] = rstd;
        if (a.stats && slab == 0) { a.stats[((long long)b * s.G + t) * 2] = (float)mean; a.stats[((long long)b * s.G + t) * 2 + 1] = rstd; }
    }
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
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float z = f[j] * ca[j] + cb[j];
            if (a.silu) z = siluf_(z);
            if (a.drop_p > 0.f) {
                unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + cx * VEC + j;
                z = dropout_keep(a.seed, idx, a.thresh24) ? z * keep_scale : 0.f;
            }
            f[j] = z;
        }
        stg16(yb + (long long)p * s.y_ld, Elem<T>::pack(f));
    }
}

// ---- single-launch forward: one block owns (sample b, a chunk of GPB groups) = all HW pixels x seg_ch = GPB*cpg channels.
// The slice (<= 64 KiB) is DMA'd into LDS once, statistics are taken from LDS (two-pass: mean, then centred variance),
// and y is produced from the LDS copy: exactly one HBM read of x and one write of y, no partial-sum workspace, one launch.
struct GnFused {
    int GPB, seg_ch, seg_vecs;     // groups per block, channels / 16-byte vectors per pixel segment
    int nta;                       // active threads: largest multiple of seg_vecs <= GN_THREADS (a thread keeps one vector column)
    int rows_per_iter;             // nta / seg_vecs pixels per sweep of the block
    int tpg;                       // threads cooperating on one group's statistics (GN_THREADS / GPB)
};

template <typename T>
__global__ __launch_bounds__(GN_THREADS)
void gn_fused_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, GnShape s, GnFused f, GnApply a) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int slice_bytes = s.HW * f.seg_ch * ES;
    float* sh_a = reinterpret_cast<float*>(gsm + slice_bytes);       // [seg_ch] scale
    float* sh_b = sh_a + f.seg_ch;                                    // [seg_ch] shift
    float* sh_red = sh_b + f.seg_ch;                                  // [GN_THREADS / 64][GPB] cross-wave partials
    float* sh_mean = sh_red + (GN_THREADS / 64) * 32;                 // [GPB]
    float* sh_rstd = sh_mean + 32;                                    // [GPB]
    const int b = blockIdx.y, c0 = blockIdx.x * f.seg_ch, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nvec = s.HW * f.seg_vecs;
    const bool active = tid < f.nta;
    const int j = tid % f.seg_vecs, prow = tid / f.seg_vecs;          // this thread's vector column and first pixel

    // phase 1: slice -> LDS (direct-to-LDS loads, vector v lands at byte 16 v)
    {
        const T* xb = x + ((long long)b * s.HW) * s.x_ld + c0;
        const unsigned long long ad = (unsigned long long)xb;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
        const long long ext = ((long long)(s.HW - 1) * s.x_ld + f.seg_ch) * ES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                                           __builtin_amdgcn_readfirstlane((int)ext), 0x00020000);
        int pp = prow;
        for (int v0 = 0; v0 < nvec; v0 += f.nta, pp += f.rows_per_iter) {
            // lanes past the slice (or past the last whole vector column set) stay masked: an LDS-DMA lane always writes
            if (active && pp < s.HW) {
                const unsigned off = (unsigned)(((long long)pp * s.x_ld + j * VEC) * ES);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(gsm + (v0 + wave * 64) * 16), 16, off, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // phase 2: statistics from LDS — tpg threads per group, each strides over the pixels of its group's cpg channels
    {
        const int g = tid / f.tpg, t = tid - g * f.tpg;                // tpg * GPB == GN_THREADS
        const int chunks = s.cpg * ES / 8;                            // 8-byte pieces of one pixel's group segment
        const char* base = gsm + g * s.cpg * ES;
        const int pitch = f.seg_ch * ES;
        auto reduce_group = [&](float v) -> float {
            // deterministic: xor-butterfly inside the (<= 64 lane) thread set of the group, then fixed-order sum over waves
            const int span = f.tpg < 64 ? f.tpg : 64;
            for (int o = span >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (f.tpg <= 64) return v;
            __syncthreads();
            if (lane == 0) sh_red[wave] = v;
            __syncthreads();
            const int w0 = g * (f.tpg >> 6);
            float r = 0.f;
            for (int w = 0; w < (f.tpg >> 6); ++w) r += sh_red[w0 + w];
            return r;
        };
        float sum = 0.f;
        for (int p = t; p < s.HW; p += f.tpg)
            for (int q = 0; q < chunks; ++q) {
                const uint2 u = *reinterpret_cast<const uint2*>(base + p * pitch + q * 8);
                if (ES == 2) sum += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) + (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u));
                else sum += __uint_as_float(u.x) + __uint_as_float(u.y);
            }
        const float n = (float)s.HW * (float)s.cpg;
        const float mean = reduce_group(sum) / n;
        float sq = 0.f;
        for (int p = t; p < s.HW; p += f.tpg)
            for (int q = 0; q < chunks; ++q) {
                const uint2 u = *reinterpret_cast<const uint2*>(base + p * pitch + q * 8);
                if (ES == 2) {
                    const float e0 = __uint_as_float(u.x << 16) - mean, e1 = __uint_as_float(u.x & 0xffff0000u) - mean;
                    const float e2 = __uint_as_float(u.y << 16) - mean, e3 = __uint_as_float(u.y & 0xffff0000u) - mean;
                    sq += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
                } else {
                    const float e0 = __uint_as_float(u.x) - mean, e1 = __uint_as_float(u.y) - mean;
                    sq += e0 * e0 + e1 * e1;
                }
            }
        const float var = reduce_group(sq) / n;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        if (t == 0) {
            sh_mean[g] = mean; sh_rstd[g] = rstd;
            if (a.stats) {
                const long long gi = (long long)b * s.G + blockIdx.x * f.GPB + g;
                a.stats[gi * 2] = mean; a.stats[gi * 2 + 1] = rstd;
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < f.seg_ch; c += GN_THREADS) {
        const int g = c / s.cpg;
        const float sc = sh_rstd[g] * a.gamma[c0 + c];
        sh_a[c] = sc; sh_b[c] = a.beta[c0 + c] - sh_mean[g] * sc;
    }
    __syncthreads();

    // phase 3: y = silu(x * a_c + b_c) [* keep / (1 - p)] from the LDS copy
    if (active) {
        float ca[VEC], cb[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ca[e] = sh_a[j * VEC + e]; cb[e] = sh_b[j * VEC + e]; }
        const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
        T* yb = y + ((long long)b * s.HW) * s.y_ld + c0 + j * VEC;
        for (int p = prow; p < s.HW; p += f.rows_per_iter) {
            float v[VEC];
            Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gsm + (p * f.seg_vecs + j) * 16), v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float z = v[e] * ca[e] + cb[e];
                if (a.silu) z = siluf_(z);
                if (a.drop_p > 0.f) {
                    const unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + c0 + j * VEC + e;
                    z = dropout_keep(a.seed, idx, a.thresh24) ? z * keep_scale : 0.f;
                }
                v[e] = z;
            }
            stg16(yb + (long long)p * s.y_ld, Elem<T>::pack(v));
        }
    }
}

constexpr int GN_FUSED_SLICE = 64 * 1024;     // LDS bytes of activations per block: two blocks per CU

// picks the group chunk; false when no chunk of this geometry fits (falls back to the two-launch path)
static bool gn_fused_plan(const GnShape& s, int esize, GnFused& f, size_t& lds_bytes) {
    const int vec = 16 / esize;
    if ((s.cpg * esize) % 8) return false;
    int best = 0;
    for (int gpb = 1; gpb <= s.G && gpb <= 32; gpb <<= 1) {
        if (s.G % gpb || GN_THREADS % gpb) continue;
        const int seg_ch = gpb * s.cpg;
        if (seg_ch % vec) continue;
        if ((long long)s.HW * seg_ch * esize > GN_FUSED_SLICE) break;
        if (seg_ch / vec > GN_THREADS) break;
        best = gpb;
    }
    if (!best) return false;
    // prefer >= 512 blocks (two per CU) as long as a pixel segment stays >= 128 bytes
    while (best > 1 && (long long)s.B * (s.G / best) < 512 && (best / 2) * s.cpg * esize >= 128 && ((best / 2) * s.cpg) % vec == 0) best >>= 1;
    f.GPB = best; f.seg_ch = best * s.cpg; f.seg_vecs = f.seg_ch / vec;
    f.nta = (GN_THREADS / f.seg_vecs) * f.seg_vecs; f.rows_per_iter = f.nta / f.seg_vecs; f.tpg = GN_THREADS / best;
    lds_bytes = (size_t)s.HW * f.seg_ch * esize + (2 * f.seg_ch + (GN_THREADS / 64) * 32 + 64) * sizeof(float);
    return true;
}

// ---- backward
// dz = dy * mask/(1-p) * silu'(z), z = gamma*xhat + beta.  Per (b, c): A1 = sum dz*xhat, A2 = sum dz.
template <typename T>
__global__ void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, GnShape s, long long dy_ld,
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
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
    const T* db = dy + ((long long)b * s.HW) * dy_ld + cx * VEC;
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
                dz = dropout_keep(a.seed, idx, a.thresh24) ? dz * keep_scale : 0.f;
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
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, GnApply a, int accumulate) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float sh1[GN_MAXC], sh2[GN_MAXC];
    __shared__ float sh_c1[64], sh_c2[64];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int cx = threadIdx.x, py = threadIdx.y, PY = blockDim.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    const int p0 = slab * s.pix_per_slab, p1 = min(s.HW, p0 + s.pix_per_slab);
    // group coefficients of this sample from the per-slab channel sums (every block of the sample recomputes them: a few
    // KB from L2 instead of a third launch); the slab-0 block also owns the dgamma / dbeta contribution of the sample
    for (int c = t; c < s.C; c += nt) {
        float a1 = 0.f, a2 = 0.f;
        for (int i = 0; i < s.S; ++i) {
            const float* o = partial + (((long long)b * s.S + i) * s.C + c) * 2;
            a1 += o[0]; a2 += o[1];
        }
        const float gmc = a.gamma[c];
        sh1[c] = a1 * gmc; sh2[c] = a2 * gmc;
        if (slab == 0) {
            if (dgamma) atomicAdd(dgamma + c, a1);
            if (dbeta) atomicAdd(dbeta + c, a2);
        }
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
    const T* xb = x + ((long long)b * s.HW) * s.x_ld + cx * VEC;
    const T* db = dy + ((long long)b * s.HW) * dy_ld + cx * VEC;
    T* ob = dx + ((long long)b * s.HW) * dx_ld + cx * VEC;
    for (int p = p0 + py; p < p1; p += PY) {
        float f[VEC], d[VEC], o[VEC];
        Elem<T>::unpack(ldg16(xb + (long long)p * s.x_ld), f);
        Elem<T>::unpack(ldg16(db + (long long)p * dy_ld), d);
        if (accumulate) Elem<T>::unpack(ldg16(ob + (long long)p * dx_ld), o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (f[j] - mean[j]) * rstd[j];
            float dz = d[j];
            if (a.drop_p > 0.f) {
                unsigned long long idx = ((unsigned long long)b * s.HW + p) * s.C + cx * VEC + j;
                dz = dropout_keep(a.seed, idx, a.thresh24) ? dz * keep_scale : 0.f;
            }
            if (a.silu) dz *= silu_gradf_(gm[j] * xh + bt[j]);
            const float r = rstd[j] * (dz * gm[j] - xh * c1[j] - c2[j]);
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
    // enough slabs to give every CU several blocks, but at least ~4 pixel-iterations of work each
    int S = (1024 + B - 1) / B;
    int maxS = (HW + 4 * py - 1) / (4 * py);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > 256) S = 256;
    int pps = (HW + S - 1) / S;
    S = (HW + pps - 1) / pps;
    s.B = B; s.HW = HW; s.C = C; s.G = G; s.cpg = C / G; s.x_ld = x_ld; s.y_ld = y_ld; s.S = S; s.pix_per_slab = pps;
    grid = dim3(S, B);
    return DDPM_OK;
}

static GnApply make_apply(const float* gamma, const float* beta, float eps, int silu, float drop_p, unsigned long long seed, float* stats) {
    GnApply a; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu; a.drop_p = drop_p; a.seed = seed; a.stats = stats;
    double th = (double)drop_p * 16777216.0;
    a.thresh24 = th <= 0 ? 0u : (th >= 16777216.0 ? 16777216u : (unsigned)(th + 0.5));
    return a;
}

// floats of scratch the forward ([B][S][G][2]) and backward ([B][S][C][2] + [B][G][2]) launches need
extern "C" long long ddpm_gn_workspace_floats(int B, int HW, int C, int G, int dtype) {
    GnShape s; dim3 block, grid;
    if (gn_geometry(B, HW, C, G, C, C, dtype == DDPM_BF16 ? 2 : 4, s, block, grid)) return -1;
    return (long long)B * s.S * C * 2 + (long long)B * G * 2;
}

extern "C" int ddpm_groupnorm_silu_fwd(const void* x, long long x_ld, void* y, long long y_ld, const float* gamma, const float* beta,
                                       float* stats, float* workspace, int B, int HW, int C, int G, float eps, int silu,
                                       float drop_p, unsigned long long seed, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !workspace) return DDPM_ERR_NULL;
    if (!aligned16(x) || !aligned16(y)) return DDPM_ERR_ALIGN;
    GnShape s; dim3 block, grid;
    const int es = dtype == DDPM_BF16 ? 2 : 4;
    if (dtype != DDPM_BF16 && dtype != DDPM_F32) return DDPM_ERR_DTYPE;
    int rc = gn_geometry(B, HW, C, G, x_ld, y_ld, es, s, block, grid);
    if (rc) return rc;
    GnApply a = make_apply(gamma, beta, eps, silu, drop_p, seed, stats);
    hipStream_t st = (hipStream_t)stream;
    GnFused f; size_t lds = 0;
    static const bool no_fused = getenv("DDPM_GN_NO_FUSED") != nullptr;
    // measured (scripts/microbench.py): the single-launch kernel wins while the tensor is small (launch / latency bound
    // 8x8 and 4x4 levels); on the big levels the two-launch path streams at the HBM rate and stays ahead
    const bool small = (long long)B * HW * C * es <= (12ll << 20);
    if (!no_fused && small && gn_fused_plan(s, es, f, lds)) {
        const dim3 fgrid(G / f.GPB, B);
        if (dtype == DDPM_BF16) {
            static bool attr = false;
            if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, GN_FUSED_SLICE + 4096) != hipSuccess) return DDPM_ERR_LAUNCH; attr = true; }
            hipLaunchKernelGGL(gn_fused_fwd_kernel<bf16_t>, fgrid, dim3(GN_THREADS), lds, st, (const bf16_t*)x, (bf16_t*)y, s, f, a);
        } else {
            static bool attr = false;
            if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, GN_FUSED_SLICE + 4096) != hipSuccess) return DDPM_ERR_LAUNCH; attr = true; }
            hipLaunchKernelGGL(gn_fused_fwd_kernel<float>, fgrid, dim3(GN_THREADS), lds, st, (const float*)x, (float*)y, s, f, a);
        }
        return check_launch();
    }
    if (dtype == DDPM_BF16) {
        hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, s, workspace);
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, s, workspace, a);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<float>, grid, block, 0, st, (const float*)x, s, workspace);
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, block, 0, st, (const float*)x, (float*)y, s, workspace, a);
    }
    return check_launch();
}

extern "C" int ddpm_groupnorm_silu_bwd(const void* x, long long x_ld, const void* dy, long long dy_ld, void* dx, long long dx_ld,
                                       const float* gamma, const float* beta, const float* stats, float* dgamma, float* dbeta,
                                       float* workspace, int B, int HW, int C, int G, int silu, float drop_p, unsigned long long seed,
                                       int accumulate, int dtype, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || !stats || !workspace) return DDPM_ERR_NULL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return DDPM_ERR_ALIGN;
    if (dtype != DDPM_BF16 && dtype != DDPM_F32) return DDPM_ERR_DTYPE;
    GnShape s; dim3 block, grid;
    const int es = dtype == DDPM_BF16 ? 2 : 4;
    int rc = gn_geometry(B, HW, C, G, x_ld, x_ld, es, s, block, grid);
    if (rc) return rc;
    if (dy_ld % (16 / es) || dx_ld % (16 / es)) return DDPM_ERR_ALIGN;
    GnApply a = make_apply(gamma, beta, 0.f, silu, drop_p, seed, nullptr);
    hipStream_t st = (hipStream_t)stream;
    float* partial = workspace;                                   // [B][S][C][2]
    if (dtype == DDPM_BF16)
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, s, dy_ld, stats, a, partial);
    else
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)dy, s, dy_ld, stats, a, partial);
    if (dtype == DDPM_BF16)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, s, dy_ld, dx_ld, stats, partial, dgamma, dbeta, a, accumulate);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)dy, (float*)dx, s, dy_ld, dx_ld, stats, partial, dgamma, dbeta, a, accumulate);
    return check_launch();
}
