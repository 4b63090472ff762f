// Fused optimizer-side kernels over flat fp32 buffers: global-norm partials, and one pass that
// applies clip scale + Adam + EMA.  Reference semantics: nn.utils.clip_grad_norm_(max_norm) -> torch.optim.Adam
// (train.py:128: lr, betas, eps=1e-8, no weight decay) -> EMA.update (ddpm_torch/utils/train.py:300-305).
#include "common.h"

// partial[blockIdx] = sum of squares of a slice (fp32 accumulate per thread, wave + LDS reduce)
__global__ void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += g[i] * g[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[0] += sum(partial[0..nblk))   (single block)
__global__ void sum_partials_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += sh[0] + sh[1] + sh[2] + sh[3];
}
// total_sq[0] += ||g||^2 ; workspace needs >= 1024 floats
extern "C" int ddpm_sumsq_accumulate(const float* g, long long n, float* total_sq, float* workspace, void* stream) {
    if (!g || !total_sq || !workspace) return DDPM_ERR_NULL;
    if (n <= 0) return DDPM_OK;
    long long want = (n + 255) / 256;
    const int nblk = (int)(want > 1024 ? 1024 : want);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, n, workspace);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nblk, total_sq);
    return check_launch();
}

// One pass over a parameter tensor: g *= clip ; Adam(m, v) ; p -= step ; shadow += (1-d)(p - shadow).
// clip = min(1, max_norm / (sqrt(total_sq[0]) + 1e-6)) is computed on the device from the norm accumulated above,
// so there is no host synchronisation between backward and the update.
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                float* __restrict__ shadow, long long n, const float* __restrict__ total_sq, float max_norm,
                                float lr, float beta1, float beta2, float eps, float bc1, float bc2, float ema_w) {
    float clip = 1.f;
    if (total_sq && max_norm > 0.f) {
        const float c = max_norm / (sqrtf(total_sq[0]) + 1e-6f);
        clip = c < 1.f ? c : 1.f;
    }
    const float step = lr / bc1;
    const float sqrt_bc2 = sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        const float pi = p[i] - step * (mi / denom);
        p[i] = pi;
        if (shadow) shadow[i] += ema_w * (pi - shadow[i]);
    }
}
extern "C" int ddpm_adam_ema_step(float* p, const float* g, float* m, float* v, float* shadow, long long n, const float* total_sq,
                                  float max_norm, float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                                  float ema_w, void* stream) {
    if (!p || !g || !m || !v) return DDPM_ERR_NULL;
    if (n <= 0) return DDPM_OK;
    long long want = (n + 255) / 256;
    const int nblk = (int)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, n, total_sq, max_norm, lr, beta1, beta2,
                       eps, bias_corr1, bias_corr2, ema_w);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------- multi-tensor forms
// One launch over ALL parameter tensors (304 for the CIFAR UNet).  table[i] = {p, g, m, v, shadow (or 0), numel} as int64;
// grid = (blocks per tensor, n_tensors), grid-stride inside a tensor.
// total_sq is a bank of MT_SUMSQ_LANES accumulators: ~5000 blocks ending in an atomic on ONE address serialise at ~27 ns each
// (the kernel took 127 us for 143 MB); striped over 64 addresses the atomics vanish under the streaming read.
constexpr int MT_SUMSQ_LANES = 64;
// Round 5: the squared gradient norm is BIT-DETERMINISTIC.  Earlier the blocks added their partial sums to the bank with fp32 atomics:
// whatever order they arrived in decided the last bits of the norm, hence of the clip coefficient — and two data-parallel replicas holding
// identical (all-reduced) gradients could step to parameters one ulp apart (seen on the CIFAR geometry with two ranks: 1.2e-7 after four
// steps, and nothing ever pulls replicas back together).  Now every block STORES its partial in its own slot behind the bank and one
// 1024-thread block adds the slots in a fixed order into total_sq[0] (the other 63 lanes are zeroed, so consumers that add the bank up are
// unchanged).  total_sq therefore needs MT_SUMSQ_LANES + (blocks of the producing launch) floats; ddpm_mt_sumsq_slots() says how many.
__global__ __launch_bounds__(1024) void sumsq_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ total_sq) {
    __shared__ float sh[1024];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) acc += partial[i];                 // thread t owns slots t, t + 1024, ...: a fixed order
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {                                            // fixed tree
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < MT_SUMSQ_LANES) total_sq[threadIdx.x] = threadIdx.x == 0 ? sh[0] : 0.f;
}
int ddpm_sumsq_finish_launch(const float* partial, int n, float* total_sq, void* stream) {
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, n, total_sq);
    return check_launch();
}
constexpr int MT_SUMSQ_BLOCKS = 16;                     // blocks per tensor of mt_sumsq_kernel
__global__ void mt_sumsq_kernel(const long long* __restrict__ table, float* __restrict__ partial) {
    __shared__ float sh[4];
    const long long* row = table + 6 * (long long)blockIdx.y;
    const float* g = reinterpret_cast<const float*>(row[1]);
    const long long n = row[5];
    float acc = 0.f;
    const long long nv = (reinterpret_cast<unsigned long long>(g) & 15) == 0 ? n >> 2 : 0;      // float4 body when aligned
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (long long i = (nv << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += g[i] * g[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// hyper (optional, device memory): {lr, bias_corr1, bias_corr2, ema_w} — overrides the by-value arguments, so that a hipGraph
// captured once can be replayed with the step-dependent scalars of every training step (LR schedule, bias corrections, EMA
// warm-up) written to memory by the host before the replay.
__global__ void mt_adam_ema_kernel(const long long* __restrict__ table, const float* __restrict__ total_sq, float max_norm, float lr,
                                   float beta1, float beta2, float eps, float bc1, float bc2, float ema_w, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; ema_w = hyper[3]; }
    const long long* row = table + 6 * (long long)blockIdx.y;
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* m = reinterpret_cast<float*>(row[2]);
    float* v = reinterpret_cast<float*>(row[3]);
    float* shadow = reinterpret_cast<float*>(row[4]);
    const long long n = row[5];
    float clip = 1.f;
    if (total_sq && max_norm > 0.f) {
        float tot = 0.f;
        for (int i = 0; i < MT_SUMSQ_LANES; ++i) tot += total_sq[i];           // fixed order: every block gets the same value
        const float c = max_norm / (sqrtf(tot) + 1e-6f);
        clip = c < 1.f ? c : 1.f;
    }
    const float step = lr / bc1;
    const float sqrt_bc2 = sqrtf(bc2);
    auto upd = [&](float& pi, float gi, float& mi, float& vi, float& si) {
        gi *= clip;
        mi = beta1 * mi + (1.f - beta1) * gi;
        vi = beta2 * vi + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;             // torch.optim.Adam: (sqrt(v) / sqrt(bias_correction2)) + eps
        pi = pi - step * (mi / denom);
        si += ema_w * (pi - si);
    };
    // 16-byte body when every array is aligned (parameters, flat gradient slices and optimizer state all are), scalar tail
    const unsigned long long al = reinterpret_cast<unsigned long long>(p) | reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(m) |
                                  reinterpret_cast<unsigned long long>(v) | reinterpret_cast<unsigned long long>(shadow);
    const long long nv = (al & 15) == 0 ? n >> 2 : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        float4 s4 = shadow ? reinterpret_cast<float4*>(shadow)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        upd(p4.x, g4.x, m4.x, v4.x, s4.x); upd(p4.y, g4.y, m4.y, v4.y, s4.y);
        upd(p4.z, g4.z, m4.z, v4.z, s4.z); upd(p4.w, g4.w, m4.w, v4.w, s4.w);
        reinterpret_cast<float4*>(p)[i] = p4; reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
        if (shadow) reinterpret_cast<float4*>(shadow)[i] = s4;
    }
    for (long long i = (nv << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i], si = shadow ? shadow[i] : 0.f;
        upd(pi, g[i], mi, vi, si);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (shadow) shadow[i] = si;
    }
}
// floats a total_sq buffer needs for a producing launch over n_tensors rows (the 64-lane bank + one slot per block; the unpack kernel
// runs 64 blocks per row, the norm pass 16)
extern "C" int ddpm_mt_sumsq_slots(int n_tensors) { return MT_SUMSQ_LANES + 64 * (n_tensors > 0 ? n_tensors : 0); }
// total_sq: ddpm_mt_sumsq_slots(n_tensors) floats; on return (stream-ordered) total_sq[0] holds the sum, lanes 1..63 are zero.
extern "C" int ddpm_mt_grad_sumsq(const long long* table, int n_tensors, float* total_sq, long long total_sq_floats, void* stream) {
    if (!table || !total_sq) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;
    if (total_sq_floats < ddpm_mt_sumsq_slots(n_tensors)) return DDPM_ERR_SHAPE;
    hipLaunchKernelGGL(mt_sumsq_kernel, dim3(MT_SUMSQ_BLOCKS, n_tensors), dim3(256), 0, (hipStream_t)stream, table, total_sq + MT_SUMSQ_LANES);
    const int rc = check_launch();
    return rc ? rc : ddpm_sumsq_finish_launch(total_sq + MT_SUMSQ_LANES, MT_SUMSQ_BLOCKS * n_tensors, total_sq, stream);
}
extern "C" int ddpm_mt_adam_ema(const long long* table, int n_tensors, const float* total_sq, float max_norm, float lr, float beta1,
                                float beta2, float eps, float bias_corr1, float bias_corr2, float ema_w, const float* hyper_dev, void* stream) {
    if (!table) return DDPM_ERR_NULL;
    if (n_tensors <= 0) return DDPM_OK;
    hipLaunchKernelGGL(mt_adam_ema_kernel, dim3(32, n_tensors), dim3(256), 0, (hipStream_t)stream, table, total_sq, max_norm, lr, beta1, beta2, eps,
                       bias_corr1, bias_corr2, ema_w, hyper_dev);
    return check_launch();
}
