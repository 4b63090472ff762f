// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the DDPM hot path.
// Wave = 64 lanes everywhere in this tree; nothing here is meant to build for another target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DDPM_OK 0
#define DDPM_ERR_SHAPE 1      // dimension / divisibility contract violated
#define DDPM_ERR_DTYPE 2      // unknown dtype enum
#define DDPM_ERR_ALIGN 3      // pointer or pitch not 16-byte aligned
#define DDPM_ERR_LAUNCH 4     // hipGetLastError() after launch
#define DDPM_ERR_NULL 5

#define DDPM_F32 0
#define DDPM_BF16 1

typedef unsigned short bf16_t;   // raw bfloat16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// fp32 -> bf16, round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (one instruction per pair)
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    hw_f32x2 v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// Element traits: VEC = elements per 16-byte vector.
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v; v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]); v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
        return v;
    }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v; v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]); v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
        return v;
    }
};

__device__ __forceinline__ u32x4 ldg16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void stg16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 zero16() { u32x4 v; v.x = v.y = v.z = v.w = 0u; return v; }

// 64-lane wave reductions (butterfly over the whole wavefront).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + __expf(-z)); }
__device__ __forceinline__ float siluf_(float z) { return z * sigmoidf_(z); }
// hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division: for values that are rounded to bf16 right after
#ifdef GN_ABL_NO_TRANS       // timing-only ablation (scripts/gpu_gn_abl.sh): what do the transcendentals cost?  WRONG numerics by construction
__device__ __forceinline__ float silu_fast_(float z) { return z * fminf(fmaxf(0.5f + 0.25f * z, 0.f), 1.f); }
#else
__device__ __forceinline__ float silu_fast_(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
#endif
// d/dz [z*sigmoid(z)] = s*(1 + z*(1-s))
__device__ __forceinline__ float silu_gradf_(float z) { float s = sigmoidf_(z); return s * (1.0f + z * (1.0f - s)); }

// Exact unsigned division by a runtime constant chosen on the host, valid for n < 2^31.
// Power of two: q = n >> shift (mul == 0).  Otherwise q = (n * mul) >> shift with
// l = ceil(log2 d), shift = 31 + l, mul = floor(2^shift / d) + 1  (< 2^32 because d is not a power of 2).
struct FastDiv {
    unsigned mul, shift, d;
};
// q = (n * mul) >> shift for EVERY divisor (a power of two 2^l is mul = 2^31, shift = 31 + l): no "is it a power of two" branch — the kernels
// evaluate these on wave-uniform values, where each such branch ends a basic block and with it the batching of the kernel-argument
// loads around it (the prologue of the tile GEMMs was ten separate s_load / s_waitcnt round trips).
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    return (unsigned)(((unsigned long long)n * f.mul) >> f.shift);
}
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f; f.d = d ? d : 1; f.mul = 1u << 31; f.shift = 31;
    if (d <= 1) return f;
    unsigned l = 0; while ((1ull << l) < d) ++l;
    f.shift = 31 + l;
    if ((1ull << l) == d) return f;
    f.mul = (unsigned)((1ull << f.shift) / d + 1);
    return f;
}

// Counter-based dropout RNG shared by forward and backward (and by the test hook that dumps the mask):
// keep(seed, idx) is a pure function, so backward regenerates the mask instead of storing it.  One 32-bit hash word serves the
// PAIR of adjacent elements (2q, 2q + 1) — 16 bits each, P(keep) = 1 - thresh16 / 65536 — because the two 32-bit integer
// multiplies of the mixer are quarter-rate VALU instructions and the GroupNorm kernels that draw the masks are VALU-bound.
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned dropout_word(unsigned long long seed, unsigned long long pair) {
    const unsigned lo = (unsigned)pair, hi = (unsigned)(pair >> 32);
    return mix32(lo ^ mix32(hi + (unsigned)seed) ^ (unsigned)(seed >> 32) * 0x9E3779B9u);
}
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, unsigned thresh16) {
    const unsigned w = dropout_word(seed, idx >> 1);
    return ((idx & 1) ? (w >> 16) : (w & 0xffffu)) >= thresh16;
}
// The same word for pair indices below 2^32 (every tensor of the hot path): the seed-dependent part is loop-invariant.
//   dropout_word(seed, q) == dropout_word32(dropout_h0(seed), (unsigned)q)   for q < 2^32
__device__ __forceinline__ unsigned dropout_h0(unsigned long long seed) {
    return mix32((unsigned)seed) ^ ((unsigned)(seed >> 32) * 0x9E3779B9u);
}
#ifdef GN_ABL_NO_HASH         // timing-only ablation: what does the mask hash cost?
__device__ __forceinline__ unsigned dropout_word32(unsigned h0, unsigned pair) { return (pair ^ h0) * 0x10001u; }
#else
__device__ __forceinline__ unsigned dropout_word32(unsigned h0, unsigned pair) { return mix32(pair ^ h0); }
#endif
static inline unsigned dropout_thresh16(float p) {
    const double th = (double)p * 65536.0;
    return th <= 0 ? 0u : (th >= 65536.0 ? 65536u : (unsigned)(th + 0.5));
}
// d/dz [z*sigmoid(z)] with the hardware reciprocal (1 ulp) instead of the IEEE division
__device__ __forceinline__ float silu_grad_fast_(float z) {
#ifdef GN_ABL_NO_TRANS
    const float s = fminf(fmaxf(0.5f + 0.25f * z, 0.f), 1.f);
#else
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
#endif
    return s * (1.0f + z * (1.0f - s));
}

// "Done once per DEVICE" flag for the launchers' hipFuncSetAttribute(MaxDynamicSharedMemorySize) opt-ins: the attribute belongs to the
// (function, device) pair, so a process that drives a second GPU has to set it there too.  Reads like the bool it replaces
// (`static DevOnce set; if (!set) { ...; set = true; }`); devices are folded modulo 64, setting an attribute twice is harmless.
#include <atomic>
struct DevOnce {
    std::atomic<unsigned long long> mask{0};
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool operator!() const { return !(mask.load(std::memory_order_acquire) & bit()); }
    DevOnce& operator=(bool) { mask.fetch_or(bit(), std::memory_order_release); return *this; }
};

static inline int check_launch() { return hipGetLastError() == hipSuccess ? DDPM_OK : DDPM_ERR_LAUNCH; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// optim.hip: total_sq[0] = sum of partial[0 .. n) in a fixed order, total_sq[1 .. 63] = 0 (one 1024-thread block)
int ddpm_sumsq_finish_launch(const float* partial, int n, float* total_sq, void* stream);
// probe.hip: compute units the persistent launchers leave to a concurrent collective (ddpm_set_reserved_cus); 0 by default
int ddpm_reserved_cus();
static inline int ddpm_cu_budget(int wanted) { const int left = 256 - ddpm_reserved_cus(); return wanted < left ? wanted : (left > 1 ? left : 1); }
// pointwise.hip: launcher of the persistent 1x1-conv kernel (-1: geometry not covered)
int ddpm_pointwise_launch(const void* x, long long x_ld, const void* w, void* y, long long y_ld, const float* bias, const void* residual,
                          long long res_ld, int accumulate, int M, int N, int K, int dry, void* stream);
// edgeconv.hip: launchers of the few-output-channel / few-input-channel 3x3 kernels (-1: geometry not covered)
int ddpm_edgeconv_few_out_launch(const void* x, long long x_ld, const void* w, void* y, const float* bias, int B, int H, int W, int C, int N,
                                 int dry, void* stream);
int ddpm_edgeconv_few_in_launch(const void* x, long long x_ld, const void* w, void* y, long long y_ld, const float* bias, int B, int H, int W,
                                int C, int N, int dry, void* stream);
// conv3x3.hip: launcher of the persistent stationary-halo 3x3 kernel (-1: geometry not covered)
int ddpm_conv3x3_stream_launch(const void* x, long long x_ld, const void* w, void* y, long long y_ld, const float* bias, const float* rowbias,
                               long long rowbias_ld, const void* residual, long long res_ld, int accumulate, int B, int H, int W, int C, int N,
                               int upsample, int xcd, int dry, void* stream);
