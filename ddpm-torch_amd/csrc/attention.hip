// Single-head attention of AttentionBlock (ddpm_torch/models/unet.py:41-52 of tqch/ddpm-torch) WITHOUT any L x L tensor in
// memory — forward with the row log-sum-exp saved for training, and a flash-style backward in two kernels (gfx950, bf16).
//
//     S = Q K^T / sqrt(C)      P = softmax_keys(S)      O = P V                                  (forward)
//     dV = P^T dO      dP = dO V^T      dS = P o (dP - rowsum(dP o P))      dQ = dS K / sqrt(C)      dK = dS^T Q / sqrt(C)
//
// The reference's attention runs at 16x16 (L = 256; 8x8 / 4x4 in the middle blocks), so ONE row block of the score matrix —
// 128 rows x L columns — fits on chip.  All three kernels are the same two products around an LDS-resident bf16 tile:
//   P1  T[128 x L]  = X_blk[128 x C] . Y[L x C]^T     both operands channel-contiguous: plain ds_read_b128 fragments;
//                     the accumulators stay in registers for the element-wise step (softmax / exp(S - lse) / dS) and are then
//                     written to the tile as bf16;
//   P2  Z[128 x C]  = tile[128 x L] . Y[L x C]        Y is pixel-major in memory (k is the slow index): staged as stored and
//                     read with the hardware transpose (ds_read_b64_tr_b16);
// with the row-block role played by queries (forward, dQ) or by keys (dK / dV: the transposed score block K_blk Q^T, whose
// softmax normalisation comes from the saved log-sum-exp and whose row-sum term is D[q] = sum_c dO[q,c] O[q,c]).
// 512 threads = 8 waves as 2 (row halves of 64) x 4 (column tiles, round robin); operand chunks are register-staged one chunk
// ahead into a double-buffered LDS stage (one barrier per chunk).  Every L <= 256 and C <= 512 with C % 32 == 0, L % 16 == 0
// goes through here (rows / keys beyond L are masked, loads beyond the tensors return zeros through the buffer descriptor).
#include "common.h"
#include <string.h>

namespace {

constexpr unsigned OOB = 0x7ffffff0u;
constexpr int RB = 128;                        // rows of the score block owned by a workgroup
constexpr int NT = 512;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Opnd {                                   // [L rows][C] bf16, row pitch ld (elements); rsrc spans one image
    __amdgpu_buffer_rsrc_t rsrc;
    long long ld;
};

__device__ __forceinline__ Opnd make_opnd(const bf16_t* base, long long ld, int L, int C) {
    const unsigned long long ad = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
    const long long bytes = ((long long)(L - 1) * ld + C) * 2;
    Opnd o;
    o.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    o.ld = ld;
    return o;
}
__device__ __forceinline__ u32x4 ldv(const Opnd& o, unsigned off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(o.rsrc, off, 0, 0));
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

struct AttnArgs {
    const bf16_t *q, *k, *v;  long long qkv_ld;   // packed projection buffer: q at +0, k at +C, v at +2C (pointers already offset)
    const bf16_t *o, *d_o;    long long o_ld, do_ld;
    bf16_t *out;              long long out_ld;    // forward: O
    bf16_t *dq, *dk, *dv;     long long dqkv_ld;
    float *lse, *dvec;                             // [B][L] fp32: row log-sum-exp of the scaled logits; D = rowsum(dO o O)
    int B, L, C, Lpad;                             // Lpad = L rounded up to 32
    float scale;
};

// ---- LDS layout
// tile : bf16 [RB][Lpad], 16-byte chunks XOR-swizzled by (row & KM) (KM = min(Lpad / 8, 16) - 1): the 16 rows a ds_read_b128
//        lane group touches at one k position fall on 16 different 16-byte slots
// stage: 2 buffers; P1 chunk = X [RB][64 c] + Y [Lpad][64 c] with 128-byte rows and the gemm kernels' swizzle ((row >> 1) & 7);
//        P2 chunk = Y [KR][C] as stored (rows of 2C bytes), chunks XOR-swizzled by (krow & 3) << 2 for the transpose reads
__device__ __forceinline__ int tile_addr(int row, int col, int pitch_b, int km) {
    return row * pitch_b + ((((col >> 3) ^ (row & km))) << 4) + (col & 7) * 2;
}

// T accumulators of P1: acc[i][j], i = row tile (2 per wave: rows wr*64 + i*32), j-th column tile of this wave (ct = wc + 4 j)
template <int NJ>
struct P1 {
    f32x16 acc[2][NJ];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x16)(0.f);
    }
};

// P1: acc += X_blk[rows r0 .. r0+127][:] . Y[0 .. Lpad)[:]^T over all channels, 64 at a time.
template <int NJ>
__device__ __forceinline__ void product_nt(P1<NJ>& t, const Opnd& X, int r0, const Opnd& Y, int Lpad, int C, char* stage, int tid) {
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int xs_bytes = RB * 128, ys_bytes = Lpad * 128, buf_bytes = xs_bytes + ys_bytes;
    u32x4 vx[2], vy[4];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + NT * i, row = v >> 3, lc = (v & 7) ^ ((row >> 1) & 7);
            const int c = c0 + lc * 8;
            vx[i] = ldv(X, c < C ? (unsigned)(((long long)(r0 + row) * X.ld + c) * 2) : OOB);    // rows beyond the tensor: zeros (descriptor range)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid + NT * i, row = v >> 3, lc = (v & 7) ^ ((row >> 1) & 7);
            const int c = c0 + lc * 8;
            vy[i] = (row < Lpad && c < C) ? ldv(Y, (unsigned)(((long long)row * Y.ld + c) * 2)) : zero16();
        }
    };
    auto put = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(buf + (tid + NT * i) * 16) = vx[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((tid + NT * i) < Lpad * 8) *reinterpret_cast<u32x4*>(buf + xs_bytes + (tid + NT * i) * 16) = vy[i];
    };
    const int nchunks = (C + 63) / 64;
    fetch(0);
    put(stage);
    __syncthreads();
    const int sw = (lane >> 5) ^ ((lane >> 1) & 7);
    for (int ch = 0; ch < nchunks; ++ch) {
        const char* cur = stage + (ch & 1) * buf_bytes;
        if (ch + 1 < nchunks) fetch((ch + 1) * 64);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            u32x4 fa[2], fb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[i] = *reinterpret_cast<const u32x4*>(cur + (wr * 64 + i * 32 + (lane & 31)) * 128 + (((2 * kc) ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ct = wc + 4 * j;
                fb[j] = ct * 32 < Lpad ? *reinterpret_cast<const u32x4*>(cur + xs_bytes + (ct * 32 + (lane & 31)) * 128 + (((2 * kc) ^ sw) << 4)) : zero16();
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    t.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fa[i]), as_frag(fb[j]), t.acc[i][j], 0, 0, 0);
        }
        if (ch + 1 < nchunks) put(stage + ((ch + 1) & 1) * buf_bytes);
        __syncthreads();
    }
}

// P2: Z[128 x C] = tile[128 x Lpad] . Y[0 .. Lpad)[0 .. C): acc2[i][j] for column tiles ct = wc + 4 j (32 channels each).
template <int NJ2>
__device__ __forceinline__ void product_nn(f32x16 (&z)[2][NJ2], const char* tile, int pitch_b, int km, const Opnd& Y, int L, int Lpad, int C,
                                           char* stage, int tid) {
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int KR = C > 256 ? 32 : 64;                   // k rows per chunk: KR * C * 2 <= 32 KiB
    const int row_b = C * 2, cpr = C >> 3;              // bytes / 16-byte chunks per staged row
    const int nvec = KR * cpr;                          // 16-byte vectors per chunk (<= 4 per thread)
    const int skey = C % 128 == 0 ? 2 : 0;              // chunk swizzle (krow & 3) << 2: only when a row is a multiple of 16 chunks
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ2; ++j) z[i][j] = (f32x16)(0.f);
    u32x4 vy[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid + NT * i;
            const int kr = v / cpr, pc = v - kr * cpr;
            const int lc = skey ? (pc ^ ((kr & 3) << 2)) : pc;
            vy[i] = (v < nvec && k0 + kr < L) ? ldv(Y, (unsigned)(((long long)(k0 + kr) * Y.ld + lc * 8) * 2)) : zero16();
        }
    };
    auto put = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tid + NT * i < nvec) *reinterpret_cast<u32x4*>(buf + (tid + NT * i) * 16) = vy[i];
    };
    const int nchunks = (Lpad + KR - 1) / KR;
    const int buf_bytes = KR * row_b;
    fetch(0);
    put(stage);
    __syncthreads();
    const int i16 = lane & 15;
    for (int ch = 0; ch < nchunks; ++ch) {
        const char* cur = stage + (ch & 1) * buf_bytes;
        if (ch + 1 < nchunks) fetch((ch + 1) * KR);
        for (int ks = 0; ks < KR / 16; ++ks) {
            const int kg = ch * KR + ks * 16;                          // first k (column of the tile) of this step
            if (kg >= Lpad) break;
            u32x4 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wr * 64 + i * 32 + (lane & 31);
                fa[i] = *reinterpret_cast<const u32x4*>(tile + row * pitch_b + ((((kg >> 3) + (lane >> 5)) ^ (row & km)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ2; ++j) {
                const int ct = wc + 4 * j;
                if (ct * 32 >= C) continue;
                const int mcol = ct * 32 + ((lane >> 4) & 1) * 16 + (i16 & 3) * 4;
                const int krow = ks * 16 + (lane >> 5) * 8 + (i16 >> 2);
                const int pch = skey ? ((mcol >> 3) ^ ((krow & 3) << 2)) : (mcol >> 3);
                const char* p = cur + krow * row_b + pch * 16 + (mcol & 7) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * row_b));
                u32x4 fb;
                fb.x = __builtin_bit_cast(uint2, lo).x; fb.y = __builtin_bit_cast(uint2, lo).y;
                fb.z = __builtin_bit_cast(uint2, hi).x; fb.w = __builtin_bit_cast(uint2, hi).y;
#pragma unroll
                for (int i = 0; i < 2; ++i) z[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fa[i]), as_frag(fb), z[i][j], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) put(stage + ((ch + 1) & 1) * buf_bytes);
        __syncthreads();
    }
}

// accumulator (reg r, lane l) of a 32x32 tile = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int NJ, typename F>
__device__ __forceinline__ void write_tile(const P1<NJ>& t, char* tile, int pitch_b, int km, int Lpad, int tid, F&& f) {
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ct = wc + 4 * j;
            if (ct * 32 >= Lpad) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + i * 32 + acc_row(r, lane), col = ct * 32 + (lane & 31);
                *reinterpret_cast<bf16_t*>(tile + tile_addr(row, col, pitch_b, km)) = f2bf(f(t.acc[i][j][r], i, j, r, row, col));
            }
        }
}

template <int NJ2>
__device__ __forceinline__ void store_rows(const f32x16 (&z)[2][NJ2], bf16_t* out, long long ld, int r0, int L, int C, int tid) {
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            const int ct = wc + 4 * j;
            if (ct * 32 >= C) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + wr * 64 + i * 32 + acc_row(r, lane);
                if (row < L) out[(long long)row * ld + ct * 32 + (lane & 31)] = f2bf(z[i][j][r]);
            }
        }
}

// LDS carve-up shared by the three kernels
struct Lds {
    char* tile; char* stage; float* red;            // red: [2 * 4][RB] fp32 scratch for the cross-wave row reductions + [RB] row vector
    int pitch_b, km;
};
__device__ __forceinline__ Lds carve(char* smem, int Lpad) {
    Lds l;
    l.pitch_b = Lpad * 2;
    const int nch = Lpad >> 3;
    l.km = (nch < 16 ? nch : 16) - 1;
    l.tile = smem;
    l.stage = smem + RB * l.pitch_b;
    l.red = nullptr;
    return l;
}

// ------------------------------------------------------------------------------------------------ forward
template <int NJ1, int NJ2>
__global__ __launch_bounds__(NT, 2)
void attn_fwd_lse_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int b = blockIdx.y, r0 = blockIdx.x * RB;
    Lds l = carve(smem, a.Lpad);
    const Opnd Q = make_opnd(a.q + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd K = make_opnd(a.k + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd V = make_opnd(a.v + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    P1<NJ1> s;
    s.zero();
    product_nt<NJ1>(s, Q, r0, K, a.Lpad, a.C, l.stage, tid);
    // ---- softmax over the keys of each row.  A row lives in ONE lane half of the 4 column-tile waves of its row half: reduce over
    // this wave's columns (registers + butterfly over 32 lanes), then over the 4 waves through LDS.
    float* red = reinterpret_cast<float*>(l.stage);            // the stage is idle between the products: [4 wc][RB] max, then sums
    float rmax[2][16], rsum[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) {
                const int col = (wc + 4 * j) * 32 + (lane & 31);
                const float v = col < a.L ? s.acc[i][j][r] * a.scale : -3.0e38f;
                s.acc[i][j][r] = v;
                m = fmaxf(m, v);
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            rmax[i][r] = m;
            if ((lane & 31) == 0) red[wc * RB + wr * 64 + i * 32 + acc_row(r, lane)] = m;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + acc_row(r, lane);
            const float m = fmaxf(fmaxf(red[row], red[RB + row]), fmaxf(red[2 * RB + row], red[3 * RB + row]));
            rmax[i][r] = m;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) {
                const int col = (wc + 4 * j) * 32 + (lane & 31);
                const float e = col < a.L ? __expf(s.acc[i][j][r] - m) : 0.f;
                s.acc[i][j][r] = e;
                sum += e;
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            rsum[i][r] = sum;
        }
    __syncthreads();                                            // everyone has read the maxima: reuse the scratch for the sums
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((lane & 31) == 0) red[wc * RB + wr * 64 + i * 32 + acc_row(r, lane)] = rsum[i][r];
    __syncthreads();
    float rinv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + acc_row(r, lane);
            const float tot = (red[row] + red[RB + row]) + (red[2 * RB + row] + red[3 * RB + row]);       // fixed order
            rinv[i][r] = 1.0f / tot;
            if (a.lse && wc == 0 && (lane & 31) == 0 && r0 + row < a.L) a.lse[(long long)b * a.L + r0 + row] = rmax[i][r] + __logf(tot);
        }
    __syncthreads();                                            // scratch reads done before the stage is refilled by the next product
    write_tile<NJ1>(s, l.tile, l.pitch_b, l.km, a.Lpad, tid, [&](float e, int i, int, int r, int, int) { return e * rinv[i][r]; });
    __syncthreads();
    f32x16 z[2][NJ2];
    product_nn<NJ2>(z, l.tile, l.pitch_b, l.km, V, a.L, a.Lpad, a.C, l.stage, tid);
    store_rows<NJ2>(z, a.out + (long long)b * a.L * a.out_ld, a.out_ld, r0, a.L, a.C, tid);
}

// ------------------------------------------------------------------------------------------------ backward: dQ (and D)
template <int NJ1, int NJ2>
__global__ __launch_bounds__(NT, 2)
void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int b = blockIdx.y, r0 = blockIdx.x * RB;
    Lds l = carve(smem, a.Lpad);
    const Opnd Q = make_opnd(a.q + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd K = make_opnd(a.k + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd V = make_opnd(a.v + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd dO = make_opnd(a.d_o + (long long)b * a.L * a.do_ld, a.do_ld, a.L, a.C);
    // ---- D[q] = sum_c dO[q,c] O[q,c] for the block's rows: 4 threads per row, written to memory for the dK / dV kernel too
    float* dloc = reinterpret_cast<float*>(l.tile);               // [RB] (the tile is not in use yet)
    {
        const int row = tid >> 2, part = tid & 3, q = r0 + row;
        float acc = 0.f;
        if (q < a.L) {
            const bf16_t* po = a.o + ((long long)b * a.L + q) * a.o_ld;
            const bf16_t* pd = a.d_o + ((long long)b * a.L + q) * a.do_ld;
            for (int c = part * 8; c < a.C; c += 32) {
                float fo[8], fd[8];
                Elem<bf16_t>::unpack(ldg16(po + c), fo); Elem<bf16_t>::unpack(ldg16(pd + c), fd);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += fo[e] * fd[e];
            }
        }
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
        if (part == 0) { dloc[row] = acc; if (q < a.L) a.dvec[(long long)b * a.L + q] = acc; }
    }
    __syncthreads();
    float drow[2][16], lrow[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + acc_row(r, lane), q = r0 + row;
            drow[i][r] = dloc[row];
            lrow[i][r] = q < a.L ? a.lse[(long long)b * a.L + q] : 0.f;
        }
    __syncthreads();
    P1<NJ1> p;
    p.zero();
    product_nt<NJ1>(p, Q, r0, K, a.Lpad, a.C, l.stage, tid);      // S
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (wc + 4 * j) * 32 + (lane & 31);
                p.acc[i][j][r] = col < a.L ? __expf(p.acc[i][j][r] * a.scale - lrow[i][r]) : 0.f;      // P
            }
    P1<NJ1> dp;
    dp.zero();
    product_nt<NJ1>(dp, dO, r0, V, a.Lpad, a.C, l.stage, tid);    // dP
    write_tile<NJ1>(dp, l.tile, l.pitch_b, l.km, a.Lpad, tid,
                    [&](float v, int i, int j, int r, int, int) { return p.acc[i][j][r] * (v - drow[i][r]) * a.scale; });      // dS / sqrt(C)
    __syncthreads();
    f32x16 z[2][NJ2];
    product_nn<NJ2>(z, l.tile, l.pitch_b, l.km, K, a.L, a.Lpad, a.C, l.stage, tid);
    store_rows<NJ2>(z, a.dq + (long long)b * a.L * a.dqkv_ld, a.dqkv_ld, r0, a.L, a.C, tid);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <int NJ1, int NJ2>
__global__ __launch_bounds__(NT, 2)
void attn_bwd_dkv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wc = wave & 3;
    const int b = blockIdx.y, r0 = blockIdx.x * RB;              // r0: first KEY of the block
    Lds l = carve(smem, a.Lpad);
    const Opnd Q = make_opnd(a.q + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd K = make_opnd(a.k + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd V = make_opnd(a.v + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd dO = make_opnd(a.d_o + (long long)b * a.L * a.do_ld, a.do_ld, a.L, a.C);
    // per-COLUMN (= per query) constants of the transposed score block: one query per lane and column tile
    float lcol[NJ1], dcol[NJ1];
#pragma unroll
    for (int j = 0; j < NJ1; ++j) {
        const int q = (wc + 4 * j) * 32 + (lane & 31);
        lcol[j] = q < a.L ? a.lse[(long long)b * a.L + q] : 0.f;
        dcol[j] = q < a.L ? a.dvec[(long long)b * a.L + q] : 0.f;
    }
    P1<NJ1> p;
    p.zero();
    product_nt<NJ1>(p, K, r0, Q, a.Lpad, a.C, l.stage, tid);      // S^T = K_blk Q^T
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (wc + 4 * j) * 32 + (lane & 31);
                p.acc[i][j][r] = q < a.L ? __expf(p.acc[i][j][r] * a.scale - lcol[j]) : 0.f;            // P^T
            }
    write_tile<NJ1>(p, l.tile, l.pitch_b, l.km, a.Lpad, tid, [&](float v, int, int, int, int, int) { return v; });
    __syncthreads();
    f32x16 z[2][NJ2];
    product_nn<NJ2>(z, l.tile, l.pitch_b, l.km, dO, a.L, a.Lpad, a.C, l.stage, tid);                     // dV = P^T dO
    store_rows<NJ2>(z, a.dv + (long long)b * a.L * a.dqkv_ld, a.dqkv_ld, r0, a.L, a.C, tid);
    P1<NJ1> dp;
    dp.zero();
    product_nt<NJ1>(dp, V, r0, dO, a.Lpad, a.C, l.stage, tid);    // dP^T = V_blk dO^T     (every wave is past its tile reads: product_nn ends in a barrier)
    write_tile<NJ1>(dp, l.tile, l.pitch_b, l.km, a.Lpad, tid,
                    [&](float v, int i, int j, int r, int, int) { return p.acc[i][j][r] * (v - dcol[j]) * a.scale; });        // dS^T / sqrt(C)
    __syncthreads();
    product_nn<NJ2>(z, l.tile, l.pitch_b, l.km, Q, a.L, a.Lpad, a.C, l.stage, tid);                      // dK = dS^T Q
    store_rows<NJ2>(z, a.dk + (long long)b * a.L * a.dqkv_ld, a.dqkv_ld, r0, a.L, a.C, tid);
}

static int check_common(const void* qkv, long long ld, int B, int L, int C, int dtype) {
    if (!qkv) return DDPM_ERR_NULL;
    if (dtype != DDPM_BF16) return DDPM_ERR_DTYPE;
    if (B <= 0 || L <= 0 || L > 256 || L % 16 || C <= 0 || C > 512 || C % 32 || ld < 3 * C) return DDPM_ERR_SHAPE;
    if (!aligned16(qkv) || ld % 8) return DDPM_ERR_ALIGN;
    if ((long long)L * ld * 2 > 0x7ffffff0ll) return DDPM_ERR_SHAPE;
    return DDPM_OK;
}

static size_t lds_bytes(int Lpad, int C) {
    const size_t tile = (size_t)RB * Lpad * 2;
    const size_t p1 = 2 * (size_t)(RB * 128 + Lpad * 128);
    const size_t p2 = 2 * (size_t)((C > 256 ? 32 : 64) * C * 2);
    const size_t red = 4 * RB * sizeof(float);
    size_t stage = p1 > p2 ? p1 : p2;
    if (stage < red) stage = red;
    return tile + stage;
}

#define ATTN_DISPATCH(KERNEL, grid, lds, st, a)                                                                            \
    do {                                                                                                                   \
        const int nj1 = (a.Lpad / 32 + 3) / 4, nj2 = (a.C / 32 + 3) / 4;                                                   \
        auto go = [&](auto k) -> int {                                                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DDPM_ERR_LAUNCH; \
            hipLaunchKernelGGL(k, grid, dim3(NT), lds, st, a);                                                             \
            return check_launch();                                                                                         \
        };                                                                                                                 \
        if (nj1 <= 1 && nj2 <= 1) return go(&KERNEL<1, 1>);                                                                \
        if (nj1 <= 1 && nj2 <= 2) return go(&KERNEL<1, 2>);                                                                \
        if (nj1 <= 1) return go(&KERNEL<1, 4>);                                                                            \
        if (nj2 <= 1) return go(&KERNEL<2, 1>);                                                                            \
        if (nj2 <= 2) return go(&KERNEL<2, 2>);                                                                            \
        return go(&KERNEL<2, 4>);                                                                                          \
    } while (0)

}  // namespace

// Forward with the row log-sum-exp (training; also serves every geometry the inference kernel ddpm_attention_fwd does not):
//   out[b][i][:] = sum_j softmax_j(q_i . k_j * scale) v_j ;  lse[b][i] = log sum_j exp(q_i . k_j * scale)  (optional)
// qkv: packed [B][L][ld] with q at channel 0, k at C, v at 2C.  bf16; L <= 256, L % 16 == 0; C <= 512, C % 32 == 0.
extern "C" int ddpm_attention_fwd_lse(const void* qkv, long long ld, void* out, long long out_ld, float* lse, int B, int L, int C, float scale,
                                      int dtype, void* stream) {
    int rc = check_common(qkv, ld, B, L, C, dtype);
    if (rc) return rc;
    if (!out) return DDPM_ERR_NULL;
    if (!aligned16(out) || out_ld % 8) return DDPM_ERR_ALIGN;
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)qkv; a.k = a.q + C; a.v = a.q + 2 * C; a.qkv_ld = ld;
    a.out = (bf16_t*)out; a.out_ld = out_ld; a.lse = lse;
    a.B = B; a.L = L; a.C = C; a.Lpad = (L + 31) / 32 * 32; a.scale = scale;
    const dim3 grid((L + RB - 1) / RB, B);
    const size_t lds = lds_bytes(a.Lpad, C);
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(attn_fwd_lse_kernel, grid, lds, st, a);
}

// Backward of the above given the saved output o, its gradient d_o and lse: writes dq / dk / dv into the packed gradient buffer
// dqkv [B][L][dqkv_ld] (dq at channel 0, dk at C, dv at 2C).  dvec: [B][L] fp32 workspace (receives D = rowsum(dO o O)).
extern "C" int ddpm_attention_bwd(const void* qkv, long long ld, const void* o, long long o_ld, const void* d_o, long long do_ld,
                                  const float* lse, float* dvec, void* dqkv, long long dqkv_ld, int B, int L, int C, float scale,
                                  int dtype, void* stream) {
    int rc = check_common(qkv, ld, B, L, C, dtype);
    if (rc) return rc;
    if (!o || !d_o || !lse || !dvec || !dqkv) return DDPM_ERR_NULL;
    if (!aligned16(o) || !aligned16(d_o) || !aligned16(dqkv) || o_ld % 8 || do_ld % 8 || dqkv_ld % 8 || dqkv_ld < 3 * C) return DDPM_ERR_ALIGN;
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)qkv; a.k = a.q + C; a.v = a.q + 2 * C; a.qkv_ld = ld;
    a.o = (const bf16_t*)o; a.o_ld = o_ld; a.d_o = (const bf16_t*)d_o; a.do_ld = do_ld;
    a.dq = (bf16_t*)dqkv; a.dk = a.dq + C; a.dv = a.dq + 2 * C; a.dqkv_ld = dqkv_ld;
    a.lse = const_cast<float*>(lse); a.dvec = dvec;
    a.B = B; a.L = L; a.C = C; a.Lpad = (L + 31) / 32 * 32; a.scale = scale;
    const dim3 grid((L + RB - 1) / RB, B);
    const size_t lds = lds_bytes(a.Lpad, C);
    hipStream_t st = (hipStream_t)stream;
    {
        auto run = [&]() -> int { ATTN_DISPATCH(attn_bwd_dq_kernel, grid, lds, st, a); };
        rc = run();
        if (rc) return rc;
    }
    ATTN_DISPATCH(attn_bwd_dkv_kernel, grid, lds, st, a);
}
