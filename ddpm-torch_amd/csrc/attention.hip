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

// ---- operand staging: LDS-DMA rings with counted waits.  (First version: global -> registers -> LDS, one chunk ahead, one barrier per
// chunk.  With ONE block per CU — the grid is 2 x B blocks and the tile + stages fill the LDS — every chunk's load latency (~2 us from
// L2 / HBM) was exposed behind ~0.5 us of MFMA work: 41 us for the forward whose MFMA time is ~4 us.)  Now every chunk of a product
// is requested as early as the ring allows — P1 keeps two 48-KiB chunks in flight beside the one being multiplied (the score tile's
// LDS is free while P1 runs), P2 four 16-KiB chunks — and a wave waits for exactly the chunk it is about to read.
// All LDS reads of the products are inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of any ds_read it generates while an LDS-DMA
// may be outstanding (it cannot tell which bytes the DMA writes), which would drain the ring at every step.
__device__ __forceinline__ void wait_vm_rt(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
__device__ __forceinline__ u32x4 lds_rd16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ uint2 lds_rd_tr(unsigned addr) {          // ds_read_b64_tr_b16: see read_frag<T, true> in gemm.hip
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// (the fence keeps hipcc from moving an MFMA, which depends on the reads' registers but not on the wait, ahead of it)
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
template <int N> __device__ __forceinline__ void lds_wait_n() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }

#ifdef ATTN_TIMING
__device__ unsigned long long* g_attn_timing = nullptr;        // debug builds only (scripts/attn_timeline.py): [block][8] stamps
#define ATTN_STAMP(slot) do { if (g_attn_timing && threadIdx.x == 0) g_attn_timing[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define ATTN_STAMP(slot)
#endif

constexpr int P1_RING = 3, P2_RING = 5;
constexpr int P2_SLOT = 2 * NT * 16;          // every thread issues two DMA vectors per chunk (past the chunk: zeros), so a slot is 16 KiB whatever C
__device__ __host__ __forceinline__ int p1_slot_bytes(int Lpad) { return RB * 128 + ((Lpad + 63) / 64) * 8192; }      // X chunk + Y chunk in whole 512-vector instructions
__device__ __host__ __forceinline__ int p2_rows(int C) { return C > 256 ? 16 : (C > 128 ? 32 : 64); }                   // k rows per chunk: <= 1024 vectors (16 KiB)

// P1: acc += X_blk[rows r0 .. r0+127][:] . Y[0 .. Lpad)[:]^T over all channels, 64 at a time.  `lds` = the block's whole dynamic LDS:
// the ring occupies it from offset 0 (no score tile is live while a P1 runs).
template <int NJ, bool PIPE = true>
__device__ __forceinline__ void product_nt(P1<NJ>& t, const Opnd& X, int r0, const Opnd& Y, int Lpad, int C, char* lds, int tid) {
    const int lane = tid & 63, wr = (tid >> 8) & 1, wc = (tid >> 6) & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nyi = (Lpad + 63) / 64;                            // DMA instructions per thread for the Y chunk (rows >= L: zeros)
    const int xs_bytes = RB * 128, slot_bytes = p1_slot_bytes(Lpad), per = 2 + nyi;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int nchunks = (C + 63) / 64;
    // source offsets of this thread's vectors at chunk 0 (LDS position v = tid + 512 i -> row v >> 3, logical chunk (v & 7) ^ ((row >> 1) & 7))
    unsigned xo[2], yo[4]; int xc[2], yc[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + NT * i, row = v >> 3;
        xc[i] = ((v & 7) ^ ((row >> 1) & 7)) * 8;
        xo[i] = (unsigned)(((long long)(r0 + row) * X.ld + xc[i]) * 2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid + NT * i, row = v >> 3;
        yc[i] = ((v & 7) ^ ((row >> 1) & 7)) * 8;
        yo[i] = row < Lpad ? (unsigned)(((long long)row * Y.ld + yc[i]) * 2) : OOB;
    }
    auto issue = [&](int ch) {
        char* slot = lds + (ch % P1_RING) * slot_bytes;
        const int c0 = ch * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned o = c0 + xc[i] < C ? xo[i] + (unsigned)(c0 * 2) : OOB;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(X.rsrc, (__attribute__((address_space(3))) void*)(slot + (i * NT + wave * 64) * 16), 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < nyi) {
                unsigned o = (yo[i] != OOB && c0 + yc[i] < C) ? yo[i] + (unsigned)(c0 * 2) : OOB;
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(Y.rsrc, (__attribute__((address_space(3))) void*)(slot + xs_bytes + (i * NT + wave * 64) * 16), 16, o, 0, 0, 0);
            }
        }
    };
    issue(0);
    if (nchunks > 1) issue(1);
    const int sw = (lane >> 5) ^ ((lane >> 1) & 7);
    const unsigned arow = (unsigned)((wr * 64 + (lane & 31)) * 128), brow = (unsigned)(xs_bytes + (wc * 32 + (lane & 31)) * 128);
    bool jok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) jok[j] = (wc + 4 * j) * 32 < Lpad;
#ifdef ATTN_TIMING
    unsigned long long tq[5] = {0, 0, 0, 0, 0};
#define P1_T(k) do { if (ch == 1) tq[k] = clock64(); } while (0)
#else
#define P1_T(k)
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        P1_T(0);
        wait_vm_rt(ch + 1 < nchunks ? per : 0);                   // chunk ch has landed (this wave's part); chunk ch + 1 may stay in flight
        P1_T(1);
        __builtin_amdgcn_s_barrier();                             // ... everyone's part, and every wave is done with chunk ch - 1
        P1_T(2);
        if (ch + 2 < nchunks) issue(ch + 2);                      // into the slot of chunk ch - 1
        P1_T(3);
        const unsigned cur = lds0 + (unsigned)((ch % P1_RING) * slot_bytes);
        // fragment reads one K-step ahead of the MFMAs (two register sets): the wave's LDS latency hides behind its own four MFMAs
        // instead of only behind the partner wave's
        constexpr int NS = PIPE ? 2 : 1;                                 // (one set where the caller's accumulators leave no room)
        u32x4 fa[NS][2], fb[NS][NJ];
        auto rd = [&](int kc, int set) {
            const unsigned ko = (unsigned)(((2 * kc) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i] = lds_rd16(cur + arow + (unsigned)(i * 32 * 128) + ko);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[set][j] = lds_rd16(cur + (jok[j] ? brow + (unsigned)(j * 128 * 128) : (unsigned)xs_bytes) + ko);   // (column tiles past Lpad: any valid address, no MFMA)
        };
        rd(0, 0);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            constexpr int cs = NS == 2 ? 1 : 0;
            if (NS == 1) { if (kc > 0) rd(kc, 0); lds_wait_n<0>(); }
            else if (kc < 3) { rd(kc + 1, (kc + 1) & 1); lds_wait_n<2 + NJ>(); } else lds_wait_n<0>();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (jok[j]) t.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fa[kc & cs][i]), as_frag(fb[kc & cs][j]), t.acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        P1_T(4);
    }
#ifdef ATTN_TIMING
    if (g_attn_timing && threadIdx.x == 0 && tq[4])
        g_attn_timing[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = ((tq[1] - tq[0]) & 0xffff) | (((tq[2] - tq[1]) & 0xffff) << 16) | (((tq[3] - tq[2]) & 0xffff) << 32) | (((tq[4] - tq[3]) & 0xffff) << 48);
#endif
#undef P1_T
    __syncthreads();                                              // every wave is past its reads: the caller may reuse the LDS
}

// P2: Z[128 x C] = tile[128 x Lpad] . Y[0 .. Lpad)[0 .. C): acc2[i][j] for column tiles ct = wc + 4 j (32 channels each).
// `ring`: P2_RING slots of P2_SLOT bytes behind the tile.
template <int NJ2>
__device__ __forceinline__ void product_nn(f32x16 (&z)[2][NJ2], const char* tile, int pitch_b, int km, const Opnd& Y, int L, int Lpad, int C,
                                           char* ring, int tid) {
    const int lane = tid & 63, wr = (tid >> 8) & 1, wc = (tid >> 6) & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KR = p2_rows(C);
    const int row_b = C * 2, cpr = C >> 3;              // bytes / 16-byte chunks per staged row
    const int nvec = KR * cpr;                          // 16-byte vectors per chunk (<= 2 per thread)
    const int skey = C % 128 == 0 ? 2 : 0;              // chunk swizzle (krow & 3) << 2: only when a row is a multiple of 16 chunks
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ2; ++j) z[i][j] = (f32x16)(0.f);
    const int nchunks = (Lpad + KR - 1) / KR;
    int vkr[2]; unsigned vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + NT * i;
        const int kr = v / cpr, pc = v - kr * cpr;
        const int lc = skey ? (pc ^ ((kr & 3) << 2)) : pc;
        vkr[i] = v < nvec ? kr : (1 << 20);
        vo[i] = (unsigned)(((long long)kr * Y.ld + lc * 8) * 2);
    }
    const unsigned kstep = (unsigned)((long long)KR * Y.ld * 2);
    auto issue = [&](int ch) {
        char* slot = ring + (ch % P2_RING) * P2_SLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned o = ch * KR + vkr[i] < L ? vo[i] + (unsigned)ch * kstep : OOB;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Y.rsrc, (__attribute__((address_space(3))) void*)(slot + (i * NT + wave * 64) * 16), 16, o, 0, 0, 0);
        }
    };
    for (int c = 0; c < P2_RING - 1 && c < nchunks; ++c) issue(c);
    const int i16 = lane & 15;
    const unsigned tile0 = (unsigned)(size_t)tile, ring0 = (unsigned)(size_t)ring;
    bool jok[NJ2];
#pragma unroll
    for (int j = 0; j < NJ2; ++j) jok[j] = (wc + 4 * j) * 32 < C;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int ahead = min(ch + P2_RING - 2, nchunks - 1) - ch;      // chunks younger than ch already requested
        wait_vm_rt(2 * ahead);
        __builtin_amdgcn_s_barrier();
        if (ch + P2_RING - 1 < nchunks) issue(ch + P2_RING - 1);   // into the slot of chunk ch - 1 (everyone is past it)
        const unsigned cur = ring0 + (unsigned)((ch % P2_RING) * P2_SLOT);
        // fragment reads one k-step ahead of the MFMAs, as in P1 (a k-step past Lpad reads valid LDS and is never multiplied)
        constexpr int NS = NJ2 <= 2 ? 2 : 1;                             // (one register set for the 512-channel tensors: two would spill)
        u32x4 fa[NS][2]; uint2 fb[NS][NJ2][2];
        const int nks = min(KR / 16, (Lpad - ch * KR + 15) / 16);
        auto rd = [&](int ks, int set) {
            const int kg = ch * KR + ks * 16;                          // first k (column of the tile) of this step
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wr * 64 + i * 32 + (lane & 31);
                fa[set][i] = lds_rd16(tile0 + (unsigned)(row * pitch_b + ((((kg >> 3) + (lane >> 5)) ^ (row & km)) << 4)));
            }
#pragma unroll
            for (int j = 0; j < NJ2; ++j) {
                const int ct = jok[j] ? wc + 4 * j : wc;
                const int mcol = ct * 32 + ((lane >> 4) & 1) * 16 + (i16 & 3) * 4;
                const int krow = ks * 16 + (lane >> 5) * 8 + (i16 >> 2);
                const int pch = skey ? ((mcol >> 3) ^ ((krow & 3) << 2)) : (mcol >> 3);
                const unsigned p = cur + (unsigned)(krow * row_b + pch * 16 + (mcol & 7) * 2);
                fb[set][j][0] = lds_rd_tr(p); fb[set][j][1] = lds_rd_tr(p + (unsigned)(4 * row_b));
            }
        };
        rd(0, 0);
        for (int ks = 0; ks < nks; ks += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (ks + h >= nks) break;
                constexpr int cs = NS == 2 ? 1 : 0;
                const int cur_set = h & cs;
                if (NS == 1) { if (ks + h > 0) rd(ks + h, 0); lds_wait_n<0>(); }
                else if (ks + h + 1 < nks) { rd(ks + h + 1, (h + 1) & 1); lds_wait_n<2 + 2 * NJ2>(); } else lds_wait_n<0>();
#pragma unroll
                for (int j = 0; j < NJ2; ++j) {
                    if (!jok[j]) continue;
                    u32x4 b4; b4.x = fb[cur_set][j][0].x; b4.y = fb[cur_set][j][0].y; b4.z = fb[cur_set][j][1].x; b4.w = fb[cur_set][j][1].y;
#pragma unroll
                    for (int i = 0; i < 2; ++i) z[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fa[cur_set][i]), as_frag(b4), z[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
}

// Reduction of CNT per-lane values over the 2 * OFF lanes that share lane bits above OFF, all values at once: at every step a lane
// keeps the half of its values its side of the exchange is responsible for and sends the other half, so CNT - 1 exchanges (all of a
// step independent of each other) replace CNT butterflies of log2 steps.  Lane l ends with the total of value index l & (CNT - 1).
template <int CNT, int OFF, typename Op>
__device__ __forceinline__ float half_reduce(const float (&v)[CNT], int lane, Op&& op) {
    if constexpr (CNT == 1) return v[0];
    else {
        const bool bit = (lane & OFF) != 0;
        float o[CNT / 2];
#pragma unroll
        for (int k = 0; k < CNT / 2; ++k) {
            const float send = bit ? v[k] : v[k + CNT / 2], keep = bit ? v[k + CNT / 2] : v[k];
            o[k] = op(keep, __shfl_xor(send, OFF, 64));
        }
        return half_reduce<CNT / 2, OFF / 2>(o, lane, op);
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

// Block -> (sample b, first row r0).  The row blocks of one sample read the same K / V (P1's Y operand, P2's Y operand): consecutive
// workgroup ids go round-robin over the 8 XCDs, so with the plain (row block, sample) grid the two blocks of a 256-token sample sit
// on different L2s and each pulls K and V from the Infinity Cache.  Remapped, the blocks of a sample are ids 8 apart: same XCD,
// dispatched back to back — the second one's operand chunks hit in L2.
__device__ __forceinline__ void attn_block(int B, int& b, int& r0) {
    const int gx = gridDim.x;
    if (gx > 1 && (B & 7) == 0) {
        const int L = blockIdx.y * gx + blockIdx.x, k = L >> 3;
        r0 = (k % gx) * RB;
        b = (k / gx) * 8 + (L & 7);
    } else { b = blockIdx.y; r0 = blockIdx.x * RB; }
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
    int b, r0;
    attn_block(a.B, b, r0);
    Lds l = carve(smem, a.Lpad);
    const Opnd Q = make_opnd(a.q + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd K = make_opnd(a.k + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    const Opnd V = make_opnd(a.v + (long long)b * a.L * a.qkv_ld, a.qkv_ld, a.L, a.C);
    ATTN_STAMP(0);
    P1<NJ1> s;
    s.zero();
    product_nt<NJ1, (NJ2 <= 2)>(s, Q, r0, K, a.Lpad, a.C, smem, tid);
    ATTN_STAMP(1);
    // ---- softmax over the keys of each row.  A row lives in ONE lane half of the 4 column-tile waves of its row half: reduce over
    // this wave's columns (registers, then the 32 lanes of the half), then over the 4 waves through LDS.
    // The lane step is a HALVING butterfly over all 32 rows of the lane at once (half_reduce: 16 + 8 + 4 + 2 + 1 exchanges, lane k
    // ends with the total of row k) instead of 32 separate 5-step butterflies: those were 320 dependent ds_bpermute per thread and
    // pass, issued one row after the other — 15.7 us of the kernel's 41, whatever L (scripts/attn_timeline.py).
    float* red = reinterpret_cast<float*>(l.stage);            // the ring is idle between the products: [4 wc][RB] maxima, then [4 wc][RB] sums
    const int kk = lane & 31, myrow = wr * 64 + (kk >> 4) * 32 + acc_row(kk & 15, lane);
    float vals[32];
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
            vals[i * 16 + r] = m;
        }
    red[wc * RB + myrow] = half_reduce<32, 16>(vals, lane, [](float x, float y) { return fmaxf(x, y); });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + acc_row(r, lane);
            const float m = fmaxf(fmaxf(red[row], red[RB + row]), fmaxf(red[2 * RB + row], red[3 * RB + row]));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) {
                const int col = (wc + 4 * j) * 32 + (lane & 31);
                const float e = col < a.L ? __expf(s.acc[i][j][r] - m) : 0.f;
                s.acc[i][j][r] = e;
                sum += e;
            }
            vals[i * 16 + r] = sum;
        }
    float* rsum = red + 4 * RB;
    rsum[wc * RB + myrow] = half_reduce<32, 16>(vals, lane, [](float x, float y) { return x + y; });
    __syncthreads();
    float rinv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + acc_row(r, lane);
            const float tot = (rsum[row] + rsum[RB + row]) + (rsum[2 * RB + row] + rsum[3 * RB + row]);       // fixed order
            rinv[i][r] = __builtin_amdgcn_rcpf(tot);
        }
    if (a.lse && tid < RB && r0 + tid < a.L) {                  // one thread per row: coalesced store
        const float m = fmaxf(fmaxf(red[tid], red[RB + tid]), fmaxf(red[2 * RB + tid], red[3 * RB + tid]));
        const float tot = (rsum[tid] + rsum[RB + tid]) + (rsum[2 * RB + tid] + rsum[3 * RB + tid]);
        a.lse[(long long)b * a.L + r0 + tid] = m + __logf(tot);
    }
    __syncthreads();                                            // scratch reads done before the ring is refilled by the next product
    ATTN_STAMP(2);
    write_tile<NJ1>(s, l.tile, l.pitch_b, l.km, a.Lpad, tid, [&](float e, int i, int, int r, int, int) { return e * rinv[i][r]; });
    __syncthreads();
    ATTN_STAMP(3);
    f32x16 z[2][NJ2];
    product_nn<NJ2>(z, l.tile, l.pitch_b, l.km, V, a.L, a.Lpad, a.C, l.stage, tid);
    ATTN_STAMP(4);
    store_rows<NJ2>(z, a.out + (long long)b * a.L * a.out_ld, a.out_ld, r0, a.L, a.C, tid);
    ATTN_STAMP(5);
#ifdef ATTN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ATTN_STAMP(6);
}

// ------------------------------------------------------------------------------------------------ backward: dQ (and D)
template <int NJ1, int NJ2>
__global__ __launch_bounds__(NT, 2)
void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    int b, r0;
    attn_block(a.B, b, r0);
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
    product_nt<NJ1, (NJ2 <= 2)>(p, Q, r0, K, a.Lpad, a.C, smem, tid);      // S
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
    product_nt<NJ1, (NJ2 <= 2)>(dp, dO, r0, V, a.Lpad, a.C, smem, tid);    // dP
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
    int b, r0;                                                   // r0: first KEY of the block
    attn_block(a.B, b, r0);
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
    product_nt<NJ1, (NJ2 <= 2)>(p, K, r0, Q, a.Lpad, a.C, smem, tid);      // S^T = K_blk Q^T
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
    product_nt<NJ1, (NJ2 <= 2)>(dp, V, r0, dO, a.Lpad, a.C, smem, tid);    // dP^T = V_blk dO^T     (every wave is past its tile reads: product_nn ends in a barrier)
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
    const int nch1 = (C + 63) / 64, nch2 = (Lpad + p2_rows(C) - 1) / p2_rows(C);
    const size_t p1 = (size_t)(nch1 < P1_RING ? nch1 : P1_RING) * p1_slot_bytes(Lpad);            // from offset 0: no tile is live during a P1
    size_t p2 = (size_t)(nch2 < P2_RING ? nch2 : P2_RING) * P2_SLOT;                               // behind the tile
    const size_t red = 8 * RB * sizeof(float);
    if (p2 < red) p2 = red;
    return p1 > tile + p2 ? p1 : tile + p2;
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

#ifdef ATTN_TIMING
extern "C" int ddpm_debug_set_attn_timing(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_timing), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif

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
