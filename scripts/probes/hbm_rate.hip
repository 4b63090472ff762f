// Probe: what do the HBM-bound kernels of the step have to work with?  GroupNorm (LDS-staged, csrc/norm.hip) and the 1x1 convs run
// as chip-wide PHASES — every block loads its slice, then every block stores — and reach ~3 TB/s, while the loop-structured kernels
// (Adam, wgrad_reduce) reach 4.5-5.5 TB/s.  This probe separates the candidates: pure read / pure write / copy, as a grid-stride
// stream and as one burst per block (contiguous 64-KiB chunks or the GroupNorm slice pattern: 1024 rows of 64 B, 256 B apart),
// plain vs nontemporal stores, a 32-MiB working set reused by every launch (Infinity-Cache warm, like an activation the previous
// kernel just wrote) vs 16 rotating buffer pairs (1 GiB: HBM proper), and the burst kernel with half of the blocks delayed.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_rate hbm_rate.hip      (run on the GPU box; prints one line per variant)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr size_t TENSOR = 32u << 20;                     // bytes per buffer: 128 x 32 x 32 x 128 bf16
constexpr int ROT = 16;

__device__ __forceinline__ u32x4 ld(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
template <int NT> __device__ __forceinline__ void st(char* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = v;
}

// ---- loop-structured: U vectors in flight per thread, grid-stride over the buffer
template <int U>
__global__ __launch_bounds__(256) void rd_stream(const char* __restrict__ src, char* __restrict__ dst, size_t bytes) {
    const size_t nvec = bytes / 16, stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride * U) {
        u32x4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = v + u * stride < nvec ? ld(src + (v + u * stride) * 16) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= t[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) st<0>(dst, acc);
}
template <int U, int NT>
__global__ __launch_bounds__(256) void wr_stream(const char* __restrict__ src, char* __restrict__ dst, size_t bytes) {
    const size_t nvec = bytes / 16, stride = (size_t)gridDim.x * 256;
    const u32x4 val = {threadIdx.x, blockIdx.x, 3, 4};
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (v + u * stride < nvec) st<NT>(dst + (v + u * stride) * 16, val);
    }
}
template <int U, int NT>
__global__ __launch_bounds__(256) void cp_stream(const char* __restrict__ src, char* __restrict__ dst, size_t bytes) {
    const size_t nvec = bytes / 16, stride = (size_t)gridDim.x * 256;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride * U) {
        u32x4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = v + u * stride < nvec ? ld(src + (v + u * stride) * 16) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) if (v + u * stride < nvec) st<NT>(dst + (v + u * stride) * 16, t[u] + 1u);
    }
}

// ---- phase-structured: 512-thread blocks, 8 vectors per thread = 64 KiB per block, everything issued at once
// PAT 0: the block's 64 KiB are contiguous.  PAT 1: GroupNorm slice — block (sample b = blk / 4, chunk c = blk % 4) takes rows of
// 64 B (4 lanes) at 256-B pitch, 1024 rows.  PAT 2: the same with the XCD remap of gn_block_slice (chunks of a sample 8 ids apart).
template <int PAT>
__device__ __forceinline__ size_t burst_off(int i, int nblk) {
    const int tid = threadIdx.x;
    if (PAT == 0) return (size_t)blockIdx.x * 65536 + (size_t)(i * 512 + tid) * 16;
    int b, c;
    if (PAT == 1) { b = blockIdx.x >> 2; c = blockIdx.x & 3; }
    else { const int L = blockIdx.x, k = L >> 3; c = k & 3; b = (k >> 2) * 8 + (L & 7); }
    const int j = tid & 3, p = (tid >> 2) + i * 128;
    return (size_t)b * 262144 + (size_t)p * 256 + c * 64 + j * 16;
}
__device__ __forceinline__ void delay_if(int delay_clk, int nblk) {
    if (delay_clk > 0 && (int)blockIdx.x >= nblk / 2) {
        const long long t0 = wall_clock64();          // 100 MHz
        while (wall_clock64() - t0 < delay_clk) __builtin_amdgcn_s_sleep(8);
    }
}
// MODE 0: read only; 1: write only; 2: read, barrier, write (the forward GroupNorm skeleton); 3: read src AND dst, barrier, write (backward)
template <int PAT, int MODE, int NT>
__global__ __launch_bounds__(512) void burst(const char* __restrict__ src, char* __restrict__ dst, const char* __restrict__ src2, int delay_clk) {
    const int nblk = gridDim.x;
    delay_if(delay_clk, nblk);
    u32x4 t[8], t2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        t[i] = MODE != 1 ? ld(src + burst_off<PAT>(i, nblk)) : u32x4{(unsigned)i, threadIdx.x, blockIdx.x, 7};
        if (MODE == 3) t2[i] = ld(src2 + burst_off<PAT>(i, nblk));
    }
    if (MODE == 0) {
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= t[i];
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) st<0>(dst, acc);
        return;
    }
    if (MODE >= 2) __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) st<NT>(dst + burst_off<PAT>(i, nblk), MODE == 3 ? t[i] + t2[i] : t[i] + 1u);
}

// ---- the loop-structured alternative for the same slice work: 256-thread blocks, each takes 1/4 of a slice's rows (16 KiB), 2048 blocks
template <int NT>
__global__ __launch_bounds__(256) void slice_small(const char* __restrict__ src, char* __restrict__ dst) {
    const int blk = blockIdx.x, q = blk & 3, s = blk >> 2, b = s >> 2, c = s & 3, tid = threadIdx.x;
    u32x4 t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = q * 256 + (tid >> 2) + i * 64;
        t[i] = ld(src + (size_t)b * 262144 + (size_t)p * 256 + c * 64 + (tid & 3) * 16);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = q * 256 + (tid >> 2) + i * 64;
        st<NT>(dst + (size_t)b * 262144 + (size_t)p * 256 + c * 64 + (tid & 3) * 16, t[i] + 1u);
    }
}

static char* bufs[3][ROT];
template <typename F>
static void run(const char* name, double bytes, int rot, F&& launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch(i % rot);
    hipDeviceSynchronize();
    const int n = 32;
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) launch(i % rot);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / n;
    printf("%-58s ws=%4d MiB  %8.2f us  %6.2f TB/s\n", name, (int)(rot * 2 * (TENSOR >> 20)), us, bytes / us * 1e-6);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < ROT; ++r) {
            if (hipMalloc(&bufs[k][r], TENSOR) != hipSuccess) { printf("alloc failed\n"); return 1; }
            hipMemset(bufs[k][r], k + 1, TENSOR);
        }
    hipDeviceSynchronize();
    const double T = (double)TENSOR;
    for (int rot : {1, ROT}) {
#define SRC bufs[0][r]
#define DST bufs[1][r]
#define SR2 bufs[2][r]
        run("read   stream U=4 (2048 x 256)", T, rot, [&](int r) { hipLaunchKernelGGL(rd_stream<4>, dim3(2048), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("read   stream U=8 (2048 x 256)", T, rot, [&](int r) { hipLaunchKernelGGL(rd_stream<8>, dim3(2048), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("read   stream U=8 (1024 x 256)", T, rot, [&](int r) { hipLaunchKernelGGL(rd_stream<8>, dim3(1024), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("write  stream U=4 (2048 x 256)", T, rot, [&](int r) { hipLaunchKernelGGL((wr_stream<4, 0>), dim3(2048), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("write  stream U=4 nontemporal", T, rot, [&](int r) { hipLaunchKernelGGL((wr_stream<4, 1>), dim3(2048), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("copy   stream U=4 (2048 x 256)", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((cp_stream<4, 0>), dim3(2048), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("copy   stream U=4 nontemporal", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((cp_stream<4, 1>), dim3(2048), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("copy   stream U=8 (1024 x 256)", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((cp_stream<8, 0>), dim3(1024), dim3(256), 0, 0, SRC, DST, TENSOR); });
        run("burst  read   contiguous 64 KiB / block", T, rot, [&](int r) { hipLaunchKernelGGL((burst<0, 0, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read   GN slices", T, rot, [&](int r) { hipLaunchKernelGGL((burst<1, 0, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read   GN slices, XCD remap", T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 0, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  write  contiguous", T, rot, [&](int r) { hipLaunchKernelGGL((burst<0, 1, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  write  contiguous nontemporal", T, rot, [&](int r) { hipLaunchKernelGGL((burst<0, 1, 1>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  write  GN slices", T, rot, [&](int r) { hipLaunchKernelGGL((burst<1, 1, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  write  GN slices, XCD remap", T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 1, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read|write contiguous", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<0, 2, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read|write GN slices", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<1, 2, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read|write GN slices, XCD remap", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 2, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read|write GN slices, remap, nontemporal", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 2, 1>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("burst  read|write GN slices, remap, half delayed 4 us", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 2, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 400); });
        run("burst  read|write GN slices, remap, half delayed 8 us", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 2, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 800); });
        run("burst  2 reads|write GN slices, remap (backward)", 3 * T, rot, [&](int r) { hipLaunchKernelGGL((burst<2, 3, 0>), dim3(512), dim3(512), 0, 0, SRC, DST, SR2, 0); });
        run("slices as 2048 small blocks (16 KiB each)", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((slice_small<0>), dim3(2048), dim3(256), 0, 0, SRC, DST); });
        run("slices as 2048 small blocks, nontemporal", 2 * T, rot, [&](int r) { hipLaunchKernelGGL((slice_small<1>), dim3(2048), dim3(256), 0, 0, SRC, DST); });
        printf("\n");
    }
    return 0;
}
