// Probe: how fast can ONE CU pull an L2-resident operand tile on chip, by access pattern and mechanism?
// (the small-grid GEMM kernels are bound by exactly this: scripts/g64_timeline.py shows ~30 clk of CU time per 1-KiB LDS-DMA instruction)
// Operand = 256 rows x 4608 B (a 3x3 / 256-channel weight matrix, 1.18 MB, shared by all blocks -> L2-resident), a block walks 64 rows
// along K in 128-B steps; per wave-instruction (1 KiB):
//   mode 0: LDS-DMA, 8 rows x 128 B (what Loader::issue does)        mode 1: LDS-DMA, 1 KiB contiguous (upper bound of the mechanism)
//   mode 2: VGPR buffer_load_dwordx4, 8 rows x 128 B                  mode 3: VGPR, 32 rows x 32 B (a 32x32x16 MFMA fragment read straight from memory)
//   mode 4: VGPR, 16 rows x 64 B (a 16x16x32 fragment)                mode 5: VGPR, 1 KiB contiguous
// Build: hipcc --offload-arch=gfx950 -O3 -o feed_rate feed_rate.hip   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr unsigned PITCH = 4608, ROWS = 256;

// a K-step of the 64-row tile = 8 KiB = 8 wave-instructions; wave w takes instructions w, w + nw, ... of the stream
template <int MODE>
__device__ __forceinline__ unsigned offset_of(unsigned q, unsigned row0, int lane) {          // q = running instruction index of the block
    const unsigned kstep = (q >> 3) % (PITCH / 128), part = q & 7;
    unsigned row, col;
    if (MODE == 0 || MODE == 2) { row = part * 8 + (lane >> 3); col = (lane & 7) * 16; }
    else if (MODE == 3) { row = (part >> 2) * 32 + (lane & 31); col = (part & 3) * 32 + (lane >> 5) * 16; }
    else if (MODE == 4) { row = (part >> 1) * 16 + (lane & 15); col = (part & 1) * 64 + (lane >> 4) * 16; }
    else return ((q * 1024u) % (PITCH * 64u)) + row0 * PITCH + lane * 16;
    return (row0 + row) * PITCH + kstep * 128 + col;
}

template <int MODE, int D>
__global__ __launch_bounds__(512) void k(const char* src, unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const unsigned long long ad = (unsigned long long)src;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ad), hi = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)(PITCH * ROWS), 0x00020000);
    const unsigned row0 = (blockIdx.x & 3) * 64;
    unsigned q = wave, acc = 0;
    u32x4 prev[D];
#pragma unroll
    for (int d = 0; d < D; ++d) prev[d] = u32x4{0, 0, 0, 0};
    char* ring = smem + wave * (2 * D * 1024);
    for (int it = 0; it < iters; ++it) {
        if (MODE <= 1) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const unsigned o = offset_of<MODE>(q, row0, lane);        // (a template call INSIDE the builtin's argument list breaks hipcc's host pass)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + ((it & 1) * D + d) * 1024), 16, o, 0, 0, 0);
                q += nw;
            }
            if (D == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            u32x4 cur[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { cur[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offset_of<MODE>(q, row0, lane), 0, 0)); q += nw; }
#pragma unroll
            for (int d = 0; d < D; ++d) { acc ^= prev[d].x ^ prev[d].w; prev[d] = cur[d]; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < D; ++d) acc ^= prev[d].y;
    if (MODE <= 1) acc ^= reinterpret_cast<unsigned*>(smem)[tid];
    if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE, int D>
void run(int nw, int blocks, const char* src, unsigned* out) {
    const int iters = 3000;
    const int lds = MODE <= 1 ? nw * 2 * D * 1024 : 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, D>), dim3(blocks), dim3(nw * 64), lds, 0, src, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, D>), dim3(blocks), dim3(nw * 64), lds, 0, src, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us_it = ms * 1e3 / iters, bytes = (double)nw * D * 1024;
    static const char* names[] = {"LDS-DMA 8 rows x 128 B", "LDS-DMA 1 KiB contiguous", "VGPR 8 rows x 128 B", "VGPR 32 rows x 32 B", "VGPR 16 rows x 64 B", "VGPR 1 KiB contiguous"};
    printf("%-26s waves=%d x %d instr/iter, %3d blocks: %.3f us/iter -> %6.1f GB/s/CU = %5.1f B/clk @2.2GHz (%5.1f clk per 1-KiB instruction per CU)\n",
           names[MODE], nw, D, blocks, us_it, bytes / us_it / 1e3, bytes / (us_it * 2200), us_it * 2200 / (nw * D));
}
int main() {
    char* src; unsigned* out;
    hipMalloc(&src, PITCH * ROWS + 65536); hipMemset(src, 1, PITCH * ROWS + 65536); hipMalloc(&out, 1 << 16);
    for (int blocks : {128, 256}) {
        run<0, 4>(4, blocks, src, out); run<0, 4>(8, blocks, src, out);
        run<1, 4>(4, blocks, src, out); run<1, 4>(8, blocks, src, out);
        run<2, 4>(4, blocks, src, out); run<2, 4>(8, blocks, src, out); run<2, 8>(4, blocks, src, out);
        run<3, 4>(4, blocks, src, out); run<3, 4>(8, blocks, src, out); run<3, 8>(4, blocks, src, out);
        run<4, 4>(4, blocks, src, out); run<4, 4>(8, blocks, src, out); run<4, 8>(4, blocks, src, out);
        run<5, 4>(4, blocks, src, out); run<5, 8>(4, blocks, src, out);
    }
    return 0;
}
