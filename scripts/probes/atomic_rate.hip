// Probe: throughput of coalesced fp32 atomic adds (no return) on gfx950, private regions vs regions shared by S blocks,
// against plain 4-byte and 16-byte stores of the same volume.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* buf, int share, int iters) {
    // each block covers a 64 KiB tile (16384 floats); blocks (blockIdx.x / share) share a tile
    float* tile = buf + (size_t)(blockIdx.x / share) * 16384;
    for (int it = 0; it < iters; ++it)
        for (int i = threadIdx.x; i < 16384; i += 512) {
            if (MODE == 0) atomicAdd(tile + i, 1.0f);
            else if (MODE == 1) tile[i] = (float)it;
        }
    if (MODE == 2)
        for (int it = 0; it < iters; ++it)
            for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<float4*>(tile)[i] = make_float4(it, 1, 2, 3);
}
template <int MODE>
void run(const char* name, float* buf, int blocks, int share) {
    const int iters = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, buf, share, 1);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, buf, share, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 16384 * iters;
    printf("%-22s blocks=%4d share=%2d: %8.1f us  %7.1f G elem/s  (%.2f TB/s)\n", name, blocks, share, ms * 1e3, n / ms / 1e6, n * 4 / ms / 1e9);
}
int main() {
    float* buf; hipMalloc(&buf, 512u << 20); hipMemset(buf, 0, 512u << 20);
    for (int share : {1, 2, 14, 56}) run<0>("atomicAdd f32", buf, 504, share);
    run<0>("atomicAdd f32", buf, 2016, 1);
    run<1>("store 4B", buf, 504, 1);
    run<2>("store 16B", buf, 504, 1);
    run<2>("store 16B", buf, 504, 14);
    return 0;
}
