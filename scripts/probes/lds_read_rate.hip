// Probe: LDS read bandwidth of one CU by instruction — ds_read_b128, ds_read_b64, ds_read_b64_tr_b16 (the hardware-transpose read the
// weight-gradient kernels feed their MFMAs with) — 8 waves, conflict-free address patterns (the kernels' own), 16 reads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o lds_read_rate lds_read_rate.hip && ./lds_read_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, unsigned* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
    unsigned ad;
    if (MODE == 0) ad = base + lane * 16;                       // b128: 1 KiB contiguous per instruction
    else if (MODE == 1) ad = base + lane * 8;                   // b64: 512 B contiguous
    else {                                                      // tr_b16 b64: wgrad3x3's halo fragment pattern (64-byte rows)
        const int i16 = lane & 15, q4 = i16 >> 2, h = lane >> 5;
        const int mcol = ((lane >> 4) & 1) * 16 + (i16 & 3) * 4, k_lo = 8 * h + q4;
        ad = base + k_lo * 64 + (mcol >> 3) * 16 + (mcol & 7) * 2;
    }
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(ad), "n"((r & 3) * 1024) : "memory");
            else if (MODE == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*reinterpret_cast<uint2*>(&v[r])) : "v"(ad), "n"((r & 7) * 512) : "memory");
            else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*reinterpret_cast<uint2*>(&v[r])) : "v"(ad), "n"((r & 3) * 1024) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc.x ^= v[r].x; acc.y ^= v[r].y; }
    }
    const unsigned long long t1 = clock64();
    if (acc.x == 0x12345u && acc.y == 0x54321u) sink[tid] = acc.x;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 4096);
    const int iters = 2000;
    const char* names[3] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16"};
    const int bytes[3] = {1024, 512, 512};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 64 * 1024, 0, out, sink, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 64 * 1024, 0, out, sink, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 64 * 1024, 0, out, sink, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[2048]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < 2048; ++i) mx = h[i] > mx ? h[i] : mx;
        const double clk = (double)mx;               // clock64() counts shader clocks here (an MFMA 32x32x16 measures 32.3 of them: mfma_issue.hip)
        printf("%-20s %8.1f clk per wave-instruction and CU (8 waves, 16 in flight each) = %6.1f B/clk/CU\n",
               names[mode], clk / (iters * 16.0 * 8), bytes[mode] * 8.0 * iters * 16 / clk);
    }
    return 0;
}
