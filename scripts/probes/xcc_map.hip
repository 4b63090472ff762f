// Probe: which XCD (XCC_ID hardware register) does workgroup `blockIdx.x` of a 1-D / 2-D grid run on?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.y * gridDim.x + blockIdx.x] = (int)(xcc & 0xf);
    }
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    int h[4096];
    for (int pass = 0; pass < 2; ++pass) {
        dim3 grid = pass == 0 ? dim3(64, 1) : dim3(9, 7);
        hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, d);
        hipMemcpy(h, d, grid.x * grid.y * 4, hipMemcpyDeviceToHost);
        printf("grid (%d,%d): ", grid.x, grid.y);
        for (unsigned i = 0; i < grid.x * grid.y; ++i) printf("%d ", h[i]);
        printf("\n");
    }
    return 0;
}
