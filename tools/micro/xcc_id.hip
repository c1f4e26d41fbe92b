#include <hip/hip_runtime.h>
__global__ void k(unsigned *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    unsigned cu;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = cu; }
}
int main() {
    unsigned *d; hipMalloc(&d, 8 * 1024); hipMemset(d, 0xff, 8 * 1024);
    for (int g : {15, 200, 256}) {
        hipLaunchKernelGGL(k, dim3(g), dim3(512), 0, 0, d);
        unsigned h[2048]; hipMemcpy(h, d, 8 * g, hipMemcpyDeviceToHost);
        printf("grid %d: xcc of blocks 0..31:", g);
        for (int i = 0; i < 32 && i < g; i++) printf(" %u", h[2 * i] & 15);
        int bad = 0; for (int i = 0; i < g; i++) bad += ((h[2 * i] & 15) != (unsigned)(i % 8));
        printf("  | blocks with xcc != block %% 8: %d\n", bad);
    }
    return 0;
}
