// HBM microbenchmark: R read streams + W write streams of float4 per thread, persistent grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { float *p[16]; };

template <int R, int W>
__global__ __launch_bounds__(256) void k_stream(Ptrs P, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; r++) {
            float4 v = reinterpret_cast<const float4 *>(P.p[r])[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
#pragma unroll
        for (int w = 0; w < W; w++) {
            float4 o = acc; o.x += w;
            reinterpret_cast<float4 *>(P.p[R + w])[i] = o;
        }
        if (W == 0 && acc.x == 1234.5f) P.p[15][0] = acc.x;
    }
}

template <int R, int W>
void run(Ptrs P, size_t n4, int grid, const char *tag)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, P, n4);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, P, n4);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    double bytes = (double)(R + W) * n4 * 16;
    printf("%-10s R=%2d W=%2d grid=%5d  %.3f ms  %.0f GB/s\n", tag, R, W, grid, ms, bytes / ms / 1e6);
}

int main(int argc, char **argv)
{
    size_t n = (size_t)5056 * 5000;   // one 5000^2 plane, pitched
    size_t n4 = n / 4;
    Ptrs P;
    for (int i = 0; i < 16; i++) { CK(hipMalloc(&P.p[i], n * 4)); CK(hipMemset(P.p[i], 0, n * 4)); }
    for (int grid : {1024, 2048, 4096, 8192}) {
        run<1, 1>(P, n4, grid, "copy");
        run<2, 1>(P, n4, grid, "triad");
        run<4, 0>(P, n4, grid, "read4");
        run<9, 4>(P, n4, grid, "passA-like");
        run<10, 4>(P, n4, grid, "passB-like");
        run<5, 2>(P, n4, grid, "half");
    }
    return 0;
}
