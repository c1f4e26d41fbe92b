// HBM microbenchmark 2: fixed total footprint (1.4 GB, >> 256 MiB Infinity Cache), split into K equal
// streams (R reads + W writes), optional per-stream skew of the base address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Ptrs { float *p[16]; };

template <int R, int W>
__global__ __launch_bounds__(256) void k_stream(Ptrs P, size_t n4, int reverse)
{
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (size_t)gridDim.x * 256) {
        size_t i = reverse ? n4 - 1 - j : j;
        float4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; r++) {
            float4 v = reinterpret_cast<const float4 *>(P.p[r])[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
#pragma unroll
        for (int w = 0; w < W; w++) { float4 o = acc; o.x += w; reinterpret_cast<float4 *>(P.p[R + w])[i] = o; }
        if (W == 0 && acc.x == 1234.5f) P.p[15][0] = acc.x;
    }
}

template <int R, int W>
void run(float *arena, size_t total_floats, size_t skew, int grid, const char *tag)
{
    const int K = R + W;
    size_t per = (total_floats / K) / 1024 * 1024;
    Ptrs P;
    for (int i = 0; i < K; i++) P.p[i] = arena + i * (per + skew);
    P.p[15] = arena;
    size_t n4 = per / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, P, n4, 0);
    CK(hipEventRecord(a));
    const int reps = 6;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, P, n4, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    // alternating direction (cache reuse between consecutive passes)
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, P, n4, i & 1);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms2; CK(hipEventElapsedTime(&ms2, a, b)); ms2 /= reps;
    double bytes = (double)K * n4 * 16;
    printf("%-8s R=%2d W=%2d skew=%6zu grid=%5d  fwd %.3f ms %5.0f GB/s | alternating %.3f ms %5.0f GB/s\n", tag, R, W, skew, grid, ms, bytes / ms / 1e6, ms2, bytes / ms2 / 1e6);
}

int main()
{
    size_t total = (size_t)350 * 1000 * 1000;   // floats = 1.4 GB
    float *arena; CK(hipMalloc(&arena, (total + 16 * 1024 * 1024) * 4)); CK(hipMemset(arena, 0, (total + 16 * 1024 * 1024) * 4));
    for (size_t skew : {(size_t)0, (size_t)64, (size_t)1088, (size_t)17 * 1024}) {
        run<1, 1>(arena, total, skew, 2048, "copy");
        run<2, 1>(arena, total, skew, 2048, "triad");
        run<4, 2>(arena, total, skew, 2048, "6");
        run<6, 2>(arena, total, skew, 2048, "8");
        run<9, 4>(arena, total, skew, 2048, "13");
        run<10, 4>(arena, total, skew, 2048, "14");
    }
    for (int grid : {512, 1024, 4096, 16384}) run<10, 4>(arena, total, 1088, grid, "14");
    return 0;
}
