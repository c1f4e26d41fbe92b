// h2d_pitch.hip -- what the host-buffer path of octane_vof_run pays for its transfers: a 5000 x 5000 float frame
// host -> device and back, as one linear copy, as a 2-D copy into / out of a pitched plane (what plan_load_inputs does: the plan's rows
// are padded to 64 floats) and as a linear copy + a device-side repack kernel; pageable and pinned host memory.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/h2d_pitch tools/micro/h2d_pitch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void repack(const float *src, int sp, float *dst, int dp, int w, int h)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x + 3 < w) *(float4 *)&dst[(size_t)y * dp + x] = *(const float4 *)&src[(size_t)y * sp + x];
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 5000, pitch = (n + 63) / 64 * 64;
    const size_t dense = (size_t)n * n * 4, pitched = (size_t)pitch * n * 4;
    float *d_pitched, *d_dense, *h_pin, *h_page = (float *)malloc(dense);
    CK(hipMalloc(&d_pitched, pitched)); CK(hipMalloc(&d_dense, dense)); CK(hipHostMalloc(&h_pin, dense));
    memset(h_page, 1, dense); memset(h_pin, 1, dense);
    hipStream_t s; CK(hipStreamCreate(&s));
    auto time = [&](const char *what, auto fn) {
        double best = 1e9;
        for (int r = 0; r < 5; r++) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            fn();
            CK(hipStreamSynchronize(s));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (r > 0 && ms < best) best = ms;
        }
        printf("%-64s %7.3f ms  %6.1f GB/s\n", what, best, dense / best / 1e6);
        return 0;
    };
    for (int pin = 0; pin < 2; pin++) {
        float *h = pin ? h_pin : h_page;
        printf("-- %s host memory, %d x %d floats (%.0f MB)\n", pin ? "pinned" : "pageable", n, n, dense / 1e6);
        time("H2D linear", [&] { (void)hipMemcpyAsync(d_dense, h, dense, hipMemcpyHostToDevice, s); });
        time("H2D 2-D into the pitched plane (plan_load_inputs today)", [&] { (void)hipMemcpy2DAsync(d_pitched, (size_t)pitch * 4, h, (size_t)n * 4, (size_t)n * 4, n, hipMemcpyHostToDevice, s); });
        time("H2D linear + device repack into the pitched plane", [&] { (void)hipMemcpyAsync(d_dense, h, dense, hipMemcpyHostToDevice, s);
             hipLaunchKernelGGL(repack, dim3((n / 4 + 255) / 256, n), dim3(256), 0, s, d_dense, n, d_pitched, pitch, n, n); });
        time("D2H linear", [&] { (void)hipMemcpyAsync(h, d_dense, dense, hipMemcpyDeviceToHost, s); });
        time("D2H 2-D out of the pitched plane (today)", [&] { (void)hipMemcpy2DAsync(h, (size_t)n * 4, d_pitched, (size_t)pitch * 4, (size_t)n * 4, n, hipMemcpyDeviceToHost, s); });
        time("device repack + D2H linear", [&] { hipLaunchKernelGGL(repack, dim3((n / 4 + 255) / 256, n), dim3(256), 0, s, d_pitched, pitch, d_dense, n, n, n);
             (void)hipMemcpyAsync(h, d_dense, dense, hipMemcpyDeviceToHost, s); });
    }
    return 0;
}
