// What a grid-wide barrier costs on this GPU, as a function of the number of workgroups: the question behind "several PCG
// iterations of a small level in one persistent kernel" (DESIGN 9).  Two barriers are timed: cooperative groups'
// grid.sync() and a hand-rolled sense-reversing one on a device-scope atomic (what a persistent solver would use, with
// the data it exchanges made visible by the same release / acquire pair).
//   hipcc --offload-arch=gfx950 -O3 -o gridsync tools/micro/gridsync.hip && ./gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k_cg(int iters, float *out)
{
    cg::grid_group g = cg::this_grid();
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        acc += (float)i;
        g.sync();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_atomic(int iters, unsigned *counter, float *out, int *timeout_flag)
{
    float acc = 0.f;
    const unsigned n = gridDim.x;
    for (int i = 0; i < iters; i++) {
        acc += (float)i;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(i + 1) * n;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1L << 24)) { *timeout_flag = 1; break; }      // never hang the GPU
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main()
{
    float *out; unsigned *counter; int *flag;
    (void)hipMalloc((void **)&out, 4096 * sizeof(float));
    (void)hipMalloc((void **)&counter, sizeof(unsigned));
    (void)hipMalloc((void **)&flag, sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int grid : {8, 40, 64, 128, 256, 400, 512}) {
        float ms_cg = -1.f, ms_at = -1.f;
        {
            int it = iters;
            void *args[] = {&it, &out};
            hipLaunchCooperativeKernel((const void *)k_cg, dim3(grid), dim3(256), args, 0, nullptr);      // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((const void *)k_cg, dim3(grid), dim3(256), args, 0, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            if (e == hipSuccess && hipGetLastError() == hipSuccess) hipEventElapsedTime(&ms_cg, e0, e1);
        }
        {
            hipMemset(counter, 0, sizeof(unsigned)); hipMemset(flag, 0, sizeof(int));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(256), 0, 0, iters, counter, out, flag);
            hipEventRecord(e1); hipEventSynchronize(e1);
            int h = 0; hipMemcpy(&h, flag, sizeof h, hipMemcpyDeviceToHost);
            if (hipGetLastError() == hipSuccess && !h) hipEventElapsedTime(&ms_at, e0, e1);
        }
        printf("grid %4d: grid.sync %6.2f us per barrier, atomic barrier %6.2f us\n", grid, ms_cg * 1e3f / iters, ms_at * 1e3f / iters);
    }
    return 0;
}
