// What the PROLOGUE of a per-iteration PCG launch costs, three ways ("issue launch k + 1's state load + 7 x 512 partial
// fold from a single wave into a 56-byte scalar block that every workgroup then reads, instead of every workgroup folding").
// A chain of dependent launches of a kernel that does nothing but the prologue and the epilogue of k_pcg_fused_q_dma:
//   mode 0  nothing (what a launch of this grid costs)
//   mode 1  EVERY workgroup folds the previous launch's 7 x nparts fp64 partials (14 loads per thread at 512 partials, wave shuffles +
//           an LDS tree) and writes its own seven partials -- the production form
//   mode 2  ONE wave (workgroup 0, wave 0) folds them and publishes {7 doubles, tag = launch number} with write-through stores; thread 0
//           of every workgroup polls the tag (bounded), then the workgroup reads the 56 bytes -- the consuming-side single fold
//   mode 3  as mode 2, but the single wave's fold is done by the LAST launch's workgroup 0 at its end (producing side), the poll is gone:
//           the next launch just reads 56 bytes -- the floor of any single-fold scheme (round 4's last-workgroup fold without its atomics)
// Every launch's result depends on the previous one's (a running checksum), so nothing can be elided.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/fold_probe tools/micro/fold_probe.hip && tools/micro/bin/fold_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kKinds = 7, kMaxParts = 2048;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

struct Pub { double v[kKinds]; unsigned long long tag; };

template <int MODE>
__global__ __launch_bounds__(256) void k_prologue(const double *pin, double *pout, Pub *pub, int nparts, int k, int *timeout_flag)
{
    __shared__ double s_red[4 * kKinds];
    __shared__ double s_tot[kKinds];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double tot[kKinds];
#pragma unroll
    for (int j = 0; j < kKinds; j++) tot[j] = 1.0;
    if (MODE == 1) {
        double v[kKinds];
#pragma unroll
        for (int j = 0; j < kKinds; j++) v[j] = 0.;
        for (int i = tid; i < nparts; i += 256) {
#pragma unroll
            for (int j = 0; j < kKinds; j++) v[j] += pin[(size_t)j * kMaxParts + i];
        }
#pragma unroll
        for (int j = 0; j < kKinds; j++) { v[j] = wave_sum(v[j]); if (lane == 0) s_red[j * 4 + wv] = v[j]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kKinds; j++) tot[j] = s_red[j * 4] + s_red[j * 4 + 1] + s_red[j * 4 + 2] + s_red[j * 4 + 3];
    } else if (MODE == 2) {
        Pub *slot = pub + (k & 1);
        if (blockIdx.x == 0 && wv == 0) {
            double v[kKinds];
#pragma unroll
            for (int j = 0; j < kKinds; j++) v[j] = 0.;
            for (int i = lane; i < nparts; i += 64) {
#pragma unroll
                for (int j = 0; j < kKinds; j++) v[j] += pin[(size_t)j * kMaxParts + i];
            }
#pragma unroll
            for (int j = 0; j < kKinds; j++) v[j] = wave_sum(v[j]);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < kKinds; j++) __hip_atomic_store(&slot->v[j], v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slot->tag, (unsigned long long)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) {
            long spins = 0;
            while (__hip_atomic_load(&slot->tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)k) {
                if (++spins > (1L << 22)) { *timeout_flag = 1; break; }          // never hang the GPU
            }
#pragma unroll
            for (int j = 0; j < kKinds; j++) s_tot[j] = __hip_atomic_load(&slot->v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kKinds; j++) tot[j] = s_tot[j];
    } else if (MODE == 3) {
        const Pub *slot = pub + ((k + 1) & 1);          // what the previous launch's epilogue published
        if (tid < kKinds) s_tot[tid] = slot->v[tid];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kKinds; j++) tot[j] = s_tot[j];
    }
    // "tile work": none.  Epilogue: the workgroup's own seven partials (depend on what was folded)
    if (MODE != 0) {
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < kKinds; j++) pout[(size_t)j * kMaxParts + blockIdx.x] = tot[j] * 1e-3 + (double)(blockIdx.x & 7);
        }
    }
    if (MODE == 3 && blockIdx.x == 0 && wv == 0) {
        // (stand-in for "the workgroup that finishes last folds": here workgroup 0 folds the PREVIOUS block again -- same loads, same
        // arithmetic, no wait -- and publishes for the next launch; a real scheme needs an arrival counter on top: this is its floor)
        double v[kKinds];
#pragma unroll
        for (int j = 0; j < kKinds; j++) v[j] = 0.;
        for (int i = lane; i < nparts; i += 64) {
#pragma unroll
            for (int j = 0; j < kKinds; j++) v[j] += pin[(size_t)j * kMaxParts + i];
        }
#pragma unroll
        for (int j = 0; j < kKinds; j++) v[j] = wave_sum(v[j]);
        if (lane == 0) {
            Pub *mine = pub + (k & 1);
#pragma unroll
            for (int j = 0; j < kKinds; j++) mine->v[j] = v[j];
        }
    }
}

template <int MODE>
static float run(int grid, int nparts, int launches, double *parts, Pub *pub, int *flag)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {                    // the first pass warms up
        (void)hipEventRecord(e0, nullptr);
        for (int k = 0; k < launches; k++) {
            const double *pin = parts + (size_t)((k + 1) & 1) * kKinds * kMaxParts;
            double *pout = parts + (size_t)(k & 1) * kKinds * kMaxParts;
            hipLaunchKernelGGL(k_prologue<MODE>, dim3(grid), dim3(256), 0, nullptr, pin, pout, pub, nparts, k + 2, flag);
        }
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / launches;
}

int main()
{
    double *parts; Pub *pub; int *flag;
    (void)hipMalloc((void **)&parts, 2 * (size_t)kKinds * kMaxParts * sizeof(double));
    (void)hipMalloc((void **)&pub, 2 * sizeof(Pub));
    (void)hipMalloc((void **)&flag, sizeof(int));
    (void)hipMemset(parts, 0, 2 * (size_t)kKinds * kMaxParts * sizeof(double));
    (void)hipMemset(pub, 0, 2 * sizeof(Pub));
    (void)hipMemset(flag, 0, sizeof(int));
    const int launches = 2000;
    printf("# us per launch of a chain of %d dependent launches, 256 threads per workgroup; nparts = grid\n", launches);
    printf("# grid   empty   every-wg-fold   single-wave-fold+poll   publish-at-end(no wait)\n");
    for (int grid : {128, 256, 512, 768, 1024}) {
        const float a = run<0>(grid, grid, launches, parts, pub, flag);
        const float b = run<1>(grid, grid, launches, parts, pub, flag);
        const float c = run<2>(grid, grid, launches, parts, pub, flag);
        const float d = run<3>(grid, grid, launches, parts, pub, flag);
        int f = 0;
        (void)hipMemcpy(&f, flag, sizeof(int), hipMemcpyDeviceToHost);
        printf("%6d  %6.2f   %6.2f (+%.2f)   %6.2f (+%.2f)%s   %6.2f (+%.2f)\n", grid, a, b, b - a, c, c - a, f ? " TIMEOUT" : "", d, d - a);
        (void)hipMemset(flag, 0, sizeof(int));
    }
    return 0;
}
