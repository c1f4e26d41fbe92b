// pingpong.hip -- what does one hand-off between two workgroups cost, through the XCD's L2 and through the fabric?  (round 3)
//
// The persistent mid-level solve exchanges its partial sums and edge pixels with agent-scope accesses (sc1: the XCDs' L2s are not
// coherent with each other, so the data has to go through the fabric), and what is left of a small level's iteration is mostly that
// round trip.  Workgroups on the SAME XCD could meet in its L2 (workgroup-scope accesses, sc0: past the L1, coherent in the L2).
// Two single-wave workgroups bounce a counter: A writes n to flag0, B waits for it and writes n to flag1, A waits for that, n + 1 ...
// for pairs on one XCD (blocks 0 and 8 of the grid: workgroups are dealt round-robin over the XCDs) and on two (blocks 0 and 1), with
//   agent   stores and loads at agent scope (what the solve does today),
//   wg      stores and loads at workgroup scope,
//   mixed   agent-scope stores (visible to every XCD), workgroup-scope loads.
// Every wait is bounded.    hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE> __device__ __forceinline__ void st(unsigned long long *p, unsigned long long v)
{
    if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE> __device__ __forceinline__ unsigned long long ld(const unsigned long long *p)
{
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long *flags, int partner_block, int rounds, unsigned *info)
{
    const int b = blockIdx.x;
    if (b != 0 && b != partner_block) return;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) info[b == 0 ? 0 : 1] = x & 15;
    unsigned long long *mine = flags + (b == 0 ? 0 : 64), *theirs = flags + (b == 0 ? 64 : 0);      // 512 bytes apart
    if (threadIdx.x != 0) return;
    for (int n = 1; n <= rounds; n++) {
        if (b == 0) st<MODE>(mine, (unsigned long long)n);
        long spins = 0;
        while (ld<MODE>(theirs) < (unsigned long long)n) { if (++spins > (1L << 22)) { info[2] = 1; return; } }
        if (b != 0) st<MODE>(mine, (unsigned long long)n);
    }
}

int main()
{
    unsigned long long *flags; unsigned *info;
    CK(hipMalloc((void **)&flags, 4096)); CK(hipMalloc((void **)&info, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 20000;
    const char *names[3] = {"agent stores, agent loads", "wg stores, wg loads", "agent stores, wg loads"};
    for (int partner : {8, 1, 16, 4}) {
        for (int mode = 0; mode < 3; mode++) {
            CK(hipMemset(flags, 0, 4096)); CK(hipMemset(info, 0, 64));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(64), dim3(64), 0, 0, flags, partner, rounds, info);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(64), dim3(64), 0, 0, flags, partner, rounds, info);
            else hipLaunchKernelGGL(k<2>, dim3(64), dim3(64), 0, 0, flags, partner, rounds, info);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h[3]; CK(hipMemcpy(h, info, 12, hipMemcpyDeviceToHost));
            printf("blocks 0 and %2d (XCD %u and %u), %-26s: %s%.3f us per round trip (two hand-offs)\n", partner, h[0], h[1], names[mode], h[2] ? "TIMED OUT " : "", ms * 1e3 / rounds);
        }
    }
    return 0;
}
