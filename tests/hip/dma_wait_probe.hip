// dma_wait_probe.hip -- a minimal kernel with the LDS-DMA + counted-wait idiom of octane_amd/csrc/pcg_fused_q_dma.hip, for the self-test
// of tools/check_dma_wait.py (tests/test_capi_cpu.py): per trip of a loop one global_load_lds, then PROBE_LOADS register loads, then
// the marked wait `s_waitcnt vmcnt(PROBE_WAIT)` + `s_setprio 0`, then the LDS read.  The checker has to accept PROBE_WAIT <=
// PROBE_LOADS and reject PROBE_WAIT > PROBE_LOADS (the wait could pass with the DMA still in flight).  PROBE_COND makes one of the
// loads conditional: a path with one load fewer, which the checker has to find.  Compiled, disassembled, never launched.
#include <hip/hip_runtime.h>
#ifndef PROBE_LOADS
#define PROBE_LOADS 4
#endif
#ifndef PROBE_WAIT
#define PROBE_WAIT 4
#endif
#ifndef PROBE_COND
#define PROBE_COND 0
#endif
#define STR2(x) #x
#define STR(x) STR2(x)

extern "C" __global__ __launch_bounds__(64) void dma_wait_probe(const float *__restrict__ src, const float *__restrict__ p0, const float *__restrict__ p1,
                                                                const float *__restrict__ p2, const float *__restrict__ p3, const float *__restrict__ p4,
                                                                const float *__restrict__ p5, float *__restrict__ out, int trips, int flag)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * 4];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_float *)tile);
    const int lane = threadIdx.x;
    const float *const planes[6] = {p0, p1, p2, p3, p4, p5};
    float acc = 0.f;
    for (int t = 0; t < trips; t++) {
        const float *g = src + (size_t)t * 256 + lane * 4;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(base) : "memory");
        float v[PROBE_LOADS];
#pragma unroll
        for (int i = 0; i < PROBE_LOADS; i++) {
            v[i] = 0.f;
            if (!(PROBE_COND && i == 1) || flag) v[i] = planes[i][t * 64 + lane];     // PROBE_COND: a path with one load fewer
        }
        asm volatile("s_waitcnt vmcnt(" STR(PROBE_WAIT) ")\n\ts_setprio 0" ::: "memory");
        __syncthreads();
        acc += tile[(lane * 4 + t) & 255];
#pragma unroll
        for (int i = 0; i < PROBE_LOADS; i++) acc += v[i];
        __syncthreads();
    }
    out[blockIdx.x * 64 + lane] = acc;
}
