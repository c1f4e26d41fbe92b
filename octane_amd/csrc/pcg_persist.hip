// pcg_persist.hip -- a whole PCG solve of a mid-size pyramid level in ONE launch, the level resident on chip.  gfx950.
//
// Behavioural spec: ref src/oct_variational_optical_flow.cu:1105-1195 (the PCG loop of one linearisation, then u += dx,
// v += dy), in the one-reduction-per-iteration form of k_pcg_fused (pcg_kernels.hip): same operator, same recurrences, same
// stop test, same iteration count.
//
// Why: between the levels one workgroup can hold (k_pcg_solve_small, <= 6144 pixels) and the levels that stream from HBM
// (>= 2 Mpixel) lie the levels whose PCG iteration is pure latency: 10-15 us per launch at 156^2 .. 625^2 and 33 us at
// 1250^2 for a working set that fits the chip's registers and LDS several times over (256 CUs x (512 KB + 160 KB)).  Here
// the level is cut into sub-domains of 64 columns x up to 128 rows, one 512-thread workgroup (one CU) each; r, p, q and the
// operator of a pixel stay in its thread's registers and x in LDS for the whole solve.  Per iteration a workgroup
//   * folds the G x 7 partial sums of the previous iteration (every workgroup folds all of them in the same order, so all
//     take the same alpha, beta and stop decision),
//   * recomputes p_k on its one-pixel ring from r, q, p of the neighbouring sub-domains' edge pixels (published by their
//     owners at the end of the previous iteration: same inputs, same operations, same bits as the owner's own p_k),
//   * updates x, r, p of its own pixels, forms q = A p from an LDS tile and the seven partial sums,
//   * publishes its edge pixels and its partial sums and meets the other workgroups at a grid barrier:
// ONE barrier per iteration, nothing else leaves the chip.
//
// Hand-off between workgroups inside the launch (cdna_hip_programming.md, Guideline 16, form R1): the payload (edges, partial
// sums) is stored write-through (agent-scope relaxed atomic stores = sc1), every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at its own barrier, ONE lane adds 1 to the grid counter; a consumer polls that
// ONE word (relaxed, with s_sleep, bounded), then ONE agent-scope acquire + vmcnt(0) + workgroup barrier, and reads the
// payload with agent-scope (sc1) loads.  The counter, the abort word and nothing else are zeroed by a memset node ahead of
// every launch.  All G workgroups have to be resident at once: G <= number of CUs, and launches of this kernel on one device
// are serialised among themselves by an event chain on the host side (vof_plan.hip); a barrier that does not complete
// within 0.25 s (a co-tenant process holding CUs with the same kind of kernel) sets the abort word, every workgroup leaves,
// and the host reports an error instead of hanging the GPU.
//
// The same kernel runs a sub-range of iterations per launch with the complete state stored to / loaded from the level's
// planes (full_state, "stepped" form: the kernel boundary then provides the visibility): that is the per-launch form this
// kernel is checked against bit for bit (tests/test_gpu_persist.py) -- any stale read through the in-launch hand-off would
// show up as a difference.
#include "vof_kernels.hpp"
#include "device_util.hpp"

namespace octane {

constexpr int kMidT = 512;              // threads per workgroup: 8 waves, two per SIMD, up to 256 VGPRs each
constexpr int kMidW = 64;               // columns of a sub-domain: one wavefront per row
constexpr int kMidRG = kMidT / kMidW;   // rows one slot of all threads covers
constexpr int kMidLP = kMidW + 2;       // LDS row of the p tile: west ring pixel, 64 columns, east ring pixel
constexpr int kMidEdge = 128;           // longest edge of a sub-domain (rows: 16 slots x 8)
constexpr unsigned long long kMidTimeoutTicks = 25000000ull;   // 0.25 s of the 100 MHz wall clock

// plane base + 32-bit byte offset: the scalar-base addressing form (one VGPR of offset for every plane instead of a 64-bit
// address pair per plane and pixel; a level's planes are far smaller than 4 GiB)
__device__ __forceinline__ const float *at(const float *base, unsigned byte_off) { return (const float *)((const char *)base + byte_off); }
__device__ __forceinline__ float *at(float *base, unsigned byte_off) { return (float *)((char *)base + byte_off); }

__device__ __forceinline__ void st_agent(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float *p)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Arrive at grid barrier number `phase` (1-based within the launch) and wait for everybody.  Every wave has stored its
// payload before the call.  Returns false (uniformly) when the launch has been aborted.
__device__ __forceinline__ bool mid_grid_barrier(const MidArgs &A, unsigned long long target, int *s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(A.ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(A.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63) == 0) {
                if (__hip_atomic_load(A.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                if (wall_clock64() - t0 > kMidTimeoutTicks) {
                    __hip_atomic_store(A.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

template <int P, bool UNITW>
__global__ __launch_bounds__(kMidT) void k_pcg_solve_mid(LevelPtrs L, MidArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    constexpr int ROWS = P * kMidRG;                       // rows a sub-domain can hold
    float *s_pu = s_mem, *s_pv = s_pu + (ROWS + 2) * kMidLP;
    float *s_xu = s_pv + (ROWS + 2) * kMidLP, *s_xv = s_xu + ROWS * kMidW;
    float *s_edge = s_xv + ROWS * kMidW;                   // r_u, r_v, q_u, q_v of the first / last row [0..7] and of the west / east column [8..15]: 16 x 128
    double *s_red = reinterpret_cast<double *>(s_edge + 16 * kMidEdge);     // 8 * kPartKinds doubles (offset is a multiple of 16 bytes)
    int *s_flag = reinterpret_cast<int *>(s_red + (kMidT / 64) * kPartKinds);

    const int tid = threadIdx.x, c_ = tid & (kMidW - 1), rg_ = tid >> 6;
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int wg = blockIdx.x, bx = wg % A.gx, by = wg / A.gx;
    const int x0 = bx * kMidW, y0 = by * A.bh;
    const int sw = min(kMidW, w - x0), sh = min(A.bh, h - y0);

    // The solve's scalars between the launches of the stepped form, double-buffered by the parity of the first iteration (a
    // one-iteration launch has no barrier, so workgroup 0 may write the new state before another workgroup has read the old)
    PcgState st = L.st[A.k0 & 1];
    if (A.k0 == 0) { st.rz = 0.f; st.stopped = 0; st.iters = 0; }
    if (st.stopped) {                                      // stepped form: the loop ended in an earlier launch (uniform)
        if (blockIdx.x == 0 && threadIdx.x == 0) L.st[A.k1 & 1] = st;
        return;
    }

    // ---- the sub-domain's state: operator, r (p, q, x in the stepped form) -> registers / LDS
    float ru[P], rv[P], qu[P], qv[P], a1[P], a2[P], a4[P];      // p of the own pixels lives in the LDS tile only
    float wS[UNITW ? 1 : P], wW[UNITW ? 1 : P], wE[UNITW ? 1 : P], wN[UNITW ? 1 : P];   // merged neighbour weights (ref .cu:929-1001)
    for (int i = tid; i < (ROWS + 2) * kMidLP; i += kMidT) { s_pu[i] = 0.f; s_pv[i] = 0.f; }
    __syncthreads();
    const int par0 = (A.k0 + 1) & 1;                       // parity of iteration k0 - 1: where the stepped form left its state
    {
    const int c = c_, rg = rg_, x = x0 + c;
    const bool colok = c < sw;
#pragma unroll
    for (int s = 0; s < P; s++) {
        const int ly = s * kMidRG + rg, y = y0 + ly;
        const bool ok = colok && ly < sh;
        ru[s] = rv[s] = qu[s] = qv[s] = 0.f;
        float pu0 = 0.f, pv0 = 0.f;
        a1[s] = a4[s] = 1.f; a2[s] = 0.f;
        if (!UNITW) { wS[s] = wW[s] = wE[s] = wN[s] = 0.f; }
        float xu0 = 0.f, xv0 = 0.f;
        if (ok) {
            const unsigned o = (unsigned)(y * pitch + x) * 4u;
            a1[s] = *at(L.a1, o); a2[s] = *at(L.a2, o); a4[s] = *at(L.a4, o);
            if (!UNITW) {
                const float wxc = *at(L.wx, o), wyc = *at(L.wy, o);
                const float wys = (y > 0) ? *at(L.wy, o - 4u * (unsigned)pitch) : 0.f, wxw = (x > 0) ? *at(L.wx, o - 4u) : 0.f;
                wS[s] = (y == h - 1) ? wys + wyc : wys;
                wW[s] = (x == w - 1) ? wxw + wxc : wxw;
                wE[s] = (x == 0) ? wxc + wxc : wxc;
                wN[s] = (y == 0) ? wyc + wyc : wyc;
            }
            if (A.k0 == 0) {
                ru[s] = *at(L.rb_u[0], o); rv[s] = *at(L.rb_v[0], o);   // r_0 = the right-hand side the assembly wrote
            } else {
                ru[s] = *at(L.rb_u[par0], o); rv[s] = *at(L.rb_v[par0], o);
                pu0 = *at(L.pf_u[par0], o); pv0 = *at(L.pf_v[par0], o);
                qu[s] = *at(L.qb_u[par0], o); qv[s] = *at(L.qb_v[par0], o);
                xu0 = *at(L.xu, o); xv0 = *at(L.xv, o);
            }
        }
        s_xu[ly * kMidW + c] = xu0; s_xv[ly * kMidW + c] = xv0;
        if (ok) { s_pu[(ly + 1) * kMidLP + c + 1] = pu0; s_pv[(ly + 1) * kMidLP + c + 1] = pv0; }     // p_{k0-1} (zero at k0 = 0)
        __builtin_amdgcn_sched_barrier(0);                 // slot after slot: all slots' loads in flight at once would not fit the registers
    }
    }
    // ---- this thread's ring pixel (threads 0 .. 127 + 2 ROWS): where it lives, whose edge it is, its preconditioner entries
    int r_lds = -1, r_idx = 0, r_nb = 0, r_side = 0;
    float r_iu = 0.f, r_iv = 0.f;
    unsigned r_off = 0;
    {
        int rx = -1, ry = -1;
        if (tid < kMidW) { rx = x0 + tid; ry = y0 - 1; r_idx = tid; r_nb = wg - A.gx; r_side = 1; if (tid < sw && ry >= 0) r_lds = tid + 1; }
        else if (tid < 2 * kMidW) { const int j = tid - kMidW; rx = x0 + j; ry = y0 + sh; r_idx = j; r_nb = wg + A.gx; r_side = 0;
                                    if (j < sw && ry < h) r_lds = (sh + 1) * kMidLP + j + 1; }
        else if (tid < 2 * kMidW + ROWS) { const int j = tid - 2 * kMidW; rx = x0 - 1; ry = y0 + j; r_idx = j; r_nb = wg - 1; r_side = 3;
                                           if (j < sh && rx >= 0) r_lds = (j + 1) * kMidLP; }
        else if (tid < 2 * kMidW + 2 * ROWS) { const int j = tid - 2 * kMidW - ROWS; rx = x0 + sw; ry = y0 + j; r_idx = j; r_nb = wg + 1; r_side = 2;
                                               if (j < sh && rx < w) r_lds = (j + 1) * kMidLP + sw + 1; }
        if (r_lds >= 0) {
            r_off = (unsigned)(ry * pitch + rx) * 4u;
            r_iu = 1.0f / *at(L.a1, r_off); r_iv = 1.0f / *at(L.a4, r_off);
        }
    }
    const unsigned e_nb_off = (unsigned)(r_lds >= 0 ? r_nb : wg) * (2 * 4 * 6 * kMidEdge) * 4u;   // byte offset of the neighbour's block

    float rz_prev = st.rz;                                 // (r.z) of iteration k - 1 as that iteration formed it
    float alpha = 0.f;
    int iters = 0, k = A.k0;
    unsigned long long phase = 0;
    bool stopped = false, aborted = false;
    __syncthreads();

    for (;; k++) {
        // Everything a slot derives from its position (predicates, LDS addresses, the reciprocals of the diagonal) is invariant
        // over the iterations, and the compiler would hoist all of it out of this loop into registers it does not have (P = 16:
        // 13 arrays of 16 are live already).  The empty asm statements make the position and the diagonal opaque once per
        // iteration, so those values are formed again where they are used -- a few integer operations and two divisions per pixel.
        int c = c_, rg = rg_;
        asm volatile("" : "+v"(c), "+v"(rg));
        const int x = x0 + c;
        const bool colok = c < sw;
#pragma unroll
        for (int s = 0; s < P; s++) asm volatile("" : "+v"(a1[s]), "+v"(a4[s]));
        // ---- the scalars of iteration k from the sums of iteration k - 1 (ref .cu:1131-1178; recurrences: pcg_kernels.hip)
        float nalpha = 0.f, beta = 0.f, rz_new, rr;
        const bool first = (k == 0);
        if (first) {
            double t[2];
            fold_band_partials_multi<2, kMidT>(L.band_parts, kPartBlock + kPartRz, kMaxParts, A.nparts_asm, 1, s_red, t);
            rz_new = (float)t[0]; rr = (float)t[1];
        } else {
            double v[kPartKinds], t[kPartKinds];
            const double *src = A.parts + (size_t)((k + 1) & 1) * kPartKinds * kMidMaxG;
#pragma unroll
            for (int j = 0; j < kPartKinds; j++) v[j] = (tid < A.G) ? ld_agent(src + j * kMidMaxG + tid) : 0.;
            block_sum_multi<kPartKinds, kMidT>(v, s_red, t);
            const double rzd = t[0], rrd = t[1], pq = t[2], qz = t[3], qmq = t[4], rq = t[5], qq = t[6];
            alpha = rz_prev / (float)pq;                   // ref .cu:1169
            nalpha = (float)(-1. * (double)alpha);         // ref .cu:1174
            const double a = (double)alpha;
            rz_new = (float)(rzd - 2. * a * qz + a * a * qmq);
            rr = (float)(rrd - 2. * a * rq + a * a * qq);
            beta = rz_new / rz_prev;
        }
        const bool active = (k < A.kcap) && (rr > A.tol);  // ref .cu:1131
        if (!active) {                                      // the loop is over: x still owes alpha_{k-1} p_{k-1}
            if (!first) {
#pragma unroll
                for (int s = 0; s < P; s++) {
                    const int ly = s * kMidRG + rg;
                    if (colok && ly < sh) {
                        const int li = (ly + 1) * kMidLP + c + 1;
                        s_xu[ly * kMidW + c] = alpha * s_pu[li] + s_xu[ly * kMidW + c];  // jVecPVec(p0,x0,x0,alphak), ref .cu:1172
                        s_xv[ly * kMidW + c] = alpha * s_pv[li] + s_xv[ly * kMidW + c];
                    }
                }
            }
            stopped = true;
            break;
        }
        if (k >= A.k1) break;                               // stepped form: this launch's share is done
        iters++;
        // ---- p_k on the ring, from the neighbouring sub-domains' edge pixels of iteration k - 1
        if (r_lds >= 0) {
            float pku, pkv;
            if (first) {
                pku = r_iu * *at(L.rb_u[0], r_off); pkv = r_iv * *at(L.rb_v[0], r_off);
            } else {
                const unsigned eo = e_nb_off + (unsigned)(((((k + 1) & 1) * 4 + r_side) * 6) * kMidEdge + r_idx) * 4u;
                float r0 = ld_agent(at(A.edges, eo)), r1 = ld_agent(at(A.edges, eo + 4u * kMidEdge));
                const float q0 = ld_agent(at(A.edges, eo + 8u * kMidEdge)), q1 = ld_agent(at(A.edges, eo + 12u * kMidEdge));
                const float p0 = ld_agent(at(A.edges, eo + 16u * kMidEdge)), p1 = ld_agent(at(A.edges, eo + 20u * kMidEdge));
                r0 = nalpha * q0 + r0; r1 = nalpha * q1 + r1;
                const float zu = r_iu * r0, zv = r_iv * r1;
                pku = beta * p0 + zu; pkv = beta * p1 + zv;
            }
            s_pu[r_lds] = pku; s_pv[r_lds] = pkv;
        }
        // ---- own pixels: x += alpha p, r -= alpha q, p = M^-1 r + beta p; direct sums of r.z and r.r.  Threads without a pixel
        // in a slot hold zeros there (and a unit diagonal) and compute along: only stores are predicated.
        double acc[kPartKinds];
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) acc[j] = 0.;
#pragma unroll
        for (int s = 0; s < P; s++) {
            const int ly = s * kMidRG + rg;
            const bool ok = colok && ly < sh;
            const int li = (ly + 1) * kMidLP + c + 1;
            float pou = 0.f, pov = 0.f;                                                      // p_{k-1} of the own pixel
            if (!first) {
                if (ok) {
                    pou = s_pu[li]; pov = s_pv[li];
                    s_xu[ly * kMidW + c] = alpha * pou + s_xu[ly * kMidW + c];            // ref .cu:1172
                    s_xv[ly * kMidW + c] = alpha * pov + s_xv[ly * kMidW + c];
                }
                ru[s] = nalpha * qu[s] + ru[s];                                              // ref .cu:1174
                rv[s] = nalpha * qv[s] + rv[s];
            }
            const float iu = 1.0f / a1[s], iv = 1.0f / a4[s];
            const float zu = iu * ru[s], zv = iv * rv[s];
            const float pnu = first ? zu : beta * pou + zu;
            const float pnv = first ? zv : beta * pov + zv;
            if (ok) { s_pu[li] = pnu; s_pv[li] = pnv; }
            float d = 0.f; d += ru[s] * zu; d += rv[s] * zv; acc[0] += (double)d;
            d = 0.f; d += ru[s] * ru[s]; d += rv[s] * rv[s]; acc[1] += (double)d;
            __builtin_amdgcn_sched_barrier(0);             // one slot after the other
        }
        __syncthreads();
        // ---- q = A p and the sums that carry q
        const int par = k & 1;
#pragma unroll
        for (int s = 0; s < P; s++) {
            const int ly = s * kMidRG + rg, y = y0 + ly;
            const bool ok = colok && ly < sh;
            const int li = (ly + 1) * kMidLP + c + 1;
            float sumu = 0.f, sumv = 0.f;
            if (y > 0) { const float ws = UNITW ? ((y == h - 1) ? -2.f : -1.f) : wS[s]; sumu += ws * s_pu[li - kMidLP]; sumv += ws * s_pv[li - kMidLP]; }
            if (x > 0) { const float ww = UNITW ? ((x == w - 1) ? -2.f : -1.f) : wW[s]; sumu += ww * s_pu[li - 1]; sumv += ww * s_pv[li - 1]; }
            const float pcu = s_pu[li], pcv = s_pv[li];         // the pixel's own p_k (a ring or unused cell where it has none: masked below)
            sumu += a1[s] * pcu; sumv += a2[s] * pcu;
            sumu += a2[s] * pcv; sumv += a4[s] * pcv;
            if (x < w - 1) { const float we = UNITW ? ((x == 0) ? -2.f : -1.f) : wE[s]; sumu += we * s_pu[li + 1]; sumv += we * s_pv[li + 1]; }
            if (y < h - 1) { const float wn = UNITW ? ((y == 0) ? -2.f : -1.f) : wN[s]; sumu += wn * s_pu[li + kMidLP]; sumv += wn * s_pv[li + kMidLP]; }
            sumu = ok ? sumu : 0.f; sumv = ok ? sumv : 0.f;     // no pixel here: the neighbours in LDS are somebody else's
            qu[s] = sumu; qv[s] = sumv;
            const float iu = 1.0f / a1[s], iv = 1.0f / a4[s];
            const float zu = iu * ru[s], zv = iv * rv[s];
            float d = 0.f; d += pcu * sumu; d += pcv * sumv; acc[2] += (double)d;
            d = 0.f; d += sumu * zu; d += sumv * zv; acc[3] += (double)d;
            d = 0.f; d += sumu * (iu * sumu); d += sumv * (iv * sumv); acc[4] += (double)d;
            d = 0.f; d += ru[s] * sumu; d += rv[s] * sumv; acc[5] += (double)d;
            d = 0.f; d += sumu * sumu; d += sumv * sumv; acc[6] += (double)d;
            // r and q of the edge pixels go to LDS first (p is there already); the exchange buffer is written below, coalesced
            if (ok) {
                if (ly == 0) { s_edge[0 * kMidEdge + c] = ru[s]; s_edge[2 * kMidEdge + c] = rv[s]; s_edge[4 * kMidEdge + c] = sumu; s_edge[6 * kMidEdge + c] = sumv; }
                if (ly == sh - 1) { s_edge[1 * kMidEdge + c] = ru[s]; s_edge[3 * kMidEdge + c] = rv[s]; s_edge[5 * kMidEdge + c] = sumu; s_edge[7 * kMidEdge + c] = sumv; }
                if (c == 0) { s_edge[(8 + 0) * kMidEdge + ly] = ru[s]; s_edge[(8 + 2) * kMidEdge + ly] = rv[s]; s_edge[(8 + 4) * kMidEdge + ly] = sumu; s_edge[(8 + 6) * kMidEdge + ly] = sumv; }
                if (c == sw - 1) { s_edge[(8 + 1) * kMidEdge + ly] = ru[s]; s_edge[(8 + 3) * kMidEdge + ly] = rv[s]; s_edge[(8 + 5) * kMidEdge + ly] = sumu; s_edge[(8 + 7) * kMidEdge + ly] = sumv; }
                if (A.full_state) {                            // stepped form: the whole state goes back to the planes
                    const unsigned o = (unsigned)(y * pitch + x) * 4u;
                    *at(L.rb_u[par], o) = ru[s]; *at(L.rb_v[par], o) = rv[s]; *at(L.pf_u[par], o) = pcu; *at(L.pf_v[par], o) = pcv;
                    *at(L.qb_u[par], o) = sumu; *at(L.qb_v[par], o) = sumv; *at(L.xu, o) = s_xu[ly * kMidW + c]; *at(L.xv, o) = s_xv[ly * kMidW + c];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        double tot[kPartKinds];
        block_sum_multi<kPartKinds, kMidT>(acc, s_red, tot);   // its barriers also complete s_edge
        // ---- publish: side 0 / 1 = first / last row, 2 / 3 = west / east column; arrays r_u r_v q_u q_v p_u p_v; write-through
        {
            const int side = tid >> 7, i = tid & (kMidEdge - 1);      // 128 threads per side
            const int len = (side < 2) ? sw : sh;
            if (i < len) {
                const unsigned eo = (unsigned)(((wg * 2 + par) * 4 + side) * 6 * kMidEdge + i) * 4u;
                const int eb = (side < 2) ? side : 8 + (side - 2);
#pragma unroll
                for (int a = 0; a < 4; a++) st_agent(at(A.edges, eo + (unsigned)a * 4u * kMidEdge), s_edge[(eb + 2 * a) * kMidEdge + i]);
                const int li = (side == 0) ? kMidLP + i + 1 : (side == 1) ? sh * kMidLP + i + 1 : (side == 2) ? (i + 1) * kMidLP + 1 : (i + 1) * kMidLP + sw;
                st_agent(at(A.edges, eo + 16u * kMidEdge), s_pu[li]); st_agent(at(A.edges, eo + 20u * kMidEdge), s_pv[li]);
            }
        }
        if (tid == 0) {
            double *dst = A.parts + (size_t)par * kPartKinds * kMidMaxG + wg;
#pragma unroll
            for (int j = 0; j < kPartKinds; j++) st_agent(dst + j * kMidMaxG, tot[j]);
        }
        rz_prev = rz_new;
        const bool last_of_launch = (k + 1 >= A.k1) && (A.k1 < A.kcap);   // stepped form: the next launch folds these sums
        if (last_of_launch) { k++; break; }
        phase++;
        if (!mid_grid_barrier(A, phase * (unsigned long long)A.G, s_flag)) { aborted = true; break; }
    }
    if (aborted) return;
    __syncthreads();
    // ---- the end of the solve: u += dx, v += dy (ref .cu:1185-1195); the stepped form only carries its scalars on
    if (stopped) {
        if (k > 0) {
            const int c = c_, rg = rg_, x = x0 + c;
            const bool colok = c < sw;
#pragma unroll
            for (int s = 0; s < P; s++) {
                const int ly = s * kMidRG + rg, y = y0 + ly;
                if (colok && ly < sh) {
                    const unsigned o = (unsigned)(y * pitch + x) * 4u;
                    const float dx = s_xu[ly * kMidW + c], dy = s_xv[ly * kMidW + c];
                    *at(L.u, o) = *at(L.u, o) + dx; *at(L.v, o) = *at(L.v, o) + dy;
                    *at(L.xu, o) = dx; *at(L.xv, o) = dy;        // kept for the debug tap
                }
            }
        }
    }
    if (wg == 0 && tid == 0) {
        PcgState n; n.rz = rz_prev; n.stopped = stopped ? 1 : 0; n.iters = st.iters + iters; n.pad = 0;
        L.st[A.k1 & 1] = n;
        *L.iter_total += iters;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
static size_t mid_lds_bytes(int P)
{
    const int rows = P * kMidRG;
    return (size_t)(2 * (rows + 2) * kMidLP + 2 * rows * kMidW + 16 * kMidEdge) * sizeof(float) + (size_t)(kMidT / 64) * kPartKinds * sizeof(double) + 16;
}

// Sub-domain grid of a w x h level on a device with `ncu` CUs: 64-column strips, as many rows of sub-domains as keep every
// workgroup on a CU of its own, P (slots of 8 rows) from {4, 8, 12, 16}.  0 = the level does not fit.
int pcg_mid_config(int w, int h, int ncu, int force_p, MidGeom *g)
{
    if (ncu > kMidMaxG) ncu = kMidMaxG;
    const int gx = (w + kMidW - 1) / kMidW;
    if (gx > ncu || (long)w * h <= 0) return 0;
    static const int kP[4] = {4, 8, 12, 16};
    for (int i = 0; i < 4; i++) {
        const int P = kP[i];
        if (force_p && P != force_p) continue;
        const int rows = P * kMidRG;
        int gy = (h + rows - 1) / rows;
        if (gx * gy > ncu) continue;
        // balance: equal shares of rows, in whole slots
        int bh = (h + gy - 1) / gy;
        bh = (bh + kMidRG - 1) / kMidRG * kMidRG;
        if (bh > rows) bh = rows;
        gy = (h + bh - 1) / bh;
        g->gx = gx; g->gy = gy; g->bh = bh; g->P = P; g->G = gx * gy;
        return 1;
    }
    return 0;
}

void pcg_mid_configure()
{
#define MID_ATTR(P, U) (void)hipFuncSetAttribute((const void *)k_pcg_solve_mid<P, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds_bytes(P))
    MID_ATTR(4, false); MID_ATTR(4, true); MID_ATTR(8, false); MID_ATTR(8, true);
    MID_ATTR(12, false); MID_ATTR(12, true); MID_ATTR(16, false); MID_ATTR(16, true);
#undef MID_ATTR
}

size_t pcg_mid_workspace_bytes()
{
    // [counter + abort word: 16 bytes, zeroed before every launch] [partials 2 x 7 x G doubles] [edges G x 2 x 4 x 6 x 128 floats]
    return 256 + (size_t)2 * kPartKinds * kMidMaxG * sizeof(double) + (size_t)kMidMaxG * 2 * 4 * 6 * kMidEdge * sizeof(float);
}

// Iterations [k0, k1) of one solve; k0 = 0 and k1 = cgiters is the whole solve in one launch (plus the flow update).
hipError_t launch_pcg_solve_mid(hipStream_t s, const LevelPtrs &L, const MidGeom &g, void *workspace, int k0, int k1, int kcap,
                                int nparts_asm, float tol)
{
    MidArgs A;
    A.gx = g.gx; A.gy = g.gy; A.bh = g.bh; A.G = g.G;
    A.k0 = k0; A.k1 = k1; A.kcap = kcap; A.nparts_asm = nparts_asm;
    A.full_state = (k0 != 0 || k1 != kcap) ? 1 : 0;
    A.tol = tol;
    char *ws = static_cast<char *>(workspace);
    A.ctr = reinterpret_cast<unsigned long long *>(ws);
    A.abort_word = reinterpret_cast<unsigned int *>(ws + 8);
    A.parts = reinterpret_cast<double *>(ws + 256);
    A.edges = reinterpret_cast<float *>(ws + 256 + (size_t)2 * kPartKinds * kMidMaxG * sizeof(double));
    hipError_t e = hipMemsetAsync(ws, 0, 8, s);            // the counter only: a raised abort word stays up until the host has read it
    if (e != hipSuccess) return e;
    const size_t lds = mid_lds_bytes(g.P);
#define MID_LAUNCH(P) \
    do { if (L.unit_w) hipLaunchKernelGGL((k_pcg_solve_mid<P, true>), dim3(g.G), dim3(kMidT), lds, s, L, A); \
         else hipLaunchKernelGGL((k_pcg_solve_mid<P, false>), dim3(g.G), dim3(kMidT), lds, s, L, A); } while (0)
    switch (g.P) {
    case 4: MID_LAUNCH(4); break;
    case 8: MID_LAUNCH(8); break;
    case 12: MID_LAUNCH(12); break;
    default: MID_LAUNCH(16); break;
    }
#undef MID_LAUNCH
    return hipGetLastError();
}

}  // namespace octane
