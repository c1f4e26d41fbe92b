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
// Hand-off between workgroups inside the launch (cdna_hip_programming.md, Guideline 16, form R2: "the data IS the flag"):
// everything that crosses workgroups -- six floats per edge pixel, seven doubles per workgroup -- travels as 8-byte granules
// {tag, 32 bits of payload}, each written by ONE aligned agent-scope (write-through) store and read by agent-scope loads that
// are repeated until the tag is the one of the iteration waited for.  No flag, no counter, no fence, no drain, no grid
// barrier: the all-to-all exchange of the partial sums is what keeps the workgroups within one iteration of each other (a
// workgroup can only start iteration k + 1 when every workgroup has published the sums of iteration k, i.e. has finished
// reading what iteration k - 1 left in the buffers that iteration k + 1 overwrites; buffers alternate by iteration parity).
// tag = seq * (cgiters + 2) + k + 1 with seq a per-workspace solve counter, so a granule of an earlier solve is never taken
// for a fresh one.  All G workgroups have to be resident at once: G <= number of CUs, and launches of this kernel on one
// device are serialised among themselves by an event chain on the host side (vof_plan.hip); a wait that does not complete
// within 0.25 s (a co-tenant process holding CUs with the same kind of kernel) raises the abort word, every workgroup leaves,
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

// One granule: tag in the upper, 32 payload bits in the lower half; ONE aligned 8-byte write-through store.
__device__ __forceinline__ void st_granule(unsigned long long *g, unsigned tag, unsigned bits)
{
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long *g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long *gat(unsigned long long *base, unsigned byte_off) { return (unsigned long long *)((char *)base + byte_off); }

// Bounded wait of a polling lane: false once the launch is to be abandoned (somebody's wait timed out, ours included).
__device__ __forceinline__ bool mid_keep_waiting(const MidArgs &A, unsigned &spins, unsigned long long &t0)
{
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 127u) != 0u) return true;
    if (__hip_atomic_load(A.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    const unsigned long long now = wall_clock64();
    if (t0 == 0ull) { t0 = now; return true; }
    if (now - t0 > kMidTimeoutTicks) {
        __hip_atomic_store(A.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}

// 1 / x, correctly rounded for every normal x whose reciprocal is normal (checked against the division on all of them by
// octane_selftest_rcp / tests/test_gpu_persist.py): the hardware estimate (1 ulp) and one Newton step in fused arithmetic.
// Three instructions instead of the eleven of an IEEE division; the diagonal of the operator is >= 1, far inside that range.
__device__ __forceinline__ float rcp_exact(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__global__ void k_selftest_rcp(unsigned long long *out)     // out[0] = patterns compared, out[1] = mismatches, out[2] = first mismatch
{
    unsigned long long n = 0, bad = 0, firstbad = 0;
    // positive normal floats from 2^-125 up to 2^125 (the reciprocal stays normal)
    const unsigned lo = 0x01000000u, hi = 0x7E000000u;
    for (unsigned long long b = lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < hi; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const float want = 1.0f / x, got = rcp_exact(x);
        n++;
        if (__float_as_uint(want) != __float_as_uint(got)) { if (!bad) firstbad = b; bad++; }
    }
    atomicAdd(&out[0], n); atomicAdd(&out[1], bad);
    if (bad) atomicMax(&out[2], firstbad);
}

int pcg_selftest_rcp(hipStream_t s, unsigned long long *host3)
{
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 3 * sizeof(unsigned long long)) != hipSuccess) return -1;
    int rc = -1;
    if (hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), s) == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(256), 0, s, d);
        if (hipMemcpyAsync(host3, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) rc = 0;
    }
    (void)hipFree(d);
    return rc;
}

// q = A p of one pixel from the LDS tile (same operations, in the same order, as every other form of A p in this library) and
// the five sums that carry q.  INTERIOR: the sub-domain touches no border of the level, so no neighbour is missing and no weight
// is a merged border weight.
template <bool UNITW, bool INTERIOR>
__device__ __forceinline__ void mid_stencil(const float *s_pu, const float *s_pv, int li, int x, int y, int w, int h, float a1, float a2, float a4,
                                            float wS, float wW, float wE, float wN, float &pcu, float &pcv, float &sumu, float &sumv)
{
    sumu = 0.f; sumv = 0.f;
    if (INTERIOR) {
        const float ws = UNITW ? -1.f : wS, ww = UNITW ? -1.f : wW, we = UNITW ? -1.f : wE, wn = UNITW ? -1.f : wN;
        sumu += ws * s_pu[li - kMidLP]; sumv += ws * s_pv[li - kMidLP];
        sumu += ww * s_pu[li - 1]; sumv += ww * s_pv[li - 1];
        pcu = s_pu[li]; pcv = s_pv[li];
        sumu += a1 * pcu; sumv += a2 * pcu;
        sumu += a2 * pcv; sumv += a4 * pcv;
        sumu += we * s_pu[li + 1]; sumv += we * s_pv[li + 1];
        sumu += wn * s_pu[li + kMidLP]; sumv += wn * s_pv[li + kMidLP];
    } else {
        if (y > 0) { const float ws = UNITW ? ((y == h - 1) ? -2.f : -1.f) : wS; sumu += ws * s_pu[li - kMidLP]; sumv += ws * s_pv[li - kMidLP]; }
        if (x > 0) { const float ww = UNITW ? ((x == w - 1) ? -2.f : -1.f) : wW; sumu += ww * s_pu[li - 1]; sumv += ww * s_pv[li - 1]; }
        pcu = s_pu[li]; pcv = s_pv[li];
        sumu += a1 * pcu; sumv += a2 * pcu;
        sumu += a2 * pcv; sumv += a4 * pcv;
        if (x < w - 1) { const float we = UNITW ? ((x == 0) ? -2.f : -1.f) : wE; sumu += we * s_pu[li + 1]; sumv += we * s_pv[li + 1]; }
        if (y < h - 1) { const float wn = UNITW ? ((y == 0) ? -2.f : -1.f) : wN; sumu += wn * s_pu[li + kMidLP]; sumv += wn * s_pv[li + kMidLP]; }
    }
}

template <int P, bool UNITW>
__global__ __launch_bounds__(kMidT) void k_pcg_solve_mid(LevelPtrs L, MidArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    constexpr int ROWS = P * kMidRG;                       // rows a sub-domain can hold
    float *s_pu = s_mem, *s_pv = s_pu + (ROWS + 2) * kMidLP;
    float *s_xu = s_pv + (ROWS + 2) * kMidLP, *s_xv = s_xu + ROWS * kMidW;
    float *s_edge = s_xv + ROWS * kMidW;                   // r_u, r_v, q_u, q_v of the first / last row [0..7] and of the west / east column [8..15]: 16 x 128
    float *s_acc = s_edge + 16 * kMidEdge;                 // every thread's seven partial sums, [kind][thread]
    double *s_tot = reinterpret_cast<double *>(s_acc + kPartKinds * kMidT);   // the seven folded sums of the previous iteration (+ scratch)
    int *s_flag = reinterpret_cast<int *>(s_tot + 32);     // raised by a lane whose wait was abandoned

    const int tid = threadIdx.x, c_ = tid & (kMidW - 1), rg_ = tid >> 6, wv = tid >> 6, lane = tid & 63;
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int wg = blockIdx.x, bx = wg % A.gx, by = wg / A.gx;
    const int x0 = bx * kMidW, y0 = by * A.bh;
    const int sw = min(kMidW, w - x0), sh = min(A.bh, h - y0);
    const bool interior = x0 > 0 && x0 + sw < w && y0 > 0 && y0 + sh < h;     // no pixel of the sub-domain lies on the level's border

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
    if (tid == 0) *s_flag = 0;
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
            r_iu = rcp_exact(*at(L.a1, r_off)); r_iv = rcp_exact(*at(L.a4, r_off));
        }
    }
    const unsigned e_nb_off = (unsigned)(r_lds >= 0 ? r_nb : wg) * (2 * 4 * 6 * kMidEdge) * 8u;   // byte offset of the neighbour's block of granules

    float rz_prev = st.rz;                                 // (r.z) of iteration k - 1 as that iteration formed it
    float alpha = 0.f;
    int iters = 0, k = A.k0;
    bool stopped = false, aborted = false;
    __syncthreads();

    for (;; k++) {
        // Everything a slot derives from its position (predicates, LDS addresses) is invariant over the iterations, and the compiler
        // would hoist all of it out of this loop into registers it does not have (P = 16: 11 arrays of 16 are live already).  The empty
        // asm statement makes the position opaque once per iteration, so those values are formed again where they are used.
        int c = c_, rg = rg_;
        asm volatile("" : "+v"(c), "+v"(rg));
        const int x = x0 + c;
        const bool colok = c < sw;
        const bool first = (k == 0);
        // ---- wait for what iteration k - 1 left: the G x 7 partial sums (wave j < 7 sweeps sum j: lane l takes workgroups l, l + 64,
        // l + 128, l + 192) and this thread's ring pixel.  All loads of a pass are issued before the first tag is looked at.
        float nalpha = 0.f, beta = 0.f, rz_new, rr;
        unsigned rbits[6] = {0, 0, 0, 0, 0, 0};
        if (first) {
            double t[2];
            fold_band_partials_multi<2, kMidT>(L.band_parts, kPartBlock + kPartRz, kMaxParts, A.nparts_asm, 1, s_tot + 8, t);   // scratch: 16 doubles
            rz_new = (float)t[0]; rr = (float)t[1];
        } else {
            const unsigned want = A.tag0 + (unsigned)k;                      // tag of iteration k - 1
            const bool sweeper = wv < kPartKinds, ringer = r_lds >= 0;
            const unsigned po = (unsigned)((((k + 1) & 1) * 2 * kPartKinds + 2 * wv) * kMidMaxG + lane) * 8u;
            const unsigned eo = e_nb_off + (unsigned)(((((k + 1) & 1) * 4 + r_side) * 6) * kMidEdge + r_idx) * 8u;
            unsigned long long g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            unsigned spins = 0; unsigned long long t0 = 0ull;
            for (;;) {
                bool ok = true;
                if (sweeper) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (lane + 64 * i < A.G) {
                            g[2 * i] = ld_granule(gat(A.parts, po + (unsigned)(64 * i) * 8u));
                            g[2 * i + 1] = ld_granule(gat(A.parts, po + (unsigned)(kMidMaxG + 64 * i) * 8u));
                        }
                    }
                }
                if (ringer) {
                    unsigned long long e[6];
#pragma unroll
                    for (int a = 0; a < 6; a++) e[a] = ld_granule(gat(A.edges, eo + (unsigned)a * kMidEdge * 8u));
#pragma unroll
                    for (int a = 0; a < 6; a++) { rbits[a] = (unsigned)e[a]; ok = ok && (unsigned)(e[a] >> 32) == want; }
                }
                if (sweeper) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (lane + 64 * i < A.G) ok = ok && (unsigned)(g[2 * i] >> 32) == want && (unsigned)(g[2 * i + 1] >> 32) == want;
                }
                if (ok) break;
                if (!mid_keep_waiting(A, spins, t0)) { *s_flag = 1; break; }
            }
            if (sweeper) {
                double v = 0.;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double d = __longlong_as_double((long long)((g[2 * i + 1] << 32) | (g[2 * i] & 0xffffffffull)));
                    v += (lane + 64 * i < A.G) ? d : 0.;
                }
                v = wave_sum(v);
                if (lane == 0) s_tot[wv] = v;
            }
            __syncthreads();
            if (*s_flag) { aborted = true; break; }
            const double rzd = s_tot[0], rrd = s_tot[1], pq = s_tot[2], qz = s_tot[3], qmq = s_tot[4], rq = s_tot[5], qq = s_tot[6];
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
                float r0 = __uint_as_float(rbits[0]), r1 = __uint_as_float(rbits[1]);
                const float q0 = __uint_as_float(rbits[2]), q1 = __uint_as_float(rbits[3]);
                const float p0 = __uint_as_float(rbits[4]), p1 = __uint_as_float(rbits[5]);
                r0 = nalpha * q0 + r0; r1 = nalpha * q1 + r1;
                const float zu = r_iu * r0, zv = r_iv * r1;
                pku = beta * p0 + zu; pkv = beta * p1 + zv;
            }
            s_pu[r_lds] = pku; s_pv[r_lds] = pkv;
        }
        // ---- own pixels: x += alpha p, r -= alpha q, p = M^-1 r + beta p; direct sums of r.z and r.r.  Threads without a pixel
        // in a slot hold zeros there (and a unit diagonal) and compute along: only stores are predicated.
        float acc[kPartKinds];
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) acc[j] = 0.f;
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
            const float iu = rcp_exact(a1[s]), iv = rcp_exact(a4[s]);
            const float zu = iu * ru[s], zv = iv * rv[s];
            const float pnu = first ? zu : beta * pou + zu;
            const float pnv = first ? zv : beta * pov + zv;
            if (ok) { s_pu[li] = pnu; s_pv[li] = pnv; }
            float d = 0.f; d += ru[s] * zu; d += rv[s] * zv; acc[0] += d;
            d = 0.f; d += ru[s] * ru[s]; d += rv[s] * rv[s]; acc[1] += d;
        }
        __syncthreads();
        // ---- q = A p and the sums that carry q.  (The reciprocals of the diagonal and z are formed again rather than kept across the
        // barrier -- four registers per slot; the empty asm statements keep the compiler from "saving" that work.)
#pragma unroll
        for (int s = 0; s < P; s++) {
            asm volatile("" : "+v"(a1[s]), "+v"(a2[s]), "+v"(a4[s]));
            if (!UNITW) asm volatile("" : "+v"(wS[UNITW ? 0 : s]), "+v"(wW[UNITW ? 0 : s]), "+v"(wE[UNITW ? 0 : s]), "+v"(wN[UNITW ? 0 : s]));
        }
        asm volatile("" : "+v"(c), "+v"(rg));
        const int par = k & 1;
#define MID_STENCIL_LOOP(INTERIOR)                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < P; s++) {                                                                                \
            const int ly = s * kMidRG + rg, y = y0 + ly;                                                                               \
            const bool ok = colok && ly < sh;                                                                                          \
            const int li = (ly + 1) * kMidLP + c + 1;                                                                                  \
            float pcu, pcv, sumu, sumv;                                                                                                \
            mid_stencil<UNITW, INTERIOR>(s_pu, s_pv, li, x, y, w, h, a1[s], a2[s], a4[s], UNITW ? 0.f : wS[UNITW ? 0 : s],             \
                                         UNITW ? 0.f : wW[UNITW ? 0 : s], UNITW ? 0.f : wE[UNITW ? 0 : s], UNITW ? 0.f : wN[UNITW ? 0 : s], \
                                         pcu, pcv, sumu, sumv);                                                                        \
            sumu = ok ? sumu : 0.f; sumv = ok ? sumv : 0.f;     /* no pixel here: the neighbours in LDS are somebody else's */         \
            qu[s] = sumu; qv[s] = sumv;                                                                                                \
            const float iu = rcp_exact(a1[s]), iv = rcp_exact(a4[s]);                                                                  \
            const float zu = iu * ru[s], zv = iv * rv[s];                                                                              \
            float d = 0.f; d += pcu * sumu; d += pcv * sumv; acc[2] += d;                                                              \
            d = 0.f; d += sumu * zu; d += sumv * zv; acc[3] += d;                                                                      \
            d = 0.f; d += sumu * (iu * sumu); d += sumv * (iv * sumv); acc[4] += d;                                                    \
            d = 0.f; d += ru[s] * sumu; d += rv[s] * sumv; acc[5] += d;                                                                \
            d = 0.f; d += sumu * sumu; d += sumv * sumv; acc[6] += d;                                                                  \
            /* r and q of the edge pixels go to LDS first (p is there already); the granules are written below, coalesced */           \
            if (ok) {                                                                                                                  \
                if (ly == 0) { s_edge[0 * kMidEdge + c] = ru[s]; s_edge[2 * kMidEdge + c] = rv[s]; s_edge[4 * kMidEdge + c] = sumu; s_edge[6 * kMidEdge + c] = sumv; } \
                if (ly == sh - 1) { s_edge[1 * kMidEdge + c] = ru[s]; s_edge[3 * kMidEdge + c] = rv[s]; s_edge[5 * kMidEdge + c] = sumu; s_edge[7 * kMidEdge + c] = sumv; } \
                if (c == 0) { s_edge[(8 + 0) * kMidEdge + ly] = ru[s]; s_edge[(8 + 2) * kMidEdge + ly] = rv[s]; s_edge[(8 + 4) * kMidEdge + ly] = sumu; s_edge[(8 + 6) * kMidEdge + ly] = sumv; } \
                if (c == sw - 1) { s_edge[(8 + 1) * kMidEdge + ly] = ru[s]; s_edge[(8 + 3) * kMidEdge + ly] = rv[s]; s_edge[(8 + 5) * kMidEdge + ly] = sumu; s_edge[(8 + 7) * kMidEdge + ly] = sumv; } \
                if (A.full_state) {                            /* stepped form: the whole state goes back to the planes */             \
                    const unsigned o = (unsigned)(y * pitch + x) * 4u;                                                                 \
                    *at(L.rb_u[par], o) = ru[s]; *at(L.rb_v[par], o) = rv[s]; *at(L.pf_u[par], o) = pcu; *at(L.pf_v[par], o) = pcv;     \
                    *at(L.qb_u[par], o) = sumu; *at(L.qb_v[par], o) = sumv; *at(L.xu, o) = s_xu[ly * kMidW + c]; *at(L.xv, o) = s_xv[ly * kMidW + c]; \
                }                                                                                                                      \
            }                                                                                                                          \
        }
        if (interior) { MID_STENCIL_LOOP(true) } else { MID_STENCIL_LOOP(false) }
#undef MID_STENCIL_LOOP
        // ---- the workgroup's seven sums: every thread's subtotal (float, over its <= 16 pixels) through LDS, then wave j adds up
        // sum j over the 512 threads in a fixed order, in double
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) s_acc[j * kMidT + tid] = acc[j];
        __syncthreads();                                     // completes s_acc and s_edge
        const unsigned tag = A.tag0 + (unsigned)k + 1u;
        if (wv < kPartKinds) {
            double v = 0.;
#pragma unroll
            for (int i = 0; i < kMidT / 64; i++) v += (double)s_acc[wv * kMidT + lane + 64 * i];
            v = wave_sum(v);
            if (lane < 2) {                                    // lane 0 / 1: low / high half of sum wv
                const unsigned long long b64 = (unsigned long long)__double_as_longlong(v);
                st_granule(gat(A.parts, (unsigned)((par * 2 * kPartKinds + 2 * wv + lane) * kMidMaxG + wg) * 8u), tag, lane ? (unsigned)(b64 >> 32) : (unsigned)b64);
            }
        }
        // ---- publish the edges: side 0 / 1 = first / last row, 2 / 3 = west / east column; arrays r_u r_v q_u q_v p_u p_v
        {
            const int side = tid >> 7, i = tid & (kMidEdge - 1);      // 128 threads per side
            const int len = (side < 2) ? sw : sh;
            if (i < len) {
                const unsigned eo = (unsigned)(((wg * 2 + par) * 4 + side) * 6 * kMidEdge + i) * 8u;
                const int eb = (side < 2) ? side : 8 + (side - 2);
#pragma unroll
                for (int a = 0; a < 4; a++) st_granule(gat(A.edges, eo + (unsigned)a * kMidEdge * 8u), tag, __float_as_uint(s_edge[(eb + 2 * a) * kMidEdge + i]));
                const int li = (side == 0) ? kMidLP + i + 1 : (side == 1) ? sh * kMidLP + i + 1 : (side == 2) ? (i + 1) * kMidLP + 1 : (i + 1) * kMidLP + sw;
                st_granule(gat(A.edges, eo + 4u * kMidEdge * 8u), tag, __float_as_uint(s_pu[li]));
                st_granule(gat(A.edges, eo + 5u * kMidEdge * 8u), tag, __float_as_uint(s_pv[li]));
            }
        }
        rz_prev = rz_new;
        const bool last_of_launch = (k + 1 >= A.k1) && (A.k1 < A.kcap);   // stepped form: the next launch folds these sums
        if (last_of_launch) { k++; break; }
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
    return (size_t)(2 * (rows + 2) * kMidLP + 2 * rows * kMidW + 16 * kMidEdge + kPartKinds * kMidT) * sizeof(float) + 32 * sizeof(double) + 16;
}

// Sub-domain grid of a w x h level on a device with `ncu` CUs: 64-column strips, as many rows of sub-domains as keep every
// workgroup on a CU of its own, P (slots of 8 rows) from {4, 6, .. 16}.  0 = the level does not fit.
int pcg_mid_config(int w, int h, int ncu, int force_p, MidGeom *g)
{
    if (ncu > kMidMaxG) ncu = kMidMaxG;
    const int gx = (w + kMidW - 1) / kMidW;
    if (gx > ncu || (long)w * h <= 0) return 0;
    static const int kP[7] = {4, 6, 8, 10, 12, 14, 16};
    for (int i = 0; i < 7; i++) {
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
    MID_ATTR(4, false); MID_ATTR(4, true); MID_ATTR(6, false); MID_ATTR(6, true); MID_ATTR(8, false); MID_ATTR(8, true);
    MID_ATTR(10, false); MID_ATTR(10, true); MID_ATTR(12, false); MID_ATTR(12, true); MID_ATTR(14, false); MID_ATTR(14, true);
    MID_ATTR(16, false); MID_ATTR(16, true);
#undef MID_ATTR
}

size_t pcg_mid_workspace_bytes()
{
    // [abort word] [partial sums: 2 parities x 14 granules x G] [edge pixels: G x 2 parities x 4 sides x 6 arrays x 128 granules]
    return 256 + (size_t)2 * 2 * kPartKinds * kMidMaxG * 8 + (size_t)kMidMaxG * 2 * 4 * 6 * kMidEdge * 8;
}

// Iterations [k0, k1) of one solve; k0 = 0 and k1 = cgiters is the whole solve in one launch (plus the flow update).
hipError_t launch_pcg_solve_mid(hipStream_t s, const LevelPtrs &L, const MidGeom &g, void *workspace, unsigned seq, int k0, int k1, int kcap,
                                int nparts_asm, float tol)
{
    MidArgs A;
    A.gx = g.gx; A.gy = g.gy; A.bh = g.bh; A.G = g.G;
    A.k0 = k0; A.k1 = k1; A.kcap = kcap; A.nparts_asm = nparts_asm;
    A.full_state = (k0 != 0 || k1 != kcap) ? 1 : 0;
    A.tol = tol;
    char *ws = static_cast<char *>(workspace);
    A.abort_word = reinterpret_cast<unsigned int *>(ws + 8);
    A.parts = reinterpret_cast<unsigned long long *>(ws + 256);
    A.edges = reinterpret_cast<unsigned long long *>(ws + 256 + (size_t)2 * 2 * kPartKinds * kMidMaxG * 8);
    A.tag0 = seq * (unsigned)(kcap + 2);                   // granule tags of this solve: tag0 + 1 .. tag0 + kcap
    const size_t lds = mid_lds_bytes(g.P);
#define MID_LAUNCH(P) \
    do { if (L.unit_w) hipLaunchKernelGGL((k_pcg_solve_mid<P, true>), dim3(g.G), dim3(kMidT), lds, s, L, A); \
         else hipLaunchKernelGGL((k_pcg_solve_mid<P, false>), dim3(g.G), dim3(kMidT), lds, s, L, A); } while (0)
    switch (g.P) {
    case 4: MID_LAUNCH(4); break;
    case 6: MID_LAUNCH(6); break;
    case 8: MID_LAUNCH(8); break;
    case 10: MID_LAUNCH(10); break;
    case 12: MID_LAUNCH(12); break;
    case 14: MID_LAUNCH(14); break;
    default: MID_LAUNCH(16); break;
    }
#undef MID_LAUNCH
    return hipGetLastError();
}

}  // namespace octane
