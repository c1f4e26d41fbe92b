// pcg_persist.hip -- a whole PCG solve of a mid-size pyramid level in ONE launch, the level resident on chip.  gfx950.
//
// Behavioural spec: ref src/oct_variational_optical_flow.cu:1105-1195 (the PCG loop of one linearisation, then u += dx,
// v += dy), in the one-reduction-per-iteration form of k_pcg_fused (pcg_kernels.hip): same operator, same recurrences, same
// stop test, same iteration count.
//
// Why: between the levels one workgroup can hold (k_pcg_solve_small, <= 6144 pixels) and the levels that stream from HBM
// (>= 2 Mpixel) lie the levels whose PCG iteration is pure latency: 10-15 us per launch at 156^2 .. 625^2 and 33 us at
// 1250^2 for a working set that fits the chip's registers and LDS several times over (256 CUs x (512 KB + 160 KB)).  Here
// the level is cut into sub-domains of 64 columns x 8 .. 128 rows (1 .. 16 slots of 8 rows per thread: the fewest that fit the
// CUs, pcg_mid_config), one 512-thread workgroup (one CU) each; r, q and the operator of a pixel stay in its thread's registers
// (from 12 slots on the four neighbour weights in an L2 workspace, MID_WREG_P), p and x in LDS for the whole solve.  Per iteration a workgroup
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

// Diagnostic build (pcg_persist_diag.hip includes this file with MID_DIAG defined and the entry points renamed): thread 0 of
// every workgroup reads the shader clock at the seams of an iteration and adds up where its time goes (tools/probe_mid_stamps.py).
#ifdef MID_DIAG
__device__ unsigned long long g_mid_stamps[32];     // [0..15] sub-domains on the fast path, [16..31] the others
#define MID_STAMP(i) do { if (tid == 0) { const unsigned long long now_ = clock64(); s_stamp[i] += now_ - s_stamp[15]; s_stamp[15] = now_; } } while (0)
#else
#define MID_STAMP(i) do { } while (0)
#endif

#ifndef MID_KEEP_RCP_P
#define MID_KEEP_RCP_P 10  // ... and up to this many keep the reciprocals of the diagonal (two registers per slot) instead of forming them twice per iteration
#endif
// Sub-domains of more than MID_WREG_P slots do not have the registers for the whole operator (11 values per pixel: 154 registers at
// 14 slots, and the compiler spilled 36-468 bytes per lane into scratch, reloaded in the middle of the stencil loop: 1250^2 ran at
// 20.5 us per iteration, 2.4 times the time per slot of 1000^2).  Their four merged neighbour weights live in a workspace instead,
// laid out [workgroup][slot][weight][thread] -- every thread reads back exactly what it wrote, whole wavefronts of consecutive
// floats, out of its XCD's L2 (3.4 MB per XCD at 1250^2) -- and are requested two slots before they are used.
#ifndef MID_WREG_P
#define MID_WREG_P 10
#endif
#ifndef MID_HOIST_P
#define MID_HOIST_P 6     // sub-domains of up to this many slots per thread have the registers to keep what is invariant over the iterations
#endif
constexpr int kMidT = 512;              // threads per workgroup: 8 waves, two per SIMD, up to 256 VGPRs each
constexpr int kMidW = 64;               // columns of a sub-domain: one wavefront per row
constexpr int kMidRG = kMidT / kMidW;   // rows one slot of all threads covers
constexpr int kMidLP = kMidW + 2;       // LDS row of the p tile: west ring pixel, 64 columns, east ring pixel
constexpr int kMidEdge = 128;           // longest edge of a sub-domain (rows: 16 slots x 8)
// The partial sums are an all-to-all: every workgroup reads every workgroup's 14 granules, G x G x 14 reads of the same 28 KB per
// iteration, and at G = 128 .. 256 the few memory channels behind those lines serialise them (the wait for the sums grew linearly with
// G: 0.9 us at 2 workgroups, 3.4 us at 128).  So every workgroup publishes its sums MID_REPS times, into copies an odd number of 4 KB
// pages apart, and reads the copy (workgroup index mod MID_REPS): MID_REPS times fewer readers per line.
#ifndef MID_REPS
#define MID_REPS 8
#endif
constexpr unsigned kMidRepStride = (2u * 2u * kPartKinds * kMidMaxG * 8u) + 4096u;      // bytes between two copies: 57344 + 4096 = 15 pages
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

#ifndef MID_DIAG
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
#endif

typedef float v2f __attribute__((ext_vector_type(2)));     // (u, v) of one pixel: one packed instruction per pair of operations
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }

template <int P, bool UNITW>
__global__ __launch_bounds__(kMidT) void k_pcg_solve_mid(LevelPtrs L, MidArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    constexpr int ROWS = P * kMidRG;                       // rows a sub-domain can hold
    v2f *s_p = reinterpret_cast<v2f *>(s_mem);             // p tile with its one-pixel ring: (ROWS + 2) x 66 pixels of (u, v)
    v2f *s_x = s_p + (ROWS + 2) * kMidLP;                  // x: ROWS x 64
    v2f *s_edge = s_x + ROWS * kMidW;                      // r and q of the edge pixels: [side 0..3][r, q][128]
    float *s_acc = reinterpret_cast<float *>(s_edge + 4 * 2 * kMidEdge);   // every thread's seven partial sums, [kind][thread]
    double *s_tot = reinterpret_cast<double *>(s_acc + kPartKinds * kMidT);   // the seven folded sums of the previous iteration (+ scratch)
    int *s_flag = reinterpret_cast<int *>(s_tot + 32);     // raised by a lane whose wait was abandoned
#ifdef MID_DIAG
    unsigned long long *s_stamp = reinterpret_cast<unsigned long long *>(s_flag + 4);
    if (threadIdx.x == 0) { for (int i = 0; i < 15; i++) s_stamp[i] = 0ull; }
#endif

    const int tid = threadIdx.x, c_ = tid & (kMidW - 1), rg_ = tid >> 6, wv = tid >> 6, lane = tid & 63;
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int wg = blockIdx.x, bx = wg % A.gx, by = wg / A.gx;
    const int x0 = bx * kMidW, y0 = by * A.bh;
    const int sw = min(kMidW, w - x0), sh = min(A.bh, h - y0);
    // a sub-domain that touches no border of the level and fills all 64 x ROWS pixel slots needs no predicate and no border weight
    const bool fast = x0 > 0 && x0 + sw < w && y0 > 0 && y0 + sh < h && sw == kMidW && sh == ROWS;

    // The solve's scalars between the launches of the stepped form, double-buffered by the parity of the first iteration (a
    // one-iteration launch has no barrier, so workgroup 0 may write the new state before another workgroup has read the old)
#ifdef OCTANE_DIAG
    // diagnostic library only (round 4: the product kernel carries no test hook): a workgroup that never shows up -- the others have to
    // give up, not hang (tests/persist_fault_worker.py)
    if (A.fault && wg == A.G - 1) return;
#endif
    PcgState st = L.st[A.k0 & 1];
    if (A.k0 == 0) { st.rz = 0.f; st.stopped = 0; st.iters = 0; }
    if (st.stopped) {                                      // stepped form: the loop ended in an earlier launch (uniform)
        if (blockIdx.x == 0 && threadIdx.x == 0) L.st[A.k1 & 1] = st;
        return;
    }

    // ---- the sub-domain's state: operator, r (p, q, x in the stepped form) -> registers / LDS.  A slot without a pixel (ragged last
    // row / column of sub-domains) holds r = q = 0, a unit diagonal and zero weights: it computes along and stays zero.
    v2f r2[P], q2[P];                                      // p of the own pixels lives in the LDS tile only
    float a1[P], a2[P], a4[P];
    constexpr bool WL2 = P > MID_WREG_P;                   // the weights live in the workspace, not in registers
    constexpr int PW = WL2 ? 1 : P;
    float wS[PW], wW[PW], wE[PW], wN[PW];                  // merged neighbour weights (ref .cu:929-1001); 0 where the level has no such neighbour
    float *const wsp = A.wspill + (size_t)wg * (16 * 4 * kMidT) + tid;   // + (slot * 4 + weight) * kMidT
    for (int i = tid; i < (ROWS + 2) * kMidLP; i += kMidT) s_p[i] = mk2(0.f, 0.f);
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
            r2[s] = mk2(0.f, 0.f); q2[s] = mk2(0.f, 0.f);
            v2f p0 = mk2(0.f, 0.f), xx0 = mk2(0.f, 0.f);
            a1[s] = a4[s] = 1.f; a2[s] = 0.f;
            float ws_ = 0.f, ww_ = 0.f, we_ = 0.f, wn_ = 0.f;
            if (ok) {
                const unsigned o = (unsigned)(y * pitch + x) * 4u;
                a1[s] = *at(L.a1, o); a2[s] = *at(L.a2, o); a4[s] = *at(L.a4, o);
                if (UNITW) {                               // al1 == 1: every weight is exactly -1 (ref .cu:837-864), merged border weights -2
                    ws_ = (y > 0) ? ((y == h - 1) ? -2.f : -1.f) : 0.f;
                    ww_ = (x > 0) ? ((x == w - 1) ? -2.f : -1.f) : 0.f;
                    we_ = (x < w - 1) ? ((x == 0) ? -2.f : -1.f) : 0.f;
                    wn_ = (y < h - 1) ? ((y == 0) ? -2.f : -1.f) : 0.f;
                } else {
                    const float wxc = *at(L.wx, o), wyc = *at(L.wy, o);
                    const float wys = (y > 0) ? *at(L.wy, o - 4u * (unsigned)pitch) : 0.f, wxw = (x > 0) ? *at(L.wx, o - 4u) : 0.f;
                    ws_ = (y > 0) ? ((y == h - 1) ? wys + wyc : wys) : 0.f;
                    ww_ = (x > 0) ? ((x == w - 1) ? wxw + wxc : wxw) : 0.f;
                    we_ = (x < w - 1) ? ((x == 0) ? wxc + wxc : wxc) : 0.f;
                    wn_ = (y < h - 1) ? ((y == 0) ? wyc + wyc : wyc) : 0.f;
                }
                if (A.k0 == 0) {
                    r2[s] = mk2(*at(L.rb_u[0], o), *at(L.rb_v[0], o));      // r_0 = the right-hand side the assembly wrote
                } else {
                    r2[s] = mk2(*at(L.rb_u[par0], o), *at(L.rb_v[par0], o));
                    p0 = mk2(*at(L.pf_u[par0], o), *at(L.pf_v[par0], o));
                    q2[s] = mk2(*at(L.qb_u[par0], o), *at(L.qb_v[par0], o));
                    xx0 = mk2(*at(L.xu, o), *at(L.xv, o));
                }
            }
            if (WL2) {
                if (!(UNITW && fast)) { wsp[(s * 4 + 0) * kMidT] = ws_; wsp[(s * 4 + 1) * kMidT] = ww_; wsp[(s * 4 + 2) * kMidT] = we_; wsp[(s * 4 + 3) * kMidT] = wn_; }
            } else {
                wS[WL2 ? 0 : s] = ws_; wW[WL2 ? 0 : s] = ww_; wE[WL2 ? 0 : s] = we_; wN[WL2 ? 0 : s] = wn_;
            }
            s_x[ly * kMidW + c] = xx0;
            if (ok) s_p[(ly + 1) * kMidLP + c + 1] = p0;       // p_{k0-1} (zero at k0 = 0)
        }
    }
    // ---- this thread's ring pixel (threads 0 .. 127 + 2 ROWS): where it lives, whose edge it is, its preconditioner entries
    int r_lds = -1, r_idx = 0, r_nb = 0, r_side = 0;
    v2f r_i = mk2(0.f, 0.f);
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
            r_i = mk2(rcp_exact(*at(L.a1, r_off)), rcp_exact(*at(L.a4, r_off)));
        }
    }
    const unsigned e_nb_off = (unsigned)(r_lds >= 0 ? r_nb : wg) * (2 * 4 * 6 * kMidEdge) * 8u;   // byte offset of the neighbour's block of granules

    v2f i2s[P <= MID_KEEP_RCP_P ? P : 1];
    if (P <= MID_KEEP_RCP_P) {
#pragma unroll
        for (int s = 0; s < P; s++) i2s[s] = mk2(rcp_exact(a1[s]), rcp_exact(a4[s]));
    }
    float rz_prev = st.rz;                                 // (r.z) of iteration k - 1 as that iteration formed it
    float alpha = 0.f;
    int iters = 0, k = A.k0;
    bool stopped = false, aborted = false;
    __syncthreads();

    for (;; k++) {
        // Everything a slot derives from its position (predicates, LDS addresses) and from the operator (packed pairs, reciprocals) is
        // invariant over the iterations, and the compiler would hoist all of it out of this loop into registers it does not have
        // (P = 14: 11 arrays of 14 are live already).  The empty asm statements make those inputs opaque once per phase, so the
        // derived values are formed again where they are used.
        int c = c_, rg = rg_;
        if (P > MID_HOIST_P) asm volatile("" : "+v"(c), "+v"(rg));
        const bool colok = c < sw;
        const bool first = (k == 0);
#ifdef MID_DIAG
        if (tid == 0) s_stamp[15] = clock64();
#endif
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
            const unsigned po = (unsigned)((((k + 1) & 1) * 2 * kPartKinds + 2 * wv) * kMidMaxG + lane) * 8u + (unsigned)(wg % MID_REPS) * kMidRepStride;
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
            MID_STAMP(0);                                    // wave 0's wait for the sums (and its ring pixels) of iteration k - 1
#ifdef MID_DIAG
            if (tid == 0) s_stamp[7] += spins;               // failed polling rounds of thread 0
#endif
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
            MID_STAMP(1);                                    // ... until the slowest wave has them, folded
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
                    if (colok && ly < sh) s_x[ly * kMidW + c] = alpha * s_p[(ly + 1) * kMidLP + c + 1] + s_x[ly * kMidW + c];   // ref .cu:1172
                }
            }
            stopped = true;
            break;
        }
        if (k >= A.k1) break;                               // stepped form: this launch's share is done
        iters++;
        // ---- p_k on the ring, from the neighbouring sub-domains' edge pixels of iteration k - 1
        if (r_lds >= 0) {
            v2f pk;
            if (first) {
                pk = r_i * mk2(*at(L.rb_u[0], r_off), *at(L.rb_v[0], r_off));
            } else {
                v2f rr2 = mk2(__uint_as_float(rbits[0]), __uint_as_float(rbits[1]));
                const v2f qq2 = mk2(__uint_as_float(rbits[2]), __uint_as_float(rbits[3]));
                const v2f pp2 = mk2(__uint_as_float(rbits[4]), __uint_as_float(rbits[5]));
                rr2 = nalpha * qq2 + rr2;
                pk = beta * pp2 + r_i * rr2;
            }
            s_p[r_lds] = pk;
        }
        // ---- own pixels: x += alpha p, r -= alpha q, p = M^-1 r + beta p; r.z and r.r.  Packed (u, v) arithmetic: one rounding per
        // product and per sum, exactly as the scalar form; the seven sums are kept per component and per thread in float and meet in
        // double across threads.
        v2f acc[kPartKinds];
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) acc[j] = mk2(0.f, 0.f);
#define MID_UPDATE_LOOP(FAST)                                                                                                          \
        _Pragma("unroll") for (int s = 0; s < P; s++) {                                                                                \
            const int ly = s * kMidRG + rg;                                                                                            \
            const bool ok = FAST || (colok && ly < sh);                                                                                \
            const int li = (ly + 1) * kMidLP + c + 1;                                                                                  \
            v2f po = mk2(0.f, 0.f);                                                                /* p_{k-1} of the own pixel */      \
            if (!first) {                                                                                                              \
                if (ok) { po = s_p[li]; s_x[ly * kMidW + c] = alpha * po + s_x[ly * kMidW + c]; }    /* ref .cu:1172 */                  \
                r2[s] = nalpha * q2[s] + r2[s];                                                    /* ref .cu:1174 */                  \
            }                                                                                                                          \
            const v2f i2 = (P <= MID_KEEP_RCP_P) ? i2s[s] : mk2(rcp_exact(a1[s]), rcp_exact(a4[s]));                                   \
            const v2f z2 = i2 * r2[s];                                                                                                 \
            const v2f pn = first ? z2 : beta * po + z2;                                                                                \
            if (ok) s_p[li] = pn;                                                                                                      \
            acc[0] += r2[s] * z2; acc[1] += r2[s] * r2[s];                                                                             \
        }
        if (fast) { MID_UPDATE_LOOP(true) } else { MID_UPDATE_LOOP(false) }
#undef MID_UPDATE_LOOP
        MID_STAMP(2);                                        // scalars, ring pixel, update loop
        __syncthreads();
        MID_STAMP(3);
        // ---- q = A p and the sums that carry q (the reciprocals of the diagonal and z are formed again rather than kept across the barrier)
#pragma unroll
        for (int s = 0; s < P; s++) {
            if (P > MID_HOIST_P && !WL2) asm volatile("" : "+v"(a1[s]), "+v"(a2[s]), "+v"(a4[s]), "+v"(wS[WL2 ? 0 : s]), "+v"(wW[WL2 ? 0 : s]), "+v"(wE[WL2 ? 0 : s]), "+v"(wN[WL2 ? 0 : s]));
            if (WL2) asm volatile("" : "+v"(a1[s]), "+v"(a2[s]), "+v"(a4[s]));
        }
        if (P > MID_HOIST_P) asm volatile("" : "+v"(c), "+v"(rg));
        const int par = k & 1;
#define MID_WLOAD(dst, s_) do { if (WL2 && (s_) < P) { dst[0] = wp[((s_) * 4 + 0) * kMidT]; dst[1] = wp[((s_) * 4 + 1) * kMidT];            \
                                                     dst[2] = wp[((s_) * 4 + 2) * kMidT]; dst[3] = wp[((s_) * 4 + 3) * kMidT]; } } while (0)
        const float *wp = wsp;                               // (opaque once per iteration: these loads are not to be hoisted out of the k loop)
        if (WL2) asm volatile("" : "+v"(wp));
#define MID_STENCIL_LOOP(FAST)                                                                                                         \
        float wq[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};    /* the weights of slot s, s + 1, s + 2 in turn */ \
        if (!(FAST && UNITW)) { MID_WLOAD(wq[0], 0); MID_WLOAD(wq[1], 1); }                                                            \
        _Pragma("unroll") for (int s = 0; s < P; s++) {                                                                                \
            if (!(FAST && UNITW)) MID_WLOAD(wq[(s + 2) % 3], s + 2);                                                                   \
            const int ly = s * kMidRG + rg, y = y0 + ly;                                                                               \
            const bool ok = FAST || (colok && ly < sh);                                                                                \
            const int li = (ly + 1) * kMidLP + c + 1;                                                                                  \
            const float ws = (FAST && UNITW) ? -1.f : (WL2 ? wq[s % 3][0] : wS[WL2 ? 0 : s]), ww = (FAST && UNITW) ? -1.f : (WL2 ? wq[s % 3][1] : wW[WL2 ? 0 : s]); \
            const float we = (FAST && UNITW) ? -1.f : (WL2 ? wq[s % 3][2] : wE[WL2 ? 0 : s]), wn = (FAST && UNITW) ? -1.f : (WL2 ? wq[s % 3][3] : wN[WL2 ? 0 : s]); \
            v2f sum = mk2(0.f, 0.f);                                                                                                   \
            sum += ws * s_p[li - kMidLP];                                                                                              \
            sum += ww * s_p[li - 1];                                                                                                   \
            const v2f pc = s_p[li];                                                                                                    \
            sum += mk2(a1[s], a2[s]) * pc.x;                                                                                           \
            sum += mk2(a2[s], a4[s]) * pc.y;                                                                                           \
            sum += we * s_p[li + 1];                                                                                                   \
            sum += wn * s_p[li + kMidLP];                                                                                              \
            if (!FAST) sum = ok ? sum : mk2(0.f, 0.f);   /* a slot without a pixel: its LDS cell may be a ring cell of the neighbours */ \
            q2[s] = sum;                                                                                                               \
            const v2f i2 = (P <= MID_KEEP_RCP_P) ? i2s[s] : mk2(rcp_exact(a1[s]), rcp_exact(a4[s]));                                   \
            const v2f z2 = i2 * r2[s];                                                                                                 \
            acc[2] += pc * sum; acc[3] += sum * z2; acc[4] += sum * (i2 * sum); acc[5] += r2[s] * sum; acc[6] += sum * sum;            \
            /* r and q of the edge pixels go to LDS first (p is there already); the granules are written below, coalesced */           \
            if (ok) {                                                                                                                  \
                if (ly == 0) { s_edge[(0 * 2 + 0) * kMidEdge + c] = r2[s]; s_edge[(0 * 2 + 1) * kMidEdge + c] = sum; }                 \
                if (ly == sh - 1) { s_edge[(1 * 2 + 0) * kMidEdge + c] = r2[s]; s_edge[(1 * 2 + 1) * kMidEdge + c] = sum; }            \
                if (c == 0) { s_edge[(2 * 2 + 0) * kMidEdge + ly] = r2[s]; s_edge[(2 * 2 + 1) * kMidEdge + ly] = sum; }                \
                if (c == sw - 1) { s_edge[(3 * 2 + 0) * kMidEdge + ly] = r2[s]; s_edge[(3 * 2 + 1) * kMidEdge + ly] = sum; }           \
                if (A.full_state) {                            /* stepped form: the whole state goes back to the planes */             \
                    const unsigned o = (unsigned)(y * pitch + x0 + c) * 4u;                                                            \
                    const v2f xx = s_x[ly * kMidW + c];                                                                                \
                    *at(L.rb_u[par], o) = r2[s].x; *at(L.rb_v[par], o) = r2[s].y; *at(L.pf_u[par], o) = pc.x; *at(L.pf_v[par], o) = pc.y; \
                    *at(L.qb_u[par], o) = sum.x; *at(L.qb_v[par], o) = sum.y; *at(L.xu, o) = xx.x; *at(L.xv, o) = xx.y;                 \
                }                                                                                                                      \
            }                                                                                                                          \
        }
        if (fast) { MID_STENCIL_LOOP(true) } else { MID_STENCIL_LOOP(false) }
#undef MID_STENCIL_LOOP
#undef MID_WLOAD
        MID_STAMP(4);                                        // stencil loop
        // ---- the workgroup's seven sums: every thread's subtotal through LDS, then wave j adds up sum j over the 512 threads in a
        // fixed order, in double
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) s_acc[j * kMidT + tid] = acc[j].x + acc[j].y;
        __syncthreads();                                     // completes s_acc and s_edge
        MID_STAMP(5);
        const unsigned tag = A.tag0 + (unsigned)k + 1u;
        // ---- publish: side 0 / 1 = first / last row, 2 / 3 = west / east column (128 threads per side); arrays r_u r_v q_u q_v p_u p_v.
        // Small sub-domains send the edges first (their stores are on their way while the sums are still being added up: 125^2 0.95 ->
        // 0.86 ms per 270 iterations); with six slots or more the edges are long, every workgroup waits for everybody's sums, and the
        // sums go first (1000^2: 2.75 against 3.21 ms the other way round)
#define MID_PUBLISH_EDGES() \
        { \
            const int side = tid >> 7, i = tid & (kMidEdge - 1); \
            const int len = (side < 2) ? sw : sh; \
            if (i < len) { \
                const unsigned eo = (unsigned)(((wg * 2 + par) * 4 + side) * 6 * kMidEdge + i) * 8u; \
                const v2f er = s_edge[(side * 2 + 0) * kMidEdge + i], eq = s_edge[(side * 2 + 1) * kMidEdge + i]; \
                const int li = (side == 0) ? kMidLP + i + 1 : (side == 1) ? sh * kMidLP + i + 1 : (side == 2) ? (i + 1) * kMidLP + 1 : (i + 1) * kMidLP + sw; \
                const v2f ep = s_p[li]; \
                st_granule(gat(A.edges, eo), tag, __float_as_uint(er.x)); \
                st_granule(gat(A.edges, eo + 1u * kMidEdge * 8u), tag, __float_as_uint(er.y)); \
                st_granule(gat(A.edges, eo + 2u * kMidEdge * 8u), tag, __float_as_uint(eq.x)); \
                st_granule(gat(A.edges, eo + 3u * kMidEdge * 8u), tag, __float_as_uint(eq.y)); \
                st_granule(gat(A.edges, eo + 4u * kMidEdge * 8u), tag, __float_as_uint(ep.x)); \
                st_granule(gat(A.edges, eo + 5u * kMidEdge * 8u), tag, __float_as_uint(ep.y)); \
            } \
        }
        if (P <= 4) { MID_PUBLISH_EDGES(); }
        // ---- the workgroup's seven sums
        if (wv < kPartKinds) {
            double v = 0.;
#pragma unroll
            for (int i = 0; i < kMidT / 64; i++) v += (double)s_acc[wv * kMidT + lane + 64 * i];
            v = wave_sum(v);
            if (lane < 2 * MID_REPS) {                         // even / odd lane: low / high half of sum wv, lane / 2: the copy
                const unsigned long long b64 = (unsigned long long)__double_as_longlong(v);
                const int half = lane & 1;
                st_granule(gat(A.parts, (unsigned)((par * 2 * kPartKinds + 2 * wv + half) * kMidMaxG + wg) * 8u + (unsigned)(lane >> 1) * kMidRepStride), tag,
                           half ? (unsigned)(b64 >> 32) : (unsigned)b64);
            }
        }
        if (P > 4) { MID_PUBLISH_EDGES(); }
#undef MID_PUBLISH_EDGES
        MID_STAMP(6);                                        // workgroup sums and edges published
#ifdef MID_DIAG
        if (tid == 0) s_stamp[14] += 1;
#endif
        rz_prev = rz_new;
        const bool last_of_launch = (k + 1 >= A.k1) && (A.k1 < A.kcap);   // stepped form: the next launch folds these sums
        if (last_of_launch) { k++; break; }
    }
    if (aborted) return;
    __syncthreads();
#ifdef MID_DIAG
    if (tid == 0) { for (int i = 0; i < 15; i++) atomicAdd(&g_mid_stamps[(fast ? 0 : 16) + i], s_stamp[i]); }
#endif
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
                    const v2f d = s_x[ly * kMidW + c];
                    *at(L.u, o) = *at(L.u, o) + d.x; *at(L.v, o) = *at(L.v, o) + d.y;
                    *at(L.xu, o) = d.x; *at(L.xv, o) = d.y;      // kept for the debug tap
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
    return (size_t)(2 * (rows + 2) * kMidLP + 2 * rows * kMidW + 2 * 4 * 2 * kMidEdge + kPartKinds * kMidT) * sizeof(float) + 32 * sizeof(double) + 16 + 16 * sizeof(unsigned long long);
}

#ifndef MID_DIAG
static int g_mid_min_p = 1;
void set_mid_min_p(int p) { g_mid_min_p = p < 1 ? 1 : p; }
// Sub-domain grid of a w x h level on a device with `ncu` CUs: 64-column strips, as many rows of sub-domains as keep every
// workgroup on a CU of its own, P (slots of 8 rows) from {1, 2, 4, 6, .. 16}.  0 = the level does not fit.
int pcg_mid_config(int w, int h, int ncu, int force_p, MidGeom *g)
{
    if (ncu > kMidMaxG) ncu = kMidMaxG;
    const int gx = (w + kMidW - 1) / kMidW;
    if (gx > ncu || (long)w * h <= 0) return 0;
    static const int kP[9] = {1, 2, 4, 6, 8, 10, 12, 14, 16};
    for (int i = 0; i < 9; i++) {
        const int P = kP[i];
        if (force_p > 0 && P != force_p) continue;
        if (force_p <= 0 && P < (force_p < 0 ? -force_p : g_mid_min_p)) continue;        // force_p < 0: at least that many slots
        const int rows = P * kMidRG;
        int gy = (h + rows - 1) / rows;
        if (gx * gy > ncu) continue;
        // An iteration costs ~3.4 us of exchange and reductions plus ~0.45 us per slot; the exchange grows with the number of workgroups
        // (+2 us from 8 to 128), so sub-domains of one or two slots pay only up to 128 workgroups (tools/mid_minp.py, round 3:
        // 63^2 1.20 -> 0.93 ms per 270 iterations, 250^2 1.34 -> 1.15, 313^2 with two slots 1.38 -> 1.24; 500^2 is faster with four)
        if (P < 4 && gx * gy > 128) continue;
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

#endif

void pcg_mid_configure()
{
#define MID_ATTR(P, U) (void)hipFuncSetAttribute((const void *)k_pcg_solve_mid<P, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds_bytes(P))
    MID_ATTR(1, false); MID_ATTR(1, true); MID_ATTR(2, false); MID_ATTR(2, true);
    MID_ATTR(4, false); MID_ATTR(4, true); MID_ATTR(6, false); MID_ATTR(6, true); MID_ATTR(8, false); MID_ATTR(8, true);
    MID_ATTR(10, false); MID_ATTR(10, true); MID_ATTR(12, false); MID_ATTR(12, true); MID_ATTR(14, false); MID_ATTR(14, true);
    MID_ATTR(16, false); MID_ATTR(16, true);
#undef MID_ATTR
}

#ifndef MID_DIAG
size_t pcg_mid_workspace_bytes()
{
    // [abort word] [MID_REPS copies of the partial sums: 2 parities x 14 granules x G, + one page] [edge pixels: G x 2 parities x 4 sides x 6 arrays x 128 granules] [weights of the 12-16-slot sub-domains: G x 16 slots x 4 x 512 floats]
    return 256 + (size_t)MID_REPS * kMidRepStride + (size_t)kMidMaxG * 2 * 4 * 6 * kMidEdge * 8 + (size_t)kMidMaxG * 16 * 4 * kMidT * sizeof(float);
}

#endif
// Iterations [k0, k1) of one solve; k0 = 0 and k1 = cgiters is the whole solve in one launch (plus the flow update).
static int g_mid_fault = 0;
#if defined(OCTANE_DIAG) && !defined(MID_DIAG)
void set_mid_fault(int v) { g_mid_fault = v != 0; }
#endif

hipError_t launch_pcg_solve_mid(hipStream_t s, const LevelPtrs &L, const MidGeom &g, void *workspace, unsigned seq, int k0, int k1, int kcap,
                                int nparts_asm, float tol)
{
    MidArgs A;
    A.fault = g_mid_fault;
    A.gx = g.gx; A.gy = g.gy; A.bh = g.bh; A.G = g.G;
    A.k0 = k0; A.k1 = k1; A.kcap = kcap; A.nparts_asm = nparts_asm;
    A.full_state = (k0 != 0 || k1 != kcap) ? 1 : 0;
    A.tol = tol;
    char *ws = static_cast<char *>(workspace);
    A.abort_word = reinterpret_cast<unsigned int *>(ws + 8);
    A.parts = reinterpret_cast<unsigned long long *>(ws + 256);
    A.edges = reinterpret_cast<unsigned long long *>(ws + 256 + (size_t)MID_REPS * kMidRepStride);
    A.wspill = reinterpret_cast<float *>(ws + 256 + (size_t)MID_REPS * kMidRepStride + (size_t)kMidMaxG * 2 * 4 * 6 * kMidEdge * 8);
    A.tag0 = seq * (unsigned)(kcap + 2);                   // granule tags of this solve: tag0 + 1 .. tag0 + kcap
    const size_t lds = mid_lds_bytes(g.P);
#define MID_LAUNCH(P) \
    do { if (L.unit_w) hipLaunchKernelGGL((k_pcg_solve_mid<P, true>), dim3(g.G), dim3(kMidT), lds, s, L, A); \
         else hipLaunchKernelGGL((k_pcg_solve_mid<P, false>), dim3(g.G), dim3(kMidT), lds, s, L, A); } while (0)
    switch (g.P) {
    case 1: MID_LAUNCH(1); break;
    case 2: MID_LAUNCH(2); break;
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

#ifdef MID_DIAG
// read (and clear) the stamps: [0..6] cycles per seam summed over the workgroups' thread 0 and the iterations, [7] failed polling
// rounds of thread 0, [14] iterations x workgroups; the same at [16..] for the sub-domains that touch a border of the level or are
// not full (the predicated path)
int pcg_mid_stamps(hipStream_t s, unsigned long long *out32)
{
    static const unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbolAsync(out32, HIP_SYMBOL(g_mid_stamps), sizeof zero, 0, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_mid_stamps), zero, sizeof zero, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace octane
