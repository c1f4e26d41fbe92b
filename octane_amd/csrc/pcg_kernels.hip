// pcg_kernels.hip -- the inner linear solver: Jacobi-preconditioned CG, matrix-free, gfx950.
//
// Behavioural spec: ref src/oct_variational_optical_flow.cu:1105-1195 (PCG on an assembled CSR matrix, 12 grid
// barriers and 5 atomics-based dot products per iteration, A*p computed twice).  The iterates here are the same --
// same operator, same preconditioner, same recurrences, same iteration count, same stop test -- but one iteration
// is two streaming kernels over row-pitched planes:
//
//   pass A(k):  beta = (r.z)_k / (r.z)_{k-1};  p <- M^-1 r + beta p;  q <- A p;  partials of p.q
//   pass B(k):  alpha = (r.z)_k / (p.q);  x <- x + alpha p;  r <- r - alpha q;  partials of r.M^-1 r, r.r
//
// (the reference's z0.bcu and bcu.z0 products are the previous/current r.z, so three of its five dot products are
// redundant).  A acts through five coefficient planes (see k_assemble).
//
// Reductions are two-stage and deterministic with no atomics, fences or extra launches: each persistent workgroup
// writes one double partial; every workgroup of the NEXT kernel folds the (at most 2048) partials in the same fixed
// order (device_util.hpp), so all of them agree bitwise on alpha/beta and on the stop decision.  The kernel
// boundary provides the visibility.
//
// HBM bytes per pixel per iteration: A reads r(2) p(2) a1 a2 a4 wx wy = 36 B and writes p(2) q(2)
// = 16 B; B reads r(2) p(2) q(2) mu mv = 32 B (+ x(2) p_prev(2) = 16 B every second iteration) and writes r(2)
// (+ x(2) every second iteration).  Single-use planes (x, q, mu/mv, a2) move with streaming hints so that the
// planes both passes share (r, p) keep the Infinity Cache; pass B walks the frame backwards for the same reason.
//
// Pass A comes in three forms chosen per level (pass_a_choice): a latency-oriented tiled form below 1 Mpixel, 128 x 16
// tiles, and k_pcg_pass_a_ring, a marching form with 8 % instead of 30 % re-fetch that wins at 4-12 Mpixel only
// (EXPERIMENTS.md 8).  The tiled and marching forms skip the wx / wy planes in the first GNC step, where every weight is -1,
// and the tiled form can work on a row band of the level (vof_tiled.hip).
// Also here: k_pcg_solve_small (a whole solve in one workgroup for the coarsest levels) and k_flow_update.
#include <vector>

#include "vof_kernels.hpp"
#include "device_util.hpp"

namespace octane {

constexpr int kLRow = kTileX + 8;   // LDS row: [3 pad][west halo][kTileX interior][east halo][3 pad]
constexpr int kLInt = 4;            // interior starts 16-byte aligned

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// Streaming ("nt") variants for data with no reuse before it would be evicted anyway: they keep single-use
// planes from displacing the planes pass A and pass B share (r, p) in the Infinity Cache.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float *p)
{
    f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float *p, float4 v)
{
    f4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
}
// plane base + 32-bit byte offset: the scalar-base addressing form, one VGPR of offset shared by every plane instead of
// a 64-bit address pair per plane and group (planes addressed this way are smaller than 4 GiB)
__device__ __forceinline__ const float *at(const float *base, unsigned byte_off) { return (const float *)((const char *)base + byte_off); }
__device__ __forceinline__ float *at(float *base, unsigned byte_off) { return (float *)((char *)base + byte_off); }
__device__ __forceinline__ float4 ld4_if(const float *p, bool nt) { return nt ? ld4_nt(p) : ld4(p); }
__device__ __forceinline__ void st4_if(float *p, float4 v, bool nt) { if (nt) st4_nt(p, v); else st4(p, v); }

// p_new = z + beta * p_old with z = M^-1 r  (ref .cu:1117/1138 then jVecPVec(p0,z0,p0,Bk) at :1146)
// The preconditioner entry is re-derived from the diagonal here (pass A reads a1/a4 anyway for
// A p) with one correctly rounded float division.  The reference rounds 1./M through double
// first; the two agree except when the double quotient sits exactly on a float rounding
// boundary (probability ~2^-29 per value, 1 ulp then).
__device__ __forceinline__ float direction(float r, float pold, float diag, float beta, bool first)
{
    float z = rcp_exact(diag) * r;          // the bits of 1.0f / diag in three instructions (device_util.hpp)
    return first ? z : beta * pold + z;
}

// Pass A.  A 256-thread workgroup iteration covers a 128 x 8R tile: thread (lx, ly) owns the 4 pixels at columns
// 4*lx.. of rows ly, ly+8, ..  p_new = M^-1 r + beta p goes into an LDS tile with a one-pixel halo (recomputed from
// r, p_old and the diagonal of the neighbouring pixels), then q = A p_new is formed from LDS + registers.  R = 2
// (128 x 16) is the default: against R = 1 the two halo rows are amortised over 16 rows and there are half as many
// barriers per pixel (-6 % time); R = 4 needs 236 VGPRs and loses more in occupancy than it saves.
// UNITW: the first GNC step (al1 == 1) makes every neighbour weight exactly -1 (k_assemble: a7 = a8 =
// (float)(-1 * (1 + 0 * psi)) ), so the wx / wy planes hold a constant and are not read: 44 instead of 52 B/pixel for
// a third of all iterations, same bits.
template <int R, bool WIDE, bool UNITW>
__global__ __launch_bounds__(256) void k_pcg_pass_a(LevelPtrs L, int k, int nparts_prev, float tol)
{
    constexpr int TY = WIDE ? kTileY : kTileY * R;          // tile rows
    constexpr int TX = WIDE ? kTileX * R : kTileX;          // tile columns
    constexpr int LROW = TX + 8;                             // LDS row: [3 pad][west][TX interior][east][3 pad]
    constexpr int HL = TX / 4;                               // float4 lanes per halo row
    __shared__ __attribute__((aligned(16))) float s_pu[(TY + 2) * LROW];
    __shared__ __attribute__((aligned(16))) float s_pv[(TY + 2) * LROW];
    __shared__ double s_red[8];
    const int tid = threadIdx.x;

    const PcgState prev = L.st[k & 1];                               // requested together with the partials
    const float rz_new = (float)fold_band_partials_256(L.band_parts, kPartRz, nparts_prev, L.nbands, s_red);
    const float rr = (float)fold_band_partials_256(L.band_parts, kPartRr, nparts_prev, L.nbands, s_red);
    const bool active = (prev.stopped == 0) && (rr > tol);          // ref .cu:1131
    if (!active) {
        if (blockIdx.x == 0 && tid == 0) { PcgState n = prev; n.stopped = 1; L.st[(k + 1) & 1] = n; }
        return;
    }
    const bool first = (k == 0);
    const float beta = first ? 0.f : rz_new / prev.rz;
    if (blockIdx.x == 0 && tid == 0) {
        PcgState n; n.rz = rz_new; n.stopped = 0; n.iters = prev.iters + 1; n.pad = 0;
        L.st[(k + 1) & 1] = n;
    }

    // This launch covers rows [y0, y1) of the frame (the whole frame for a plain plan).  Rows y0-1 and y1 of a
    // band's inner edges belong to the neighbouring bands: r there is read from the neighbour's own plane (its
    // pass B finished before this launch started), the coefficients come from the band's own (overlapping)
    // assembly, and p is kept up to date right here --
    // the halo rows of p_new every tile recomputes anyway are stored when they lie outside the band, with the
    // very arithmetic the owning band uses, so p needs no exchange.
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int by0 = L.y0, by1 = L.y1;
    const int tiles_x = (w + TX - 1) / TX, tiles_y = (by1 - by0 + TY - 1) / TY;
    const int ntiles = tiles_x * tiles_y;
    const int lx = tid & 31, ly = tid >> 5;
    double acc = 0.;
    const float *__restrict__ pin_u = L.pu[k & 1];
    const float *__restrict__ pin_v = L.pv[k & 1];
    float *__restrict__ pout_u = L.pu[(k + 1) & 1];
    float *__restrict__ pout_v = L.pv[(k + 1) & 1];

    const ItemRange tr = item_range(ntiles, L.xcd_bands == 1);
    for (int t = tr.first; t < tr.end; t += tr.step) {
        const int tx0 = (t % tiles_x) * TX, ty0 = by0 + (t / tiles_x) * TY;
        float a1[R][4], a4[R][4], a2[R][4], wxc[R][4], wyc[R][4], wys[R][4], npu[R][4], npv[R][4];
        float wxw[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int x = tx0 + lx * 4 + (WIDE ? kTileX * q : 0);
            const int y = ty0 + ly + (WIDE ? 0 : kTileY * q);
            const int lrow1 = ly + (WIDE ? 0 : kTileY * q) + 1, lcol = kLInt + lx * 4 + (WIDE ? kTileX * q : 0);
            const bool rowok = (y < by1) && (x < w);
            const size_t o = (size_t)y * pitch + x;
            float ru[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0};
            wxw[q] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) { a1[q][e] = 1.f; a4[q][e] = 1.f; a2[q][e] = 0.f; wxc[q][e] = 0.f; wyc[q][e] = 0.f; wys[q][e] = 0.f; npu[q][e] = 0.f; npv[q][e] = 0.f; }
            if (rowok) {
                *(float4 *)ru = ld4(L.ru + o);
                *(float4 *)rv = ld4(L.rv + o);
                *(float4 *)a1[q] = ld4(L.a1 + o);
                *(float4 *)a4[q] = ld4(L.a4 + o);
                *(float4 *)a2[q] = ld4_if(L.a2 + o, L.nt_hints & 8);
                if (UNITW) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { wxc[q][e] = -1.f; wyc[q][e] = -1.f; wys[q][e] = -1.f; }
                    wxw[q] = -1.f;
                } else {
                    *(float4 *)wxc[q] = ld4(L.wx + o);
                    *(float4 *)wyc[q] = ld4(L.wy + o);
                    if (y > 0) *(float4 *)wys[q] = ld4(L.wy + o - pitch);
                    if (x > 0) wxw[q] = L.wx[o - 1];
                }
                float pu[4] = {0, 0, 0, 0}, pv[4] = {0, 0, 0, 0};
                if (!first) { *(float4 *)pu = ld4(pin_u + o); *(float4 *)pv = ld4(pin_v + o); }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool ok = (x + e) < w;
                    npu[q][e] = ok ? direction(ru[e], pu[e], a1[q][e], beta, first) : 0.f;
                    npv[q][e] = ok ? direction(rv[e], pv[e], a4[q][e], beta, first) : 0.f;
                }
            }
            st4(&s_pu[lrow1 * LROW + lcol], *(float4 *)npu[q]);
            st4(&s_pv[lrow1 * LROW + lcol], *(float4 *)npv[q]);
        }
        // one-pixel halo of p_new, recomputed from r, p_old and the diagonal
        if (tid < 2 * HL) {                              // rows above and below the tile
            const int hy = (tid < HL) ? ty0 - 1 : ty0 + TY;
            const int hx = tx0 + (tid % HL) * 4;
            const int lrow = (tid < HL) ? 0 : TY + 1;
            float hu[4] = {0, 0, 0, 0}, hv[4] = {0, 0, 0, 0};
            if (hy >= 0 && hy < h && hx < w) {
                const size_t ho = (size_t)hy * pitch + hx;
                float r0[4], r1[4], d0[4], d1[4], q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0};
                // r of a neighbouring band's row comes straight from that band's plane
                const float *hru = (hy < by0) ? L.ru_up : (hy >= by1) ? L.ru_dn : L.ru;
                const float *hrv = (hy < by0) ? L.rv_up : (hy >= by1) ? L.rv_dn : L.rv;
                *(float4 *)r0 = ld4(hru + ho); *(float4 *)r1 = ld4(hrv + ho);
                *(float4 *)d0 = ld4(L.a1 + ho); *(float4 *)d1 = ld4(L.a4 + ho);
                if (!first) { *(float4 *)q0 = ld4(pin_u + ho); *(float4 *)q1 = ld4(pin_v + ho); }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool ok = (hx + e) < w;
                    hu[e] = ok ? direction(r0[e], q0[e], d0[e], beta, first) : 0.f;
                    hv[e] = ok ? direction(r1[e], q1[e], d1[e], beta, first) : 0.f;
                }
                if (hy < by0 || hy >= by1) {                 // a neighbouring band's row: keep our copy of p current
                    st4(pout_u + ho, *(float4 *)hu);
                    st4(pout_v + ho, *(float4 *)hv);
                }
            }
            st4(&s_pu[lrow * LROW + kLInt + (tid % HL) * 4], *(float4 *)hu);
            st4(&s_pv[lrow * LROW + kLInt + (tid % HL) * 4], *(float4 *)hv);
        } else if (tid < 2 * HL + 2 * TY) {              // columns left and right of the tile
            const int side = (tid - 2 * HL) / TY, row = (tid - 2 * HL) % TY;
            const int hy = ty0 + row;
            const int hx = side ? tx0 + TX : tx0 - 1;
            float hu = 0.f, hv = 0.f;
            if (hy < by1 && hx >= 0 && hx < w) {
                const size_t ho = (size_t)hy * pitch + hx;
                const float q0 = first ? 0.f : pin_u[ho], q1 = first ? 0.f : pin_v[ho];
                hu = direction(L.ru[ho], q0, L.a1[ho], beta, first);
                hv = direction(L.rv[ho], q1, L.a4[ho], beta, first);
            }
            const int lcol = side ? kLInt + TX : kLInt - 1;
            s_pu[(row + 1) * LROW + lcol] = hu;
            s_pv[(row + 1) * LROW + lcol] = hv;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int x = tx0 + lx * 4 + (WIDE ? kTileX * q : 0);
            const int y = ty0 + ly + (WIDE ? 0 : kTileY * q);
            const int lrow = ly + (WIDE ? 0 : kTileY * q), lc = kLInt + lx * 4 + (WIDE ? kTileX * q : 0);
            if ((y < by1) && (x < w)) {
                float su[4], sv[4], nu[4], nv[4];
                *(float4 *)su = ld4(&s_pu[lrow * LROW + lc]);
                *(float4 *)sv = ld4(&s_pv[lrow * LROW + lc]);
                *(float4 *)nu = ld4(&s_pu[(lrow + 2) * LROW + lc]);
                *(float4 *)nv = ld4(&s_pv[(lrow + 2) * LROW + lc]);
                const float uwest = s_pu[(lrow + 1) * LROW + lc - 1];
                const float vwest = s_pv[(lrow + 1) * LROW + lc - 1];
                const float ueast = s_pu[(lrow + 1) * LROW + lc + 4];
                const float veast = s_pv[(lrow + 1) * LROW + lc + 4];
                float qu[4], qv[4];
                float rowdot = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = x + e;
                    const float pwu = (e == 0) ? uwest : npu[q][(e + 3) & 3], pwv = (e == 0) ? vwest : npv[q][(e + 3) & 3];
                    const float peu = (e == 3) ? ueast : npu[q][(e + 1) & 3], pev = (e == 3) ? veast : npv[q][(e + 1) & 3];
                    const float a5 = (e == 0) ? wxw[q] : wxc[q][(e + 3) & 3];
                    const float wS = (y == h - 1) ? wys[q][e] + wyc[q][e] : wys[q][e];
                    const float wW = (i == w - 1) ? a5 + wxc[q][e] : a5;
                    const float wE = (i == 0) ? wxc[q][e] + wxc[q][e] : wxc[q][e];
                    const float wN = (y == 0) ? wyc[q][e] + wyc[q][e] : wyc[q][e];
                    float sumu = 0.f, sumv = 0.f;
                    if (y > 0) { sumu += wS * su[e]; sumv += wS * sv[e]; }
                    if (i > 0) { sumu += wW * pwu; sumv += wW * pwv; }
                    sumu += a1[q][e] * npu[q][e]; sumv += a2[q][e] * npu[q][e];
                    sumu += a2[q][e] * npv[q][e]; sumv += a4[q][e] * npv[q][e];
                    if (i < w - 1) { sumu += wE * peu; sumv += wE * pev; }
                    if (y < h - 1) { sumu += wN * nu[e]; sumv += wN * nv[e]; }
                    qu[e] = sumu; qv[e] = sumv;
                    if (i < w) { rowdot += npu[q][e] * sumu; rowdot += npv[q][e] * sumv; }
                }
                const size_t o = (size_t)y * pitch + x;
                st4(pout_u + o, *(float4 *)npu[q]);
                st4(pout_v + o, *(float4 *)npv[q]);
                st4_if(L.qu + o, *(float4 *)qu, L.nt_hints & 16);
                st4_if(L.qv + o, *(float4 *)qv, L.nt_hints & 16);
                acc += (double)rowdot;
            }
        }
        __syncthreads();
    }
    const double tot = block_sum_256(acc, s_red);
    if (tid == 0) L.part_pq[blockIdx.x] = tot;
}

template <int R>
struct PassATile {            // everything one workgroup iteration loads, per thread
    float ru[R][4], rv[R][4], pu[R][4], pv[R][4];
    float a1[R][4], a4[R][4], a2[R][4], wxc[R][4], wyc[R][4], wys[R][4];
    float wxw[R];
    float hr[6][4];           // halo-row lanes (tid < 64): r_u r_v a1 a4 p_u p_v of one float4 group
    float hs[6];              // halo-column lanes: the same six values of one pixel
};

// Latency-oriented form of pass A for the levels below one Mpixel (R = 1 there): loads and arithmetic are separate
// phases so that the first tile's operands can be requested before the reduction partials and the solve state --
// one memory round trip instead of three.  It holds everything in registers at once (185 VGPRs), which does not
// matter at these sizes (one or two workgroups per CU).
template <int R>
__global__ __launch_bounds__(256) void k_pcg_pass_a_lat(LevelPtrs L, int k, int nparts_prev, float tol)
{
    constexpr int TY = kTileY * R;
    __shared__ __attribute__((aligned(16))) float s_pu[(TY + 2) * kLRow];
    __shared__ __attribute__((aligned(16))) float s_pv[(TY + 2) * kLRow];
    __shared__ double s_red[8];
    const int tid = threadIdx.x;
    const bool first = (k == 0);
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int tiles_x = (w + kTileX - 1) / kTileX, tiles_y = (h + TY - 1) / TY;
    const int ntiles = tiles_x * tiles_y;
    const int lx = tid & 31, ly = tid >> 5;
    const float *__restrict__ pin_u = L.pu[k & 1];
    const float *__restrict__ pin_v = L.pv[k & 1];
    float *__restrict__ pout_u = L.pu[(k + 1) & 1];
    float *__restrict__ pout_v = L.pv[(k + 1) & 1];
    const ItemRange tr = item_range(ntiles, L.xcd_bands == 1);

    // Phase 1 of a tile: nothing but loads -- none of them depends on beta or on the stop decision.
    auto load_tile = [&](int t, PassATile<R> &g) {
        const int tx0 = (t % tiles_x) * kTileX, ty0 = (t / tiles_x) * TY;
        const int x = tx0 + lx * 4;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int y = ty0 + ly + kTileY * q;
            const size_t o = (size_t)y * pitch + x;
            g.wxw[q] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                g.ru[q][e] = 0.f; g.rv[q][e] = 0.f; g.pu[q][e] = 0.f; g.pv[q][e] = 0.f;
                g.a1[q][e] = 1.f; g.a4[q][e] = 1.f; g.a2[q][e] = 0.f; g.wxc[q][e] = 0.f; g.wyc[q][e] = 0.f; g.wys[q][e] = 0.f;
            }
            if ((y < h) && (x < w)) {
                *(float4 *)g.ru[q] = ld4(L.ru + o);
                *(float4 *)g.rv[q] = ld4(L.rv + o);
                *(float4 *)g.a1[q] = ld4(L.a1 + o);
                *(float4 *)g.a4[q] = ld4(L.a4 + o);
                *(float4 *)g.a2[q] = ld4_if(L.a2 + o, L.nt_hints & 8);
                *(float4 *)g.wxc[q] = ld4(L.wx + o);
                *(float4 *)g.wyc[q] = ld4(L.wy + o);
                if (y > 0) *(float4 *)g.wys[q] = ld4(L.wy + o - pitch);
                if (x > 0) g.wxw[q] = L.wx[o - 1];
                if (!first) { *(float4 *)g.pu[q] = ld4(pin_u + o); *(float4 *)g.pv[q] = ld4(pin_v + o); }
            }
        }
        if (tid < 64) {                                  // rows above and below the tile
            const int hy = (tid < 32) ? ty0 - 1 : ty0 + TY;
            const int hx = tx0 + (tid & 31) * 4;
#pragma unroll
            for (int c = 0; c < 6; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) g.hr[c][e] = (c == 2 || c == 3) ? 1.f : 0.f;
            if (hy >= 0 && hy < h && hx < w) {
                const size_t ho = (size_t)hy * pitch + hx;
                *(float4 *)g.hr[0] = ld4(L.ru + ho); *(float4 *)g.hr[1] = ld4(L.rv + ho);
                *(float4 *)g.hr[2] = ld4(L.a1 + ho); *(float4 *)g.hr[3] = ld4(L.a4 + ho);
                if (!first) { *(float4 *)g.hr[4] = ld4(pin_u + ho); *(float4 *)g.hr[5] = ld4(pin_v + ho); }
            }
        } else if (tid < 64 + 2 * TY) {                  // columns left and right of the tile
            const int side = (tid - 64) / TY, row = (tid - 64) % TY;
            const int hy = ty0 + row;
            const int hx = side ? tx0 + kTileX : tx0 - 1;
            g.hs[0] = 0.f; g.hs[1] = 0.f; g.hs[2] = 1.f; g.hs[3] = 1.f; g.hs[4] = 0.f; g.hs[5] = 0.f;
            if (hy < h && hx >= 0 && hx < w) {
                const size_t ho = (size_t)hy * pitch + hx;
                g.hs[0] = L.ru[ho]; g.hs[1] = L.rv[ho]; g.hs[2] = L.a1[ho]; g.hs[3] = L.a4[ho];
                if (!first) { g.hs[4] = pin_u[ho]; g.hs[5] = pin_v[ho]; }
            }
        }
    };

    // At the coarse levels a pass is a chain of memory round trips; the first tile's operands are therefore
    // requested before the reduction partials and the solve state, so that all of it is one round trip.
    PassATile<R> g;
    bool preloaded = false;
    if (tr.first < tr.end) { load_tile(tr.first, g); preloaded = true; }

    const PcgState prev = L.st[k & 1];
    const float rz_new = (float)fold_partials_256(L.part_rz, nparts_prev, s_red);
    const float rr = (float)fold_partials_256(L.part_rr, nparts_prev, s_red);
    // the reference's loop test: while (residc > tol && ki < iters)   (ref .cu:1131)
    const bool active = (prev.stopped == 0) && (rr > tol);
    if (!active) {
        if (blockIdx.x == 0 && tid == 0) { PcgState n = prev; n.stopped = 1; L.st[(k + 1) & 1] = n; }
        return;
    }
    const float beta = first ? 0.f : rz_new / prev.rz;
    if (blockIdx.x == 0 && tid == 0) {
        PcgState n; n.rz = rz_new; n.stopped = 0; n.iters = prev.iters + 1; n.pad = 0;
        L.st[(k + 1) & 1] = n;
    }
    double acc = 0.;

    for (int t = tr.first; t < tr.end; t += tr.step) {
        if (!(preloaded && t == tr.first)) load_tile(t, g);
        const int tx0 = (t % tiles_x) * kTileX, ty0 = (t / tiles_x) * TY;
        const int x = tx0 + lx * 4;
        // Phase 2: p_new = M^-1 r + beta p for the tile and its one-pixel halo, into LDS
        float npu[R][4], npv[R][4];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int y = ty0 + ly + kTileY * q;
            const bool rowok = (y < h) && (x < w);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = rowok && (x + e) < w;
                npu[q][e] = ok ? direction(g.ru[q][e], g.pu[q][e], g.a1[q][e], beta, first) : 0.f;
                npv[q][e] = ok ? direction(g.rv[q][e], g.pv[q][e], g.a4[q][e], beta, first) : 0.f;
            }
            st4(&s_pu[(ly + kTileY * q + 1) * kLRow + kLInt + lx * 4], *(float4 *)npu[q]);
            st4(&s_pv[(ly + kTileY * q + 1) * kLRow + kLInt + lx * 4], *(float4 *)npv[q]);
        }
        if (tid < 64) {
            const int hy = (tid < 32) ? ty0 - 1 : ty0 + TY;
            const int hx = tx0 + (tid & 31) * 4;
            const int lrow = (tid < 32) ? 0 : TY + 1;
            const bool in = hy >= 0 && hy < h && hx < w;
            float hu[4], hv[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = in && (hx + e) < w;
                hu[e] = ok ? direction(g.hr[0][e], g.hr[4][e], g.hr[2][e], beta, first) : 0.f;
                hv[e] = ok ? direction(g.hr[1][e], g.hr[5][e], g.hr[3][e], beta, first) : 0.f;
            }
            st4(&s_pu[lrow * kLRow + kLInt + (tid & 31) * 4], *(float4 *)hu);
            st4(&s_pv[lrow * kLRow + kLInt + (tid & 31) * 4], *(float4 *)hv);
        } else if (tid < 64 + 2 * TY) {
            const int side = (tid - 64) / TY, row = (tid - 64) % TY;
            const int hy = ty0 + row;
            const int hx = side ? tx0 + kTileX : tx0 - 1;
            const bool in = hy < h && hx >= 0 && hx < w;
            const float hu = in ? direction(g.hs[0], g.hs[4], g.hs[2], beta, first) : 0.f;
            const float hv = in ? direction(g.hs[1], g.hs[5], g.hs[3], beta, first) : 0.f;
            const int lcol = side ? kLInt + kTileX : kLInt - 1;
            s_pu[(row + 1) * kLRow + lcol] = hu;
            s_pv[(row + 1) * kLRow + lcol] = hv;
        }
        __syncthreads();
        // Phase 3: q = A p_new.  Row entries in the reference's storage order -- south, west, block, east, north --
        // with the merged border weights of ref .cu:929-1001
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int y = ty0 + ly + kTileY * q;
            const int lrow = ly + kTileY * q;
            if ((y < h) && (x < w)) {
                float su[4], sv[4], nu[4], nv[4];
                *(float4 *)su = ld4(&s_pu[lrow * kLRow + kLInt + lx * 4]);
                *(float4 *)sv = ld4(&s_pv[lrow * kLRow + kLInt + lx * 4]);
                *(float4 *)nu = ld4(&s_pu[(lrow + 2) * kLRow + kLInt + lx * 4]);
                *(float4 *)nv = ld4(&s_pv[(lrow + 2) * kLRow + kLInt + lx * 4]);
                const float uwest = s_pu[(lrow + 1) * kLRow + kLInt + lx * 4 - 1];
                const float vwest = s_pv[(lrow + 1) * kLRow + kLInt + lx * 4 - 1];
                const float ueast = s_pu[(lrow + 1) * kLRow + kLInt + lx * 4 + 4];
                const float veast = s_pv[(lrow + 1) * kLRow + kLInt + lx * 4 + 4];
                float qu[4], qv[4];
                float rowdot = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = x + e;
                    const float pwu = (e == 0) ? uwest : npu[q][(e + 3) & 3], pwv = (e == 0) ? vwest : npv[q][(e + 3) & 3];
                    const float peu = (e == 3) ? ueast : npu[q][(e + 1) & 3], pev = (e == 3) ? veast : npv[q][(e + 1) & 3];
                    const float a5 = (e == 0) ? g.wxw[q] : g.wxc[q][(e + 3) & 3];
                    const float wS = (y == h - 1) ? g.wys[q][e] + g.wyc[q][e] : g.wys[q][e];
                    const float wW = (i == w - 1) ? a5 + g.wxc[q][e] : a5;
                    const float wE = (i == 0) ? g.wxc[q][e] + g.wxc[q][e] : g.wxc[q][e];
                    const float wN = (y == 0) ? g.wyc[q][e] + g.wyc[q][e] : g.wyc[q][e];
                    float sumu = 0.f, sumv = 0.f;
                    if (y > 0) { sumu += wS * su[e]; sumv += wS * sv[e]; }
                    if (i > 0) { sumu += wW * pwu; sumv += wW * pwv; }
                    sumu += g.a1[q][e] * npu[q][e]; sumv += g.a2[q][e] * npu[q][e];
                    sumu += g.a2[q][e] * npv[q][e]; sumv += g.a4[q][e] * npv[q][e];
                    if (i < w - 1) { sumu += wE * peu; sumv += wE * pev; }
                    if (y < h - 1) { sumu += wN * nu[e]; sumv += wN * nv[e]; }
                    qu[e] = sumu; qv[e] = sumv;
                    if (i < w) { rowdot += npu[q][e] * sumu; rowdot += npv[q][e] * sumv; }
                }
                const size_t o = (size_t)y * pitch + x;
                st4(pout_u + o, *(float4 *)npu[q]);
                st4(pout_v + o, *(float4 *)npv[q]);
                st4_if(L.qu + o, *(float4 *)qu, L.nt_hints & 16);
                st4_if(L.qv + o, *(float4 *)qv, L.nt_hints & 16);
                acc += (double)rowdot;
            }
        }
        __syncthreads();
    }
    const double tot = block_sum_256(acc, s_red);
    if (tid == 0) L.part_pq[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------
// Pass A, marching form with an LDS row ring (OCTANE_TUNE_PASS_A=3).
//
// Same decomposition as k_pcg_pass_a_march (1024-pixel strips, contiguous runs of rows, equal shares), but
// the rolling window of p_new rows lives in a 4-slot LDS ring instead of registers: row y+1 is computed
// and written to the ring one step before q(y) needs it, all horizontal and vertical neighbours are LDS
// reads, and the operand loads of row y+2 are in flight while q(y) is evaluated.  One barrier per row.
// Nothing is fetched twice inside a run; re-fetch is 2 pixels per row (strip ends) + 2 reduced rows per
// run.
// ---------------------------------------------------------------------------------------------
constexpr int kMarchW = 1024;
constexpr int kRingRow = kMarchW + 8;     // [3 pad][west px][1024 interior][east px][3 pad]

struct RawRow { float4 ru, rv, pu, pv, a1, a4; float s[6]; };   // s: the strip-end pixel's r,p,a (2 lanes only)

template <bool UNITW>
__global__ __launch_bounds__(256, 3) void k_pcg_pass_a_ring(LevelPtrs L, int k, int nparts_prev, float tol)
{
    __shared__ double s_red[8];
    __shared__ __attribute__((aligned(16))) float s_ru[4][kRingRow];
    __shared__ __attribute__((aligned(16))) float s_rv[4][kRingRow];
    __shared__ float s_wxe[2][4];        // [row parity][wave]: wx of the wave's last pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const float rz_new = (float)fold_partials_256(L.part_rz, nparts_prev, s_red);
    const float rr = (float)fold_partials_256(L.part_rr, nparts_prev, s_red);
    const PcgState prev = L.st[k & 1];
    const bool active = (prev.stopped == 0) && (rr > tol);          // ref .cu:1131
    if (!active) {
        if (blockIdx.x == 0 && tid == 0) { PcgState n = prev; n.stopped = 1; L.st[(k + 1) & 1] = n; }
        return;
    }
    const bool first = (k == 0);
    const float beta = first ? 0.f : rz_new / prev.rz;
    if (blockIdx.x == 0 && tid == 0) {
        PcgState n; n.rz = rz_new; n.stopped = 0; n.iters = prev.iters + 1; n.pad = 0;
        L.st[(k + 1) & 1] = n;
    }

    const int w = L.w, h = L.h, pitch = L.pitch;
    const int strips = (w + kMarchW - 1) / kMarchW;
    const long total_rows = (long)strips * h;
    const float *__restrict__ pin_u = L.pu[k & 1];
    const float *__restrict__ pin_v = L.pv[k & 1];
    float *__restrict__ pout_u = L.pu[(k + 1) & 1];
    float *__restrict__ pout_v = L.pv[(k + 1) & 1];
    double acc = 0.;

    long vr = total_rows * blockIdx.x / gridDim.x;
    const long vr_end = total_rows * (blockIdx.x + 1) / gridDim.x;
    while (vr < vr_end) {
        const int sidx = (int)(vr / h);
        const int y0 = (int)(vr - (long)sidx * h);
        const int y1 = (int)min((long)h, y0 + (vr_end - vr));
        vr += y1 - y0;
        const int x0 = sidx * kMarchW;
        const int x = x0 + tid * 4;
        const bool colok = x < w;
        const bool west_side = (tid == 0) && (x0 > 0);
        const bool east_side = (tid == 255) && (x0 + kMarchW < w);
        const bool side = west_side || east_side;
        const int xs = west_side ? x0 - 1 : x0 + kMarchW;

        // issue the operand loads a row needs for p_new
        auto issue = [&](int yy, RawRow &r) {
            const size_t o = (size_t)yy * pitch + x;
            if (colok) {
                r.ru = ld4(L.ru + o); r.rv = ld4(L.rv + o); r.a1 = ld4(L.a1 + o); r.a4 = ld4(L.a4 + o);
                if (!first) { r.pu = ld4(pin_u + o); r.pv = ld4(pin_v + o); }
            }
            if (side) {
                const size_t so = (size_t)yy * pitch + xs;
                r.s[0] = L.ru[so]; r.s[1] = L.rv[so]; r.s[2] = L.a1[so]; r.s[3] = L.a4[so];
                if (!first) { r.s[4] = pin_u[so]; r.s[5] = pin_v[so]; }
            }
        };
        // turn them into p_new and put the row into its ring slot
        auto deposit = [&](int yy, const RawRow &r) {
            const int slot = yy & 3;
            float ru[4], rv[4], pu[4] = {0, 0, 0, 0}, pv[4] = {0, 0, 0, 0}, a1[4], a4[4], du[4], dv[4];
            *(float4 *)ru = r.ru; *(float4 *)rv = r.rv; *(float4 *)a1 = r.a1; *(float4 *)a4 = r.a4;
            if (!first) { *(float4 *)pu = r.pu; *(float4 *)pv = r.pv; }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = colok && (x + e) < w;
                du[e] = ok ? direction(ru[e], pu[e], a1[e], beta, first) : 0.f;
                dv[e] = ok ? direction(rv[e], pv[e], a4[e], beta, first) : 0.f;
            }
            st4(&s_ru[slot][4 + tid * 4], *(float4 *)du);
            st4(&s_rv[slot][4 + tid * 4], *(float4 *)dv);
            if (side) {
                const float su = direction(r.s[0], first ? 0.f : r.s[4], r.s[2], beta, first);
                const float sv = direction(r.s[1], first ? 0.f : r.s[5], r.s[3], beta, first);
                s_ru[slot][west_side ? 3 : 4 + kMarchW] = su;
                s_rv[slot][west_side ? 3 : 4 + kMarchW] = sv;
            }
        };

        RawRow raw;
        float a1c[4] = {1, 1, 1, 1}, a4c[4] = {1, 1, 1, 1}, a2c[4] = {0, 0, 0, 0}, wxc[4] = {0, 0, 0, 0}, wyc[4] = {0, 0, 0, 0};
        float wym[4] = {0, 0, 0, 0};
        float4 a2n = make_float4(0, 0, 0, 0), wxn = a2n, wyn = a2n;
        float wx_side_c = 0.f, wx_side_n = 0.f;
        if (UNITW) {        // first GNC step: every neighbour weight is exactly -1, the planes are not read
#pragma unroll
            for (int e = 0; e < 4; e++) { wxc[e] = -1.f; wyc[e] = -1.f; wym[e] = -1.f; }
            wxn = make_float4(-1.f, -1.f, -1.f, -1.f); wyn = wxn;
            wx_side_c = -1.f; wx_side_n = -1.f;
        }

        __syncthreads();                             // a previous run may still be reading the ring
        if (y0 > 0) {
            issue(y0 - 1, raw);
            if (!UNITW && colok) *(float4 *)wym = ld4(L.wy + (size_t)(y0 - 1) * pitch + x);
            deposit(y0 - 1, raw);
        }
        {
            const size_t oc = (size_t)y0 * pitch + x;
            issue(y0, raw);
            if (colok) {
                *(float4 *)a2c = ld4_if(L.a2 + oc, L.nt_hints & 8);
                if (!UNITW) { *(float4 *)wxc = ld4(L.wx + oc); *(float4 *)wyc = ld4(L.wy + oc); }
            }
            if (!UNITW && west_side) wx_side_c = L.wx[oc - 1];
            deposit(y0, raw);
            *(float4 *)a1c = raw.a1; *(float4 *)a4c = raw.a4;
            if (lane == 63) s_wxe[y0 & 1][wave] = wxc[3];
        }
        if (y0 + 1 < h) issue(y0 + 1, raw);
        for (int y = y0; y < y1; ++y) {
            const bool have_next = (y + 1 < y1);
            float4 a1n = make_float4(1, 1, 1, 1), a4n = a1n;
            if (y + 1 < h) {
                deposit(y + 1, raw);                 // waits for the loads issued one step ago
                a1n = raw.a1; a4n = raw.a4;
            }
            if (have_next) {
                const size_t on = (size_t)(y + 1) * pitch + x;
                if (colok) {
                    a2n = ld4_if(L.a2 + on, L.nt_hints & 8);
                    if (!UNITW) { wxn = ld4(L.wx + on); wyn = ld4(L.wy + on); }
                }
                if (!UNITW && west_side) wx_side_n = L.wx[on - 1];
                if (y + 2 < h) issue(y + 2, raw);    // in flight while q(y) is evaluated
            }
            __syncthreads();
            const int sm = (y - 1) & 3, sc = y & 3, sn = (y + 1) & 3;
            float wxw = __shfl_up(wxc[3], 1, 64);
            if (lane == 0) wxw = (wave > 0) ? s_wxe[y & 1][wave - 1] : wx_side_c;
            if (colok) {
                float pmu[4], pmv[4], pcu[4], pcv[4], pnu[4], pnv[4];
                *(float4 *)pmu = ld4(&s_ru[sm][4 + tid * 4]); *(float4 *)pmv = ld4(&s_rv[sm][4 + tid * 4]);
                *(float4 *)pcu = ld4(&s_ru[sc][4 + tid * 4]); *(float4 *)pcv = ld4(&s_rv[sc][4 + tid * 4]);
                *(float4 *)pnu = ld4(&s_ru[sn][4 + tid * 4]); *(float4 *)pnv = ld4(&s_rv[sn][4 + tid * 4]);
                const float uwest = s_ru[sc][3 + tid * 4], vwest = s_rv[sc][3 + tid * 4];
                const float ueast = s_ru[sc][8 + tid * 4], veast = s_rv[sc][8 + tid * 4];
                float qu[4], qv[4];
                float rowdot = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = x + e;
                    const float pwu = (e == 0) ? uwest : pcu[(e + 3) & 3], pwv = (e == 0) ? vwest : pcv[(e + 3) & 3];
                    const float peu = (e == 3) ? ueast : pcu[(e + 1) & 3], pev = (e == 3) ? veast : pcv[(e + 1) & 3];
                    const float a5 = (e == 0) ? wxw : wxc[(e + 3) & 3];
                    const float wS = (y == h - 1) ? wym[e] + wyc[e] : wym[e];
                    const float wW = (i == w - 1) ? a5 + wxc[e] : a5;
                    const float wE = (i == 0) ? wxc[e] + wxc[e] : wxc[e];
                    const float wN = (y == 0) ? wyc[e] + wyc[e] : wyc[e];
                    float sumu = 0.f, sumv = 0.f;
                    if (y > 0) { sumu += wS * pmu[e]; sumv += wS * pmv[e]; }
                    if (i > 0) { sumu += wW * pwu; sumv += wW * pwv; }
                    sumu += a1c[e] * pcu[e]; sumv += a2c[e] * pcu[e];
                    sumu += a2c[e] * pcv[e]; sumv += a4c[e] * pcv[e];
                    if (i < w - 1) { sumu += wE * peu; sumv += wE * pev; }
                    if (y < h - 1) { sumu += wN * pnu[e]; sumv += wN * pnv[e]; }
                    qu[e] = sumu; qv[e] = sumv;
                    if (i < w) { rowdot += pcu[e] * sumu; rowdot += pcv[e] * sumv; }
                }
                const size_t o = (size_t)y * pitch + x;
                st4(pout_u + o, *(float4 *)pcu);
                st4(pout_v + o, *(float4 *)pcv);
                st4_if(L.qu + o, *(float4 *)qu, L.nt_hints & 16);
                st4_if(L.qv + o, *(float4 *)qv, L.nt_hints & 16);
                acc += (double)rowdot;
            }
            // roll to the next row
#pragma unroll
            for (int e = 0; e < 4; e++) wym[e] = wyc[e];
            *(float4 *)a1c = a1n; *(float4 *)a4c = a4n;
            *(float4 *)a2c = a2n; *(float4 *)wxc = wxn; *(float4 *)wyc = wyn;
            wx_side_c = wx_side_n;
            if (have_next && lane == 63) s_wxe[(y + 1) & 1][wave] = wxc[3];
        }
    }
    const double tot = block_sum_256(acc, s_red);
    if (tid == 0) L.part_pq[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused PCG iteration: ONE kernel per iteration instead of pass A + pass B.
//
//   G(k):  alpha_{k-1} = (r.z)_{k-1} / (p.q)_{k-1};   x += alpha_{k-1} p_{k-1};   r_k = r_{k-1} - alpha_{k-1} q_{k-1}     [k >= 1]
//          beta_k = (r.z)_k / (r.z)_{k-1};   p_k = M^-1 r_k + beta_k p_{k-1};   q_k = A p_k
//          partials of p_k.q_k, q_k.z_k, q_k.M^-1 q_k, r_k.q_k, q_k.q_k and (directly) r_k.z_k, r_k.r_k
//
// Pass B's only reason to be a kernel of its own is that beta_k needs (r.z)_k, a sum over the residual pass B has just
// formed.  But r_k = r_{k-1} - alpha q_{k-1} with everything on the right known before pass B runs, so
//          (r.z)_k = (r.z)_{k-1} - 2 alpha (q.z)_{k-1} + alpha^2 (q.M^-1 q)_{k-1}          (M^-1 is diagonal)
//          (r.r)_k = (r.r)_{k-1} - 2 alpha (r.q)_{k-1} + alpha^2 (q.q)_{k-1}
// are plain algebra on sums the previous kernel can form while it has q in registers -- no symmetry or orthogonality of
// the operator is assumed (the single-reduction CG variants that do assume it diverge here, EXPERIMENTS.md 8).  The base
// values (r.z)_{k-1}, (r.r)_{k-1} are the DIRECT sums the previous kernel formed over the residual it wrote, so nothing
// is chained: the recurrence value differs from a direct sum over r_k only by the rounding of r_k's elements (~1e-10
// relative), far below what the reference's own float atomics do to the same numbers.  Element by element x, r, p and q
// are computed exactly as the two-pass kernels compute them; alpha, beta and the stop test are formed as there from
// floats of those sums (ref .cu:1131-1178).
//
// Traffic: reads r q p x (8 B each) + a1 a2 a4 wx wy (20 B), writes r p q x = 84 B/pixel/iteration instead of 104, and
// half as many kernel boundaries -- which is what the small, latency-bound levels are made of.  r and q are
// double-buffered (rb, qb): a workgroup recomputes r_k and p_k on its tile's one-pixel ring from the OLD values of the
// neighbouring tiles while those are being overwritten.  The last iteration's x update is left to k_flow_update_fused.
// Row bands: like pass A, plus one sync per iteration instead of two; a band keeps its own copies of r_k and p_k on the
// neighbouring bands' edge rows and reads only q_{k-1} of those rows from the neighbour.
// ---------------------------------------------------------------------------------------------------------------------
template <int R, bool UNITW>
__global__ __launch_bounds__(256) void k_pcg_fused(LevelPtrs L, int k, int nparts_prev, float tol)
{
    constexpr int TY = kTileY * R, TX = kTileX, LROW = TX + 8, HL = TX / 4;
    __shared__ __attribute__((aligned(16))) float s_pu[(TY + 2) * LROW];
    __shared__ __attribute__((aligned(16))) float s_pv[(TY + 2) * LROW];
    __shared__ double s_red[4 * kPartKinds];
    const int tid = threadIdx.x;
    const bool first = (k == 0);

    const PcgState prev = L.st[k & 1];
    if (prev.stopped) {                              // uniform: the loop ended in an earlier launch
        if (blockIdx.x == 0 && tid == 0) L.st[(k + 1) & 1] = prev;
        return;
    }
    // Every launch reads the sums of the previous one and writes its own, and workgroups of one launch do not all run
    // at the same time: the partials are double-buffered by iteration parity (the assembly writes the "-1" block).
    const int pin_off = ((k + 1) & 1) * kPartBlock, pout_off = (k & 1) * kPartBlock;
    float alpha = 0.f, nalpha = 0.f, beta = 0.f, rz_new, rr;
    if (first) {                                     // r_0 = rhs: the assembly's direct sums
        double t[2];
        fold_band_partials_multi_256<2>(L.band_parts, pin_off + kPartRz, kMaxParts, nparts_prev, L.nbands, s_red, t);
        rz_new = (float)t[0]; rr = (float)t[1];
    } else {
        double t[kPartKinds];                        // kinds in block order: rz rr pq qz qmq rq qq
        fold_band_partials_multi_256<kPartKinds>(L.band_parts, pin_off, kMaxParts, nparts_prev, L.nbands, s_red, t);
        const double rzd = t[0], rrd = t[1], pq = t[2], qz = t[3], qmq = t[4], rq = t[5], qq = t[6];
        alpha = prev.rz / (float)pq;                 // ref .cu:1169
        nalpha = (float)(-1. * (double)alpha);       // ref .cu:1174
        const double a = (double)alpha;
        rz_new = (float)(rzd - 2. * a * qz + a * a * qmq);
        rr = (float)(rrd - 2. * a * rq + a * a * qq);
        beta = rz_new / prev.rz;
    }
    const bool active = rr > tol || OCT_STOP_HELD_OPEN(tol);   // ref .cu:1131 (held open only by the diagnostic library's solo-band timing)
    if (blockIdx.x == 0 && tid == 0) {
        PcgState n; n.rz = rz_new; n.stopped = active ? 0 : 1; n.iters = prev.iters + (active ? 1 : 0); n.pad = 0;
        L.st[(k + 1) & 1] = n;
        if (!first) L.alpha[(k - 1) & 1] = alpha;
    }
    if (first && !active) return;                    // nothing ran, nothing to update

    const int w = L.w, h = L.h, pitch = L.pitch;
    const int by0 = L.y0, by1 = L.y1;
    const int tiles_x = (w + TX - 1) / TX, tiles_y = (by1 - by0 + TY - 1) / TY;
    const int ntiles = tiles_x * tiles_y;
    const int lx = tid & 31, ly = tid >> 5;
    const int ko = (k + 1) & 1, kn = k & 1;          // old = k-1 (same parity as k+1), new = k
    const float *__restrict__ rin_u = first ? L.rb_u[0] : L.rb_u[ko];
    const float *__restrict__ rin_v = first ? L.rb_v[0] : L.rb_v[ko];
    float *__restrict__ rout_u = L.rb_u[kn];
    float *__restrict__ rout_v = L.rb_v[kn];
    const float *__restrict__ qin_u = L.qb_u[ko];
    const float *__restrict__ qin_v = L.qb_v[ko];
    float *__restrict__ qout_u = L.qb_u[kn];
    float *__restrict__ qout_v = L.qb_v[kn];
    const float *__restrict__ pin_u = L.pf_u[(k + 2) % 3];          // p_{k-1}
    const float *__restrict__ pin_v = L.pf_v[(k + 2) % 3];
    const float *__restrict__ pin2_u = L.pf_u[(k + 1) % 3];         // p_{k-2}
    const float *__restrict__ pin2_v = L.pf_v[(k + 1) % 3];
    float *__restrict__ pout_u = L.pf_u[k % 3];
    float *__restrict__ pout_v = L.pf_v[k % 3];
    // x handling.  Immediate: x += alpha_{k-1} p_{k-1} in every launch.  Deferred (default): odd launches leave their update
    // pending and even launches apply two in the reference's order, x <- alpha_{k-1} p_{k-1} + (alpha_{k-2} p_{k-2} + x),
    // reading p_{k-2} from the third p buffer: same operations, same roundings, but x moves every second launch only
    // (12 instead of 16 B/pixel/iteration).  A launch that ends the loop applies whatever is pending.
    const bool defer = L.defer_x != 0;
    const bool x_two = defer && !first && (k & 1) == 0;               // k >= 2, even: two updates
    const bool x_one = !first && (!defer || ((k & 1) == 1 && !active));   // one update: immediate mode, or an odd launch that stops
    const bool x_read = x_two ? (k > 2) : (defer ? (k >= 3) : (k > 1));  // x holds something already
    const float alpha2 = x_two ? L.alpha[(k - 2) & 1] : 0.f;          // alpha_{k-2}, stored by the previous launch
    double acc_pq = 0., acc_qz = 0., acc_qmq = 0., acc_rq = 0., acc_qq = 0., acc_rz = 0., acc_rr = 0.;

    const ItemRange tr = item_range_walk(ntiles, L.xcd_bands);
    for (int t = tr.first; t < tr.end; t += tr.step) {
        const int tx0 = (t % tiles_x) * TX, ty0 = by0 + (t / tiles_x) * TY;
        float a1[R][4], a4[R][4], a2[R][4], wxc[R][4], wyc[R][4], wys[R][4], npu[R][4], npv[R][4], nru[R][4], nrv[R][4];
        float wxw[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int x = tx0 + lx * 4, y = ty0 + ly + kTileY * q;
            const int lrow1 = ly + kTileY * q + 1, lcol = kLInt + lx * 4;
            const bool rowok = (y < by1) && (x < w);
            const size_t o = (size_t)y * pitch + x;
            wxw[q] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) { a1[q][e] = 1.f; a4[q][e] = 1.f; a2[q][e] = 0.f; wxc[q][e] = 0.f; wyc[q][e] = 0.f; wys[q][e] = 0.f;
                                          npu[q][e] = 0.f; npv[q][e] = 0.f; nru[q][e] = 0.f; nrv[q][e] = 0.f; }
            if (rowok) {
                float ru[4], rv[4];
                *(float4 *)ru = ld4(rin_u + o);
                *(float4 *)rv = ld4(rin_v + o);
                *(float4 *)a1[q] = ld4(L.a1 + o);
                *(float4 *)a4[q] = ld4(L.a4 + o);
                *(float4 *)a2[q] = ld4_if(L.a2 + o, L.nt_hints & 8);
                if (UNITW) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { wxc[q][e] = -1.f; wyc[q][e] = -1.f; wys[q][e] = -1.f; }
                    wxw[q] = -1.f;
                } else {
                    *(float4 *)wxc[q] = ld4(L.wx + o);
                    *(float4 *)wyc[q] = ld4(L.wy + o);
                    if (y > 0) *(float4 *)wys[q] = ld4(L.wy + o - pitch);
                    if (x > 0) wxw[q] = L.wx[o - 1];
                }
                float pu[4] = {0, 0, 0, 0}, pv[4] = {0, 0, 0, 0};
                if (!first) {
                    float qu[4], qv[4];
                    *(float4 *)pu = ld4(pin_u + o); *(float4 *)pv = ld4(pin_v + o);
                    *(float4 *)qu = ld4(qin_u + o); *(float4 *)qv = ld4(qin_v + o);
                    if (x_two || x_one) {
                        float xu[4] = {0, 0, 0, 0}, xv[4] = {0, 0, 0, 0};
                        if (x_read) { *(float4 *)xu = ld4_if(L.xu + o, L.nt_hints & 1); *(float4 *)xv = ld4_if(L.xv + o, L.nt_hints & 1); }
                        if (x_two) {                               // the previous launch's update first
                            float ou[4], ov[4];
                            *(float4 *)ou = ld4(pin2_u + o); *(float4 *)ov = ld4(pin2_v + o);
#pragma unroll
                            for (int e = 0; e < 4; e++) { xu[e] = alpha2 * ou[e] + xu[e]; xv[e] = alpha2 * ov[e] + xv[e]; }
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            xu[e] = alpha * pu[e] + xu[e];         // jVecPVec(p0,x0,x0,alphak), ref .cu:1172
                            xv[e] = alpha * pv[e] + xv[e];
                        }
                        st4_if(L.xu + o, *(float4 *)xu, L.nt_hints & 1);
                        st4_if(L.xv + o, *(float4 *)xv, L.nt_hints & 1);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        ru[e] = nalpha * qu[e] + ru[e];            // jVecPVec(dummyvec,bcu,rk,-alphak), ref .cu:1174
                        rv[e] = nalpha * qv[e] + rv[e];
                    }
                    if (active) { st4(rout_u + o, *(float4 *)ru); st4(rout_v + o, *(float4 *)rv); }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool ok = (x + e) < w;
                    nru[q][e] = ok ? ru[e] : 0.f; nrv[q][e] = ok ? rv[e] : 0.f;
                    npu[q][e] = ok ? direction(ru[e], pu[e], a1[q][e], beta, first) : 0.f;
                    npv[q][e] = ok ? direction(rv[e], pv[e], a4[q][e], beta, first) : 0.f;
                }
            }
            st4(&s_pu[lrow1 * LROW + lcol], *(float4 *)npu[q]);
            st4(&s_pv[lrow1 * LROW + lcol], *(float4 *)npv[q]);
        }
        if (!active) continue;                           // the loop has ended: only x needed its last update (uniform)
        // one-pixel ring of p_k, recomputed from the old r, q, p of the neighbouring pixels
        if (tid < 2 * HL) {                              // rows above and below the tile
            const int hy = (tid < HL) ? ty0 - 1 : ty0 + TY;
            const int hx = tx0 + (tid % HL) * 4;
            const int lrow = (tid < HL) ? 0 : TY + 1;
            float hu[4] = {0, 0, 0, 0}, hv[4] = {0, 0, 0, 0};
            if (hy >= 0 && hy < h && hx < w) {
                const size_t ho = (size_t)hy * pitch + hx;
                const bool outside = (hy < by0 || hy >= by1);          // a neighbouring band's row
                float r0[4], r1[4], d0[4], d1[4], p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
                *(float4 *)r0 = ld4(rin_u + ho); *(float4 *)r1 = ld4(rin_v + ho);
                *(float4 *)d0 = ld4(L.a1 + ho); *(float4 *)d1 = ld4(L.a4 + ho);
                if (!first) {
                    float q0[4], q1[4];
                    // q of a neighbouring band's row comes from that band's plane; r and p of it are this band's own copies
                    const float *hqu = (hy < by0) ? L.qup_u[ko] : (hy >= by1) ? L.qdn_u[ko] : qin_u;
                    const float *hqv = (hy < by0) ? L.qup_v[ko] : (hy >= by1) ? L.qdn_v[ko] : qin_v;
                    *(float4 *)q0 = ld4(hqu + ho); *(float4 *)q1 = ld4(hqv + ho);
                    *(float4 *)p0 = ld4(pin_u + ho); *(float4 *)p1 = ld4(pin_v + ho);
#pragma unroll
                    for (int e = 0; e < 4; e++) { r0[e] = nalpha * q0[e] + r0[e]; r1[e] = nalpha * q1[e] + r1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool ok = (hx + e) < w;
                    hu[e] = ok ? direction(r0[e], p0[e], d0[e], beta, first) : 0.f;
                    hv[e] = ok ? direction(r1[e], p1[e], d1[e], beta, first) : 0.f;
                }
                if (outside) {                           // keep this band's copies of r_k and p_k on that row current
                    if (!first) { st4(rout_u + ho, *(float4 *)r0); st4(rout_v + ho, *(float4 *)r1); }
                    st4(pout_u + ho, *(float4 *)hu); st4(pout_v + ho, *(float4 *)hv);
                }
            }
            st4(&s_pu[lrow * LROW + kLInt + (tid % HL) * 4], *(float4 *)hu);
            st4(&s_pv[lrow * LROW + kLInt + (tid % HL) * 4], *(float4 *)hv);
        } else if (tid < 2 * HL + 2 * TY) {              // columns left and right of the tile
            const int side = (tid - 2 * HL) / TY, row = (tid - 2 * HL) % TY;
            const int hy = ty0 + row;
            const int hx = side ? tx0 + TX : tx0 - 1;
            float hu = 0.f, hv = 0.f;
            if (hy < by1 && hx >= 0 && hx < w) {
                const size_t ho = (size_t)hy * pitch + hx;
                float r0 = rin_u[ho], r1 = rin_v[ho], p0 = 0.f, p1 = 0.f;
                if (!first) {
                    r0 = nalpha * qin_u[ho] + r0; r1 = nalpha * qin_v[ho] + r1;
                    p0 = pin_u[ho]; p1 = pin_v[ho];
                }
                hu = direction(r0, p0, L.a1[ho], beta, first);
                hv = direction(r1, p1, L.a4[ho], beta, first);
            }
            const int lcol = side ? kLInt + TX : kLInt - 1;
            s_pu[(row + 1) * LROW + lcol] = hu;
            s_pv[(row + 1) * LROW + lcol] = hv;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int x = tx0 + lx * 4, y = ty0 + ly + kTileY * q;
            const int lrow = ly + kTileY * q, lc = kLInt + lx * 4;
            if ((y < by1) && (x < w)) {
                float su[4], sv[4], nu[4], nv[4];
                *(float4 *)su = ld4(&s_pu[lrow * LROW + lc]);
                *(float4 *)sv = ld4(&s_pv[lrow * LROW + lc]);
                *(float4 *)nu = ld4(&s_pu[(lrow + 2) * LROW + lc]);
                *(float4 *)nv = ld4(&s_pv[(lrow + 2) * LROW + lc]);
                const float uwest = s_pu[(lrow + 1) * LROW + lc - 1];
                const float vwest = s_pv[(lrow + 1) * LROW + lc - 1];
                const float ueast = s_pu[(lrow + 1) * LROW + lc + 4];
                const float veast = s_pv[(lrow + 1) * LROW + lc + 4];
                float qu[4], qv[4];
                float d_pq = 0.f, d_qz = 0.f, d_qmq = 0.f, d_rq = 0.f, d_qq = 0.f, d_rz = 0.f, d_rr = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = x + e;
                    const float pwu = (e == 0) ? uwest : npu[q][(e + 3) & 3], pwv = (e == 0) ? vwest : npv[q][(e + 3) & 3];
                    const float peu = (e == 3) ? ueast : npu[q][(e + 1) & 3], pev = (e == 3) ? veast : npv[q][(e + 1) & 3];
                    const float a5 = (e == 0) ? wxw[q] : wxc[q][(e + 3) & 3];
                    const float wS = (y == h - 1) ? wys[q][e] + wyc[q][e] : wys[q][e];
                    const float wW = (i == w - 1) ? a5 + wxc[q][e] : a5;
                    const float wE = (i == 0) ? wxc[q][e] + wxc[q][e] : wxc[q][e];
                    const float wN = (y == 0) ? wyc[q][e] + wyc[q][e] : wyc[q][e];
                    float sumu = 0.f, sumv = 0.f;
                    if (y > 0) { sumu += wS * su[e]; sumv += wS * sv[e]; }
                    if (i > 0) { sumu += wW * pwu; sumv += wW * pwv; }
                    sumu += a1[q][e] * npu[q][e]; sumv += a2[q][e] * npu[q][e];
                    sumu += a2[q][e] * npv[q][e]; sumv += a4[q][e] * npv[q][e];
                    if (i < w - 1) { sumu += wE * peu; sumv += wE * pev; }
                    if (y < h - 1) { sumu += wN * nu[e]; sumv += wN * nv[e]; }
                    qu[e] = sumu; qv[e] = sumv;
                    if (i < w) {
                        const float iu = rcp_exact(a1[q][e]), iv = rcp_exact(a4[q][e]);
                        const float zu = iu * nru[q][e], zv = iv * nrv[q][e];
                        d_pq += npu[q][e] * sumu; d_pq += npv[q][e] * sumv;
                        d_qz += sumu * zu; d_qz += sumv * zv;
                        d_qmq += sumu * (iu * sumu); d_qmq += sumv * (iv * sumv);
                        d_rq += nru[q][e] * sumu; d_rq += nrv[q][e] * sumv;
                        d_qq += sumu * sumu; d_qq += sumv * sumv;
                        d_rz += nru[q][e] * zu; d_rz += nrv[q][e] * zv;
                        d_rr += nru[q][e] * nru[q][e]; d_rr += nrv[q][e] * nrv[q][e];
                    }
                }
                const size_t o = (size_t)y * pitch + x;
                st4(pout_u + o, *(float4 *)npu[q]);
                st4(pout_v + o, *(float4 *)npv[q]);
                st4(qout_u + o, *(float4 *)qu);
                st4(qout_v + o, *(float4 *)qv);
                acc_pq += (double)d_pq; acc_qz += (double)d_qz; acc_qmq += (double)d_qmq; acc_rq += (double)d_rq;
                acc_qq += (double)d_qq; acc_rz += (double)d_rz; acc_rr += (double)d_rr;
            }
        }
        __syncthreads();
    }
    if (!active) return;
    double *own = L.part_own + pout_off;                 // this launch's block: kinds are kMaxParts apart
    const double accs[kPartKinds] = {acc_rz, acc_rr, acc_pq, acc_qz, acc_qmq, acc_rq, acc_qq};
    double tot[kPartKinds];
    block_sum_multi_256<kPartKinds>(accs, s_red, tot);
    if (tid == 0) {
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) own[j * kMaxParts + blockIdx.x] = tot[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused PCG iteration WITHOUT a stored q (levels and row bands of 3 Mpixel and more).
//
// q_{k-1} = A p_{k-1} is written by one launch only to be read once by the next (r_k = r_{k-1} - alpha q_{k-1}).  p_{k-1}
// is stored anyway, so the next launch can form q_{k-1} again -- same inputs, same operations, same bits -- and the 16
// B/pixel of q traffic (of 80) disappear: 64 B/pixel/iteration.  The price is a second stencil per pixel (free, the kernel
// is bandwidth-bound) and wider halos: a workgroup needs r_k and p_k on its tile's one-pixel ring, so it forms q_{k-1}
// there too, from p_{k-1} two pixels out.  The ring is handled as 100 more float4 groups of the same per-group routine
// (the row above, the row below, the group left and right of every row), not as scalar special cases:
//   phase 0   p_{k-1} of the tile + 2 rows / 4 columns around it -> LDS
//   phase 1   per group of the tile and its ring: q_{k-1} (stencil on that LDS tile), r_k, z_k, p_k -> second LDS tile;
//             tile groups also update x and store r_k, p_k
//   phase 2   tile groups: q_k (stencil on the second LDS tile) and the seven partial sums; q_k is NOT stored
// The second LDS tile has two buffers used alternately, so a tile costs two barriers, not three.  r_k and p_k leave with
// streaming stores (nothing of a launch this size is still cached when the next one reads it); the file is built with
// LLVM's max-ilp scheduling (Makefile).  What else was tried on this kernel, and what it did: EXPERIMENTS.md 8.
// Everything else (scalars from the previous launch's sums, double-buffered r / partials, triple-buffered p, deferred x)
// is k_pcg_fused's.  In row bands (BANDED) the two rows beyond a band edge are read from the neighbour's planes in place.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kQTY = 2 * kTileY;            // 16 tile rows
constexpr int kQCols = kTileX + 16;         // LDS row: 8 floats of margin either side of the 128 tile columns
constexpr int kQOff = 8;                    // LDS column of the tile's first pixel

struct QCoef { float a1[4], a2[4], a4[4], wx[4], wy[4], wys[4]; float wxw; };

// the 5-point operator on one float4 group at frame position (x0, y), from an LDS tile whose row `lrow` / column `lcol`
// hold the group's own pixels (same arithmetic, in the same order, as every other form of A p in this file)
__device__ __forceinline__ void stencil_group(const float *s_u, const float *s_v, int lrow, int lcol, int x0, int y, int w, int h,
                                              const QCoef &c, float (&qu)[4], float (&qv)[4])
{
    float cu[4], cv[4], su[4], sv[4], nu[4], nv[4];
    *(float4 *)cu = ld4(&s_u[lrow * kQCols + lcol]); *(float4 *)cv = ld4(&s_v[lrow * kQCols + lcol]);
    *(float4 *)su = ld4(&s_u[(lrow - 1) * kQCols + lcol]); *(float4 *)sv = ld4(&s_v[(lrow - 1) * kQCols + lcol]);
    *(float4 *)nu = ld4(&s_u[(lrow + 1) * kQCols + lcol]); *(float4 *)nv = ld4(&s_v[(lrow + 1) * kQCols + lcol]);
    const float uwest = s_u[lrow * kQCols + lcol - 1], vwest = s_v[lrow * kQCols + lcol - 1];
    const float ueast = s_u[lrow * kQCols + lcol + 4], veast = s_v[lrow * kQCols + lcol + 4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int i = x0 + e;
        const float pwu = (e == 0) ? uwest : cu[(e + 3) & 3], pwv = (e == 0) ? vwest : cv[(e + 3) & 3];
        const float peu = (e == 3) ? ueast : cu[(e + 1) & 3], pev = (e == 3) ? veast : cv[(e + 1) & 3];
        const float a5 = (e == 0) ? c.wxw : c.wx[(e + 3) & 3];
        const float wS = (y == h - 1) ? c.wys[e] + c.wy[e] : c.wys[e];
        const float wW = (i == w - 1) ? a5 + c.wx[e] : a5;
        const float wE = (i == 0) ? c.wx[e] + c.wx[e] : c.wx[e];
        const float wN = (y == 0) ? c.wy[e] + c.wy[e] : c.wy[e];
        float sumu = 0.f, sumv = 0.f;
        if (y > 0) { sumu += wS * su[e]; sumv += wS * sv[e]; }
        if (i > 0) { sumu += wW * pwu; sumv += wW * pwv; }
        sumu += c.a1[e] * cu[e]; sumv += c.a2[e] * cu[e];
        sumu += c.a2[e] * cv[e]; sumv += c.a4[e] * cv[e];
        if (i < w - 1) { sumu += wE * peu; sumv += wE * pev; }
        if (y < h - 1) { sumu += wN * nu[e]; sumv += wN * nv[e]; }
        qu[e] = sumu; qv[e] = sumv;
    }
}

template <bool UNITW, bool BANDED>
__global__ __launch_bounds__(256) void k_pcg_fused_q(LevelPtrs L, int k, int nparts_prev, float tol)
{
    constexpr int TY = kQTY, TX = kTileX;
    __shared__ __attribute__((aligned(16))) float s_ou[(TY + 4) * kQCols], s_ov[(TY + 4) * kQCols];   // p_{k-1}: rows ty0-2 .. ty0+TY+1
    // p_k: rows ty0-1 .. ty0+TY, two buffers used alternately -- a workgroup's fast waves may stage and compute the next tile
    // while its slow ones still read this one's p_k in phase 2, which saves the barrier at the end of a tile
    constexpr int NSZ = (TY + 2) * kQCols;
    __shared__ __attribute__((aligned(16))) float s_nu2[2 * NSZ], s_nv2[2 * NSZ];
    __shared__ double s_red[4 * kPartKinds];
    const int tid = threadIdx.x;
    const bool first = (k == 0);

    const PcgState prev = L.st[k & 1];
    if (prev.stopped) {
        if (blockIdx.x == 0 && tid == 0) L.st[(k + 1) & 1] = prev;
        return;
    }
    const int pin_off = ((k + 1) & 1) * kPartBlock, pout_off = (k & 1) * kPartBlock;
    float alpha = 0.f, nalpha = 0.f, beta = 0.f, rz_new, rr;
    if (first) {
        double t[2];
        fold_band_partials_multi_256<2>(L.band_parts, pin_off + kPartRz, kMaxParts, nparts_prev, L.nbands, s_red, t);
        rz_new = (float)t[0]; rr = (float)t[1];
    } else {
        double t[kPartKinds];
        fold_band_partials_multi_256<kPartKinds>(L.band_parts, pin_off, kMaxParts, nparts_prev, L.nbands, s_red, t);
        const double rzd = t[0], rrd = t[1], pq = t[2], qz = t[3], qmq = t[4], rq = t[5], qq = t[6];
        alpha = prev.rz / (float)pq;                 // ref .cu:1169
        nalpha = (float)(-1. * (double)alpha);       // ref .cu:1174
        const double a = (double)alpha;
        rz_new = (float)(rzd - 2. * a * qz + a * a * qmq);
        rr = (float)(rrd - 2. * a * rq + a * a * qq);
        beta = rz_new / prev.rz;
    }
    const bool active = rr > tol || OCT_STOP_HELD_OPEN(tol);   // ref .cu:1131 (held open only by the diagnostic library's solo-band timing)
    if (blockIdx.x == 0 && tid == 0) {
        PcgState n; n.rz = rz_new; n.stopped = active ? 0 : 1; n.iters = prev.iters + (active ? 1 : 0); n.pad = 0;
        L.st[(k + 1) & 1] = n;
        if (!first) L.alpha[(k - 1) & 1] = alpha;
    }
    if (first && !active) return;

    const int w = L.w, h = L.h, pitch = L.pitch;
    // Row bands (BANDED): the tiles cover the band's own rows [y0, y1); what a tile needs from the two rows beyond a band
    // edge -- r_{k-1} on the ring row, p_{k-1} on the ring row and the one after -- is read in place from the neighbouring
    // band's planes, which the previous launch completed (one phase boundary per iteration).  The operator on the ring
    // row is this band's own (the assembly covers one halo row), except wy of the row above the upper ring row.
    const int y0 = BANDED ? L.y0 : 0, y1 = BANDED ? L.y1 : h;
    const int tiles_x = (w + TX - 1) / TX, tiles_y = (y1 - y0 + TY - 1) / TY;
    const int ntiles = tiles_x * tiles_y;
    const int ko = (k + 1) & 1, kn = k & 1;
    const float *__restrict__ rin_u = first ? L.rb_u[0] : L.rb_u[ko];
    const float *__restrict__ rin_v = first ? L.rb_v[0] : L.rb_v[ko];
    float *__restrict__ rout_u = L.rb_u[kn];
    float *__restrict__ rout_v = L.rb_v[kn];
    const float *__restrict__ pin_u = L.pf_u[(k + 2) % 3];          // p_{k-1}
    const float *__restrict__ pin_v = L.pf_v[(k + 2) % 3];
    const float *__restrict__ pin2_u = L.pf_u[(k + 1) % 3];         // p_{k-2}
    const float *__restrict__ pin2_v = L.pf_v[(k + 1) % 3];
    float *__restrict__ pout_u = L.pf_u[k % 3];
    float *__restrict__ pout_v = L.pf_v[k % 3];
    const bool defer = L.defer_x != 0;
    const bool x_two = defer && !first && (k & 1) == 0;
    const bool x_one = !first && (!defer || ((k & 1) == 1 && !active));
    const bool x_read = x_two ? (k > 2) : (defer ? (k >= 3) : (k > 1));
    const float alpha2 = x_two ? L.alpha[(k - 2) & 1] : 0.f;
    double acc_pq = 0., acc_qz = 0., acc_qmq = 0., acc_rq = 0., acc_qq = 0., acc_rz = 0., acc_rr = 0.;

    const ItemRange tr = item_range_walk(ntiles, L.xcd_bands);
    int parity = 0;
    for (int t = tr.first; t < tr.end; t += tr.step, parity ^= 1) {
        float *const s_nu = s_nu2 + parity * NSZ, *const s_nv = s_nv2 + parity * NSZ;
        const int tx0 = (t % tiles_x) * TX, ty0 = y0 + (t / tiles_x) * TY;
        // ---- loads of the thread's two tile groups first (r_{k-1} and the operator; addresses of groups beyond a ragged
        // edge are clamped into the frame, their values never used), so that they are in flight while phase 0 waits for p
        QCoef c3[2];
        float r3u[2][4], r3v[2][4];
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            const int gx = tid & 31, gy = (tid >> 5) + kTileY * slot;
            const int x0 = tx0 + 4 * gx, y = ty0 + gy;
            const bool valid = y < y1 && x0 < w;
            const unsigned o = valid ? (unsigned)(y * pitch + x0) * 4u : 0u;
            QCoef &c = c3[slot];
            *(float4 *)r3u[slot] = ld4(at(rin_u, o)); *(float4 *)r3v[slot] = ld4(at(rin_v, o));
            *(float4 *)c.a1 = ld4(at(L.a1, o)); *(float4 *)c.a4 = ld4(at(L.a4, o));
            // no streaming hint by default (bit 256, not the stored-q kernels' bit 8): the neighbouring tiles' rings read these
            // lines too, -1.5 % without it.  The switch stays because the kernel is 3 % slower without the branch (sic).
            *(float4 *)c.a2 = ld4_if(at(L.a2, o), L.nt_hints & 256);
            if (UNITW) {
#pragma unroll
                for (int e = 0; e < 4; e++) { c.wx[e] = -1.f; c.wy[e] = -1.f; c.wys[e] = -1.f; }
                c.wxw = -1.f;
            } else {
                *(float4 *)c.wx = ld4(at(L.wx, o)); *(float4 *)c.wy = ld4(at(L.wy, o));
                *(float4 *)c.wys = ld4(at(L.wy, (valid && y > 0) ? o - 4u * (unsigned)pitch : o));     // unused in the frame's first row
                c.wxw = *at(L.wx, (valid && x0 > 0) ? o - 4u : o);                                      // unused in its first column
            }
        }
        // ---- phase 0: p_{k-1} on the tile + 2 rows / one float4 group around it (zero outside the frame)
        if (!first) {
            constexpr int GW = TX / 4 + 2;                        // groups per staged row: one left, one right of the tile
            for (int i = tid; i < GW * (TY + 4); i += 256) {
                const int gx = i % GW - 1, gy = i / GW - 2;
                const int x0 = tx0 + 4 * gx, y = ty0 + gy;
                float4 pu = make_float4(0, 0, 0, 0), pv = pu;
                if (y >= 0 && y < h && x0 >= 0 && x0 < w) {
                    const unsigned o = (unsigned)(y * pitch + x0) * 4u;
                    if (BANDED && y < y0) { pu = ld4(at(L.pup_u[(k + 2) % 3], o)); pv = ld4(at(L.pup_v[(k + 2) % 3], o)); }
                    else if (BANDED && y >= y1) { pu = ld4(at(L.pdn_u[(k + 2) % 3], o)); pv = ld4(at(L.pdn_v[(k + 2) % 3], o)); }
                    else { pu = ld4(at(pin_u, o)); pv = ld4(at(pin_v, o)); }   // planes are padded to a multiple of 64 floats: in bounds
                    if (x0 + 3 >= w) {                            // beyond the frame's last column: zero, as the other forms do
                        if (x0 + 1 >= w) { pu.y = 0.f; pv.y = 0.f; }
                        if (x0 + 2 >= w) { pu.z = 0.f; pv.z = 0.f; }
                        pu.w = 0.f; pv.w = 0.f;
                    }
                }
                st4(&s_ou[(gy + 2) * kQCols + kQOff + 4 * gx], pu);
                st4(&s_ov[(gy + 2) * kQCols + kQOff + 4 * gx], pv);
            }
            __syncthreads();
        }
        // ---- phase 1: the ring group (one each for the first 100 threads), then the two tile groups
#pragma unroll
        for (int sl = 0; sl < 3; sl++) {
            const int slot = (sl + 2) % 3;
            int gx, gy;
            const bool own = slot < 2;
            if (own) { gx = tid & 31; gy = (tid >> 5) + kTileY * slot; }
            else if (tid < 34) { gx = tid - 1; gy = -1; }
            else if (tid < 68) { gx = tid - 35; gy = TY; }
            else if (tid < 84) { gx = -1; gy = tid - 68; }
            else if (tid < 100) { gx = TX / 4; gy = tid - 84; }
            else { gx = 0; gy = -9; }                              // no ring group for this thread
            const int x0 = tx0 + 4 * gx, y = ty0 + gy;
            const bool valid = (gy >= -1) && y >= 0 && y < h && x0 >= 0 && x0 < w && (!own || y < y1);
            QCoef cr;
            float ru[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0}, pnu[4] = {0, 0, 0, 0}, pnv[4] = {0, 0, 0, 0};
            if (!own) {
#pragma unroll
                for (int e = 0; e < 4; e++) { cr.a1[e] = 1.f; cr.a4[e] = 1.f; cr.a2[e] = 0.f; cr.wx[e] = 0.f; cr.wy[e] = 0.f; cr.wys[e] = 0.f; }
                cr.wxw = 0.f;
                if (valid) {
                    const unsigned o = (unsigned)(y * pitch + x0) * 4u;
                    const int kr = first ? 0 : ko;
                    if (BANDED && y < y0) { *(float4 *)ru = ld4(at(L.rup_u[kr], o)); *(float4 *)rv = ld4(at(L.rup_v[kr], o)); }
                    else if (BANDED && y >= y1) { *(float4 *)ru = ld4(at(L.rdn_u[kr], o)); *(float4 *)rv = ld4(at(L.rdn_v[kr], o)); }
                    else { *(float4 *)ru = ld4(at(rin_u, o)); *(float4 *)rv = ld4(at(rin_v, o)); }
                    *(float4 *)cr.a1 = ld4(at(L.a1, o)); *(float4 *)cr.a4 = ld4(at(L.a4, o));
                    *(float4 *)cr.a2 = ld4(at(L.a2, o));
                    if (UNITW) {
#pragma unroll
                        for (int e = 0; e < 4; e++) { cr.wx[e] = -1.f; cr.wy[e] = -1.f; cr.wys[e] = -1.f; }
                        cr.wxw = -1.f;
                    } else {
                        *(float4 *)cr.wx = ld4(at(L.wx, o)); *(float4 *)cr.wy = ld4(at(L.wy, o));
                        if (y > 0) *(float4 *)cr.wys = ld4(at((BANDED && y < y0) ? L.wy_up : L.wy, o - 4u * (unsigned)pitch));
                        if (x0 > 0) cr.wxw = *at(L.wx, o - 4u);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) { ru[e] = r3u[slot & 1][e]; rv[e] = r3v[slot & 1][e]; }
            }
            const QCoef &c = own ? c3[slot & 1] : cr;
            if (valid) {
                const unsigned o = (unsigned)(y * pitch + x0) * 4u;
                float pu[4] = {0, 0, 0, 0}, pv[4] = {0, 0, 0, 0};
                if (!first) {
                    float qu[4], qv[4];
                    stencil_group(s_ou, s_ov, gy + 2, kQOff + 4 * gx, x0, y, w, h, c, qu, qv);        // q_{k-1}, again
                    *(float4 *)pu = ld4(&s_ou[(gy + 2) * kQCols + kQOff + 4 * gx]);
                    *(float4 *)pv = ld4(&s_ov[(gy + 2) * kQCols + kQOff + 4 * gx]);
                    if (own && (x_two || x_one)) {
                        float xu[4] = {0, 0, 0, 0}, xv[4] = {0, 0, 0, 0};
                        if (x_read) { *(float4 *)xu = ld4_if(at(L.xu, o), L.nt_hints & 1); *(float4 *)xv = ld4_if(at(L.xv, o), L.nt_hints & 1); }
                        if (x_two) {
                            float ou[4], ov[4];
                            *(float4 *)ou = ld4_nt(at(pin2_u, o)); *(float4 *)ov = ld4_nt(at(pin2_v, o));
#pragma unroll
                            for (int e = 0; e < 4; e++) { xu[e] = alpha2 * ou[e] + xu[e]; xv[e] = alpha2 * ov[e] + xv[e]; }
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) { xu[e] = alpha * pu[e] + xu[e]; xv[e] = alpha * pv[e] + xv[e]; }   // ref .cu:1172
                        st4_if(at(L.xu, o), *(float4 *)xu, L.nt_hints & 1);
                        st4_if(at(L.xv, o), *(float4 *)xv, L.nt_hints & 1);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) { ru[e] = nalpha * qu[e] + ru[e]; rv[e] = nalpha * qv[e] + rv[e]; }    // ref .cu:1174
                    if (own && active) { st4_nt(at(rout_u, o), *(float4 *)ru); st4_nt(at(rout_v, o), *(float4 *)rv); }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool ok = (x0 + e) < w;
                    if (!ok) { ru[e] = 0.f; rv[e] = 0.f; }
                    pnu[e] = ok ? direction(ru[e], pu[e], c.a1[e], beta, first) : 0.f;
                    pnv[e] = ok ? direction(rv[e], pv[e], c.a4[e], beta, first) : 0.f;
                }
                if (own && active) { st4_nt(at(pout_u, o), *(float4 *)pnu); st4_nt(at(pout_v, o), *(float4 *)pnv); }
            }
            if (gy >= -1) {
                st4(&s_nu[(gy + 1) * kQCols + kQOff + 4 * gx], *(float4 *)pnu);
                st4(&s_nv[(gy + 1) * kQCols + kQOff + 4 * gx], *(float4 *)pnv);
            }
            if (own) {
#pragma unroll
                for (int e = 0; e < 4; e++) { r3u[slot & 1][e] = ru[e]; r3v[slot & 1][e] = rv[e]; }     // r_k, for the sums of phase 2
            }
        }
        __syncthreads();
        // ---- phase 2: q_k on the tile and the partial sums (q_k is not stored: the next launch forms it again)
        if (active) {
#pragma unroll
            for (int slot = 0; slot < 2; slot++) {
                const int gx = tid & 31, gy = (tid >> 5) + kTileY * slot;
                const int x0 = tx0 + 4 * gx, y = ty0 + gy;
                if (y < y1 && x0 < w) {
                    float qu[4], qv[4];
                    stencil_group(s_nu, s_nv, gy + 1, kQOff + 4 * gx, x0, y, w, h, c3[slot], qu, qv);
                    float pku[4], pkv[4];
                    *(float4 *)pku = ld4(&s_nu[(gy + 1) * kQCols + kQOff + 4 * gx]); *(float4 *)pkv = ld4(&s_nv[(gy + 1) * kQCols + kQOff + 4 * gx]);
                    float d_pq = 0.f, d_qz = 0.f, d_qmq = 0.f, d_rq = 0.f, d_qq = 0.f, d_rz = 0.f, d_rr = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (x0 + e < w) {
                            const float iu = rcp_exact(c3[slot].a1[e]), iv = rcp_exact(c3[slot].a4[e]);
                            const float zu = iu * r3u[slot][e], zv = iv * r3v[slot][e];
                            d_pq += pku[e] * qu[e]; d_pq += pkv[e] * qv[e];
                            d_qz += qu[e] * zu; d_qz += qv[e] * zv;
                            d_qmq += qu[e] * (iu * qu[e]); d_qmq += qv[e] * (iv * qv[e]);
                            d_rq += r3u[slot][e] * qu[e]; d_rq += r3v[slot][e] * qv[e];
                            d_qq += qu[e] * qu[e]; d_qq += qv[e] * qv[e];
                            d_rz += r3u[slot][e] * zu; d_rz += r3v[slot][e] * zv;
                            d_rr += r3u[slot][e] * r3u[slot][e]; d_rr += r3v[slot][e] * r3v[slot][e];
                        }
                    }
                    acc_pq += (double)d_pq; acc_qz += (double)d_qz; acc_qmq += (double)d_qmq; acc_rq += (double)d_rq;
                    acc_qq += (double)d_qq; acc_rz += (double)d_rz; acc_rr += (double)d_rr;
                }
            }
        }
    }
    if (!active) return;
    double *own_blk = L.part_own + pout_off;
    const double accs[kPartKinds] = {acc_rz, acc_rr, acc_pq, acc_qz, acc_qmq, acc_rq, acc_qq};
    double tot[kPartKinds];
    block_sum_multi_256<kPartKinds>(accs, s_red, tot);
    if (tid == 0) {
#pragma unroll
        for (int j = 0; j < kPartKinds; j++) own_blk[j * kMaxParts + blockIdx.x] = tot[j];
    }
}

// u += dx, v += dy after a solve made of `nlaunched` fused kernels (ref .cu:1185-1195).  A solve that ran into its
// iteration cap leaves the last alpha p to be added here; one that met the tolerance has a complete x.
__global__ __launch_bounds__(256) void k_flow_update_fused(LevelPtrs L, int nlaunched, int nparts)
{
    __shared__ double s_red[8];
    const int K = nlaunched;
    const PcgState st = L.st[K & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) *L.iter_total += st.iters;
    const int n = st.iters;
    if (n == 0) return;
    // A solve that met the tolerance has a complete x (the launch that ended the loop applied what was pending).  One that
    // ran into its iteration cap (all K launches ran an iteration) still owes alpha_{K-1} p_{K-1}, and with deferred x
    // updates also alpha_{K-2} p_{K-2} when the last launch was an odd one.
    const bool capped = (st.stopped == 0);
    const bool defer = L.defer_x != 0;
    const bool owe2 = capped && defer && K >= 2 && ((K - 1) & 1) == 1;
    bool have_x = true;
    if (capped) have_x = defer ? (owe2 ? (K >= 4) : (K >= 3)) : (K >= 2);
    float a1 = 0.f, a2 = 0.f;
    if (capped)
        a1 = st.rz / (float)fold_band_partials_256(L.band_parts, ((K - 1) & 1) * kPartBlock + kPartPq, nparts, L.nbands, s_red);
    if (owe2) a2 = L.alpha[(K - 2) & 1];
    const float *__restrict__ p1u = L.pf_u[(K + 2) % 3];            // p_{K-1}
    const float *__restrict__ p1v = L.pf_v[(K + 2) % 3];
    const float *__restrict__ p2u = L.pf_u[(K + 1) % 3];            // p_{K-2}
    const float *__restrict__ p2v = L.pf_v[(K + 1) % 3];
    const int w = L.w, pitch = L.pitch;
    const int gw = (w + 3) / 4;
    const long ngroups = (long)gw * (L.y1 - L.y0);
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long)gridDim.x * 256) {
        const int y = (int)(g / gw), x = (int)(g - (long)y * gw) * 4;
        const size_t o = (size_t)(L.y0 + y) * pitch + x;
        float4 u = ld4(L.u + o), v = ld4(L.v + o), dx = make_float4(0, 0, 0, 0), dy = dx;
        if (have_x) { dx = ld4(L.xu + o); dy = ld4(L.xv + o); }
        if (owe2) {
            const float4 pu = ld4(p2u + o), pv = ld4(p2v + o);
            dx.x = a2 * pu.x + dx.x; dx.y = a2 * pu.y + dx.y; dx.z = a2 * pu.z + dx.z; dx.w = a2 * pu.w + dx.w;
            dy.x = a2 * pv.x + dy.x; dy.y = a2 * pv.y + dy.y; dy.z = a2 * pv.z + dy.z; dy.w = a2 * pv.w + dy.w;
        }
        if (capped) {
            const float4 pu = ld4(p1u + o), pv = ld4(p1v + o);
            dx.x = a1 * pu.x + dx.x; dx.y = a1 * pu.y + dx.y; dx.z = a1 * pu.z + dx.z; dx.w = a1 * pu.w + dx.w;
            dy.x = a1 * pv.x + dy.x; dy.y = a1 * pv.y + dy.y; dy.z = a1 * pv.z + dy.z; dy.w = a1 * pv.w + dy.w;
            if (!L.lean) { st4(L.xu + o, dx); st4(L.xv + o, dy); }     // keep x complete for the debug tap
        }
        u.x = u.x + dx.x; u.y = u.y + dx.y; u.z = u.z + dx.z; u.w = u.w + dx.w;
        v.x = v.x + dy.x; v.y = v.y + dy.y; v.z = v.z + dy.z; v.w = v.w + dy.w;
        st4(L.u + o, u); st4(L.v + o, v);
    }
}

struct BOperands {
    float4 ru, rv, pu, pv, qu, qv, mu, mv, xu, xv, ou, ov;   // ou/ov: the previous iteration's p (deferred x update)
    size_t o;
    int x;
    bool valid;
};

// Operand loads of one 256-group chunk of pass B.  None of them depends on alpha, so the first chunk's loads
// are issued BEFORE the partials are folded and the next chunk's before the current one is computed: the
// reduction's round trip and the arithmetic hide under memory latency instead of adding to it.
// x handling of pass B(k).  Immediate mode: x += alpha_k p_k every iteration, as the reference does.  Deferred mode
// (default): even iterations leave x alone; odd iterations apply the two pending updates in the reference's order,
//   x <- alpha_k p_k + (alpha_{k-1} p_{k-1} + x),
// reading p_{k-1} from the other half of the p ping-pong.  Same operations, same roundings, but x is read and
// written every second iteration only (4 B/pixel/iteration less traffic).  A solve that ends on an even iteration
// has one update pending; k_flow_update applies it.
enum { XMODE_NOW_FIRST = 0, XMODE_NOW = 1, XMODE_SKIP = 2, XMODE_PAIR_FIRST = 3, XMODE_PAIR = 4 };
__device__ __forceinline__ int pass_b_xmode(int defer, int k)
{
    if (!defer) return k == 0 ? XMODE_NOW_FIRST : XMODE_NOW;
    if ((k & 1) == 0) return XMODE_SKIP;
    return k == 1 ? XMODE_PAIR_FIRST : XMODE_PAIR;
}

__device__ __forceinline__ void pass_b_issue(const LevelPtrs &L, int k, int xmode, const ItemRange &cr, int ci,
                                             long ngroups, int gw, int pitch, BOperands &b)
{
    b.valid = false;
    if (ci >= cr.end) return;
    const int c = L.reverse_b ? cr.end - 1 - (ci - cr.base) : ci;
    const long g = (long)c * 256 + threadIdx.x;
    if (g >= ngroups) return;
    const int y = (int)(g / gw);
    b.x = (int)(g - (long)y * gw) * 4;
    b.o = (size_t)(L.y0 + y) * pitch + b.x;
    b.valid = true;
    b.ru = ld4(L.ru + b.o); b.rv = ld4(L.rv + b.o);
    b.pu = ld4(L.pu[(k + 1) & 1] + b.o); b.pv = ld4(L.pv[(k + 1) & 1] + b.o);
    b.qu = ld4_if(L.qu + b.o, L.nt_hints & 2); b.qv = ld4_if(L.qv + b.o, L.nt_hints & 2);
    b.mu = ld4_if(L.mu + b.o, L.nt_hints & 4); b.mv = ld4_if(L.mv + b.o, L.nt_hints & 4);
    if (xmode == XMODE_NOW || xmode == XMODE_PAIR) { b.xu = ld4_if(L.xu + b.o, L.nt_hints & 1); b.xv = ld4_if(L.xv + b.o, L.nt_hints & 1); }
    else { b.xu = make_float4(0, 0, 0, 0); b.xv = make_float4(0, 0, 0, 0); }
    if (xmode == XMODE_PAIR_FIRST || xmode == XMODE_PAIR) { b.ou = ld4(L.pu[k & 1] + b.o); b.ov = ld4(L.pv[k & 1] + b.o); }
    else { b.ou = make_float4(0, 0, 0, 0); b.ov = make_float4(0, 0, 0, 0); }
}

#ifdef OCTANE_DIAG      // the two-pass form (pass A + pass B per iteration) exists in the diagnostic library only: the product always runs one kernel per iteration
__global__ __launch_bounds__(256) void k_pcg_pass_b(LevelPtrs L, int k, int nparts_a)
{
    __shared__ double s_red[8];
    const int xmode = pass_b_xmode(L.defer_x, k);
    const int w = L.w, pitch = L.pitch;
    const int gw = (w + 3) / 4;
    const long ngroups = (long)gw * (L.y1 - L.y0);         // float4 groups of the rows this launch owns
    // Pass B walks the frame from the end to the start and pass A from the start to the end, so
    // each pass begins on the planes the previous one touched last (p, q, r and the diagonal are
    // still in the 256 MiB Infinity Cache there).
    const int nchunks = (int)((ngroups + 255) / 256);
    const ItemRange cr = item_range(nchunks, L.xcd_bands == 1);
    BOperands cur, nxt;
    pass_b_issue(L, k, xmode, cr, cr.first, ngroups, gw, pitch, cur);

    const PcgState st = L.st[(k + 1) & 1];          // requested together with the partials: one round trip
    const float pq = (float)fold_band_partials_256(L.band_parts, kPartPq, nparts_a, L.nbands, s_red);
    if (st.stopped) return;
    const float alpha = st.rz / pq;                        // ref .cu:1169
    const float nalpha = (float)(-1. * (double)alpha);     // ref .cu:1174
    const float alpha_prev = (xmode >= XMODE_PAIR_FIRST) ? L.alpha[(k - 1) & 1] : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) L.alpha[k & 1] = alpha;
    double acc_rz = 0., acc_rr = 0.;
    for (int ci = cr.first; ci < cr.end; ci += cr.step) {
        pass_b_issue(L, k, xmode, cr, ci + cr.step, ngroups, gw, pitch, nxt);
        if (cur.valid) {
            float xu[4], xv[4], ru[4], rv[4], pu[4], pv[4], qu[4], qv[4], mu[4], mv[4], ou[4], ov[4];
            *(float4 *)ou = cur.ou; *(float4 *)ov = cur.ov;
            *(float4 *)xu = cur.xu; *(float4 *)xv = cur.xv; *(float4 *)ru = cur.ru; *(float4 *)rv = cur.rv;
            *(float4 *)pu = cur.pu; *(float4 *)pv = cur.pv; *(float4 *)qu = cur.qu; *(float4 *)qv = cur.qv;
            *(float4 *)mu = cur.mu; *(float4 *)mv = cur.mv;
            float srz = 0.f, srr = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (xmode >= XMODE_PAIR_FIRST) {           // the previous iteration's update first
                    xu[e] = alpha_prev * ou[e] + xu[e];
                    xv[e] = alpha_prev * ov[e] + xv[e];
                }
                xu[e] = alpha * pu[e] + xu[e];             // jVecPVec(p0,x0,x0,alphak), ref .cu:1172
                xv[e] = alpha * pv[e] + xv[e];
                ru[e] = nalpha * qu[e] + ru[e];            // jVecPVec(dummyvec,bcu,rk,-alphak), ref .cu:1174
                rv[e] = nalpha * qv[e] + rv[e];
                if (cur.x + e < w) {
                    const float zu = mu[e] * ru[e], zv = mv[e] * rv[e];
                    srz += ru[e] * zu; srz += rv[e] * zv;
                    srr += ru[e] * ru[e]; srr += rv[e] * rv[e];
                }
            }
            if (xmode != XMODE_SKIP) {
                st4_if(L.xu + cur.o, *(float4 *)xu, L.nt_hints & 1); st4_if(L.xv + cur.o, *(float4 *)xv, L.nt_hints & 1);
            }
            st4(L.ru + cur.o, *(float4 *)ru); st4(L.rv + cur.o, *(float4 *)rv);
            acc_rz += (double)srz; acc_rr += (double)srr;
        }
        cur = nxt;
    }
    const double trz = block_sum_256(acc_rz, s_red);
    const double trr = block_sum_256(acc_rr, s_red);
    if (threadIdx.x == 0) { L.part_rz[blockIdx.x] = trz; L.part_rr[blockIdx.x] = trr; }
}
#endif

// u += dx, v += dy after a solve (ref .cu:1185-1195).  x is only meaningful if at least one
// iteration ran; the reference's x0 stays zero otherwise.
#ifdef OCTANE_DIAG
__global__ __launch_bounds__(256) void k_flow_update(LevelPtrs L, int nlaunched)
{
    const PcgState st = L.st[nlaunched & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) *L.iter_total += st.iters;
    const int n = st.iters;
    if (n == 0) return;
    // deferred-x mode: an odd number of executed iterations leaves alpha_{n-1} p_{n-1} pending
    const bool pending = L.defer_x && (n & 1);
    const float apend = pending ? L.alpha[(n - 1) & 1] : 0.f;
    const float *__restrict__ ppu = L.pu[n & 1];
    const float *__restrict__ ppv = L.pv[n & 1];
    const int w = L.w, pitch = L.pitch;
    const int gw = (w + 3) / 4;
    const long ngroups = (long)gw * (L.y1 - L.y0);
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long)gridDim.x * 256) {
        const int y = (int)(g / gw), x = (int)(g - (long)y * gw) * 4;
        const size_t o = (size_t)(L.y0 + y) * pitch + x;
        float4 u = ld4(L.u + o), v = ld4(L.v + o), dx, dy;
        if (pending && n == 1) { dx = make_float4(0, 0, 0, 0); dy = dx; }
        else { dx = ld4(L.xu + o); dy = ld4(L.xv + o); }
        if (pending) {
            const float4 pu = ld4(ppu + o), pv = ld4(ppv + o);
            dx.x = apend * pu.x + dx.x; dx.y = apend * pu.y + dx.y; dx.z = apend * pu.z + dx.z; dx.w = apend * pu.w + dx.w;
            dy.x = apend * pv.x + dy.x; dy.y = apend * pv.y + dy.y; dy.z = apend * pv.z + dy.z; dy.w = apend * pv.w + dy.w;
            st4(L.xu + o, dx); st4(L.xv + o, dy);          // keep x complete for the debug tap
        }
        u.x = u.x + dx.x; u.y = u.y + dx.y; u.z = u.z + dx.z; u.w = u.w + dx.w;
        v.x = v.x + dy.x; v.y = v.y + dy.y; v.z = v.z + dy.z; v.w = v.w + dy.w;
        st4(L.u + o, u); st4(L.v + o, v);
    }
}
#endif

static int stream_grid_size(int w, int h)
{
    long groups = (long)((w + 3) / 4) * h;
    return balanced_grid((groups + 255) / 256);
}

// Pass B keeps two chunks of operands in registers (~124 VGPRs): 4 workgroups per CU are resident, so its
// persistent grid is capped at 256 x 4.
// Residency caps of the two passes' persistent grids.  Defaults fill the chip (3 x 256 and 4 x 256 workgroups); a
// caller that runs two plans side by side (octane_vof_batch_run's lanes) lowers them so that the other lane's
// latency-bound kernels find free wave slots instead of queueing behind a whole bandwidth-bound launch.
static int g_cap_a = 768, g_cap_b = 1024;
void set_pass_caps(int cap_a, int cap_b)
{
    g_cap_a = (cap_a >= 64 && cap_a <= 768) ? cap_a : 768;
    g_cap_b = (cap_b >= 64 && cap_b <= 1024) ? cap_b : 1024;
}

static int pass_b_grid_size(int w, int h)
{
    long chunks = ((long)((w + 3) / 4) * h + 255) / 256;
    int g = balanced_grid(chunks);
    const long cap = g_cap_b;
    if (g > cap) { long rounds = (chunks + cap - 1) / cap; g = (int)((chunks + rounds - 1) / rounds); }
    return g;
}

// Which form of pass A a level runs.  Default (0), from per-level timings of every form on the same arena
// (octane_vof_plan_probe, several boxes):
//   below 1 Mpixel        1  latency form, 128 x 8 tiles (7-15 % faster than taller tiles: more workgroups)
//   1 .. 4 Mpixel         2  128 x 16 tiles
//   4 .. 12 Mpixel        3  LDS-ring marching: 64.5-65.3 us at 2500^2 against 72-75 us for every tiled form
//   12 Mpixel and up      2  128 x 16 tiles (the marching form is 3-8 % slower at 5000^2)
// OCTANE_TUNE_PASS_A forces 1, 2, 4 (128 x 32), 5 (256 x 8) or 3.
static int g_pass_a_variant = 0;
void set_pass_a_variant(int v) { g_pass_a_variant = (v >= 0 && v <= 5) ? v : 0; }
static int pass_a_choice(int w, int h)
{
    if (g_pass_a_variant != 0) return g_pass_a_variant;
    const long npix = (long)w * h;
    if (npix < (1L << 20)) return 1;
    if (npix >= (4L << 20) && npix < (12L << 20)) return 3;
    return 2;
}

static int g_unit_cap = 768;
void set_unit_w_cap(int c) { g_unit_cap = (c >= 256 && c <= kMaxParts) ? c : 768; }

// Grid of the unit-weight launches (first GNC step).  The tiled form needs 128 instead of 152 VGPRs there, so four
// instead of three workgroups per CU are resident.
int pcg_grid_size_unit_w(int w, int h)
{
    const int variant = pass_a_choice(w, h);
    if (variant != 2) return pcg_grid_size(w, h);
    const long items = (long)((w + kTileX - 1) / kTileX) * ((h + kTileY * 2 - 1) / (kTileY * 2));
    const long cap = g_unit_cap;
    if (items <= cap) return (int)items;
    const long rounds = (items + cap - 1) / cap;
    return (int)((items + rounds - 1) / rounds);
}

int pcg_grid_size(int w, int h)
{
    const int variant = pass_a_choice(w, h);
    if (variant == 3) {   // marching: at least 8 rows per workgroup, 3 workgroups per CU resident (168 VGPRs)
        long rows = (long)((w + kMarchW - 1) / kMarchW) * h;
        long g = rows / 8;
        if (g < 1) g = 1;
        return (int)(g > g_cap_a ? g_cap_a : g);
    }
    const int R = (variant == 5) ? 2 : variant;          // 5 = two sub-tiles side by side (256 x 8)
    const long items = (variant == 5) ? (long)((w + 2 * kTileX - 1) / (2 * kTileX)) * ((h + kTileY - 1) / kTileY)
                                      : (long)((w + kTileX - 1) / kTileX) * ((h + kTileY * R - 1) / (kTileY * R));
    const long cap = (R == 2) ? g_cap_a : (g_cap_a < 512 ? g_cap_a : 512);   // 185 / 148 / 236 VGPRs: 2 / 3 / 2 workgroups per CU resident
    if (items <= cap) return (int)items;
    const long rounds = (items + cap - 1) / cap;
    return (int)((items + rounds - 1) / rounds);
}

// Grid of the 128 x 16 tiled form over `rows` rows: what a row band of a level launches (launch_pcg_pass_a below).
int pcg_band_grid_size(int w, int rows)
{
    const long items = (long)((w + kTileX - 1) / kTileX) * ((rows + 2 * kTileY - 1) / (2 * kTileY));
    const long cap = 768;
    if (items <= cap) return (int)(items < 1 ? 1 : items);
    const long rounds = (items + cap - 1) / cap;
    return (int)((items + rounds - 1) / rounds);
}

#ifdef OCTANE_DIAG
void launch_pcg_pass_a(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol)
{
    if (L.nbands > 1 || L.y0 != 0 || L.y1 != L.h) {   // a row band: only the tiled form knows about bands
        if (L.unit_w) hipLaunchKernelGGL((k_pcg_pass_a<2, false, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL((k_pcg_pass_a<2, false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        return;
    }
    switch (pass_a_choice(L.w, L.h)) {
    case 1: hipLaunchKernelGGL(k_pcg_pass_a_lat<1>, dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol); break;
    case 5: hipLaunchKernelGGL((k_pcg_pass_a<2, true, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol); break;
    case 4: hipLaunchKernelGGL((k_pcg_pass_a<4, false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol); break;
    case 3:
        if (L.unit_w) hipLaunchKernelGGL(k_pcg_pass_a_ring<true>, dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL(k_pcg_pass_a_ring<false>, dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        break;
    default:
        if (L.unit_w) hipLaunchKernelGGL((k_pcg_pass_a<2, false, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL((k_pcg_pass_a<2, false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        break;
    }
}

void launch_pcg_pass_b(hipStream_t s, const LevelPtrs &L, int k, int nparts_a, int grid)
{
    hipLaunchKernelGGL(k_pcg_pass_b, dim3(grid), dim3(256), 0, s, L, k, nparts_a);
}
#endif  // OCTANE_DIAG

// Fused PCG: grid (shared by every launch of a solve, it is also the number of partials), launch, flow update.
static int g_fused_q = 1;            // large levels of a plain plan recompute q instead of storing it (k_pcg_fused_q); 0 = always store q
void set_fused_q(int v) { g_fused_q = v != 0; }
static int g_fused_r = 0;            // tuning: 0 = by size, 1 / 2 = force the 128 x 8 / 128 x 16 tile
void set_fused_rows(int r) { g_fused_r = (r == 1 || r == 2) ? r : 0; }
static int fused_rows(int w, int rows) { return g_fused_r ? g_fused_r : (((long)w * rows < (1L << 20)) ? 1 : 2); }
// q is recomputed where that pays.  Round 1 (register-staged kernel): from 3 M pixels (1250^2 33 -> 35 us; 2000^2 +2 %; 2500^2
// 112 -> 99 us; 5000^2 395 -> 339 us).  Round 2 (LDS-DMA kernel): from 2 M pixels, i.e. every level above the persistent solve's
// range (1500^2: 49.4 -> 41.4 us per launch, 1700^2: 56.2 -> 45.9); at 1 M pixels, where only plans beside other lanes get
// (their persistent solves are capped), the stored-q form is still as good (64 x 2000^2: 181.4 against 179.1 Mpix/s)
static long g_fused_q_min = 2L << 20; // pixels from which q is recomputed
void set_fused_q_min(long px) { g_fused_q_min = px > 0 ? px : (2L << 20); }
int pcg_fused_q_form(int w, int rows, int h)
{
    return g_fused_q && fused_rows(w, rows) == 2 && (long)w * rows >= g_fused_q_min &&
           (long)(w + 64) * h < (1L << 30);               // 32-bit byte offsets inside a plane
}

int pcg_fused_grid_size(int w, int rows, int unit_w, int q_form)
{
    const int R = fused_rows(w, rows);                                // small levels: more, smaller tiles
    const long items = (long)((w + kTileX - 1) / kTileX) * ((rows + kTileY * R - 1) / (kTileY * R));
    // never more workgroups than are resident at once (a second wave of a persistent grid runs on a half-empty chip):
    // 128 x 16 tiles need 180 / 163 VGPRs (2 / 3 workgroups per CU), 128 x 8 tiles 135 / 122 (3 / 4)
    // (the q-recomputing form of the 128 x 16 tile: 191 / 220 VGPRs and 65 KB of LDS, 2 per CU)
    const bool qform = q_form && R == 2;
    const long cap = 256 * (R == 2 ? ((unit_w && !qform) ? 3 : 2) : (unit_w ? 4 : 3));
    if (items <= cap) return (int)(items < 1 ? 1 : items);
    const long rounds = (items + cap - 1) / cap;
    // stored-q forms: as many workgroups as give every one the same number of tiles; the q-recomputing form fills the chip
    // and lets the last round run on fewer workgroups (measured on a kernel trace: -0.7 % over the two levels that use it)
    int g = qform ? (int)cap : (int)((items + rounds - 1) / rounds);
    if (grid_multiple() > 1 && g >= 8 * grid_multiple()) g = g / grid_multiple() * grid_multiple();   // XCD bands want a multiple of 8
    return g;
}

// Which workgroups walk the border-column tiles.  The tiles of the frame's first and last tile column are the expensive ones of the LDS-DMA
// kernel (no DMA staging, the bordered operator: ~1 us more than an interior tile), and the walk "tile t -> workgroup" is periodic in the
// tile-column count: at 5000 pixels (40 columns, 512 workgroups) 64 workgroups own ALL left-border tiles, five of their 24-25, and finish
// last in every launch.  Rotating the columns of tile row r by r (a permutation within the row, so every tile is still done exactly once)
// spreads them: at most two per workgroup at 5000^2 (-1.1 % per launch, profiles/r6_row_rotation.txt) -- but at 10848 pixels (85 columns)
// the plain walk is the more even one, so the host counts both and the kernel rotates only where that lowers the maximum.
int pcg_row_rotation(int w, int rows, int grid, int walk_mode) { return pcg_row_rotation_count(w, rows, grid, walk_mode, nullptr); }
int pcg_row_rotation_count(int w, int rows, int grid, int walk_mode, int *out3)
{
    const int tiles_x = (w + kTileX - 1) / kTileX, tiles_y = (rows + 2 * kTileY - 1) / (2 * kTileY);
    const long nt = (long)tiles_x * tiles_y;
    if (out3) { out3[0] = tiles_x; out3[1] = out3[2] = -1; }
    if (tiles_x < 3 || grid < 1 || grid > kMaxParts || (grid % tiles_x == 0 && tiles_x > 1)) return 0;      // (columns dividing the grid: the kernel rotates by round)
    if (!(walk_mode == 0 || ((walk_mode == 3 || walk_mode == 4) && (grid & 63) == 0))) return 0;           // walks this count does not model
    std::vector<int> plain((size_t)grid, 0), rot((size_t)grid, 0);
    const int run = walk_mode == 3 ? 4 : 8;
    for (int b = 0; b < grid; b++) {
        long first = b;
        if (walk_mode == 3 || walk_mode == 4) { const int x = b & 7, j = b >> 3; first = ((long)(j / run) * 8 + x) * run + (j % run); }
        for (long t = first; t < nt; t += grid) {
            const int row = (int)(t / tiles_x), col = (int)(t % tiles_x), colr = (col + row) % tiles_x;
            plain[b] += (col == 0 || col == tiles_x - 1);
            rot[b] += (colr == 0 || colr == tiles_x - 1);
        }
    }
    int mp = 0, mr = 0;
    for (int b = 0; b < grid; b++) { mp = plain[b] > mp ? plain[b] : mp; mr = rot[b] > mr ? rot[b] : mr; }
    if (out3) { out3[1] = mp; out3[2] = mr; }
    return mr < mp ? 1 : 0;
}

#ifdef OCTANE_DIAG
static int g_q_diag = 0;              // diagnostic build only: route whole-level q-form launches to the copy in pcg_fused_q_diag.hip
void set_q_diag(int v) { g_q_diag = v != 0; }
#endif
static int g_q_dma = 1;               // q-form launches by the LDS-DMA form (pcg_fused_q_dma.hip): same bits, 9 % faster
void set_q_dma(int v) { g_q_dma = v != 0; }
void launch_pcg_fused(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol)
{
    const bool small = fused_rows(L.w, L.y1 - L.y0) == 1;
    const bool whole = L.nbands == 1 && L.y0 == 0 && L.y1 == L.h;
#ifdef OCTANE_DIAG
    if (g_q_diag && L.q_form && whole) { launch_pcg_fused_q_diag(s, L, k, nparts_prev, grid, tol); return; }
#endif
    if (g_q_dma && !L.no_dma && L.q_form) { launch_pcg_fused_q_dma(s, L, k, nparts_prev, grid, tol); return; }      // whole levels and row bands
    if (L.q_form) {                                                    // q = A p is not stored but formed again
        if (whole) {
            if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused_q<true, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
            else hipLaunchKernelGGL((k_pcg_fused_q<false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        } else {
            if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused_q<true, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
            else hipLaunchKernelGGL((k_pcg_fused_q<false, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        }
    } else if (small) {
        if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused<1, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL((k_pcg_fused<1, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    } else {
        if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused<2, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL((k_pcg_fused<2, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    }
}

void launch_flow_update_fused(hipStream_t s, const LevelPtrs &L, int nlaunched, int nparts)
{
    hipLaunchKernelGGL(k_flow_update_fused, dim3(stream_grid_size(L.w, L.y1 - L.y0)), dim3(256), 0, s, L, nlaunched, nparts);
}

#ifdef OCTANE_DIAG
void launch_flow_update(hipStream_t s, const LevelPtrs &L, int nlaunched)
{
    hipLaunchKernelGGL(k_flow_update, dim3(stream_grid_size(L.w, L.y1 - L.y0)), dim3(256), 0, s, L, nlaunched);
}
#endif

// ---------------------------------------------------------------------------------------------
// Whole PCG solve in one workgroup, for the coarsest pyramid levels (<= kSmallMaxPix pixels).
//
// At 39x39 or 78x78 pixels a PCG pass is 5-7 us of launch + latency for a few hundred nanoseconds of
// work, 60 times per solve.  Here one 512-thread workgroup keeps r, x, p and the operator in registers
// (up to 12 pixels per thread), exchanges p through LDS and runs all cgiters iterations, the stop test and
// the flow update u += dx, v += dy (ref .cu:1105-1195) between workgroup barriers: one launch per solve.
// Same recurrences, same operator, same stop rule as the two-pass kernels; only the (fp64) summation
// order of the dot products differs.
// ---------------------------------------------------------------------------------------------
constexpr int kSmallThreads = 512;             // 2 waves per SIMD -> 256 VGPRs per lane
constexpr int kSmallMaxPerThread = 12;         // 15 register arrays per pixel
constexpr int kSmallMaxPix = kSmallThreads * kSmallMaxPerThread;

__device__ __forceinline__ double block_sum_1024(double v, double *scratch)   // over kSmallThreads lanes
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.;
#pragma unroll
    for (int i = 0; i < kSmallThreads / 64; i++) t += scratch[i];
    return t;
}

template <int kSmallPerThread>
__global__ __launch_bounds__(kSmallThreads) void k_pcg_solve_small(LevelPtrs L, int maxit, float tol)
{
    extern __shared__ float s_mem[];               // p_u, p_v, x_u, x_v, wx, wy: npix floats each
    __shared__ double s_red[16];
    const int w = L.w, h = L.h, pitch = L.pitch, npix = w * h;
    float *s_pu = s_mem, *s_pv = s_mem + npix, *s_xu = s_mem + 2 * npix, *s_xv = s_mem + 3 * npix;
    float *s_wx = s_mem + 4 * npix, *s_wy = s_mem + 5 * npix;
    const int tid = threadIdx.x;

    float ru[kSmallPerThread], rv[kSmallPerThread], pu[kSmallPerThread], pv[kSmallPerThread];
    float a1[kSmallPerThread], a2[kSmallPerThread], a4[kSmallPerThread], mu[kSmallPerThread], mv[kSmallPerThread];
    double d_rz = 0., d_rr = 0.;
#pragma unroll
    for (int s = 0; s < kSmallPerThread; s++) {
        const int n = tid + s * kSmallThreads;
        ru[s] = rv[s] = pu[s] = pv[s] = 0.f;
        a1[s] = a4[s] = 1.f; a2[s] = mu[s] = mv[s] = 0.f;
        if (n < npix) {
            const size_t o = (size_t)(n / w) * pitch + (n % w);
            ru[s] = L.ru[o]; rv[s] = L.rv[o];
            a1[s] = L.a1[o]; a2[s] = L.a2[o]; a4[s] = L.a4[o]; mu[s] = L.mu[o]; mv[s] = L.mv[o];
            s_wx[n] = L.wx[o]; s_wy[n] = L.wy[o];
            s_xu[n] = 0.f; s_xv[n] = 0.f;
            const float zu = mu[s] * ru[s], zv = mv[s] * rv[s];
            float t = 0.f; t += ru[s] * zu; t += rv[s] * zv; d_rz += (double)t;
            t = 0.f; t += ru[s] * ru[s]; t += rv[s] * rv[s]; d_rr += (double)t;
        }
    }
    float rz, rr;
    {   // both sums with one pair of barriers (each in the order block_sum_1024 takes)
        const double two[2] = {d_rz, d_rr};
        double tot[2];
        block_sum_multi<2, kSmallThreads>(two, s_red, tot);
        rz = (float)tot[0]; rr = (float)tot[1];
    }
    float rz_old = 0.f;
    int it = 0;
    while (rr > tol && it < maxit) {                                   // ref .cu:1131
        const float beta = (it == 0) ? 0.f : rz / rz_old;
#pragma unroll
        for (int s = 0; s < kSmallPerThread; s++) {
            const int n = tid + s * kSmallThreads;
            if (n < npix) {
                const float zu = mu[s] * ru[s], zv = mv[s] * rv[s];
                pu[s] = (it == 0) ? zu : beta * pu[s] + zu;
                pv[s] = (it == 0) ? zv : beta * pv[s] + zv;
                s_pu[n] = pu[s]; s_pv[n] = pv[s];
            }
        }
        __syncthreads();
        float qu[kSmallPerThread], qv[kSmallPerThread];
        double d_pq = 0.;
#pragma unroll
        for (int s = 0; s < kSmallPerThread; s++) {
            const int n = tid + s * kSmallThreads;
            qu[s] = qv[s] = 0.f;
            if (n < npix) {
                const int j = n / w, i = n - j * w;
                // merged border weights exactly as the two-pass kernel forms them (ref .cu:929-1001)
                const float wxc = s_wx[n], wyc = s_wy[n];
                float sumu = 0.f, sumv = 0.f;
                if (j > 0) { const float a6 = s_wy[n - w]; const float wS = (j == h - 1) ? a6 + wyc : a6; sumu += wS * s_pu[n - w]; sumv += wS * s_pv[n - w]; }
                if (i > 0) { const float a5 = s_wx[n - 1]; const float wW = (i == w - 1) ? a5 + wxc : a5; sumu += wW * s_pu[n - 1]; sumv += wW * s_pv[n - 1]; }
                sumu += a1[s] * pu[s]; sumv += a2[s] * pu[s];
                sumu += a2[s] * pv[s]; sumv += a4[s] * pv[s];
                if (i < w - 1) { const float wE = (i == 0) ? wxc + wxc : wxc; sumu += wE * s_pu[n + 1]; sumv += wE * s_pv[n + 1]; }
                if (j < h - 1) { const float wN = (j == 0) ? wyc + wyc : wyc; sumu += wN * s_pu[n + w]; sumv += wN * s_pv[n + w]; }
                qu[s] = sumu; qv[s] = sumv;
                float t = 0.f; t += pu[s] * sumu; t += pv[s] * sumv; d_pq += (double)t;
            }
        }
        const float pq = (float)block_sum_1024(d_pq, s_red);          // its barriers also order the LDS reads of p
        const float alpha = rz / pq;                                    // before the next iteration's writes
        const float nalpha = (float)(-1. * (double)alpha);
        d_rz = 0.; d_rr = 0.;
#pragma unroll
        for (int s = 0; s < kSmallPerThread; s++) {
            const int n = tid + s * kSmallThreads;
            if (n < npix) {
                s_xu[n] = alpha * pu[s] + s_xu[n];                      // own element only: no hazard
                s_xv[n] = alpha * pv[s] + s_xv[n];
                ru[s] = nalpha * qu[s] + ru[s];
                rv[s] = nalpha * qv[s] + rv[s];
                const float zu = mu[s] * ru[s], zv = mv[s] * rv[s];
                float t = 0.f; t += ru[s] * zu; t += rv[s] * zv; d_rz += (double)t;
                t = 0.f; t += ru[s] * ru[s]; t += rv[s] * rv[s]; d_rr += (double)t;
            }
        }
        rz_old = rz;
        {
            const double two[2] = {d_rz, d_rr};
            double tot[2];
            block_sum_multi<2, kSmallThreads>(two, s_red, tot);
            rz = (float)tot[0]; rr = (float)tot[1];
        }
        it++;
    }
    if (it > 0) {                                                      // ref .cu:1185-1195
#pragma unroll
        for (int s = 0; s < kSmallPerThread; s++) {
            const int n = tid + s * kSmallThreads;
            if (n < npix) {
                const size_t o = (size_t)(n / w) * pitch + (n % w);
                L.u[o] = L.u[o] + s_xu[n];
                L.v[o] = L.v[o] + s_xv[n];
                L.xu[o] = s_xu[n]; L.xv[o] = s_xv[n];                   // kept for the debug tap
            }
        }
    }
    if (tid == 0) *L.iter_total += it;
}

bool pcg_small_applicable(int w, int h) { return (long)w * h <= kSmallMaxPix; }

// dynamic LDS above 64 KiB has to be allowed per kernel; done once per plan, outside any stream capture
void pcg_small_configure()
{
    (void)hipFuncSetAttribute((const void *)k_pcg_solve_small<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * kSmallMaxPix * 4);
    (void)hipFuncSetAttribute((const void *)k_pcg_solve_small<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * kSmallMaxPix * 4);
    (void)hipFuncSetAttribute((const void *)k_pcg_solve_small<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * kSmallMaxPix * 4);
}

void launch_pcg_solve_small(hipStream_t s, const LevelPtrs &L, int maxit, float tol)
{
    const size_t lds = (size_t)6 * L.w * L.h * sizeof(float);       // <= 147456 B of the CU's 160 KiB
    const int per = (L.w * L.h + kSmallThreads - 1) / kSmallThreads;
    if (per <= 3) hipLaunchKernelGGL(k_pcg_solve_small<3>, dim3(1), dim3(kSmallThreads), lds, s, L, maxit, tol);
    else if (per <= 6) hipLaunchKernelGGL(k_pcg_solve_small<6>, dim3(1), dim3(kSmallThreads), lds, s, L, maxit, tol);
    else hipLaunchKernelGGL(k_pcg_solve_small<12>, dim3(1), dim3(kSmallThreads), lds, s, L, maxit, tol);
}

int pcg_b_grid_size(int w, int h) { return pass_b_grid_size(w, h); }

}  // namespace octane
