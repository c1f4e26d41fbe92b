// pcg_fused_q_diag.hip -- DIAGNOSTIC copy of the q-recomputing PCG iteration (k_pcg_fused_q of pcg_kernels.hip), never on the
// product path.  The production kernel is sensitive to everything around its loads (a run-time choice of cache policy per
// load -- `nt ? ld4_nt(p) : ld4(p)` on r, a1, a4, wx, wy and p -- cost it 25 %, 0.326 -> 0.417 ms: the loads of a tile were no
// longer issued back to back), so experiments live here:
//   * STAMP: every wave reads the shader clock at the seams of a tile and adds up where its time goes (tools/probe_stamps.py);
//   * the slab tile walk (xcd >= 5: device_util.hpp, SlabWalk), selected for octane_vof_plan_probe by
//     octane_vof_tune(plan, "q_diag", 1) (tools/probe_q.py).
// Results of the diagnostic kernel are the production kernel's (same arithmetic); its timings are not.
#include "vof_kernels.hpp"
#include "device_util.hpp"

namespace octane {
namespace {

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
    float z = (1.0f / diag) * r;
    return first ? z : beta * pold + z;
}

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

}  // namespace

// Diagnostic build (STAMP, launched by octane_vof_plan_probe_stamps only): every wave reads the shader clock at the seams of a
// tile and adds up where its time goes; one atomic add per wave and seam at the end.  [0] loads of the own groups issued, [1] p
// staged (includes the wait for p), [2] first barrier, [3] ring group, [4] own groups, [5] second barrier, [6] phase 2,
// [7] tiles, [8] everything before the first tile (fold of the partial sums), [9] the final reduction
__device__ unsigned long long g_q_stamps[16];

template <bool UNITW, bool BANDED, bool STAMP = false>
__global__ __launch_bounds__(256) void k_pcg_fused_q_diag(LevelPtrs L, int k, int nparts_prev, float tol)
{
    constexpr int TY = kQTY, TX = kTileX;
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
    if (STAMP) st_last = clock64();
#define Q_STAMP(i) do { if (STAMP) { const unsigned long long now_ = clock64(); st_acc[i] += now_ - st_last; st_last = now_; } } while (0)
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
    // xcd_bands >= 5: vertical slabs per XCD inside super-rows of (xcd_bands & 63) tile rows (device_util.hpp, SlabWalk)
    const bool slabs = L.xcd_bands >= 5 && (gridDim.x & 7) == 0 && tiles_x >= 8;
    SlabWalk sw = slab_walk_begin(tiles_x, tiles_y, slabs ? (L.xcd_bands & 63) : 1);
    int parity = 0;
    Q_STAMP(8);
    for (int t = tr.first; ; t += tr.step, parity ^= 1) {
        int tcol, trow;
        if (slabs) {
            if (!slab_walk_tile(sw, tcol, trow)) break;
            slab_walk_next(sw);
        } else {
            if (t >= tr.end) break;
            tcol = t % tiles_x; trow = t / tiles_x;
        }
        if (STAMP) st_acc[7] += 1;
        float *const s_nu = s_nu2 + parity * NSZ, *const s_nv = s_nv2 + parity * NSZ;
        const int tx0 = tcol * TX, ty0 = y0 + trow * TY;
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
        Q_STAMP(0);
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
            Q_STAMP(1);
            __syncthreads();
            Q_STAMP(2);
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
            if (STAMP && sl == 0) Q_STAMP(3);
        }
        Q_STAMP(4);
        __syncthreads();
        Q_STAMP(5);
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
                            const float iu = 1.0f / c3[slot].a1[e], iv = 1.0f / c3[slot].a4[e];
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
        Q_STAMP(6);
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
    if (STAMP) {
        Q_STAMP(9);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int j = 0; j < 10; j++) atomicAdd(&g_q_stamps[j], st_acc[j]);
        }
    }
#undef Q_STAMP
}

// Diagnostic (tools/probe_stamps.py): one stamped launch of the q-recomputing kernel on whatever the planes hold
int pcg_fused_q_stamps(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol, unsigned long long *out16)
{
    unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_q_stamps), zero, sizeof zero, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused_q_diag<true, false, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    else hipLaunchKernelGGL((k_pcg_fused_q_diag<false, false, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    if (hipMemcpyFromSymbolAsync(out16, HIP_SYMBOL(g_q_stamps), sizeof zero, 0, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}


// the unstamped diagnostic kernel in the place of the production one (whole-level launches only)
void launch_pcg_fused_q_diag(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol)
{
    if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused_q_diag<true, false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    else hipLaunchKernelGGL((k_pcg_fused_q_diag<false, false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
}

}  // namespace octane
