// pcg_fused_q_dma.hip -- the q-recomputing PCG iteration (k_pcg_fused_q of pcg_kernels.hip, whole levels) with the p_{k-1} tile of
// the NEXT tile fetched by LDS-DMA (global_load_lds_dwordx4) while the current tile is in its phase 2, instead of global -> registers
// -> LDS at the start of every tile.  Same arithmetic, same tile walk, same partial sums: same bits as the register-staged kernel.
// Where a wave's time went in that kernel (tools/probe_stamps.py): ~6.9 k of ~35 k cycles per tile in "stage p" -- issue 11 loads,
// wait for all of them, write LDS, barrier -- a phase that moves 16 of the tile's 52-76 B/pixel and overlaps with nothing.
#include "vof_kernels.hpp"
#include "device_util.hpp"
// Compile-time switches of this kernel's text (tools/sweep_variants.sh builds and times all 32 combinations with both scheduling
// strategies; the kernel's speed moves by +-5 % with how the text reaches the register allocator, so the choice is measured, not
// argued).  Sweep of round 2, ms per finest-level launch at 5000^2 / 2000^2 on one box: neither border-free phase 0.318 / 0.0619,
// phase 1 only 0.298 / 0.0611, both 0.2952 / 0.0607, both + rotated columns 0.2960 / 0.0578 (the defaults below, max-ilp scheduling).
#ifndef Q_P1
#define Q_P1 1          // border-free form of phase 1 for tiles strictly inside the frame
#endif
#ifndef Q_P2
#define Q_P2 1          // the same for phase 2
#endif
#ifndef Q_LB
#define Q_LB 1          // second launch bound (minimum waves per SIMD); tests/test_capi_cpu.py checks the built kernel's occupancy instead
#endif
#ifndef Q_A2BR
#define Q_A2BR 1        // the a2 load behind the run-time streaming switch (0: a plain load; measured both ways, see Q_VMN)
#endif
#ifndef Q_VMN
#define Q_VMN 1         // phase 0 waits for the DMA only (vmcnt = the own loads issued since), not for the tile's own loads too:
                        // 0.2947 -> 0.2910 ms per launch at 5000^2 with the a2 switch, 0.3044 -> 0.2930 without it
#endif
#ifndef Q_VMCNT_W
#define Q_VMCNT_W 18    // own register loads of a tile issued after the DMA: 9 per slot (r_u r_v a1 a4 a2 wx wy wy-above wx-west) ...
#endif
#ifndef Q_VMCNT_U
#define Q_VMCNT_U 10    // ... 5 per slot where the weights are the constant -1
#endif
#ifndef Q_REV
#define Q_REV 1         // launches with odd k walk the tiles BACKWARDS (the same tile -> workgroup map, mirrored; 2: the even ones do; 0: none):
                        // a launch starts on the tiles whose r, p and operator lines the previous launch touched last, so its first
                        // fronts read them out of the 256 MB Infinity Cache instead of HBM.  Round 6, 5000^2 (profiles/r6_reverse_walk.txt):
                        // 291.5 -> 287.0 us per launch; with Q_TAILC 285.3-286.1 (-2.0 %); nothing at 2500^2 / 2000^2 (not HBM-bound).
#endif
#ifndef Q_TAILC
#define Q_TAILC 6       // r_k / p_k of a workgroup's last Q_TAILC tiles are stored WITHOUT the streaming hint, so that they are still in the
                        // cache when the next launch starts there (~4.9 Mpixel of a launch's tail fit beside what else the tail allocates);
                        // only on walks of Q_TAILMIN rounds or more: on a level of 6 rounds (2500^2) it is +3 % (the whole level allocates)
#endif
#ifndef Q_TAILMIN
#define Q_TAILMIN 12
#endif
// Measured and not kept (round 6): the streaming hint on the tail tiles' p_{k-1} DMA (their lines are dead afterwards) gives the reverse
// walk's gain back (291.6 us); on the own r loads as well it is the run-time-policy trap of EXPERIMENTS 8 (+8 %).
#ifndef Q_RROT
#define Q_RROT 1         // rotate the tile columns by the tile row where the host asks for it (L.row_rot); 0: never (A/B builds)
#endif
#define Q_STR2(x) #x
#define Q_STR(x) Q_STR2(x)
#ifndef Q_ROT
#define Q_ROT 1         // rotate the tile columns by the round number when the column count divides the grid
#endif

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
// A p) as the correctly rounded float reciprocal.  The reference rounds 1./M through double
// first; the two agree except when the double quotient sits exactly on a float rounding
// boundary (probability ~2^-29 per value, 1 ulp then).  rcp_exact (device_util.hpp) gives the bits of 1.0f / diag in three
// instructions instead of the division's eleven: four reciprocals per pixel and launch, 32 of the kernel's 236 lane-instructions.
__device__ __forceinline__ float direction(float r, float pold, float diag, float beta, bool first)
{
    float z = rcp_exact(diag) * r;
    return first ? z : beta * pold + z;
}

constexpr int kQTY = 2 * kTileY;            // 16 tile rows
constexpr int kQCols = kTileX + 16;         // LDS row: 8 floats of margin either side of the 128 tile columns
constexpr int kQOff = 8;                    // LDS column of the tile's first pixel

struct QCoef { float a1[4], a2[4], a4[4], wx[4], wy[4], wys[4]; float wxw; };

// the 5-point operator on one float4 group at frame position (x0, y), from an LDS tile whose row `lrow` / column `lcol`
// hold the group's own pixels (same arithmetic, in the same order, as every other form of A p in this file).
// INTERIOR: the group and its four neighbours lie strictly inside the frame -- no merged border weight, no missing neighbour:
// the same sums without the 21 compare / select instructions per pixel that only the frame's outermost pixels need.
template <bool INTERIOR>
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
        const float wS = (!INTERIOR && y == h - 1) ? c.wys[e] + c.wy[e] : c.wys[e];
        const float wW = (!INTERIOR && i == w - 1) ? a5 + c.wx[e] : a5;
        const float wE = (!INTERIOR && i == 0) ? c.wx[e] + c.wx[e] : c.wx[e];
        const float wN = (!INTERIOR && y == 0) ? c.wy[e] + c.wy[e] : c.wy[e];
        float sumu = 0.f, sumv = 0.f;
        if (INTERIOR || y > 0) { sumu += wS * su[e]; sumv += wS * sv[e]; }
        if (INTERIOR || i > 0) { sumu += wW * pwu; sumv += wW * pwv; }
        sumu += c.a1[e] * cu[e]; sumv += c.a2[e] * cu[e];
        sumu += c.a2[e] * cv[e]; sumv += c.a4[e] * cv[e];
        if (INTERIOR || i < w - 1) { sumu += wE * peu; sumv += wE * pev; }
        if (INTERIOR || y < h - 1) { sumu += wN * nu[e]; sumv += wN * nv[e]; }
        qu[e] = sumu; qv[e] = sumv;
    }
}


constexpr int kRingGroups = 100;          // 34 above, 34 below, 16 left, 16 right of a 128 x 16 tile
constexpr int kRingOps = 9;

// every float4 group the staging touches (2 rows above / below, one group left / right of the tile) lies inside the level
__device__ __forceinline__ bool tile_is_interior(int tx0, int ty0, int w, int h, int y1)
{
    // (strictly inside on the right: the last pixel of the ring group east of the tile is not the frame's last column, so no pixel the
    // tile touches -- own or ring -- has a merged border weight or a missing neighbour: stencil_group<true>)
    // (a row band: all 16 rows of the tile are the band's own; the rows staged from beyond its edges are the neighbouring band's)
    return tx0 >= 4 && tx0 + kTileX + 4 < w && ty0 >= 2 && ty0 + kQTY + 2 <= h && ty0 + kQTY <= y1;
}

// LDS-DMA of the staged p tile: one instruction per staged row and component, lanes 0..33 <-> the row's 34 float4 groups (544
// contiguous bytes at column kQOff - 4 of the padded LDS row), rows dealt over the four waves.  The destination is M0 + lane * 16.
// A row band takes the staged rows above its first / below its last row from the neighbouring band's planes (up_* / dn_*; the
// band's own planes for a whole level).
template <bool BANDED>
__device__ __forceinline__ void dma_p_tile(const float *pin_u, const float *pin_v, const float *up_u, const float *up_v, const float *dn_u,
                                           const float *dn_v, int y0, int y1, float *s_ou, float *s_ov, int tx0, int ty0, int pitch, int lane, int wv)
{
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned base_u = (unsigned)(unsigned long)(lds_float *)s_ou, base_v = (unsigned)(unsigned long)(lds_float *)s_ov;
    if (lane < kTileX / 4 + 2) {
        const int x0 = tx0 + 4 * (lane - 1);
        for (int r = wv; r < kQTY + 4; r += 4) {
            const int y = ty0 + r - 2;
            const size_t o = (size_t)y * pitch + x0;
            const float *gu = ((BANDED && y < y0) ? up_u : (BANDED && y >= y1) ? dn_u : pin_u) + o;
            const float *gv = ((BANDED && y < y0) ? up_v : (BANDED && y >= y1) ? dn_v : pin_v) + o;
            const unsigned du = __builtin_amdgcn_readfirstlane(base_u + (unsigned)(r * kQCols + kQOff - 4) * 4u);
            const unsigned dv = __builtin_amdgcn_readfirstlane(base_v + (unsigned)(r * kQCols + kQOff - 4) * 4u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gu), "s"(du) : "memory");
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gv), "s"(dv) : "memory");
        }
    }
}

// LDS-DMA of the ring groups' operands: lane l of instruction i carries ring group 64 i + l (the ring group of thread 64 i + l in
// phase 1); operand planes dealt over the four waves.  NOPS = 5 where the weights are the constant -1.
__device__ __forceinline__ void dma_one(const float *g, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}
// In a row band the residual of a ring row beyond the band's edge is the neighbouring band's (rup / rdn) and so is wy of the row
// above the upper ring row (wy_up); every other operand of the ring rows is the band's own (its assembly covers one halo row).
struct RingBand { const float *rup_u, *rup_v, *rdn_u, *rdn_v, *wy_up; int y0, y1; };
template <int NOPS, bool BANDED>
__device__ __forceinline__ void dma_ring(const float *const (&plane)[kRingOps], const int (&shift)[kRingOps], const RingBand &rb, float *s_ring,
                                         int tx0, int ty0, int pitch, int lane, int wv)
{
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned base = (unsigned)(unsigned long)(lds_float *)s_ring;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int j = 64 * i + lane;                       // ring group
        int gx, gy;
        if (j < 34) { gx = j - 1; gy = -1; }
        else if (j < 68) { gx = j - 35; gy = kQTY; }
        else if (j < 84) { gx = -1; gy = j - 68; }
        else { gx = kTileX / 4; gy = j - 84; }
        if (j < kRingGroups) {
            const int y = ty0 + gy;
            const long o = (long)y * pitch + tx0 + 4 * gx;
#pragma unroll
            for (int op = 0; op < NOPS; op++) {
                if ((op & 3) == wv) {                      // uniform: this wave's operands
                    const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)((op * kRingGroups + 64 * i) * 16));
                    const float *pl = plane[op];
                    if (BANDED && op == 0) pl = y < rb.y0 ? rb.rup_u : y >= rb.y1 ? rb.rdn_u : pl;
                    if (BANDED && op == 1) pl = y < rb.y0 ? rb.rup_v : y >= rb.y1 ? rb.rdn_v : pl;
                    if (BANDED && op == 7) pl = y < rb.y0 ? rb.wy_up : pl;
                    dma_one(pl + o + shift[op], dst);
                }
            }
        }
    }
}

}  // namespace

// Two waves per SIMD (two workgroups per CU) is the design point.  The register allocator does not know that: one experimental build
// came out with 256 VGPRs + 2 AGPRs, ran at one workgroup per CU (0.377 instead of 0.30 ms per launch), and nothing but the
// "Occupancy" line of the assembly said so.  The Makefile keeps the compiler's resource remarks of this file
// (pcg_fused_q_dma.usage.txt) and tests/test_capi_cpu.py fails when the kernel's occupancy is not 2.
template <bool UNITW, bool BANDED>
__global__ __launch_bounds__(256, Q_LB) void k_pcg_fused_q_dma(LevelPtrs L, int k, int nparts_prev, float tol)
{
    constexpr int TY = kQTY, TX = kTileX;
    __shared__ __attribute__((aligned(16))) float s_ou[(TY + 4) * kQCols], s_ov[(TY + 4) * kQCols];   // p_{k-1}: rows ty0-2 .. ty0+TY+1
    // p_k: rows ty0-1 .. ty0+TY, two buffers used alternately -- a workgroup's fast waves may stage and compute the next tile
    // while its slow ones still read this one's p_k in phase 2, which saves the barrier at the end of a tile
    constexpr int NSZ = (TY + 2) * kQCols;
    __shared__ __attribute__((aligned(16))) float s_nu2[2 * NSZ], s_nv2[2 * NSZ];
    __shared__ double s_red[4 * kPartKinds];
    // operands of the 100 ring groups of a tile (r_u r_v a1 a4 a2 wx wy, wy of the row above, wx of the group to the west), fetched
    // by LDS-DMA together with the p tile: [operand][ring group]
    __shared__ __attribute__((aligned(16))) float s_ring[kRingOps * kRingGroups * 4];
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
    // When the number of tile columns divides the grid (2000 or 2048 pixels wide: 16 columns, 512 workgroups) the round-robin walk
    // gives a workgroup the same column in every round, and the 1 / 8 of the workgroups that own the frame's first and last column
    // -- the tiles that stage through registers and run the bordered operator -- finish last in every launch.  The columns of a
    // tile row are then rotated by the round number (a permutation within the row, so every tile is still done exactly once).
    const bool rotate = Q_ROT && tr.step == (int)gridDim.x && tr.step % tiles_x == 0 && tiles_x > 1;
    const bool rev = Q_REV != 0 && ((k & 1) != 0) == (Q_REV == 1);
    // Otherwise the columns of tile row r may be rotated by r -- where the host found that this spreads the border-column tiles (register-
    // staged, bordered operator) more evenly over the workgroups (pcg_row_rotation, pcg_kernels.hip): at 5000^2 64 of the 512 workgroups
    // own ALL left-border tiles and finish last in every launch; rotated, no workgroup has more than two (-1.1 % per launch).
    const bool rowrot = Q_RROT && !rotate && L.row_rot != 0;
    int parity = 0, round = 0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // p_{k-1} of an interior tile (every staged float4 group inside the level) comes by LDS-DMA, issued one phase ahead: for the
    // first tile here, for every later one at the start of the previous tile's phase 2 (s_ou / s_ov are free then: phase 1 has
    // read them and the barrier before phase 2 has been passed).  Nothing else is loaded between the DMA and the wait for it at
    // the tile's phase 0, so the wait does not hold anything else up.  Border tiles stage through registers as before.
    // ring operands: r_u r_v a1 a4 a2 | wx wy wy(row above) wx(group to the west); the last four only where the weights vary
    const float *const ring_plane[kRingOps] = {rin_u, rin_v, L.a1, L.a4, L.a2, L.wx, L.wy, L.wy, L.wx};
    const int ring_shift[kRingOps] = {0, 0, 0, 0, 0, 0, 0, -pitch, -4};
    const float *const up_u = L.pup_u[(k + 2) % 3], *const up_v = L.pup_v[(k + 2) % 3];      // the neighbouring bands' p_{k-1} (row bands only)
    const float *const dn_u = L.pdn_u[(k + 2) % 3], *const dn_v = L.pdn_v[(k + 2) % 3];
    const RingBand rb = {L.rup_u[ko], L.rup_v[ko], L.rdn_u[ko], L.rdn_v[ko], L.wy_up, y0, y1};
    bool dma_cur = false;
    if (!first && tr.first < tr.end) {
        const int ft = rev ? ntiles - 1 - tr.first : tr.first;
        const int ftx0 = ((ft % tiles_x + (rowrot ? ft / tiles_x : 0)) % tiles_x) * TX, fty0 = y0 + (ft / tiles_x) * TY;      // round 0: no rotation
        dma_cur = tile_is_interior(ftx0, fty0, w, h, y1);
        if (dma_cur) {
            dma_p_tile<BANDED>(pin_u, pin_v, up_u, up_v, dn_u, dn_v, y0, y1, s_ou, s_ov, ftx0, fty0, pitch, lane, wv);
            dma_ring<UNITW ? 5 : kRingOps, BANDED>(ring_plane, ring_shift, rb, s_ring, ftx0, fty0, pitch, lane, wv);
        }
    }
    for (int t = tr.first; t < tr.end; t += tr.step, parity ^= 1, round++) {
        float *const s_nu = s_nu2 + parity * NSZ, *const s_nv = s_nv2 + parity * NSZ;
        const int tt = rev ? ntiles - 1 - t : t;
        const bool tailc = Q_TAILC > 0 && tr.end - tr.base >= Q_TAILMIN * tr.step && t + Q_TAILC * tr.step >= tr.end;
        const int tx0 = ((tt % tiles_x + (rotate ? round : (rowrot ? tt / tiles_x : 0))) % tiles_x) * TX, ty0 = y0 + (tt / tiles_x) * TY;
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
            *(float4 *)c.a2 = Q_A2BR ? ld4_if(at(L.a2, o), L.nt_hints & 256) : ld4(at(L.a2, o));
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
        if (!first && dma_cur) {
            // The DMA was issued before this tile's own loads and loads return in order: once no more than the own loads issued since
            // (5 or 9 per slot, all unconditional) are outstanding, the DMA has landed -- the ring groups, which need nothing else,
            // then run under the own loads' latency.  (Q_VMN 0: wait for everything.)
            // `s_setprio 0` (the priority every wave has anyway) directly after the wait is the MARKER tools/check_dma_wait.py looks for:
            // it proves on the built code object that on every path at least N vector-memory instructions lie between the last
            // global_load_lds and this wait (tests/test_capi_cpu.py; the compiler does not track the inline-asm DMA).
            if (Q_VMN) { if (UNITW) asm volatile("s_waitcnt vmcnt(" Q_STR(Q_VMCNT_U) ")\n\ts_setprio 0" ::: "memory");
                         else asm volatile("s_waitcnt vmcnt(" Q_STR(Q_VMCNT_W) ")\n\ts_setprio 0" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_setprio 0" ::: "memory");
            __syncthreads();
        } else if (!first) {
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
        // ---- phase 1: the ring group (one each for the first 100 threads), then the two tile groups.  Twice in the text: for a tile
        // strictly inside the frame (INT: every group valid, no border case in the operator; these are the tiles whose staging came by
        // DMA) and for the tiles along the frame's border.
        const bool interior = dma_cur && !first;
        if (Q_P1 && interior) {
#define Q_INT true
#include "pcg_fused_q_phase1.inc"
#undef Q_INT
        } else {
#define Q_INT false
#include "pcg_fused_q_phase1.inc"
#undef Q_INT
        }
        __syncthreads();
        // ---- the next tile's p_{k-1}: LDS-DMA now, to land under phase 2 and the next tile's own loads
        {
            const int tn = t + tr.step;
            bool dma_next = false;
            if (!first && tn < tr.end) {
                const int tnn = rev ? ntiles - 1 - tn : tn;
                const int ntx0 = ((tnn % tiles_x + (rotate ? round + 1 : (rowrot ? tnn / tiles_x : 0))) % tiles_x) * TX, nty0 = y0 + (tnn / tiles_x) * TY;
                dma_next = tile_is_interior(ntx0, nty0, w, h, y1);
                if (dma_next) {
                    dma_p_tile<BANDED>(pin_u, pin_v, up_u, up_v, dn_u, dn_v, y0, y1, s_ou, s_ov, ntx0, nty0, pitch, lane, wv);
                    dma_ring<UNITW ? 5 : kRingOps, BANDED>(ring_plane, ring_shift, rb, s_ring, ntx0, nty0, pitch, lane, wv);
                }
            }
            dma_cur = dma_next;
        }
        // ---- phase 2: q_k on the tile and the partial sums (q_k is not stored: the next launch forms it again)
        if (active) {
            if (Q_P2 && interior) {
#define Q_INT true
#include "pcg_fused_q_phase2.inc"
#undef Q_INT
            } else {
#define Q_INT false
#include "pcg_fused_q_phase2.inc"
#undef Q_INT
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


void launch_pcg_fused_q_dma(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol)
{
    const bool whole = L.nbands == 1 && L.y0 == 0 && L.y1 == L.h;
    if (whole) {
        if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused_q_dma<true, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL((k_pcg_fused_q_dma<false, false>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    } else {                   // a row band (vof_tiled.hip)
        if (L.unit_w) hipLaunchKernelGGL((k_pcg_fused_q_dma<true, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
        else hipLaunchKernelGGL((k_pcg_fused_q_dma<false, true>), dim3(grid), dim3(256), 0, s, L, k, nparts_prev, tol);
    }
}

}  // namespace octane
