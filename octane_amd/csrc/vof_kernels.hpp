// vof_kernels.hpp -- launch interface between the host driver (vof_plan.hip) and the
// gfx950 kernels (vof_kernels.hip, pcg_kernels.hip).  Internal; the public boundary is
// include/octane_vof.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace octane {

// Environment knobs.  The PRODUCT library reads only the ones include/octane_vof.h documents (OCTANE_VOF_CACHE, OCTANE_VOF_BANDS,
// OCTANE_TILED_TRANSPORT, OCTANE_TILED_SELFCHECK, OCTANE_MP_TIMEOUT_S, OCTANE_PIX2UV_FMAD, OCTANE_TUNE_MIN_BAND_PIXELS,
// OCTANE_TUNE_PERSIST_MAXG and the bisect pair OCTANE_TUNE_Q_DMA / OCTANE_TUNE_PERSIST).  Every other OCTANE_TUNE_* variable is a
// developer knob of the DIAGNOSTIC library (make DIAG=1, liboctane_vof_diag.so: what tools/ loads) and does not exist in the
// product: a stray variable in a production environment cannot change which kernels run.
// Solo-band timing (vof_tiled.hip, diagnostic library only) runs a band's launch sequence on inconsistent neighbour data; a negative
// tolerance then keeps every launch of the fused PCG kernels working whatever the sums say (a NaN included), so that the timeline is
// the real one.  In the product the macro is `false` and the kernels' text is what it was.
#ifdef OCTANE_DIAG
#define OCT_STOP_HELD_OPEN(tol) ((tol) < 0.f)
#else
#define OCT_STOP_HELD_OPEN(tol) false
#endif

inline const char *tune_env(const char *name)
{
#ifdef OCTANE_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// A 2-D float plane in HBM: w x h pixels, rows `pitch` floats apart (pitch % 64 == 0 for
// planes the library owns, so every row starts on a 256-byte boundary).
struct Plane {
    float *p;
    int w, h, pitch;
};

constexpr int kMaxChan = 3;
constexpr int kMaxParts = 2048;      // upper bound on persistent blocks == reduction partials
                                     // (256 CUs x 8 resident 256-thread workgroups)
constexpr int kMaxBands = 8;         // row bands of one frame solved side by side (vof_tiled.hip)
// A band's partial block: kPartKinds arrays of kMaxParts doubles.  rz, rr: direct sums over the current residual (written
// by the assembly, pass B and the fused pass); pq: p.q; the last four only by the fused pass (pcg_kernels.hip):
// q.z, q.M^-1 q, r.q and q.q, from which the NEXT residual's r.z and r.r follow without another sweep.
constexpr int kPartKinds = 7;
constexpr int kPartBlock = kPartKinds * kMaxParts;   // doubles per block; every band owns TWO blocks (see k_pcg_fused)
constexpr int kPartRz = 0, kPartRr = kMaxParts, kPartPq = 2 * kMaxParts, kPartQz = 3 * kMaxParts, kPartQmq = 4 * kMaxParts,
              kPartRq = 5 * kMaxParts, kPartQq = 6 * kMaxParts;
constexpr int kBandAlign = 32;       // band boundaries are multiples of this many rows (a whole number of pass A tiles)

// PCG tile geometry: 256 threads, each owning 4 consecutive pixels of one row.
constexpr int kTileX = 128;
constexpr int kTileY = 8;

// Scalars of one PCG solve, double-buffered by iteration parity (see pcg_kernels.hip).
struct PcgState {
    float rz;        // r.z of the last completed iteration (denominator of the next beta)
    int stopped;     // sticky: the reference's while-condition failed
    int iters;       // iterations executed in this solve
    int pad;
};

struct LevelPtrs {       // everything one pyramid level's solve touches
    int w, h, pitch, nc;
    size_t cstride;                 // floats between channel planes
    const float *img1, *img2;       // nc planes each
    const float *gx1, *gy1, *gx2, *gy2, *gxx, *gxy, *gyy;
    float *u, *v;
    const float *ut, *vt;           // first-guess hint (only read when lambdac != 0)
    float *a1, *a2, *a4, *wx, *wy;  // per-pixel operator coefficients
    float *mu, *mv;                 // Jacobi preconditioner 1/a1, 1/a4 as the reference rounds it (ref .cu:141-149)
    float *ru, *rv, *qu, *qv, *xu, *xv;             // r (starts as rhs), q = A p, x
    float *pu[2], *pv[2];                           // search direction, ping-pong by iteration parity:
                                                    // pass A(k) reads p[k&1] (halo too) and writes p[(k+1)&1]
    double *part_rz, *part_rr, *part_pq;            // where THIS launch writes its per-workgroup partials (= own block + kind)
    double *part_own;                               // base of this band's two partial blocks
    // Fused one-kernel-per-iteration PCG (k_pcg_fused): r and q are double-buffered like p, because a workgroup reads
    // the old values of its neighbours' pixels while those workgroups write the new ones.  r_k is in rb[k & 1] (r_0 =
    // the rhs, written by the assembly into rb[0] == ru/rv), q_k in qb[k & 1].
    float *rb_u[2], *rb_v[2], *qb_u[2], *qb_v[2];
    float *pf_u[3], *pf_v[3];       // p of the fused kernel: p_k in pf[k % 3]; three halves because with defer_x every second
                                    // launch applies two x updates at once and reads p_{k-2} besides p_{k-1}
    const float *qup_u[2], *qup_v[2], *qdn_u[2], *qdn_v[2];   // the q planes rows y0-1 / y1 are read from (neighbouring bands)
    // q-recomputing fused PCG in row bands: the neighbours' planes, read in place on the two rows beyond a band edge
    const float *rup_u[2], *rup_v[2], *rdn_u[2], *rdn_v[2];   // r double buffer of the band above / below
    const float *pup_u[3], *pup_v[3], *pdn_u[3], *pdn_v[3];   // p triple buffer of the band above / below
    const float *wy_up;                                       // wy of the band above (its row y0 - 2 is not assembled here)
    int q_form;                     // the host's choice for this level: 1 = k_pcg_fused_q (q recomputed), 0 = q stored
    // Row band of the level this launch works on (vof_tiled.hip); a plain plan has one band covering the frame.
    // Planes are always addressed with frame coordinates: a band's neighbours' rows exist in its planes as halos.
    int y0, y1;                     // rows this band owns: PCG passes, flow update and the dot products cover [y0, y1)
    int ya0, ya1;                   // rows the assembly fills: the owned rows plus one halo row on inner edges
    int nbands;                     // reductions fold the partial blocks of all bands, in band order
    const double *band_parts[kMaxBands];            // every band's partial block ([rz|rr|pq]); other bands' blocks live
                                                    // in their own memory (peer-mapped when on another device)
    const float *ru_up, *rv_up, *ru_dn, *rv_dn;     // the r planes pass A reads rows y0-1 / y1 from: the neighbouring
                                                    // bands' own planes (this band's for a plain plan)
    PcgState *st;
    float *alpha;                   // alpha of the last two iterations (pass B of iteration k writes alpha[k&1])
    int defer_x;                    // fold x += alpha p of two iterations into every second pass B
    long long *iter_total;
    int reverse_b;                  // pass B walks the frame backwards (Infinity-Cache reuse)
    int nt_hints;                   // bit mask of streaming-load/store hints (tuning)
    int xcd_bands;                  // 1: give each XCD (blockIdx % 8) one contiguous band of the frame; 2 (fused PCG): one run of tiles per workgroup
    int unit_w;                     // this linearisation has al1 == 1: wx == wy == -1 everywhere, pass A need not read them
    int lean;                       // the fused kernels are the only readers: the assembly skips the planes they never read
                                    // (mu, mv; wx, wy while unit_w) and the flow update does not write x back
    int row_rot;                    // LDS-DMA PCG kernel: the tile columns of tile row r are rotated by r (a permutation within the row) -- set by
                                    // the host where it spreads the border-column tiles more evenly over the workgroups (pcg_row_rotation)
    int no_dma;                     // row bands whose first-contact self-check (vof_tiled.hip) found LDS-DMA from the neighbouring band's
                                    // memory wanting: q-form launches take the register-staged kernel (host-side dispatch only)
};

struct AssembleParams {
    double al1, alpha, loa;   // GNC weight, alpha, lambda/alpha
    double ralpha;            // the correctly rounded 1 / alpha (host division), for the three-instruction division by alpha
    float lambdac;
    int dozim;
    int fast_math;            // bit 0: x / alpha as x * ralpha + one exact residual step; bit 1: 1 / (s + 1) by v_rcp_f64 + two Newton
                              // steps; bit 2: 1 / sqrt(x + 1e-6) by v_rsq_f64 + two Newton steps.  Each bit is set only after the
                              // form has reproduced the IEEE sequence of the reference on EVERY float input (assemble_math_selftest)
};
// Exhaustive device self-test of the three fast forms above against the divisions / square roots the reference's expressions compile
// to, for this alpha: out8 = {patterns, mismatches} x {x / alpha over all finite floats x; 1 / (s + 1) over all floats s >= 0;
// 1 / sqrt(x + 1e-6) over all floats x >= 0}, [6] = a mismatching bit pattern, [7] = its test.  Returns 0 when it ran.
int  assemble_math_selftest(hipStream_t s, double alpha, unsigned long long *out8);
// the bits of AssembleParams::fast_math that are safe for this alpha on the current device (runs the self-test once per process and
// alpha; OCTANE_TUNE_ASM_FAST=0 turns all of them off)
int  assemble_fast_math_bits(double alpha);

void launch_copy2d(hipStream_t s, const float *src, int spitch, float *dst, int dpitch, int w, int h);
void launch_scale_copy2d(hipStream_t s, const float *src, int spitch, float *dst, int dpitch, int w, int h, float scale);
void launch_blur_rows_sampled(hipStream_t s, const float *src, int sw, int sh, int spitch,
                              float *dst, int dw, int dpitch, const float *gk, int fs, float factor);
void launch_blur_cols_sampled(hipStream_t s, const float *src, int sw, int sh, int spitch,
                              float *dst, int dh, int dpitch, const float *gk, int fs, float factor, float postscale, int do_scale);
void launch_gradient(hipStream_t s, const float *f, float *gx, float *gy, int w, int h, int pitch, int nc, size_t cstride);
void launch_upsample(hipStream_t s, const float *coarse, int cw, int ch, int cpitch,
                     float *fine, int fw, int fh, int fpitch, float sf);
void set_max_blocks(int n);
void set_q_dma(int v);                   // the q-form kernel with LDS-DMA staging of p (pcg_fused_q_dma.hip)
void launch_pcg_fused_q_dma(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol);
// 1 where rotating the tile columns of tile row r by r lowers the largest number of border-column tiles (frame's first / last tile column:
// register-staged, bordered operator, ~1 us more than an interior tile) any ONE workgroup of a `grid`-workgroup launch walks; host arithmetic
int  pcg_row_rotation(int w, int rows, int grid, int walk_mode);
int  pcg_row_rotation_count(int w, int rows, int grid, int walk_mode, int *out3);   // the same + {tile columns, max border tiles per workgroup plain, rotated}
void set_q_diag(int v);                  // diagnostics: the q-form kernel's copy in pcg_fused_q_diag.hip instead of the production one
void launch_pcg_fused_q_diag(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol);
int  pcg_fused_q_stamps(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol, unsigned long long *out16);   // diagnostic
void set_grid_multiple(int m);           // tuning knob (<= kMaxParts)
int  grid_multiple();
int  balanced_grid(long work_items);   // persistent grid: every block gets the same number of items (+-1)
int  pcg_grid_size(int w, int h);
int  pcg_grid_size_unit_w(int w, int h);     // pass A grid of the unit-weight (first GNC step) launches
void set_unit_w_cap(int c);
void set_pass_caps(int cap_a, int cap_b);   // residency caps of the pass A / pass B persistent grids (defaults 768 / 1024)
int  pcg_band_grid_size(int w, int rows);   // pass A grid for a row band (always the 128 x 16 tiled form)
int  pcg_b_grid_size(int w, int h);
void set_pass_a_variant(int v);       // tuning knob: tile rows per thread 1 | 2 (default) | 4; 3 = LDS-ring marching experiment
int  assemble_grid_size(int w, int h);
void launch_assemble(hipStream_t s, const LevelPtrs &L, const AssembleParams &P, int grid);
#ifdef OCTANE_DIAG      // the two-pass form of a PCG iteration: diagnostic library only (the product runs one kernel per iteration)
void launch_pcg_pass_a(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol);
void launch_pcg_pass_b(hipStream_t s, const LevelPtrs &L, int k, int nparts_a, int grid);
void launch_flow_update(hipStream_t s, const LevelPtrs &L, int niter_launched);
#endif
void set_fused_q_min(long px);               // tuning: level size from which q is recomputed (default 3 * 2^20 pixels)
void set_fused_q(int v);                     // tuning: the q-recomputing form of the fused kernel on large levels
void set_fused_rows(int r);                  // tuning: tile rows of the fused kernel (0 = by level size)
int  pcg_fused_q_form(int w, int rows, int h);            // 1: a level / band of this size recomputes q (pcg_kernels.hip)
int  pcg_fused_grid_size(int w, int rows, int unit_w, int q_form);   // fused one-kernel-per-iteration PCG
void launch_pcg_fused(hipStream_t s, const LevelPtrs &L, int k, int nparts_prev, int grid, float tol);
void launch_flow_update_fused(hipStream_t s, const LevelPtrs &L, int niter_launched, int nparts);
// ---- whole solve of a mid-size level in one launch, the level resident on chip (pcg_persist.hip) ----
constexpr int kMidMaxG = 256;        // workgroups of the persistent solve: one per CU
struct MidGeom { int gx, gy, bh, P, G; };      // sub-domain grid, rows per sub-domain, 8-row slots per thread, workgroups
struct MidArgs {
    int gx, gy, bh, G;
    int k0, k1, kcap;                // this launch runs iterations [k0, k1) of a solve capped at kcap
    int nparts_asm;                  // partial sums the assembly wrote (r.z, r.r of the right-hand side)
    int full_state;                  // stepped form: the complete state is stored to / loaded from the level's planes
    int fault;                       // diagnostic library only (the product kernel ignores it): the last workgroup leaves at once, the others' waits then have to give up
    float tol;
    unsigned tag0;                   // granule tags of this solve are tag0 + iteration + 1
    unsigned int *abort_word;        // raised when a wait timed out
    unsigned long long *parts;       // [2 parities][14 = 7 sums x low / high half][kMidMaxG] granules
    unsigned long long *edges;       // [G][2 parities][4 sides][6 arrays][128] granules
    float *wspill;                   // sub-domains of more than 10 slots: the four merged neighbour weights, [G][slot][4][512 threads]
};
int  pcg_mid_config(int w, int h, int ncu, int force_p, MidGeom *g);
void set_mid_min_p(int p);          // smallest slot count pcg_mid_config may choose (developer knob OCTANE_TUNE_PERSIST_MINP)
void set_mid_fault(int v);               // diagnostic library only, see MidArgs::fault
void pcg_mid_configure();
size_t pcg_mid_workspace_bytes();
hipError_t launch_pcg_solve_mid(hipStream_t s, const LevelPtrs &L, const MidGeom &g, void *workspace, unsigned seq, int k0, int k1, int kcap,
                                int nparts_asm, float tol);
// the stamped diagnostic build of the same source (pcg_persist_diag.hip)
void pcg_mid_configure_diag();
hipError_t launch_pcg_solve_mid_diag(hipStream_t s, const LevelPtrs &L, const MidGeom &g, void *workspace, unsigned seq, int k0, int k1, int kcap,
                                int nparts_asm, float tol);
int pcg_mid_stamps(hipStream_t s, unsigned long long *out16);

int  pcg_selftest_rcp(hipStream_t s, unsigned long long *host3);
bool pcg_small_applicable(int w, int h);
void pcg_small_configure();
void launch_pcg_solve_small(hipStream_t s, const LevelPtrs &L, int maxit, float tol);   // whole solve + flow update, one workgroup

struct NavArgs {
    double pph, req, rpol, lam0;
    float xScale, xOffset, yScale, yOffset;
    float lat1, lon1, lon0, R;
    int minX, minY, nx, ny;
};
void launch_pix2uv(hipStream_t s, const NavArgs &nav, double t1, double t2, const float *u, const float *v,
                   int mode, short *ur, short *vr, short *ur2, short *vr2, long n);
// the same source compiled with fused multiply-adds, as nvcc's default -fmad=true builds the reference (pix2uv_kernel.hip)
void launch_pix2uv_fsites(hipStream_t s, const NavArgs &nav, double t1, double t2, const float *u, const float *v,
                          int mode, short *ur, short *vr, short *ur2, short *vr2, long n);   // strict build, the two float sites fused
void launch_pix2uv_fmad(hipStream_t s, const NavArgs &nav, double t1, double t2, const float *u, const float *v,
                        int mode, short *ur, short *vr, short *ur2, short *vr2, long n);

struct NavcalArgs {
    float xScale, xOffset, yScale, yOffset, radScale, radOffset, rpol, req, H, lam0, fk1, fk2, bc1, bc2, kap1;
    float maxin, minin, maxout, minout, subpoint_slope, subpoint_int;
    int cal, donav, nx, ny, minx, maxx, miny, maxy;
};
void launch_navcal(hipStream_t s, const NavcalArgs &A, const short *x, const short *y, const short *data2,
                   float *data3, float *lat, float *lon, short *data2s);

struct ProjNavcalArgs {          // polar (mode 1) / mercator (mode 2) navigation; lon0, lat1 in radians
    float xScale, xOffset, yScale, yOffset, R, lon0, lat1;
    int donav, mode, nx, ny, minx, maxx, miny, maxy;
};
void launch_proj_navcal(hipStream_t s, const ProjNavcalArgs &A, const short *x, const short *y, const float *data2,
                        float *data3, float *lat, float *lon);

struct Uv2pixArgs {
    double secs, req, req2, rpol, rpol2, eval, lam0, pph;
    float xscale, xoffset, yscale, yoffset;
    int nx, ny;
};
void launch_uv2pix(hipStream_t s, const Uv2pixArgs &A, const float *u, const float *v, const float *lat, const float *lon,
                   const short *gx, const short *gy, float *upix, float *vpix);

// patch matching (-sosm): `spiral` holds 2*count ints (n, m in visiting order) followed by (2 srad + 1)^2 visiting
// indices laid out [n + srad][m + srad] (-1 = never visited)
void launch_sosm(hipStream_t s, const float *g1, const float *g2, float *u, float *v, int nx, int ny, int rad, int srad,
                 const int *spiral, int count);

struct SrsalArgs { double gk[37]; double sigpix2; };
void launch_srsal(hipStream_t s, const float *u, const float *v, const float *cth, int nx, int ny, const SrsalArgs &A,
                  float *uo, float *vo);

}  // namespace octane
