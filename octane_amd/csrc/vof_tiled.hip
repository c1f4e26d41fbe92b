// vof_tiled.hip -- one frame solved by several GPUs: row bands of the fine pyramid levels, exact parity.
//
// BASELINE.json configs[3] / SURVEY.md 8e "spatial tiles of one frame": the reference has no counterpart (it is
// single-GPU, ref src/oct_variational_optical_flow.cu:1251-1265); what has to be preserved is its arithmetic --
// one global PCG per linearisation with global dot products (ref .cu:1105-1195), not independent tiles blended at
// seams.  The scheme:
//
//   * every band (one per device; several bands may share a device, which is how a one-GPU box tests this file)
//     holds full-size planes and addresses them with frame coordinates;
//   * the cheap, wide-reach phases of a level (pyramid blur + decimation, gradients, bicubic flow up-sampling: ~6 %
//     of a pyramid) are REPLICATED -- every band computes them for the whole level, nothing is exchanged for them,
//     and the warp of the assembly may land anywhere in the frame;
//   * levels below `min_band_pixels` are replicated entirely (their passes are latency-bound; exchanging halos for
//     them would cost more than it saves) -- every band then holds the same bits;
//   * on the banded levels a band owns rows [y0, y1) (multiples of kBandAlign).  Its assembly also fills rows y0-1
//     and y1, so the coefficients, the preconditioner and the initial residual of the halo rows are bit-identical
//     copies of the owner's without an exchange; pass A keeps its own copy of p on those rows current
//     (pcg_kernels.hip).  What crosses bands per PCG iteration is READ IN PLACE by the consuming kernel through
//     peer-mapped pointers, not copied: the per-workgroup partials of the two reductions ({p.q} written by pass A,
//     {r.z, r.r} by pass B: every band folds all bands' partials in the same order and so takes the same alpha, beta
//     and stop decision) and ONE row of r per inner edge (pass A reads it from the neighbour's plane).  Copied, per
//     linearisation: two rows of u, v per inner edge; per level: the bands of the flow, all-gathered for the next
//     level's up-sampling.
//
// Ordering: every cross-band read is made by a kernel launched after an event wait on the producing band's stream,
// i.e. after the producing kernel completed -- the visibility point HIP defines for device memory shared between
// peers; nothing relies on stores becoming visible while a kernel runs.  No host synchronisation with the GPU inside
// a pyramid, no device atomics, no flags polled by kernels.  One host thread per band issues that band's work; the
// threads only meet each other (a spin barrier per phase) so that events are recorded before the other bands'
// streams are told to wait for them.  Copies are stream-ordered peer copies (hipMemcpyPeerAsync over xGMI between
// devices, plain device copies between bands that share a device).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/octane_vof.h"
#include "vof_kernels.hpp"
#include "vof_plan.hpp"

using namespace octane;

#define TILED_TRY(expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                 \
            return OCTANE_E_HIP;                                                               \
        }                                                                                      \
    } while (0)

struct BandRows { int y0, y1; };

struct octane_vof_tiled {
    int nx = 0, ny = 0, nc = 0, nbands = 0;
    octane_vof_params prm;
    long min_band_pixels = 0;
    std::vector<int> dev;                       // device of band b
    std::vector<octane_vof_plan *> pl;          // one full-size plan per band
    std::vector<double *> parts;                // per band: its partial block [rz|rr|pq] x kMaxParts, read by every band
    std::vector<hipEvent_t> ev;                 // per band
    std::vector<std::vector<BandRows>> rows;    // [level][band]; empty vector = level is replicated
    int last_cur = 0;
    int loaded = 0;
    long long copies = 0;                       // peer copies issued by the last solve (diagnostic)
};

static int round_to(int v, int m) { return (v + m / 2) / m * m; }

// How `rows` rows of a level are cut into `nbands` bands: edges[0] = 0 <= edges[1] <= ... <= edges[nbands] = rows, inner
// edges at multiples of 32 rows (whole pass A tiles) as close to equal shares as that allows.  Returns 1 and fills
// edges[0..nbands], or 0 when the level cannot be banded (a single band, or a band would get fewer than 32 rows) --
// such levels are solved redundantly by every band.  Pure host arithmetic: needs no GPU.
extern "C" int octane_vof_band_partition(int rows, int nbands, int *edges)
{
    if (!edges || rows < 1 || nbands < 1 || nbands > kMaxBands) return OCTANE_E_INVALID;
    if (nbands < 2 || rows < nbands * kBandAlign) return 0;
    edges[0] = 0;
    edges[nbands] = rows;
    for (int b = 1; b < nbands; b++) edges[b] = round_to((int)((long)rows * b / nbands), kBandAlign);
    for (int b = 0; b < nbands; b++)
        if (edges[b + 1] - edges[b] < kBandAlign) return 0;
    return 1;
}

extern "C" int octane_vof_tiled_destroy(octane_vof_tiled *t)
{
    if (!t) return OCTANE_OK;
    for (int b = 0; b < (int)t->pl.size(); b++) {
        (void)hipSetDevice(t->dev[b]);
        if (t->pl[b] && t->pl[b]->own_stream) (void)hipStreamSynchronize(t->pl[b]->own_stream);
        if (b < (int)t->ev.size() && t->ev[b]) (void)hipEventDestroy(t->ev[b]);
        if (b < (int)t->parts.size() && t->parts[b]) (void)hipFree(t->parts[b]);
        if (t->pl[b]) octane_vof_plan_destroy(t->pl[b]);
    }
    delete t;
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_create(octane_vof_tiled **out, int nx, int ny, int nchan, const octane_vof_params *p,
                                       int nbands, const int *devices, long long min_band_pixels)
{
    if (!out || !p || nbands < 1 || nbands > kMaxBands) {
        set_last_error("octane_vof_tiled_create: invalid argument (1 <= nbands <= 8)");
        return OCTANE_E_INVALID;
    }
    *out = nullptr;
    const int ndev = octane_device_count();
    if (ndev == 0) { set_last_error("No gpus available for use"); return OCTANE_E_NODEVICE; }
    octane_vof_tiled *t = new octane_vof_tiled();
    t->nx = nx; t->ny = ny; t->nc = nchan; t->nbands = nbands; t->prm = *p;
    // Below ~12 Mpixel one PCG iteration of the whole level (< 0.25 ms) is cheaper than issuing a banded one.
    t->min_band_pixels = min_band_pixels > 0 ? (long)min_band_pixels : (12L << 20);
    if (const char *e = getenv("OCTANE_TUNE_MIN_BAND_PIXELS")) t->min_band_pixels = atol(e);
    for (int b = 0; b < nbands; b++) {
        int d = devices ? devices[b] : b % ndev;
        if (d < 0 || d > ndev - 1) d = 0;                     // as the reference treats a bad -set_device (.cu:1260)
        t->dev.push_back(d);
    }
    // peer access between every pair of distinct devices (an error here only means "already enabled")
    for (int a = 0; a < nbands; a++)
        for (int b = 0; b < nbands; b++)
            if (t->dev[a] != t->dev[b]) {
                int can = 0;
                if (hipSetDevice(t->dev[a]) == hipSuccess && hipDeviceCanAccessPeer(&can, t->dev[a], t->dev[b]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(t->dev[b], 0);
                (void)hipGetLastError();
                if (!can) {      // the PCG kernels read the other bands' partials and edge rows in place
                    set_last_error("octane_vof_tiled_create: devices " + std::to_string(t->dev[a]) + " and " +
                                   std::to_string(t->dev[b]) + " cannot access each other's memory");
                    delete t;
                    return OCTANE_E_INVALID;
                }
            }
    t->pl.assign(nbands, nullptr);
    t->parts.assign(nbands, nullptr);
    t->ev.assign(nbands, nullptr);
    int rc = OCTANE_OK;
    for (int b = 0; b < nbands && rc == OCTANE_OK; b++) {
        octane_vof_params pb = *p;
        pb.device = t->dev[b];
        // placement trials only when the band has its device to itself (they allocate four arenas)
        int sharing = 0;
        for (int c = 0; c < nbands; c++) sharing += (t->dev[c] == t->dev[b]);
        rc = plan_create_ex(&t->pl[b], nx, ny, nchan, &pb, sharing > 1 ? 1 : 4);
        if (rc != OCTANE_OK) break;
        if (hipSetDevice(t->dev[b]) != hipSuccess ||
            hipMalloc((void **)&t->parts[b], (size_t)3 * kMaxParts * sizeof(double)) != hipSuccess ||
            hipMemset(t->parts[b], 0, (size_t)3 * kMaxParts * sizeof(double)) != hipSuccess ||
            hipEventCreateWithFlags(&t->ev[b], hipEventDisableTiming) != hipSuccess) {
            set_last_error("octane_vof_tiled_create: device allocation failed");
            rc = OCTANE_E_NOMEM;
        }
    }
    for (int b = 0; b < nbands && rc == OCTANE_OK; b++)      // the fills above ran on the null stream
        if (hipSetDevice(t->dev[b]) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = OCTANE_E_HIP;
    if (rc != OCTANE_OK) { octane_vof_tiled_destroy(t); return rc; }

    // Row bands per level (octane_vof_band_partition); levels it declines stay replicated.
    const std::vector<LevelInfo> &lev = t->pl[0]->lev;
    t->rows.resize(lev.size());
    for (size_t k = 0; k < lev.size(); k++) {
        const LevelInfo &li = lev[k];
        if ((long)li.w * li.h < t->min_band_pixels) continue;
        int edges[kMaxBands + 1];
        if (octane_vof_band_partition(li.h, nbands, edges) != 1) continue;
        std::vector<BandRows> r(nbands);
        for (int b = 0; b < nbands; b++) { r[b].y0 = edges[b]; r[b].y1 = edges[b + 1]; }
        t->rows[k] = r;
    }
    *out = t;
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_banded_levels(const octane_vof_tiled *t)
{
    if (!t) return -1;
    int n = 0;
    for (auto &r : t->rows) n += !r.empty();
    return n;
}

extern "C" int octane_vof_tiled_band_rows(const octane_vof_tiled *t, int level, int band, int *y0, int *y1)
{
    if (!t || level < 0 || level >= (int)t->rows.size() || band < 0 || band >= t->nbands || !y0 || !y1) return OCTANE_E_INVALID;
    if (t->rows[level].empty()) { *y0 = 0; *y1 = t->pl[0]->lev[level].h; return 0; }
    *y0 = t->rows[level][band].y0; *y1 = t->rows[level][band].y1;
    return 1;
}

extern "C" long long octane_vof_tiled_last_iterations(octane_vof_tiled *t)
{
    if (!t) return -1;
    return *t->pl[0]->h_iters;
}

extern "C" long long octane_vof_tiled_last_copies(octane_vof_tiled *t) { return t ? t->copies : -1; }

extern "C" size_t octane_vof_tiled_device_bytes(const octane_vof_tiled *t)
{
    return t ? t->pl[0]->arena_bytes + (size_t)3 * kMaxParts * sizeof(double) : 0;
}

// ---- transport -----------------------------------------------------------------------------------------------
// One host thread per band issues that band's launches, copies and event operations.  The threads never wait for
// the GPU; they meet at a spin barrier once per phase only so that "record my event" is issued before "make your
// stream wait for my event" (waiting on an event that has not been recorded yet is a no-op in HIP).
struct SpinBarrier {
    std::atomic<int> arrived{0};
    std::atomic<int> generation{0};
    int n = 1;
    void wait()
    {
        const int gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            arrived.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == gen)
                if (++spins > 4096) std::this_thread::yield();
        }
    }
};

struct BandRun {                     // per-solve state shared by the band threads
    octane_vof_tiled *t;
    SpinBarrier bar;
    std::atomic<int> failed{0};      // first failing band + 1
    std::mutex mu;
    std::string error;
    int rc = OCTANE_OK;
    std::vector<long long> copies;
    std::vector<int> cur;
};

// HIP calls of a band thread: after the first failure anywhere every thread keeps walking the same control flow
// (so the barriers still match up) but issues nothing more.
#define BAND_HIP(expr)                                                                         \
    do {                                                                                       \
        if (!R.failed.load(std::memory_order_relaxed)) {                                       \
            hipError_t e_ = (expr);                                                            \
            if (e_ != hipSuccess) band_fail(R, b, OCTANE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)

static void band_fail(BandRun &R, int b, int rc, const std::string &msg)
{
    std::lock_guard<std::mutex> g(R.mu);
    if (!R.failed.load()) { R.rc = rc; R.error = msg; R.failed.store(b + 1); }
}

static void copy_band(BandRun &R, int b, void *dst, int dband, const void *src, size_t bytes)
{
    octane_vof_tiled *t = R.t;
    R.copies[b]++;
    hipStream_t s = t->pl[b]->own_stream;
    if (t->dev[dband] == t->dev[b]) BAND_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    else BAND_HIP(hipMemcpyPeerAsync(dst, t->dev[dband], src, t->dev[b], bytes, s));
}

// Everything band b has issued so far becomes a dependency of whatever the other bands issue next, and vice versa.
static void sync_bands(BandRun &R, int b)
{
    octane_vof_tiled *t = R.t;
    BAND_HIP(hipEventRecord(t->ev[b], t->pl[b]->own_stream));
    R.bar.wait();
    for (int c = 0; c < t->nbands; c++)
        if (c != b) BAND_HIP(hipStreamWaitEvent(t->pl[b]->own_stream, t->ev[c], 0));
    R.bar.wait();        // nobody re-records its event before everybody has queued the waits on it
}

// Rows [ya, yb) of a plane (same layout in every band's arena) from band b to band dst.
static void send_rows(BandRun &R, int b, float *from_base, float *to_base, int pitch, int ya, int yb, int dst)
{
    copy_band(R, b, to_base + (size_t)ya * pitch, dst, from_base + (size_t)ya * pitch, (size_t)(yb - ya) * pitch * sizeof(float));
}

// ---- one banded level, as band b's thread issues it ---------------------------------------------------------------
static void solve_level_banded(BandRun &R, int b, int k, int cur, const LevelCtx &ctx, bool finest)
{
    octane_vof_tiled *t = R.t;
    const int nb = t->nbands;
    const octane_vof_params &prm = t->prm;
    octane_vof_plan *pl = t->pl[b];
    hipStream_t s = pl->own_stream;
    const LevelInfo &li = pl->lev[k];
    const std::vector<BandRows> &rows = t->rows[k];
    LevelPtrs L;
    plan_fill_level_ptrs(pl, k, cur, ctx, L);
    L.y0 = rows[b].y0; L.y1 = rows[b].y1;
    L.ya0 = (b == 0) ? 0 : L.y0 - 1;
    L.ya1 = (b == nb - 1) ? li.h : L.y1 + 1;
    L.nbands = nb;
    double *own = t->parts[b];
    L.part_rz = own + kPartRz; L.part_rr = own + kPartRr; L.part_pq = own + kPartPq;
    for (int c = 0; c < kMaxBands; c++) L.band_parts[c] = t->parts[c < nb ? c : b];
    L.ru_up = t->pl[b > 0 ? b - 1 : b]->ru; L.rv_up = t->pl[b > 0 ? b - 1 : b]->rv;
    L.ru_dn = t->pl[b < nb - 1 ? b + 1 : b]->ru; L.rv_dn = t->pl[b < nb - 1 ? b + 1 : b]->rv;
    int maxrows = 0;
    for (int c = 0; c < nb; c++) maxrows = rows[c].y1 - rows[c].y0 > maxrows ? rows[c].y1 - rows[c].y0 : maxrows;
    // every band launches the same grids, so that the slots hold the same number of partials (idle workgroups
    // contribute zeros)
    const int g_asm = assemble_grid_size(li.w, maxrows + 2);
    const int g_a = pcg_band_grid_size(li.w, maxrows);
    const int g_b = pcg_b_grid_size(li.w, maxrows);

    for (int gnc = 0; gnc < 3; gnc++) {                 // ref .cu:604-606
        AssembleParams ap;
        ap.al1 = 1. - 0.5 * gnc;
        ap.alpha = prm.alpha;
        ap.loa = prm.lambda / prm.alpha;                // ref .cu:1230
        ap.lambdac = li.lambdac;
        ap.dozim = prm.dozim != 0;
        L.unit_w = (pl->use_unit_w && gnc == 0) ? 1 : 0;
        for (int l = 0; l < prm.liters; l++) {          // ref .cu:608
            if (!R.failed.load()) launch_assemble(s, L, ap, g_asm);
            sync_bands(R, b);
            for (int it = 0; it < prm.cgiters; it++) {  // ref .cu:1131-1182
                if (!R.failed.load()) launch_pcg_pass_a(s, L, it, it == 0 ? g_asm : g_b, g_a, pl->tol);
                sync_bands(R, b);
                if (!R.failed.load()) launch_pcg_pass_b(s, L, it, g_a, g_b);
                sync_bands(R, b);
            }
            if (!R.failed.load()) launch_flow_update(s, L, prm.cgiters);          // ref .cu:1185-1195
            // the next assembly reads u, v two rows beyond the band (one for the halo row it fills, one for that
            // row's own 3 x 3 neighbourhood)
            if (b > 0) {
                send_rows(R, b, pl->U[cur], t->pl[b - 1]->U[cur], li.pitch, L.y0, L.y0 + 2, b - 1);
                send_rows(R, b, pl->V[cur], t->pl[b - 1]->V[cur], li.pitch, L.y0, L.y0 + 2, b - 1);
            }
            if (b < nb - 1) {
                send_rows(R, b, pl->U[cur], t->pl[b + 1]->U[cur], li.pitch, L.y1 - 2, L.y1, b + 1);
                send_rows(R, b, pl->V[cur], t->pl[b + 1]->V[cur], li.pitch, L.y1 - 2, L.y1, b + 1);
            }
            sync_bands(R, b);
        }
    }
    // Level done: the next level's up-sampling (replicated) needs the whole flow on every band; after the finest
    // level only band 0, which hands the result out, does.
    for (int c = 0; c < nb; c++) {
        if (c == b || (finest && c != 0)) continue;
        send_rows(R, b, pl->U[cur], t->pl[c]->U[cur], li.pitch, L.y0, L.y1, c);
        send_rows(R, b, pl->V[cur], t->pl[c]->V[cur], li.pitch, L.y0, L.y1, c);
    }
    sync_bands(R, b);
}

// The whole pyramid of band b.
static void band_worker(BandRun &R, int b)
{
    octane_vof_tiled *t = R.t;
    octane_vof_plan *pl = t->pl[b];
    const int nlev = (int)pl->lev.size();
    BAND_HIP(hipSetDevice(t->dev[b]));
    BAND_HIP(hipMemsetAsync(pl->d_iters, 0, sizeof(long long), pl->own_stream));
    pl->evs_used = 0;
    int cur = 0;
    LevelCtx ctx;
    for (int k = 0; k < nlev; k++) {
        if (!R.failed.load()) {
            const int rc = plan_level_setup(pl, pl->own_stream, k, cur, ctx);
            if (rc) band_fail(R, b, rc, "plan_level_setup failed");
        } else if (k > 0) {
            cur ^= 1;      // keep the bookkeeping in step with the healthy bands
        }
        if (t->rows[k].empty()) {
            if (!R.failed.load()) {
                const int rc = plan_level_solve(pl, pl->own_stream, k, cur, ctx, false);
                if (rc) band_fail(R, b, rc, "plan_level_solve failed");
            }
        } else {
            sync_bands(R, b);          // a band's halo rows must not be written while it still sets the level up
            solve_level_banded(R, b, k, cur, ctx, k == nlev - 1);
        }
    }
    if (b == 0) BAND_HIP(hipMemcpyAsync(pl->h_iters, pl->d_iters, sizeof(long long), hipMemcpyDeviceToHost, pl->own_stream));
    BAND_HIP(hipGetLastError());
    R.cur[b] = cur;
}

static hipError_t copy_from_band0(octane_vof_tiled *t, void *dst, int dband, const void *src, size_t bytes, hipStream_t s)
{
    if (t->dev[dband] == t->dev[0]) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    return hipMemcpyPeerAsync(dst, t->dev[dband], src, t->dev[0], bytes, s);
}

extern "C" int octane_vof_tiled_load(octane_vof_tiled *t, const float *img1, const float *img2, const float *u, const float *v, int mem)
{
    if (!t || !img1 || !img2 || !u || !v || (mem != OCTANE_MEM_HOST && mem != OCTANE_MEM_DEVICE)) {
        set_last_error("octane_vof_tiled_load: invalid argument");
        return OCTANE_E_INVALID;
    }
    // Host buffers: every band uploads the frame itself.  Device buffers (dense, on band 0's device): band 0
    // repacks them into its pitched planes and the other bands copy those planes.
    for (int b = 0; b < t->nbands; b++) {
        TILED_TRY(hipSetDevice(t->dev[b]));
        octane_vof_plan *pl = t->pl[b];
        if (mem == OCTANE_MEM_HOST || b == 0) {
            const int rc = plan_load_inputs(pl, img1, img2, u, v, mem, pl->own_stream);
            if (rc) return rc;
            if (b == 0 && mem == OCTANE_MEM_DEVICE) TILED_TRY(hipStreamSynchronize(pl->own_stream));
        } else {
            octane_vof_plan *p0 = t->pl[0];
            const size_t img_bytes = ((size_t)(t->nc - 1) * pl->plane0 + (size_t)pl->pitch0 * t->ny) * sizeof(float);
            const size_t one = (size_t)pl->pitch0 * t->ny * sizeof(float);
            TILED_TRY(copy_from_band0(t, pl->img1p, b, p0->img1p, img_bytes, pl->own_stream));
            TILED_TRY(copy_from_band0(t, pl->img2p, b, p0->img2p, img_bytes, pl->own_stream));
            TILED_TRY(copy_from_band0(t, pl->uh, b, p0->uh, one, pl->own_stream));
            TILED_TRY(copy_from_band0(t, pl->vh, b, p0->vh, one, pl->own_stream));
        }
    }
    for (int b = 0; b < t->nbands; b++) {
        TILED_TRY(hipSetDevice(t->dev[b]));
        TILED_TRY(hipStreamSynchronize(t->pl[b]->own_stream));
    }
    t->loaded = 1;
    return OCTANE_OK;
}

// The pyramid on the loaded inputs.  Asynchronous: returns once everything is issued; octane_vof_tiled_wait or
// octane_vof_tiled_fetch synchronise.
extern "C" int octane_vof_tiled_solve(octane_vof_tiled *t)
{
    if (!t || !t->loaded) { set_last_error("octane_vof_tiled_solve: no inputs loaded"); return OCTANE_E_INVALID; }
    const int nb = t->nbands;
    BandRun R;
    R.t = t;
    R.bar.n = nb;
    R.copies.assign(nb, 0);
    R.cur.assign(nb, 0);
    if (nb == 1) {
        band_worker(R, 0);
    } else {
        std::vector<std::thread> th;
        for (int b = 0; b < nb; b++) th.emplace_back([&R, b]() { band_worker(R, b); });
        for (auto &x : th) x.join();
    }
    t->copies = 0;
    for (int b = 0; b < nb; b++) t->copies += R.copies[b];
    if (R.failed.load()) {
        set_last_error("octane_vof_tiled_solve (band " + std::to_string(R.failed.load() - 1) + "): " + R.error);
        (void)octane_vof_tiled_wait(t);
        return R.rc;
    }
    (void)hipSetDevice(t->dev[0]);
    t->last_cur = R.cur[0];
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_wait(octane_vof_tiled *t)
{
    if (!t) return OCTANE_E_INVALID;
    for (int b = 0; b < t->nbands; b++) {
        TILED_TRY(hipSetDevice(t->dev[b]));
        TILED_TRY(hipStreamSynchronize(t->pl[b]->own_stream));
    }
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_fetch(octane_vof_tiled *t, float *u, float *v, int mem)
{
    if (!t || !u || !v || (mem != OCTANE_MEM_HOST && mem != OCTANE_MEM_DEVICE)) return OCTANE_E_INVALID;
    int rc = octane_vof_tiled_wait(t);
    if (rc) return rc;
    octane_vof_plan *p0 = t->pl[0];
    TILED_TRY(hipSetDevice(t->dev[0]));
    const size_t dense_row = (size_t)t->nx * sizeof(float), pitched_row = (size_t)p0->pitch0 * sizeof(float);
    const int cur = t->last_cur;
    if (mem == OCTANE_MEM_HOST) {
        TILED_TRY(hipMemcpy2DAsync(u, dense_row, p0->U[cur], pitched_row, dense_row, t->ny, hipMemcpyDeviceToHost, p0->own_stream));
        TILED_TRY(hipMemcpy2DAsync(v, dense_row, p0->V[cur], pitched_row, dense_row, t->ny, hipMemcpyDeviceToHost, p0->own_stream));
    } else {
        launch_copy2d(p0->own_stream, p0->U[cur], p0->pitch0, u, t->nx, t->nx, t->ny);
        launch_copy2d(p0->own_stream, p0->V[cur], p0->pitch0, v, t->nx, t->nx, t->ny);
    }
    TILED_TRY(hipStreamSynchronize(p0->own_stream));
    TILED_TRY(hipGetLastError());
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_run(octane_vof_tiled *t, const float *img1, const float *img2, float *u, float *v, int mem)
{
    int rc = octane_vof_tiled_load(t, img1, img2, u, v, mem);
    if (rc == OCTANE_OK) rc = octane_vof_tiled_solve(t);
    if (rc == OCTANE_OK) rc = octane_vof_tiled_fetch(t, u, v, mem);
    return rc;
}
