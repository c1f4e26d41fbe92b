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
//     copies of the owner's without an exchange.  What crosses bands per PCG iteration is READ IN PLACE by the
//     consuming kernel through peer-mapped pointers, not copied: the per-workgroup partial sums (every band folds all
//     bands' partials in the same order and so takes the same alpha, beta and stop decision) and a few rows per inner
//     edge -- fused kernel, bands of >= 3 Mpixel (q recomputed): r on the row beyond the edge, p on the two rows
//     beyond it, wy of the row above the upper one; fused kernel, smaller bands (q stored; halo rows of r and p kept
//     current locally): one row of q; two-pass form: one row of r, pass A keeps its own copy of p on the halo rows
//     (pcg_kernels.hip).  Copied, per linearisation: two rows of u, v per inner edge; per level: the bands of the flow,
//     all-gathered for the next level's up-sampling.
//
// Transport (OCTANE_TILED_TRANSPORT): "inplace" (default) is the scheme above -- the consuming kernel dereferences the
// neighbour's memory.  "copy" is the fallback SURVEY 5 asks for, there to tell a protocol error from a visibility problem of
// in-kernel peer reads should a multi-GPU run ever disagree with the plain plan: the same rows and partial blocks are PULLED
// into this band's own planes / a local mirror by stream-ordered runtime copies (hipMemcpyPeerAsync, or hipMemcpyAsync on an
// IPC mapping) after the phase boundary, and every kernel reads local memory only.  Same arithmetic, same fold order, same
// bits.  (RCCL itself is not used for this: ncclAllGather + ncclSend / Recv would carry exactly these copies, but RCCL
// refuses two ranks on one device, so on the one-GPU boxes this code is tested on it could never run.)
//
// Ordering: every cross-band read is made by a kernel launched after an event wait on the producing band's stream,
// i.e. after the producing kernel completed -- the visibility point HIP defines for device memory shared between
// peers; nothing relies on stores becoming visible while a kernel runs.  No host synchronisation with the GPU inside
// a pyramid, no device atomics, no flags polled by kernels.  One host thread per band issues that band's work; the
// threads only meet each other (a spin barrier per phase) so that events are recorded before the other bands'
// streams are told to wait for them.  Copies are stream-ordered peer copies (hipMemcpyPeerAsync over xGMI between
// devices, plain device copies between bands that share a device).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/octane_vof_dev.h"
#include "../../include/octane_extras.h"
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
constexpr long kDefaultMinBandPixels = 4L << 20;     // levels below this many pixels are solved redundantly by every band (octane_vof_tiled_create)

extern "C" const char *octane_vof_transport_name(int transport)
{
    return transport == OCTANE_TRANSPORT_INPLACE ? "inplace" : transport == OCTANE_TRANSPORT_COPY ? "copy" :
           transport == OCTANE_TRANSPORT_COLLECTIVE ? "collective" : "unknown";
}
// OCTANE_TILED_TRANSPORT=inplace|copy|collective forces a transport (-1: not set, -2: set to something else)
static int forced_transport()
{
    const char *e = getenv("OCTANE_TILED_TRANSPORT");
    if (!e || !*e || std::string(e) == "auto") return -1;
    for (int t = 0; t < 3; t++) if (std::string(e) == octane_vof_transport_name(t)) return t;
    return -2;
}
static bool selfcheck_enabled()
{
    const char *e = getenv("OCTANE_TILED_SELFCHECK");
    return !(e && std::string(e) == "0");
}
static int test_break_bits()
{
#ifdef OCTANE_DIAG
    if (const char *e = getenv("OCTANE_TEST_BREAK_TRANSPORT")) return atoi(e);
#endif
    return 0;
}
static void info_reset(octane_vof_transport_info &in, int nbands)
{
    std::memset(&in, 0, sizeof in);
    in.q_dma = 1; in.peer_ok = 1; in.nbands = nbands; in.ndevices = 1;
    for (int i = 0; i < 4; i++) in.check_rel_l2[i] = -1.;
}

struct octane_vof_tiled {
    int nx = 0, ny = 0, nc = 0, nbands = 0;
    octane_vof_params prm;
    long min_band_pixels = 0;
    std::vector<int> dev;                       // device of band b
    std::vector<octane_vof_plan *> pl;          // one full-size plan per band
    std::vector<double *> parts;                // per band: its partial block [rz|rr|pq] x kMaxParts, read by every band
    std::vector<double *> mirror;               // per band: local copies of every band's partial blocks (copy transport)
    std::vector<hipEvent_t> ev;                 // per band
    std::vector<std::vector<BandRows>> rows;    // [level][band]; empty vector = level is replicated
    int last_cur = 0;
    int loaded = 0;
    long long copies = 0;                       // peer copies issued by the last solve (diagnostic)
    bool aborted = false;                       // a persistent mid-level solve of the last solve gave up: its flow is not valid
    int transport = OCTANE_TRANSPORT_INPLACE;   // what the self-check (or OCTANE_TILED_TRANSPORT) chose
    bool no_dma = false;                        // q-form band launches by the register-staged kernel
    octane_vof_transport_info info;
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
        if (b < (int)t->mirror.size() && t->mirror[b]) (void)hipFree(t->mirror[b]);
        if (t->pl[b]) octane_vof_plan_destroy(t->pl[b]);
    }
    delete t;
    return OCTANE_OK;
}

// ---- first-contact self-check ----------------------------------------------------------------------------------------
// A synthetic pair every rank can build for itself, bit for bit (the ranks share the node's libm): separable cosines, image 2 = image 1
// moved by (1.25, -0.75) pixels.
static void check_scene(int w, int h, std::vector<float> &a, std::vector<float> &b)
{
    const double lx[4] = {23., 37., 61., 97.}, ly[4] = {29., 41., 53., 89.}, dx = 1.25, dy = -0.75;
    std::vector<double> cx[4], cy[4], sx[4], sy[4];
    for (int k = 0; k < 4; k++) {
        cx[k].resize(w); sx[k].resize(w); cy[k].resize(h); sy[k].resize(h);
        for (int i = 0; i < w; i++) { cx[k][i] = std::cos(6.283185307179586 * i / lx[k] + 0.7 * k); sx[k][i] = std::cos(6.283185307179586 * (i - dx) / lx[k] + 0.7 * k); }
        for (int j = 0; j < h; j++) { cy[k][j] = std::cos(6.283185307179586 * j / ly[k] + 1.3 * k); sy[k][j] = std::cos(6.283185307179586 * (j - dy) / ly[k] + 1.3 * k); }
    }
    a.resize((size_t)w * h); b.resize((size_t)w * h);
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            double va = 127.5, vb = 127.5;
            for (int k = 0; k < 4; k++) { va += 25. * cx[k][i] * cy[k][j]; vb += 25. * sx[k][i] * sy[k][j]; }
            a[(size_t)j * w + i] = (float)va; b[(size_t)j * w + i] = (float)vb;
        }
}
static double flow_rel_l2(const std::vector<float> &u, const std::vector<float> &v, const std::vector<float> &uo, const std::vector<float> &vo, bool *finite)
{
    double num = 0., den = 0.;
    bool fin = true;
    for (size_t i = 0; i < u.size(); i++) {
        const double du = (double)u[i] - uo[i], dv = (double)v[i] - vo[i];
        num += du * du + dv * dv; den += (double)uo[i] * uo[i] + (double)vo[i] * vo[i];
        fin = fin && std::isfinite(u[i]) && std::isfinite(v[i]);
    }
    if (finite) *finite = fin;
    return den > 0. ? std::sqrt(num / den) : (num > 0. ? 1. : 0.);
}
// The check frame: two pyramid levels, both banded.  `large`: the finer level's bands hold 2 Mpixel -- the size from which band
// launches are the q-recomputing kernel that fetches the neighbour's rows by LDS-DMA (pcg_fused_q_form) -- the coarser level's
// bands run the stored-q kernel; small: both levels on the stored-q kernel.
static void check_geometry(int nbands, bool large, int *w, int *h) { *w = large ? 2048 : 512; *h = (large ? 1024 : 128) * nbands; }
static octane_vof_params check_params(const octane_vof_params &p, int device)
{
    octane_vof_params c = p;
    c.kiters = 2; c.liters = 1; c.cgiters = 6; c.device = device;
    return c;
}
constexpr int kCheckBandedLevels = 2;    // both levels of the check frame (kiters = 2, threshold 1 pixel) are banded
constexpr double kCheckBar = 2e-5;      // banded against plain: two groupings of the same fp64 partial sums (bit-identical in every run so far)
struct Candidate { int transport; bool no_dma; };

static int plain_check_flow(const octane_vof_params &cp, int w, int h, const std::vector<float> &a, const std::vector<float> &b,
                            std::vector<float> &u, std::vector<float> &v, long long *iters)
{
    octane_vof_plan *pl = nullptr;
    int rc = plan_create_ex(&pl, w, h, 1, &cp, 1);
    if (rc != OCTANE_OK) return rc;
    u.assign((size_t)w * h, 0.f); v.assign((size_t)w * h, 0.f);
    rc = octane_vof_plan_run(pl, a.data(), b.data(), u.data(), v.data(), OCTANE_MEM_HOST, nullptr);
    if (rc == OCTANE_OK) *iters = octane_vof_plan_last_iterations(pl);
    octane_vof_plan_destroy(pl);
    return rc;
}

static int tiled_create_impl(octane_vof_tiled **out, int nx, int ny, int nchan, const octane_vof_params *p,
                             int nbands, const int *devices, long long min_band_pixels, bool allow_selfcheck);

struct CheckResult { int transport; bool no_dma; octane_vof_transport_info info; };
static std::mutex g_check_mu;
static std::vector<std::pair<std::string, CheckResult>> g_check_cache;     // per process: (devices, size class, drill bits) -> verdict

// Thread form: the bands of `t` as they will run, on a small frame, against the plain plan; picks t->transport / t->no_dma.
static int tiled_selfcheck(octane_vof_tiled *t)
{
    const int nb = t->nbands;
    bool large = false;                  // does a banded level of the real plan run the q-recomputing kernel?
    for (size_t k = 0; k < t->rows.size(); k++) {
        if (t->rows[k].empty()) continue;
        int maxrows = 0;
        for (auto &r : t->rows[k]) maxrows = std::max(maxrows, r.y1 - r.y0);
        if (t->pl[0]->use_fused && pcg_fused_q_form(t->pl[0]->lev[k].w, maxrows, t->pl[0]->lev[k].h)) large = true;
    }
    std::string key = (large ? "L" : "S") + std::to_string(test_break_bits()) + (t->info.peer_ok ? "p" : "n");
    for (int b = 0; b < nb; b++) key += "," + std::to_string(t->dev[b]);
    {
        std::lock_guard<std::mutex> g(g_check_mu);
        for (auto &e : g_check_cache)
            if (e.first == key) {
                t->transport = e.second.transport; t->no_dma = e.second.no_dma;
                const int nd = t->info.ndevices, pk = t->info.peer_ok;
                t->info = e.second.info; t->info.ndevices = nd; t->info.peer_ok = pk; t->info.nbands = nb;
                if (t->info.selfcheck < 0) { set_last_error("octane_vof_tiled_create: the first-contact self-check failed under every transport on these devices (earlier in this process)"); return OCTANE_E_HIP; }
                return OCTANE_OK;
            }
    }
    int w, h;
    check_geometry(nb, large, &w, &h);
    std::vector<float> a, b2, up, vp, u, v;
    check_scene(w, h, a, b2);
    long long its_plain = 0;
    const octane_vof_params cp = check_params(t->prm, t->dev[0]);
    int rc = plain_check_flow(cp, w, h, a, b2, up, vp, &its_plain);
    if (rc != OCTANE_OK) return rc;
    std::vector<Candidate> cand;
    if (t->info.peer_ok) { cand.push_back({OCTANE_TRANSPORT_INPLACE, false}); if (large) cand.push_back({OCTANE_TRANSPORT_INPLACE, true}); }
    cand.push_back({OCTANE_TRANSPORT_COPY, false});
    octane_vof_tiled *c = nullptr;
    rc = tiled_create_impl(&c, w, h, 1, &cp, nb, t->dev.data(), 1, false);
    if (rc != OCTANE_OK) return rc;
    if (octane_vof_tiled_banded_levels(c) != kCheckBandedLevels) {     // a check frame without banded levels would pass every transport vacuously
        fprintf(stderr, "octane: row-band self-check NOT RUN: the check frame (%d x %d, %d bands) has %d banded level(s), expected %d\n",
                w, h, nb, octane_vof_tiled_banded_levels(c), kCheckBandedLevels);
        octane_vof_tiled_destroy(c);
        t->info.selfcheck = 0;
        return OCTANE_OK;
    }
    int chosen = -1;
    std::string log;
    for (size_t i = 0; i < cand.size() && chosen < 0; i++) {
        c->transport = cand[i].transport; c->no_dma = cand[i].no_dma;
        u.assign((size_t)w * h, 0.f); v.assign((size_t)w * h, 0.f);
        rc = octane_vof_tiled_run(c, a.data(), b2.data(), u.data(), v.data(), OCTANE_MEM_HOST);
        bool fin = false;
        const double d = rc == OCTANE_OK ? flow_rel_l2(u, v, up, vp, &fin) : 1.;
        const long long its = rc == OCTANE_OK ? octane_vof_tiled_last_iterations(c) : -1;
        if (i < 4) t->info.check_rel_l2[i] = d;
        t->info.candidates_tried = (int)i + 1;
        const bool pass = rc == OCTANE_OK && fin && d <= kCheckBar && its == its_plain;
        log += std::string(i ? "; " : "") + octane_vof_transport_name(cand[i].transport) + (cand[i].no_dma ? " without LDS-DMA from the neighbour" : "") +
               (pass ? ": ok" : rc != OCTANE_OK ? std::string(": error (") + octane_last_error() + ")" :
                       ": rel L2 " + std::to_string(d) + ", iterations " + std::to_string(its) + " / " + std::to_string(its_plain));
        if (pass) chosen = (int)i;
    }
    octane_vof_tiled_destroy(c);
    t->info.selfcheck = chosen < 0 ? -1 : chosen == 0 ? 1 : 2;
    if (chosen >= 0) { t->transport = cand[chosen].transport; t->no_dma = cand[chosen].no_dma; }
    t->info.transport = t->transport; t->info.q_dma = t->no_dma ? 0 : 1;
    if (chosen != 0)
        fprintf(stderr, "octane: row-band self-check on devices [%s] (%d x %d, %d bands): %s -> %s\n", key.c_str(), w, h, nb, log.c_str(),
                chosen < 0 ? "NO transport reproduces the plain plan" : "using the last of these");
    {
        std::lock_guard<std::mutex> g(g_check_mu);
        g_check_cache.push_back({key, {t->transport, t->no_dma, t->info}});
    }
    if (chosen < 0) { set_last_error("octane_vof_tiled_create: the first-contact self-check failed under every transport: " + log); return OCTANE_E_HIP; }
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_create(octane_vof_tiled **out, int nx, int ny, int nchan, const octane_vof_params *p,
                                       int nbands, const int *devices, long long min_band_pixels)
{
    return tiled_create_impl(out, nx, ny, nchan, p, nbands, devices, min_band_pixels, true);
}

static int tiled_create_impl(octane_vof_tiled **out, int nx, int ny, int nchan, const octane_vof_params *p,
                             int nbands, const int *devices, long long min_band_pixels, bool allow_selfcheck)
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
    info_reset(t->info, nbands);
    // Which levels are worth banding (round 4, measured with four virtual bands at 10848^2, profiles/r4_tiled_threshold.txt): a level's
    // PCG iteration as bands costs the band's kernel (10.7 us + 12 ps x its pixels) + ~15 us of phase boundary, replicated it costs the
    // whole level's kernel on EVERY device.  The 2712^2 level (7.4 Mpixel: 99 us replicated, ~48 us as four bands, ~37 us as eight)
    // pays -- banding it lowers even the total work of four bands sharing one GPU, 692 -> 646 ms per pyramid; the 1356^2 level
    // (1.8 Mpixel: a persistent solve of ~17 us per iteration) does not.  Until round 3 the threshold was 12 Mpixel.
    t->min_band_pixels = min_band_pixels > 0 ? (long)min_band_pixels : kDefaultMinBandPixels;
    // (the environment's threshold is for the caller's plan only: the self-check's small frame must keep its banded levels -- ADVICE r4)
    if (allow_selfcheck) if (const char *e = getenv("OCTANE_TUNE_MIN_BAND_PIXELS")) t->min_band_pixels = atol(e);
    for (int b = 0; b < nbands; b++) {
        int d = devices ? devices[b] : b % ndev;
        if (d < 0 || d > ndev - 1) d = 0;                     // as the reference treats a bad -set_device (.cu:1260)
        t->dev.push_back(d);
    }
    {
        std::vector<int> seen;
        for (int d : t->dev) if (std::find(seen.begin(), seen.end(), d) == seen.end()) seen.push_back(d);
        t->info.ndevices = (int)seen.size();
    }
    // Peer access between every pair of distinct devices ("already enabled" is fine).  The in-place transport needs it: its kernels
    // dereference the other bands' memory.  Without it (no peer path, or enabling failed) the bands still solve the frame -- by the
    // copy transport, whose hipMemcpyPeerAsync the runtime stages itself -- and transport_info says peer_ok = 0.
    std::string peer_note;
    for (int a = 0; a < nbands; a++)
        for (int b = 0; b < nbands; b++)
            if (t->dev[a] != t->dev[b]) {
                int can = 0;
                hipError_t pe = hipSetDevice(t->dev[a]);
                if (pe == hipSuccess) pe = hipDeviceCanAccessPeer(&can, t->dev[a], t->dev[b]);
                if (pe == hipSuccess && can) {
                    pe = hipDeviceEnablePeerAccess(t->dev[b], 0);
                    if (pe == hipErrorPeerAccessAlreadyEnabled) pe = hipSuccess;
                }
                (void)hipGetLastError();
                if ((pe != hipSuccess || !can) && t->info.peer_ok) {
                    t->info.peer_ok = 0;
                    peer_note = "device " + std::to_string(t->dev[a]) + " cannot access the memory of device " + std::to_string(t->dev[b]) +
                                (pe != hipSuccess ? std::string(": ") + hipGetErrorString(pe) : std::string(" (no peer path)"));
                }
            }
    t->pl.assign(nbands, nullptr);
    t->parts.assign(nbands, nullptr);
    t->mirror.assign(nbands, nullptr);
    t->ev.assign(nbands, nullptr);
    int rc = OCTANE_OK;
    for (int b = 0; b < nbands && rc == OCTANE_OK; b++) {
        octane_vof_params pb = *p;
        pb.device = t->dev[b];
        // placement trials only when the band has its device to itself (they allocate four arenas)
        int sharing = 0;
        for (int c = 0; c < nbands; c++) sharing += (t->dev[c] == t->dev[b]);
        rc = plan_create_ex(&t->pl[b], nx, ny, nchan, &pb, (sharing > 1 || !allow_selfcheck) ? 1 : 8, true);
        if (rc != OCTANE_OK) break;
        if (hipSetDevice(t->dev[b]) != hipSuccess ||
            hipMalloc((void **)&t->parts[b], (size_t)2 * kPartBlock * sizeof(double)) != hipSuccess ||
            hipMemset(t->parts[b], 0, (size_t)2 * kPartBlock * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&t->mirror[b], (size_t)nbands * 2 * kPartBlock * sizeof(double)) != hipSuccess ||
            hipMemset(t->mirror[b], 0, (size_t)nbands * 2 * kPartBlock * sizeof(double)) != hipSuccess ||
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
    // Which transport.  Forced by the environment; else the first-contact self-check decides (thread form: inplace -> inplace without
    // LDS-DMA from the neighbour -> copy); without the check: inplace where every peer is reachable, copy otherwise.
    t->transport = t->info.peer_ok ? OCTANE_TRANSPORT_INPLACE : OCTANE_TRANSPORT_COPY;
    const int forced = forced_transport();
    if (forced == -2 || forced == OCTANE_TRANSPORT_COLLECTIVE) {
        set_last_error("octane_vof_tiled_create: OCTANE_TILED_TRANSPORT must be inplace or copy here (collective: one band per process, octane_vof_mp_*)");
        octane_vof_tiled_destroy(t);
        return OCTANE_E_INVALID;
    }
    if (forced == OCTANE_TRANSPORT_INPLACE && !t->info.peer_ok) {
        set_last_error("octane_vof_tiled_create: OCTANE_TILED_TRANSPORT=inplace, but " + peer_note);
        octane_vof_tiled_destroy(t);
        return OCTANE_E_INVALID;
    }
    if (forced >= 0) { t->transport = forced; t->info.forced = 1; }
    else if (!t->info.peer_ok && allow_selfcheck)
        fprintf(stderr, "octane: row bands: %s -- the in-place transport is not available, using runtime peer copies\n", peer_note.c_str());
    t->info.transport = t->transport; t->info.q_dma = 1;
    bool any_banded = false;
    for (auto &r : t->rows) any_banded = any_banded || !r.empty();
    if (allow_selfcheck && forced < 0 && nbands > 1 && any_banded && selfcheck_enabled()) {
        rc = tiled_selfcheck(t);
        if (rc != OCTANE_OK) { octane_vof_tiled_destroy(t); return rc; }
    }
    *out = t;
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_transport_info(const octane_vof_tiled *t, octane_vof_transport_info *out)
{
    if (!t || !out) return OCTANE_E_INVALID;
    *out = t->info;
    out->transport = t->transport; out->q_dma = t->no_dma ? 0 : 1;
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
    if (t->aborted) return -2;                 // a persistent solve of the last solve gave up: see octane_vof_tiled_wait
    for (int b = 0; b < t->nbands; b++)
        if (t->pl[b]->h_mid_abort && *t->pl[b]->h_mid_abort != 0) return -2;
    return *t->pl[0]->h_iters;
}

extern "C" long long octane_vof_tiled_last_copies(octane_vof_tiled *t) { return t ? t->copies : -1; }

extern "C" size_t octane_vof_tiled_device_bytes(const octane_vof_tiled *t)
{
    return t ? t->pl[0]->arena_bytes + t->pl[0]->xarena_bytes + (size_t)2 * kPartBlock * sizeof(double) : 0;
}

// ---- how a band reaches the others ---------------------------------------------------------------------------------
// The level loop below is written once against this interface and runs in two settings:
//   * ThreadNet -- all bands in one process, one host thread per band (octane_vof_tiled_*): phase boundaries are events
//     recorded and waited on across the bands' streams, the threads only meet at a spin barrier so that "record" is
//     issued before "wait"; the host never waits for the GPU inside a pyramid;
//   * ProcNet   -- one band per process (octane_vof_mp_*, the torchrun launch): the other bands' arenas and partial
//     blocks are HIP IPC mappings, a phase boundary is "drain my stream, then meet the other processes at a barrier in
//     POSIX shared memory".
// Every band's arena is carved identically, so a plane of band c is band b's plane pointer moved by the distance
// between the two arena bases as mapped in this process.
struct SpinBarrier {
    std::atomic<int> arrived{0};
    std::atomic<int> generation{0};
    // returns false when `timeout_s` (0 = none) passed without everybody arriving, or when *abort became non-zero while
    // waiting.  A wait that gives up leaves its arrival counted: the barrier is not to be used again afterwards (the
    // process form marks itself dead and stops synchronising, ProcNet::sync).
    bool wait(int n, double timeout_s = 0., const std::atomic<int> *abort = nullptr)
    {
        const int gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            arrived.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
            return true;
        }
        long spins = 0;
        std::chrono::steady_clock::time_point t0;
        while (generation.load(std::memory_order_acquire) == gen) {
            if (++spins > 4096) {
                std::this_thread::yield();
                if ((spins & 1023) == 0) {
                    if (abort && abort->load(std::memory_order_acquire) != 0) return false;
                    if (timeout_s > 0.) {
                        if (spins == 5120) t0 = std::chrono::steady_clock::now();
                        else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
                    }
                }
            }
        }
        return true;
    }
};

struct BandNet {
    int nb = 1;
    octane_vof_params prm;
    const std::vector<std::vector<BandRows>> *rows = nullptr;     // [level][band]; empty = replicated level
    char *arena[kMaxBands] = {nullptr};                            // every band's arena base as mapped in this process
    double *parts[kMaxBands] = {nullptr};                          // every band's partial block
    std::vector<long long> copies = std::vector<long long>(kMaxBands, 0);
    std::vector<int> cur = std::vector<int>(kMaxBands, 0);
    virtual ~BandNet() {}
    virtual octane_vof_plan *plan(int b) = 0;                      // band b's plan (bands of this process only)
    virtual bool failed() = 0;
    virtual void fail(int b, int rc, const std::string &msg) = 0;
    virtual void sync(int b) = 0;                                  // phase boundary, called by every band
    virtual hipError_t copy(int b, void *dst, int dband, const void *src, size_t bytes) = 0;   // on band b's stream
    // copy transport: from band `sband`'s memory (as mapped here) into band b's own memory, on band b's stream
    virtual hipError_t pull(int b, void *dst_local, int sband, const void *src, size_t bytes) = 0;
    virtual double *mirror(int b) = 0;                             // band b's local copies of all bands' partial blocks
    // collective transport (process form only): the host program's collective library moves the bytes (octane_vof_exchange)
    virtual int xchg_all_gather(int, const void *, void *const *, size_t) { return -1; }
    virtual int xchg_sendrecv(int, int, const octane_vof_xfer *) { return -1; }
    virtual void level_mark(int, int) {}                           // band b is about to start level k (k == levels: the pyramid is issued); solo-band timing records an event
    int transport = OCTANE_TRANSPORT_INPLACE;
    bool no_dma = false;
    int brk = 0;       // diagnostic library only: OCTANE_TEST_BREAK_TRANSPORT, the drill of the self-check's downgrade
    bool solo = false; // diagnostic library only: solo-band timing (SoloNet): the "neighbours' rows" are this band's own rows, mirrored at its edges
    float *peer(int c, int b, float *plane_of_b) const
    {
        return reinterpret_cast<float *>(arena[c] + (reinterpret_cast<char *>(plane_of_b) - arena[b]));
    }
};

// HIP calls of a band: after the first failure anywhere every band keeps walking the same control flow (so the phase
// boundaries still match up) but issues nothing more.
#define BAND_HIP(expr)                                                                         \
    do {                                                                                       \
        if (!N.failed()) {                                                                     \
            hipError_t e_ = (expr);                                                            \
            if (e_ != hipSuccess) N.fail(b, OCTANE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)

// Rows [ya, yb) of one of band b's planes to the same plane of band dst.
static void send_rows(BandNet &N, int b, float *plane, int pitch, int ya, int yb, int dst)
{
    N.copies[b]++;
    BAND_HIP(N.copy(b, N.peer(dst, b, plane) + (size_t)ya * pitch, dst, plane + (size_t)ya * pitch, (size_t)(yb - ya) * pitch * sizeof(float)));
}

// copy transport: rows [ya, yb) of band `src`'s copy of one of band b's planes into band b's own plane (same rows), and the
// partial block(s) of every other band into band b's mirror
static void pull_rows(BandNet &N, int b, int src, float *plane, int pitch, int ya, int yb)
{
    if (yb <= ya) return;
    N.copies[b]++;
    BAND_HIP(N.pull(b, plane + (size_t)ya * pitch, src, N.peer(src, b, plane) + (size_t)ya * pitch, (size_t)(yb - ya) * pitch * sizeof(float)));
}
static void pull_parts(BandNet &N, int b, int parity, int first, int count)
{
    for (int c = 0; c < N.nb; c++) {
        if (c == b) continue;
        N.copies[b]++;
        const size_t off = (size_t)parity * kPartBlock + first;
        BAND_HIP(N.pull(b, N.mirror(b) + (size_t)c * 2 * kPartBlock + off, c, N.parts[c] + off, (size_t)count * sizeof(double)));
    }
}

// collective transport: the same bytes by the host program's collective library.  Every rank calls these at the same points of the
// level loop with mirrored arguments; the stream has been drained (N.sync) and the calls return when the data is in place.
static void gather_parts(BandNet &N, int b, int parity, int first, int count)
{
    if (N.failed()) return;
    const size_t off = (size_t)parity * kPartBlock + first;
    void *recv[kMaxBands];
    for (int c = 0; c < N.nb; c++) recv[c] = (c == b) ? nullptr : (void *)(N.mirror(b) + (size_t)c * 2 * kPartBlock + off);
    N.copies[b]++;
    if (N.xchg_all_gather(b, N.parts[b] + off, recv, (size_t)count * sizeof(double)) != 0)
        N.fail(b, OCTANE_E_HIP, "the collective transport's all_gather of the partial sums failed");
}
// What band b needs of a plane beyond its edges -- rows [y0 - a_hi, y0 - a_lo) from the band above, [y1 + b_lo, y1 + b_hi) from
// the band below -- is exactly what its neighbours need of it, mirrored: it sends rows [y1 - a_hi, y1 - a_lo) down and
// [y0 + b_lo, y0 + b_hi) up.  Rows of a plane are contiguous, so every transfer is one buffer.
struct RowSpec { float *plane; int a_lo, a_hi, b_lo, b_hi; };
static void exchange_rows(BandNet &N, int b, const RowSpec *spec, int nspec, int pitch, int y0, int y1)
{
    if (N.failed()) return;
    std::vector<octane_vof_xfer> ops;
    auto add = [&](int peer, int send, float *plane, int ya, int yb) {
        if (yb <= ya) return;
        octane_vof_xfer x; x.peer = peer; x.send = send; x.buf = plane + (size_t)ya * pitch; x.bytes = (size_t)(yb - ya) * pitch * sizeof(float);
        ops.push_back(x);
    };
    for (int i = 0; i < nspec; i++) {
        const RowSpec &r = spec[i];
        if (b > 0) { add(b - 1, 0, r.plane, y0 - r.a_hi, y0 - r.a_lo); add(b - 1, 1, r.plane, y0 + r.b_lo, y0 + r.b_hi); }
        if (b < N.nb - 1) { add(b + 1, 0, r.plane, y1 + r.b_lo, y1 + r.b_hi); add(b + 1, 1, r.plane, y1 - r.a_hi, y1 - r.a_lo); }
    }
    if (ops.empty()) return;
    N.copies[b] += (long long)ops.size();
    if (N.xchg_sendrecv(b, (int)ops.size(), ops.data()) != 0) N.fail(b, OCTANE_E_HIP, "the collective transport's sendrecv of edge rows failed");
}

// ---- one banded level, as band b issues it ----------------------------------------------------------------------------
static void solve_level_banded(BandNet &N, int b, int k, int cur, const LevelCtx &ctx, bool finest)
{
    const int nb = N.nb;
    const octane_vof_params &prm = N.prm;
    octane_vof_plan *pl = N.plan(b);
    hipStream_t s = pl->own_stream;
    const LevelInfo &li = pl->lev[k];
    const std::vector<BandRows> &rows = (*N.rows)[k];
    LevelPtrs L;
    plan_fill_level_ptrs(pl, k, cur, ctx, L);
    L.y0 = rows[b].y0; L.y1 = rows[b].y1;
    L.ya0 = (b == 0) ? 0 : L.y0 - 1;
    L.ya1 = (b == nb - 1) ? li.h : L.y1 + 1;
    L.nbands = nb;
    L.no_dma = N.no_dma ? 1 : 0;
    double *own = N.parts[b];
    L.part_rz = own + kPartRz; L.part_rr = own + kPartRr; L.part_pq = own + kPartPq; L.part_own = own;
    const int tr = N.transport;
    const bool coll = tr == OCTANE_TRANSPORT_COLLECTIVE;
    const bool cp = tr != OCTANE_TRANSPORT_INPLACE;    // neighbours' rows and partial blocks are brought into local memory, kernels read local
    for (int c = 0; c < kMaxBands; c++) {
        const int cc = c < nb ? c : b;
        L.band_parts[c] = (cp && cc != b) ? N.mirror(b) + (size_t)cc * 2 * kPartBlock : N.parts[cc];
    }
    const int up = b > 0 ? b - 1 : b, dn = b < nb - 1 ? b + 1 : b;
    const int rup = cp ? b : up, rdn = cp ? b : dn;    // whose planes the kernels read the rows beyond the band's edges from
    // (solo-band timing, diagnostic library: every "other band" is this band's own arena; reading row y0 - k of the band above then means
    // reading this band's row y0 - k + 2, i.e. its own edge rows again, and row y1 + k of the band below its row y1 + k - 2 -- the same
    // number of bytes at the same distance from the edge, values that stay finite and move with the solve instead of never-written rows)
    const ptrdiff_t sh_up = (N.solo && b > 0) ? (ptrdiff_t)2 * li.pitch : 0, sh_dn = (N.solo && b < nb - 1) ? -(ptrdiff_t)2 * li.pitch : 0;
    L.ru_up = N.peer(rup, b, pl->ru) + sh_up; L.rv_up = N.peer(rup, b, pl->rv) + sh_up;
    L.ru_dn = N.peer(rdn, b, pl->ru) + sh_dn; L.rv_dn = N.peer(rdn, b, pl->rv) + sh_dn;
    int maxrows = 0;
    for (int c = 0; c < nb; c++) maxrows = rows[c].y1 - rows[c].y0 > maxrows ? rows[c].y1 - rows[c].y0 : maxrows;
    // every band launches the same grids, so that the blocks hold the same number of partials (idle workgroups
    // contribute zeros)
    const int g_asm = assemble_grid_size(li.w, maxrows + 2);
    const int g_a = pcg_band_grid_size(li.w, maxrows);
    const int g_b = pcg_b_grid_size(li.w, maxrows);
    const bool fused = pl->use_fused != 0;
    // one form of the fused kernel for all bands (they read each other's planes): decided on the largest band
    L.q_form = (fused && pcg_fused_q_form(li.w, maxrows, li.h)) ? 1 : 0;
    if (fused && !L.q_form)   // q of the neighbouring bands' edge rows is read from their planes (both halves of the double buffer)
        for (int i = 0; i < 2; i++) {
            L.qup_u[i] = N.peer(rup, b, L.qb_u[i]) + sh_up; L.qup_v[i] = N.peer(rup, b, L.qb_v[i]) + sh_up;
            L.qdn_u[i] = N.peer(rdn, b, L.qb_u[i]) + sh_dn; L.qdn_v[i] = N.peer(rdn, b, L.qb_v[i]) + sh_dn;
        }
    if (L.q_form) {           // q is recomputed: r on the row beyond an edge, p on the two rows beyond it, wy of the row above the upper one
        for (int i = 0; i < 2; i++) {
            L.rup_u[i] = N.peer(rup, b, L.rb_u[i]) + sh_up; L.rup_v[i] = N.peer(rup, b, L.rb_v[i]) + sh_up;
            L.rdn_u[i] = N.peer(rdn, b, L.rb_u[i]) + sh_dn; L.rdn_v[i] = N.peer(rdn, b, L.rb_v[i]) + sh_dn;
        }
        for (int i = 0; i < 3; i++) {
            L.pup_u[i] = N.peer(rup, b, L.pf_u[i]) + sh_up; L.pup_v[i] = N.peer(rup, b, L.pf_v[i]) + sh_up;
            L.pdn_u[i] = N.peer(rdn, b, L.pf_u[i]) + sh_dn; L.pdn_v[i] = N.peer(rdn, b, L.pf_v[i]) + sh_dn;
        }
        L.wy_up = N.peer(rup, b, pl->wy) + sh_up;
    }
#ifdef OCTANE_DIAG
    // Drill of the self-check's downgrade (diagnostic library only, OCTANE_TEST_BREAK_TRANSPORT): bit 0 -- the in-place transport
    // folds its OWN partial block in place of band 0's / band nb-1's (wrong alpha and beta: what a peer mapping that returns stale
    // data looks like); bit 1 -- with the LDS-DMA kernel the rows beyond the lower edge come from this band's own, never-written halo
    // rows; bit 2 -- the copy transport leaves the other bands' partial blocks in the mirror unfilled.
    if ((N.brk & 1) && !cp && nb > 1) L.band_parts[b == 0 ? nb - 1 : 0] = N.parts[b];
    if ((N.brk & 2) && !cp && L.q_form && !N.no_dma && b < nb - 1)
        for (int i = 0; i < 3; i++) { L.pdn_u[i] = L.pf_u[i]; L.pdn_v[i] = L.pf_v[i]; }
#endif

    for (int gnc = 0; gnc < 3; gnc++) {                 // ref .cu:604-606
        AssembleParams ap;
        ap.al1 = 1. - 0.5 * gnc;
        ap.alpha = prm.alpha;
        ap.loa = prm.lambda / prm.alpha;                // ref .cu:1230
        ap.lambdac = li.lambdac;
        ap.dozim = prm.dozim != 0;
        ap.ralpha = 1. / prm.alpha;                     // correctly rounded (host division)
        ap.fast_math = pl->asm_fast;
        L.unit_w = (pl->use_unit_w && gnc == 0) ? 1 : 0;
        L.lean = fused ? 1 : 0;
        const int g_f = pcg_fused_grid_size(li.w, maxrows, L.unit_w, L.q_form);
        for (int l = 0; l < prm.liters; l++) {          // ref .cu:608
            if (!N.failed()) {
                LevelPtrs La = L;
                if (fused) { La.part_rz = L.part_own + kPartBlock + kPartRz; La.part_rr = L.part_own + kPartBlock + kPartRr; }
                launch_assemble(s, La, ap, g_asm);
            }
            N.sync(b);
            if (cp) {      // what the first launch reads of the others: the sums of their right-hand sides, wy of the row above the upper ring row
                if (!fused) N.fail(b, OCTANE_E_INVALID, "the copy and collective transports need the one-kernel PCG iteration (OCTANE_TUNE_FUSED=1)");
                if (coll) {
                    gather_parts(N, b, 1, kPartRz, 2 * kMaxParts);
                    if (L.q_form) { const RowSpec wy = {pl->wy, 1, 2, 0, 0}; exchange_rows(N, b, &wy, 1, li.pitch, L.y0, L.y1); }
                } else {
#ifdef OCTANE_DIAG
                    if (!(N.brk & 4))
#endif
                    pull_parts(N, b, 1, kPartRz, 2 * kMaxParts);
                    if (L.q_form && b > 0 && L.y0 >= 2) pull_rows(N, b, up, pl->wy, li.pitch, L.y0 - 2, L.y0 - 1);
                }
            }
            if (fused) {
                for (int it = 0; it < prm.cgiters; it++) {  // ref .cu:1131-1182: one kernel and one boundary per iteration
                    if (!N.failed()) launch_pcg_fused(s, L, it, it == 0 ? g_asm : g_f, g_f, pl->tol);
                    N.sync(b);
                    if (coll) {   // the same bytes as below, moved by the host program's collective library
                        gather_parts(N, b, it & 1, 0, kPartBlock);
                        if (L.q_form) {
                            const RowSpec sp[4] = {{L.pf_u[it % 3], 0, 2, 0, 2}, {L.pf_v[it % 3], 0, 2, 0, 2},
                                                   {L.rb_u[it & 1], 0, 1, 0, 1}, {L.rb_v[it & 1], 0, 1, 0, 1}};
                            exchange_rows(N, b, sp, 4, li.pitch, L.y0, L.y1);
                        } else {
                            const RowSpec sp[2] = {{L.qb_u[it & 1], 0, 1, 0, 1}, {L.qb_v[it & 1], 0, 1, 0, 1}};
                            exchange_rows(N, b, sp, 2, li.pitch, L.y0, L.y1);
                        }
                    } else if (cp) {   // what launch it + 1 (or the flow update) reads of the others: their sums, the rows beyond this band's edges
#ifdef OCTANE_DIAG
                        if (!(N.brk & 4))
#endif
                        pull_parts(N, b, it & 1, 0, kPartBlock);
                        for (int side = 0; side < 2; side++) {
                            const int nbr = side == 0 ? up : dn;
                            if (nbr == b) continue;
                            const int e0 = side == 0 ? L.y0 : L.y1;             // the edge: rows e0 - 2, e0 - 1 lie above it, e0, e0 + 1 below
                            if (L.q_form) {
                                const int ya = side == 0 ? e0 - 2 : e0, yb = side == 0 ? e0 : e0 + 2;
                                const int ra = side == 0 ? e0 - 1 : e0;
                                pull_rows(N, b, nbr, L.pf_u[it % 3], li.pitch, ya < 0 ? 0 : ya, yb > li.h ? li.h : yb);
                                pull_rows(N, b, nbr, L.pf_v[it % 3], li.pitch, ya < 0 ? 0 : ya, yb > li.h ? li.h : yb);
                                pull_rows(N, b, nbr, L.rb_u[it & 1], li.pitch, ra, ra + 1);
                                pull_rows(N, b, nbr, L.rb_v[it & 1], li.pitch, ra, ra + 1);
                            } else {
                                const int qa = side == 0 ? e0 - 1 : e0;
                                pull_rows(N, b, nbr, L.qb_u[it & 1], li.pitch, qa, qa + 1);
                                pull_rows(N, b, nbr, L.qb_v[it & 1], li.pitch, qa, qa + 1);
                            }
                        }
                    }
                }
                if (!N.failed()) launch_flow_update_fused(s, L, prm.cgiters, g_f);     // ref .cu:1185-1195
            } else {
#ifdef OCTANE_DIAG      // the two-pass form: diagnostic library only
                for (int it = 0; it < prm.cgiters; it++) {
                    if (!N.failed()) launch_pcg_pass_a(s, L, it, it == 0 ? g_asm : g_b, g_a, pl->tol);
                    N.sync(b);
                    if (!N.failed()) launch_pcg_pass_b(s, L, it, g_a, g_b);
                    N.sync(b);
                }
                if (!N.failed()) launch_flow_update(s, L, prm.cgiters);
#else
                (void)g_a; (void)g_b;
                N.fail(b, OCTANE_E_INVALID, "the two-pass form of the PCG iteration is built into the diagnostic library only");
#endif
            }
            // the next assembly reads u, v two rows beyond the band (one for the halo row it fills, one for that
            // row's own 3 x 3 neighbourhood)
            if (coll) {
                N.sync(b);
                const RowSpec sp[2] = {{pl->U[cur], 0, 2, 0, 2}, {pl->V[cur], 0, 2, 0, 2}};
                exchange_rows(N, b, sp, 2, li.pitch, L.y0, L.y1);
            } else {
                if (b > 0) {
                    send_rows(N, b, pl->U[cur], li.pitch, L.y0, L.y0 + 2, b - 1);
                    send_rows(N, b, pl->V[cur], li.pitch, L.y0, L.y0 + 2, b - 1);
                }
                if (b < nb - 1) {
                    send_rows(N, b, pl->U[cur], li.pitch, L.y1 - 2, L.y1, b + 1);
                    send_rows(N, b, pl->V[cur], li.pitch, L.y1 - 2, L.y1, b + 1);
                }
                N.sync(b);
            }
        }
    }
    // Level done: the next level's up-sampling (replicated) needs the whole flow on every band; after the finest
    // level only band 0, which hands the result out, does.
    if (coll) {
        std::vector<octane_vof_xfer> ops;
        auto add = [&](int peer, int send, float *plane, int ya, int yb) {
            octane_vof_xfer x; x.peer = peer; x.send = send; x.buf = plane + (size_t)ya * li.pitch; x.bytes = (size_t)(yb - ya) * li.pitch * sizeof(float);
            ops.push_back(x);
        };
        for (int c = 0; c < nb; c++) {
            if (c == b) continue;
            if (!(finest && c != 0)) { add(c, 1, pl->U[cur], L.y0, L.y1); add(c, 1, pl->V[cur], L.y0, L.y1); }
            if (!(finest && b != 0)) { add(c, 0, pl->U[cur], rows[c].y0, rows[c].y1); add(c, 0, pl->V[cur], rows[c].y0, rows[c].y1); }
        }
        N.copies[b] += (long long)ops.size();
        if (!N.failed() && !ops.empty() && N.xchg_sendrecv(b, (int)ops.size(), ops.data()) != 0)
            N.fail(b, OCTANE_E_HIP, "the collective transport's gather of the flow bands failed");
    } else {
        for (int c = 0; c < nb; c++) {
            if (c == b || (finest && c != 0)) continue;
            send_rows(N, b, pl->U[cur], li.pitch, L.y0, L.y1, c);
            send_rows(N, b, pl->V[cur], li.pitch, L.y0, L.y1, c);
        }
    }
    N.sync(b);
}

// The whole pyramid of band b.
static void band_worker(BandNet &N, int b)
{
    octane_vof_plan *pl = N.plan(b);
    const int nlev = (int)pl->lev.size();
    BAND_HIP(hipSetDevice(pl->device));
    BAND_HIP(hipMemsetAsync(pl->d_iters, 0, sizeof(long long), pl->own_stream));
    // replicated levels run the persistent mid-level solve through plan_level_solve: its abort word is per run (vof_plan.hpp)
    if (persist_begin_run(pl, pl->own_stream) != OCTANE_OK) N.fail(b, OCTANE_E_HIP, "clearing the persistent solve's abort word failed");
    pl->evs_used = 0;
    int cur = 0;
    LevelCtx ctx;
    for (int k = 0; k < nlev; k++) {
        N.level_mark(b, k);
        if (!N.failed()) {
            const int rc = plan_level_setup(pl, pl->own_stream, k, cur, ctx);
            if (rc) N.fail(b, rc, "plan_level_setup failed");
        } else if (k > 0) {
            cur ^= 1;      // keep the bookkeeping in step with the healthy bands
        }
        if ((*N.rows)[k].empty()) {
            if (!N.failed()) {
                const int rc = plan_level_solve(pl, pl->own_stream, k, cur, ctx, false);
                if (rc) N.fail(b, rc, "plan_level_solve failed");
            }
        } else {
            N.sync(b);                 // a band's halo rows must not be written while it still sets the level up
            solve_level_banded(N, b, k, cur, ctx, k == nlev - 1);
        }
    }
    N.level_mark(b, nlev);
    if (b == 0) BAND_HIP(hipMemcpyAsync(pl->h_iters, pl->d_iters, sizeof(long long), hipMemcpyDeviceToHost, pl->own_stream));
    if (persist_end_run(pl, pl->own_stream) != OCTANE_OK) N.fail(b, OCTANE_E_HIP, "fetching the persistent solve's abort word failed");
    BAND_HIP(hipGetLastError());
    N.cur[b] = cur;
}

// ---- all bands in this process: one host thread per band --------------------------------------------------------
struct ThreadNet : BandNet {
    octane_vof_tiled *t;
    SpinBarrier bar;
    std::atomic<int> failed_{0};     // first failing band + 1
    std::mutex mu;
    std::string error;
    int rc = OCTANE_OK;
    explicit ThreadNet(octane_vof_tiled *t_) : t(t_)
    {
        nb = t->nbands; prm = t->prm; rows = &t->rows;
        for (int b = 0; b < nb; b++) { arena[b] = reinterpret_cast<char *>(t->pl[b]->shared_base()); parts[b] = t->parts[b]; }
        transport = t->transport; no_dma = t->no_dma; brk = test_break_bits();
    }
    double *mirror(int b) override { return t->mirror[b]; }
    hipError_t pull(int b, void *dst_local, int sband, const void *src, size_t bytes) override
    {
        hipStream_t s = t->pl[b]->own_stream;
        if (t->dev[sband] == t->dev[b]) return hipMemcpyAsync(dst_local, src, bytes, hipMemcpyDeviceToDevice, s);
        return hipMemcpyPeerAsync(dst_local, t->dev[b], src, t->dev[sband], bytes, s);
    }
    octane_vof_plan *plan(int b) override { return t->pl[b]; }
    bool failed() override { return failed_.load(std::memory_order_relaxed) != 0; }
    void fail(int b, int code, const std::string &msg) override
    {
        std::lock_guard<std::mutex> g(mu);
        if (!failed_.load()) { rc = code; error = msg; failed_.store(b + 1); }
    }
    // Everything band b has issued so far becomes a dependency of whatever the other bands issue next, and vice versa.
    void sync(int b) override
    {
        BandNet &N = *this;
        BAND_HIP(hipEventRecord(t->ev[b], t->pl[b]->own_stream));
        bar.wait(nb);
        for (int c = 0; c < nb; c++)
            if (c != b) BAND_HIP(hipStreamWaitEvent(t->pl[b]->own_stream, t->ev[c], 0));
        bar.wait(nb);       // nobody re-records its event before everybody has queued the waits on it
    }
    hipError_t copy(int b, void *dst, int dband, const void *src, size_t bytes) override
    {
        hipStream_t s = t->pl[b]->own_stream;
        if (t->dev[dband] == t->dev[b]) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
        return hipMemcpyPeerAsync(dst, t->dev[dband], src, t->dev[b], bytes, s);
    }
};

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
    t->aborted = false;
    ThreadNet N(t);
    if (nb == 1) {
        band_worker(N, 0);
    } else {
        std::vector<std::thread> th;
        for (int b = 0; b < nb; b++) th.emplace_back([&N, b]() { band_worker(N, b); });
        for (auto &x : th) x.join();
    }
    t->copies = 0;
    for (int b = 0; b < nb; b++) t->copies += N.copies[b];
    if (N.failed()) {
        set_last_error("octane_vof_tiled_solve (band " + std::to_string(N.failed_.load() - 1) + "): " + N.error);
        (void)octane_vof_tiled_wait(t);
        return N.rc;
    }
    (void)hipSetDevice(t->dev[0]);
    t->last_cur = N.cur[0];
    return OCTANE_OK;
}

extern "C" int octane_vof_tiled_wait(octane_vof_tiled *t)
{
    if (!t) return OCTANE_E_INVALID;
    for (int b = 0; b < t->nbands; b++) {
        TILED_TRY(hipSetDevice(t->dev[b]));
        TILED_TRY(hipStreamSynchronize(t->pl[b]->own_stream));
    }
    // A persistent solve of a replicated level that gave up on ANY band (its workgroups return without applying the update) makes
    // the whole frame invalid: the bands no longer hold the same bits.  Every band's word is checked and cleared.
    int rc = OCTANE_OK;
    for (int b = 0; b < t->nbands; b++) {
        TILED_TRY(hipSetDevice(t->dev[b]));
        if (persist_check(t->pl[b]) != OCTANE_OK) rc = OCTANE_E_HIP;      // sets octane_last_error
    }
    if (rc != OCTANE_OK) t->aborted = true;
    (void)hipSetDevice(t->dev[0]);
    return t->aborted ? OCTANE_E_HIP : OCTANE_OK;
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

#ifdef OCTANE_DIAG
// ---- solo-band timing (diagnostic library only; round 5) --------------------------------------------------------
// The pool this is developed on gives one GPU per box, so an N-GPU solve of one frame cannot be timed.  What CAN be measured on
// hardware is the compute term of the scaling model: band b's OWN launch sequence of an N-band solve -- the replicated level set-ups and
// coarse levels, its band's assemblies, PCG launches, flow updates, event records at every phase boundary -- with the neighbours' rows
// and partial blocks left static (every "other band" is mapped onto this band's own arena and partial block: the kernels read the
// same number of bytes from the same row offsets they would read from a neighbour, only the values are wrong).  Wrong flow, right
// timeline.  The arena is zeroed first (the creation poison is NaN: a NaN in a halo row would trip the stop test and empty the launches),
// and the run is only a measurement if every solve ran its full iteration count, which is returned for the caller to check.
// What it cannot show: the cost of waiting for the slowest band at a boundary, xGMI latency of the in-kernel peer reads, and the
// peer copies (their bytes are counted and returned for tools/tiled_model.py to price).
struct SoloNet : BandNet {
    octane_vof_plan *pl = nullptr;
    double *mir = nullptr;
    hipEvent_t ev = nullptr;
    int band = 0, rc = OCTANE_OK;
    bool bad = false;
    std::string error;
    long long peer_bytes = 0, boundaries = 0;
    std::vector<hipEvent_t> lev_ev;            // one timing event per level start + one at the end of the pyramid
    void level_mark(int, int k) override
    {
        if ((int)lev_ev.size() <= k) { lev_ev.resize(k + 1, nullptr); }
        if (!lev_ev[k] && hipEventCreate(&lev_ev[k]) != hipSuccess) { lev_ev[k] = nullptr; return; }
        (void)hipEventRecord(lev_ev[k], pl->own_stream);
    }
    octane_vof_plan *plan(int) override { return pl; }
    bool failed() override { return bad; }
    void fail(int, int code, const std::string &msg) override { if (!bad) { bad = true; rc = code; error = msg; } }
    double *mirror(int) override { return mir; }
    void sync(int b) override
    {   // what a boundary costs THIS band's host thread and stream when nobody is late: record, then a wait that is already satisfied
        BandNet &N = *this;
        boundaries++;
        BAND_HIP(hipEventRecord(ev, pl->own_stream));
        BAND_HIP(hipStreamWaitEvent(pl->own_stream, ev, 0));
    }
    // A send to a neighbouring band would cross xGMI: counted, not made.  The two rows of u, v a band SENDS per inner edge after a flow
    // update are also what it would RECEIVE there: the same rows are copied into its own halo rows beyond that edge (a local copy of the
    // same size), so that the next assembly reads a flow that moves with the band instead of the previous level's.  (Halo sends are two
    // rows, < 1 MB; the flow bands gathered at the end of a level are whole bands and are only counted.)
    hipError_t copy(int b, void *dst, int dband, const void *src, size_t bytes) override
    {
        peer_bytes += (long long)bytes;
        if (bytes >= ((size_t)1 << 20) || (dband != b - 1 && dband != b + 1)) return hipSuccess;
        char *halo = const_cast<char *>(static_cast<const char *>(src)) + (dband < b ? -(ptrdiff_t)bytes : (ptrdiff_t)bytes);
        (void)dst;
        return hipMemcpyAsync(halo, src, bytes, hipMemcpyDeviceToDevice, pl->own_stream);
    }
    hipError_t pull(int, void *, int, const void *, size_t bytes) override { peer_bytes += (long long)bytes; return hipSuccess; }
};

extern "C" int octane_vof_solo_band_time(int nx, int ny, int nchan, const octane_vof_params *p, int nbands, int band, long long min_band_pixels,
                                         const float *img1, const float *img2, int mem, int reps, double *ms_out,
                                         long long *iterations, long long *peer_copy_bytes, long long *boundaries, int *banded_levels,
                                         int *band_rows_finest, double *level_ms /* [32]: GPU time of each level (coarsest first) in the LAST repetition, or NULL */)
{
    if (!p || !img1 || !img2 || !ms_out || nbands < 1 || nbands > kMaxBands || band < 0 || band >= nbands || reps < 1) return OCTANE_E_INVALID;
    octane_vof_plan *pl = nullptr;
    int rc = plan_create_ex(&pl, nx, ny, nchan, p, 8, true);
    if (rc != OCTANE_OK) return rc;
    SoloNet N;
    N.pl = pl; N.band = band; N.nb = nbands; N.prm = *p; N.solo = true;
    pl->tol = -1.f;        // OCT_STOP_HELD_OPEN: the band's fused PCG launches all do their work, whatever the inconsistent sums say
    std::vector<std::vector<BandRows>> rows(pl->lev.size());
    const long minpix = min_band_pixels > 0 ? (long)min_band_pixels : kDefaultMinBandPixels;
    int nbanded = 0;
    for (size_t k = 0; k < pl->lev.size(); k++) {
        const LevelInfo &li = pl->lev[k];
        int edges[kMaxBands + 1];
        if ((long)li.w * li.h < minpix || octane_vof_band_partition(li.h, nbands, edges) != 1) continue;
        std::vector<BandRows> r(nbands);
        for (int b = 0; b < nbands; b++) { r[b].y0 = edges[b]; r[b].y1 = edges[b + 1]; }
        rows[k] = r; nbanded++;
    }
    N.rows = &rows;
    double *parts = nullptr;
    auto cleanup = [&]() {
        if (parts) (void)hipFree(parts);
        if (N.mir) (void)hipFree(N.mir);
        if (N.ev) (void)hipEventDestroy(N.ev);
        octane_vof_plan_destroy(pl);
    };
    if (hipSetDevice(pl->device) != hipSuccess ||
        hipMalloc((void **)&parts, (size_t)2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipMemset(parts, 0, (size_t)2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&N.mir, (size_t)nbands * 2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipMemset(N.mir, 0, (size_t)nbands * 2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipEventCreateWithFlags(&N.ev, hipEventDisableTiming) != hipSuccess ||
        hipMemset(pl->arena, 0, pl->arena_bytes) != hipSuccess || (pl->xarena && hipMemset(pl->xarena, 0, pl->xarena_bytes) != hipSuccess) ||
        hipDeviceSynchronize() != hipSuccess) {
        set_last_error("octane_vof_solo_band_time: device allocation failed");
        cleanup();
        return OCTANE_E_NOMEM;
    }
    for (int c = 0; c < kMaxBands; c++) { N.arena[c] = reinterpret_cast<char *>(pl->shared_base()); N.parts[c] = parts; }
    N.transport = OCTANE_TRANSPORT_INPLACE;
    rc = plan_load_inputs(pl, img1, img2, nullptr, nullptr, mem, pl->own_stream);
    if (rc == OCTANE_OK && hipStreamSynchronize(pl->own_stream) != hipSuccess) rc = OCTANE_E_HIP;
    for (int r = 0; r < reps && rc == OCTANE_OK; r++) {
        N.peer_bytes = 0; N.boundaries = 0;
        const auto t0 = std::chrono::steady_clock::now();
        band_worker(N, band);
        if (band != 0) (void)hipMemcpyAsync(pl->h_iters, pl->d_iters, sizeof(long long), hipMemcpyDeviceToHost, pl->own_stream);
        if (hipStreamSynchronize(pl->own_stream) != hipSuccess) rc = OCTANE_E_HIP;
        ms_out[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (N.bad) { set_last_error("octane_vof_solo_band_time: " + N.error); rc = N.rc; }
        if (rc == OCTANE_OK && persist_check(pl) != OCTANE_OK) rc = OCTANE_E_HIP;
    }
    if (level_ms) {
        for (int k = 0; k < 32; k++) level_ms[k] = -1.;
        for (size_t k = 0; k + 1 < N.lev_ev.size() && k < 32; k++) {
            float ms = 0.f;
            if (N.lev_ev[k] && N.lev_ev[k + 1] && hipEventElapsedTime(&ms, N.lev_ev[k], N.lev_ev[k + 1]) == hipSuccess) level_ms[k] = ms;
        }
    }
    for (auto &e : N.lev_ev) if (e) (void)hipEventDestroy(e);
    if (iterations) *iterations = rc == OCTANE_OK ? *pl->h_iters : -1;
    if (peer_copy_bytes) *peer_copy_bytes = N.peer_bytes;
    if (boundaries) *boundaries = N.boundaries;
    if (banded_levels) *banded_levels = nbanded;
    if (band_rows_finest) *band_rows_finest = rows.back().empty() ? pl->lev.back().h : rows.back()[band].y1 - rows.back()[band].y0;
    cleanup();
    return rc;
}
#endif

// =====================================================================================================================
// One band per PROCESS (the one-process-per-GPU launch): octane_vof_mp_*.
//
// Same level loop, same kernels, same halo / partial protocol as above; what differs is how a band reaches the others.
// Every rank creates its band (a full-size plan + a partial block), exports the two allocations as HIP IPC handles, the
// host program all-gathers the handles (torch.distributed, MPI, a file -- 128 bytes per rank) and every rank maps the
// other ranks' allocations.  A phase boundary is: drain my stream (hipStreamSynchronize), then meet the other ranks at a
// barrier that lives in a POSIX shared-memory object.  That costs a host round trip per boundary (two per PCG iteration)
// where the in-process form costs an event wait, and in exchange needs nothing but shared memory between the ranks: no
// collective library on the data path (RCCL stays what the launcher uses for rendezvous and timing).
// The barrier gives up after 120 s (OCTANE_MP_TIMEOUT_S) so that a rank that died cannot leave the others spinning: the
// rank that times out marks the group DEAD in the shared object; from then on no rank of the group waits for or issues
// anything more, octane_vof_mp_run returns an error on every rank still alive, and the group cannot be used again (the
// launcher is expected to exit non-zero and start fresh ranks).  A rank that merely FAILED (a HIP error, bad inputs) is
// different: it keeps walking the protocol so that the others see the flag and return at once, and the group stays usable.
// =====================================================================================================================
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct MpShared {                 // lives in the shared-memory object; zero-filled at creation by rank 0
    SpinBarrier bar;
    std::atomic<int> failed;      // first failing rank + 1 (cleared by rank 0 for the next call)
    std::atomic<int> ready;       // rank 0 sets it once the object is initialised
    std::atomic<int> dead;        // a rank timed out at a phase boundary: the barrier state is undefined, nobody synchronises again
    std::atomic<unsigned long long> nonce;   // of rank 0's IPC handle: tells this run's object from one a crashed run left behind
    std::atomic<unsigned long long> devid[kMaxBands];   // a hash of each rank's PCI bus id: ranks sharing a GPU share its CUs
    std::atomic<int> ipc[kMaxBands];                    // 1: the rank opened every IPC mapping, 2: it could not; +4: it has a collective library
};

struct octane_vof_mp {
    int rank = 0, world = 1, device = 0;
    int nx = 0, ny = 0, nc = 0;
    octane_vof_params prm;
    octane_vof_plan *pl = nullptr;
    double *parts = nullptr;
    double *mirror = nullptr;      // local copies of every rank's partial blocks (copy transport)
    char *arena[kMaxBands] = {nullptr};
    double *parts_all[kMaxBands] = {nullptr};
    bool connected = false;
    bool dead = false;             // a phase boundary timed out: every further call fails at once
    double timeout_s = 120.;
    std::vector<std::vector<BandRows>> rows;
    std::string shm_name;
    MpShared *shm = nullptr;
    int last_cur = 0;
    long long copies = 0;
    bool aborted = false;          // a persistent mid-level solve of the last run gave up (on any rank): its flow is not valid
    int transport = OCTANE_TRANSPORT_INPLACE;
    bool no_dma = false;
    bool ipc_ok = true;            // every rank opened every IPC mapping (agreed among the ranks in octane_vof_mp_connect)
    octane_vof_exchange ex;        // the host program's collective library (all zero: none registered)
    long long min_band_pixels = 0;
    octane_vof_transport_info info;
};

struct MpHandles { hipIpcMemHandle_t arena, parts; };
static_assert(sizeof(MpHandles) == OCTANE_MP_HANDLE_BYTES, "OCTANE_MP_HANDLE_BYTES must match two hipIpcMemHandle_t");

struct ProcNet : BandNet {
    octane_vof_mp *m;
    std::string error;
    int rc = OCTANE_OK;
    explicit ProcNet(octane_vof_mp *m_) : m(m_)
    {
        nb = m->world; prm = m->prm; rows = &m->rows;
        for (int b = 0; b < nb; b++) { arena[b] = m->arena[b]; parts[b] = m->parts_all[b]; }
        transport = m->transport; no_dma = m->no_dma; brk = test_break_bits();
    }
    double *mirror(int) override { return m->mirror; }
    int xchg_all_gather(int, const void *send, void *const *recv, size_t bytes) override
    {
        return m->ex.all_gather ? m->ex.all_gather(m->ex.user, send, recv, bytes) : -1;
    }
    int xchg_sendrecv(int, int n, const octane_vof_xfer *ops) override
    {
        return m->ex.sendrecv ? m->ex.sendrecv(m->ex.user, n, ops) : -1;
    }
    hipError_t pull(int, void *dst_local, int, const void *src, size_t bytes) override
    {
        return hipMemcpyAsync(dst_local, src, bytes, hipMemcpyDefault, m->pl->own_stream);     // src is an IPC mapping
    }
    octane_vof_plan *plan(int) override { return m->pl; }
    bool dead() const { return m->shm->dead.load(std::memory_order_acquire) != 0; }
    bool failed() override { return m->shm->failed.load(std::memory_order_relaxed) != 0 || dead(); }
    void fail(int b, int code, const std::string &msg) override
    {
        if (rc == OCTANE_OK) { rc = code; error = msg; }
        int expect = 0;
        m->shm->failed.compare_exchange_strong(expect, b + 1);
    }
    void sync(int b) override
    {
        BandNet &N = *this;
        if (dead()) return;                         // the walk goes on (nothing is issued: failed()), but nobody waits any more
        BAND_HIP(hipStreamSynchronize(m->pl->own_stream));
        if (!m->shm->bar.wait(nb, m->timeout_s, &m->shm->dead)) {
            const bool first = !dead();
            m->shm->dead.store(1, std::memory_order_release);
            fail(b, OCTANE_E_HIP, first ? "timed out waiting for the other ranks at a phase boundary"
                                        : "another rank timed out at a phase boundary");
        }
    }
    hipError_t copy(int, void *dst, int, const void *src, size_t bytes) override
    {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, m->pl->own_stream);     // dst is an IPC mapping
    }
};

extern "C" int octane_vof_mp_destroy(octane_vof_mp *m)
{
    if (!m) return OCTANE_OK;
    (void)hipSetDevice(m->device);
    if (m->pl && m->pl->own_stream) (void)hipStreamSynchronize(m->pl->own_stream);
    for (int b = 0; b < m->world && b < kMaxBands; b++) {
        if (b == m->rank) continue;
        if (m->arena[b]) (void)hipIpcCloseMemHandle(m->arena[b]);
        if (m->parts_all[b]) (void)hipIpcCloseMemHandle(m->parts_all[b]);
    }
    if (m->shm) {
        // nobody unmaps what another rank may still read (a dead group has stopped reading)
        if (m->connected && !m->dead && m->shm->dead.load() == 0) (void)m->shm->bar.wait(m->world, 20., &m->shm->dead);
        munmap(m->shm, sizeof(MpShared));
        if (m->rank == 0) shm_unlink(m->shm_name.c_str());
    }
    if (m->parts) (void)hipFree(m->parts);
    if (m->mirror) (void)hipFree(m->mirror);
    if (m->pl) octane_vof_plan_destroy(m->pl);
    delete m;
    return OCTANE_OK;
}

static int mp_create_impl(octane_vof_mp **out, int nx, int ny, int nchan, const octane_vof_params *p, int rank, int world,
                          long long min_band_pixels, const char *shm_name, bool check_group);

extern "C" int octane_vof_mp_create(octane_vof_mp **out, int nx, int ny, int nchan, const octane_vof_params *p, int rank, int world,
                                    long long min_band_pixels, const char *shm_name)
{
    return mp_create_impl(out, nx, ny, nchan, p, rank, world, min_band_pixels, shm_name, false);
}

static int mp_create_impl(octane_vof_mp **out, int nx, int ny, int nchan, const octane_vof_params *p, int rank, int world,
                          long long min_band_pixels, const char *shm_name, bool check_group)
{
    if (!out || !p || world < 1 || world > kMaxBands || rank < 0 || rank >= world || !shm_name || shm_name[0] != '/') {
        set_last_error("octane_vof_mp_create: invalid argument (1 <= world <= 8, shm_name like /octane_1234)");
        return OCTANE_E_INVALID;
    }
    *out = nullptr;
    octane_vof_mp *m = new octane_vof_mp();
    m->rank = rank; m->world = world; m->nx = nx; m->ny = ny; m->nc = nchan; m->prm = *p; m->shm_name = shm_name;
    m->min_band_pixels = min_band_pixels;
    std::memset(&m->ex, 0, sizeof m->ex);
    info_reset(m->info, world);
    int rc = plan_create_ex(&m->pl, nx, ny, nchan, p, 1, true);
    if (rc != OCTANE_OK) { delete m; return rc; }
    m->device = m->pl->device;
    if (hipMalloc((void **)&m->parts, (size_t)2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipMemset(m->parts, 0, (size_t)2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&m->mirror, (size_t)world * 2 * kPartBlock * sizeof(double)) != hipSuccess ||
        hipMemset(m->mirror, 0, (size_t)world * 2 * kPartBlock * sizeof(double)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        set_last_error("octane_vof_mp_create: device allocation failed");
        octane_vof_mp_destroy(m);
        return OCTANE_E_NOMEM;
    }
    m->arena[rank] = reinterpret_cast<char *>(m->pl->shared_base());      // what the other ranks map: the band's SHARED planes only (vof_plan.hpp)
    m->parts_all[rank] = m->parts;
    long minpix = min_band_pixels > 0 ? (long)min_band_pixels : kDefaultMinBandPixels;
    if (!check_group) if (const char *e = getenv("OCTANE_TUNE_MIN_BAND_PIXELS")) minpix = atol(e);     // never for the self-check's frame
    m->rows.resize(m->pl->lev.size());
    for (size_t k = 0; k < m->pl->lev.size(); k++) {
        const LevelInfo &li = m->pl->lev[k];
        if ((long)li.w * li.h < minpix) continue;
        int edges[kMaxBands + 1];
        if (octane_vof_band_partition(li.h, world, edges) != 1) continue;
        std::vector<BandRows> r(world);
        for (int b = 0; b < world; b++) { r[b].y0 = edges[b]; r[b].y1 = edges[b + 1]; }
        m->rows[k] = r;
    }
    if (const char *e = getenv("OCTANE_MP_TIMEOUT_S")) { const double v = atof(e); if (v > 0.) m->timeout_s = v; }
    // The shared-memory object.  Rank 0 creates and initialises it HERE (removing whatever an earlier, crashed run left
    // under the name); the other ranks open it in octane_vof_mp_connect, i.e. after the host program's all-gather of the
    // handles, which rank 0 only joins after this function has returned -- so they can only ever see this run's object --
    // and they check the nonce rank 0 derived from its own IPC handle against the all-gathered handle.
    if (rank == 0) {
        shm_unlink(shm_name);
        const int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(MpShared)) != 0) {
            if (fd >= 0) close(fd);
            set_last_error(std::string("octane_vof_mp_create: cannot create shared memory object ") + shm_name);
            octane_vof_mp_destroy(m);
            return OCTANE_E_INVALID;
        }
        void *mem = mmap(nullptr, sizeof(MpShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (mem == MAP_FAILED) {
            set_last_error("octane_vof_mp_create: mmap of the shared memory object failed");
            octane_vof_mp_destroy(m);
            return OCTANE_E_INVALID;
        }
        m->shm = static_cast<MpShared *>(mem);
        new (m->shm) MpShared();
        m->shm->failed.store(0);
        m->shm->dead.store(0);
        m->shm->nonce.store(0);
        for (int b = 0; b < kMaxBands; b++) { m->shm->devid[b].store(0); m->shm->ipc[b].store(0); }
    }
    *out = m;
    return OCTANE_OK;
}

extern "C" int octane_vof_mp_set_exchange(octane_vof_mp *m, const octane_vof_exchange *ex)
{
    if (!m || m->connected || (ex && (!ex->all_gather || !ex->sendrecv))) {
        set_last_error("octane_vof_mp_set_exchange: call it between octane_vof_mp_create and octane_vof_mp_connect, with both callbacks set");
        return OCTANE_E_INVALID;
    }
    if (ex) { m->ex = *ex; m->ex.name[sizeof m->ex.name - 1] = 0; }
    else std::memset(&m->ex, 0, sizeof m->ex);
    std::memcpy(m->info.exchange, m->ex.name, sizeof m->info.exchange);
    return OCTANE_OK;
}

static unsigned long long handle_nonce(const MpHandles &h)     // FNV-1a of rank 0's handles, never 0
{
    unsigned long long x = 1469598103934665603ull;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&h);
    for (size_t i = 0; i < sizeof h; i++) { x ^= p[i]; x *= 1099511628211ull; }
    return x ? x : 1;
}

extern "C" int octane_vof_mp_handles(octane_vof_mp *m, void *buf)
{
    if (!m || !buf) return OCTANE_E_INVALID;
    TILED_TRY(hipSetDevice(m->device));
    MpHandles h;
    std::memset(&h, 0, sizeof h);
    TILED_TRY(hipIpcGetMemHandle(&h.arena, m->pl->shared_base()));
    TILED_TRY(hipIpcGetMemHandle(&h.parts, m->parts));
    std::memcpy(buf, &h, sizeof h);
    if (m->rank == 0) {                          // the object is ready for the others once the nonce is in place
        m->shm->nonce.store(handle_nonce(h), std::memory_order_release);
        m->shm->ready.store(1, std::memory_order_release);
    }
    return OCTANE_OK;
}

extern "C" int octane_vof_mp_connect(octane_vof_mp *m, const void *all_handles)
{
    if (!m || !all_handles || m->connected) { set_last_error("octane_vof_mp_connect: invalid argument"); return OCTANE_E_INVALID; }
    TILED_TRY(hipSetDevice(m->device));
    const MpHandles *h = static_cast<const MpHandles *>(all_handles);
    if (m->rank != 0) {       // rank 0's object exists by now (see octane_vof_mp_create); a few retries cover slow filesystems only
        int fd = -1;
        for (int tries = 0; tries < 500 && fd < 0; tries++) {
            fd = shm_open(m->shm_name.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(MpShared)) { close(fd); fd = -1; }
            }
            if (fd < 0) usleep(10000);
        }
        if (fd < 0) { set_last_error("octane_vof_mp_connect: cannot open shared memory object " + m->shm_name); return OCTANE_E_INVALID; }
        void *mem = mmap(nullptr, sizeof(MpShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (mem == MAP_FAILED) { set_last_error("octane_vof_mp_connect: mmap of the shared memory object failed"); return OCTANE_E_INVALID; }
        m->shm = static_cast<MpShared *>(mem);
        if (m->shm->ready.load(std::memory_order_acquire) != 1 || m->shm->nonce.load(std::memory_order_acquire) != handle_nonce(h[0])) {
            set_last_error("octane_vof_mp_connect: shared memory object " + m->shm_name + " is not this run's (stale object of a crashed run?)");
            munmap(m->shm, sizeof(MpShared)); m->shm = nullptr;
            return OCTANE_E_INVALID;
        }
    }
    // The other ranks' allocations.  A mapping that cannot be opened (no IPC between these two processes / devices) is not fatal when
    // the host program registered a collective library: the ranks then agree below on the collective transport, which needs none.
    bool ipc_ok = true;
    std::string ipc_err;
    const int forced = forced_transport();
    if (forced == -2) { set_last_error("octane_vof_mp_connect: OCTANE_TILED_TRANSPORT must be inplace, copy or collective"); return OCTANE_E_INVALID; }
    const bool skip_ipc = forced == OCTANE_TRANSPORT_COLLECTIVE
#ifdef OCTANE_DIAG
                          || (test_break_bits() & 8)         // drill: "IPC unavailable"
#endif
                          ;
    for (int b = 0; b < m->world && ipc_ok; b++) {
        if (b == m->rank) continue;
        void *pa = nullptr, *pp = nullptr;
        hipError_t e = skip_ipc ? hipErrorNotSupported : hipIpcOpenMemHandle(&pa, h[b].arena, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&pp, h[b].parts, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            ipc_ok = false;
            ipc_err = std::string("hipIpcOpenMemHandle of rank ") + std::to_string(b) + "'s memory: " + hipGetErrorString(e);
            if (pa) (void)hipIpcCloseMemHandle(pa);
            (void)hipGetLastError();
            break;
        }
        m->arena[b] = static_cast<char *>(pa);
        m->parts_all[b] = static_cast<double *>(pp);
    }
    m->shm->ipc[m->rank].store((ipc_ok ? 1 : 2) + (m->ex.all_gather ? 4 : 0), std::memory_order_release);
    m->connected = true;
    {   // which GPU this rank drives, for the others to see
        char bus[64] = {0};
        (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, m->device);
        unsigned long long hsh = 1469598103934665603ull;
        for (const char *c = bus; *c; c++) { hsh ^= (unsigned char)*c; hsh *= 1099511628211ull; }
        m->shm->devid[m->rank].store(hsh ? hsh : 1, std::memory_order_release);
    }
    if (!m->shm->bar.wait(m->world, m->timeout_s, &m->shm->dead)) {
        m->shm->dead.store(1); m->dead = true;
        set_last_error("octane_vof_mp_connect: the other ranks did not arrive");
        return OCTANE_E_HIP;
    }
    {   // what every rank can do decides what all of them do
        bool all_ipc = true, all_ex = true;
        for (int b = 0; b < m->world; b++) {
            const int f = m->shm->ipc[b].load(std::memory_order_acquire);
            all_ipc = all_ipc && (f & 3) == 1; all_ex = all_ex && (f & 4) != 0;
        }
        m->ipc_ok = all_ipc;
        m->info.peer_ok = all_ipc ? 1 : 0;
        if (!all_ipc) {                         // close what was opened: nobody will read through it
            for (int b = 0; b < m->world; b++) {
                if (b == m->rank) continue;
                if (m->arena[b]) { (void)hipIpcCloseMemHandle(m->arena[b]); m->arena[b] = nullptr; }
                if (m->parts_all[b]) { (void)hipIpcCloseMemHandle(m->parts_all[b]); m->parts_all[b] = nullptr; }
            }
        }
        if (!all_ex) std::memset(&m->ex, 0, sizeof m->ex);          // registered on some ranks only: nobody uses it
        const bool have_ex = all_ex;
        if (forced >= 0) {
            if ((forced != OCTANE_TRANSPORT_COLLECTIVE && !all_ipc) || (forced == OCTANE_TRANSPORT_COLLECTIVE && !have_ex)) {
                set_last_error(std::string("octane_vof_mp_connect: OCTANE_TILED_TRANSPORT=") + octane_vof_transport_name(forced) + " is not available: " +
                               (forced == OCTANE_TRANSPORT_COLLECTIVE ? "no collective library registered on every rank (octane_vof_mp_set_exchange)" : ipc_err));
                return OCTANE_E_INVALID;
            }
            m->transport = forced; m->info.forced = 1;
        } else if (!all_ipc) {
            if (!have_ex) {
                set_last_error("octane_vof_mp_connect: " + (ipc_err.empty() ? std::string("another rank could not open the IPC mappings") : ipc_err) +
                               ", and no collective library is registered (octane_vof_mp_set_exchange) to carry the bands' exchange instead");
                return OCTANE_E_HIP;
            }
            m->transport = OCTANE_TRANSPORT_COLLECTIVE;
            if (m->rank == 0) fprintf(stderr, "octane: row bands, one per process: HIP IPC mappings are not available (%s) -- using the collective transport (%s)\n",
                                      ipc_err.empty() ? "on another rank" : ipc_err.c_str(), m->ex.name);
        }
        m->info.transport = m->transport;
        std::memcpy(m->info.exchange, m->ex.name, sizeof m->info.exchange);
    }
    {   // Ranks that share a GPU (a rehearsal on a one-GPU box) share its CUs: their persistent PCG solves of the replicated
        // levels run at the same time and cannot be serialised across processes, so each may only hold its share of the CUs --
        // otherwise two of them could wait for each other's workgroups to become resident (pcg_persist.hip).
        int sharing = 0;
        const unsigned long long mine = m->shm->devid[m->rank].load(std::memory_order_acquire);
        for (int b = 0; b < m->world; b++) sharing += (m->shm->devid[b].load(std::memory_order_acquire) == mine);
        int distinct = 0;
        for (int b = 0; b < m->world; b++) {
            bool dup = false;
            for (int c = 0; c < b; c++) dup = dup || m->shm->devid[c].load(std::memory_order_acquire) == m->shm->devid[b].load(std::memory_order_acquire);
            distinct += !dup;
        }
        m->info.ndevices = distinct;
        if (sharing > 1) {
            const int cap = m->pl->ncu / sharing;
            if (cap < m->pl->persist_max_g) m->pl->persist_max_g = cap;
        }
    }
    return OCTANE_OK;
}

extern "C" int octane_vof_mp_transport_info(const octane_vof_mp *m, octane_vof_transport_info *out)
{
    if (!m || !out) return OCTANE_E_INVALID;
    *out = m->info;
    out->transport = m->transport; out->q_dma = m->no_dma ? 0 : 1; out->nbands = m->world;
    return OCTANE_OK;
}

// The first-contact self-check of the process form (include/octane_vof.h): the ranks build a second, small group with the host
// program's byte all-gather, solve the check frame under each candidate transport and keep the first one rank 0 finds within the bar
// of the plain plan's flow (every rank's own return code can veto).  Collective; the verdicts travel by the same all-gather.
extern "C" int octane_vof_mp_selfcheck(octane_vof_mp *m, octane_allgather_bytes_fn ag, void *user)
{
    if (!m || !m->connected || !ag) { set_last_error("octane_vof_mp_selfcheck: call it after octane_vof_mp_connect, with an all-gather"); return OCTANE_E_INVALID; }
    bool any_banded = false, large = false;
    for (size_t k = 0; k < m->rows.size(); k++) {
        if (m->rows[k].empty()) continue;
        any_banded = true;
        int maxrows = 0;
        for (auto &r : m->rows[k]) maxrows = std::max(maxrows, r.y1 - r.y0);
        if (m->pl->use_fused && pcg_fused_q_form(m->pl->lev[k].w, maxrows, m->pl->lev[k].h)) large = true;
    }
    if (m->world < 2 || !any_banded || m->info.forced || !selfcheck_enabled()) return OCTANE_OK;
    TILED_TRY(hipSetDevice(m->device));
    const int nb = m->world;
    int w, h;
    check_geometry(nb, large, &w, &h);
    const octane_vof_params cp = check_params(m->prm, m->device);
    std::vector<Candidate> cand;
    if (m->ipc_ok) {
        cand.push_back({OCTANE_TRANSPORT_INPLACE, false});
        if (large) cand.push_back({OCTANE_TRANSPORT_INPLACE, true});
        cand.push_back({OCTANE_TRANSPORT_COPY, false});
    }
    if (m->ex.all_gather) cand.push_back({OCTANE_TRANSPORT_COLLECTIVE, false});
    // the check group: same ranks, same devices, same collective library, a shared-memory object of its own
    octane_vof_mp *c = nullptr;
    int rc = mp_create_impl(&c, w, h, 1, &cp, m->rank, nb, 1, (m->shm_name + "_chk").c_str(), true);
    std::vector<unsigned char> all((size_t)nb * OCTANE_MP_HANDLE_BYTES);
    unsigned char mine[OCTANE_MP_HANDLE_BYTES] = {0};
    if (rc == OCTANE_OK && m->ex.all_gather) rc = octane_vof_mp_set_exchange(c, &m->ex);
    if (rc == OCTANE_OK) rc = octane_vof_mp_handles(c, mine);
    // every rank joins every all-gather whatever happened to it before: a rank that failed says so in its verdict below
    const int ag_rc = ag(user, mine, all.data(), OCTANE_MP_HANDLE_BYTES);
    if (rc == OCTANE_OK && ag_rc != 0) { set_last_error("octane_vof_mp_selfcheck: the host program's all-gather failed"); rc = OCTANE_E_HIP; }
    if (rc == OCTANE_OK) rc = octane_vof_mp_connect(c, all.data());
    std::vector<float> a, b2, up, vp, u, v;
    long long its_plain = 0;
    if (rc == OCTANE_OK) {
        check_scene(w, h, a, b2);
        if (m->rank == 0) rc = plain_check_flow(cp, w, h, a, b2, up, vp, &its_plain);
    }
    std::vector<int> verdicts(nb);
    // one word per rank: bit 0 = the check group stands, bit 1 = its IPC mappings are open, bit 2 = its frame has the banded levels it
    // is meant to have.  The in-place and copy candidates dereference the other ranks' memory: they run only where EVERY rank of the
    // check group has the mappings (the real group's ipc_ok says nothing about a second set of mappings -- ADVICE r4).
    const bool banded_ok = rc == OCTANE_OK && c && octane_vof_mp_banded_levels(c) == kCheckBandedLevels;
    int word = (rc == OCTANE_OK ? 1 : 0) | ((rc == OCTANE_OK && c && c->ipc_ok) ? 2 : 0) | (banded_ok ? 4 : 0);
    if (ag(user, &word, verdicts.data(), sizeof(int)) != 0) { word = 0; for (int r = 0; r < nb; r++) verdicts[r] = 0; }
    int setup_ok = 1, ipc_all = 1, banded_all = 1;
    for (int r = 0; r < nb; r++) { setup_ok = setup_ok && (verdicts[r] & 1); ipc_all = ipc_all && (verdicts[r] & 2); banded_all = banded_all && (verdicts[r] & 4); }
    int chosen = -1;
    std::string log;
    if (setup_ok && !banded_all) {
        if (m->rank == 0)
            fprintf(stderr, "octane: row-band self-check NOT RUN: the check frame (%d x %d, %d ranks) does not have %d banded levels on every rank\n", w, h, nb, kCheckBandedLevels);
        if (c) octane_vof_mp_destroy(c);
        TILED_TRY(hipSetDevice(m->device));
        m->info.selfcheck = 0;
        return OCTANE_OK;
    }
    if (setup_ok) {
        for (size_t i = 0; i < cand.size() && chosen < 0; i++) {
            if (cand[i].transport != OCTANE_TRANSPORT_COLLECTIVE && !ipc_all) {      // every rank skips the same candidates
                if (m->rank == 0)
                    log += std::string(log.empty() ? "" : "; ") + octane_vof_transport_name(cand[i].transport) + ": skipped (the check group has no IPC mappings on every rank)";
                if (i < 4) m->info.check_rel_l2[i] = -1.;
                m->info.candidates_tried = (int)i + 1;
                continue;
            }
            c->transport = cand[i].transport; c->no_dma = cand[i].no_dma;
            u.assign((size_t)w * h, 0.f); v.assign((size_t)w * h, 0.f);
            const int rrc = octane_vof_mp_run(c, a.data(), b2.data(), nullptr, nullptr, u.data(), v.data(), OCTANE_MEM_HOST);
            int pass = rrc == OCTANE_OK ? 1 : 0;
            double d = -1.;
            if (m->rank == 0 && pass) {
                bool fin = false;
                d = flow_rel_l2(u, v, up, vp, &fin);
                const long long its = octane_vof_mp_last_iterations(c);
                pass = fin && d <= kCheckBar && its == its_plain;
                log += std::string(log.empty() ? "" : "; ") + octane_vof_transport_name(cand[i].transport) + (cand[i].no_dma ? " without LDS-DMA from the neighbour" : "") +
                       (pass ? ": ok" : ": rel L2 " + std::to_string(d) + ", iterations " + std::to_string(its) + " / " + std::to_string(its_plain));
            } else if (m->rank == 0) {
                log += std::string(log.empty() ? "" : "; ") + octane_vof_transport_name(cand[i].transport) + ": error (" + octane_last_error() + ")";
            }
            if (i < 4) m->info.check_rel_l2[i] = d;
            m->info.candidates_tried = (int)i + 1;
            if (ag(user, &pass, verdicts.data(), sizeof(int)) != 0) { pass = 0; for (int r = 0; r < nb; r++) verdicts[r] = 0; }
            bool all_pass = true;
            for (int r = 0; r < nb; r++) all_pass = all_pass && verdicts[r] != 0;
            if (c->dead || (c->shm && c->shm->dead.load() != 0)) break;        // the check group is gone: nothing more can be tried
            if (all_pass) chosen = (int)i;
        }
    }
    if (c) octane_vof_mp_destroy(c);
    TILED_TRY(hipSetDevice(m->device));
    if (!setup_ok) {
        m->info.selfcheck = -1;
        set_last_error("octane_vof_mp_selfcheck: the check group could not be set up on every rank" + (rc != OCTANE_OK ? std::string(": ") + octane_last_error() : std::string()));
        return OCTANE_E_HIP;
    }
    m->info.selfcheck = chosen < 0 ? -1 : chosen == 0 ? 1 : 2;
    if (chosen >= 0) { m->transport = cand[chosen].transport; m->no_dma = cand[chosen].no_dma; }
    m->info.transport = m->transport; m->info.q_dma = m->no_dma ? 0 : 1;
    if (chosen != 0 && m->rank == 0)
        fprintf(stderr, "octane: row-band self-check, %d ranks (%d x %d): %s -> %s\n", nb, w, h, log.c_str(),
                chosen < 0 ? "NO transport reproduces the plain plan" : "using the last of these");
    if (chosen < 0) { set_last_error("octane_vof_mp_selfcheck: the first-contact self-check failed under every transport" + (log.empty() ? std::string() : ": " + log)); return OCTANE_E_HIP; }
    return OCTANE_OK;
}

extern "C" int octane_vof_mp_banded_levels(const octane_vof_mp *m)
{
    if (!m) return -1;
    int n = 0;
    for (auto &r : m->rows) n += !r.empty();
    return n;
}

extern "C" long long octane_vof_mp_last_iterations(octane_vof_mp *m) { return !m ? -1 : m->aborted ? -2 : *m->pl->h_iters; }

// Every rank passes the whole pair (host buffers, or dense device buffers on its own device) and the first guess; the
// flow arrives in u / v on rank 0 only (other ranks' u / v are left alone).  Collective: every rank must call it.
extern "C" int octane_vof_mp_run(octane_vof_mp *m, const float *img1, const float *img2, const float *u0, const float *v0,
                                 float *u, float *v, int mem)
{
    if (!m || !m->connected || !img1 || !img2 || ((u0 == nullptr) != (v0 == nullptr)) || (m->rank == 0 && (!u || !v)) ||
        (mem != OCTANE_MEM_HOST && mem != OCTANE_MEM_DEVICE)) {
        set_last_error("octane_vof_mp_run: invalid argument (or octane_vof_mp_connect not called)");
        return OCTANE_E_INVALID;
    }
    if (m->dead || m->shm->dead.load(std::memory_order_acquire) != 0) {
        m->dead = true;
        set_last_error("octane_vof_mp_run: this group of ranks timed out at a phase boundary earlier and cannot be used again; start fresh ranks");
        return OCTANE_E_HIP;
    }
    TILED_TRY(hipSetDevice(m->device));
    octane_vof_plan *pl = m->pl;
    ProcNet N(m);
    {   // a rank that cannot load its inputs still walks the protocol, so that the others fail fast instead of timing out
        const int rc = plan_load_inputs(pl, img1, img2, u0, v0, mem, pl->own_stream);
        if (rc) N.fail(m->rank, rc, "loading the inputs failed");
    }
    N.sync(m->rank);                      // nobody's halo rows are written before everybody has its inputs in place
    m->aborted = false;
    band_worker(N, m->rank);
    // a persistent solve of a replicated level that gave up on this rank invalidates the frame for every rank: raise the group's
    // failure flag before the closing boundary, so that all ranks return the error together
    if (hipStreamSynchronize(pl->own_stream) == hipSuccess && persist_check(pl) != OCTANE_OK) {
        m->aborted = true;
        N.fail(m->rank, OCTANE_E_HIP, "a persistent PCG solve was abandoned (GPU shared with another process?): the flow of this run is not valid; "
                                      "OCTANE_TUNE_PERSIST=0 selects one launch per iteration");
    }
    N.sync(m->rank);
    m->copies = N.copies[m->rank];
    m->last_cur = N.cur[m->rank];
    if (N.failed()) {
        m->aborted = true;                        // whatever failed: this run's flow and iteration count are not valid
        set_last_error("octane_vof_mp_run (rank " + std::to_string(m->rank) + "): " + (N.error.empty() ? std::string("another rank failed") : N.error));
        if (N.dead()) {                           // a rank is gone: no more synchronisation of any kind, the group is finished
            m->dead = true;
            (void)hipStreamSynchronize(pl->own_stream);
            return N.rc != OCTANE_OK ? N.rc : OCTANE_E_HIP;
        }
        m->shm->bar.wait(m->world, 20., &m->shm->dead);   // everybody has seen the flag before rank 0 clears it for the next call
        if (m->rank == 0) m->shm->failed.store(0);
        return N.rc != OCTANE_OK ? N.rc : OCTANE_E_HIP;
    }
    if (m->rank == 0) {
        const size_t dense_row = (size_t)m->nx * sizeof(float), pitched_row = (size_t)pl->pitch0 * sizeof(float);
        const int cur = m->last_cur;
        if (mem == OCTANE_MEM_HOST) {
            TILED_TRY(hipMemcpy2DAsync(u, dense_row, pl->U[cur], pitched_row, dense_row, m->ny, hipMemcpyDeviceToHost, pl->own_stream));
            TILED_TRY(hipMemcpy2DAsync(v, dense_row, pl->V[cur], pitched_row, dense_row, m->ny, hipMemcpyDeviceToHost, pl->own_stream));
        } else {
            launch_copy2d(pl->own_stream, pl->U[cur], pl->pitch0, u, m->nx, m->nx, m->ny);
            launch_copy2d(pl->own_stream, pl->V[cur], pl->pitch0, v, m->nx, m->nx, m->ny);
        }
    }
    TILED_TRY(hipStreamSynchronize(pl->own_stream));
    TILED_TRY(hipGetLastError());
    return OCTANE_OK;
}
