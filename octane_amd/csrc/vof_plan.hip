// vof_plan.hip -- host driver and C-ABI (include/octane_vof.h) of the variational flow core.
//
// Control flow restated from ref src/oct_variational_optical_flow.cu:487-1210 (level loop, three
// GNC steps, liters re-linearisations, cgiters PCG iterations) and :1213-1473 (host wrapper).
// The reference does all of this inside one cooperative launch with ~40 grid barriers per PCG
// iteration; here each phase is a stream-ordered launch and the host never synchronises inside
// a pyramid: the PCG stop test, alpha and beta live on the device (pcg_kernels.hip).
#include <hip/hip_runtime.h>

#include <atomic>
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

static thread_local std::string g_last_error;
void octane::set_last_error(const std::string &msg) { g_last_error = msg; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return OCTANE_E_HIP;                                                               \
        }                                                                                      \
    } while (0)

extern "C" const char *octane_last_error(void) { return g_last_error.c_str(); }

extern "C" int octane_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void octane_vof_default_params(octane_vof_params *p)
{
    if (!p) return;
    p->alpha = 5.; p->lambda = 1.; p->lambdac = 0.; p->scaleF = 0.5; p->scsig = 400.;
    p->kiters = 4; p->liters = 3; p->cgiters = 30; p->dozim = 1; p->device = 0;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Gaussian taps exactly as the reference's single thread computes them (ref .cu:207-228);
// 2*fs+1 values, normalised over all of them although only the first 2*fs are ever applied.
static void gauss_taps(float factor, int fs, float *gk)
{
    float sigma = (float)(0.6 * std::sqrt(1.0 / (double)(factor * factor) - 1.0));
    float s = (float)(2.0 * (double)sigma * (double)sigma);
    float sum = 0.0f;
    for (int x = -fs; x <= fs; x++) {
        float r = (float)x;
        gk[x + fs] = (float)((double)expf(-(r * r) / s) / (3.14159265358979323846 * (double)s));
        sum += gk[x + fs];
    }
    for (int i = 0; i < 2 * fs + 1; ++i) gk[i] /= sum;
}

extern "C" int octane_vof_plan_destroy(octane_vof_plan *pl)
{
    if (!pl) return OCTANE_OK;
    (void)hipSetDevice(pl->device);
    if (pl->own_stream) (void)hipStreamSynchronize(pl->own_stream);
    if (pl->side_stream) { (void)hipStreamSynchronize(pl->side_stream); (void)hipStreamDestroy(pl->side_stream); }
    if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
    for (int i = 0; i < 2; i++) { if (pl->ev_img[i]) (void)hipEventDestroy(pl->ev_img[i]); if (pl->ev_solved[i]) (void)hipEventDestroy(pl->ev_solved[i]); }
    for (auto &e : pl->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (pl->ev_t0) (void)hipEventDestroy(pl->ev_t0);
    if (pl->ev_t1) (void)hipEventDestroy(pl->ev_t1);
    if (pl->ev_s0) (void)hipEventDestroy(pl->ev_s0);
    if (pl->ev_s1) (void)hipEventDestroy(pl->ev_s1);
    if (pl->graph_exec) (void)hipGraphExecDestroy(pl->graph_exec);
    if (pl->arena) (void)hipFree(pl->arena);
    if (pl->xarena) (void)hipFree(pl->xarena);
    if (pl->d_taps) (void)hipFree(pl->d_taps);
    if (pl->d_parts) (void)hipFree(pl->d_parts);
    if (pl->d_state) (void)hipFree(pl->d_state);
    if (pl->d_alpha) (void)hipFree(pl->d_alpha);
    if (pl->d_iters) (void)hipFree(pl->d_iters);
    if (pl->h_iters) (void)hipHostFree(pl->h_iters);
    if (pl->d_mid) (void)hipFree(pl->d_mid);
    if (pl->h_mid_abort) (void)hipHostFree(pl->h_mid_abort);
    if (pl->own_stream) (void)hipStreamDestroy(pl->own_stream);
    delete pl;
    return OCTANE_OK;
}

static void fill_level_ptrs(octane_vof_plan *pl, const LevelInfo &li, int cur, const LevelCtx &c, LevelPtrs &L)
{
    L.w = li.w; L.h = li.h; L.pitch = li.pitch; L.nc = pl->nc; L.cstride = pl->plane0;
    L.img1 = c.lev1; L.img2 = c.lev2;
    L.gx1 = c.gx1; L.gy1 = c.gy1; L.gx2 = c.gx2; L.gy2 = c.gy2;
    L.gxx = c.gxx; L.gxy = c.gxy; L.gyy = c.gyy;
    L.u = pl->U[cur]; L.v = pl->V[cur]; L.ut = c.ut; L.vt = c.vt;
    L.a1 = pl->a1; L.a2 = pl->a2; L.a4 = pl->a4; L.wx = pl->wx; L.wy = pl->wy; L.mu = pl->mu; L.mv = pl->mv;
    L.ru = pl->ru; L.rv = pl->rv;
    L.pu[0] = pl->pu[0]; L.pu[1] = pl->pu[1]; L.pv[0] = pl->pv[0]; L.pv[1] = pl->pv[1];
    L.qu = pl->qu; L.qv = pl->qv; L.xu = pl->xu; L.xv = pl->xv;
    L.part_rz = pl->d_parts; L.part_rr = pl->d_parts + kMaxParts; L.part_pq = pl->d_parts + 2 * kMaxParts;
    L.part_own = pl->d_parts;
    for (int b = 0; b < kMaxBands; b++) L.band_parts[b] = pl->d_parts;
    L.ru_up = L.ru_dn = pl->ru; L.rv_up = L.rv_dn = pl->rv;
    L.rb_u[0] = pl->ru; L.rb_v[0] = pl->rv; L.rb_u[1] = pl->ru2; L.rb_v[1] = pl->rv2;
    L.qb_u[0] = pl->qu; L.qb_v[0] = pl->qv; L.qb_u[1] = pl->qu2; L.qb_v[1] = pl->qv2;
    for (int i = 0; i < 2; i++) { L.qup_u[i] = L.qdn_u[i] = L.qb_u[i]; L.qup_v[i] = L.qdn_v[i] = L.qb_v[i]; }
    L.pf_u[0] = pl->pu[0]; L.pf_u[1] = pl->pu[1]; L.pf_u[2] = pl->pu3;
    L.pf_v[0] = pl->pv[0]; L.pf_v[1] = pl->pv[1]; L.pf_v[2] = pl->pv3;
    for (int i = 0; i < 2; i++) { L.rup_u[i] = L.rdn_u[i] = L.rb_u[i]; L.rup_v[i] = L.rdn_v[i] = L.rb_v[i]; }
    for (int i = 0; i < 3; i++) { L.pup_u[i] = L.pdn_u[i] = L.pf_u[i]; L.pup_v[i] = L.pdn_v[i] = L.pf_v[i]; }
    L.wy_up = pl->wy;
    L.q_form = 0;
    L.y0 = 0; L.y1 = li.h; L.ya0 = 0; L.ya1 = li.h; L.nbands = 1;
    L.st = pl->d_state; L.iter_total = pl->d_iters; L.alpha = pl->d_alpha; L.defer_x = pl->defer_x;
    L.reverse_b = pl->reverse_b;
    L.xcd_bands = pl->xcd_bands;
    L.nt_hints = pl->nt_hints;
    L.unit_w = 0;
    L.lean = 0;
    L.row_rot = 0;
    L.no_dma = 0;
}

void octane::plan_fill_level_ptrs(octane_vof_plan *pl, int k, int cur, const LevelCtx &c, LevelPtrs &L)
{
    fill_level_ptrs(pl, pl->lev[k], cur, c, L);
}

// Times a few PCG iterations of the finest level on whatever the arena holds (the values do not matter, only
// the addresses).  Returns milliseconds per iteration, or -1 on failure.
static double probe_level(octane_vof_plan *pl, int level, int reps, double *a_ms, double *b_ms);

static double probe_placement(octane_vof_plan *pl)
{
    return probe_level(pl, (int)pl->lev.size() - 1, 5, nullptr, nullptr);
}

// Times `reps` PCG iterations (pass A + pass B) of one pyramid level on whatever the planes hold; the stop test is
// kept open by rewriting the partials before every pass A.  Returns ms per iteration (first iteration excluded);
// with a_ms/b_ms non-null the two passes are also timed separately (one event pair per launch).
static double probe_level(octane_vof_plan *pl, int level, int reps, double *a_ms, double *b_ms)
{
    const LevelInfo &li = pl->lev[level];
    LevelPtrs L;
    const LevelCtx pc = {pl->img1p, pl->img2p, pl->uh, pl->vh, pl->gx1, pl->gy1, pl->gx2, pl->gy2, pl->gxx, pl->gxy, pl->gyy};
    fill_level_ptrs(pl, li, 0, pc, L);
    const int g_a = pcg_grid_size(li.w, li.h), g_b = pcg_b_grid_size(li.w, li.h);
    L.q_form = pcg_fused_q_form(li.w, li.h, li.h);
    const int g_f = pcg_fused_grid_size(li.w, li.h, 0, L.q_form);
    L.row_rot = L.q_form ? pcg_row_rotation(li.w, li.h, g_f, pl->xcd_bands) : 0;
    const bool fused = pl->use_fused && !(a_ms && b_ms);      // the split timing is defined for the two-pass form only
    // sums that keep the stop test open and every scalar finite whatever the planes hold: with all seven sums equal
    // to 1 the fused kernel's recurrences give alpha = 1, r.z = 0, r.r = 0 + ... -> use values that stay positive
    std::vector<double> ones(2 * (size_t)kPartBlock, 0.0);
    for (int half = 0; half < 2; half++) {
        double *blk = ones.data() + (size_t)half * kPartBlock;
        blk[kPartRz] = 4.0; blk[kPartRr] = 4.0; blk[kPartPq] = 1.0; blk[kPartQz] = 1.0; blk[kPartQmq] = 1.0; blk[kPartRq] = 1.0; blk[kPartQq] = 1.0;
    }
    if (!fused) for (size_t i = 0; i < 2 * (size_t)kMaxParts; i++) ones[i] = 1.0;
    PcgState st[2];
    st[0].rz = 1.f; st[0].stopped = 0; st[0].iters = 0; st[0].pad = 0;
    st[1] = st[0];
    hipStream_t s = pl->own_stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.;
    double ms_out = -1.;
    bool ok = true;
    std::vector<hipEvent_t> ev, fev;
    const bool split = a_ms && b_ms;
    if (split) {
        ev.resize(3 * (size_t)reps);
        for (auto &e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    }
    for (int it = 1; it <= reps && ok; it++) {     // it = 1 is a warm-up
        if (it == 2) ok = hipEventRecord(e0, s) == hipSuccess;
        // keep the stop test open whatever the previous pass wrote
        ok = ok && hipMemcpyAsync(pl->d_parts, ones.data(), (fused ? 2 * (size_t)kPartBlock : 2 * (size_t)kMaxParts) * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(pl->d_state, st, sizeof(st), hipMemcpyHostToDevice, s) == hipSuccess;
        if (fused) {       // the kernel the solve actually runs: one block of partials -> rz = 4 - 2 + 1 > 0, rr likewise
            // timed launch by launch (an event pair each): the two small copies above are not part of the figure
            if (fev.empty()) { fev.resize(2 * (size_t)reps); for (auto &e : fev) ok = ok && hipEventCreate(&e) == hipSuccess; }
            ok = ok && hipEventRecord(fev[2 * (it - 1)], s) == hipSuccess;
            launch_pcg_fused(s, L, it + 1, 1, g_f, 0.f);
            ok = ok && hipEventRecord(fev[2 * (it - 1) + 1], s) == hipSuccess;
            continue;
        }
#ifdef OCTANE_DIAG
        if (split) ok = ok && hipEventRecord(ev[3 * (it - 1)], s) == hipSuccess;
        launch_pcg_pass_a(s, L, it, g_b, g_a, 0.f);
        if (split) ok = ok && hipEventRecord(ev[3 * (it - 1) + 1], s) == hipSuccess;
        launch_pcg_pass_b(s, L, it, g_a, g_b);
        if (split) ok = ok && hipEventRecord(ev[3 * (it - 1) + 2], s) == hipSuccess;
#else
        (void)g_a; (void)g_b; ok = false;        // the two-pass form is not in the product library
#endif
    }
    ok = ok && hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess;
    if (ok) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ms_out = ms / (reps - 1);
        if (!fev.empty()) {        // the fused kernel's own time: mean over the launches after the warm-up
            double sum = 0.;
            for (int it = 2; it <= reps; it++) { float x = 0.f; (void)hipEventElapsedTime(&x, fev[2 * (it - 1)], fev[2 * (it - 1) + 1]); sum += x; }
            ms_out = sum / (reps - 1);
        }
        if (split) {
            double sa = 0., sb = 0.;
            for (int it = 2; it <= reps; it++) {
                float x = 0.f;
                (void)hipEventElapsedTime(&x, ev[3 * (it - 1)], ev[3 * (it - 1) + 1]); sa += x;
                (void)hipEventElapsedTime(&x, ev[3 * (it - 1) + 1], ev[3 * (it - 1) + 2]); sb += x;
            }
            *a_ms = sa / (reps - 1); *b_ms = sb / (reps - 1);
        }
    } else {
        (void)hipGetLastError();
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    for (auto &e : fev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms_out;
}

// Runs when the library is loaded.  The batch entry keeps several streams busy per device; the HIP runtime gives a
// process GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and two streams on one queue do not overlap.  Ask
// for 8, unless the user said something else; without effect if the runtime was initialised before this library came in.
__attribute__((constructor)) static void octane_runtime_defaults()
{
    (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

extern "C" int octane_vof_plan_create(octane_vof_plan **out, int nx, int ny, int nchan, const octane_vof_params *p)
{
    return octane::plan_create_ex(out, nx, ny, nchan, p, 8);
}

int octane::plan_create_ex(octane_vof_plan **out, int nx, int ny, int nchan, const octane_vof_params *p, int placement_trials, bool band_plan)
{
    if (!out || !p || nx < 2 || ny < 2 || nchan < 1 || nchan > kMaxChan || p->kiters < 1 || p->kiters > 24 ||
        p->liters < 0 || p->cgiters < 0 || !(p->alpha != 0.) || !(p->scaleF > 0. && p->scaleF <= 1.)) {
        g_last_error = "octane_vof_plan_create: invalid argument";
        return OCTANE_E_INVALID;
    }
    *out = nullptr;
    // the kernels address a plane with 32-bit byte offsets (one VGPR of offset shared by all planes): a plane has to stay below 4 GiB,
    // i.e. below 2^30 pixels incl. the row padding (32768 x 32752 would be the first frame too large; the reference's int indices
    // (.cu:613) give out at 2^31 / 12 CSR entries per pixel, 13377 x 13377, long before)
    if ((unsigned long long)round_up(nx, 64) * (unsigned long long)ny >= (1ull << 30)) {
        g_last_error = "octane_vof_plan_create: frame too large (a plane of round_up(nx, 64) x ny floats has to stay below 4 GiB)";
        return OCTANE_E_INVALID;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    int dev = p->device;
    if (dev > ndev - 1 || dev < 0) dev = 0;      // ref .cu:1260-1264
    HIP_TRY(hipSetDevice(dev));

    octane_vof_plan *pl = new octane_vof_plan();
    pl->prm = *p; pl->nx = nx; pl->ny = ny; pl->nc = nchan; pl->device = dev;
    pl->pitch0 = round_up(nx, 64);
    pl->plane0 = (size_t)pl->pitch0 * ny;
    pl->tol = (float)(0.0001 * 0.0001);          // ref .cu:1353
    if (const char *e = tune_env("OCTANE_TUNE_MAXBLOCKS")) set_max_blocks(atoi(e));   // developer tuning knobs
    if (const char *e = tune_env("OCTANE_TUNE_REVERSE_B")) pl->reverse_b = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_PASS_A")) set_pass_a_variant(atoi(e));
    if (const char *e = tune_env("OCTANE_TUNE_XCD")) pl->xcd_bands = atoi(e);
    if (const char *e = tune_env("OCTANE_TUNE_SMALL")) pl->use_small = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_SMALL_MAX")) pl->small_max_pixels = atol(e);
    if (const char *e = tune_env("OCTANE_TUNE_NT")) pl->nt_hints = atoi(e);
    if (const char *e = tune_env("OCTANE_TUNE_DEFER_X")) pl->defer_x = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_GRAPH")) pl->use_graph = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_UNIT_W")) pl->use_unit_w = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_FUSED")) pl->use_fused = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_FUSED_ROWS")) set_fused_rows(atoi(e));
    if (const char *e = tune_env("OCTANE_TUNE_FUSED_Q")) set_fused_q(atoi(e));
    if (const char *e = tune_env("OCTANE_TUNE_FUSED_Q_MIN")) set_fused_q_min(atol(e));
    if (const char *e = getenv("OCTANE_TUNE_Q_DMA")) set_q_dma(atoi(e));
    if (const char *e = getenv("OCTANE_TUNE_PERSIST")) pl->use_persist = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_PERSIST_STEP")) pl->persist_step = atoi(e);
    if (const char *e = tune_env("OCTANE_TUNE_PERSIST_P")) pl->persist_p = atoi(e);
    if (const char *e = tune_env("OCTANE_TUNE_PERSIST_MINP")) set_mid_min_p(atoi(e));
    if (const char *e = tune_env("OCTANE_TUNE_OVERLAP")) pl->use_overlap = atoi(e) != 0;
    if (const char *e = tune_env("OCTANE_TUNE_PERSIST_MAX")) pl->persist_max_pixels = atol(e);
    if (const char *e = getenv("OCTANE_TUNE_PERSIST_MAXG")) pl->persist_max_g = atoi(e);
    {
        hipDeviceProp_t prop;
        pl->ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    if (const char *e = tune_env("OCTANE_TUNE_UNIT_W_CAP")) set_unit_w_cap(atoi(e));
    {
        const char *ea = tune_env("OCTANE_TUNE_CAP_A"), *eb = tune_env("OCTANE_TUNE_CAP_B");
        if (ea || eb) set_pass_caps(ea ? atoi(ea) : 768, eb ? atoi(eb) : 1024);
    }
    pcg_small_configure();
    pcg_mid_configure();
#ifdef OCTANE_DIAG
    pcg_mid_configure_diag();
#endif
    set_grid_multiple(pl->xcd_bands == 1 ? 8 : 1);
    // the assembly's fast exact forms (three-instruction division by alpha etc.): each is used only if it reproduces the reference's
    // IEEE sequence on every float input for THIS alpha -- a ~10 ms device self-test, once per process and alpha (vof_kernels.hip)
    pl->asm_fast = assemble_fast_math_bits(p->alpha);
    memset(&pl->prof, 0, sizeof(pl->prof));

    const float scale = (float)p->scaleF;
    const float lambdaco = (float)(p->lambdac / p->alpha);   // ref .cu:1236
    std::vector<float> taps;
    for (int k = 0; k < p->kiters; k++) {
        LevelInfo li;
        li.factor = (float)std::pow((double)scale, (double)(p->kiters - k - 1));   // ref .cu:488
        li.w = (int)((double)nx * (double)li.factor + 0.5);                        // ref .cu:52-53
        li.h = (int)((double)ny * (double)li.factor + 0.5);
        li.pitch = round_up(li.w > 0 ? li.w : 1, 64);
        li.lambdac = (float)((double)lambdaco * std::pow(0.5, (double)k));         // ref .cu:494
        li.fs = 0; li.tap_off = 0;
        if (li.w < 2 || li.h < 2) {
            g_last_error = "a pyramid level would be smaller than 2 pixels; lower kiters";
            delete pl;
            return OCTANE_E_TOOSMALL;
        }
        if (k < p->kiters - 1) {
            float sigma = (float)(1.0 / std::sqrt(2. * (double)li.factor));        // ref .cu:521-526
            int fs = (int)(2 * sigma);
            if (fs < 5) fs = 5;
            li.fs = fs;
            li.tap_off = (int)taps.size();
            taps.resize(taps.size() + 2 * fs + 1);
            gauss_taps(li.factor, fs, taps.data() + li.tap_off);
        }
        pl->lev.push_back(li);
    }

    const int nc = nchan;
    // The second set of a level's flow-independent planes (9 nc + 2) only where the one-level-ahead overlap can run: a band
    // of a row-band solve sets its levels up through plan_level_setup (set 0 only), and a plan created with the overlap off never
    // forks the side stream.  They are the last planes of the arena, so every other plane keeps its offset.
    if (band_plan) pl->use_overlap = 0;
    pl->has_bset = pl->use_overlap ? 1 : 0;
    const size_t nplanes = (size_t)(2 * nc + 2) + 2 * nc + 7 * nc + 6 + 17 + 1 + 6 + (pl->has_bset ? (size_t)(9 * nc + 2) : 0);
    size_t skew = 0;                              // developer knob: stagger the planes' base addresses (floats)
    if (const char *e = tune_env("OCTANE_TUNE_SKEW")) skew = (size_t)atol(e) / 64 * 64;
    size_t align_f = 0;                           // developer knob: round the plane stride up to a multiple (bytes)
    if (const char *e = tune_env("OCTANE_TUNE_PLANE_ALIGN")) align_f = (size_t)atol(e) / 4;
    size_t stride = pl->plane0;
    if (align_f) stride = (stride + align_f - 1) / align_f * align_f;
    stride += skew;
    // band plans: the 30 planes from U[0] on (flow, operator, CG vectors: what neighbouring bands read) live in `xarena`, see vof_plan.hpp
    constexpr size_t kSharedPlanes = 6 + 17 + 1 + 6;
    constexpr size_t kGuardFloats = (size_t)1 << 20;      // 4 MB in front of the shared planes: an access just below U[0] stays inside the allocation
    const size_t head_planes = band_plan ? nplanes - kSharedPlanes : nplanes;
    pl->arena_bytes = head_planes * stride * sizeof(float) + (size_t)(4 << 20);
    hipError_t e = hipMalloc((void **)&pl->arena, pl->arena_bytes);
    if (e == hipSuccess && band_plan) {
        pl->xarena_bytes = (kGuardFloats + kSharedPlanes * stride) * sizeof(float) + (size_t)(4 << 20);
        e = hipMalloc((void **)&pl->xarena, pl->xarena_bytes);
    }
    if (e != hipSuccess) {
        g_last_error = std::string("hipMalloc of the plan arena failed: ") + hipGetErrorString(e);
        octane_vof_plan_destroy(pl);
        return OCTANE_E_NOMEM;
    }
    // Poison the arena with NaNs: a kernel that consumed a value nobody wrote would show up as a
    // NaN flow field instead of a silently run-dependent one (tests rely on this).
    if (hipMemset(pl->arena, 0xFF, pl->arena_bytes) != hipSuccess || (pl->xarena && hipMemset(pl->xarena, 0xFF, pl->xarena_bytes) != hipSuccess)) {
        g_last_error = "hipMemset of the plan arena failed";
        octane_vof_plan_destroy(pl);
        return OCTANE_E_HIP;
    }
    // plane pointers for an arena starting at `base` (and, for a band plan, shared planes starting in the allocation `xbase`)
    auto carve = [&](float *base, float *xbase) {
        // planes sit at the same OFFSETS in every arena (row bands address a neighbour's plane as "my pointer moved by the
        // distance between the arena bases", BandNet::peer), so nothing here may depend on the absolute address
        float *cur = base;
        // multi-channel fields are addressed as base + c * plane0, so their planes stay plane0 apart
        auto take = [&](size_t n) { float *r = cur; cur += (n - 1) * pl->plane0 + stride; return r; };
        pl->img1p = take(nc); pl->img2p = take(nc); pl->uh = take(1); pl->vh = take(1);
        pl->lev1 = take(nc); pl->lev2 = take(nc);
        pl->gx1 = take(nc); pl->gy1 = take(nc); pl->gx2 = take(nc); pl->gy2 = take(nc);
        pl->gxx = take(nc); pl->gxy = take(nc); pl->gyy = take(nc);
        if (xbase) cur = xbase + kGuardFloats;       // band plan: from here on the planes neighbouring bands read
        pl->U[0] = take(1); pl->U[1] = take(1); pl->V[0] = take(1); pl->V[1] = take(1);
        pl->ut = take(1); pl->vt = take(1);
        pl->a1 = take(1); pl->a2 = take(1); pl->a4 = take(1); pl->wx = take(1); pl->wy = take(1); pl->mu = take(1); pl->mv = take(1);
        pl->ru = take(1); pl->rv = take(1);
        pl->pu[0] = take(1); pl->pu[1] = take(1); pl->pv[0] = take(1); pl->pv[1] = take(1);
        pl->qu = take(1); pl->qv = take(1); pl->xu = take(1); pl->xv = take(1);
        pl->tmp = take(1);
        pl->ru2 = take(1); pl->rv2 = take(1); pl->qu2 = take(1); pl->qv2 = take(1);
        pl->pu3 = take(1); pl->pv3 = take(1);
        // (new planes go here, at the end: the offsets of the ones above are what row bands address each other's arenas by)
        if (pl->has_bset) {
            pl->lev1b = take(nc); pl->lev2b = take(nc); pl->utb = take(1); pl->vtb = take(1);
            pl->gx1b = take(nc); pl->gy1b = take(nc); pl->gx2b = take(nc); pl->gy2b = take(nc);
            pl->gxxb = take(nc); pl->gxyb = take(nc); pl->gyyb = take(nc);
        } else {          // never used as a second set (run_on_stream checks has_bset): alias the first
            pl->lev1b = pl->lev1; pl->lev2b = pl->lev2; pl->utb = pl->ut; pl->vtb = pl->vt;
            pl->gx1b = pl->gx1; pl->gy1b = pl->gy1; pl->gx2b = pl->gx2; pl->gy2b = pl->gy2;
            pl->gxxb = pl->gxx; pl->gxyb = pl->gxy; pl->gyyb = pl->gyy;
        }
    };
    carve(pl->arena, pl->xarena);

    int rc = OCTANE_OK;
    do {
        size_t ntaps = taps.empty() ? 1 : taps.size();
        if (hipMalloc((void **)&pl->d_taps, ntaps * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (!taps.empty() &&
            hipMemcpy(pl->d_taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMalloc((void **)&pl->d_parts, 2 * kPartBlock * sizeof(double)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&pl->d_state, 2 * sizeof(PcgState)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&pl->d_alpha, 2 * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMemset(pl->d_alpha, 0, 2 * sizeof(float)) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMalloc((void **)&pl->d_iters, sizeof(long long)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipHostMalloc((void **)&pl->h_iters, sizeof(long long)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        *pl->h_iters = 0;
        if (hipMalloc(&pl->d_mid, pcg_mid_workspace_bytes()) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMemset(pl->d_mid, 0, pcg_mid_workspace_bytes()) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipHostMalloc((void **)&pl->h_mid_abort, 16) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        *pl->h_mid_abort = 0;
        if (hipMemset(pl->d_parts, 0, 2 * kPartBlock * sizeof(double)) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemset(pl->d_state, 0, 2 * sizeof(PcgState)) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamCreateWithFlags(&pl->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamCreateWithFlags(&pl->side_stream, hipStreamNonBlocking) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        for (int i = 0; i < 2 && rc == OCTANE_OK; i++) {
            if (hipEventCreateWithFlags(&pl->ev_img[i], hipEventDisableTiming) != hipSuccess) rc = OCTANE_E_HIP;
            else if (hipEventCreateWithFlags(&pl->ev_solved[i], hipEventDisableTiming) != hipSuccess) rc = OCTANE_E_HIP;
        }
        if (rc != OCTANE_OK) break;
    } while (0);
    if (rc != OCTANE_OK) {
        g_last_error = "octane_vof_plan_create: device allocation failed";
        octane_vof_plan_destroy(pl);
        return rc;
    }
    // Placement trials.  Where the arena lands in physical memory decides how the concurrent streams of a PCG kernel
    // spread over the HBM channels: arenas come in two kinds, 0.41 or 0.45-0.46 ms per fused iteration at 5000^2, and
    // consecutive allocations of one process tend to be of the same kind (EXPERIMENTS.md 8).  For large frames up to eight
    // candidate arenas are therefore allocated, a few PCG iterations are timed on each, the fastest is kept and the
    // others are freed.
    int trials = placement_trials;
    if (const char *e = tune_env("OCTANE_TUNE_PLACEMENT_TRIALS")) trials = atoi(e) < trials ? atoi(e) : trials;
    constexpr int kMaxTrials = 8;
    if (const char *e = tune_env("OCTANE_TUNE_PLACEMENT_TRIALS_FORCE")) trials = atoi(e);       // experiments: more than the caller's default
    if (trials > kMaxTrials) trials = kMaxTrials;
    // (a band plan's candidates are candidates for its SHARED allocation -- the operator and the CG vectors are what a PCG launch streams)
    const size_t trial_bytes = pl->shared_bytes();
    {   // the candidates exist side by side while they are timed: keep that within 48 GB
        const long fit = (long)(((size_t)48 << 30) / trial_bytes);
        if (trials > fit) trials = (int)fit;
    }
    if (trials > 1 && (long)nx * ny >= (1L << 22)) {
        float *cand[kMaxTrials] = {pl->shared_base()};
        double ms[kMaxTrials] = {0};
        int ncand = 1;
        // experiment (OCTANE_TUNE_ARENA_CONTIG=1): odd candidates ask for physically contiguous memory
        const bool try_contig = tune_env("OCTANE_TUNE_ARENA_CONTIG") && atoi(tune_env("OCTANE_TUNE_ARENA_CONTIG")) != 0;
        for (int t = 1; t < trials; t++) {
            hipError_t ae = (try_contig && (t & 1)) ? hipExtMallocWithFlags((void **)&cand[t], trial_bytes, hipDeviceMallocContiguous)
                                                    : hipMalloc((void **)&cand[t], trial_bytes);
            if (ae != hipSuccess && try_contig && (t & 1)) { (void)hipGetLastError(); ae = hipMalloc((void **)&cand[t], trial_bytes); if (tune_env("OCTANE_TUNE_VERBOSE")) fprintf(stderr, "[octane] contiguous candidate %d refused\n", t); }
            if (ae != hipSuccess) { (void)hipGetLastError(); cand[t] = nullptr; break; }
            ncand = t + 1;
        }
        auto carve_cand = [&](float *c) { if (pl->xarena) carve(pl->arena, c); else carve(c, nullptr); };
        int best = 0;
        for (int t = 0; t < ncand; t++) {
            carve_cand(cand[t]);
            ms[t] = probe_placement(pl);
            if (ms[t] > 0 && (ms[best] <= 0 || ms[t] < ms[best])) best = t;
        }
        if (tune_env("OCTANE_TUNE_VERBOSE")) {
            fprintf(stderr, "[octane] placement trials, ms per PCG iteration:");
            for (int t = 0; t < ncand; t++) fprintf(stderr, " %.4f@%p", ms[t], (void *)cand[t]);
            fprintf(stderr, " -> candidate %d\n", best);
        }
        pl->ntrials = ncand;
        for (int t = 0; t < ncand; t++) pl->trial_ms[t] = ms[t];
        for (int t = 0; t < ncand; t++)
            if (t != best) (void)hipFree(cand[t]);
        if (pl->xarena) pl->xarena = cand[best]; else pl->arena = cand[best];
        carve_cand(cand[best]);
        if (hipMemset(cand[best], 0xFF, trial_bytes) != hipSuccess) {      // restore the poison the probe disturbed
            g_last_error = "hipMemset of the plan arena failed";
            octane_vof_plan_destroy(pl);
            return OCTANE_E_HIP;
        }
    }
    // hipMemset on device memory is enqueued on the null stream and may return before it has run; the plan's own
    // stream does not synchronise with the null stream, so without this a first upload could be overtaken by the
    // poison fill (seen with two plans created back to back).
    if (hipDeviceSynchronize() != hipSuccess) {
        g_last_error = "octane_vof_plan_create: device synchronisation failed";
        octane_vof_plan_destroy(pl);
        return OCTANE_E_HIP;
    }
    *out = pl;
    return OCTANE_OK;
}

extern "C" size_t octane_vof_plan_device_bytes(const octane_vof_plan *pl) { return pl ? pl->arena_bytes + pl->xarena_bytes : 0; }

extern "C" int octane_vof_plan_placement_trials(const octane_vof_plan *pl, double *ms, int cap)
{
    if (!pl) return 0;
    for (int t = 0; t < pl->ntrials && t < cap; t++)
        if (ms) ms[t] = pl->trial_ms[t];
    return pl->ntrials;
}

extern "C" int octane_vof_plan_set_trace(octane_vof_plan *pl, octane_vof_trace_fn fn, void *user)
{
    if (!pl) return OCTANE_E_INVALID;
    pl->trace = fn; pl->trace_user = user;
    return OCTANE_OK;
}

extern "C" int octane_vof_plan_set_profiling(octane_vof_plan *pl, int enable)
{
    if (!pl) return OCTANE_E_INVALID;
    pl->profiling = enable ? 1 : 0;
    return OCTANE_OK;
}

// ---- debug tap: copy planes to a dense host buffer and hand them to the callback -------------
static int emit(octane_vof_plan *pl, hipStream_t s, const char *tag, int k, int gnc, int l,
                std::initializer_list<const float *> planes, int w, int h, int pitch)
{
    if (!pl->trace || (pl->trace_levels > 0 && k >= pl->trace_levels)) return OCTANE_OK;
    std::vector<float> host((size_t)w * h * planes.size());
    HIP_TRY(hipStreamSynchronize(s));
    size_t i = 0;
    for (const float *p : planes) {
        HIP_TRY(hipMemcpy2D(host.data() + i * (size_t)w * h, (size_t)w * sizeof(float), p, (size_t)pitch * sizeof(float),
                            (size_t)w * sizeof(float), h, hipMemcpyDeviceToHost));
        i++;
    }
    pl->trace(pl->trace_user, tag, k, gnc, l, host.data(), w, h, (int)planes.size());
    return OCTANE_OK;
}

static int emit_chan(octane_vof_plan *pl, hipStream_t s, const char *tag, int k, const float *base, int w, int h, int pitch)
{
    if (!pl->trace) return OCTANE_OK;
    if (pl->nc == 1) return emit(pl, s, tag, k, -1, -1, {base}, w, h, pitch);
    if (pl->nc == 2) return emit(pl, s, tag, k, -1, -1, {base, base + pl->plane0}, w, h, pitch);
    return emit(pl, s, tag, k, -1, -1, {base, base + pl->plane0, base + 2 * pl->plane0}, w, h, pitch);
}

// ---- profiling helpers ---------------------------------------------------------------------
enum { EV_PASS_A = 0, EV_PASS_B = 1, EV_ASM = 2, EV_UPD = 3 };

static EvPair *ev_begin(octane_vof_plan *pl, hipStream_t s, int kind, bool on)
{
    if (!on) return nullptr;
    if (pl->evs_used == pl->evs.size()) {
        EvPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        p.kind = kind;
        pl->evs.push_back(p);
    }
    EvPair *p = &pl->evs[pl->evs_used++];
    p->kind = kind;
    (void)hipEventRecord(p->a, s);
    return p;
}
static void ev_end(EvPair *p, hipStream_t s) { if (p) (void)hipEventRecord(p->b, s); }

int octane::plan_level_images(octane_vof_plan *pl, hipStream_t s, int k, int which, LevelCtx &c)
{
    const octane_vof_params &prm = pl->prm;
    const int nc = pl->nc;
    const int nlev = (int)pl->lev.size();
    const bool hint = (prm.lambdac != 0.);
    const LevelInfo &li = pl->lev[k];
    const bool finest = (k == nlev - 1);
    float *lev1 = which ? pl->lev1b : pl->lev1, *lev2 = which ? pl->lev2b : pl->lev2;
    // without the hint term only level 0 forms the decimated first guess (and nothing reads it afterwards): one pair of planes will do
    float *ut = (which && hint) ? pl->utb : pl->ut, *vt = (which && hint) ? pl->vtb : pl->vt;
    c.gx1 = which ? pl->gx1b : pl->gx1; c.gy1 = which ? pl->gy1b : pl->gy1; c.gx2 = which ? pl->gx2b : pl->gx2; c.gy2 = which ? pl->gy2b : pl->gy2;
    c.gxx = which ? pl->gxxb : pl->gxx; c.gxy = which ? pl->gxyb : pl->gxy; c.gyy = which ? pl->gyyb : pl->gyy;
    if (finest) {  // ref .cu:504-517: the finest level uses the inputs themselves
        c.lev1 = pl->img1p; c.lev2 = pl->img2p; c.ut = pl->uh; c.vt = pl->vh;
    } else {       // ref .cu:519-563
        const float *gk = pl->d_taps + li.tap_off;
        // channel 0 only: the reference's zoom_out samples channel 0 for every channel (.cu:406)
        launch_blur_rows_sampled(s, pl->img1p, pl->nx, pl->ny, pl->pitch0, pl->tmp, li.w, li.pitch, gk, li.fs, li.factor);
        launch_blur_cols_sampled(s, pl->tmp, li.w, pl->ny, li.pitch, lev1, li.h, li.pitch, gk, li.fs, li.factor, 1.f, 0);
        launch_blur_rows_sampled(s, pl->img2p, pl->nx, pl->ny, pl->pitch0, pl->tmp, li.w, li.pitch, gk, li.fs, li.factor);
        launch_blur_cols_sampled(s, pl->tmp, li.w, pl->ny, li.pitch, lev2, li.h, li.pitch, gk, li.fs, li.factor, 1.f, 0);
        for (int ch = 1; ch < nc; ch++) {
            launch_copy2d(s, lev1, li.pitch, lev1 + ch * pl->plane0, li.pitch, li.w, li.h);
            launch_copy2d(s, lev2, li.pitch, lev2 + ch * pl->plane0, li.pitch, li.w, li.h);
        }
        if (hint || k == 0) {   // decimated first guess, scaled to this level's pixel size
            launch_blur_rows_sampled(s, pl->uh, pl->nx, pl->ny, pl->pitch0, pl->tmp, li.w, li.pitch, gk, li.fs, li.factor);
            launch_blur_cols_sampled(s, pl->tmp, li.w, pl->ny, li.pitch, ut, li.h, li.pitch, gk, li.fs, li.factor, li.factor, 1);
            launch_blur_rows_sampled(s, pl->vh, pl->nx, pl->ny, pl->pitch0, pl->tmp, li.w, li.pitch, gk, li.fs, li.factor);
            launch_blur_cols_sampled(s, pl->tmp, li.w, pl->ny, li.pitch, vt, li.h, li.pitch, gk, li.fs, li.factor, li.factor, 1);
        }
        c.lev1 = lev1; c.lev2 = lev2; c.ut = ut; c.vt = vt;
    }
    // ref .cu:587-595; d/dy of gx2 is dead (overwritten by the fourth call), so it is not stored
    launch_gradient(s, c.lev1, c.gx1, c.gy1, li.w, li.h, li.pitch, nc, pl->plane0);
    launch_gradient(s, c.lev2, c.gx2, c.gy2, li.w, li.h, li.pitch, nc, pl->plane0);
    launch_gradient(s, c.gx2, c.gxx, nullptr, li.w, li.h, li.pitch, nc, pl->plane0);
    launch_gradient(s, c.gy2, c.gxy, c.gyy, li.w, li.h, li.pitch, nc, pl->plane0);
    return OCTANE_OK;
}

// ref .cu:498-503: bicubic up-sample of the previous level's flow, divided by scaleF (flips `cur`)
static void plan_level_upsample(octane_vof_plan *pl, hipStream_t s, int k, int &cur)
{
    if (k == 0) return;
    const LevelInfo &li = pl->lev[k], &lo = pl->lev[k - 1];
    launch_upsample(s, pl->U[cur], lo.w, lo.h, lo.pitch, pl->U[cur ^ 1], li.w, li.h, li.pitch, (float)pl->prm.scaleF);
    launch_upsample(s, pl->V[cur], lo.w, lo.h, lo.pitch, pl->V[cur ^ 1], li.w, li.h, li.pitch, (float)pl->prm.scaleF);
    cur ^= 1;
}

// ref .cu:576-585: the coarsest level starts from the decimated first guess
static void plan_level_first_guess(octane_vof_plan *pl, hipStream_t s, int k, int cur, const LevelCtx &c)
{
    if (k != 0) return;
    const LevelInfo &li = pl->lev[0];
    launch_copy2d(s, c.ut, li.pitch, pl->U[cur], li.pitch, li.w, li.h);
    launch_copy2d(s, c.vt, li.pitch, pl->V[cur], li.pitch, li.w, li.h);
}

int octane::plan_level_setup(octane_vof_plan *pl, hipStream_t s, int k, int &cur, LevelCtx &c)
{
    plan_level_upsample(pl, s, k, cur);
    const int rc = plan_level_images(pl, s, k, 0, c);
    if (rc) return rc;
    plan_level_first_guess(pl, s, k, cur, c);
    return OCTANE_OK;
}

// Launches of the persistent solve need ALL their workgroups resident at once (they meet at grid barriers).  Two such
// launches running side by side on one device -- the lanes of a batch, virtual row bands sharing a GPU -- could each hold
// CUs the other is waiting for, so within this process they are serialised among themselves per device: every launch waits
// for the event the previous one recorded (on whatever stream that was) and records the next.  Everything else overlaps as
// before.  (Another PROCESS on the same GPU is not covered: the kernel's barriers are bounded and abort the solve.)
// A plan that runs beside other plans on the same device (the lanes of octane_vof_batch_run, bench.py --lanes): its persistent
// solves are serialised with theirs (persist_launch below), so only the tiny levels keep them (<= 16 workgroups: 157^2 and below)
// and everything one workgroup can hold goes to the single-workgroup solve, whose launches overlap freely.
static void plan_lane_mode(octane_vof_plan *pl) { pl->persist_max_g = 16; pl->small_max_pixels = 6144; }
// TWO plans side by side on one device (the two lanes of octane_vof_batch_run, bench.py --workload batch64 / --lanes 2; round 5): each
// caps its persistent solves at HALF the compute units and launches them WITHOUT waiting for the other's -- 2 x ncu / 2 workgroups are
// always co-resident, so the two solves run concurrently instead of queueing behind each other.  Measured, two lanes of 2000^2 pairs
// (tools/lanes_concurrent.py, profiles/r5_lanes_concurrent.txt): serialised + uncapped (rounds 2-4) 200.2 Mpix/s, concurrent + capped at
// half 219.1 (+9.4 %), serialised + capped 195.3, concurrent + uncapped 163.3 (a solve is abandoned now and then: 2 x 240 workgroups do
// not always become resident beside the other lane's streaming kernels).  The cap changes the sub-domain grid of the mid-size levels and
// with it the grouping of their fp64 partial sums: flows equal the default plan's to the last bits of the PCG scalars, not bit for bit.
static void plan_lane_pair_mode(octane_vof_plan *pl) { pl->persist_max_g = pl->ncu > 1 ? pl->ncu / 2 : 128; pl->persist_chain = 0; }
static std::mutex g_persist_mu;
#ifdef OCTANE_DIAG
static int g_persist_diag = 0;      // octane_vof_tune(plan, "persist_diag", 1): the stamped build of the persistent solve (diagnostic library only)
#endif
static hipEvent_t g_persist_ev[64] = {nullptr};
static int persist_launch(octane_vof_plan *pl, hipStream_t s, const LevelPtrs &L, const MidGeom &mg, unsigned seq, int k0, int k1, int kcap, int nparts_asm)
{
    std::lock_guard<std::mutex> g(g_persist_mu);
    const int d = pl->device & 63;
    if (!pl->persist_chain) {      // EXPERIMENT (round 5, tune "persist_chain" 0): this plan's persistent launches neither wait for nor hold up the others'
#ifdef OCTANE_DIAG
        hipError_t e0 = (g_persist_diag ? launch_pcg_solve_mid_diag : launch_pcg_solve_mid)(s, L, mg, pl->d_mid, seq, k0, k1, kcap, nparts_asm, pl->tol);
#else
        hipError_t e0 = launch_pcg_solve_mid(s, L, mg, pl->d_mid, seq, k0, k1, kcap, nparts_asm, pl->tol);
#endif
        if (e0 != hipSuccess) { g_last_error = std::string("persistent solve: ") + hipGetErrorString(e0); return OCTANE_E_HIP; }
        return OCTANE_OK;
    }
    if (!g_persist_ev[d]) {
        if (hipEventCreateWithFlags(&g_persist_ev[d], hipEventDisableTiming) != hipSuccess) { g_last_error = "persistent solve: hipEventCreate failed"; return OCTANE_E_HIP; }
    } else if (hipStreamWaitEvent(s, g_persist_ev[d], 0) != hipSuccess) {
        g_last_error = "persistent solve: hipStreamWaitEvent failed"; return OCTANE_E_HIP;
    }
#ifdef OCTANE_DIAG
    hipError_t e = (g_persist_diag ? launch_pcg_solve_mid_diag : launch_pcg_solve_mid)(s, L, mg, pl->d_mid, seq, k0, k1, kcap, nparts_asm, pl->tol);
#else
    hipError_t e = launch_pcg_solve_mid(s, L, mg, pl->d_mid, seq, k0, k1, kcap, nparts_asm, pl->tol);
#endif
    if (e == hipSuccess) e = hipEventRecord(g_persist_ev[d], s);
    if (e != hipSuccess) { g_last_error = std::string("persistent solve: ") + hipGetErrorString(e); return OCTANE_E_HIP; }
    return OCTANE_OK;
}

int octane::plan_level_solve(octane_vof_plan *pl, hipStream_t s, int k, int cur, const LevelCtx &c, bool pf)
{
    const octane_vof_params &prm = pl->prm;
    const LevelInfo &li = pl->lev[k];
    LevelPtrs L;
    fill_level_ptrs(pl, li, cur, c, L);

    const int g_asm = assemble_grid_size(li.w, li.h);
    L.q_form = pcg_fused_q_form(li.w, li.h, li.h);
    const int g_f_plain = pcg_fused_grid_size(li.w, li.h, 0, L.q_form), g_f_unit = pcg_fused_grid_size(li.w, li.h, 1, L.q_form);
    L.row_rot = L.q_form ? pcg_row_rotation(li.w, li.h, g_f_plain, pl->xcd_bands) : 0;     // (the q-form's grid is the same with and without unit weights)
    const int g_a_plain = pcg_grid_size(li.w, li.h);
    const int g_a_unit = pcg_grid_size_unit_w(li.w, li.h);
    const int g_b = pcg_b_grid_size(li.w, li.h);
    // mid-size levels: the whole solve in one persistent launch, the level resident on chip (pcg_persist.hip)
    MidGeom mg;
    const bool mid_ok = pl->use_persist && pl->use_fused && !pl->use_graph && pl->d_mid && (long)li.w * li.h <= pl->persist_max_pixels &&
                        pcg_mid_config(li.w, li.h, pl->ncu < pl->persist_max_g ? pl->ncu : pl->persist_max_g, pl->persist_p, &mg) == 1;
    // the coarsest levels: one workgroup holds the whole level (k_pcg_solve_small, <= 6144 pixels).  With up to three pixels per thread
    // (<= 1536 pixels) nothing beats it (39^2: 2.9 us per iteration against 3.4 for five one-slot persistent sub-domains); with 6 or 12
    // pixels per thread the persistent sub-domains are faster (45^2: 3.4 against 3.6, 55^2: 3.4 against 4.2, 63^2: 3.4 against 6.0;
    // tools/small_vs_mid.py), so the single workgroup keeps only what is smaller, or everything if the persistent solve is not available
    const bool small = pl->use_small && pcg_small_applicable(li.w, li.h) && (!mid_ok || (long)li.w * li.h <= pl->small_max_pixels);
    const bool mid = !small && mid_ok;

    for (int gnc = 0; gnc < 3; gnc++) {                 // ref .cu:604-606
        AssembleParams ap;
        ap.al1 = 1. - 0.5 * gnc;
        ap.alpha = prm.alpha;
        ap.loa = prm.lambda / prm.alpha;                // ref .cu:1230
        ap.lambdac = li.lambdac;
        ap.dozim = prm.dozim != 0;
        ap.ralpha = 1. / prm.alpha;                     // correctly rounded (host division)
        ap.fast_math = pl->asm_fast;
        L.unit_w = (pl->use_unit_w && gnc == 0) ? 1 : 0;   // al1 == 1: all neighbour weights are exactly -1
        L.lean = (pl->use_fused && !small && !pl->trace) ? 1 : 0;
        const int g_a = L.unit_w ? g_a_unit : g_a_plain;
        const int g_f = L.unit_w ? g_f_unit : g_f_plain;
        for (int l = 0; l < prm.liters; l++) {          // ref .cu:608
            EvPair *e = ev_begin(pl, s, EV_ASM, pf);
            if (pl->use_fused && !small) {     // the fused kernels double-buffer the partials; the rhs sums go to the "-1" block
                LevelPtrs La = L;
                La.part_rz = L.part_own + kPartBlock + kPartRz; La.part_rr = L.part_own + kPartBlock + kPartRr;
                launch_assemble(s, La, ap, g_asm);
            } else {
                launch_assemble(s, L, ap, g_asm);
            }
            ev_end(e, s);
            if (pl->trace) {
                int rc = emit(pl, s, "coef7", k, gnc, l, {pl->a1, pl->a2, pl->a4, pl->wx, pl->wy, pl->ru, pl->rv}, li.w, li.h, li.pitch);
                if (rc) return rc;
            }
            if (small) {       // coarsest levels: the whole solve and the flow update in one workgroup
                launch_pcg_solve_small(s, L, prm.cgiters, pl->tol);
            } else if (mid) {  // the whole solve and the flow update in one launch of one workgroup per sub-domain
                const int step = pl->persist_step > 0 ? pl->persist_step : (prm.cgiters > 0 ? prm.cgiters : 1);
                const unsigned seq = pl->mid_seq++;            // one tag range per solve, shared by the launches of the stepped form
                for (int k0 = 0; k0 < prm.cgiters || k0 == 0; k0 += step) {
                    const int k1 = k0 + step < prm.cgiters ? k0 + step : prm.cgiters;
                    e = ev_begin(pl, s, EV_PASS_A, pf);
                    const int rc = persist_launch(pl, s, L, mg, seq, k0, k1, prm.cgiters, g_asm);
                    ev_end(e, s);
                    if (rc) return rc;
                    if (k1 >= prm.cgiters) break;
                }
            } else if (pl->use_fused) {
                std::vector<double> sums;                   // debug tap only: what every launch summed and decided
                for (int it = 0; it < prm.cgiters; it++) {  // ref .cu:1131-1182, one kernel per iteration
                    e = ev_begin(pl, s, EV_PASS_A, pf);
                    launch_pcg_fused(s, L, it, it == 0 ? g_asm : g_f, g_f, pl->tol);
                    ev_end(e, s);
                    if (pl->trace && L.nbands == 1) {
                        // Tag "pcg_sums": per launch ten doubles -- the launch's own DIRECT sums over the state it formed (r.z, r.r, p.q, q.z,
                        // q.M^-1 q, r.q, q.q: the seven partial-sum kinds folded on the host in index order), then the PcgState it left
                        // (rz = the r.z it USED: the assembly's direct sum at launch 0, the one-step recurrence afterwards; stopped; iters).
                        // tests/test_gpu_parity.py forms the next launch's predicted r.z / r.r from row k and compares them with row k + 1's
                        // direct sums (the second arithmetic freedom of DESIGN 4).  Delivered as nx = 20 floats per row = the doubles' bytes.
                        HIP_TRY(hipStreamSynchronize(s));
                        std::vector<double> blk((size_t)kPartBlock);
                        HIP_TRY(hipMemcpy(blk.data(), L.part_own + (size_t)(it & 1) * kPartBlock, (size_t)kPartBlock * sizeof(double), hipMemcpyDeviceToHost));
                        PcgState st;
                        HIP_TRY(hipMemcpy(&st, &L.st[(it + 1) & 1], sizeof(st), hipMemcpyDeviceToHost));
                        for (int j = 0; j < kPartKinds; j++) {
                            double t = 0.;
                            for (int i = 0; i < g_f; i++) t += blk[(size_t)j * kMaxParts + i];
                            sums.push_back(t);
                        }
                        sums.push_back((double)st.rz); sums.push_back((double)st.stopped); sums.push_back((double)st.iters);
                    }
                }
                if (pl->trace && !sums.empty() && !(pl->trace_levels > 0 && k >= pl->trace_levels))
                    pl->trace(pl->trace_user, "pcg_sums", k, gnc, l, reinterpret_cast<const float *>(sums.data()), 20, (int)(sums.size() / 10), 1);
                e = ev_begin(pl, s, EV_UPD, pf);
                launch_flow_update_fused(s, L, prm.cgiters, g_f);   // ref .cu:1185-1195 (+ the last x update)
                ev_end(e, s);
            } else {
#ifdef OCTANE_DIAG      // pass A + pass B per iteration: what the fused kernels are compared with (tune "fused" 0); diagnostic library only
                for (int it = 0; it < prm.cgiters; it++) {  // ref .cu:1131-1182
                    e = ev_begin(pl, s, EV_PASS_A, pf);
                    launch_pcg_pass_a(s, L, it, it == 0 ? g_asm : g_b, g_a, pl->tol);
                    ev_end(e, s);
                    e = ev_begin(pl, s, EV_PASS_B, pf);
                    launch_pcg_pass_b(s, L, it, g_a, g_b);
                    ev_end(e, s);
                }
                e = ev_begin(pl, s, EV_UPD, pf);
                launch_flow_update(s, L, prm.cgiters);      // ref .cu:1185-1195
                ev_end(e, s);
#else
                (void)g_a; (void)g_b;
                g_last_error = "the two-pass form of the PCG iteration is built into the diagnostic library only";
                return OCTANE_E_INVALID;
#endif
            }
            if (pl->trace) {
                int rc;
                if ((rc = emit(pl, s, "dx2", k, gnc, l, {pl->xu, pl->xv}, li.w, li.h, li.pitch))) return rc;
                if ((rc = emit(pl, s, "u", k, gnc, l, {pl->U[cur]}, li.w, li.h, li.pitch))) return rc;
                if ((rc = emit(pl, s, "v", k, gnc, l, {pl->V[cur]}, li.w, li.h, li.pitch))) return rc;
            }
        }
    }
    return OCTANE_OK;
}

static int run_on_stream(octane_vof_plan *pl, hipStream_t s)
{
    const int nlev = (int)pl->lev.size();
    int cur = 0;       // U[cur], V[cur] hold the flow of the level being solved
    pl->evs_used = 0;
    const bool prof = pl->profiling != 0;
    if (prof) {
        if (!pl->ev_t0) { (void)hipEventCreate(&pl->ev_t0); (void)hipEventCreate(&pl->ev_t1); (void)hipEventCreate(&pl->ev_s0); (void)hipEventCreate(&pl->ev_s1); }
        (void)hipEventRecord(pl->ev_t0, s);
    }
    HIP_TRY(hipMemsetAsync(pl->d_iters, 0, sizeof(long long), s));
    // the abort word of the persistent solves is per RUN: cleared here, on the stream, before the first of them (a caller that
    // drives the plan with device buffers and its own synchronisation never reaches persist_check, which used to be the only
    // place that cleared it: one abort then made every later solve that polled 128 times give up at once); this run's value
    // is copied to the host at the end of the run
    if (pl->d_mid) HIP_TRY(hipMemsetAsync(static_cast<char *>(pl->d_mid) + 8, 0, sizeof(unsigned), s));
    double setup_ms = 0.;

    // Everything a level needs apart from the flow -- its pyramid images, the decimated first guess, seven gradient fields -- depends on
    // the inputs alone.  The coarse levels' solves are latency-bound launches that leave the GPU idle (R1: 11 of 120 ms), so a side
    // stream prepares level k + 1 in the second set of planes while level k is solved out of the first, and so on alternately; the
    // solve of level k waits for the event of its images, the side stream waits for the solve that last read the set it overwrites.
    const bool overlap = pl->use_overlap && pl->has_bset && !prof && !pl->trace && pl->side_stream && nlev > 1 && !pl->use_graph;
    LevelCtx next_c;
    // An early return with an error must not leave work of this run on the side stream behind it: whoever uses the plan
    // next -- its destruction, a re-solve after an abandoned persistent solve -- would race that work.  The normal end needs no join:
    // every side-stream piece is waited for by the event of the level that consumes it.
    struct SideJoin {
        hipStream_t side; bool armed;
        ~SideJoin() { if (armed && side) (void)hipStreamSynchronize(side); }
    } side_join{pl->side_stream, overlap};
    if (overlap) {
        hipStream_t side = pl->side_stream;
        HIP_TRY(hipEventRecord(pl->ev_fork, s));                 // the inputs are in place (and the previous run on this stream is over)
        HIP_TRY(hipStreamWaitEvent(side, pl->ev_fork, 0));
        int rc = plan_level_images(pl, side, 0, 0, next_c);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(pl->ev_img[0], side));
    }

    for (int k = 0; k < nlev; k++) {
        const LevelInfo &li = pl->lev[k];
        const bool finest = (k == nlev - 1);
        LevelCtx c;
        if (overlap) {
            c = next_c;
            if (k + 1 < nlev) {                                  // level k + 1 into the other set, once the solve of level k - 1 has left it
                hipStream_t side = pl->side_stream;
                if (k >= 1) HIP_TRY(hipStreamWaitEvent(side, pl->ev_solved[(k + 1) & 1], 0));
                int rc = plan_level_images(pl, side, k + 1, (k + 1) & 1, next_c);
                if (rc) return rc;
                HIP_TRY(hipEventRecord(pl->ev_img[(k + 1) & 1], side));
            }
            plan_level_upsample(pl, s, k, cur);
            // (ev_img[k & 1] was recorded for level k before the call above re-recorded the OTHER event; a wait takes the event's
            // latest record at the time of the call, so it has to come after that record and before the next one of the same event)
            HIP_TRY(hipStreamWaitEvent(s, pl->ev_img[k & 1], 0));
            plan_level_first_guess(pl, s, k, cur, c);
            int rc = plan_level_solve(pl, s, k, cur, c, false);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(pl->ev_solved[k & 1], s));
            continue;
        }
        if (prof) (void)hipEventRecord(pl->ev_s0, s);
        int rc = plan_level_setup(pl, s, k, cur, c);
        if (rc) return rc;
        if (prof) {
            (void)hipEventRecord(pl->ev_s1, s);
            (void)hipEventSynchronize(pl->ev_s1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, pl->ev_s0, pl->ev_s1);
            setup_ms += ms;
        }

        if (pl->trace) {
            if ((rc = emit_chan(pl, s, "img1", k, c.lev1, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "img2", k, c.lev2, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gx1", k, c.gx1, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gy1", k, c.gy1, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gx2", k, c.gx2, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gy2", k, c.gy2, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gxx", k, c.gxx, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gxy", k, c.gxy, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit_chan(pl, s, "gyy", k, c.gyy, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit(pl, s, "u0", k, -1, -1, {pl->U[cur]}, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit(pl, s, "v0", k, -1, -1, {pl->V[cur]}, li.w, li.h, li.pitch))) return rc;
        }

        if ((rc = plan_level_solve(pl, s, k, cur, c, prof && finest))) return rc;
        if (pl->trace) {
            if ((rc = emit(pl, s, "ulev", k, -1, -1, {pl->U[cur]}, li.w, li.h, li.pitch))) return rc;
            if ((rc = emit(pl, s, "vlev", k, -1, -1, {pl->V[cur]}, li.w, li.h, li.pitch))) return rc;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pl->h_iters, pl->d_iters, sizeof(long long), hipMemcpyDeviceToHost, s));
    if (pl->d_mid) HIP_TRY(hipMemcpyAsync(pl->h_mid_abort, static_cast<char *>(pl->d_mid) + 8, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    if (prof) {
        (void)hipEventRecord(pl->ev_t1, s);
        HIP_TRY(hipEventSynchronize(pl->ev_t1));
        octane_vof_profile pr;
        memset(&pr, 0, sizeof(pr));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, pl->ev_t0, pl->ev_t1);
        pr.total_ms = ms; pr.setup_ms = setup_ms;
        pr.finest_pixels = (long long)pl->nx * pl->ny;
        pl->pcg_launch_ms.clear();
        for (size_t i = 0; i < pl->evs_used; i++) {
            (void)hipEventElapsedTime(&ms, pl->evs[i].a, pl->evs[i].b);
            switch (pl->evs[i].kind) {
            case EV_PASS_A: pr.pass_a_ms += ms; pr.pass_a_launches++; pl->pcg_launch_ms.push_back(ms); break;
            case EV_PASS_B: pr.pass_b_ms += ms; pr.pass_b_launches++; break;
            case EV_ASM: pr.assemble_ms += ms; pr.assemble_launches++; break;
            default: pr.update_ms += ms; pr.update_launches++; break;
            }
        }
        pl->prof = pr;
    }
    // the result lives in U[cur], V[cur]; remember which for the copy-out
    pl->prof.finest_pixels = (long long)pl->nx * pl->ny;
    side_join.armed = false;
    return cur;   // >= 0
}

extern "C" int octane_vof_plan_get_profile(octane_vof_plan *pl, octane_vof_profile *out)
{
    if (!pl || !out) return OCTANE_E_INVALID;
    *out = pl->prof;
    return OCTANE_OK;
}

// The finest-level PCG launches of the last profiled run, one duration each, in launch order (solve after solve: 3 GNC steps x liters
// solves of cgiters launches); returns how many there are and writes the first min(count, cap) into ms.
extern "C" int octane_vof_plan_get_launch_times(octane_vof_plan *pl, float *ms, int cap)
{
    if (!pl || (cap > 0 && !ms)) return OCTANE_E_INVALID;
    const int n = (int)pl->pcg_launch_ms.size();
    for (int i = 0; i < n && i < cap; i++) ms[i] = pl->pcg_launch_ms[i];
    return n;
}

// A persistent solve gives up when its workgroups cannot all become resident (its grid barrier is bounded): the flow of that
// run is not valid.  Reported by every call that has synchronised with the run; the word is cleared so that the plan stays usable.
int octane::persist_check(octane_vof_plan *pl)
{
    if (!pl->h_mid_abort || *pl->h_mid_abort == 0) return OCTANE_OK;
    *pl->h_mid_abort = 0;
    (void)hipMemset(static_cast<char *>(pl->d_mid) + 8, 0, 4);
    g_last_error = "a persistent PCG solve could not get all its workgroups resident within 0.25 s (another process with the same kind of "
                   "kernel on this GPU?): the result of this run is not valid; OCTANE_TUNE_PERSIST=0 selects one launch per iteration";
    return OCTANE_E_HIP;
}

// The row-band paths (vof_tiled.hip) run plan_level_solve themselves, not run_on_stream: they clear the word at the start of a
// solve and fetch it at its end with these two, on the band's stream.
int octane::persist_begin_run(octane_vof_plan *pl, hipStream_t s)
{
    if (pl->d_mid) HIP_TRY(hipMemsetAsync(static_cast<char *>(pl->d_mid) + 8, 0, sizeof(unsigned), s));
    return OCTANE_OK;
}
int octane::persist_end_run(octane_vof_plan *pl, hipStream_t s)
{
    if (pl->d_mid) HIP_TRY(hipMemcpyAsync(pl->h_mid_abort, static_cast<char *>(pl->d_mid) + 8, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return OCTANE_OK;
}

static int run_on_stream(octane_vof_plan *pl, hipStream_t s);
constexpr int kPersistOffRuns = 16;        // runs a plan makes with one launch per iteration after an abandoned persistent solve

static void persist_switch_off_for_a_while(octane_vof_plan *pl)
{
    pl->use_persist = 0;
    pl->persist_off_runs = kPersistOffRuns;
    pl->persist_abandoned_total++;
    if (pl->graph_exec) { (void)hipGraphExecDestroy(pl->graph_exec); pl->graph_exec = nullptr; }
    static std::atomic<int> said{0};
    if (said.fetch_add(1) == 0)
        fprintf(stderr, "octane: a persistent PCG solve was abandoned (GPU shared with another process?); solving the pair again with one launch per "
                        "iteration, and so for this plan's next %d runs (OCTANE_TUNE_PERSIST=0 selects that from the start)\n", kPersistOffRuns);
}

// Device-buffer runs are asynchronous: nobody looks at the abort word until the caller synchronises and asks.  The first call that does
// (octane_vof_plan_wait, octane_vof_plan_last_iterations) finds it raised, and -- VERDICT r3 item 8 -- repairs the run as the host
// path does: the inputs of that run are still in the plan's own planes (the run only reads them), so the pyramid is made again on the
// run's stream with one launch per iteration and its flow copied into the output buffers that run was given (they must still be the
// caller's: they are what it is waiting for).  Returns OCTANE_OK when there was nothing to repair or the repair succeeded.
static int heal_abandoned_run(octane_vof_plan *pl)
{
    if (!pl->h_mid_abort || *pl->h_mid_abort == 0) return OCTANE_OK;
    if (pl->last_mem != OCTANE_MEM_DEVICE || !pl->last_u || !pl->last_v || !pl->use_persist) return persist_check(pl);
    (void)persist_check(pl);                       // clears the word (and leaves the explanation in octane_last_error)
    persist_switch_off_for_a_while(pl);
    hipStream_t s = pl->last_stream;
    const int cur = run_on_stream(pl, s);
    if (cur < 0) return cur;
    launch_copy2d(s, pl->U[cur], pl->pitch0, pl->last_u, pl->nx, pl->nx, pl->ny);
    launch_copy2d(s, pl->V[cur], pl->pitch0, pl->last_v, pl->nx, pl->nx, pl->ny);
    HIP_TRY(hipStreamSynchronize(s));
    return persist_check(pl);                      // one launch per iteration cannot be abandoned; anything else is an error
}

extern "C" int octane_vof_plan_wait(octane_vof_plan *pl)
{
    if (!pl) return OCTANE_E_INVALID;
    HIP_TRY(hipSetDevice(pl->device));
    HIP_TRY(hipStreamSynchronize(pl->own_stream));
    // the stream of the last device-buffer run, the NULL stream included (what callers on PyTorch's default stream pass: own_stream is
    // non-blocking, so draining it says nothing about the NULL stream -- ADVICE r4)
    if (pl->last_mem == OCTANE_MEM_DEVICE && pl->last_stream != pl->own_stream) HIP_TRY(hipStreamSynchronize(pl->last_stream));
    return heal_abandoned_run(pl);
}

extern "C" long long octane_vof_plan_last_iterations(octane_vof_plan *pl)
{
    if (!pl) return -1;
    // the abort word and the count belong to a run that may still be in flight on the caller's stream: wait for it before reading them
    if (pl->last_mem == OCTANE_MEM_DEVICE) { (void)hipSetDevice(pl->device); if (hipStreamSynchronize(pl->last_stream) != hipSuccess) return -1; }
    if (pl->h_mid_abort && *pl->h_mid_abort != 0) {
        // the run's persistent solve was abandoned: make the run again (device buffers; see heal_abandoned_run) -- -2 only if that fails
        (void)hipSetDevice(pl->device);
        if (heal_abandoned_run(pl) != OCTANE_OK) return -2;
    }
    return *pl->h_iters;
}

extern "C" int octane_vof_plan_persist_state(const octane_vof_plan *pl, int *abandoned_total)
{
    if (!pl) return OCTANE_E_INVALID;
    if (abandoned_total) *abandoned_total = pl->persist_abandoned_total;
    return pl->persist_off_runs > 0 ? -pl->persist_off_runs : (pl->use_persist ? 1 : 0);
}

int octane::plan_load_inputs(octane_vof_plan *pl, const float *img1, const float *img2, const float *u, const float *v,
                             int mem, hipStream_t s)
{
    const int nx = pl->nx, ny = pl->ny, nc = pl->nc, p0 = pl->pitch0;
    const size_t dense_row = (size_t)nx * sizeof(float), pitched_row = (size_t)p0 * sizeof(float);
    if (mem == OCTANE_MEM_HOST) {   // ref .cu:1330-1352 (element-wise fills of managed memory there)
        for (int c = 0; c < nc; c++) {
            HIP_TRY(hipMemcpy2DAsync(pl->img1p + c * pl->plane0, pitched_row, img1 + (size_t)c * nx * ny, dense_row, dense_row, ny, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpy2DAsync(pl->img2p + c * pl->plane0, pitched_row, img2 + (size_t)c * nx * ny, dense_row, dense_row, ny, hipMemcpyHostToDevice, s));
        }
        if (u) {
            HIP_TRY(hipMemcpy2DAsync(pl->uh, pitched_row, u, dense_row, dense_row, ny, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpy2DAsync(pl->vh, pitched_row, v, dense_row, dense_row, ny, hipMemcpyHostToDevice, s));
        }
    } else {
        for (int c = 0; c < nc; c++) {
            launch_copy2d(s, img1 + (size_t)c * nx * ny, nx, pl->img1p + c * pl->plane0, p0, nx, ny);
            launch_copy2d(s, img2 + (size_t)c * nx * ny, nx, pl->img2p + c * pl->plane0, p0, nx, ny);
        }
        if (u) {
            launch_copy2d(s, u, nx, pl->uh, p0, nx, ny);
            launch_copy2d(s, v, nx, pl->vh, p0, nx, ny);
        }
    }
    if (!u) {       // no first guess: zero flow, as oct_optical_flow does without -firstguess (ref oct_optical_flow.cc:38-48)
        HIP_TRY(hipMemsetAsync(pl->uh, 0, pl->plane0 * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(pl->vh, 0, pl->plane0 * sizeof(float), s));
    }
    return OCTANE_OK;
}

extern "C" int octane_vof_plan_run(octane_vof_plan *pl, const float *img1, const float *img2,
                                   float *u, float *v, int mem, void *hip_stream)
{
    if (!u || !v) { g_last_error = "octane_vof_plan_run: invalid argument"; return OCTANE_E_INVALID; }
    return octane_vof_plan_solve(pl, img1, img2, u, v, u, v, mem, hip_stream);
}

static int plan_solve_attempt(octane_vof_plan *pl, const float *img1, const float *img2, const float *u0, const float *v0,
                              float *u, float *v, int mem, void *hip_stream, int attempt);

extern "C" int octane_vof_plan_solve(octane_vof_plan *pl, const float *img1, const float *img2, const float *u0, const float *v0,
                                     float *u, float *v, int mem, void *hip_stream)
{
    return plan_solve_attempt(pl, img1, img2, u0, v0, u, v, mem, hip_stream, 0);
}

static int plan_solve_attempt(octane_vof_plan *pl, const float *img1, const float *img2, const float *u0, const float *v0,
                              float *u, float *v, int mem, void *hip_stream, int attempt)
{
    if (!pl || !img1 || !img2 || !u || !v || ((u0 == nullptr) != (v0 == nullptr)) || (mem != OCTANE_MEM_HOST && mem != OCTANE_MEM_DEVICE)) {
        g_last_error = "octane_vof_plan_solve: invalid argument";
        return OCTANE_E_INVALID;
    }
    HIP_TRY(hipSetDevice(pl->device));
    // Device buffers: the work is ordered on exactly the stream the caller names (NULL is HIP's
    // null stream, which is what PyTorch's default stream is).  Host buffers: the call blocks
    // anyway, so NULL selects the plan's private stream.
    hipStream_t s = (mem == OCTANE_MEM_DEVICE || hip_stream) ? (hipStream_t)hip_stream : pl->own_stream;
    if (hip_stream == OCTANE_STREAM_OWN) s = pl->own_stream;
    if (attempt == 0 && pl->persist_off_runs > 0 && --pl->persist_off_runs == 0) pl->use_persist = 1;     // the co-tenant may be gone: try again
    const int nx = pl->nx, ny = pl->ny, p0 = pl->pitch0;
    const size_t dense_row = (size_t)nx * sizeof(float), pitched_row = (size_t)p0 * sizeof(float);
    {
        const int rc = plan_load_inputs(pl, img1, img2, u0, v0, mem, s);
        if (rc) return rc;
    }
    // The launch sequence of a pyramid is fixed for a plan (every pointer and size is the plan's own), so it is
    // captured once into a hipGraph and replayed: the ~4500 launches of a pyramid then cost a kernel boundary
    // each (~1.5 us) instead of a host launch (~3.5 us), which is what the coarse levels' 5-8 us kernels were
    // waiting on.  Debug taps and per-kernel profiling need host work between launches and use the eager path.
    int cur;
    if (pl->use_graph && !pl->trace && !pl->profiling) {
        if (!pl->graph_exec) {
            hipGraph_t graph = nullptr;
            // captured on the plan's private stream (the caller's may be the null stream, which cannot
            // capture); the instantiated graph is then launched into whatever stream the caller names
            hipStream_t cs = pl->own_stream;
            HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            const int c = run_on_stream(pl, cs);
            hipError_t e = hipStreamEndCapture(cs, &graph);
            if (c < 0 || e != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                if (c >= 0) g_last_error = std::string("hipStreamEndCapture: ") + hipGetErrorString(e);
                return c < 0 ? c : OCTANE_E_HIP;
            }
            e = hipGraphInstantiate(&pl->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) {
                pl->graph_exec = nullptr;
                g_last_error = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
                return OCTANE_E_HIP;
            }
            pl->graph_cur = c;
        }
        HIP_TRY(hipGraphLaunch(pl->graph_exec, s));
        cur = pl->graph_cur;
    } else {
        cur = run_on_stream(pl, s);
        if (cur < 0) return cur;
    }
    if (mem == OCTANE_MEM_HOST) {   // ref .cu:1432-1438
        HIP_TRY(hipStreamSynchronize(s));             // the abort word of this run is on the host now
        if (int rc = persist_check(pl)) {
            // A persistent solve gave up (its workgroups did not all become resident in time: another process with the same kind
            // of kernel on this GPU, or a pre-empted queue).  The caller's buffers are still intact (u / v, which may be the first
            // guess, are written below only after this check), so solve the pair again with one launch per iteration, once; the
            // plan stays that way.
            if (!pl->use_persist || attempt > 0) return rc;
            persist_switch_off_for_a_while(pl);
            return plan_solve_attempt(pl, img1, img2, u0, v0, u, v, mem, hip_stream, 1);
        }
        HIP_TRY(hipMemcpy2DAsync(u, dense_row, pl->U[cur], pitched_row, dense_row, ny, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpy2DAsync(v, dense_row, pl->V[cur], pitched_row, dense_row, ny, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    } else {
        launch_copy2d(s, pl->U[cur], p0, u, nx, nx, ny);
        launch_copy2d(s, pl->V[cur], p0, v, nx, nx, ny);
    }
    pl->last_mem = mem; pl->last_u = u; pl->last_v = v; pl->last_stream = s;
    HIP_TRY(hipGetLastError());
    return OCTANE_OK;
}

// One-shot entry.  The reference allocates and frees all device state per call (.cu:1268-1472); here the plan of the
// last call is kept and reused when the next call has the same shape and parameters (a time series through the same
// host program), because creating and destroying a multi-GB arena costs as much as 0.5 s at 10848^2.
// OCTANE_VOF_CACHE=0 restores allocate-per-call; octane_vof_release_cache() frees the kept plan.
static std::mutex g_cache_mu;
static octane_vof_plan *g_cache_plan = nullptr;
static octane_vof_params g_cache_prm;
static int g_cache_nx = 0, g_cache_ny = 0, g_cache_nc = 0;
static bool g_cache_placed = false;     // the kept plan has had its placement trials (from its first reuse on)

static bool same_params(const octane_vof_params &a, const octane_vof_params &b)
{
    return a.alpha == b.alpha && a.lambda == b.lambda && a.lambdac == b.lambdac && a.scaleF == b.scaleF && a.scsig == b.scsig &&
           a.kiters == b.kiters && a.liters == b.liters && a.cgiters == b.cgiters && a.dozim == b.dozim && a.device == b.device;
}

extern "C" void octane_vof_release_cache(void)
{
    std::lock_guard<std::mutex> g(g_cache_mu);
    if (g_cache_plan) { octane_vof_plan_destroy(g_cache_plan); g_cache_plan = nullptr; }
}

extern "C" int octane_vof_run(const float *img1, const float *img2, int nx, int ny, int nchan,
                              float *u, float *v, const octane_vof_params *p)
{
    if (!u || !v) { g_last_error = "octane_vof_run: invalid argument"; return OCTANE_E_INVALID; }
    return octane_vof_solve(img1, img2, nx, ny, nchan, u, v, u, v, p);
}

// The same one-shot call with the first guess apart from the result; u0 = v0 = NULL is the zero first guess of a run without
// -firstguess (ref src/oct_optical_flow.cc:38-48), which is then not uploaded: two of the call's six PCIe transfers (3.5 of 10.5 ms at
// 5000^2; the transfers run at the link's 57 GB/s each way and nothing of them can hide behind the solve, which needs all inputs at
// its coarsest level and has both outputs only after its finest: profiles/r4_h2d_pitch.txt).
extern "C" int octane_vof_solve(const float *img1, const float *img2, int nx, int ny, int nchan, const float *u0, const float *v0,
                                float *u, float *v, const octane_vof_params *p)
{
    if (!p) { g_last_error = "octane_vof_solve: invalid argument"; return OCTANE_E_INVALID; }
    const char *e = getenv("OCTANE_VOF_CACHE");
    const bool use_cache = !(e && atoi(e) == 0);
    // A kept plan lives for many pairs (a time series through one host program), so it gets the placement trials a plan created through
    // octane_vof_plan_create gets (up to eight candidate arenas timed, the fastest kept: worth up to 11 % per pair, ~0.3 s once at
    // 5000^2) -- but only once it IS reused: the first call of a process takes the first arena it gets (a single-pair `octane` run
    // would pay the trials and never recover them -- ADVICE r4), the second call with the same geometry re-creates the plan with the
    // trials, every later one finds it placed.  Only the allocate-per-call form (OCTANE_VOF_CACHE=0, or a second thread while the kept
    // plan is busy) never runs them.
    if (use_cache && g_cache_mu.try_lock()) {
        std::lock_guard<std::mutex> g(g_cache_mu, std::adopt_lock);
        if (g_cache_plan && !(g_cache_nx == nx && g_cache_ny == ny && g_cache_nc == nchan && same_params(g_cache_prm, *p))) {
            octane_vof_plan_destroy(g_cache_plan);
            g_cache_plan = nullptr;
        }
        if (g_cache_plan && !g_cache_placed) {      // first reuse: now the trials pay
            octane_vof_plan_destroy(g_cache_plan);
            g_cache_plan = nullptr;
            const int rc = plan_create_ex(&g_cache_plan, nx, ny, nchan, p, 8);
            if (rc != OCTANE_OK) { g_cache_plan = nullptr; return rc; }
            g_cache_placed = true;
        }
        if (!g_cache_plan) {
            const int rc = plan_create_ex(&g_cache_plan, nx, ny, nchan, p, 1);
            if (rc != OCTANE_OK) { g_cache_plan = nullptr; return rc; }
            g_cache_prm = *p; g_cache_nx = nx; g_cache_ny = ny; g_cache_nc = nchan;
            g_cache_placed = false;
        }
        const int rc = octane_vof_plan_solve(g_cache_plan, img1, img2, u0, v0, u, v, OCTANE_MEM_HOST, nullptr);
        if (rc != OCTANE_OK) { octane_vof_plan_destroy(g_cache_plan); g_cache_plan = nullptr; }   // do not keep a plan that failed
        return rc;
    }
    // cache disabled, or another thread is inside the cached plan: a private plan for this call
    octane_vof_plan *pl = nullptr;
    int rc = plan_create_ex(&pl, nx, ny, nchan, p, 1);
    if (rc != OCTANE_OK) return rc;
    rc = octane_vof_plan_solve(pl, img1, img2, u0, v0, u, v, OCTANE_MEM_HOST, nullptr);
    octane_vof_plan_destroy(pl);
    return rc;
}

extern "C" int octane_vof_batch_run(int npairs, const float *const *img1, const float *const *img2,
                                    int nx, int ny, int nchan, float *const *u, float *const *v,
                                    const octane_vof_params *p, int ndevices, const int *devices)
{
    if (npairs < 0 || !img1 || !img2 || !u || !v || !p || ndevices < 1) {
        g_last_error = "octane_vof_batch_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    // Frames up to ~8 Mpixel leave the GPU latency-bound on their coarse levels, so each device gets several lanes
    // (plans on their private streams -- distinct hardware queues --, one host thread each) whose kernels interleave.
    // Round 1 (one launch per PCG iteration on every level): three or four lanes, as many as the runtime has hardware queues for
    // (GPU_MAX_HW_QUEUES, 4 by default: 147 / 161 / 145 Mpix/s with 2 / 3 / 4 lanes at 2000^2; 166 / 171 / 160 with 3 / 4 / 6 and 8
    // queues).  Round 2: the mid-size levels of a plan are persistent solves, which are serialised per device but short, and TWO
    // lanes with those solves uncapped beat every other combination (64 x 2000^2, same box: 2 lanes 188.6, 3 lanes 182.4, 4 lanes
    // 177.3; with the round-1 arrangement -- persistent solves only on the tiny levels -- 155 / 176 / 181 for 2 / 3 / 4 lanes).
    // Beyond 64 Mpixel one lane.
    const long px = (long)nx * ny;
    int lanes = px <= (64L << 20) ? 2 : 1;
    if (const char *e = tune_env("OCTANE_TUNE_BATCH_LANES")) lanes = atoi(e) > 0 ? atoi(e) : lanes;
    const int nworkers = ndevices * lanes;
    std::vector<int> rcs(nworkers, OCTANE_OK);
    std::vector<std::string> errs(nworkers);
    std::vector<std::thread> workers;
    for (int wk = 0; wk < nworkers; wk++) {
        workers.emplace_back([&, wk]() {
            const int d = wk % ndevices, lane = wk / ndevices;
            octane_vof_params prm = *p;
            prm.device = devices ? devices[d] : d;
            octane_vof_plan *pl = nullptr;
            // pair b belongs to device b % ndevices; a device's pairs alternate between its lanes
            const int first = d + ndevices * lane, step = ndevices * lanes;
            if (first >= npairs) return;
            int rc = octane_vof_plan_create(&pl, nx, ny, nchan, &prm);
            // more than two lanes (OCTANE_TUNE_BATCH_LANES): a persistent solve holds its CUs for a whole solve and such launches are
            // serialised per device, so only the tiny levels keep it then (plan_lane_mode)
            if (rc == OCTANE_OK && lanes > 2 && !getenv("OCTANE_TUNE_PERSIST_MAXG")) plan_lane_mode(pl);
            if (rc == OCTANE_OK && lanes == 2 && !getenv("OCTANE_TUNE_PERSIST_MAXG")) plan_lane_pair_mode(pl);      // round 5: concurrent, each on half the CUs
            for (int b = first; rc == OCTANE_OK && b < npairs; b += step)
                rc = octane_vof_plan_run(pl, img1[b], img2[b], u[b], v[b], OCTANE_MEM_HOST, nullptr);
            if (rc != OCTANE_OK) errs[wk] = g_last_error;
            rcs[wk] = rc;
            octane_vof_plan_destroy(pl);
        });
    }
    for (auto &t : workers) t.join();
    for (int wk = 0; wk < nworkers; wk++)
        if (rcs[wk] != OCTANE_OK) { g_last_error = errs[wk]; return rcs[wk]; }
    return OCTANE_OK;
}

// ---------------------------------------------------------------------------------------------
// pix2uv (ref src/oct_pix2uv_cuda.cu:265-370)
// ---------------------------------------------------------------------------------------------
extern "C" int octane_pix2uv_run(const octane_nav *nav, double t1, double t2, const float *u, const float *v,
                                 int pixuv, int mode, short *ur, short *vr, short *ur2, short *vr2,
                                 float *dT, int *sector_moved, int device)
{
    // which build of the kernel (pix2uv_kernel.hip): 0 strict, 1 every multiply-add the compiler may fuse, 2 the two float sites only
    int build = (mode & OCTANE_NAV_FMAD) ? 1 : (mode & OCTANE_NAV_FMAD_FLOAT) ? 2 : 0;
    mode &= ~(OCTANE_NAV_FMAD | OCTANE_NAV_FMAD_FLOAT);
    if (const char *e = getenv("OCTANE_PIX2UV_FMAD")) build = (atoi(e) >= 0 && atoi(e) <= 2) ? atoi(e) : build;
    if (!nav || !u || !v || !ur || !vr || nav->nx < 1 || nav->ny < 1 || mode < 0 || mode > 2 ||
        (pixuv == 0 && (!ur2 || !vr2))) {
        g_last_error = "octane_pix2uv_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    const long n = (long)nav->nx * nav->ny;
    if (dT) *dT = (float)(t2 - t1);
    if (sector_moved) *sector_moved = 0;
    const float dx = nav->xOffset - nav->g2xOffset, dy = nav->yOffset - nav->g2yOffset;
    const bool same = (((double)(dx * dx) < (0.00001 * 0.00001)) && ((double)(dy * dy) < (0.00001 * 0.00001)));   // ref p2u:295
    if (!same) {                      // ref p2u:358-368
        for (long k = 0; k < n; k++) { ur[k] = 0; vr[k] = 0; }
        if (ur2 && vr2) for (long k = 0; k < n; k++) { ur2[k] = 0; vr2[k] = 0; }
        if (sector_moved) *sector_moved = 1;
        return OCTANE_OK;
    }
    if (pixuv != 0) {                 // ref p2u:348-356: no navigation, host-only in the reference too
        for (long k = 0; k < n; k++) { ur[k] = (short)(100 * u[k]); vr[k] = (short)(100 * v[k]); }
        return OCTANE_OK;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    if (device > ndev - 1 || device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    float *du = nullptr, *dv = nullptr;
    short *dout = nullptr;
    hipStream_t s = nullptr;
    int rc = OCTANE_OK;
    do {
        if (hipMalloc((void **)&du, n * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&dv, n * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&dout, 4 * n * sizeof(short)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipStreamCreate(&s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(du, u, n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(dv, v, n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        NavArgs a;
        a.pph = nav->pph; a.req = nav->req; a.rpol = nav->rpol; a.lam0 = nav->lam0;
        a.xScale = nav->xScale; a.xOffset = nav->xOffset; a.yScale = nav->yScale; a.yOffset = nav->yOffset;
        a.lat1 = nav->lat1; a.lon1 = nav->lon1; a.lon0 = nav->lon0; a.R = nav->R;
        a.minX = nav->minX; a.minY = nav->minY; a.nx = nav->nx; a.ny = nav->ny;
        (build == 1 ? launch_pix2uv_fmad : build == 2 ? launch_pix2uv_fsites : launch_pix2uv)(s, a, t1, t2, du, dv, mode, dout, dout + n, dout + 2 * n, dout + 3 * n, n);
        if (hipGetLastError() != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(ur, dout, n * sizeof(short), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(vr, dout + n, n * sizeof(short), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(ur2, dout + 2 * n, n * sizeof(short), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(vr2, dout + 3 * n, n * sizeof(short), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
    } while (0);
    if (rc != OCTANE_OK) g_last_error = "octane_pix2uv_run: HIP failure";
    if (s) (void)hipStreamDestroy(s);
    if (du) (void)hipFree(du);
    if (dv) (void)hipFree(dv);
    if (dout) (void)hipFree(dout);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// navcal (ref src/oct_navcal_cuda.cu:100-207) and the band range table (ref src/oct_normalize_geo.cc:9-88)
// ---------------------------------------------------------------------------------------------
extern "C" int octane_bandminmax(int band, float *maxch, float *minch)
{
    // {max, min} radiance per ABI band 1..16; bands 7 and 8 carry the reference's "meteorological" ranges
    static const float tab[16][2] = {
        {804.03605737f, -25.93664701f}, {628.98723908f, -20.28991094f}, {373.16695681f, -12.03764377f},
        {140.19342584f, -4.52236858f},  {94.84802665f, -3.05961376f},   {29.78947040f, -0.96095066f},
        {2.f, 0.f},                     {6.f, 3.f},                     {44.998f, -0.2472f},
        {79.831f, -0.2871f},            {134.93f, -0.3909f},            {108.44f, -0.4617f},
        {185.5699f, -1.6443f},          {198.71f, -0.5154f},            {212.28f, -0.5262f},
        {170.19f, -1.5726f}};
    if (band < 1 || band > 16 || !maxch || !minch) return OCTANE_E_INVALID;
    *maxch = tab[band - 1][0];
    *minch = tab[band - 1][1];
    return OCTANE_OK;
}

extern "C" int octane_navcal_run(const short *data2, const short *x, const short *y, int nx, int ny,
                                 const octane_navcal_params *p, float *data3, float *lat, float *lon,
                                 short *data2s, short *xs, short *ys, int device)
{
    if (!data2 || !x || !y || !p || !data3 || !lat || !lon || !data2s || !xs || !ys || nx < 1 || ny < 1 ||
        p->minx < 0 || p->miny < 0 || p->maxx > nx || p->maxy > ny || p->maxx <= p->minx || p->maxy <= p->miny ||
        p->cal < 0 || p->cal > 3) {
        g_last_error = "octane_navcal_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    if (device > ndev - 1 || device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    const long n = (long)nx * ny, n2 = (long)(p->maxx - p->minx) * (p->maxy - p->miny);
    for (int i = p->minx; i < p->maxx; i++) xs[i - p->minx] = x[i];       // ref nav:155-162
    for (int j = p->miny; j < p->maxy; j++) ys[j - p->miny] = y[j];
    NavcalArgs A;
    A.xScale = p->xScale; A.xOffset = p->xOffset; A.yScale = p->yScale; A.yOffset = p->yOffset;
    A.radScale = p->radScale; A.radOffset = p->radOffset; A.rpol = p->rpol; A.req = p->req; A.H = p->H; A.lam0 = p->lam0;
    A.fk1 = p->fk1; A.fk2 = p->fk2; A.bc1 = p->bc1; A.bc2 = p->bc2; A.kap1 = p->kap1;
    A.maxin = p->maxin; A.minin = p->minin; A.maxout = p->maxout; A.minout = p->minout;
    A.subpoint_slope = (float)(1. / (0.021 - 0.0212));                     // ref nav:168-169
    A.subpoint_int = (float)(1. - 0.021 * (double)A.subpoint_slope);
    A.cal = p->cal; A.donav = p->donav; A.nx = nx; A.ny = ny;
    A.minx = p->minx; A.maxx = p->maxx; A.miny = p->miny; A.maxy = p->maxy;
    short *d_in = nullptr, *d_xy = nullptr, *d_s = nullptr;
    float *d_out = nullptr;
    hipStream_t s = nullptr;
    int rc = OCTANE_OK;
    do {
        if (hipMalloc((void **)&d_in, n * sizeof(short)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&d_xy, (size_t)(nx + ny) * sizeof(short)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&d_s, n2 * sizeof(short)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&d_out, 3 * n2 * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipStreamCreate(&s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d_in, data2, n * sizeof(short), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d_xy, x, nx * sizeof(short), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d_xy + nx, y, ny * sizeof(short), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        launch_navcal(s, A, d_xy, d_xy + nx, d_in, d_out, d_out + n2, d_out + 2 * n2, d_s);
        if (hipGetLastError() != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(data3, d_out, n2 * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(lat, d_out + n2, n2 * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(lon, d_out + 2 * n2, n2 * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(data2s, d_s, n2 * sizeof(short), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
    } while (0);
    if (rc != OCTANE_OK) g_last_error = "octane_navcal_run: HIP failure";
    if (s) (void)hipStreamDestroy(s);
    if (d_in) (void)hipFree(d_in);
    if (d_xy) (void)hipFree(d_xy);
    if (d_s) (void)hipFree(d_s);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

extern "C" int octane_proj_navcal_run(const float *data2, const short *x, const short *y, int nx, int ny,
                                      const octane_proj_navcal_params *p, float *data3, float *lat, float *lon,
                                      short *data2s, short *xs, short *ys, int device)
{
    if (!data2 || !x || !y || !p || !data3 || !lat || !lon || !data2s || !xs || !ys || nx < 1 || ny < 1 ||
        p->minx < 0 || p->miny < 0 || p->maxx > nx || p->maxy > ny || p->maxx <= p->minx || p->maxy <= p->miny ||
        (p->mode != OCTANE_NAV_POLAR && p->mode != OCTANE_NAV_MERC)) {
        g_last_error = "octane_proj_navcal_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    if (device > ndev - 1 || device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    const long n = (long)nx * ny, n2 = (long)(p->maxx - p->minx) * (p->maxy - p->miny);
    for (int i = p->minx; i < p->maxx; i++) xs[i - p->minx] = x[i];       // ref pnav:122-129, mnav:104-111
    for (int j = p->miny; j < p->maxy; j++) ys[j - p->miny] = y[j];
    for (long k = 0; k < n2; k++) data2s[k] = 0;                          // ref pnav:119, mnav:101
    const double PI = 3.14159265359, DTOR = PI / 180.;
    ProjNavcalArgs A;
    A.xScale = p->xScale; A.xOffset = p->xOffset; A.yScale = p->yScale; A.yOffset = p->yOffset; A.R = p->R;
    A.lon0 = (float)(p->lon0 * DTOR);                                     // ref pnav:138, mnav:119: degrees -> radians, as float arguments
    A.lat1 = (float)(p->lat1 * DTOR);
    A.donav = p->donav; A.mode = p->mode; A.nx = nx; A.ny = ny;
    A.minx = p->minx; A.maxx = p->maxx; A.miny = p->miny; A.maxy = p->maxy;
    float *d_in = nullptr, *d_out = nullptr;
    short *d_xy = nullptr;
    hipStream_t s = nullptr;
    int rc = OCTANE_OK;
    do {
        if (hipMalloc((void **)&d_in, n * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&d_xy, (size_t)(nx + ny) * sizeof(short)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&d_out, 3 * n2 * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipStreamCreate(&s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d_in, data2, n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d_xy, x, nx * sizeof(short), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d_xy + nx, y, ny * sizeof(short), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        launch_proj_navcal(s, A, d_xy, d_xy + nx, d_in, d_out, d_out + n2, d_out + 2 * n2);
        if (hipGetLastError() != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(data3, d_out, n2 * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(lat, d_out + n2, n2 * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(lon, d_out + 2 * n2, n2 * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
    } while (0);
    if (rc != OCTANE_OK) g_last_error = "octane_proj_navcal_run: HIP failure";
    if (s) (void)hipStreamDestroy(s);
    if (d_in) (void)hipFree(d_in);
    if (d_xy) (void)hipFree(d_xy);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// patch matching, -sosm (ref src/oct_patch_match_optical_flow.cc:56-156)
// ---------------------------------------------------------------------------------------------
extern "C" int octane_sosm_run(const float *img1, const float *img2, int nx, int ny, float *u, float *v, int rad, int srad, int device)
{
    if (!img1 || !img2 || !u || !v || nx < 1 || ny < 1 || rad < 0 || srad < 0 || rad > 16 || srad > 16) {
        g_last_error = "octane_sosm_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    if (device > ndev - 1 || device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    // The spiral of ref pm:107-137 as a table.  The window test `(-SXD2 < n <= SXD2)` is a chained comparison
    // (bool <= int) in the reference and is evaluated as written.
    const int SX = 2 * srad + 1, SY = 2 * srad + 1, SXD2 = SX / 2, SYD2 = SY / 2;
    std::vector<int> tab(2 * (size_t)SX * SY + (size_t)SX * SY, -1);
    int count = 0;
    {
        int n = 0, m = 0, dn = 0, dm = -1;
        for (int ic = 0; ic < SX * SY; ic++) {
            if (((-SXD2 < n) <= SXD2) && ((-SYD2 < m) <= SYD2)) { tab[2 * count] = n; tab[2 * count + 1] = m; count++; }
            if ((n == m) || ((n < 0) && (n == -m)) || ((n > 0) && (n == 1 - m))) { const int odn = dn; dn = -dm; dm = odn; }
            n += dn; m += dm;
        }
    }
    std::vector<int> dev_tab(2 * (size_t)count + (size_t)SX * SY, -1);
    for (int c = 0; c < count; c++) {
        dev_tab[2 * c] = tab[2 * c]; dev_tab[2 * c + 1] = tab[2 * c + 1];
        const int n = tab[2 * c], m = tab[2 * c + 1];
        if (n >= -srad && n <= srad && m >= -srad && m <= srad) {
            int &slot = dev_tab[2 * count + (n + srad) * SY + (m + srad)];
            if (slot < 0) slot = c;                                         // first visit decides ties
        }
    }
    const size_t n = (size_t)nx * ny;
    float *d = nullptr;
    int *d_tab = nullptr;
    hipStream_t s = nullptr;
    int rc = OCTANE_OK;
    do {
        if (hipMalloc((void **)&d, 4 * n * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&d_tab, dev_tab.size() * sizeof(int)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipStreamCreate(&s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        const float *src[4] = {img1, img2, u, v};
        bool ok = true;
        for (int i = 0; i < 4 && ok; i++) ok = hipMemcpyAsync(d + i * n, src[i], n * sizeof(float), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(d_tab, dev_tab.data(), dev_tab.size() * sizeof(int), hipMemcpyHostToDevice, s) == hipSuccess;
        if (!ok) { rc = OCTANE_E_HIP; break; }
        launch_sosm(s, d, d + n, d + 2 * n, d + 3 * n, nx, ny, rad, srad, d_tab, count);
        if (hipGetLastError() != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(u, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(v, d + 3 * n, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
    } while (0);
    if (rc != OCTANE_OK) g_last_error = "octane_sosm_run: HIP failure";
    if (s) (void)hipStreamDestroy(s);
    if (d) (void)hipFree(d);
    if (d_tab) (void)hipFree(d_tab);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// uv2pix (ref src/oct_pix2uv_cuda.cu:372-476) and srsal (ref src/oct_srsal_cuda.cu:73-147)
// ---------------------------------------------------------------------------------------------
extern "C" int octane_uv2pix_run(const octane_nav *nav, double t1, double t2, float *u, float *v,
                                 const float *lat, const float *lon, const short *gx, const short *gy, int device)
{
    if (!nav || !u || !v || !lat || !lon || !gx || !gy || nav->nx < 1 || nav->ny < 1) {
        g_last_error = "octane_uv2pix_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    const long n = (long)nav->nx * nav->ny;
    if (!((nav->xOffset == nav->g2xOffset) && (nav->yOffset == nav->g2yOffset))) {   // ref p2u:421,457-467
        for (long k = 0; k < n; k++) { u[k] = 0.f; v[k] = 0.f; }
        return OCTANE_OK;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    if (device > ndev - 1 || device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    Uv2pixArgs A;
    A.secs = t2 - t1; A.req = nav->req; A.rpol = nav->rpol; A.lam0 = nav->lam0; A.pph = nav->pph;
    A.req2 = nav->req * nav->req; A.rpol2 = nav->rpol * nav->rpol;
    const double e = std::sqrt((A.req2 - A.rpol2) / (A.req2));           // ref p2u:417-419,428
    A.eval = e * e;
    A.xscale = nav->xScale; A.xoffset = nav->xOffset; A.yscale = nav->yScale; A.yoffset = nav->yOffset;
    A.nx = nav->nx; A.ny = nav->ny;
    float *d = nullptr;
    short *dxy = nullptr;
    hipStream_t s = nullptr;
    int rc = OCTANE_OK;
    do {
        if (hipMalloc((void **)&d, 6 * n * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipMalloc((void **)&dxy, (size_t)(nav->nx + nav->ny) * sizeof(short)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipStreamCreate(&s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        const float *src[4] = {u, v, lat, lon};
        bool ok = true;
        for (int i = 0; i < 4 && ok; i++) ok = hipMemcpyAsync(d + i * n, src[i], n * sizeof(float), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(dxy, gx, nav->nx * sizeof(short), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(dxy + nav->nx, gy, nav->ny * sizeof(short), hipMemcpyHostToDevice, s) == hipSuccess;
        if (!ok) { rc = OCTANE_E_HIP; break; }
        launch_uv2pix(s, A, d, d + n, d + 2 * n, d + 3 * n, dxy, dxy + nav->nx, d + 4 * n, d + 5 * n);
        if (hipGetLastError() != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(u, d + 4 * n, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(v, d + 5 * n, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
    } while (0);
    if (rc != OCTANE_OK) g_last_error = "octane_uv2pix_run: HIP failure";
    if (s) (void)hipStreamDestroy(s);
    if (d) (void)hipFree(d);
    if (dxy) (void)hipFree(dxy);
    return rc;
}

extern "C" int octane_srsal_run(float *u, float *v, const float *cth, int nx, int ny, int device)
{
    if (!u || !v || !cth || nx < 1 || ny < 1) {
        g_last_error = "octane_srsal_run: invalid argument";
        return OCTANE_E_INVALID;
    }
    int ndev = octane_device_count();
    if (ndev == 0) { g_last_error = "No gpus available for use"; return OCTANE_E_NODEVICE; }
    if (device > ndev - 1 || device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    SrsalArgs A;
    {   // ref srsal:75-82 with oct_getGaussian_1D (ref src/oct_gaussian.cc:34-47): 37 taps, sigma 9, normalised
        const double sigpix = 20.;
        A.sigpix2 = -1. / (sigpix * sigpix * 2.);
        const double sigma = 9;
        const int wk2 = 18;
        const double sg = 2.0 * sigma * sigma;
        double sum = 0.0;
        for (int x = -wk2; x <= wk2; x++) {
            const double r = x;
            A.gk[x + wk2] = (std::exp(-(r * r) / sg)) / (M_PI * sg);
            sum += A.gk[x + wk2];
        }
        for (int i = 0; i < 37; ++i) A.gk[i] /= sum;
    }
    const long n = (long)nx * ny;
    float *d = nullptr;
    hipStream_t s = nullptr;
    int rc = OCTANE_OK;
    do {
        if (hipMalloc((void **)&d, 5 * n * sizeof(float)) != hipSuccess) { rc = OCTANE_E_NOMEM; break; }
        if (hipStreamCreate(&s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d, u, n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d + n, v, n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(d + 2 * n, cth, n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        launch_srsal(s, d, d + n, d + 2 * n, nx, ny, A, d + 3 * n, d + 4 * n);
        if (hipGetLastError() != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(u, d + 3 * n, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipMemcpyAsync(v, d + 4 * n, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = OCTANE_E_HIP; break; }
    } while (0);
    if (rc != OCTANE_OK) g_last_error = "octane_srsal_run: HIP failure";
    if (s) (void)hipStreamDestroy(s);
    if (d) (void)hipFree(d);
    return rc;
}

#ifdef OCTANE_DIAG
// ---- from here to the end of octane_vof_tune: exported by the DIAGNOSTIC library only (include/octane_vof_dev.h, OCTANE_DIAG section) ----
// Diagnostic: time the two PCG passes of one pyramid level in isolation (values in the planes are irrelevant).
extern "C" int octane_vof_plan_probe(octane_vof_plan *pl, int level, int iterations, double *pass_a_ms, double *pass_b_ms)
{
    if (!pl || level < 0 || level >= (int)pl->lev.size() || iterations < 2 || !pass_a_ms || !pass_b_ms) {
        g_last_error = "octane_vof_plan_probe: invalid argument";
        return OCTANE_E_INVALID;
    }
    HIP_TRY(hipSetDevice(pl->device));
    if (pl->use_fused) {            // one kernel per iteration: its time is reported as "pass A", pass B as 0
        const double ms = probe_level(pl, level, iterations, nullptr, nullptr);
        if (ms < 0) { g_last_error = "octane_vof_plan_probe failed"; return OCTANE_E_HIP; }
        *pass_a_ms = ms; *pass_b_ms = 0.;
        return OCTANE_OK;
    }
    if (probe_level(pl, level, iterations, pass_a_ms, pass_b_ms) < 0) { g_last_error = "octane_vof_plan_probe failed"; return OCTANE_E_HIP; }
    return OCTANE_OK;
}

// Diagnostic: where the waves of one launch of the q-recomputing PCG kernel spend their time at pyramid level `level` (shader
// clock cycles summed over all waves, per seam of a tile: pcg_kernels.hip, g_q_stamps).  even != 0 times a launch that also
// updates x.  The planes are clobbered, values are irrelevant (stop test held open).
extern "C" int octane_vof_plan_probe_stamps(octane_vof_plan *pl, int level, int even, int unit_w, unsigned long long *out16)
{
    if (!pl || level < 0 || level >= (int)pl->lev.size() || !out16) return OCTANE_E_INVALID;
    HIP_TRY(hipSetDevice(pl->device));
    const LevelInfo &li = pl->lev[level];
    LevelPtrs L;
    const LevelCtx pc = {pl->img1p, pl->img2p, pl->uh, pl->vh, pl->gx1, pl->gy1, pl->gx2, pl->gy2, pl->gxx, pl->gxy, pl->gyy};
    fill_level_ptrs(pl, li, 0, pc, L);
    L.q_form = pcg_fused_q_form(li.w, li.h, li.h);
    if (!L.q_form) { g_last_error = "octane_vof_plan_probe_stamps: the level does not run the q-recomputing kernel"; return OCTANE_E_INVALID; }
    L.unit_w = unit_w ? 1 : 0;
    const int g_f = pcg_fused_grid_size(li.w, li.h, L.unit_w, L.q_form);
    std::vector<double> ones(2 * (size_t)kPartBlock, 0.0);
    for (int half = 0; half < 2; half++) {
        double *blk = ones.data() + (size_t)half * kPartBlock;
        blk[kPartRz] = 4.0; blk[kPartRr] = 4.0; blk[kPartPq] = 1.0; blk[kPartQz] = 1.0; blk[kPartQmq] = 1.0; blk[kPartRq] = 1.0; blk[kPartQq] = 1.0;
    }
    PcgState st[2];
    st[0].rz = 1.f; st[0].stopped = 0; st[0].iters = 0; st[0].pad = 0;
    st[1] = st[0];
    hipStream_t s = pl->own_stream;
    int rc = 0;
    for (int rep = 0; rep < 3 && rc == 0; rep++) {
        HIP_TRY(hipMemcpyAsync(pl->d_parts, ones.data(), 2 * (size_t)kPartBlock * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(pl->d_state, st, sizeof(st), hipMemcpyHostToDevice, s));
        rc = pcg_fused_q_stamps(s, L, even ? 4 : 5, 1, g_f, 0.f, out16);
    }
    return rc == 0 ? OCTANE_OK : OCTANE_E_HIP;
}

#endif  // OCTANE_DIAG

// Self-test: the three-instruction reciprocal of pcg_persist.hip against the IEEE division on every positive normal float whose
// reciprocal is normal.  out3 = {patterns compared, mismatches, a mismatching bit pattern}.
#ifdef OCTANE_DIAG
// Diagnostic: where thread 0 of the persistent solve's workgroups spent its shader-clock cycles since the last call (seams of an
// iteration, pcg_persist.hip MID_STAMP; only launches made under octane_vof_tune(plan, "persist_diag", 1) count): 32 values, [0..15] the
// sub-domains on the fast path, [16..31] the predicated ones; [14] / [30] = workgroups x iterations.  Clears the counters.
extern "C" int octane_vof_mid_stamps(int device, unsigned long long *out16)
{
    if (!out16) { g_last_error = "octane_vof_mid_stamps: null output"; return OCTANE_E_INVALID; }
    if (hipSetDevice(device) != hipSuccess) { g_last_error = "octane_vof_mid_stamps: no such device"; return OCTANE_E_HIP; }
    if (hipDeviceSynchronize() != hipSuccess) return OCTANE_E_HIP;
    return pcg_mid_stamps(nullptr, out16) == 0 ? OCTANE_OK : OCTANE_E_HIP;
}

// Self-test of the assembly's fast exact forms for `alpha` (vof_kernels.hip, assemble_math_selftest): out8 = {patterns, mismatches} of
// x / alpha on every finite float x, of 1 / (s + 1) on every float s >= 0, of 1 / sqrt(x + 1e-6) on every float x >= 0 -- each against the
// IEEE division / square root the reference's expression compiles to --, [6] a mismatching bit pattern, [7] which test it belongs to.
// A form with a mismatch is not used by plans with that alpha (octane_selftest_assembly_math_bits says which are).
extern "C" int octane_selftest_assembly_math(int device, double alpha, unsigned long long *out8)
{
    if (!out8) return OCTANE_E_INVALID;
    HIP_TRY(hipSetDevice(device));
    return assemble_math_selftest(nullptr, alpha, out8) == 0 ? OCTANE_OK : OCTANE_E_HIP;
}
extern "C" int octane_selftest_assembly_math_bits(int device, double alpha)
{
    if (hipSetDevice(device) != hipSuccess) return OCTANE_E_HIP;
    return assemble_fast_math_bits(alpha);
}

extern "C" int octane_selftest_rcp(int device, unsigned long long *out3)
{
    if (!out3) return OCTANE_E_INVALID;
    HIP_TRY(hipSetDevice(device));
    return pcg_selftest_rcp(nullptr, out3) == 0 ? OCTANE_OK : OCTANE_E_HIP;
}

// Developer knob setter (the same knobs the OCTANE_TUNE_* environment variables set at plan creation).
extern "C" int octane_vof_mid_geometry(int w, int h, int ncu, int *out5)
{
    if (!out5 || w < 1 || h < 1 || ncu < 1) { g_last_error = "octane_vof_mid_geometry: invalid argument"; return OCTANE_E_INVALID; }
    MidGeom g;
    if (pcg_mid_config(w, h, ncu, 0, &g) != 1) return 0;
    out5[0] = g.gx; out5[1] = g.gy; out5[2] = g.bh; out5[3] = g.P; out5[4] = g.G;
    return 1;
}

// Host arithmetic only: does the LDS-DMA PCG kernel rotate the tile columns by the tile row on a level of w x rows pixels walked by `grid`
// workgroups (pcg_row_rotation, pcg_kernels.hip)?  out3 = {tile columns, largest number of border-column tiles one workgroup walks without
// / with the rotation}.
extern "C" int octane_vof_row_rotation(int w, int rows, int grid, int walk_mode, int *out3)
{
    if (w < 1 || rows < 1 || grid < 1) { g_last_error = "octane_vof_row_rotation: invalid argument"; return OCTANE_E_INVALID; }
    return pcg_row_rotation_count(w, rows, grid, walk_mode, out3);
}

extern "C" int octane_vof_tune(octane_vof_plan *pl, const char *key, int value)
{
    if (!key) return OCTANE_E_INVALID;
    const std::string k(key);
    if (k == "pass_a") { set_pass_a_variant(value); return OCTANE_OK; }
    if (k == "max_blocks") { set_max_blocks(value); return OCTANE_OK; }
    if (!pl) return OCTANE_E_INVALID;
    if (k == "reverse_b") pl->reverse_b = value != 0;
    else if (k == "xcd") { pl->xcd_bands = value; set_grid_multiple(value == 1 ? 8 : 1); }
    else if (k == "nt") pl->nt_hints = value;
    else if (k == "defer_x") pl->defer_x = value != 0;
    else if (k == "small") pl->use_small = value != 0;
    else if (k == "small_max") pl->small_max_pixels = value;
    else if (k == "unit_w") pl->use_unit_w = value != 0;
    else if (k == "fused") pl->use_fused = value != 0;
    else if (k == "fused_q") set_fused_q(value);
#ifdef OCTANE_DIAG
    else if (k == "q_diag") set_q_diag(value);
    else if (k == "persist_diag") g_persist_diag = value != 0;
#endif
    else if (k == "q_dma") set_q_dma(value);
    else if (k == "persist") { pl->use_persist = value != 0; pl->persist_off_runs = 0; }
    else if (k == "persist_step") pl->persist_step = value;
    else if (k == "persist_p") pl->persist_p = value;
    else if (k == "overlap") pl->use_overlap = value != 0;
    else if (k == "trace_levels") pl->trace_levels = value;          // the debug tap reports the `value` coarsest levels only (0: all)
#ifdef OCTANE_DIAG
    else if (k == "persist_fault") set_mid_fault(value);            // the fault drill's hook exists in the diagnostic library only
#endif
    else if (k == "persist_max_g") pl->persist_max_g = value;
    else if (k == "persist_chain") pl->persist_chain = value != 0;   // 0: persistent launches of this plan are not serialised with other plans' (experiment)
    else if (k == "lane_mode") return octane_vof_plan_set_lane_mode(pl, value == 2 ? 2 : (value ? 1 : 0));
    else if (k == "fused_rows") set_fused_rows(value);
    else if (k == "asm_fast") pl->asm_fast = value ? assemble_fast_math_bits(pl->prm.alpha) : 0;    // 0: IEEE divisions throughout (same bits)
    else return OCTANE_E_INVALID;
    return OCTANE_OK;
}
#endif  // OCTANE_DIAG

// How many other plans work on this plan's device at the same time (include/octane_vof.h): 0 alone, 2 beside ONE other plan (the two lanes
// of octane_vof_batch_run: persistent solves concurrent, each capped at half the compute units), 1 beside two or more (only the tiny levels
// keep the persistent solve).
extern "C" int octane_vof_plan_set_lane_mode(octane_vof_plan *pl, int mode)
{
    if (!pl || mode < 0 || mode > 2) { g_last_error = "octane_vof_plan_set_lane_mode: invalid argument"; return OCTANE_E_INVALID; }
    if (mode == 2) plan_lane_pair_mode(pl);
    else if (mode == 1) plan_lane_mode(pl);
    else { pl->persist_max_g = kMidMaxG; pl->small_max_pixels = 1536; pl->persist_chain = 1; }
    return OCTANE_OK;
}
