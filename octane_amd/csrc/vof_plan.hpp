// vof_plan.hpp -- the plan object behind include/octane_vof.h and the pieces of its level loop that the row-band
// coordinator (vof_tiled.hip) drives band by band.  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/octane_vof_dev.h"
#include "../../include/octane_extras.h"
#include "vof_kernels.hpp"

struct LevelInfo {
    float factor;
    int w, h, pitch;
    int fs;            // blur half width (unused at the finest level)
    int tap_off;       // offset of this level's taps in the device tap table
    float lambdac;
};

struct EvPair { hipEvent_t a, b; int kind; };

struct octane_vof_plan {
    octane_vof_params prm;
    int nx, ny, nc, device;
    int pitch0;
    size_t plane0;                 // floats per full-resolution plane
    std::vector<LevelInfo> lev;
    float *arena = nullptr;
    size_t arena_bytes = 0;
    // Band plans (one band of a row-band solve) keep the planes the NEIGHBOURING bands read -- the flow, the operator, the CG vectors: the
    // 30 planes from U[0] on -- in an allocation of their own, and only that one is peer-addressed / exported over HIP IPC.  Round 5: with
    // ONE 18.9 GiB arena per band the process form of a 10848^2 frame never got past octane_vof_mp_connect (two ranks on one GPU: no return
    // from the IPC mapping within 200 s, twice; a 19 GiB allocation did not open within 400 s in tools/ipc_probe.py either, while 1 ... 17
    // GiB open in < 1 ms, profiles/r5_ipc_probe.txt).  The shared part is 13.2 GiB there and the same two ranks connect in 1 s.  Mapping
    // only what is read also keeps a neighbour's stray pointer out of this band's inputs.  The band's private planes (inputs, level images,
    // gradients) stay in `arena`.  Plain plans: xarena == nullptr, everything in `arena`, layout unchanged.
    float *xarena = nullptr;
    size_t xarena_bytes = 0;
    float *shared_base() const { return xarena ? xarena : arena; }
    size_t shared_bytes() const { return xarena ? xarena_bytes : arena_bytes; }
    // planes (all plane0 floats unless noted)
    float *img1p, *img2p, *uh, *vh, *lev1, *lev2;
    float *gx1, *gy1, *gx2, *gy2, *gxx, *gxy, *gyy;
    float *U[2], *V[2], *ut, *vt;
    float *a1, *a2, *a4, *wx, *wy, *mu, *mv, *ru, *rv, *pu[2], *pv[2], *qu, *qv, *xu, *xv, *tmp;
    float *ru2, *rv2, *qu2, *qv2;  // second halves of the r / q double buffers of the fused PCG kernel
    float *pu3, *pv3;              // third p buffer of the fused kernel (deferred x update)
    // Second set of a level's flow-independent planes (pyramid images, first-guess hint, the seven gradient fields): while level k
    // is being solved out of one set, the side stream prepares level k + 1 in the other (run_on_stream)
    float *lev1b, *lev2b, *utb, *vtb, *gx1b, *gy1b, *gx2b, *gy2b, *gxxb, *gxyb, *gyyb;
    float *d_taps = nullptr;
    double *d_parts = nullptr;     // 2 * kPartBlock
    octane::PcgState *d_state = nullptr;   // 2
    float *d_alpha = nullptr;      // 2
    long long *d_iters = nullptr;
    long long *h_iters = nullptr;  // pinned
    hipStream_t own_stream = nullptr;
    hipStream_t side_stream = nullptr;      // prepares the next level's images and gradients beside the current level's solve
    hipEvent_t ev_fork = nullptr, ev_img[2] = {nullptr, nullptr}, ev_solved[2] = {nullptr, nullptr};
    int use_overlap = 1;                    // OCTANE_TUNE_OVERLAP=0: everything on one stream, level by level
    int has_bset = 1;                       // the second set of level planes exists (not in band plans, not with the overlap off at creation)
    octane_vof_trace_fn trace = nullptr;
    void *trace_user = nullptr;
    int trace_levels = 0;                   // > 0: the debug tap reports only that many coarsest levels (tune key "trace_levels")
    int profiling = 0;
    std::vector<EvPair> evs;
    size_t evs_used = 0;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;
    octane_vof_profile prof;
    float tol;
    int reverse_b = 1;
    int xcd_bands = 4;   // tile walk of the fused PCG kernels: runs of 8 adjacent tiles per XCD (device_util.hpp, item_range_walk); +0.6 %
    int use_small = 1;
    int defer_x = 1;
    int use_fused = 1;   // one fused kernel per PCG iteration (84 B/px) instead of pass A + pass B (104 B/px)
    int use_unit_w = 1;  // pass A skips the wx / wy planes while they hold the constant -1 (first GNC step)
    int use_graph = 0;   // OCTANE_TUNE_GRAPH=1: replay the pyramid as one hipGraph (measured: no throughput gain,
                         // the host already runs ahead of the GPU; useful only when calls are latency-bound)
    hipGraphExec_t graph_exec = nullptr;
    int graph_cur = 0;
    int nt_hints = 15;   // x, q, mu/mv in pass B and a2 in pass A are single-use: streaming loads/stores
    // Mid-size levels (above what one workgroup holds, up to ~1.9 Mpixel): the whole solve in ONE persistent launch with the
    // level resident on chip (pcg_persist.hip).  persist_step > 0 runs that many iterations per launch with the state in the
    // level's planes instead (the per-launch form the persistent one is checked against); persist_p forces the slot count.
    int use_persist = 1, persist_step = 0, persist_p = 0;
    long small_max_pixels = 1536;           // levels of up to this many pixels (three per thread) run the single-workgroup solve; larger ones, up to the 6144 it can hold, only when the persistent solve is off
    int persist_max_g = 1 << 20;   // cap on the workgroups (= CUs held for a whole solve) of one persistent launch: lanes of a batch lower it
    int persist_chain = 1;           // persistent launches are serialised per device among the plans of this process (vof_plan.hip, persist_launch)
    long persist_max_pixels = 2L << 20;
    int ncu = 0;                   // compute units of the device
    void *d_mid = nullptr;         // workspace of the persistent solve (abort word, granules of partial sums and edge pixels)
    unsigned mid_seq = 1;          // solves issued on that workspace: part of the granules' tags
    unsigned *h_mid_abort = nullptr;   // pinned copy of the abort word, refreshed at the end of every run
    std::vector<float> pcg_launch_ms;   // the finest-level PCG launches of the last profiled run, in launch order
    // An abandoned persistent solve (its workgroups could not all become resident: a co-tenant on the GPU) switches the persistent solve
    // off for the NEXT persist_off_runs runs of this plan, not for its whole life, and the run is made again: at once on the
    // host-buffer path, by the next synchronising call on the device-buffer path -- from the plan's own copy of the inputs, into the
    // caller's output buffers of that run (heal_abandoned_run)
    int persist_off_runs = 0;
    int persist_abandoned_total = 0;        // how often that happened to this plan
    float *last_u = nullptr, *last_v = nullptr;
    hipStream_t last_stream = nullptr;
    int last_mem = -1;
    int asm_fast = 0;    // AssembleParams::fast_math: the fast exact forms the device self-test has cleared for this plan's alpha
    int ntrials = 0;     // placement trials made when the plan was created, and what each candidate arena measured
    double trial_ms[8] = {0};
};


namespace octane {

// where level k's images, first-guess hint and gradient fields live
struct LevelCtx { const float *lev1, *lev2, *ut, *vt; float *gx1, *gy1, *gx2, *gy2, *gxx, *gxy, *gyy; };

void set_last_error(const std::string &msg);
// Plan without the placement trials (several bands may share one device, and the trials allocate 4 arenas).
// band_plan: the plan is one band of a row-band solve (vof_tiled.hip): it never runs run_on_stream, so the second set of level planes the
// one-level-ahead overlap needs (9 nc + 2 full-size planes, 5.2 GB at 10848^2) is not allocated; nor is it when OCTANE_TUNE_OVERLAP=0.
int  plan_create_ex(octane_vof_plan **out, int nx, int ny, int nchan, const octane_vof_params *p, int placement_trials, bool band_plan = false);
// Level k up to the point where the solve starts: flow up-sampling (flips `cur`), pyramid images, gradients.
// Everything is computed for the whole level -- bands replicate this work instead of exchanging halos for it.
int  plan_level_setup(octane_vof_plan *pl, hipStream_t s, int k, int &cur, LevelCtx &c);
// The part of it that does not depend on the flow (pyramid images, hint, gradients), into plane set `which` (0 or 1).
int  plan_level_images(octane_vof_plan *pl, hipStream_t s, int k, int which, LevelCtx &c);
// The three GNC steps x liters linearisations x cgiters PCG iterations of level k over the whole level.
int  plan_level_solve(octane_vof_plan *pl, hipStream_t s, int k, int cur, const LevelCtx &c, bool profile_kernels);
void plan_fill_level_ptrs(octane_vof_plan *pl, int k, int cur, const LevelCtx &c, LevelPtrs &L);
// Upload (host) or copy (device) the inputs into the plan's planes on stream s / copy the flow of U[cur],V[cur] out.
int  plan_load_inputs(octane_vof_plan *pl, const float *img1, const float *img2, const float *u, const float *v, int mem, hipStream_t s);
// The abort word of the persistent mid-level solves (pcg_persist.hip) is per run: persist_begin_run clears it on the stream before the
// run's first solve, persist_end_run copies it to the host at the run's end (both stream-ordered), and persist_check -- after the
// host has synchronised with the run -- turns a raised word into OCTANE_E_HIP + octane_last_error and clears it.  run_on_stream
// does the first two itself; the row-band paths, which call plan_level_solve directly, use these.
int  persist_begin_run(octane_vof_plan *pl, hipStream_t s);
int  persist_end_run(octane_vof_plan *pl, hipStream_t s);
int  persist_check(octane_vof_plan *pl);

}  // namespace octane
