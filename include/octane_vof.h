/*
 * octane_vof.h -- C-ABI of the MI355X-native dense variational optical-flow core.
 *
 * This is the drop-in boundary for OCTANE's hot path.  Plain pointers and sizes
 * only; no C++ types, no torch types, no exceptions cross it.  Every entry point
 * returns 0 on success or a negative OCTANE_E_* code.  The library is
 * liboctane_vof.so (octane_amd/csrc/Makefile).  The C++ shims that keep the
 * reference's own signatures (Image / OFFlags / GOESVar by value or reference)
 * live in include/octane_host.hpp and forward here.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   octane_vof_run            <- void oct_variational_optical_flow(Image,Image,float*,float*,float*,int,int,int,OFFlags)
 *                                src/oct_variational_optical_flow.cu:1213 (declared by its caller at src/oct_optical_flow.cc:12)
 *   octane_vof_plan_*         <- the per-call allocate / launch / free sequence of the same function
 *                                (.cu:1268-1328 allocations, :1431 launch, :1441-1472 frees), split so that a
 *                                caller can keep device state across image pairs
 *   octane_vof_batch_run      <- (new) the reference is single-GPU: args.setdevice -> cudaSetDevice, .cu:1251-1265
 *   octane_vof_tiled_*        <- (new) one frame over several GPUs as row bands of the fine pyramid levels, same
 *                                arithmetic as the single-GPU solve: one global PCG per linearisation (.cu:1105-1195)
 *   octane_pix2uv_run         <- void oct_pix2uv_cuda(GOESVar&,double,float*,float*,short*,short*,short*,short*,OFFlags)
 *                                src/oct_pix2uv_cuda.cu:265 (declared at src/oct_optical_flow.cc:15)
 */
#ifndef OCTANE_VOF_H
#define OCTANE_VOF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCTANE_OK            0
#define OCTANE_E_INVALID    (-1)   /* bad argument (null pointer, non-positive size, nchan not in 1..3, ...) */
#define OCTANE_E_NODEVICE   (-2)   /* no GPU visible (the reference prints a message and exit(0)s, .cu:1255-1259) */
#define OCTANE_E_HIP        (-3)   /* a HIP runtime call failed; octane_last_error() has the text */
#define OCTANE_E_TOOSMALL   (-4)   /* some pyramid level would be narrower than 2 pixels (reference indexes out of bounds there) */
#define OCTANE_E_NOMEM      (-5)

/* The OFFlags fields the solver reads (include/offlags.h; read at .cu:1229-1254). */
typedef struct octane_vof_params {
    double alpha;     /* -alpha   smoothness weight; data terms are divided by it (.cu:831-833) */
    double lambda;    /* -lambda  gradient-constancy weight */
    double lambdac;   /* -lambdac first-guess hinting weight (scaled by 0.5^level, .cu:494) */
    double scaleF;    /* pyramid scale factor (0.5; not CLI-settable in the reference) */
    double scsig;     /* read by the reference (.cu:1238) but only used when dodiscrete, which is hard-wired false */
    int kiters;       /* -kiters  pyramid levels */
    int liters;       /* -liters  re-linearisations per GNC step */
    int cgiters;      /* PCG iteration cap per solve (OFFlags.cgiters, default 30) */
    int dozim;        /* 1 = Zimmer normalisation (default), 0 = -brox */
    int device;       /* OFFlags.setdevice (0-based); out-of-range falls back to 0 as at .cu:1260-1264 */
} octane_vof_params;

/* Fills *p with the defaults of the reference CLI (src/main.cc:78-96,102). */
void octane_vof_default_params(octane_vof_params *p);

/* One-shot solve on host buffers: allocate device state, upload, solve, download, free.
 * img1/img2: nchan planes of ny rows of nx floats (Image.data[i + nx*j + nx*ny*c]).
 * u_inout/v_inout: nx*ny floats each; in = first guess, out = flow in pixels.
 * Caller keeps ownership of every pointer; inputs are not modified. */
int octane_vof_run(const float *img1, const float *img2, int nx, int ny, int nchan,
                   float *u_inout, float *v_inout, const octane_vof_params *p);
/* The same with the first guess apart from the result.  u0 = v0 = NULL: zero first guess (what oct_optical_flow passes without
 * -firstguess, ref src/oct_optical_flow.cc:38-48) -- and nothing is uploaded for it: two of the call's six PCIe transfers less. */
int octane_vof_solve(const float *img1, const float *img2, int nx, int ny, int nchan, const float *u0, const float *v0,
                     float *u_out, float *v_out, const octane_vof_params *p);
/* octane_vof_run keeps the plan of its last call and reuses it when the next call has the same shape and parameters
 * (creating and freeing a multi-GB arena per pair costs up to 0.5 s at 10848^2).  The first call of a process takes the first arena
 * it gets; the first REUSE re-creates the plan with the placement trials of octane_vof_plan_create (~0.3 s once at 5000^2, up to 11 %
 * per pair from then on), so a single-pair run never pays for them.  This frees the kept plan;
 * OCTANE_VOF_CACHE=0 in the environment restores allocate-per-call, as the reference does (.cu:1268-1472). */
void octane_vof_release_cache(void);

/* Plan API: device state sized for (nx, ny, nchan, params) and reused across pairs. */
typedef struct octane_vof_plan octane_vof_plan;

int octane_vof_plan_create(octane_vof_plan **plan, int nx, int ny, int nchan, const octane_vof_params *p);
int octane_vof_plan_destroy(octane_vof_plan *plan);
size_t octane_vof_plan_device_bytes(const octane_vof_plan *plan);
/* Plans of 4 Mpixel and more allocate up to eight candidate arenas, time a few PCG iterations on each and keep the
 * fastest (where an arena lands in physical memory is worth up to 11 %).  Returns how many candidates were timed and
 * copies up to `cap` of their times (ms per PCG iteration at the finest level) into ms: a throughput measured on such
 * a plan is a "best of n placements" figure and should be reported with this spread.  No reference counterpart. */
int octane_vof_plan_placement_trials(const octane_vof_plan *plan, double *ms, int cap);

#define OCTANE_MEM_HOST   0
#define OCTANE_MEM_DEVICE 1
#define OCTANE_STREAM_OWN ((void *)(long)-1)   /* hip_stream value: the stream private to the plan */
/* Solve one pair.  mem says where img1/img2/u/v live.  hip_stream is a hipStream_t.
 * OCTANE_MEM_DEVICE: the call only enqueues work on hip_stream (NULL = HIP's null stream) and
 * returns.  OCTANE_MEM_HOST: the call uploads, solves, downloads and synchronises before
 * returning (NULL = a stream private to the plan).
 * OCTANE_STREAM_OWN selects the plan's private stream for device buffers too: the inputs must be complete when the call is
 * made, and octane_vof_plan_wait() (or the next blocking call on the plan) tells when the outputs are.  Two plans
 * working side by side on one GPU (the lanes of a batch) should use it: their private streams sit on different
 * hardware queues, whereas two streams of a framework's pool may share one (ROCm maps streams onto GPU_MAX_HW_QUEUES = 4
 * queues) and then never overlap. */
int octane_vof_plan_run(octane_vof_plan *plan, const float *img1, const float *img2,
                        float *u_inout, float *v_inout, int mem, void *hip_stream);
/* The same with the first guess and the result in separate buffers; u0 = v0 = NULL means a zero first guess (what
 * oct_optical_flow uses without -firstguess, src/oct_optical_flow.cc:38-48). */
int octane_vof_plan_solve(octane_vof_plan *plan, const float *img1, const float *img2, const float *u0, const float *v0,
                          float *u_out, float *v_out, int mem, void *hip_stream);

/* Blocks until the plan's private stream and the stream of its last device-buffer run -- the NULL stream too, if that is what the
 * run was given -- are idle.  octane_vof_plan_last_iterations waits for that run's stream as well before it reads the run's count:
 * the stream handle passed to a device-buffer run therefore has to stay valid until one of the two has been called. */
int octane_vof_plan_wait(octane_vof_plan *plan);
/* Number of PCG iterations the last completed run executed (sum over all solves).
 * An ABANDONED persistent solve: the mid-size pyramid levels are solved by ONE persistent launch each whose workgroups all have to be
 * resident on the GPU at once; if they cannot become so within 0.25 s (another process running the same kind of kernel on this GPU)
 * the solve gives up instead of hanging the GPU, and the run's flow is not valid.  The library repairs that itself: host-buffer runs
 * solve the pair again before they return; device-buffer runs are repaired by the first of octane_vof_plan_wait /
 * octane_vof_plan_last_iterations called after the caller has synchronised with the run -- the pyramid is made again (one launch per
 * PCG iteration) from the plan's own copy of the inputs, on that run's stream, and written into THAT RUN'S output buffers, which
 * therefore have to stay valid until one of the two has been called.  A caller of the device-buffer path that calls neither never
 * learns of an abandoned solve: it MUST call one of them per run before it uses the flow.  -2 / OCTANE_E_HIP only when the repair
 * fails too.  After such an event the plan runs its next 16 pairs with one launch per iteration and then tries the persistent solve
 * again; octane_vof_plan_persist_state tells: 1 on, 0 off (OCTANE_TUNE_PERSIST=0 / tune), -n off for the next n runs, and how
 * often a solve of this plan was abandoned.  The row-band forms (octane_vof_tiled_*, octane_vof_mp_*) report -2 / an error and do
 * not retry. */
long long octane_vof_plan_last_iterations(octane_vof_plan *plan);
int octane_vof_plan_persist_state(const octane_vof_plan *plan, int *abandoned_total);

/* Debug tap (NULL = off, zero cost): called on the host after each stage with a copy of the stage's
 * planes: data is nplanes planes of ny rows of nx floats.  Tags match oracle/vof_oracle.c's trace. */
typedef void (*octane_vof_trace_fn)(void *user, const char *tag, int level, int gnc, int l,
                                    const float *data, int nx, int ny, int nplanes);
int octane_vof_plan_set_trace(octane_vof_plan *plan, octane_vof_trace_fn fn, void *user);

/* Per-kernel timing of the finest pyramid level, measured with HIP events on the run's stream. */
typedef struct octane_vof_profile {
    double pass_a_ms;  long long pass_a_launches;   /* PCG pass A: p = z + beta p, Ap, p.Ap      */
    double pass_b_ms;  long long pass_b_launches;   /* PCG pass B: x += a p, r -= a Ap, r.z, r.r */
    double assemble_ms; long long assemble_launches;
    double update_ms;  long long update_launches;
    double setup_ms;                                  /* all level-setup kernels, every level */
    double total_ms;                                  /* whole run, first launch to last       */
    long long finest_pixels;
} octane_vof_profile;
int octane_vof_plan_set_profiling(octane_vof_plan *plan, int enable);
int octane_vof_plan_get_profile(octane_vof_plan *plan, octane_vof_profile *out);
/* The finest-level PCG launches of the last profiled run, one duration (ms) each, in launch order: solve after solve (3 GNC steps x
 * liters solves), cgiters launches per solve.  Launch k of a solve moves other bytes than launch k + 1 (x is updated by every second
 * launch, the first GNC step's weights are the constant -1 and are not read): bench.py prices each kind on its own bytes.  Returns
 * the number of launches recorded; writes min(that, cap) values. */
int octane_vof_plan_get_launch_times(octane_vof_plan *plan, float *ms, int cap);
/* Developer knob, per plan: key in {overlap, persist, persist_p, persist_step, persist_max_g, persist_chain, lane_mode, small, small_max,
 * pass_a, max_blocks, reverse_b, xcd, nt, defer_x, unit_w, fused, fused_q, fused_rows, q_dma, asm_fast, trace_levels}.  lane_mode: 0 the plan
 * runs alone on its device; 2 beside ONE other plan (the two lanes of octane_vof_batch_run: persistent solves concurrent, each capped at half
 * the compute units); 1 beside two or more (only the tiny levels keep the persistent solve).  Results agree for every
 * setting to the last bits of the PCG scalars (the grouping of the fp64 partial sums follows the grid); only speed changes.
 *
 * ENVIRONMENT.  The product library reads exactly these variables (none is needed):
 *   OCTANE_VOF_CACHE=0            one-shot entry: allocate per call            OCTANE_VOF_BANDS=n           C++ shim: n row bands
 *   OCTANE_TILED_TRANSPORT=...    row bands: force a transport                 OCTANE_TILED_SELFCHECK=0     ... skip the first-contact check
 *   OCTANE_TUNE_MIN_BAND_PIXELS=n row bands: banding threshold                 OCTANE_MP_TIMEOUT_S=s        process form: boundary time-out
 *   OCTANE_PIX2UV_FMAD=0|1|2      which build of the navigation kernel         OCTANE_TUNE_PERSIST_MAXG=g   processes SHARING a GPU: workgroups a
 *   OCTANE_TUNE_Q_DMA=0, OCTANE_TUNE_PERSIST=0   the bisect pair: the finest levels' kernel              persistent solve may hold
 *       without LDS-DMA / one launch per PCG iteration on the mid-size levels -- the first two things to switch off when a result is in doubt.
 * Every other OCTANE_TUNE_* variable of rounds 1-4 (about thirty tuning switches) is read by the DIAGNOSTIC library only
 * (liboctane_vof_diag.so, `make -C octane_amd/csrc DIAG=1`; tools/ load it): in the product they do not exist, so a stray variable in a
 * production environment cannot change which kernels run. */
int octane_vof_tune(octane_vof_plan *plan, const char *key, int value);
/* Which sub-domain grid the persistent mid-level solve (pcg_persist.hip) takes for a level of w x h pixels on a device with `ncu`
 * compute units: out5 = {columns of sub-domains, rows of sub-domains, rows per sub-domain, slots of 8 rows per thread, workgroups}.
 * Returns 1, or 0 when the level does not fit the device (such levels run one launch per PCG iteration).  Host arithmetic only: no
 * GPU is touched.  (The plan additionally keeps levels of <= 1536 pixels for the single-workgroup solve and levels above 2 Mi pixels
 * for the streaming kernel.) */
int octane_vof_mid_geometry(int w, int h, int ncu, int *out5);
/* Self-test of the persistent PCG kernel's three-instruction reciprocal (hardware estimate + one fused Newton step) against the
 * IEEE division on every positive normal float whose reciprocal is normal: out3 = {patterns compared, mismatches, one
 * mismatching bit pattern}.  No reference counterpart (the reference divides, ref .cu:141-149). */
int octane_selftest_rcp(int device, unsigned long long *out3);
/* Self-test of the assembly kernel's fast exact forms for one alpha (ref assembly .cu:611-1097): the reference divides by alpha five
 * or six times per pixel, takes 1 / (s + 1) three times per channel (Zimmer's normalisation) and 1 / sqrt(x + 1e-6) twice, all in double
 * and rounded to float afterwards.  The library uses cheaper instruction sequences for these where -- and only where -- the
 * sequence reproduces the IEEE result on EVERY float input for that alpha; this runs the comparison: out8 = {patterns, mismatches}
 * x {x / alpha, all floats but the NaNs; 1 / (s + 1), all floats s >= 0 and +inf; 1 / sqrt(x + 1e-6), the same}, [6] = a mismatching bit pattern,
 * [7] = its test.  octane_selftest_assembly_math_bits: which forms (bit 0, 1, 2 in that order) plans with this alpha use; runs the
 * self-test the first time an alpha is seen (~10 ms), exactly as plan creation does. */
int octane_selftest_assembly_math(int device, double alpha, unsigned long long *out8);
int octane_selftest_assembly_math_bits(int device, double alpha);
/* Diagnostic: time `iterations` (>= 2) PCG iterations of pyramid level `level` (0 = coarsest) in isolation, on
 * whatever the plan's planes hold (the stop test is held open, values are irrelevant, the planes are clobbered).
 * With the one-kernel iteration (the default) its time comes back in *pass_a_ms and *pass_b_ms is 0. */
int octane_vof_plan_probe(octane_vof_plan *plan, int level, int iterations, double *pass_a_ms, double *pass_b_ms);

#ifdef OCTANE_DIAG
/* ---- Diagnostic library only (liboctane_vof_diag.so, `make -C octane_amd/csrc DIAG=1`): the product library does not export these, does
 * not contain the stamped kernel copies behind them, and rejects the tune keys "q_diag" / "persist_diag".  No reference counterpart. ---- */
/* Diagnostic: where the waves of one launch of the q-recomputing PCG kernel spend their time at pyramid level `level`: shader
 * clock cycles summed over all waves, per seam of a tile (out16[0..6]), tiles walked ([7]), prologue ([8]), epilogue ([9]).
 * even != 0 stamps a launch that also updates x.  The planes are clobbered.  No reference counterpart. */
int octane_vof_plan_probe_stamps(octane_vof_plan *plan, int level, int even, int unit_w, unsigned long long *out16);
/* Diagnostic: cycles per seam of an iteration of the persistent mid-level solve (stamped build, octane_vof_tune "persist_diag"),
 * summed over workgroups and iterations since the last call: 32 values, [0..15] interior sub-domains, [16..31] the predicated ones,
 * [14] / [30] = workgroups x iterations.  Clears the counters. */
int octane_vof_mid_stamps(int device, unsigned long long *out32);
#endif  /* OCTANE_DIAG */

/* Independent pairs sharded over GPUs: pair b runs on devices[b % ndevices]; one host thread per
 * device; no collective.  Pointer arrays have npairs entries of host buffers laid out as above. */
int octane_vof_batch_run(int npairs, const float *const *img1, const float *const *img2,
                         int nx, int ny, int nchan, float *const *u_inout, float *const *v_inout,
                         const octane_vof_params *p, int ndevices, const int *devices);

/* ---- one frame over several GPUs (BASELINE configs[3]: a full-disk pair as row bands) ----------------------------
 * `nbands` (1..8) row bands, band b on devices[b] (NULL: b modulo the device count; ids may repeat -- several bands
 * then share a device, which is how a one-GPU machine exercises this path).  Pyramid levels with fewer than
 * min_band_pixels pixels (0 = default, 4 Mpixel; 12 until round 3) are solved redundantly by every band; on the larger ones a band
 * owns a range of rows and the bands exchange, per PCG iteration, their reduction partials and one row of the
 * residual per inner edge (stream-ordered peer copies over xGMI; no host synchronisation inside a pyramid).
 * The iterates are those of the single-GPU solve up to the summation order of the dot products.
 * Every band holds full-size planes: octane_vof_tiled_device_bytes() is per band.
 *   _load   uploads (OCTANE_MEM_HOST) or copies (OCTANE_MEM_DEVICE, dense buffers on devices[0]) the pair and the
 *           first guess to every band; blocking
 *   _solve  issues one pyramid on the loaded inputs; asynchronous
 *   _wait   blocks until the bands' streams are idle
 *   _fetch  waits, then copies the flow out of band 0 (host buffers, or dense device buffers on devices[0])
 *   _run    = load + solve + fetch (u/v in-out as for octane_vof_run) */
typedef struct octane_vof_tiled octane_vof_tiled;
int octane_vof_tiled_create(octane_vof_tiled **out, int nx, int ny, int nchan, const octane_vof_params *p,
                            int nbands, const int *devices, long long min_band_pixels);
int octane_vof_tiled_destroy(octane_vof_tiled *t);
int octane_vof_tiled_load(octane_vof_tiled *t, const float *img1, const float *img2, const float *u, const float *v, int mem);
int octane_vof_tiled_solve(octane_vof_tiled *t);
int octane_vof_tiled_wait(octane_vof_tiled *t);
int octane_vof_tiled_fetch(octane_vof_tiled *t, float *u, float *v, int mem);
int octane_vof_tiled_run(octane_vof_tiled *t, const float *img1, const float *img2, float *u_inout, float *v_inout, int mem);
/* How `rows` rows are cut into nbands bands: fills edges[0..nbands] (edges[0] = 0, edges[nbands] = rows, inner edges at
 * multiples of 32 rows) and returns 1; returns 0 when such a level stays replicated (one band, or fewer than 32 rows
 * for some band); negative on bad arguments.  Host arithmetic only (no GPU needed). */
int octane_vof_band_partition(int rows, int nbands, int *edges);
int octane_vof_tiled_banded_levels(const octane_vof_tiled *t);                 /* how many levels are split into bands */
int octane_vof_tiled_band_rows(const octane_vof_tiled *t, int level, int band, int *y0, int *y1);   /* 1 banded, 0 replicated */
long long octane_vof_tiled_last_iterations(octane_vof_tiled *t);               /* PCG iterations of the last pyramid */
long long octane_vof_tiled_last_copies(octane_vof_tiled *t);                   /* peer copies the last solve issued */
size_t octane_vof_tiled_device_bytes(const octane_vof_tiled *t);

/* ---- the same row-band solve with one band per PROCESS (the one-process-per-GPU launch) -----------------------------
 * Rank r of `world` (<= 8) owns band r on the device params->device selects.  Protocol, every step collective:
 *   octane_vof_mp_create   builds the band and opens the POSIX shared-memory object `shm_name` ("/something", the same on
 *                          every rank, unique per job; rank 0 creates it) that holds the ranks' phase barrier
 *   octane_vof_mp_handles  writes OCTANE_MP_HANDLE_BYTES bytes: the HIP IPC handles of this rank's two allocations; the
 *                          host program all-gathers them in rank order (torch.distributed, MPI, ...)
 *   octane_vof_mp_connect  maps the other ranks' allocations from the gathered world * OCTANE_MP_HANDLE_BYTES bytes
 *   octane_vof_mp_run      every rank passes the whole pair (host buffers, or dense device buffers on its own device)
 *                          and the first guess (NULL, NULL = zero); the flow arrives in u / v on rank 0 only
 * Same kernels and halo / partial protocol as octane_vof_tiled_*; a phase boundary is a stream drain plus a barrier in
 * shared memory.  No collective library is involved on the data path. */
#define OCTANE_MP_HANDLE_BYTES 128
typedef struct octane_vof_mp octane_vof_mp;
int octane_vof_mp_create(octane_vof_mp **out, int nx, int ny, int nchan, const octane_vof_params *p, int rank, int world,
                         long long min_band_pixels, const char *shm_name);
int octane_vof_mp_handles(octane_vof_mp *m, void *buf);
int octane_vof_mp_connect(octane_vof_mp *m, const void *all_handles);
int octane_vof_mp_run(octane_vof_mp *m, const float *img1, const float *img2, const float *u0, const float *v0,
                      float *u_out, float *v_out, int mem);
int octane_vof_mp_banded_levels(const octane_vof_mp *m);
long long octane_vof_mp_last_iterations(octane_vof_mp *m);
int octane_vof_mp_destroy(octane_vof_mp *m);

/* ---- transports of the row-band solve, the collective one, and the first-contact self-check (round 4) ----------------
 * What crosses bands (per PCG iteration the seven partial sums of every workgroup and a few rows per inner edge; per
 * linearisation two rows of the flow; per level the bands of the flow) can travel three ways.  All three run the same kernels in
 * the same order on the same values -- one global PCG per linearisation, ref .cu:1105-1195 -- and give the same bits:
 *   INPLACE     the consuming kernel reads the neighbouring band's memory through peer / IPC mappings (LDS-DMA included)
 *   COPY        stream-ordered runtime copies (hipMemcpyPeerAsync, hipMemcpyAsync on IPC mappings) pull the same bytes into the
 *               band's own planes / a local mirror; kernels read local memory only
 *   COLLECTIVE  (one band per process only) the HOST PROGRAM'S collective library moves them: an all-gather of the partial blocks
 *               and point-to-point sends / receives of the rows, through the callbacks below -- torch.distributed in this
 *               repository (octane_amd/exchange.py: backend nccl = RCCL over xGMI on a node, gloo staged through the host);
 *               needs neither peer access nor HIP IPC.  north_star: "RCCL over xGMI only for halo exchange / result gather".
 * OCTANE_TILED_TRANSPORT=inplace|copy|collective in the environment forces one.  Otherwise creation runs a FIRST-CONTACT
 * SELF-CHECK on the devices / ranks it was given: a small two-level frame (bands of 2 Mpixel when the real plan's bands run the
 * q-recomputing LDS-DMA kernel) is solved by the plain single-device plan and as row bands under each candidate in the order
 * inplace -> inplace with register staging instead of LDS-DMA from the neighbour -> copy -> collective, and the first candidate whose
 * flow is within 2e-5 of the plain plan's with equal iteration counts is what the plan uses (one line on stderr when that is not the
 * first).  No peer access / an IPC mapping that cannot be opened removes the candidates that need it instead of failing creation.
 * OCTANE_TILED_SELFCHECK=0 skips the check (inplace where possible).  No reference counterpart (the reference is single-GPU). */
#define OCTANE_TRANSPORT_INPLACE    0
#define OCTANE_TRANSPORT_COPY       1
#define OCTANE_TRANSPORT_COLLECTIVE 2
typedef struct octane_vof_transport_info {
    int transport;            /* OCTANE_TRANSPORT_* the plan uses */
    int q_dma;                /* 1: q-form band launches fetch the neighbour's rows by LDS-DMA; 0: register-staged kernel */
    int selfcheck;            /* 0 not run (one band, forced, switched off); 1 the first candidate passed; 2 downgraded; -1 none passed */
    int candidates_tried;
    int forced;               /* OCTANE_TILED_TRANSPORT named the transport */
    int peer_ok;              /* every pair of distinct devices has peer access (thread form) / every IPC mapping opened (process form) */
    int ndevices;             /* distinct devices among the bands (process form: ranks on distinct GPUs) */
    int nbands;
    double check_rel_l2[4];   /* distance to the plain plan of candidate i of the self-check (-1: not tried) */
    char exchange[48];        /* name of the host program's collective library when one was registered, else "" */
} octane_vof_transport_info;
int octane_vof_tiled_transport_info(const octane_vof_tiled *t, octane_vof_transport_info *out);
int octane_vof_mp_transport_info(const octane_vof_mp *m, octane_vof_transport_info *out);
const char *octane_vof_transport_name(int transport);
/* The collective transport's callbacks.  Every buffer is device memory of the calling rank's GPU; the library has drained its
 * stream before a call and a call returns when its data is in place (host-synchronous).  Both are collective over the ranks:
 *   all_gather  every rank contributes `bytes` bytes at `send`; rank c's contribution has to arrive at recv[c] (c != own rank;
 *               recv[own] is NULL)
 *   sendrecv    n transfers, all posted before any is waited for; ops[i] sends (send = 1) or receives `bytes` bytes to / from rank
 *               `peer`.  Between a pair of ranks the k-th send of one matches the k-th receive of the other.
 * Return 0 on success. */
typedef struct octane_vof_xfer { int peer; int send; void *buf; size_t bytes; } octane_vof_xfer;
typedef struct octane_vof_exchange {
    void *user;
    int (*all_gather)(void *user, const void *send, void *const *recv, size_t bytes);
    int (*sendrecv)(void *user, int n, const octane_vof_xfer *ops);
    char name[48];            /* e.g. "torch.distributed/nccl"; reported by octane_vof_mp_transport_info */
} octane_vof_exchange;
/* Between octane_vof_mp_create and octane_vof_mp_connect, on every rank or on none. */
int octane_vof_mp_set_exchange(octane_vof_mp *m, const octane_vof_exchange *ex);
/* The self-check of the process form: collective, after octane_vof_mp_connect.  `ag` all-gathers bytes_per_rank bytes per rank in
 * rank order (the host program's all-gather that also carried the IPC handles); the check builds a small group of its own with it. */
typedef int (*octane_allgather_bytes_fn)(void *user, const void *mine, void *all, size_t bytes_per_rank);
int octane_vof_mp_selfcheck(octane_vof_mp *m, octane_allgather_bytes_fn ag, void *user);

/* ---- patch matching (-sosm): the second flow method behind the reference's dispatch wrapper -------------------------
 * Per pixel: centre the search at the truncated, clamped first guess (u/v in), visit the (2 srad + 1)^2 displacements
 * in the reference's spiral order, keep the first strict minimum of the (2 rad + 1)^2 sum of squared differences (fp64),
 * refine each axis with a three-point parabola.  u/v out: displacement relative to the pixel (the first guess is not
 * added back -- as in the reference).  One channel, host buffers [ny][nx], blocking.  0 <= rad, srad <= 16.
 * Replaces oct_patch_match_optical_flow (src/oct_patch_match_optical_flow.cc:56), a CPU loop in the reference. */
int octane_sosm_run(const float *img1, const float *img2, int nx, int ny, float *u_inout, float *v_inout,
                    int rad, int srad, int device);

/* ---- pix2uv: pixel displacement -> navigated wind (cm/s as short) ---- */
typedef struct octane_nav {      /* the GOESNAVVar fields oct_pix2uv_cuda.cu reads (include/goesread.h) */
    double pph, req, rpol, lam0;
    float xScale, xOffset, yScale, yOffset, g2xOffset, g2yOffset;
    float lat1, lon1, lon0, R;
    int minX, minY;
    int nx, ny;
} octane_nav;

#define OCTANE_NAV_GEOS  0   /* GOES-R fixed grid (default)   */
#define OCTANE_NAV_POLAR 1   /* -Polar                         */
#define OCTANE_NAV_MERC  2   /* -Merc                          */
/* Or ONE of these into `mode`: which build of the navigation kernel runs.  nvcc's default -fmad=true builds the reference's kernel (ref
 * src/Makefile:9,20,27 set no -fmad flag), so every a * b + c of ref p2u:13-25,40-44,99-118 MAY be one fused operation in the
 * reference's binary.  Counted site by site on the oracle (tools/pix2uv_sites.py, profiles/r5_pix2uv_sites.txt; 13 sites, 50.7 M
 * shorts): only the two FLOAT multiply-adds of the base position (xi * xScale + xOffset, yi * yScale + yOffset, ref p2u:40-41,99-100)
 * move shorts -- 2.3 % of them, by 1 cm/s -- and the eleven double sites together move one short in 50.7 M.  Hence three builds:
 *   (none)                  strict: every product and sum rounded on its own -- the oracle's strict flavour;
 *   OCTANE_NAV_FMAD_FLOAT   strict + exactly the two float sites fused: what a -fmad=true CUDA build computes up to ~2e-8 of the
 *                           shorts, whatever a compiler contracts elsewhere -- the default of the oct_pix2uv_cuda C++ shim;
 *   OCTANE_NAV_FMAD         a * b + c fused wherever THIS compiler may (float and double; -ffp-contract=fast).
 * Which of them a given CUDA build of the reference equals cannot be checked here (no CUDA); a reference built with -fmad=false is
 * the first.  OCTANE_PIX2UV_FMAD=0|1|2 in the environment (strict | all | float sites) overrides mode and shim default alike. */
#define OCTANE_NAV_FMAD        0x100
#define OCTANE_NAV_FMAD_FLOAT  0x200
/* Host buffers.  pixuv != 0 reproduces -pd (ur/vr = (short)(100*u), ur2/vr2 untouched).
 * *sector_moved is set to 1 when the x/yOffset guard (oct_pix2uv_cuda.cu:295) zeroed the outputs. */
int octane_pix2uv_run(const octane_nav *nav, double t1, double t2, const float *u, const float *v,
                      int pixuv, int mode, short *ur, short *vr, short *ur2, short *vr2,
                      float *dT, int *sector_moved, int device);

/* ---- navcal: raw ABI counts -> navigated, calibrated, 0..255-normalised solver input ----
 * Replaces void oct_navcal_cuda(short*,short*,short*,short*,short*,short*,int,int,int,int,int,int,float*,float*,
 * float*,string,int,float x19,int,OFFlags), src/oct_navcal_cuda.cu:100; called from src/oct_fileread.cc. */
#define OCTANE_CAL_RAW  0
#define OCTANE_CAL_TEMP 1   /* Planck brightness temperature (fk1, fk2, bc1, bc2) */
#define OCTANE_CAL_REF  2   /* reflectance factor (kap1)                           */
#define OCTANE_CAL_BRIT 3
typedef struct octane_navcal_params {
    float xScale, xOffset, yScale, yOffset, radScale, radOffset;
    float rpol, req, H, lam0;                  /* H = perspective point height + req */
    float fk1, fk2, bc1, bc2, kap1;
    float maxin, minin, maxout, minout;        /* normalisation: [minin,maxin] -> [minout,maxout] */
    int cal, donav;
    int minx, maxx, miny, maxy;                /* output window [minx,maxx) x [miny,maxy) of the nx x ny frame */
} octane_navcal_params;
/* Host buffers.  data2: nx*ny raw counts; x: nx, y: ny scaled fixed-grid coordinates.  Outputs sized for the
 * window: data3/lat/lon/data2s (maxx-minx)*(maxy-miny), xs (maxx-minx), ys (maxy-miny). */
int octane_navcal_run(const short *data2, const short *x, const short *y, int nx, int ny,
                      const octane_navcal_params *p, float *data3, float *lat, float *lon,
                      short *data2s, short *xs, short *ys, int device);

/* Navigation of re-mapped polar (mode OCTANE_NAV_POLAR) and mercator (OCTANE_NAV_MERC) inputs: the float pixel values
 * pass through to data3 (window [minx,maxx) x [miny,maxy)), lat / lon (degrees) come from the inverse projection, data2s
 * is zero-filled and xs / ys receive the window's coordinate shorts.  lon0 / lat1 in DEGREES as the reference's callers
 * pass them (lat1 is unused for mercator).
 * Replaces oct_polar_navcal_cuda (src/oct_polar_navcal_cuda.cu:64) and oct_merc_navcal_cuda (src/oct_merc_navcal_cuda.cu:52). */
typedef struct octane_proj_navcal_params {
    float xScale, xOffset, yScale, yOffset, lon0, lat1, R;
    int donav, mode;
    int minx, maxx, miny, maxy;
} octane_proj_navcal_params;
int octane_proj_navcal_run(const float *data2, const short *x, const short *y, int nx, int ny,
                           const octane_proj_navcal_params *p, float *data3, float *lat, float *lon,
                           short *data2s, short *xs, short *ys, int device);
/* Per-band radiance range used for the normalisation when the user gives none (src/oct_normalize_geo.cc:9-88).
 * Returns 0, or OCTANE_E_INVALID for a band outside 1..16 (the reference leaves the outputs untouched then). */
int octane_bandminmax(int band, float *maxch, float *minch);

/* ---- optional steps next to the path ----
 * octane_uv2pix_run <- void oct_uv2pix(GOESVar&,float*,float*,double,OFFlags), src/oct_pix2uv_cuda.cu:372:
 *   first-guess winds u,v (m/s, in place -> pixel displacements) at lat/lon (degrees) of an nx x ny GOES fixed-grid
 *   frame whose scaled coordinates are gx[nx], gy[ny].  If the two frames' offsets differ (sector moved) or the
 *   projected point is off the disk, the displacement is 0.
 * octane_srsal_run  <- void oct_srsal_cu(float*,float*,float*,int,int,OFFlags), src/oct_srsal_cuda.cu:73:
 *   37x37 bilateral smoothing of u,v (in place) guided by the cloud-top-height image (sigma 9 px / 20 units). */
int octane_uv2pix_run(const octane_nav *nav, double t1, double t2, float *u_inout, float *v_inout,
                      const float *lat, const float *lon, const short *gx, const short *gy, int device);
int octane_srsal_run(float *u_inout, float *v_inout, const float *cth, int nx, int ny, int device);

const char *octane_last_error(void);
int octane_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
