/*
 * octane_vof.h -- C-ABI of the MI355X-native dense variational optical-flow core: THE PRODUCT SURFACE.
 *
 * The drop-in boundary for OCTANE's hot path: plain pointers and sizes only; no C++ types, no torch types, no exceptions
 * cross it.  Every entry point returns 0 or a negative OCTANE_E_* code.  Library: liboctane_vof.so (octane_amd/csrc/Makefile).
 * The C++ shims with the reference's own signatures (Image / OFFlags / GOESVar) live in include/octane_host.hpp.
 * Measurement hooks, the debug tap, self-tests, probes and tuning switches are NOT here: include/octane_vof_dev.h.  Entry points of
 * methods SURVEY 2 marks out of scope (-sosm patch matching, polar / Mercator navigation): include/octane_extras.h.
 *
 * Reference interfaces replaced (paths relative to the reference repository; .cu = src/oct_variational_optical_flow.cu):
 *   octane_vof_run / _solve <- void oct_variational_optical_flow(Image,Image,float*,float*,float*,int,int,int,OFFlags), .cu:1213
 *                              (declared by its caller at src/oct_optical_flow.cc:12, called at :67)
 *   octane_vof_plan_*       <- the allocate / launch / free sequence of the same function (.cu:1268-1328, :1431, :1441-1472), split so
 *                              that a caller can keep device state across pairs
 *   octane_vof_batch_run, octane_vof_tiled_* / _mp_* <- (new) the reference is single-GPU (args.setdevice -> cudaSetDevice, .cu:1251-1265)
 *   octane_pix2uv_run       <- void oct_pix2uv_cuda(GOESVar&,double,float*,float*,short*,short*,short*,short*,OFFlags),
 *                              src/oct_pix2uv_cuda.cu:265 (declared at src/oct_optical_flow.cc:15, called at :91); the rest: cited below
 *
 * ENVIRONMENT.  The product library reads exactly these variables (none is needed): OCTANE_VOF_CACHE=0 (one-shot
 * entry: allocate per call), OCTANE_VOF_BANDS=n (C++ shim: n row bands), OCTANE_TILED_TRANSPORT / OCTANE_TILED_SELFCHECK=0 /
 * OCTANE_TUNE_MIN_BAND_PIXELS (row bands: force a transport / skip the first-contact check / banding threshold), OCTANE_MP_TIMEOUT_S
 * (process form: boundary time-out), OCTANE_TUNE_PERSIST_MAXG (processes SHARING a GPU: workgroups a persistent solve may hold),
 * OCTANE_PIX2UV_FMAD=0|1|2 (which build of the navigation kernel), and the bisect pair OCTANE_TUNE_Q_DMA=0 / OCTANE_TUNE_PERSIST=0 (the
 * finest levels' kernel without LDS-DMA / one launch per PCG iteration on the mid-size levels: the first things to switch off when a
 * result is in doubt).  Every other OCTANE_TUNE_* variable exists in the DIAGNOSTIC library only.
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
#define OCTANE_E_TOOSMALL   (-4)   /* some pyramid level would be narrower than 2 pixels (the reference indexes out of bounds there) */
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

void octane_vof_default_params(octane_vof_params *p);    /* the defaults of the reference CLI (src/main.cc:78-96,102) */

/* One-shot solve on host buffers: allocate device state, upload, solve, download, free (ref .cu:1213-1473).  img1/img2: nchan planes
 * of ny rows of nx floats (Image.data[i + nx*j + nx*ny*c]).  u_inout/v_inout: nx*ny floats each; in = first guess, out = flow in pixels
 * (.cu:1330-1335,1434-1438).  The caller keeps ownership of every pointer; inputs are not modified. */
int octane_vof_run(const float *img1, const float *img2, int nx, int ny, int nchan,
                   float *u_inout, float *v_inout, const octane_vof_params *p);
/* The same with the first guess apart from the result.  u0 = v0 = NULL: zero first guess (what oct_optical_flow passes without
 * -firstguess, ref src/oct_optical_flow.cc:38-48) -- nothing is uploaded for it. */
int octane_vof_solve(const float *img1, const float *img2, int nx, int ny, int nchan, const float *u0, const float *v0,
                     float *u_out, float *v_out, const octane_vof_params *p);
/* The one-shot entries keep the plan of their last call and reuse it for the same shape and parameters (an arena per pair costs up
 * to 0.5 s at 10848^2); this frees it.  OCTANE_VOF_CACHE=0 restores allocate-per-call, as the reference does (.cu:1268-1472). */
void octane_vof_release_cache(void);

/* ---- plan API: device state sized for (nx, ny, nchan, params), reused across pairs ---- */
typedef struct octane_vof_plan octane_vof_plan;
int octane_vof_plan_create(octane_vof_plan **plan, int nx, int ny, int nchan, const octane_vof_params *p);
int octane_vof_plan_destroy(octane_vof_plan *plan);
size_t octane_vof_plan_device_bytes(const octane_vof_plan *plan);

#define OCTANE_MEM_HOST   0
#define OCTANE_MEM_DEVICE 1
#define OCTANE_STREAM_OWN ((void *)(long)-1)   /* hip_stream value: the stream private to the plan */
/* Solve one pair.  mem says where img1/img2/u/v live; hip_stream is a hipStream_t.  OCTANE_MEM_DEVICE: the call only enqueues work
 * on hip_stream (NULL = HIP's null stream) and returns.  OCTANE_MEM_HOST: upload, solve, download, synchronise (NULL = the plan's
 * private stream).  OCTANE_STREAM_OWN selects the private stream for device buffers too (inputs complete at the call;
 * octane_vof_plan_wait tells when the outputs are): plans working side by side on one GPU should use it -- their private streams sit
 * on different hardware queues, two streams of a framework's pool may share one and then never overlap. */
int octane_vof_plan_run(octane_vof_plan *plan, const float *img1, const float *img2,
                        float *u_inout, float *v_inout, int mem, void *hip_stream);
/* The same with the first guess and the result in separate buffers; u0 = v0 = NULL: zero first guess. */
int octane_vof_plan_solve(octane_vof_plan *plan, const float *img1, const float *img2, const float *u0, const float *v0,
                          float *u_out, float *v_out, int mem, void *hip_stream);
/* Blocks until the plan's private stream and the stream of its last device-buffer run (the NULL stream too) are idle; the stream
 * handle given to a device-buffer run has to stay valid until this or octane_vof_plan_last_iterations has been called. */
int octane_vof_plan_wait(octane_vof_plan *plan);
/* PCG iterations of the last completed run (sum over all solves).  The mid-size levels are solved by ONE persistent launch each whose
 * workgroups must all be resident at once; if they cannot become so within 0.25 s (another process on this GPU) the solve is ABANDONED
 * instead of hanging the GPU and the library repairs the run itself: host-buffer runs solve the pair again before they return;
 * device-buffer runs are repaired by the first of octane_vof_plan_wait / _last_iterations called after the run (into THAT run's output
 * buffers) -- a device-buffer caller MUST call one of them per run before it uses the flow.  -2 only when the repair fails too.
 * octane_vof_plan_persist_state: 1 on, 0 off, -n off for the next n runs; and how often a solve was abandoned. */
long long octane_vof_plan_last_iterations(octane_vof_plan *plan);
int octane_vof_plan_persist_state(const octane_vof_plan *plan, int *abandoned_total);
/* How many OTHER plans work on this plan's device at the same time: 0 none (default); 2 one (the two lanes of octane_vof_batch_run:
 * persistent solves concurrent, each capped at half the compute units); 1 two or more (only the tiny levels keep the persistent
 * solve).  Results agree to the last bits of the PCG scalars (the fp64 partial sums follow the grid).  No reference counterpart. */
int octane_vof_plan_set_lane_mode(octane_vof_plan *plan, int mode);

/* Independent pairs sharded over GPUs: pair b runs on devices[b % ndevices]; one host thread per lane and device; no collective.
 * Pointer arrays have npairs entries of host buffers laid out as above. */
int octane_vof_batch_run(int npairs, const float *const *img1, const float *const *img2,
                         int nx, int ny, int nchan, float *const *u_inout, float *const *v_inout,
                         const octane_vof_params *p, int ndevices, const int *devices);

/* ---- one frame over several GPUs (BASELINE configs[3]: a full-disk pair as row bands; one global PCG per linearisation, .cu:1105-1195) ----
 * nbands (1..8) row bands, band b on devices[b] (NULL: b modulo the device count; ids may repeat -- several bands then share a device).
 * Levels with fewer than min_band_pixels pixels (0 = default, 4 Mpixel) are solved redundantly by every band; on the larger ones a band
 * owns a range of rows and the bands exchange, per PCG iteration, their reduction partials and a few rows per inner edge: the iterates
 * are the single-GPU solve's up to the summation order of the dot products.  _load (blocking), _solve (asynchronous), _wait, _fetch
 * (waits, then copies the flow out of band 0); _run = load + solve + fetch. */
typedef struct octane_vof_tiled octane_vof_tiled;
int octane_vof_tiled_create(octane_vof_tiled **out, int nx, int ny, int nchan, const octane_vof_params *p,
                            int nbands, const int *devices, long long min_band_pixels);
int octane_vof_tiled_destroy(octane_vof_tiled *t);
int octane_vof_tiled_load(octane_vof_tiled *t, const float *img1, const float *img2, const float *u, const float *v, int mem);
int octane_vof_tiled_solve(octane_vof_tiled *t);
int octane_vof_tiled_wait(octane_vof_tiled *t);
int octane_vof_tiled_fetch(octane_vof_tiled *t, float *u, float *v, int mem);
int octane_vof_tiled_run(octane_vof_tiled *t, const float *img1, const float *img2, float *u_inout, float *v_inout, int mem);
/* How `rows` rows are cut into nbands bands: edges[0..nbands] (inner edges at multiples of 32 rows); 1, or 0 = stays replicated. */
int octane_vof_band_partition(int rows, int nbands, int *edges);
int octane_vof_tiled_banded_levels(const octane_vof_tiled *t);
int octane_vof_tiled_band_rows(const octane_vof_tiled *t, int level, int band, int *y0, int *y1);   /* 1 banded, 0 replicated */
long long octane_vof_tiled_last_iterations(octane_vof_tiled *t);
long long octane_vof_tiled_last_copies(octane_vof_tiled *t);                   /* peer copies the last solve issued */
size_t octane_vof_tiled_device_bytes(const octane_vof_tiled *t);               /* per band */

/* ---- the same row-band solve with one band per PROCESS (the one-process-per-GPU launch) ----
 * Rank r of `world` (<= 8) owns band r on the device params->device selects.  Every step is collective: _create builds the band and
 * opens the POSIX shared-memory object shm_name (the same on every rank, unique per job: the ranks' phase barrier); _handles writes
 * OCTANE_MP_HANDLE_BYTES bytes (HIP IPC handles) which the host program all-gathers in rank order; _connect maps the other ranks'
 * allocations from them; _run: every rank passes the whole pair and the first guess (NULL, NULL = zero), the flow arrives on rank 0. */
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

/* ---- transports of the row-band solve and the first-contact self-check (no reference counterpart) ----
 * What crosses bands can travel three ways, same kernels, same order, same bits: INPLACE (the consuming kernel reads the neighbour's
 * memory through peer / IPC mappings, LDS-DMA included), COPY (stream-ordered runtime copies into local mirrors), COLLECTIVE (one band
 * per process: the HOST PROGRAM's collective library moves them through the two callbacks below -- torch.distributed / RCCL over xGMI
 * in octane_amd/exchange.py; north_star: "RCCL over xGMI only for halo exchange / result gather").  Unless OCTANE_TILED_TRANSPORT forces
 * one, creation solves a small frame under each candidate (inplace -> inplace with register staging -> copy -> collective) and keeps the
 * first that reproduces the plain plan (2e-5, equal iteration counts); missing peer access / IPC removes candidates. */
#define OCTANE_TRANSPORT_INPLACE    0
#define OCTANE_TRANSPORT_COPY       1
#define OCTANE_TRANSPORT_COLLECTIVE 2
typedef struct octane_vof_transport_info {
    int transport;            /* OCTANE_TRANSPORT_* the plan uses */
    int q_dma;                /* 1: q-form band launches fetch the neighbour's rows by LDS-DMA; 0: register-staged kernel */
    int selfcheck;            /* 0 not run (one band, forced, switched off); 1 the first candidate passed; 2 downgraded; -1 none passed */
    int candidates_tried;
    int forced;               /* OCTANE_TILED_TRANSPORT named the transport */
    int peer_ok;              /* every pair of distinct devices has peer access / every IPC mapping opened */
    int ndevices;             /* distinct devices among the bands */
    int nbands;
    double check_rel_l2[4];   /* distance to the plain plan of candidate i of the self-check (-1: not tried) */
    char exchange[48];        /* name of the host program's collective library when one was registered, else "" */
} octane_vof_transport_info;
int octane_vof_tiled_transport_info(const octane_vof_tiled *t, octane_vof_transport_info *out);
int octane_vof_mp_transport_info(const octane_vof_mp *m, octane_vof_transport_info *out);
const char *octane_vof_transport_name(int transport);
/* The collective transport's callbacks.  Buffers are device memory of the calling rank's GPU; the library has drained its stream before
 * a call, a call returns when its data is in place.  all_gather: rank c's `bytes` bytes arrive at recv[c] (recv[own] is NULL).
 * sendrecv: n transfers, all posted before any is waited for; between two ranks the k-th send matches the k-th receive.  0 = success. */
typedef struct octane_vof_xfer { int peer; int send; void *buf; size_t bytes; } octane_vof_xfer;
typedef struct octane_vof_exchange {
    void *user;
    int (*all_gather)(void *user, const void *send, void *const *recv, size_t bytes);
    int (*sendrecv)(void *user, int n, const octane_vof_xfer *ops);
    char name[48];            /* e.g. "torch.distributed/nccl" */
} octane_vof_exchange;
int octane_vof_mp_set_exchange(octane_vof_mp *m, const octane_vof_exchange *ex);   /* between _create and _connect, on every rank or none */
/* The self-check of the process form: collective, after _connect; `ag` all-gathers bytes_per_rank bytes per rank in rank order. */
typedef int (*octane_allgather_bytes_fn)(void *user, const void *mine, void *all, size_t bytes_per_rank);
int octane_vof_mp_selfcheck(octane_vof_mp *m, octane_allgather_bytes_fn ag, void *user);

/* ---- pix2uv: pixel displacement -> navigated wind (cm/s as short) <- oct_pix2uv_cuda, src/oct_pix2uv_cuda.cu:265 ---- */
typedef struct octane_nav {      /* the GOESNAVVar fields oct_pix2uv_cuda.cu reads (include/goesread.h) */
    double pph, req, rpol, lam0;
    float xScale, xOffset, yScale, yOffset, g2xOffset, g2yOffset;
    float lat1, lon1, lon0, R;
    int minX, minY;
    int nx, ny;
} octane_nav;
#define OCTANE_NAV_GEOS  0   /* GOES-R fixed grid (default) */
#define OCTANE_NAV_POLAR 1   /* -Polar */
#define OCTANE_NAV_MERC  2   /* -Merc */
/* Or ONE of these into `mode`: which build of the navigation kernel runs.  nvcc's default -fmad=true builds the reference's kernel (ref
 * src/Makefile:9,20,27), so every a * b + c of ref p2u:13-25,40-44,99-118 MAY be fused there; counted site by site on the oracle
 * (profiles/r5_pix2uv_sites.txt) only the two FLOAT multiply-adds of the base position (ref p2u:40-41,99-100) move shorts (2.3 %, by
 * 1 cm/s).  (none): strict; OCTANE_NAV_FMAD_FLOAT: strict + exactly those two sites fused (the oct_pix2uv_cuda shim's default);
 * OCTANE_NAV_FMAD: fused wherever THIS compiler may.  OCTANE_PIX2UV_FMAD=0|1|2 (strict | all | float sites) overrides both. */
#define OCTANE_NAV_FMAD        0x100
#define OCTANE_NAV_FMAD_FLOAT  0x200
/* Host buffers.  pixuv != 0 reproduces -pd (ur/vr = (short)(100*u), ur2/vr2 untouched).  *sector_moved is set to 1 when the
 * x/yOffset guard (oct_pix2uv_cuda.cu:295) zeroed the outputs. */
int octane_pix2uv_run(const octane_nav *nav, double t1, double t2, const float *u, const float *v,
                      int pixuv, int mode, short *ur, short *vr, short *ur2, short *vr2,
                      float *dT, int *sector_moved, int device);

/* ---- navcal: raw ABI counts -> navigated, calibrated, 0..255-normalised solver input
 * <- void oct_navcal_cuda(short* x6, int x6, float* x3, string, int, float x19, int, OFFlags), src/oct_navcal_cuda.cu:100 ---- */
#define OCTANE_CAL_RAW  0
#define OCTANE_CAL_TEMP 1   /* Planck brightness temperature (fk1, fk2, bc1, bc2) */
#define OCTANE_CAL_REF  2   /* reflectance factor (kap1) */
#define OCTANE_CAL_BRIT 3
typedef struct octane_navcal_params {
    float xScale, xOffset, yScale, yOffset, radScale, radOffset;
    float rpol, req, H, lam0;                  /* H = perspective point height + req */
    float fk1, fk2, bc1, bc2, kap1;
    float maxin, minin, maxout, minout;        /* normalisation: [minin,maxin] -> [minout,maxout] */
    int cal, donav;
    int minx, maxx, miny, maxy;                /* output window [minx,maxx) x [miny,maxy) of the nx x ny frame */
} octane_navcal_params;
int octane_navcal_run(const short *data2 /* nx*ny raw counts */, const short *x /* nx */, const short *y /* ny scaled coordinates */, int nx, int ny,
                      const octane_navcal_params *p, float *data3, float *lat, float *lon,      /* host buffers; outputs sized for the window */
                      short *data2s, short *xs, short *ys, int device);
/* Per-band radiance range of the normalisation (src/oct_normalize_geo.cc:9-88).  OCTANE_E_INVALID for a band outside 1..16. */
int octane_bandminmax(int band, float *maxch, float *minch);

/* ---- optional steps next to the path ----
 * octane_uv2pix_run <- oct_uv2pix(GOESVar&,float*,float*,double,OFFlags), src/oct_pix2uv_cuda.cu:372: first-guess winds (m/s, in place
 *   -> pixel displacements) at lat/lon of an nx x ny fixed-grid frame with scaled coordinates gx[nx], gy[ny].
 * octane_srsal_run  <- oct_srsal_cu(float*,float*,float*,int,int,OFFlags), src/oct_srsal_cuda.cu:73: 37x37 bilateral smoothing in place. */
int octane_uv2pix_run(const octane_nav *nav, double t1, double t2, float *u_inout, float *v_inout,
                      const float *lat, const float *lon, const short *gx, const short *gy, int device);
int octane_srsal_run(float *u_inout, float *v_inout, const float *cth, int nx, int ny, int device);

const char *octane_last_error(void);    /* text of the calling thread's last error */
int octane_device_count(void);          /* ref cudaGetDeviceCount, .cu:1251 */
#ifdef __cplusplus
}
#endif
#endif
