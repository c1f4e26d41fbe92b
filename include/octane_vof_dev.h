/*
 * octane_vof_dev.h -- the DEVELOPER surface of the flow core: measurement hooks and the debug tap (exported by the product library,
 * liboctane_vof.so, because bench.py's roofline and the bit-exactness tests are taken ON the product), and -- under OCTANE_DIAG --
 * self-tests, probes, tuning switches and stamped kernel copies, which only the diagnostic library exports (liboctane_vof_diag.so,
 * `make -C octane_amd/csrc DIAG=1`: the same sources + -DOCTANE_DIAG=1; tools/ and the form-against-form tests load it, the product
 * never).  None of this has a reference counterpart: the reference's boundary is the two functions include/octane_vof.h starts with
 * (src/oct_optical_flow.cc:12-17).
 */
#ifndef OCTANE_VOF_DEV_H
#define OCTANE_VOF_DEV_H

#include "octane_vof.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement and the debug tap: exported by the product library ---------------------------------------------------------- */
/* Plans of 4 Mpixel and more allocate up to eight candidate arenas, time a few PCG iterations on each and keep the fastest (where an
 * arena lands in physical memory is worth up to 11 %).  Returns how many candidates were timed and copies up to `cap` of their times
 * (ms per PCG iteration at the finest level): a throughput measured on such a plan is a "best of n placements" figure. */
int octane_vof_plan_placement_trials(const octane_vof_plan *plan, double *ms, int cap);

/* Debug tap (NULL = off, zero cost): called on the host after each stage with a copy of the stage's planes: nplanes planes of ny rows
 * of nx floats.  Tags match oracle/vof_oracle.c's trace; "pcg_sums" (levels run one launch per iteration): per launch ten DOUBLES
 * delivered as 20 floats -- the launch's direct sums r.z r.r p.q q.z q.M^-1q r.q q.q, then the r.z it used, stopped, iterations. */
typedef void (*octane_vof_trace_fn)(void *user, const char *tag, int level, int gnc, int l,
                                    const float *data, int nx, int ny, int nplanes);
int octane_vof_plan_set_trace(octane_vof_plan *plan, octane_vof_trace_fn fn, void *user);

/* Per-kernel timing of the finest pyramid level, measured with HIP events on the run's stream. */
typedef struct octane_vof_profile {
    double pass_a_ms;  long long pass_a_launches;   /* the PCG iteration kernel (pass A of the two-pass form) */
    double pass_b_ms;  long long pass_b_launches;   /* pass B of the two-pass form (diagnostic library); 0 otherwise */
    double assemble_ms; long long assemble_launches;
    double update_ms;  long long update_launches;
    double setup_ms;                                  /* all level-setup kernels, every level */
    double total_ms;                                  /* whole run, first launch to last */
    long long finest_pixels;
} octane_vof_profile;
int octane_vof_plan_set_profiling(octane_vof_plan *plan, int enable);
int octane_vof_plan_get_profile(octane_vof_plan *plan, octane_vof_profile *out);
/* The finest-level PCG launches of the last profiled run, one duration (ms) each, in launch order (3 GNC steps x liters solves x
 * cgiters launches).  Launch k of a solve moves other bytes than launch k + 1 (x is updated by every second launch, the first GNC
 * step's weights are the constant -1 and are not read): bench.py prices each kind on its own bytes.  Returns the number recorded. */
int octane_vof_plan_get_launch_times(octane_vof_plan *plan, float *ms, int cap);

#ifdef OCTANE_DIAG
/* ---- diagnostic library only (liboctane_vof_diag.so): the product library does not export these --------------------------------- */
/* Developer knob, per plan: key in {overlap, persist, persist_p, persist_step, persist_max_g, persist_chain, lane_mode, small, small_max,
 * pass_a, max_blocks, reverse_b, xcd, nt, defer_x, unit_w, fused, fused_q, fused_rows, q_dma, asm_fast, trace_levels, q_diag,
 * persist_diag, persist_fault}.  "fused" 0 selects the two-pass form of the PCG iteration (pass A + pass B, direct sums), which only
 * this library contains.  Results agree for every setting to the last bits of the PCG scalars; only speed changes.  The ~30
 * OCTANE_TUNE_* environment variables of rounds 1-4 set the same knobs at plan creation, in this library only. */
int octane_vof_tune(octane_vof_plan *plan, const char *key, int value);
/* Which sub-domain grid the persistent mid-level solve (pcg_persist.hip) takes for a level of w x h pixels on a device with `ncu`
 * compute units: out5 = {columns, rows of sub-domains, rows per sub-domain, slots of 8 rows per thread, workgroups}.  Returns 1, or 0
 * when the level does not fit the device.  Host arithmetic only. */
int octane_vof_mid_geometry(int w, int h, int ncu, int *out5);
/* Does the LDS-DMA PCG kernel rotate the tile columns by the tile row on a level of w x rows pixels walked by `grid` workgroups (walk_mode:
 * the plan's tile walk, 4 by default)?  It does where that lowers the largest number of border-column tiles any one workgroup walks;
 * out3 = {tile columns, that maximum without / with the rotation} (-1: not counted).  Host arithmetic only. */
int octane_vof_row_rotation(int w, int rows, int grid, int walk_mode, int *out3);
/* Self-test of the three-instruction reciprocal (hardware estimate + one fused Newton step) against the IEEE division on every
 * positive normal float whose reciprocal is normal: out3 = {patterns compared, mismatches, one mismatching bit pattern}. */
int octane_selftest_rcp(int device, unsigned long long *out3);
/* Self-test of the assembly kernel's fast exact forms for one alpha (ref assembly .cu:611-1097: x / alpha, 1 / (s + 1), 1 / sqrt(x + 1e-6)
 * in double, rounded to float): out8 = {patterns, mismatches} x the three forms on EVERY float input, [6] = a mismatching bit pattern,
 * [7] = its test.  _bits: which forms (bit 0, 1, 2) plans with this alpha use -- plan creation runs the same check (~10 ms per alpha). */
int octane_selftest_assembly_math(int device, double alpha, unsigned long long *out8);
int octane_selftest_assembly_math_bits(int device, double alpha);
/* Time `iterations` (>= 2) PCG iterations of pyramid level `level` (0 = coarsest) in isolation, on whatever the plan's planes hold (the
 * stop test is held open, the planes are clobbered).  With the one-kernel iteration its time comes back in *pass_a_ms. */
int octane_vof_plan_probe(octane_vof_plan *plan, int level, int iterations, double *pass_a_ms, double *pass_b_ms);
/* Where the waves of one launch of the q-recomputing PCG kernel spend their time at pyramid level `level`: shader clock cycles summed
 * over all waves, per seam of a tile (out16[0..6]), tiles walked ([7]), prologue ([8]), epilogue ([9]).  The planes are clobbered. */
int octane_vof_plan_probe_stamps(octane_vof_plan *plan, int level, int even, int unit_w, unsigned long long *out16);
/* Cycles per seam of an iteration of the persistent mid-level solve (stamped build, tune "persist_diag"), summed over workgroups and
 * iterations since the last call: 32 values, [0..15] interior sub-domains, [16..31] the predicated ones.  Clears the counters. */
int octane_vof_mid_stamps(int device, unsigned long long *out32);
#endif  /* OCTANE_DIAG */

#ifdef __cplusplus
}
#endif
#endif
