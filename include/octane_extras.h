/*
 * octane_extras.h -- C-ABI entry points OUTSIDE the hot path's scope table: methods SURVEY 2 marks out of scope that earlier
 * rounds built and the C++ shim (include/octane_host.hpp) still links -- the -sosm patch-matching flow and the navigation of re-mapped
 * polar / Mercator inputs.  Exported by liboctane_vof.so; kept apart from include/octane_vof.h so that the product surface of the path
 * reads as what it is.
 */
#ifndef OCTANE_EXTRAS_H
#define OCTANE_EXTRAS_H

#include "octane_vof.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- patch matching (-sosm) <- oct_patch_match_optical_flow, src/oct_patch_match_optical_flow.cc:56 (a CPU loop in the reference) ----
 * First guess in (centres the (2 srad + 1)^2 search), displacement relative to the pixel out; (2 rad + 1)^2 sum of squared differences in
 * fp64, the reference's spiral order and parabola refinement.  One channel, host buffers [ny][nx], blocking.  0 <= rad, srad <= 16. */
int octane_sosm_run(const float *img1, const float *img2, int nx, int ny, float *u_inout, float *v_inout,
                    int rad, int srad, int device);


/* Navigation of re-mapped polar / mercator inputs <- oct_polar_navcal_cuda (src/oct_polar_navcal_cuda.cu:64), oct_merc_navcal_cuda
 * (src/oct_merc_navcal_cuda.cu:52): pixel values pass through, lat / lon (degrees) from the inverse projection; lon0 / lat1 in DEGREES. */
typedef struct octane_proj_navcal_params {
    float xScale, xOffset, yScale, yOffset, lon0, lat1, R;
    int donav, mode;
    int minx, maxx, miny, maxy;
} octane_proj_navcal_params;
int octane_proj_navcal_run(const float *data2, const short *x, const short *y, int nx, int ny,
                           const octane_proj_navcal_params *p, float *data3, float *lat, float *lon,
                           short *data2s, short *xs, short *ys, int device);

#ifdef __cplusplus
}
#endif
#endif
