// zoom_host.cpp -- host-side resampling of a calibrated channel onto the grid of channel 1, what the readers need for
// -ic21/-ic22/-ic31/-ic32 (ref src/oct_fileread.cc:365-383 calls them on the CPU too).
//
// Behavioural spec: ref src/oct_zoom.cc:12-16 (oct_zoom_size), :51-88 (oct_zoom_out_float), :180-222
// (oct_zoom_in_float) with their helpers ref src/oct_bicubic.cc:10-33,97-150 (Catmull-Rom cell on clamped taps, the
// fraction taken against the *clamped* integer position) and ref src/oct_gaussian.cc:34-104 (taps normalised over
// 2 fs + 1 entries, the last one never applied; rows first, then columns; clamped borders).  Same signatures, same
// arithmetic in the same order: outputs are bit-identical to the reference's own functions compiled here
// (tests/test_io_zoom.py against tests/golden/ref_helpers.npz).
//
// Two deliberate differences, both where the reference's behaviour is undefined:
//  * oct_zoom_out_float stores channel `cnum` at offset cnum (ref oct_zoom.cc:85, `+cnum` where `+cnumt` was computed
//    two lines above): for cnum > 0 that overwrites channel 0 shifted by cnum pixels and leaves the channel's own plane
//    unwritten.  Here the channel goes to its plane, cnum * nxx * nyy; for cnum == 0 the two agree.
//  * the nearest-neighbour branch of oct_zoom_in_float (interp < 1, not used by the readers) indexes the source without
//    clamping and can read one element past a row; the index is clamped here.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/octane_host.hpp"

namespace {

inline int clamp_index(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }          // ref include/oct_bc.h

// ref oct_bicubic.cc:10-18
inline double cubic(const double v[4], double x)
{
    return v[1] + 0.5 * x * (v[2] - v[0] + x * (2.0 * v[0] - 5.0 * v[1] + 4.0 * v[2] - v[3] + x * (3.0 * (v[1] - v[2]) + v[3] - v[0])));
}

// ref oct_bicubic.cc:36-95 / :97-150: the same routine on a double or a float image
template <class T>
double bicubic_at(const T *img, double uu, double vv, int nx, int ny)
{
    const int xs[4] = {clamp_index((int)(uu - 1), nx), clamp_index((int)uu, nx), clamp_index((int)(uu + 1), nx), clamp_index((int)(uu + 2), nx)};
    const int ys[4] = {clamp_index((int)(vv - 1), ny), clamp_index((int)vv, ny), clamp_index((int)(vv + 1), ny), clamp_index((int)(vv + 2), ny)};
    const double fx = uu - xs[1], fy = vv - ys[1];
    double col[4];
    for (int i = 0; i < 4; i++) {             // one column of four rows at a time, interpolated along y first
        const double taps[4] = {(double)img[xs[i] + nx * ys[0]], (double)img[xs[i] + nx * ys[1]],
                                (double)img[xs[i] + nx * ys[2]], (double)img[xs[i] + nx * ys[3]]};
        col[i] = cubic(taps, fy);
    }
    const double f = cubic(col, fx);
    if (f != f) {                              // ref oct_bicubic.cc:88-93
        printf("Bicubic failure, possible data issue\n");
        exit(0);
    }
    return f;
}

// ref oct_gaussian.cc:34-47 and :49-104
void blur_in_place(std::vector<double> &img, int nx, int ny, double sigma)
{
    int fs = (int)(2 * sigma);
    if (fs < 5) fs = 5;
    const int wk = 2 * fs + 1;
    std::vector<double> gk(wk);
    const double s = 2.0 * sigma * sigma;
    double sum = 0.0;
    for (int x = -fs; x <= fs; x++) {
        const double r = x;
        gk[x + fs] = (exp(-(r * r) / s)) / (M_PI * s);
        sum += gk[x + fs];
    }
    for (int i = 0; i < wk; i++) gk[i] /= sum;
    std::vector<double> tmp((size_t)nx * ny);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            double w = 0;
            for (int k = -fs; k < fs; k++) w = w + gk[k + fs] * img[clamp_index(i + k, nx) + (size_t)nx * j];
            tmp[i + (size_t)nx * j] = w;
        }
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            double w = 0;
            for (int k = -fs; k < fs; k++) w = w + gk[k + fs] * tmp[i + (size_t)nx * clamp_index(j + k, ny)];
            img[i + (size_t)nx * j] = w;
        }
}

}  // namespace

void oct_zoom_size(int nx, int ny, int &nxx, int &nyy, double factor)
{
    nxx = (int)((double)nx * factor + 0.5);
    nyy = (int)((double)ny * factor + 0.5);
}

void oct_zoom_out_float(float *image, float *imageout, int nx, int ny, double factor, int verb, int cnum)
{
    if (verb == 1) exit(0);                                        // ref oct_zoom.cc:61
    int nxx, nyy;
    oct_zoom_size(nx, ny, nxx, nyy, factor);
    float *out = imageout + (size_t)cnum * nxx * nyy;              // see the header: the reference adds cnum only
    if (!(factor < 0.999999)) {                                    // same size: a copy (ref :78-83); the blur is never used
        for (int jj = 0; jj < nyy; jj++)
            for (int ii = 0; ii < nxx; ii++) out[ii + (size_t)nxx * jj] = image[ii + (size_t)nxx * jj];
        return;
    }
    std::vector<double> is((size_t)nx * ny);
    for (size_t i = 0; i < is.size(); i++) is[i] = image[i];
    const double sigma = 0.6 * sqrt(1.0 / (factor * factor) - 1.0);
    blur_in_place(is, nx, ny, sigma);
    for (int jj = 0; jj < nyy; jj++)
        for (int ii = 0; ii < nxx; ii++) {
            const double i2 = (double)ii / factor, j2 = (double)jj / factor;
            out[ii + (size_t)nxx * jj] = (float)bicubic_at(is.data(), i2, j2, nx, ny);
        }
}

void oct_zoom_in_float(float *flow, float *flowout, int nx, int ny, int nxx, int nyy, int cnum, int interp)
{
    const float factorx = ((float)nxx / nx), factory = ((float)nyy / ny);
    const float val1 = (float)(0.5 - 0.5 / factory), val2 = (float)(0.5 - 0.5 / factorx);     // half-pixel shift, ref :186-187
    float *out = flowout + (size_t)cnum * ((size_t)nxx * nyy);
    for (int jj1 = 0; jj1 < nyy; jj1++) {
        const float j2 = (float)((jj1 / factory) - val1);
        for (int i1 = 0; i1 < nxx; i1++) {
            const float i2 = (float)((i1 / factorx) - val2);
            float g;
            if (interp == 1) g = (float)bicubic_at(flow, i2, j2, nx, ny);
            else g = flow[clamp_index((int)(i2 + 0.5), nx) + (size_t)nx * clamp_index((int)(j2 + 0.5), ny)];
            out[i1 + (size_t)nxx * jj1] = g;
        }
    }
}
