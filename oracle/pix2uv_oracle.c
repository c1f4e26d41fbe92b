/*
 * pix2uv_oracle.c -- CPU restatement of OCTANE's pixel-displacement -> wind
 * navigation (src/oct_pix2uv_cuda.cu in /root/reference; "ref p2u:a-b" below).
 *
 * TEST INFRASTRUCTURE ONLY (see vof_oracle.c).  "parity unpinned": the
 * reference has no tests or golden vectors for this step and its translation
 * unit needs nvcc (kernel-launch chevrons + CUDA runtime), so it cannot be
 * built here.  The one recorded reference output (CONUS-like
 * navigation, u = 1.5 px, dt = 300 s -> U = 983 cm/s) is checked in
 * tests/test_oracle_pins.py.
 *
 * Arithmetic follows the reference expression by expression: the base pixel
 * position is formed in float (int*float+float, ref p2u:40-41,99-100), the
 * displaced one in double (ref p2u:43-44,102-103), latitude/longitude pass
 * through float at the haversine call (ref p2u:13,151,160), results are
 * truncated to short after x100 (ref p2u:196-197).  No FMA contraction is
 * assumed (built with -ffp-contract=off); nvcc's default -fmad could contract
 * the float multiply-add of the base position -- see DESIGN.md "pix2uv".
 */
#include <math.h>
#include "vof_oracle.h"

/* Round 5: WHICH multiply-add decides the shorts?  nvcc builds the reference's kernel with -fmad=true (its default,
 * ref src/Makefile:9,20,27 pass no -fmad=false), so any `a * b + c` below MAY be one fused operation in the reference's binary.  Each
 * such site is an explicit switch here: bit k of the site mask makes site k a fma()/fmaf(); with mask 0 every expression is the
 * two-rounding form this file has always had (the strict build's bits do not change).  tools/pix2uv_sites.py tables, per site, how
 * many shorts it moves (profiles/r5_pix2uv_sites.txt): only the two FLOAT sites of the base position do.
 *   bit  0  F1  xVal = xi * xScale + xOffset            float   ref p2u:40,76,99   (base position)
 *   bit  1  F2  yVal = yi * yScale + yOffset            float   ref p2u:41,77,100
 *   bit  2  D1  xv[0] * dt + xi                         double  ref p2u:43,79,102  (displaced position)
 *   bit  3  D2  (...) * xScale + xOffset                double
 *   bit  4  D3  xv[1] * dt + yi                         double  ref p2u:44,80,103
 *   bit  5  D4  (...) * yScale + yOffset                double
 *   bit  6  D5  sds = xVal * xVal + yVal * yVal         double  ref p2u:105
 *   bit  7  D6  cos^2 y + (req^2 / rpol^2) * sin^2 y    double  ref p2u:108        (only if pow(x, 2) is a product, as nvcc makes it)
 *   bit  8  D7  sin^2 x + cos^2 x * (...)               double  ref p2u:108
 *   bit  9  D8  c = H * H - req * req                   double  ref p2u:110
 *   bit 10  D9  d = b * b - 4 a c                       double  ref p2u:111
 *   bit 11  D10 e = (H - sx)^2 + sy^2                   double  ref p2u:118
 *   bit 12  H1  a = sin^2(..) + cos cos sin^2(..)       double  ref p2u:20         (haversine)
 * A build with -ffp-contract=fast (the "fma" flavour) additionally lets the compiler fuse what it likes: the site mask is for the
 * strict build. */
#define P2U_NSITES 13
static __thread unsigned g_sites = 0u;
void oct_oracle_pix2uv_fma_sites(unsigned mask) { g_sites = mask; }
int oct_oracle_pix2uv_nsites(void) { return P2U_NSITES; }
#define SITE(k) (g_sites & (1u << (k)))
static inline float maddf(int k, float a, float b, float c) { return SITE(k) ? fmaf(a, b, c) : a * b + c; }
static inline double madd(int k, double a, double b, double c) { return SITE(k) ? fma(a, b, c) : a * b + c; }

/* ref p2u:13-25 oct_haversine_cuda: arguments arrive as float */
static double great_circle(float lat1, float lon1, float lat2, float lon2, double rad, double rad2)
{
    const double earthrad = 6371000.00;
    double dlon = lon2 - lon1;
    double dlat = lat2 - lat1;
    double a = SITE(12) ? fma(cos(lat1 * rad) * cos(lat2 * rad), pow((sin(dlon * rad2)), 2), pow(sin(dlat * rad2), 2))
                        : (pow(sin(dlat * rad2), 2) + cos(lat1 * rad) * cos(lat2 * rad) * pow((sin(dlon * rad2)), 2));
    double c = 2. * atan2(sqrt(a), sqrt(1 - a));
    return earthrad * c;
}

/* ref p2u:27-172 oct_navpixel_uv_cuda */
static void navigate_pixel(const oct_oracle_nav *g, const double *rate, int xi, int yi, double dt,
                           double *r, double DTOR, double DTOR2, int mode)
{
    const double PI = 3.14159265359;
    double xVal, yVal;
    double latv[2], lonv[2], sds[2] = { 0., 0. };
    for (int iv = 0; iv < 2; ++iv) {
        if (iv == 0) {
            xVal = maddf(0, (float)(xi), g->xScale, g->xOffset);       /* float arithmetic */
            yVal = maddf(1, (float)(yi), g->yScale, g->yOffset);
        } else {
            xVal = madd(3, madd(2, rate[0], dt, xi), g->xScale, g->xOffset);   /* double arithmetic */
            yVal = madd(5, madd(4, rate[1], dt, yi), g->yScale, g->yOffset);
        }
        if (mode == 1) {                               /* polar, ref p2u:34-66 */
            double rho = sqrt(xVal * xVal + yVal * yVal);
            double c = asin(rho / g->R);
            if (g->lat1 > 89.9999) {
                lonv[iv] = g->lon0 * DTOR + atan2(xVal, -yVal);
            } else {
                lonv[iv] = g->lon0 * DTOR + atan2(xVal * sin(c), (rho * cos(g->lat1 * DTOR) * cos(c) - yVal * sin(g->lat1 * DTOR) * sin(c)));
            }
            if (rho > 0.0000001) {
                latv[iv] = asin(cos(c) * sin(g->lat1 * DTOR) + (yVal * sin(c) * cos(g->lat1 * DTOR) / rho));
            } else {
                latv[iv] = g->lat1 * DTOR;
            }
            latv[iv] = latv[iv] / DTOR;
            lonv[iv] = lonv[iv] / DTOR;
        } else if (mode == 2) {                        /* mercator, ref p2u:70-87 */
            latv[iv] = PI / 2. - 2. * atan(exp(-yVal / g->R));
            lonv[iv] = xVal / g->R + g->lon1;
            latv[iv] = latv[iv] / DTOR;
            lonv[iv] = lonv[iv] / DTOR;
        } else {                                       /* GOES fixed grid, ref p2u:90-139 */
            double H = g->pph + g->req;
            sds[iv] = SITE(6) ? fma(xVal, xVal, yVal * yVal) : xVal * xVal + yVal * yVal;
            double a = pow((sin(xVal)), 2) + pow(cos(xVal), 2) * (pow((cos(yVal)), 2) + (pow(g->req, 2)) / (pow(g->rpol, 2)) * pow((sin(yVal)), 2));
            if (SITE(7) || SITE(8)) {      /* the same sum with one or both of its product-sums fused (pow(x, 2) == x * x correctly rounded) */
                const double k2 = (pow(g->req, 2)) / (pow(g->rpol, 2)), s2y = pow((sin(yVal)), 2), c2y = pow((cos(yVal)), 2);
                const double inner = SITE(7) ? fma(k2, s2y, c2y) : c2y + k2 * s2y;
                a = SITE(8) ? fma(pow(cos(xVal), 2), inner, pow((sin(xVal)), 2)) : pow((sin(xVal)), 2) + pow(cos(xVal), 2) * inner;
            }
            double b = -2. * H * cos(xVal) * cos(yVal);
            double c = SITE(9) ? fma(H, H, -pow(g->req, 2)) : pow(H, 2) - pow(g->req, 2);
            double d = SITE(10) ? fma(b, b, -(4. * a * c)) : (pow(b, 2) - 4. * a * c);
            if (d >= 0) {
                double rs = (-b - sqrt(d)) / (2. * a);
                double sx = rs * cos(xVal) * cos(yVal);
                double sy = -rs * sin(xVal);
                double sz = rs * cos(xVal) * sin(yVal);
                double e = SITE(11) ? fma(H - sx, H - sx, pow(sy, 2)) : (pow((H - sx), 2) + pow(sy, 2));
                if (sz == 0 || e <= 0 || H - sx == 0) {
                    latv[iv] = -999.; lonv[iv] = -999.;
                } else {
                    latv[iv] = atan((pow(g->req, 2)) / (pow(g->rpol, 2)) * (sz / sqrt(e)));
                    lonv[iv] = g->lam0 - atan(sy / (H - sx));
                    latv[iv] = latv[iv] / DTOR;
                    lonv[iv] = lonv[iv] / DTOR;
                }
            } else {
                latv[iv] = -999.; lonv[iv] = -999.;
            }
        }
    }
    /* ref p2u:144-168 */
    if ((latv[0] < -998) || (latv[1] < -998) || (sds[0] > 0.021)) {
        r[0] = 0.; r[1] = 0.;
    } else {
        double dist = great_circle((float)latv[0], (float)lonv[0], (float)latv[0], (float)lonv[1], DTOR, DTOR2);
        r[0] = (lonv[1] >= lonv[0]) ? dist / dt : -dist / dt;
        dist = great_circle((float)latv[0], (float)lonv[0], (float)latv[1], (float)lonv[0], DTOR, DTOR2);
        r[1] = (latv[1] >= latv[0]) ? dist / dt : -dist / dt;
    }
}

/* ref p2u:265-370 oct_pix2uv_cuda (host) + p2u:173-221 octnavcalcuda (kernel) */
int oct_oracle_pix2uv(const oct_oracle_nav *nav, double t1, double t2, const float *u, const float *v,
                      int pixuv, int mode, short *ur, short *vr, short *ur2, short *vr2, float *dT)
{
    const long n = (long)nav->nx * nav->ny;
    double pi = 3.14159265;
    double rad = pi / 180.;
    double rad2 = rad / 2.;
    float dx = nav->xOffset - nav->g2xOffset, dy = nav->yOffset - nav->g2yOffset;
    int same_sector = (((double)(dx * dx) < (0.00001 * 0.00001)) && ((double)(dy * dy) < (0.00001 * 0.00001)));
    if (dT) *dT = (float)(t2 - t1);
    if (!same_sector) {                                /* ref p2u:358-368 */
        for (long k = 0; k < n; k++) { ur[k] = 0; vr[k] = 0; ur2[k] = 0; vr2[k] = 0; }
        return 1;
    }
    if (pixuv != 0) {                                  /* ref p2u:348-356: ur2/vr2 untouched */
        for (long k = 0; k < n; k++) { ur[k] = (short)(100 * u[k]); vr[k] = (short)(100 * v[k]); }
        return 0;
    }
    for (long k = 0; k < n; k++) {
        int ii = (int)(k % nav->nx), jj = (int)(k / nav->nx);
        double u1 = u[k], v1 = v[k];
        if (u1 > -9998.) {
            double rate[2], wind[2];
            rate[0] = u1 / (t2 - t1);
            rate[1] = v1 / (t2 - t1);
            navigate_pixel(nav, rate, ii + nav->minX, jj + nav->minY, t2 - t1, wind, rad, rad2, mode);
            ur[k] = (short)(100 * (wind[0]));
            vr[k] = (short)(100 * (wind[1]));
        } else {
            ur[k] = (short)(-32768);
            vr[k] = (short)(-32768);
        }
        ur2[k] = (short)(100 * u[k]);                  /* ref p2u:335-336 */
        vr2[k] = (short)(100 * v[k]);
    }
    return 0;
}
