/*
 * navcal_oracle.c -- CPU restatement of OCTANE's GOES-R navigation + calibration + normalisation step
 * (src/oct_navcal_cuda.cu:12-98 kernel, :100-207 host wrapper; "ref nav").
 *
 * TEST INFRASTRUCTURE ONLY (see vof_oracle.c).  "parity unpinned": the reference has no tests for this step and
 * the translation unit needs nvcc.  Promotions follow ISO C++ on the host: pow(float,int) is evaluated in
 * double (CUDA's device overload would return float; only lat/lon could differ).
 */
#include <math.h>
#include "vof_oracle.h"

void oct_oracle_navcal(const short *data2, const short *x, const short *y, int nx, int ny,
                       const oct_oracle_navcal_params *p, float *data3, float *lat, float *lon,
                       short *data2s, short *xs, short *ys)
{
    const double PI = 3.14159265359;
    const double DTOR = PI / 180.;
    const int ww = p->maxx - p->minx;
    const float subpoint_slope = (float)(1. / (0.021 - 0.0212));          /* ref nav:168 */
    const float subpoint_int = (float)(1. - 0.021 * (double)subpoint_slope);
    for (int j = p->miny; j < p->maxy; j++) {
        ys[j - p->miny] = y[j];
        for (int i = p->minx; i < p->maxx; i++) {
            long lxyz = (long)i + (long)nx * j;
            long lxyz2 = (long)(i - p->minx) + (long)ww * (j - p->miny);
            xs[i - p->minx] = x[i];
            data2s[lxyz2] = data2[lxyz];
            double xVal = x[i] * p->xScale + p->xOffset;                   /* float arithmetic first */
            double yVal = y[j] * p->yScale + p->yOffset;
            double subpoint_dist = xVal * xVal + yVal * yVal;
            float dVal = data2[lxyz] * p->radScale + p->radOffset;
            double dataF;
            if (p->donav == 1) {
                double a = pow((sin(xVal)), 2) + pow(cos(xVal), 2) * (pow((cos(yVal)), 2) + (pow(p->req, 2)) / (pow(p->rpol, 2)) * pow((sin(yVal)), 2));
                double b = -2. * p->H * cos(xVal) * cos(yVal);
                double c = pow(p->H, 2) - pow(p->req, 2);
                double rs = (-b - sqrt((pow(b, 2) - 4. * a * c))) / (2. * a);
                double sx = rs * cos(xVal) * cos(yVal);
                double sy = -rs * sin(xVal);
                double sz = rs * cos(xVal) * sin(yVal);
                lat[lxyz2] = (float)atan((double)((pow(p->req, 2)) / (pow(p->rpol, 2))) * (sz / sqrt((pow((p->H - sx), 2) + pow(sy, 2)))));
                lon[lxyz2] = (float)(p->lam0 - atan(sy / (p->H - sx)));
                lat[lxyz2] = (float)(lat[lxyz2] / DTOR);
                lon[lxyz2] = (float)(lon[lxyz2] / DTOR);
            } else {
                lat[lxyz2] = 0.f;
                lon[lxyz2] = 0.f;
            }
            if (p->cal == 1) dataF = (p->fk2 / (log((p->fk1 / dVal) + 1.)) - p->bc1) / p->bc2;
            else if (p->cal == 2) dataF = p->kap1 * dVal;
            else dataF = dVal;
            float sdsconst;
            if (subpoint_dist < 0.021) sdsconst = 1.f;
            else if (subpoint_dist >= 0.0212) sdsconst = 0.f;
            else sdsconst = (float)(subpoint_slope * subpoint_dist + subpoint_int);
            data3[lxyz2] = (float)(sdsconst * (((dataF - p->minin) / (p->maxin - p->minin)) * (p->maxout - p->minout) + p->minout));
        }
    }
}

/* Polar (mode 1, ref src/oct_polar_navcal_cuda.cu:11-62 kernel, :64-163 wrapper) and mercator (mode 2, ref
 * src/oct_merc_navcal_cuda.cu:11-50, :52-143) navigation of re-mapped inputs: pixel values pass through, lat / lon
 * from the inverse projection.  lon0 / lat1 arrive in degrees and are converted to float radians as the wrappers'
 * kernel arguments are.  The reference's `cos(lat1)` / `sin(lat1)` take a float and resolve to the float overloads
 * in C++; C has no overloads, hence cosf / sinf here.  `lat1 > 89.99999` compares radians: never true (kept). */
void oct_oracle_proj_navcal(const float *data2, const short *x, const short *y, int nx, int ny,
                            const oct_oracle_proj_navcal_params *p, float *data3, float *lat, float *lon,
                            short *data2s, short *xs, short *ys)
{
    const double PI = 3.14159265359;
    const double DTOR = PI / 180.;
    const int ww = p->maxx - p->minx;
    const float lon0 = (float)(p->lon0 * DTOR), lat1 = (float)(p->lat1 * DTOR);
    const float R = p->R;
    for (int j = p->miny; j < p->maxy; j++) {
        ys[j - p->miny] = y[j];
        for (int i = p->minx; i < p->maxx; i++) {
            long lxyz = (long)i + (long)nx * j;
            long lxyz2 = (long)(i - p->minx) + (long)ww * (j - p->miny);
            xs[i - p->minx] = x[i];
            data2s[lxyz2] = 0;
            double xVal = x[i] * p->xScale + p->xOffset;                   /* float arithmetic first */
            double yVal = y[j] * p->yScale + p->yOffset;
            float la = 0.f, lo = 0.f;
            if (p->donav == 1) {
                if (p->mode == 1) {
                    double rho = sqrt(xVal * xVal + yVal * yVal);
                    double c = asin(rho / R);
                    if (lat1 > 89.99999) lo = (float)(lon0 + atan2(xVal, -yVal));
                    else lo = (float)(lon0 + atan2(xVal * sin(c), (rho * cosf(lat1) * cos(c) - yVal * sinf(lat1) * sin(c))));
                    if (rho > 0.0000001) la = (float)asin(cos(c) * sinf(lat1) + (yVal * sin(c) * cosf(lat1) / rho));
                    else la = lat1;
                } else {
                    lo = (float)(xVal / R + lon0);
                    la = (float)(PI / 2. - 2. * atan(exp(-yVal / R)));
                }
                la = (float)(la / DTOR);
                lo = (float)(lo / DTOR);
            }
            lat[lxyz2] = la; lon[lxyz2] = lo;
            data3[lxyz2] = data2[lxyz];
        }
    }
}
