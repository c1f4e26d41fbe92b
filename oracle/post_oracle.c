/*
 * post_oracle.c -- CPU restatements of the two optional steps next to the flow path:
 *   oct_oracle_uv2pix  ref src/oct_pix2uv_cuda.cu:222-263 (kernel octuv2xy) + :372-476 (host oct_uv2pix)
 *   oct_oracle_srsal   ref src/oct_srsal_cuda.cu:35-71 (kernel octsrsalcuda) + :73-147 (host oct_srsal_cu),
 *                      taps from oct_getGaussian_1D (ref src/oct_gaussian.cc:34-47)
 * TEST INFRASTRUCTURE ONLY (see vof_oracle.c); "parity unpinned" (the reference has no tests for them).
 */
#include <math.h>
#include "vof_oracle.h"

void oct_oracle_uv2pix(const oct_oracle_nav *nav, double t1, double t2, float *u, float *v,
                       const float *lat, const float *lon, const short *gx, const short *gy)
{
    const long n = (long)nav->nx * nav->ny;
    if (!((nav->xOffset == nav->g2xOffset) && (nav->yOffset == nav->g2yOffset))) {
        for (long k = 0; k < n; k++) { u[k] = 0.f; v[k] = 0.f; }
        return;
    }
    const double R = 6371000.0;
    const double pi = 3.14159265;
    double rad = pi / 180.;
    double secs = t2 - t1;
    double req = nav->req, rpol = nav->rpol, lam0 = nav->lam0;
    double req2 = req * req, rpol2 = rpol * rpol;
    double eval = sqrt((req2 - rpol2) / (req2));
    eval = eval * eval;
    double H = nav->pph + req;
    for (long k = 0; k < n; k++) {
        int i = (int)(k % nav->nx), j = (int)(k / nav->nx);
        double u1 = u[k], v1 = v[k];
        double latvalv = lat[k], lonvalv = lon[k];
        double dist = sqrt(pow(u1, 2.0) + pow(v1, 2.0)) * (secs);
        double brng = (180. + (90. - (atan2(-v1, -u1) / rad))) * rad;
        double latorig = latvalv * rad;
        latvalv = asin(sin(latorig) * cos(dist / R) + cos(latorig) * sin(dist / R) * cos(brng));
        lonvalv = lonvalv * rad + (atan2((sin(brng) * sin(dist / R) * cos(latorig)), (cos(dist / R) - sin(latorig) * sin(latvalv))));
        double thtc = atan(((rpol2) / (req2)) * tan(latvalv));
        double rc = rpol / sqrt(1. - (eval)*pow(cos(thtc), 2.));
        double sx = H - rc * cos(thtc) * cos(lonvalv - lam0);
        double sy = -rc * cos(thtc) * sin(lonvalv - lam0);
        double sz = rc * sin(thtc);
        double x1, y1;
        if ((H * (H - sx)) >= (sy * sy + ((req2) / (rpol2)*sz * sz))) {
            x1 = (asin(-sy / (sqrt(sx * sx + sy * sy + sz * sz))) - nav->xOffset) / nav->xScale;
            y1 = (atan(sz / sx) - nav->yOffset) / nav->yScale;
        } else {
            x1 = -999.; y1 = -999.;
        }
        if (x1 > -998.) { u[k] = (float)(x1 - gx[i]); v[k] = (float)(y1 - gy[j]); }
        else { u[k] = 0.f; v[k] = 0.f; }
    }
}

static int reflect_index(int x, int n)   /* ref srsal:16-28 */
{
    if (x < 0) x = 0 - x;
    if (x >= n) x = n - (x - n + 1);
    return x;
}

void oct_oracle_srsal(float *u, float *v, const float *cth, int nx, int ny, float *uo, float *vo)
{
    double sigpix = 20.;
    double sigpix2 = -1. / (sigpix * sigpix * 2.);
    double filtsigma = 9;
    int filtsize = (int)(2 * filtsigma);
    double gk[37];
    {
        double s = 2.0 * filtsigma * filtsigma, sum = 0.0;
        int wk2 = 18;
        for (int x = -wk2; x <= wk2; x++) { double r = x; gk[x + wk2] = (exp(-(r * r) / s)) / (M_PI * s); sum += gk[x + wk2]; }
        for (int i = 0; i < 37; ++i) gk[i] /= sum;
    }
    for (long q = 0; q < (long)nx * ny; q++) {
        int ic = (int)(q % nx), jc = (int)(q / nx);
        float pixc = cth[q];
        double au = 0, av = 0, a2 = 0;
        for (int kc = 0; kc < 2 * filtsize + 1; kc++) {
            for (int lc = 0; lc < 2 * filtsize + 1; lc++) {
                int ivc = reflect_index(ic + kc - filtsize, nx);
                int jvc = reflect_index(jc + lc - filtsize, ny);
                ivc = ivc < 0 ? 0 : (ivc >= nx ? nx - 1 : ivc);   /* frames narrower than the window stay in bounds */
                jvc = jvc < 0 ? 0 : (jvc >= ny ? ny - 1 : jvc);
                long q2 = (long)ivc + (long)jvc * nx;
                float pixl = cth[q2];
                double pixm = pixl - pixc;
                double a1 = gk[kc] * gk[lc] * exp((pixm) * (pixm)*sigpix2);
                a2 += a1;
                au += (double)u[q2] * a1;
                av += (double)v[q2] * a1;
            }
        }
        uo[q] = (float)(au / a2);
        vo[q] = (float)(av / a2);
    }
}
