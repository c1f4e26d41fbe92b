/*
 * sosm_oracle.c -- CPU restatement of OCTANE's patch-matching ("-sosm", sum-of-squared-error minimisation) flow,
 * ref src/oct_patch_match_optical_flow.cc:12-156 ("ref pm").
 *
 * TEST INFRASTRUCTURE ONLY (see vof_oracle.c).  PINNED: the reference's own translation unit is plain C++ and is
 * compiled unmodified into oracle/_ref (oracle/Makefile `ref`, forwarding wrapper ref_wrap.cc); this restatement is
 * checked bit for bit against it (tests/test_oracle_pins.py) and against goldens generated from it
 * (tests/golden/ref_sosm.npz), which is what the GPU box -- where the reference does not exist -- uses.
 *
 * Quirks kept: the search-window test `(-SXD2 < n <= SXD2)` is a chained comparison (bool <= int), true for every
 * spiral position; the first guess only centres the search, it is not added back to the result; the strict `<` makes
 * the first minimum in spiral order win ties; sub-pixel refinement only when the minimum is strictly below both
 * neighbours; coordinates are clamped one by one (ref include/oct_bc.h).
 */
#include <math.h>
#include <stdlib.h>
#include "vof_oracle.h"

static int clampi(int x, int n) { if (x < 0) x = 0; if (x >= n) x = n - 1; return x; }   /* oct_bc<int> */

/* ref pm:12-34 jsose: images are [nx] columns of [ny] doubles there; value(i,j) = img[i + nx*j] widened */
static double sose(const float *g1, const float *g2, int i, int j, int n, int m, int nx, int ny, int rad)
{
    double s = 0;
    for (int k = 0; k < 2 * rad + 1; k++)
        for (int l = 0; l < 2 * rad + 1; l++) {
            int ic1 = clampi(i + k - rad, nx), jc1 = clampi(j + l - rad, ny);
            int ic2 = clampi(i + k + n - rad, nx), jc2 = clampi(j + l + m - rad, ny);
            double d = (double)g2[ic2 + (long)nx * jc2] - (double)g1[ic1 + (long)nx * jc1];
            s += d * d;
        }
    return s;
}

/* ref pm:36-55 jquad_interp */
static double quad_min(double y2, double y1, double y3, double x2, double x1, double x3)
{
    double C1 = (y2 - y1) / (x2 - x1);
    double C2 = (x2 * x2 - x1 * x1) / (x2 - x1);
    double a = (y3 - C1 * x3 - y1 + C1 * x1) / (x3 * x3 - C2 * x3 - x1 * x1 + C2 * x1);
    double b = C1 - a * C2;
    if (a == 0) return x2;
    return -b / (2. * a);
}

/* The spiral of ref pm:107-137 as a list: fills nm[2*count] with (n, m) of every position the window test admits,
 * in visiting order; returns count.  nm must hold 2*max(SX,SY)^2 ints. */
int oct_oracle_sosm_spiral(int srad, int *nm)
{
    const int SX = 2 * srad + 1, SY = 2 * srad + 1, SXD2 = SX / 2, SYD2 = SY / 2;
    const int big = SX > SY ? SX : SY;
    int n = 0, m = 0, dn = 0, dm = -1, count = 0;
    for (int ic = 0; ic < big * big; ic++) {
        if (((-SXD2 < n) <= SXD2) && ((-SYD2 < m) <= SYD2)) { nm[2 * count] = n; nm[2 * count + 1] = m; count++; }
        if ((n == m) || ((n < 0) && (n == -m)) || ((n > 0) && (n == 1 - m))) { int odn = dn; dn = -dm; dm = odn; }
        n += dn; m += dm;
    }
    return count;
}

void oct_oracle_sosm(const float *g1, const float *g2, float *u, float *v, int nx, int ny, int rad, int srad)
{
    const int big = 2 * srad + 1;
    int *nm = malloc(sizeof(int) * 2 * (size_t)big * big);
    const int count = oct_oracle_sosm_spiral(srad, nm);
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            long q = i + (long)nx * j;
            int ibc = clampi((int)(i + u[q]), nx);          /* float sum, truncated, clamped: ref pm:103-104 */
            int jbc = clampi((int)(j + v[q]), ny);
            double summin = 0; int nmin = 0, mmin = 0;
            for (int c = 0; c < count; c++) {
                double s = sose(g1, g2, ibc, jbc, nm[2 * c], nm[2 * c + 1], nx, ny, rad);
                if (c == 0 || s < summin) { summin = s; nmin = nm[2 * c]; mmin = nm[2 * c + 1]; }
            }
            double s1 = sose(g1, g2, ibc, jbc, nmin + 1, mmin, nx, ny, rad);
            double s2 = sose(g1, g2, ibc, jbc, nmin - 1, mmin, nx, ny, rad);
            if ((summin < s1) && (summin < s2))
                u[q] = (float)(quad_min(summin, s1, s2, (double)(i + nmin), (double)(i + nmin + 1), (double)(i + nmin - 1)) - (double)i);
            else
                u[q] = (float)nmin;
            s1 = sose(g1, g2, ibc, jbc, nmin, mmin + 1, nx, ny, rad);
            s2 = sose(g1, g2, ibc, jbc, nmin, mmin - 1, nx, ny, rad);
            if ((summin < s1) && (summin < s2))
                v[q] = (float)(quad_min(summin, s1, s2, (double)(j + mmin), (double)(j + mmin + 1), (double)(j + mmin - 1)) - (double)j);
            else
                v[q] = (float)mmin;
        }
    free(nm);
}
