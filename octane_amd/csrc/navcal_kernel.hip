// navcal_kernel.hip -- GOES-R fixed-grid navigation + calibration + 0..255 normalisation + limb taper, the step in
// front of the flow solver.  Behavioural spec: ref src/oct_navcal_cuda.cu:12-98 ("ref nav").
// One thread per pixel of the requested window; a pure stream (2 B in, 12 B out per pixel), fp64 where the
// reference's expressions are.  Built with -ffp-contract=off like the rest of the library.
//
// Promotion note: the kernel arguments req, rpol, H, lam0 are float in the reference; `pow(req,2)` is evaluated
// in double here, as ISO C++ (and the CPU oracle) do.  CUDA's pow(float,int) overload would return float; that
// affects lat/lon only (they are written to the output file, not used by the solver) and cannot be checked
// without CUDA.
#include "vof_kernels.hpp"

namespace octane {

__global__ __launch_bounds__(256) void k_navcal(NavcalArgs A, const short *__restrict__ x, const short *__restrict__ y,
                                                const short *__restrict__ data2, float *__restrict__ data3,
                                                float *__restrict__ lat, float *__restrict__ lon, short *__restrict__ data2s)
{
    const double PI = 3.14159265359;
    const double DTOR = PI / 180.;
    const int ww = A.maxx - A.minx, wh = A.maxy - A.miny;
    const long n2 = (long)ww * wh;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (long)gridDim.x * blockDim.x) {
        const int i = A.minx + (int)(q % ww), j = A.miny + (int)(q / ww);
        const long src = (long)i + (long)A.nx * j;
        const short raw = data2[src];
        data2s[q] = raw;                                          // ref nav:152 (host loop there)
        double xVal = x[i] * A.xScale + A.xOffset;                // float arithmetic, then widened (ref nav:31)
        double yVal = y[j] * A.yScale + A.yOffset;
        double subpoint_dist = xVal * xVal + yVal * yVal;
        float dVal = raw * A.radScale + A.radOffset;
        float la = 0.f, lo = 0.f;
        if (A.donav == 1) {                                       // ref nav:36-50
            double a = pow((sin(xVal)), 2.0) + pow(cos(xVal), 2.0) * (pow((cos(yVal)), 2.0) + (pow((double)A.req, 2.0)) / (pow((double)A.rpol, 2.0)) * pow((sin(yVal)), 2.0));
            double b = -2. * A.H * cos(xVal) * cos(yVal);
            double c = pow((double)A.H, 2.0) - pow((double)A.req, 2.0);
            double rs = (-b - sqrt((pow(b, 2.0) - 4. * a * c))) / (2. * a);
            double sx = rs * cos(xVal) * cos(yVal);
            double sy = -rs * sin(xVal);
            double sz = rs * cos(xVal) * sin(yVal);
            la = (float)atan(double((pow((double)A.req, 2.0)) / (pow((double)A.rpol, 2.0))) * (sz / sqrt((pow((A.H - sx), 2.0) + pow(sy, 2.0)))));
            lo = (float)(A.lam0 - atan(sy / (A.H - sx)));
            la = (float)(la / DTOR);
            lo = (float)(lo / DTOR);
        }
        lat[q] = la; lon[q] = lo;
        double dataF;
        if (A.cal == 1) dataF = (A.fk2 / (log((A.fk1 / dVal) + 1.)) - A.bc1) / A.bc2;      // brightness temperature
        else if (A.cal == 2) dataF = A.kap1 * dVal;                                        // reflectance factor
        else dataF = dVal;                                                                  // RAW / BRIT
        float sdsconst;                                           // limb taper, ref nav:81-91
        if (subpoint_dist < 0.021) sdsconst = 1.f;
        else if (subpoint_dist >= 0.0212) sdsconst = 0.f;
        else sdsconst = (float)(A.subpoint_slope * subpoint_dist + A.subpoint_int);
        data3[q] = (float)(sdsconst * (((dataF - A.minin) / (A.maxin - A.minin)) * (A.maxout - A.minout) + A.minout));   // ref nav:93
    }
}

// Navigation of re-mapped polar (mode 1, ref src/oct_polar_navcal_cuda.cu:11-62, "ref pnav") and mercator (mode 2,
// ref src/oct_merc_navcal_cuda.cu:11-50, "ref mnav") files: the pixel values pass through unchanged, lat / lon come
// from the inverse map projection.  Promotion points kept: x*xScale+xOffset in float then widened; sin / cos of the
// float lat1 are float (the C++ overloads), those of the double c are double; lat / lon are stored as float BEFORE
// the division by DTOR, which widens them again.  The polar branch `lat1 > 89.99999` compares the latitude in
// radians (the host passes lat1*DTOR), so it is never taken -- kept as written.
__global__ __launch_bounds__(256) void k_proj_navcal(ProjNavcalArgs A, const short *__restrict__ x, const short *__restrict__ y,
                                                     const float *__restrict__ data2, float *__restrict__ data3,
                                                     float *__restrict__ lat, float *__restrict__ lon)
{
    const double PI = 3.14159265359;
    const double DTOR = PI / 180.;
    const int ww = A.maxx - A.minx, wh = A.maxy - A.miny;
    const long n2 = (long)ww * wh;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (long)gridDim.x * blockDim.x) {
        const int i = A.minx + (int)(q % ww), j = A.miny + (int)(q / ww);
        const long src = (long)i + (long)A.nx * j;
        double xVal = x[i] * A.xScale + A.xOffset;
        double yVal = y[j] * A.yScale + A.yOffset;
        float la = 0.f, lo = 0.f;
        if (A.donav == 1) {
            if (A.mode == 1) {                                    // ref pnav:33-52
                const double rho = sqrt(xVal * xVal + yVal * yVal);
                const double c = asin(rho / A.R);
                if (A.lat1 > 89.99999) lo = (float)(A.lon0 + atan2(xVal, -yVal));
                else lo = (float)(A.lon0 + atan2(xVal * sin(c), (rho * cosf(A.lat1) * cos(c) - yVal * sinf(A.lat1) * sin(c))));
                if (rho > 0.0000001) la = (float)asin(cos(c) * sinf(A.lat1) + (yVal * sin(c) * cosf(A.lat1) / rho));
                else la = A.lat1;
            } else {                                              // ref mnav:30-34
                lo = (float)(xVal / A.R + A.lon0);
                la = (float)(PI / 2. - 2. * atan(exp(-yVal / A.R)));
            }
            la = (float)(la / DTOR);
            lo = (float)(lo / DTOR);
        }
        lat[q] = la; lon[q] = lo;
        data3[q] = data2[src];
    }
}

void launch_proj_navcal(hipStream_t s, const ProjNavcalArgs &A, const short *x, const short *y, const float *data2,
                        float *data3, float *lat, float *lon)
{
    const long n2 = (long)(A.maxx - A.minx) * (A.maxy - A.miny);
    long blocks = (n2 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_proj_navcal, dim3((unsigned)blocks), dim3(256), 0, s, A, x, y, data2, data3, lat, lon);
}

void launch_navcal(hipStream_t s, const NavcalArgs &A, const short *x, const short *y, const short *data2,
                   float *data3, float *lat, float *lon, short *data2s)
{
    const long n2 = (long)(A.maxx - A.minx) * (A.maxy - A.miny);
    long blocks = (n2 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_navcal, dim3((unsigned)blocks), dim3(256), 0, s, A, x, y, data2, data3, lat, lon, data2s);
}

}  // namespace octane
