// sosm_kernel.hip -- patch-matching flow ("-sosm": sum-of-squared-error minimisation), the second flow method behind
// the reference's dispatch wrapper (ref src/oct_optical_flow.cc:57-66).
//
// Behavioural spec: ref src/oct_patch_match_optical_flow.cc:12-156 ("ref pm"), a CPU loop there.  Per pixel: centre
// the search at the (truncated, clamped) first guess, visit the (2 srad + 1)^2 displacements in the reference's
// spiral order, keep the first strict minimum of the (2 rad + 1)^2 sum of squared differences (fp64, summed in the
// reference's k-outer / l-inner order so the bits match), then refine each axis with a three-point parabola when the
// minimum is strictly below both neighbours.  Coordinates are clamped one by one (oct_bc), so patches flatten at the
// frame edges exactly as in the reference.
//
// One thread per pixel.  The template form holds the pixel's (2 rad + 1)^2 patch of image 1 and the
// (2 (rad + srad) + 1)^2 window of image 2 in registers (25 + 81 floats for the default rad = srad = 2), so the 625
// squared differences of the search come from registers; the generic form reads through the cache.  No LDS tile: the
// window position depends on the per-pixel first guess.  Bound: fp64 VALU (1450 flop per pixel at the defaults).
#include "vof_kernels.hpp"
#include "device_util.hpp"

namespace octane {

__device__ __forceinline__ double quad_min(double y2, double y1, double y3, double x2, double x1, double x3)   // ref pm:36-55
{
    const double C1 = (y2 - y1) / (x2 - x1);
    const double C2 = (x2 * x2 - x1 * x1) / (x2 - x1);
    const double a = (y3 - C1 * x3 - y1 + C1 * x1) / (x3 * x3 - C2 * x3 - x1 * x1 + C2 * x1);
    const double b = C1 - a * C2;
    if (a == 0) return x2;
    return -b / (2. * a);
}

// ref pm:12-34 through the cache (used for the four refinement sums and by the generic kernel)
__device__ __forceinline__ double sose_mem(const float *__restrict__ g1, const float *__restrict__ g2, int i, int j, int n, int m,
                                           int nx, int ny, int rad)
{
    double s = 0;
    for (int k = 0; k < 2 * rad + 1; k++)
        for (int l = 0; l < 2 * rad + 1; l++) {
            const int ic1 = clampi(i + k - rad, 0, nx - 1), jc1 = clampi(j + l - rad, 0, ny - 1);
            const int ic2 = clampi(i + k + n - rad, 0, nx - 1), jc2 = clampi(j + l + m - rad, 0, ny - 1);
            const double d = (double)g2[ic2 + (size_t)nx * jc2] - (double)g1[ic1 + (size_t)nx * jc1];
            s += d * d;
        }
    return s;
}

__device__ __forceinline__ void sosm_finish(const float *__restrict__ g1, const float *__restrict__ g2, int i, int j, int ibc, int jbc,
                                            double summin, int nmin, int mmin, int nx, int ny, int rad,
                                            float *__restrict__ u, float *__restrict__ v, size_t q)
{
    double s1 = sose_mem(g1, g2, ibc, jbc, nmin + 1, mmin, nx, ny, rad);
    double s2 = sose_mem(g1, g2, ibc, jbc, nmin - 1, mmin, nx, ny, rad);
    if ((summin < s1) && (summin < s2))           // ref pm:142-147
        u[q] = (float)(quad_min(summin, s1, s2, (double)(i + nmin), (double)(i + nmin + 1), (double)(i + nmin - 1)) - (double)i);
    else
        u[q] = (float)nmin;
    s1 = sose_mem(g1, g2, ibc, jbc, nmin, mmin + 1, nx, ny, rad);
    s2 = sose_mem(g1, g2, ibc, jbc, nmin, mmin - 1, nx, ny, rad);
    if ((summin < s1) && (summin < s2))
        v[q] = (float)(quad_min(summin, s1, s2, (double)(j + mmin), (double)(j + mmin + 1), (double)(j + mmin - 1)) - (double)j);
    else
        v[q] = (float)mmin;
}

__global__ __launch_bounds__(256) void k_sosm_generic(const float *__restrict__ g1, const float *__restrict__ g2, float *__restrict__ u,
                                                      float *__restrict__ v, int nx, int ny, int rad, const int *__restrict__ spiral, int count)
{
    const long npix = (long)nx * ny;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npix; q += (long)gridDim.x * 256) {
        const int j = (int)(q / nx), i = (int)(q - (long)j * nx);
        const int ibc = clampi((int)(i + u[q]), 0, nx - 1);          // float sum, truncated, clamped: ref pm:103-104
        const int jbc = clampi((int)(j + v[q]), 0, ny - 1);
        double summin = 0.; int nmin = 0, mmin = 0;
        for (int c = 0; c < count; c++) {
            const int n = spiral[2 * c], m = spiral[2 * c + 1];
            const double s = sose_mem(g1, g2, ibc, jbc, n, m, nx, ny, rad);
            if (c == 0 || s < summin) { summin = s; nmin = n; mmin = m; }
        }
        sosm_finish(g1, g2, i, j, ibc, jbc, summin, nmin, mmin, nx, ny, rad, u, v, (size_t)q);
    }
}

template <int RAD, int SRAD>
__global__ __launch_bounds__(256) void k_sosm_fixed(const float *__restrict__ g1, const float *__restrict__ g2, float *__restrict__ u,
                                                    float *__restrict__ v, int nx, int ny, const int *__restrict__ spiral, int count)
{
    constexpr int P = 2 * RAD + 1, W = 2 * (RAD + SRAD) + 1, HW = RAD + SRAD;
    // 64 x 4 pixel tiles per workgroup pass: neighbouring lanes share cache lines of both images
    const int tiles_x = (nx + 63) / 64, tiles_y = (ny + 3) / 4;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
        const int i = (t % tiles_x) * 64 + lx, j = (t / tiles_x) * 4 + ly;
        if (i >= nx || j >= ny) continue;
        const size_t q = (size_t)i + (size_t)nx * j;
        const int ibc = clampi((int)(i + u[q]), 0, nx - 1);
        const int jbc = clampi((int)(j + v[q]), 0, ny - 1);
        float p1[P][P], w2[W][W];                 // [k][l] = (x offset, y offset), as the reference's loops run
#pragma unroll
        for (int l = 0; l < P; l++) {
            const size_t row = (size_t)nx * clampi(jbc + l - RAD, 0, ny - 1);
#pragma unroll
            for (int k = 0; k < P; k++) p1[k][l] = g1[row + clampi(ibc + k - RAD, 0, nx - 1)];
        }
#pragma unroll
        for (int l = 0; l < W; l++) {
            const size_t row = (size_t)nx * clampi(jbc + l - HW, 0, ny - 1);
#pragma unroll
            for (int k = 0; k < W; k++) w2[k][l] = g2[row + clampi(ibc + k - HW, 0, nx - 1)];
        }
        // The reference scans the displacements in spiral order and keeps the first strict minimum.  Here they are
        // evaluated in raster order (compile-time register indices) and ranked by their position in the spiral:
        // start from the spiral's first position, (0, 0), and let a candidate replace the incumbent when its sum is
        // smaller, or equal with an earlier rank -- the same winner, NaNs included (a NaN never replaces anything).
        double summin = 0.; int nmin = 0, mmin = 0, best_rank = 0;
#pragma unroll
        for (int k = 0; k < P; k++)
#pragma unroll
            for (int l = 0; l < P; l++) {
                const double d = (double)w2[k + SRAD][l + SRAD] - (double)p1[k][l];
                summin += d * d;
            }
#pragma unroll
        for (int dn = -SRAD; dn <= SRAD; dn++) {
#pragma unroll
            for (int dm = -SRAD; dm <= SRAD; dm++) {
                if (dn == 0 && dm == 0) continue;
                double s = 0;
#pragma unroll
                for (int k = 0; k < P; k++)
#pragma unroll
                    for (int l = 0; l < P; l++) {
                        const double d = (double)w2[k + dn + SRAD][l + dm + SRAD] - (double)p1[k][l];
                        s += d * d;
                    }
                const int rank = spiral[2 * count + (dn + SRAD) * (2 * SRAD + 1) + (dm + SRAD)];   // visiting index, -1 = never visited
                if (rank > 0 && (s < summin || (s == summin && rank < best_rank))) {
                    summin = s; nmin = dn; mmin = dm; best_rank = rank;
                }
            }
        }
        sosm_finish(g1, g2, i, j, ibc, jbc, summin, nmin, mmin, nx, ny, RAD, u, v, q);
    }
}

void launch_sosm(hipStream_t s, const float *g1, const float *g2, float *u, float *v, int nx, int ny, int rad, int srad,
                 const int *spiral, int count)
{
    if (rad == 2 && srad == 2) {
        const long tiles = (long)((nx + 63) / 64) * ((ny + 3) / 4);
        const int grid = (int)(tiles < 4096 ? (tiles < 1 ? 1 : tiles) : 4096);
        hipLaunchKernelGGL((k_sosm_fixed<2, 2>), dim3(grid), dim3(256), 0, s, g1, g2, u, v, nx, ny, spiral, count);
    } else {
        long blocks = ((long)nx * ny + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_sosm_generic, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, g1, g2, u, v, nx, ny, rad, spiral, count);
    }
}

}  // namespace octane
