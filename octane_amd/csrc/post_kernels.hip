// post_kernels.hip -- the two optional steps next to the flow path, gfx950:
//   k_uv2pix  first-guess winds (m/s) at known lat/lon -> pixel displacements   (ref src/oct_pix2uv_cuda.cu:222-263)
//   k_srsal   37x37 bilateral smoothing of the flow, guided by cloud-top height (ref src/oct_srsal_cuda.cu:35-71)
// Both are fp64 like the reference and built with -ffp-contract=off.
#include "vof_kernels.hpp"

namespace octane {

__global__ __launch_bounds__(256) void k_uv2pix(Uv2pixArgs A, const float *__restrict__ u, const float *__restrict__ v,
                                                const float *__restrict__ lat, const float *__restrict__ lon,
                                                const short *__restrict__ gx, const short *__restrict__ gy,
                                                float *__restrict__ upix, float *__restrict__ vpix)
{
    const double R = 6371000.0;
    const double pi = 3.14159265;
    const double rad = pi / 180.;
    const double H = A.pph + A.req;
    const long n = (long)A.nx * A.ny;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        const int i = (int)(k % A.nx), j = (int)(k / A.nx);
        double u1 = u[k], v1 = v[k];
        double latvalv = lat[k], lonvalv = lon[k];
        // great-circle displacement of the wind over `secs` (ref p2u:239-243)
        double dist = sqrt(pow(u1, 2.0) + pow(v1, 2.0)) * (A.secs);
        double brng = (180. + (90. - (atan2(-v1, -u1) / rad))) * rad;
        double latorig = latvalv * rad;
        latvalv = asin(sin(latorig) * cos(dist / R) + cos(latorig) * sin(dist / R) * cos(brng));
        lonvalv = lonvalv * rad + (atan2((sin(brng) * sin(dist / R) * cos(latorig)), (cos(dist / R) - sin(latorig) * sin(latvalv))));
        // forward fixed-grid projection (ref p2u:246-261)
        double thtc = atan(((A.rpol2) / (A.req2)) * tan(latvalv));
        double rc = A.rpol / sqrt(1. - (A.eval) * pow(cos(thtc), 2.));
        double sx = H - rc * cos(thtc) * cos(lonvalv - A.lam0);
        double sy = -rc * cos(thtc) * sin(lonvalv - A.lam0);
        double sz = rc * sin(thtc);
        double x1, y1;
        if ((H * (H - sx)) >= (sy * sy + ((A.req2) / (A.rpol2) * sz * sz))) {
            x1 = (asin(-sy / (sqrt(sx * sx + sy * sy + sz * sz))) - A.xoffset) / A.xscale;
            y1 = (atan(sz / sx) - A.yoffset) / A.yscale;
        } else {
            x1 = -999.;
            y1 = -999.;
        }
        if (x1 > -998.) {                        // ref p2u:447-454 (host loop there)
            upix[k] = (float)(x1 - gx[i]);
            vpix[k] = (float)(y1 - gy[j]);
        } else {
            upix[k] = 0.f;
            vpix[k] = 0.f;
        }
    }
}

void launch_uv2pix(hipStream_t s, const Uv2pixArgs &A, const float *u, const float *v, const float *lat, const float *lon,
                   const short *gx, const short *gy, float *upix, float *vpix)
{
    long n = (long)A.nx * A.ny, blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_uv2pix, dim3((unsigned)blocks), dim3(256), 0, s, A, u, v, lat, lon, gx, gy, upix, vpix);
}

// Bilateral filter.  One thread per pixel, taps in the reference's order (x offset outer, y offset inner) so the
// fp64 running sums round the same way; the (2*18+1)^2 = 1369 taps of a 32x8 pixel tile come from an LDS copy of
// the tile + 18-pixel reflected apron of u, v and the guide image instead of 1369 x 3 global gathers per pixel.
constexpr int kSrFs = 18;                   // filtsize = 2 * filtsigma(9), ref srsal:78-79
constexpr int kSrTX = 32, kSrTY = 8;
constexpr int kSrLW = kSrTX + 2 * kSrFs, kSrLH = kSrTY + 2 * kSrFs;

__device__ __forceinline__ int reflect(int x, int n)      // ref srsal:16-28 (differs from the solver's clamp)
{
    if (x < 0) x = 0 - x;
    if (x >= n) x = n - (x - n + 1);
    return x;
}

__global__ __launch_bounds__(256) void k_srsal(const float *__restrict__ u, const float *__restrict__ v,
                                               const float *__restrict__ cth, int nx, int ny, SrsalArgs A,
                                               float *__restrict__ uo, float *__restrict__ vo)
{
    __shared__ float s_u[kSrLH][kSrLW], s_v[kSrLH][kSrLW], s_c[kSrLH][kSrLW];
    const int tx0 = blockIdx.x * kSrTX, ty0 = blockIdx.y * kSrTY;
    for (int q = threadIdx.x; q < kSrLW * kSrLH; q += 256) {
        const int lx = q % kSrLW, ly = q / kSrLW;
        int gx = reflect(tx0 + lx - kSrFs, nx), gy = reflect(ty0 + ly - kSrFs, ny);
        gx = gx < 0 ? 0 : (gx >= nx ? nx - 1 : gx);          // frames narrower than the window: stay in bounds
        gy = gy < 0 ? 0 : (gy >= ny ? ny - 1 : gy);
        const long g = (long)gx + (long)gy * nx;
        s_u[ly][lx] = u[g]; s_v[ly][lx] = v[g]; s_c[ly][lx] = cth[g];
    }
    __syncthreads();
    const int lx = threadIdx.x % kSrTX, ly = threadIdx.x / kSrTX;
    const int ic = tx0 + lx, jc = ty0 + ly;
    if (ic >= nx || jc >= ny) return;
    const float pixc = s_c[ly + kSrFs][lx + kSrFs];
    double au = 0, av = 0, a2 = 0;
    for (int kc = 0; kc < 2 * kSrFs + 1; kc++) {
        const double gk = A.gk[kc];
        for (int lc = 0; lc < 2 * kSrFs + 1; lc++) {
            const float pixl = s_c[ly + lc][lx + kc];
            const double pixm = pixl - pixc;
            const double a1 = gk * A.gk[lc] * exp((pixm) * (pixm)*A.sigpix2);
            a2 += a1;
            au += (double)s_u[ly + lc][lx + kc] * a1;
            av += (double)s_v[ly + lc][lx + kc] * a1;
        }
    }
    const long o = (long)ic + (long)jc * nx;
    uo[o] = (float)(au / a2);
    vo[o] = (float)(av / a2);
}

void launch_srsal(hipStream_t s, const float *u, const float *v, const float *cth, int nx, int ny, const SrsalArgs &A,
                  float *uo, float *vo)
{
    dim3 g((nx + kSrTX - 1) / kSrTX, (ny + kSrTY - 1) / kSrTY);
    hipLaunchKernelGGL(k_srsal, g, dim3(256), 0, s, u, v, cth, nx, ny, A, uo, vo);
}

}  // namespace octane
