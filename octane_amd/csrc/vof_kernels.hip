// vof_kernels.hip -- level-setup and assembly kernels of the variational flow solver, gfx950.
//
// Behavioural spec: src/oct_variational_optical_flow.cu of the reference ("ref .cu:a-b").
// Nothing here is a translation of that file's structure: the reference runs one
// cooperative mega-kernel with grid-stride loops over a flat index and a grid barrier between
// phases; here every phase is its own stream-ordered launch over 2-D tiles of row-pitched
// planes, the blur is evaluated only where the decimation samples it, and the linear
// operator is kept as five coefficient planes instead of a CSR matrix.
#include "vof_kernels.hpp"
#include "device_util.hpp"
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>

namespace octane {

// ---------------------------------------------------------------------------------------
// plain 2-D copies (pitch conversion, level bookkeeping: ref .cu:506-516, 578-582)
// ---------------------------------------------------------------------------------------
__global__ void k_copy2d(const float *__restrict__ src, int spitch, float *__restrict__ dst, int dpitch,
                         int w, int h, float scale, int do_scale)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float v = src[(size_t)y * spitch + x];
    if (do_scale) v = v * scale;
    dst[(size_t)y * dpitch + x] = v;
}

void launch_copy2d(hipStream_t s, const float *src, int spitch, float *dst, int dpitch, int w, int h)
{
    dim3 b(64, 4), g((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_copy2d, g, b, 0, s, src, spitch, dst, dpitch, w, h, 1.0f, 0);
}

void launch_scale_copy2d(hipStream_t s, const float *src, int spitch, float *dst, int dpitch, int w, int h, float scale)
{
    dim3 b(64, 4), g((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_copy2d, g, b, 0, s, src, spitch, dst, dpitch, w, h, scale, 1);
}

// ---------------------------------------------------------------------------------------
// Gaussian blur + decimation (ref .cu:311-408, 521-563)
//
// The reference blurs the whole full-resolution image (rows, then columns) and then samples
// it at integer coordinates ((int)(ii/f), (int)(jj/f)) -- its bicubic collapses to a point
// sample there.  Only those samples are needed, so the row pass is evaluated at the sampled
// columns only (output dw x sh) and the column pass at the sampled rows only (dw x dh).
// Tap order (-fs .. fs-1, last tap dropped) and the running-sum order are the reference's.
// ---------------------------------------------------------------------------------------
__global__ void k_blur_rows_sampled(const float *__restrict__ src, int sw, int sh, int spitch,
                                    float *__restrict__ dst, int dw, int dpitch,
                                    const float *__restrict__ gk, int fs, float factor)
{
    int ii = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y;
    if (ii >= dw || j >= sh) return;
    int xc = clampi((int)((float)ii / factor), 0, sw - 1);
    const float *row = src + (size_t)j * spitch;
    float acc = 0.f;
    if (xc - fs >= 0 && xc + fs - 1 <= sw - 1) {
        // No tap is clamped: the lane's 2 fs taps are consecutive pixels, fetched four at a time (the lanes of a wave sit 1 / factor
        // pixels apart, so every lane touches lines of its own whatever the width of the load: a quarter of the load instructions for
        // the same lines).  Same products, same running sum, in the same order.
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const float *p = row + (xc - fs);
        int t = 0;
        for (; t + 4 <= 2 * fs; t += 4) {
            const f4u v = *reinterpret_cast<const f4u *>(p + t);
            acc = acc + gk[t] * v.x;
            acc = acc + gk[t + 1] * v.y;
            acc = acc + gk[t + 2] * v.z;
            acc = acc + gk[t + 3] * v.w;
        }
        for (; t < 2 * fs; ++t) acc = acc + gk[t] * p[t];
    } else {
        for (int t = -fs; t < fs; ++t) {
            int sx = clampi(xc + t, 0, sw - 1);
            acc = acc + gk[t + fs] * row[sx];
        }
    }
    dst[(size_t)j * dpitch + ii] = acc;
}

__global__ void k_blur_cols_sampled(const float *__restrict__ src, int sw, int sh, int spitch,
                                    float *__restrict__ dst, int dh, int dpitch,
                                    const float *__restrict__ gk, int fs, float factor,
                                    float postscale, int do_scale)
{
    int ii = blockIdx.x * blockDim.x + threadIdx.x;
    int jj = blockIdx.y;
    if (ii >= sw || jj >= dh) return;
    int yc = clampi((int)((float)jj / factor), 0, sh - 1);
    float acc = 0.f;
    for (int t = -fs; t < fs; ++t) {
        int sy = clampi(yc + t, 0, sh - 1);
        acc = acc + gk[t + fs] * src[(size_t)sy * spitch + ii];
    }
    if (do_scale) acc = acc * postscale;   // ref .cu:559-563, a separate multiply
    dst[(size_t)jj * dpitch + ii] = acc;
}

void launch_blur_rows_sampled(hipStream_t s, const float *src, int sw, int sh, int spitch,
                              float *dst, int dw, int dpitch, const float *gk, int fs, float factor)
{
    dim3 b(64), g((dw + 63) / 64, sh);
    hipLaunchKernelGGL(k_blur_rows_sampled, g, b, 0, s, src, sw, sh, spitch, dst, dw, dpitch, gk, fs, factor);
}

void launch_blur_cols_sampled(hipStream_t s, const float *src, int sw, int sh, int spitch,
                              float *dst, int dh, int dpitch, const float *gk, int fs, float factor,
                              float postscale, int do_scale)
{
    dim3 b(64), g((sw + 63) / 64, dh);
    hipLaunchKernelGGL(k_blur_cols_sampled, g, b, 0, s, src, sw, sh, spitch, dst, dh, dpitch, gk, fs, factor,
                       postscale, do_scale);
}

// ---------------------------------------------------------------------------------------
// 4th-order central differences with clamped indices (ref .cu:410-449), fp64 intermediate
// ---------------------------------------------------------------------------------------
__global__ void k_gradient(const float *__restrict__ f, float *__restrict__ gx, float *__restrict__ gy,
                           int w, int h, int pitch, size_t cstride)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float *fc = f + cstride * blockIdx.z;
    const float *row = fc + (size_t)y * pitch;
    int xp1 = min(x + 1, w - 1), xp2 = min(x + 2, w - 1), xm1 = max(x - 1, 0), xm2 = max(x - 2, 0);
    int yp1 = min(y + 1, h - 1), yp2 = min(y + 2, h - 1), ym1 = max(y - 1, 0), ym2 = max(y - 2, 0);
    double dx = ((double)(-row[xp2]) + 8. * (double)row[xp1] - 8. * (double)row[xm1] + (double)row[xm2]) / 12.0;
    double dy = ((double)(-fc[(size_t)yp2 * pitch + x]) + 8. * (double)fc[(size_t)yp1 * pitch + x]
                 - 8. * (double)fc[(size_t)ym1 * pitch + x] + (double)fc[(size_t)ym2 * pitch + x]) / 12.0;
    size_t o = cstride * blockIdx.z + (size_t)y * pitch + x;
    gx[o] = (float)dx;
    if (gy) gy[o] = (float)dy;   // the caller passes null when d/dy is dead (ref .cu:591-594)
}

void launch_gradient(hipStream_t s, const float *f, float *gx, float *gy, int w, int h, int pitch, int nc, size_t cstride)
{
    dim3 b(64, 4), g((w + 63) / 64, (h + 3) / 4, nc);
    hipLaunchKernelGGL(k_gradient, g, b, 0, s, f, gx, gy, w, h, pitch, cstride);
}

// ---------------------------------------------------------------------------------------
// Catmull-Rom bicubic flow up-sampling (ref .cu:230-309, 452-466)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1d(float v0, float v1, float v2, float v3, float x)
{
    double xd = (double)x;
    double inner3 = 3.0 * (double)(v1 - v2) + (double)v3 - (double)v0;
    double inner2 = 2.0 * (double)v0 - 5.0 * (double)v1 + 4.0 * (double)v2 - (double)v3 + xd * inner3;
    double inner1 = (double)(v2 - v0) + xd * inner2;
    return (float)((double)v1 + 0.5 * xd * inner1);
}

// Index rule of the reference: truncate toward zero, THEN clamp; the fraction is taken
// against the clamped centre index (so -1 < uu < 0 gives a degenerate stencil).
__device__ __forceinline__ float bicubic_sample(const float *__restrict__ src, int pitch, float uu, float vv, int nx, int ny)
{
    int xs[4], ys[4];
    xs[1] = clampi((int)uu, 0, nx - 1);
    ys[1] = clampi((int)vv, 0, ny - 1);
    xs[0] = clampi((int)(uu - 1), 0, nx - 1);
    ys[0] = clampi((int)(vv - 1), 0, ny - 1);
    xs[2] = clampi((int)(uu + 1), 0, nx - 1);
    ys[2] = clampi((int)(vv + 1), 0, ny - 1);
    xs[3] = clampi((int)(uu + 2), 0, nx - 1);
    ys[3] = clampi((int)(vv + 2), 0, ny - 1);
    float fy = vv - (float)ys[1];
    float fx = uu - (float)xs[1];
    float col[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        float t0 = src[(size_t)ys[0] * pitch + xs[a]];
        float t1 = src[(size_t)ys[1] * pitch + xs[a]];
        float t2 = src[(size_t)ys[2] * pitch + xs[a]];
        float t3 = src[(size_t)ys[3] * pitch + xs[a]];
        col[a] = cubic1d(t0, t1, t2, t3, fy);
    }
    return cubic1d(col[0], col[1], col[2], col[3], fx);
}

__global__ void k_upsample(const float *__restrict__ coarse, int cw, int ch, int cpitch,
                           float *__restrict__ fine, int fw, int fh, int fpitch, float sf)
{
    int ii = blockIdx.x * blockDim.x + threadIdx.x;
    int jj = blockIdx.y * blockDim.y + threadIdx.y;
    if (ii >= fw || jj >= fh) return;
    const float fx = ((float)fw / (float)cw);
    const float fy = ((float)fh / (float)ch);
    float i2 = (float)((double)((float)ii / fx) - (0.5 - 0.5 / (double)fx));
    float j2 = (float)((double)((float)jj / fy) - (0.5 - 0.5 / (double)fy));
    fine[(size_t)jj * fpitch + ii] = bicubic_sample(coarse, cpitch, i2, j2, cw, ch) / sf;
}

void launch_upsample(hipStream_t s, const float *coarse, int cw, int ch, int cpitch,
                     float *fine, int fw, int fh, int fpitch, float sf)
{
    dim3 b(64, 4), g((fw + 63) / 64, (fh + 3) / 4);
    hipLaunchKernelGGL(k_upsample, g, b, 0, s, coarse, cw, ch, cpitch, fine, fw, fh, fpitch, sf);
}

// ---------------------------------------------------------------------------------------
// Assembly (ref .cu:611-1097): one thread per pixel.
//
// Produces, instead of the reference's 12 CSR entries per pixel, five planes
//   a1, a2, a4     the 2x2 diagonal block  [a1 a2; a2 a4]
//   wx = a7        coupling to the pixel at i+1 (east);  the west coupling a5(i,j) == wx(i-1,j)
//   wy = a8        coupling to the pixel at j+1 (north); the south coupling a6(i,j) == wy(i,j-1)
// (both identities are bit-exact consequences of the reference's formulas and are asserted on
// the oracle's planes in tests/test_oracle_structure.py) plus the right-hand side, written
// straight into r (r0 = b because x0 = 0).  It also emits the block partials of b.b and
// b.(M^-1 b) and resets the solve's state, so the first PCG pass needs no separate init.
// ---------------------------------------------------------------------------------------
#ifndef ASM_ROWS
#define ASM_ROWS 1      // EXPERIMENT (round 4): with R > 1 a thread of k_assemble marches through R consecutive rows -- the 3 x 3
                        // window of u, v rolls upward in registers (six loads per row instead of eighteen) and psi'_s of the northern edge of row j IS
                        // psi'_s of the southern edge of row j + 1, the same float, computed once (1 + 3 R instead of 4 R evaluations).  Bit-identical
                        // coefficient planes (the parity suite passes with R = 4).  Measured, us per launch at 5000^2 / 2500^2 / 1250^2 (two runs each,
                        // profiles/r4_time_assembly.txt): R = 1 495-497 / 139 / 37.2; 4: 496-501 / 140-141 / 38.1; 6: 501-504 / 141-143 / 39.4;
                        // 8: 469-471 / 139 / 40.4-41.0; 12: 558-561 / 156 / 46; 16: 579-583 / 162-166 / 47 -- a fifth fewer psi' and two thirds fewer u, v
                        // loads buy nothing (what the shared psi' saves, the rolling window and the row loop spend; beyond 8 rows the registers cost
                        // occupancy), and the one gain (8 rows at 5000^2, -5 %) is a loss on every smaller level.  1 = the form of rounds 1-3, the product.
#endif
#ifndef ASM_WAVES
#define ASM_WAVES 4     // waves (= rows of 64 pixels marched by one wave each) per workgroup of k_assemble: 4 = 256 threads, a 64 x 4 tile (rounds 1-3);
                        // 8 / 16 = taller tiles, fewer halo rows fetched per tile (EXPERIMENT, round 4: tools/time_assembly.py)
#endif
constexpr int kAsmTX = 64, kAsmRows = ASM_ROWS, kAsmWaves = ASM_WAVES, kAsmThreads = 64 * kAsmWaves, kAsmTY = kAsmWaves * kAsmRows;

// psi'_s (ref .cu:73-80): (float)(1. / (double)y) with y = sqrtf(...) a float is the correctly rounded float reciprocal of y (see
// jacobi_inv in device_util.hpp: double rounding is innocuous for a quotient of floats), which rcp_exact gives in three instructions
// instead of an fp64 division; y >= 1e-3 is normal and so is its reciprocal.  (y = +inf -- a flow that has already diverged -- gives
// NaN here and 0 there.)
__device__ __forceinline__ float psi_smooth(float x)
{
    // (rcp_exact's Newton step turns 1 / inf into NaN where the division gives 0: a run that has diverged keeps the reference's value)
    const float s = sqrtf((float)((double)x + 1E-6));
    return s == __builtin_inff() ? 0.f : rcp_exact(s);
}
// psi'_d (ref .cu:96-103) as the reference's expression compiles: IEEE square root, IEEE division, in double
__device__ __forceinline__ float psi_data_ieee(float x)
{
    return (float)(1. / sqrt((double)x + 1E-6));
}
// the same value from the hardware's reciprocal-square-root estimate and two Newton steps (7 fp64 instructions instead of ~30); used
// only where assemble_math_selftest has found it equal to psi_data_ieee on every float x >= 0 (AssembleParams::fast_math bit 2)
__device__ __forceinline__ float psi_data_fast(float x)
{
    const double d = (double)x + 1E-6;
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
    return (float)(d == (double)__builtin_inff() ? 0. : y);       // x = +inf (a diverged run): the Newton steps give NaN, 1 / sqrt(inf) is 0
}
// 1 / (s + 1) in double, rounded to float (the Zimmer normalisations, ref .cu:795-803), the reference's way ...
__device__ __forceinline__ float rcp1p_ieee(float s) { return (float)(1. / ((double)s + 1.)); }
// ... and from v_rcp_f64 + two Newton steps (5 fp64 instructions instead of ~14; fast_math bit 1, same proviso)
__device__ __forceinline__ float rcp1p_fast(float s)
{
    const double d = (double)s + 1.;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
    return (float)(d == (double)__builtin_inff() ? 0. : r);       // s = +inf: 0, as the division gives (the Newton steps: NaN)
}
// x / alpha, correctly rounded, from the correctly rounded reciprocal of alpha: q = x * ralpha, then one step on the exact residual
// x - alpha * q (Markstein).  Three fp64 instructions instead of ~14; fast_math bit 0: only after the self-test has compared it with
// the division on every finite float x for THIS alpha.
__device__ __forceinline__ double div_alpha_fast(double x, double alpha, double ralpha)
{
    const double q = x * ralpha;
    // (a zero keeps its sign as the quotient does: the residual step would turn -0 into +0; an infinity stays one: the step gives NaN)
    return (q == 0. || __builtin_fabs(q) == (double)__builtin_inff()) ? q : __builtin_fma(__builtin_fma(-alpha, q, x), ralpha, q);
}
__device__ __forceinline__ float sq(float x) { return x * x; }

// Specialisations (round 3; one channel and everything that is uniform over a launch known at compile time: 574 -> 495 us at 5000^2 from
// the channel count alone).  NC: the channel count (1: the usual single-channel pair; 0: L.nc at run time).  MODE: the GNC step -- 0 al1 == 1
// (quadratic terms only), 1 the blend, 2 al1 == 0 (robust terms only), -1 decided at run time.  DOZIM / HINT: Zimmer's normalisation on /
// off, the first-guess hint term present (lambdac != 0) / absent, -1 at run time.  Same expressions, same order, same bits in every instance.
template <int NC, int MODE, int DOZIM, int HINT>
__global__ __launch_bounds__(kAsmThreads) void k_assemble(LevelPtrs L, AssembleParams P)
{
    __shared__ double s_red[2 * kAsmWaves];
    const int w = L.w, h = L.h, pitch = L.pitch;
    // rows [ya0, ya1): the whole level for a plain plan; a band also fills the first row of each neighbouring band,
    // so that pass A finds the coefficients and the initial residual of its halo rows without an exchange
    const int tiles_x = (w + kAsmTX - 1) / kAsmTX, tiles_y = (L.ya1 - L.ya0 + kAsmTY - 1) / kAsmTY;
    const int ntiles = tiles_x * tiles_y;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const double al1 = P.al1, alpha = P.alpha, loa = P.loa;
    const float lambdac = P.lambdac;
    const bool quad_only = MODE < 0 ? (al1 == 1.0) : MODE == 0, robust_only = MODE < 0 ? (al1 == 0.0) : MODE == 2;
    const bool dozim = DOZIM < 0 ? (P.dozim != 0) : DOZIM != 0;
    const bool hint = HINT < 0 ? (lambdac != 0.f) : HINT != 0;
    const double ralpha = P.ralpha;
    const bool fdiv = (P.fast_math & 1) != 0, frcp = (P.fast_math & 2) != 0, frsq = (P.fast_math & 4) != 0;   // uniform
    auto over_alpha = [&](float x) { return fdiv ? div_alpha_fast((double)x, alpha, ralpha) : (double)x / alpha; };
    double acc_rr = 0., acc_rz = 0.;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int ii = (t % tiles_x) * kAsmTX + lx;
      const int jj0 = L.ya0 + (t / tiles_x) * kAsmTY + ly * kAsmRows;
      if (ii >= w || jj0 >= L.ya1) continue;
      // mirrored neighbour columns (ref .cu:629-652)
      const int xe = (ii == w - 1) ? ii - 1 : ii + 1;
      const int xw = (ii == 0) ? ii + 1 : ii - 1;
      const float *U = L.u, *V = L.v;
      // The thread marches through kAsmRows consecutive rows.  Its 3 x 3 windows of u and v roll with it: the centre row becomes the
      // southern one, the northern the centre, and only the new northern row is loaded (the frame's last row mirrors: its "north" is
      // the row below it, which the window already holds).  And the north edge's psi'_s of one row is the south edge's of the next:
      // Un(i, j) and Us(i, j + 1) are the same four squares added in the same order (the inner sums commute), see EXPERIMENTS.md 8.
      float use = 0.f, us = 0.f, usw = 0.f, ue = 0.f, uc = 0.f, uw = 0.f, une = 0.f, un = 0.f, unw = 0.f;
      float vse = 0.f, vs = 0.f, vsw = 0.f, ve = 0.f, vc = 0.f, vw = 0.f, vne = 0.f, vn = 0.f, vnw = 0.f;
      float ps_south = 0.f;
#pragma unroll
      for (int r = 0; r < kAsmRows; r++) {
        const int jj = jj0 + r;
        if (jj >= L.ya1) break;
        const int yn = (jj == h - 1) ? jj - 1 : jj + 1;
        const int ys = (jj == 0) ? jj + 1 : jj - 1;
        const size_t rc = (size_t)jj * pitch, rn = (size_t)yn * pitch, rs = (size_t)ys * pitch;
        if (r == 0) {
            ue = U[rc + xe]; uc = U[rc + ii]; uw = U[rc + xw]; use = U[rs + xe]; us = U[rs + ii]; usw = U[rs + xw];
            ve = V[rc + xe]; vc = V[rc + ii]; vw = V[rc + xw]; vse = V[rs + xe]; vs = V[rs + ii]; vsw = V[rs + xw];
        } else {            // jj >= 1 here, so the southern row is row jj - 1: the previous centre
            use = ue; us = uc; usw = uw; ue = une; uc = un; uw = unw;
            vse = ve; vs = vc; vsw = vw; ve = vne; vc = vn; vw = vnw;
        }
        if (r > 0 && jj == h - 1) {       // mirrored north = row jj - 1 = the southern row of the window
            une = use; un = us; unw = usw; vne = vse; vn = vs; vnw = vsw;
        } else {
            une = U[rn + xe]; un = U[rn + ii]; unw = U[rn + xw];
            vne = V[rn + xe]; vn = V[rn + ii]; vnw = V[rn + xw];
        }

        float Ue = sq(ue - uc) + sq((float)(0.25 * (double)((une - use) + (un - us))))
                 + sq(ve - vc) + sq((float)(0.25 * (double)((vne - vse) + (vn - vs))));
        float Uw = sq(uc - uw) + sq((float)(0.25 * (double)((unw - usw) + (un - us))))
                 + sq(vc - vw) + sq((float)(0.25 * (double)((vnw - vsw) + (vn - vs))));
        float Un = sq(un - uc) + sq((float)(0.25 * (double)((une - unw) + (ue - uw))))
                 + sq(vn - vc) + sq((float)(0.25 * (double)((vne - vnw) + (ve - vw))));
        float Us = sq(uc - us) + sq((float)(0.25 * (double)((use - usw) + (ue - uw))))
                 + sq(vc - vs) + sq((float)(0.25 * (double)((vse - vsw) + (ve - vw))));
        // The GNC blend al1 * (quadratic terms) + (1 - al1) * (robust terms) runs with al1 = 1, 0.5, 0.  A factor that is
        // exactly 0 makes its term +-0, which changes nothing it is added to (only, possibly, the sign of an exact zero):
        // the robust terms -- four psi'_s, two psi'_d, seven fp64 divisions and six square roots -- are skipped at al1 == 1,
        // the quadratic ones -- five fp64 divisions by alpha -- at al1 == 0.  Uniform branches.
        float ps1 = 0.f, ps2 = 0.f, ps3 = 0.f, ps4 = 0.f, pstot = 0.f, snu = 0.f, snv = 0.f;
        if (!quad_only) {
            ps1 = psi_smooth(Uw); ps3 = psi_smooth(Ue); ps4 = psi_smooth(Un);
            ps2 = (r == 0) ? psi_smooth(Us) : ps_south;      // = the previous row's ps4, bit for bit
            ps_south = ps4;
            pstot = ps1 + ps2 + ps3 + ps4;
            snu = ps1 * uw + ps2 * us + ps3 * ue + ps4 * un;
            snv = ps1 * vw + ps2 * vs + ps3 * ve + ps4 * vn;
        }
        const float pstotq = 4.f;
        float snuq = uw + us + ue + un;
        float snvq = vw + vs + ve + vn;

        // warped sampling position (ref .cu:732-747)
        float xwp = (float)ii + uc;
        float ywp = (float)jj + vc;
        bool hitx = false, hity = false;
        if (xwp < 0) { xwp = 0; hitx = true; }
        if (xwp >= w) { xwp = (float)(w - 1); hitx = true; }
        if (ywp < 0) { ywp = 0; hity = true; }
        if (ywp >= h) { ywp = (float)(h - 1); hity = true; }
        int x0 = (int)xwp, y0 = (int)ywp;
        if (x0 == w - 1) x0 = w - 2;
        if (y0 == h - 1) y0 = h - 2;
        const float fx1 = (float)x0, fx2 = (float)(x0 + 1), fy1 = (float)y0, fy2 = (float)(y0 + 1);
        // ref .cu:60-63 (oct_binterp_cu) divides these by (fx2 - fx1) and (fy2 - fy1): exactly 1.0f for every cell, and x / 1.0f is x
        const float p1 = fx2 - xwp;
        const float p2 = xwp - fx1;
        const float p3 = fy2 - ywp;
        const float p4 = ywp - fy1;
        const size_t c1 = (size_t)y0 * pitch + x0, c3 = c1 + pitch;

        float t1 = 0, t2 = 0, t4 = 0, t5 = 0, t6 = 0, e1 = 0;
        float g1 = 0, g2s = 0, g4 = 0, g5 = 0, g6 = 0, e2 = 0;
        const int nchan = NC > 0 ? NC : L.nc;
        for (int c = 0; c < nchan; c++) {      // (a constant trip count of 1 ... 3 is unrolled by the compiler)
            const size_t cb = L.cstride * c;
#define OCT_BIL(F) (p3 * ((p1) * (F)[cb + c1] + (p2) * (F)[cb + c1 + 1]) + p4 * ((p1) * (F)[cb + c3] + (p2) * (F)[cb + c3 + 1]))
            float w2 = OCT_BIL(L.img2);
            float Ix = OCT_BIL(L.gx2);
            float Iy = OCT_BIL(L.gy2);
            float Ixx = OCT_BIL(L.gxx);
            float Ixy = OCT_BIL(L.gxy);
            float Iyy = OCT_BIL(L.gyy);
#undef OCT_BIL
            if (hitx) { Ix = 0.f; Ixx = 0.f; Ixy = 0.f; }
            if (hity) { Iy = 0.f; Ixy = 0.f; Iyy = 0.f; }
            float It = w2 - L.img1[cb + rc + ii];
            float Ixt = Ix - L.gx1[cb + rc + ii];
            float Iyt = Iy - L.gy1[cb + rc + ii];
            float IxIx = Ix * Ix, IyIy = Iy * Iy, IxxIxx = Ixx * Ixx, IxyIxy = Ixy * Ixy, IyyIyy = Iyy * Iyy;
            float na, nb, ncc;
            if (dozim) {
                if (frcp) { na = rcp1p_fast(IxIx + IyIy); nb = rcp1p_fast(IxxIxx + IxyIxy); ncc = rcp1p_fast(IxyIxy + IyyIyy); }
                else { na = rcp1p_ieee(IxIx + IyIy); nb = rcp1p_ieee(IxxIxx + IxyIxy); ncc = rcp1p_ieee(IxyIxy + IyyIyy); }
            } else {
                na = 1.f; nb = 1.f; ncc = 1.f;
            }
            e1 += na * It * It;
            e2 += (nb * Ixt * Ixt + ncc * Iyt * Iyt);
            t1 += (na * IxIx);
            g1 += (nb * IxxIxx + ncc * IxyIxy);
            t2 += na * Ix * Iy;
            g2s += (nb * Ixx * Ixy + ncc * Iyy * Ixy);
            t4 += (na * IyIy);
            g4 += ((nb * IxyIxy + ncc * IyyIyy));
            float naIt = -na * It;
            float nbIxt = nb * Ixt;
            float ncIyt = ncc * Iyt;
            t5 += naIt * Ix;
            g5 += -(nbIxt * Ixx + ncIyt * Ixy);
            t6 += naIt * Iy;
            g6 += -(nbIxt * Ixy + ncIyt * Iyy);
        }
        float hint_u = 0.f, hint_v = 0.f;
        if (hint) {   // lambdac == 0: 0*(finite) == 0 exactly, so the reads can be skipped
            hint_u = lambdac * (uc - L.ut[rc + ii]);
            hint_v = lambdac * (vc - L.vt[rc + ii]);
        }
        float pd = 0.f, pd2 = 0.f;
        if (!quad_only) {
            pd = (float)over_alpha(frsq ? psi_data_fast(e1) : psi_data_ieee(e1));
            pd2 = (float)(loa * (double)(frsq ? psi_data_fast(e2) : psi_data_ieee(e2)));
        }
        double qa1 = 0., qa2 = 0., qa4 = 0., qbu = 0., qbv = 0.;          // the quadratic terms
        if (!robust_only) {
            qa1 = over_alpha(t1) + loa * (double)g1 + (double)lambdac + (double)pstotq;
            qa2 = over_alpha(t2) + loa * (double)g2s;
            qa4 = over_alpha(t4) + loa * (double)g4 + (double)lambdac + (double)pstotq;
            qbu = over_alpha(t5) + loa * (double)g5 - (double)hint_u + (double)snuq - (double)(pstotq * uc);
            qbv = over_alpha(t6) + loa * (double)g6 - (double)hint_v + (double)snvq - (double)(pstotq * vc);
        }
        float a1, a2, a4, a7, a8, bu, bv;
        if (quad_only) {                    // 1 * q + 0 * r
            a1 = (float)qa1; a2 = (float)qa2; a4 = (float)qa4;
            a7 = -1.f; a8 = -1.f;           // (float)(-1 * (1 + 0 * psi)), east and north
            bu = (float)qbu; bv = (float)qbv;
        } else if (robust_only) {           // 0 * q + 1 * r
            a1 = pd * t1 + pd2 * g1 + lambdac + pstot;
            a2 = pd * t2 + pd2 * g2s;
            a4 = pd * t4 + pd2 * g4 + lambdac + pstot;
            a7 = (float)(-1 * (double)ps3);
            a8 = (float)(-1 * (double)ps4);
            bu = pd * t5 + pd2 * g5 - hint_u + snu - pstot * uc;
            bv = pd * t6 + pd2 * g6 - hint_v + snv - pstot * vc;
        } else {
            a1 = (float)((al1) * qa1 + (1 - al1) * (double)(pd * t1 + pd2 * g1 + lambdac + pstot));
            a2 = (float)((al1) * qa2 + (1 - al1) * (double)(pd * t2 + pd2 * g2s));
            a4 = (float)((al1) * qa4 + (1 - al1) * (double)(pd * t4 + pd2 * g4 + lambdac + pstot));
            a7 = (float)(-1 * (al1 + (1 - al1) * (double)ps3));   // east
            a8 = (float)(-1 * (al1 + (1 - al1) * (double)ps4));   // north
            bu = (float)(al1 * qbu + (1. - al1) * (double)(pd * t5 + pd2 * g5 - hint_u + snu - pstot * uc));
            bv = (float)(al1 * qbv + (1 - al1) * (double)(pd * t6 + pd2 * g6 - hint_v + snv - pstot * vc));
        }
        const size_t o = rc + ii;
        L.a1[o] = a1; L.a2[o] = a2; L.a4[o] = a4;
        if (!(L.lean && L.unit_w)) { L.wx[o] = a7; L.wy[o] = a8; }     // exactly -1 in the first GNC step: nobody reads them then
        L.ru[o] = bu; L.rv[o] = bv;
        // Jacobi preconditioner, M <- 1./M in double (ref .cu:141-149); stored for the two-pass form's pass B and the
        // single-workgroup solver (the fused kernel divides instead)
        const float mu = jacobi_inv(a1), mv = jacobi_inv(a4);
        if (!L.lean) { L.mu[o] = mu; L.mv[o] = mv; }
        // r.r and r.z of the initial residual (ref .cu:1115-1126, 1157): z = M r
        float zu = mu * bu;
        float zv = mv * bv;
        if (jj >= L.y0 && jj < L.y1) {      // halo rows are the neighbouring band's to count
            acc_rr += (double)(bu * bu) + (double)(bv * bv);
            acc_rz += (double)(bu * zu) + (double)(bv * zv);
        }
      }
    }
    double tot_rr, tot_rz;
    if (kAsmWaves == 4) {          // (the reduction of rounds 1-3, kept to the letter for the product's four waves)
        tot_rr = block_sum_256(acc_rr, s_red);
        tot_rz = block_sum_256(acc_rz, s_red);
    } else {
        const double acc2[2] = {acc_rr, acc_rz};
        double tot2[2];
        block_sum_multi<2, kAsmThreads>(acc2, s_red, tot2);
        tot_rr = tot2[0]; tot_rz = tot2[1];
    }
    if (threadIdx.x == 0) {
        L.part_rr[blockIdx.x] = tot_rr;
        L.part_rz[blockIdx.x] = tot_rz;
        if (blockIdx.x == 0) {
            PcgState s0; s0.rz = 0.f; s0.stopped = 0; s0.iters = 0; s0.pad = 0;
            L.st[0] = s0;
        }
    }
}

// Persistent grid for `items` equal work items: the fewest rounds that fit kMaxParts blocks, then
// as many blocks as give every block that many items (+-1) -- no half-empty last round.
static int g_max_blocks = kMaxParts;
static int g_grid_multiple = 1;
void set_max_blocks(int n) { g_max_blocks = (n >= 64 && n <= kMaxParts) ? n : kMaxParts; }
void set_grid_multiple(int m) { g_grid_multiple = m > 0 ? m : 1; }
int grid_multiple() { return g_grid_multiple; }

int balanced_grid(long items)
{
    const int cap = g_max_blocks;
    long g;
    if (items <= cap) {
        g = items < 1 ? 1 : items;
    } else {
        long rounds = (items + cap - 1) / cap;
        g = (items + rounds - 1) / rounds;
    }
    if (g_grid_multiple > 1 && g >= 8 * g_grid_multiple) {   // XCD-banded launches want a multiple of 8
        g = (g + g_grid_multiple - 1) / g_grid_multiple * g_grid_multiple;
        if (g > cap) g -= g_grid_multiple;
    }
    return (int)g;
}

int assemble_grid_size(int w, int h)
{
    return balanced_grid((long)((w + kAsmTX - 1) / kAsmTX) * ((h + kAsmTY - 1) / kAsmTY));
}

void launch_assemble(hipStream_t s, const LevelPtrs &L, const AssembleParams &P, int grid)
{
    static const bool generic_only = [] { const char *e = tune_env("OCTANE_TUNE_ASM_GENERIC"); return e && atoi(e) != 0; }();   // developer knob: A/B timing
    if (generic_only) { hipLaunchKernelGGL((k_assemble<0, -1, -1, -1>), dim3(grid), dim3(kAsmThreads), 0, s, L, P); return; }
    const int mode = P.al1 == 1.0 ? 0 : (P.al1 == 0.0 ? 2 : 1);
    const int z = P.dozim ? 1 : 0, hn = P.lambdac != 0.f ? 1 : 0;
    // Two and three channels (round 4; the reference's loop treats 1 ... 3 alike, ref .cu:749-829): the channel count and the GNC step as
    // template parameters for the default flags (Zimmer's normalisation on, no hint term) -- the channel loop is then unrolled and its
    // `cstride * c` address arithmetic folds into the 27 loads; every other combination runs the generic instance.
    if (L.nc != 1) {
#define OCT_ASM_NC(N, M) if (L.nc == N && mode == M && z == 1 && hn == 0) { hipLaunchKernelGGL((k_assemble<N, M, 1, 0>), dim3(grid), dim3(kAsmThreads), 0, s, L, P); return; }
        OCT_ASM_NC(2, 0) OCT_ASM_NC(2, 1) OCT_ASM_NC(2, 2) OCT_ASM_NC(3, 0) OCT_ASM_NC(3, 1) OCT_ASM_NC(3, 2)
#undef OCT_ASM_NC
        hipLaunchKernelGGL((k_assemble<0, -1, -1, -1>), dim3(grid), dim3(kAsmThreads), 0, s, L, P);
        return;
    }
#define OCT_ASM_CASE(M, Z, H) if (mode == M && z == Z && hn == H) { hipLaunchKernelGGL((k_assemble<1, M, Z, H>), dim3(grid), dim3(kAsmThreads), 0, s, L, P); return; }
    OCT_ASM_CASE(0, 1, 0) OCT_ASM_CASE(1, 1, 0) OCT_ASM_CASE(2, 1, 0) OCT_ASM_CASE(0, 0, 0) OCT_ASM_CASE(1, 0, 0) OCT_ASM_CASE(2, 0, 0)
    OCT_ASM_CASE(0, 1, 1) OCT_ASM_CASE(1, 1, 1) OCT_ASM_CASE(2, 1, 1) OCT_ASM_CASE(0, 0, 1) OCT_ASM_CASE(1, 0, 1) OCT_ASM_CASE(2, 0, 1)
#undef OCT_ASM_CASE
}

// ---------------------------------------------------------------------------------------
// Self-test of the assembly's fast exact forms (AssembleParams::fast_math): every float input, bit for bit against the IEEE sequence.
// ---------------------------------------------------------------------------------------
__global__ void k_selftest_asm_math(double alpha, double ralpha, unsigned long long *out)
{
    unsigned long long n0 = 0, b0 = 0, n1 = 0, b1 = 0, n2 = 0, b2 = 0, first = 0, which = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < 0x100000000ull; b += stride) {
        const unsigned u = (unsigned)b;
        const float x = __uint_as_float(u);
        if ((u & 0x7FFFFFFFu) <= 0x7F800000u) {          // x / alpha: every float but the NaNs, both signs, zeros, denormals and infinities included
            const double want = (double)x / alpha, got = div_alpha_fast((double)x, alpha, ralpha);
            n0++;
            if (__double_as_longlong(want) != __double_as_longlong(got)) { if (!(b0 + b1 + b2)) { first = b; which = 0; } b0++; }
        }
        if (u <= 0x7F800000u) {                          // s >= 0, +inf included: 1 / (s + 1) and 1 / sqrt(x + 1e-6)
            n1++;
            if (__float_as_uint(rcp1p_ieee(x)) != __float_as_uint(rcp1p_fast(x))) { if (!(b0 + b1 + b2)) { first = b; which = 1; } b1++; }
            n2++;
            if (__float_as_uint(psi_data_ieee(x)) != __float_as_uint(psi_data_fast(x))) { if (!(b0 + b1 + b2)) { first = b; which = 2; } b2++; }
        }
    }
    atomicAdd(&out[0], n0); atomicAdd(&out[1], b0); atomicAdd(&out[2], n1); atomicAdd(&out[3], b1); atomicAdd(&out[4], n2); atomicAdd(&out[5], b2);
    if (b0 + b1 + b2) { atomicMax(&out[6], first); atomicMax(&out[7], which); }
}

int assemble_fast_math_bits(double alpha)
{
    static std::mutex mu;
    static std::map<double, int> known;
    if (const char *e = tune_env("OCTANE_TUNE_ASM_FAST")) { if (atoi(e) == 0) return 0; }
    if (!(alpha > 0.) || !std::isfinite(alpha)) return 0;
    std::lock_guard<std::mutex> g(mu);
    auto it = known.find(alpha);
    if (it != known.end()) return it->second;
    unsigned long long r[8] = {0};
    int bits = 0;
    if (assemble_math_selftest(nullptr, alpha, r) == 0) {
        if (r[0] == 0xFF000002ull && r[1] == 0) bits |= 1;       // 2 x (2^31 - 2^23) finite floats and the two infinities
        if (r[2] == 0x7F800001ull && r[3] == 0) bits |= 2;       // every float >= 0 and +inf
        if (r[4] == 0x7F800001ull && r[5] == 0) bits |= 4;
    }
    known[alpha] = bits;
    return bits;
}

int assemble_math_selftest(hipStream_t s, double alpha, unsigned long long *out8)
{
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    int rc = -1;
    if (hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), s) == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_asm_math, dim3(8192), dim3(256), 0, s, alpha, 1. / alpha, d);
        if (hipMemcpyAsync(out8, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) rc = 0;
    }
    (void)hipFree(d);
    return rc;
}

}  // namespace octane
