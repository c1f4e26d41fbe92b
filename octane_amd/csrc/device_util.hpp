// device_util.hpp -- small wave64 / workgroup helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace octane {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Sum over the 64 lanes of a wavefront; every lane ends with the total, and the order of
// additions is fixed (butterfly), so the result is run-to-run deterministic.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over a 256-thread workgroup (4 waves); every thread gets the total.
// `scratch` must hold at least 4 doubles; safe to call back-to-back.
__device__ __forceinline__ double block_sum_256(double v, double *scratch)
{
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // scratch may still be read from a previous call
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}

// N sums over the workgroup with one pair of barriers; every thread gets all totals.  scratch: 4 * N doubles.
template <int N>
__device__ __forceinline__ void block_sum_multi_256(const double (&v)[N], double *scratch, double (&out)[N])
{
    double w[N];
#pragma unroll
    for (int j = 0; j < N; j++) w[j] = wave_sum(v[j]);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // scratch may still be read from a previous call
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < N; j++) scratch[j * 4 + wave] = w[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = ((scratch[j * 4] + scratch[j * 4 + 1]) + scratch[j * 4 + 2]) + scratch[j * 4 + 3];
}

// Every block folds the same `n` per-block partials (n <= kMaxParts) in the same order, so all
// blocks obtain bit-identical totals without atomics, fences or an extra launch: thread t
// adds partials t, t+256, t+512, ..., then the workgroup tree above.
__device__ __forceinline__ double fold_partials_256(const double *__restrict__ part, int n, double *scratch)
{
    double v = 0.;
    for (int i = threadIdx.x; i < n; i += 256) v += part[i];
    return block_sum_256(v, scratch);
}

// The same over the partial blocks of `nbands` row bands (vof_kernels.hpp): band after band, so every workgroup of
// every band obtains the same bits.  nbands == 1 is the fold above.  A block of another band may live in another
// device's memory; it was written by a kernel that completed before this one started (event-ordered).
__device__ __forceinline__ double fold_band_partials_256(const double *const *blocks, int kind_off, int n, int nbands,
                                                         double *scratch)
{
    double v = 0.;
    for (int b = 0; b < nbands; b++) {
        const double *__restrict__ part = blocks[b] + kind_off;
        for (int i = threadIdx.x; i < n; i += 256) v += part[i];
    }
    return block_sum_256(v, scratch);
}

// N sums at once (the fused PCG kernel needs seven): the loads of all kinds are independent, so they cost one memory
// round trip instead of N, and the workgroup tree runs once.  kind i lives kind_stride doubles after kind i-1.
template <int N>
__device__ __forceinline__ void fold_band_partials_multi_256(const double *const *blocks, int first_off, int kind_stride, int n,
                                                             int nbands, double *scratch /* >= 4 * N */, double (&out)[N])
{
    double v[N];
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = 0.;
    for (int b = 0; b < nbands; b++) {
        const double *__restrict__ part = blocks[b] + first_off;
        for (int i = threadIdx.x; i < n; i += 256) {
#pragma unroll
            for (int j = 0; j < N; j++) v[j] += part[(size_t)j * kind_stride + i];
        }
    }
    block_sum_multi_256<N>(v, scratch, out);
}

// Work-item range of this workgroup in a persistent launch.  Plain: items b, b+G, b+2G, ...  Banded: the item
// list is cut into 8 contiguous bands and workgroup b serves band b % 8 -- workgroups are dealt round-robin over
// the 8 XCDs (observed, not guaranteed: MI355X_MICROARCH.md), so each XCD's L2 then sees one compact region of
// the frame and the halo lines shared by neighbouring tiles are L2 hits instead of second fetches through the
// fabric.  Placement only affects speed, never results.
struct ItemRange { int base, first, end, step; };   // items first, first+step, ... < end of the range [base, end)
__device__ __forceinline__ ItemRange item_range(int nitems, bool banded)
{
    ItemRange r;
    if (banded && (gridDim.x & 7) == 0 && nitems >= 64) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int per = (nitems + 7) >> 3;
        const int b0 = xcd * per;
        r.base = b0;
        r.first = b0 + local;
        r.end = min(nitems, b0 + per);
        r.step = gridDim.x >> 3;
    } else {
        r.base = 0; r.first = blockIdx.x; r.end = nitems; r.step = gridDim.x;
    }
    return r;
}

// mode 0: items round-robin over workgroups; 1: one contiguous band per XCD (above); 2: one contiguous run of items per
// workgroup -- a workgroup walking along a row of tiles finds the lines its left / right halo shares with the previous
// tile still in its own L1 / L2, whatever the other workgroups are doing
__device__ __forceinline__ ItemRange item_range_walk(int nitems, int mode)
{
    if (mode == 2) {
        ItemRange r;
        const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
        r.base = 0; r.first = (int)blockIdx.x * per; r.end = min(nitems, r.first + per); r.step = 1;
        return r;
    }
    if ((mode == 3 || mode == 4) && (gridDim.x & 63) == 0) {
        // Runs of 4 (mode 3) or 8 (mode 4) consecutive items -- horizontally adjacent tiles -- on ONE XCD (blockIdx % 8),
        // within the same compact front of gridDim.x items as mode 0: the lines a tile's left / right halo shares with
        // its neighbours are then in the same L2 at the same time.
        const int run = mode == 3 ? 4 : 8;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        ItemRange r;
        r.base = 0; r.first = ((j / run) * 8 + x) * run + (j % run); r.end = nitems; r.step = gridDim.x;
        return r;
    }
    return item_range(nitems, mode == 1);
}

// Jacobi preconditioner entry exactly as the reference forms it: M = 1./M in double, stored
// as float (ref .cu:141-149).
__device__ __forceinline__ float jacobi_inv(float a) { return (float)(1. / (double)a); }

}  // namespace octane
