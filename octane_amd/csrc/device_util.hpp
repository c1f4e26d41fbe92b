// device_util.hpp -- small wave64 / workgroup helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace octane {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Sum over the 64 lanes of a wavefront; every lane ends with the total, and the order of
// additions is fixed (butterfly: lane i adds lane i ^ 32, then ^ 16, 8, 4, 2, 1), so the result is run-to-run deterministic.
// The exchange steps are register moves, not trips through the LDS crossbar (`__shfl_xor` = two ds_bpermute_b32 per step and double,
// ~100 cycles a step): gfx950's v_permlane32_swap / v_permlane16_swap for the steps across rows, DPP moves within a row of 16 lanes.
// row_ror:4 stands for "lane ^ 4" because after the steps 32, 16, 8 the lanes i and i + 8 of a row hold the same value.  Same pairs,
// same order, same bits as the __shfl_xor loop (checked on the device over 2^18 random inputs, round 3).
template <int CTRL>
__device__ __forceinline__ double wave_dpp_f64(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum(double v)
{
    const unsigned lane = __lane_id();
    {   // lane ^ 32: the swap exchanges the upper half of its first operand with the lower half of its second
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        const bool low = lane < 32;
        v += __longlong_as_double((long long)(((unsigned long long)(low ? r1[1] : r1[0]) << 32) | (low ? r0[1] : r0[0])));
    }
    {   // lane ^ 16: the same between the odd rows of the first and the even rows of the second operand
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        const bool low = (lane & 16) == 0;
        v += __longlong_as_double((long long)(((unsigned long long)(low ? r1[1] : r1[0]) << 32) | (low ? r0[1] : r0[0])));
    }
    v += wave_dpp_f64<0x128>(v);     // row_ror:8
    v += wave_dpp_f64<0x124>(v);     // row_ror:4
    v += wave_dpp_f64<0x4E>(v);      // quad_perm:[2,3,0,1]
    v += wave_dpp_f64<0xB1>(v);      // quad_perm:[1,0,3,2]
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

// N sums over a workgroup of NT threads (NT / 64 waves) with one pair of barriers; every thread gets all totals.
// scratch: (NT / 64) * N doubles.
template <int N, int NT>
__device__ __forceinline__ void block_sum_multi(const double (&v)[N], double *scratch, double (&out)[N])
{
    constexpr int NW = NT / 64;
    double w[N];
#pragma unroll
    for (int j = 0; j < N; j++) w[j] = wave_sum(v[j]);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // scratch may still be read from a previous call
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < N; j++) scratch[j * NW + wave] = w[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; j++) {
        double t = scratch[j * NW];
#pragma unroll
        for (int i = 1; i < NW; i++) t += scratch[j * NW + i];
        out[j] = t;
    }
}
// 1 / x, correctly rounded for every normal x whose reciprocal is normal (checked against the division on all of them by
// octane_selftest_rcp (pcg_persist.hip) / tests/test_gpu_persist.py): the hardware estimate (1 ulp) and one Newton step in fused arithmetic.
// Three instructions instead of the eleven of an IEEE division; the diagonal of the operator is a normal float with a normal reciprocal
// (in the robust GNC step it can be well below 1: psi' weights of a steep flow) -- inside that range.  NOT equal to the division at
// +-inf (NaN instead of 0): callers that can meet an infinity in a diverged run guard it (vof_kernels.hip psi_smooth).
__device__ __forceinline__ float rcp_exact(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

template <int N, int NT>
__device__ __forceinline__ void fold_band_partials_multi(const double *const *blocks, int first_off, int kind_stride, int n,
                                                         int nbands, double *scratch /* >= (NT / 64) * N */, double (&out)[N])
{
    double v[N];
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = 0.;
    for (int b = 0; b < nbands; b++) {
        const double *__restrict__ part = blocks[b] + first_off;
        for (int i = threadIdx.x; i < n; i += NT) {
#pragma unroll
            for (int j = 0; j < N; j++) v[j] += part[(size_t)j * kind_stride + i];
        }
    }
    block_sum_multi<N, NT>(v, scratch, out);
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

// Tile walk "slabs" (mode >= 5; 2-D tiles of a row-major frame, persistent grid whose size is a multiple of 8).
// The tile rows are grouped into super-rows of S tile rows; inside a super-row the tiles are ranked column by column and
// the ranks are cut into 8 equal shares: XCD x (workgroups with blockIdx % 8 == x -- how the hardware deals workgroups
// over the XCDs today; placement only ever affects speed) owns share x, a vertical slab of ~tiles_x / 8 columns, in EVERY
// super-row.  A tile's left / right AND upper / lower neighbours are then worked on by the same XCD at about the same
// time, so the ring lines neighbouring tiles share can be hits in that XCD's L2 instead of second fetches through the
// fabric, while all XCDs still advance through the frame together: one compact front, as with the plain round-robin walk.
// Shares differ by at most one tile per super-row.  Workgroup (x, j) takes elements j, j + G/8, j + 2G/8, ... of XCD x's
// sequence (super-row after super-row, rank order inside).
struct SlabWalk {
    int tiles_x, tiles_y, S, nfull, rows_last;      // nfull full super-rows, then one of rows_last tile rows (may be 0)
    int x, per, idx;                                // this workgroup: XCD, workgroups per XCD, current sequence element
    int lo_full, cnt_full, lo_last, cnt_last;       // XCD x's rank range in a full / in the last super-row
};
__device__ __forceinline__ SlabWalk slab_walk_begin(int tiles_x, int tiles_y, int S)
{
    SlabWalk w;
    w.tiles_x = tiles_x; w.tiles_y = tiles_y; w.S = S;
    w.nfull = tiles_y / S; w.rows_last = tiles_y - w.nfull * S;
    w.x = blockIdx.x & 7; w.per = gridDim.x >> 3; w.idx = blockIdx.x >> 3;
    const int nf = tiles_x * S, nl = tiles_x * w.rows_last;
    w.lo_full = (int)((long)w.x * nf / 8); w.cnt_full = (int)((long)(w.x + 1) * nf / 8) - w.lo_full;
    w.lo_last = (int)((long)w.x * nl / 8); w.cnt_last = (int)((long)(w.x + 1) * nl / 8) - w.lo_last;
    return w;
}
// current tile as (column, row), or false when this workgroup's walk is over; slab_walk_next advances
__device__ __forceinline__ bool slab_walk_tile(const SlabWalk &w, int &tc, int &tr)
{
    const int in_full = w.nfull * w.cnt_full;
    if (w.idx < in_full) {
        const int s = w.idx / w.cnt_full, rank = w.lo_full + (w.idx - s * w.cnt_full);
        tc = rank / w.S; tr = s * w.S + (rank - tc * w.S);
        return true;
    }
    const int off = w.idx - in_full;
    if (off >= w.cnt_last) return false;
    const int rank = w.lo_last + off;
    tc = rank / w.rows_last; tr = w.nfull * w.S + (rank - tc * w.rows_last);
    return true;
}
__device__ __forceinline__ void slab_walk_next(SlabWalk &w) { w.idx += w.per; }

// Jacobi preconditioner entry exactly as the reference forms it: M = 1./M in double, stored as float (ref .cu:141-149).
// (float)(1. / (double)a) IS the correctly rounded float reciprocal: rounding a quotient of two 24-bit numbers to 53 bits and then to 24
// cannot differ from rounding it to 24 at once (53 >= 2 * 24 + 2), checked on all 2 130 706 432 positive normal floats by
// tests/test_oracle_pins.py::test_float_reciprocal_through_double_is_the_float_division -- and rcp_exact is that reciprocal in three
// instructions (octane_selftest_rcp).  The diagonal of the operator is a normal float with a normal reciprocal.
__device__ __forceinline__ float jacobi_inv(float a) { return rcp_exact(a); }

}  // namespace octane
