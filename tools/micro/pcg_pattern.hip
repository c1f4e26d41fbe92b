// pcg_pattern.hip -- access-pattern study for the finest-level PCG iteration (round 3).  No arithmetic worth the name: every kernel here
// moves exactly the bytes of one q-recomputing PCG launch WITHOUT x work (read r(2) p(2) a1 a2 a4 wx wy = 36 B/pixel, write r(2) p(2) =
// 16 B/pixel; the p tile with its two-pixel margin and the operands of the 100 ring groups come on top, as in the production kernel)
// and touches LDS the way the stencils do, so that what is timed is the memory pattern and its overlap, not the operator.
//
//   A  the production structure (k_pcg_fused_q_dma): 256 threads, two float4 groups per thread, two workgroups per CU, own operands
//      by register loads at the start of the tile, the NEXT tile's p tile and ring operands by LDS-DMA after phase 1;
//   B  one 512-thread workgroup per CU, one float4 group per thread, everything of tile t+1 requested at the START of tile t: p tile
//      and ring operands by LDS-DMA into a second set of LDS buffers, the own operands by inline-asm loads into a second set of
//      registers (the compiler does not know they are pending: the one wait is a hand-counted vmcnt) -- a whole tile of requests in
//      flight per CU at all times;
//      (An asm load's destination counts as written when the statement ends: the compiler may copy or reuse it before the data lands --
//      cdna_hip_programming.md 5.7.  The tile loop is therefore unrolled twice with the two register sets swapping roles, no set is ever
//      copied, and both are waited for before the epilogue.  A first version that copied `cur = nxt` and kept two tiles of DMA ahead
//      ended in a memory access fault: one more reason the production kernel does not hide register loads from the compiler.)
//
// Build: hipcc --offload-arch=gfx950 -O3 -o pcg_pattern pcg_pattern.hip      Run: ./pcg_pattern [W H]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int TX = 128, TY = 16, COLS = 144, OFF = 8;     // LDS row: 8 floats of margin either side of the 128 tile columns
constexpr int NRING = 100, NOPS = 9;

struct Planes {
    const float *ru, *rv, *pu, *pv, *a1, *a2, *a4, *wx, *wy;     // read
    float *ro_u, *ro_v, *po_u, *po_v;                            // written
    double *sink;
    int w, h, pitch;
};

__device__ __forceinline__ void dma16(const float *g, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float *p)
{
    typedef __attribute__((address_space(3))) float lds_float;
    return (unsigned)(unsigned long)(lds_float *)p;
}
__device__ __forceinline__ f4v ld4(const float *p) { return *reinterpret_cast<const f4v *>(p); }
__device__ __forceinline__ void st4nt(float *p, f4v v) { __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(p)); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// the staged p tile of tile (tx0, ty0): rows ty0-2 .. ty0+TY+1, 34 float4 groups per row, into s_u / s_v; rows dealt over NW waves
template <int NW>
__device__ __forceinline__ void dma_p_tile(const Planes &P, float *s_u, float *s_v, int tx0, int ty0, int lane, int wv)
{
    if (lane < TX / 4 + 2) {
        const int x0 = clampi(tx0 + 4 * (lane - 1), 0, P.pitch - 4);
        for (int r = wv; r < TY + 4; r += NW) {
            const int y = clampi(ty0 + r - 2, 0, P.h - 1);
            const size_t o = (size_t)y * P.pitch + x0;
            const unsigned du = __builtin_amdgcn_readfirstlane(lds_addr(s_u) + (unsigned)(r * COLS + OFF - 4) * 4u);
            const unsigned dv = __builtin_amdgcn_readfirstlane(lds_addr(s_v) + (unsigned)(r * COLS + OFF - 4) * 4u);
            dma16(P.pu + o, du);
            dma16(P.pv + o, dv);
        }
    }
}
// operands of the 100 ring groups: (operand, half) pairs dealt over NW waves starting at wave W0
template <int NW, int W0>
__device__ __forceinline__ void dma_ring(const Planes &P, float *s_ring, int tx0, int ty0, int lane, int wv)
{
    const float *const plane[NOPS] = {P.ru, P.rv, P.a1, P.a4, P.a2, P.wx, P.wy, P.wy, P.wx};
    const int shift[NOPS] = {0, 0, 0, 0, 0, 0, 0, -P.pitch, -4};
#pragma unroll
    for (int pr = 0; pr < 2 * NOPS; pr++) {
        // NW = 8: waves 4..7 (two staged p rows each, against three of waves 0..3) take three pairs each, waves 0..3 the other six:
        // 17 / 17 / 16 / 16 / 16 / 16 / 16 / 16 vector-memory operations per wave and tile with the nine own loads
        const int owner = NW == 8 ? (pr < 12 ? 4 + (pr & 3) : (pr - 12) & 3) : (pr + W0) % NW;
        if (owner != wv) continue;
        const int op = pr >> 1, i = pr & 1;
        const int j = 64 * i + lane;
        int gx, gy;
        if (j < 34) { gx = j - 1; gy = -1; }
        else if (j < 68) { gx = j - 35; gy = TY; }
        else if (j < 84) { gx = -1; gy = j - 68; }
        else { gx = TX / 4; gy = j - 84; }
        if (j < NRING) {
            const int y = clampi(ty0 + gy, 1, P.h - 1), x = clampi(tx0 + 4 * gx, 4, P.pitch - 4);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(s_ring) + (unsigned)((op * NRING + 64 * i) * 16));
            dma16(plane[op] + (long)y * P.pitch + x + shift[op], dst);
        }
    }
}

// what a stencil reads of an LDS tile for one float4 group: the group, the groups above and below, the pixel to the west and to the east
__device__ __forceinline__ f4v stencil_reads(const float *s, int lrow, int lcol)
{
    f4v c = ld4(&s[lrow * COLS + lcol]), up = ld4(&s[(lrow - 1) * COLS + lcol]), dn = ld4(&s[(lrow + 1) * COLS + lcol]);
    const float wst = s[lrow * COLS + lcol - 1], est = s[lrow * COLS + lcol + 4];
    f4v r = c + up + dn;
    r.x += wst; r.w += est;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// B / C: 512 threads, one group per thread, register double-buffered own operands, DEPTH tiles of DMA ahead
// ---------------------------------------------------------------------------------------------------------------------------------------
struct Own { f4v ru, rv, a1, a4, a2, wx, wy, wys; float wxw; };

#define ASM_LD4(dst, addr) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(addr) : "memory")
#define ASM_LD1(dst, addr) asm volatile("global_load_dword %0, %1, off" : "=&v"(dst) : "v"(addr) : "memory")

__device__ __forceinline__ void own_issue(const Planes &P, Own &o, int tx0, int ty0, int gx, int gy)
{
    const int x0 = clampi(tx0 + 4 * gx, 4, P.pitch - 4), y = clampi(ty0 + gy, 1, P.h - 1);
    const size_t off = (size_t)y * P.pitch + x0;
    ASM_LD4(o.ru, P.ru + off); ASM_LD4(o.rv, P.rv + off); ASM_LD4(o.a1, P.a1 + off); ASM_LD4(o.a4, P.a4 + off);
    ASM_LD4(o.a2, P.a2 + off); ASM_LD4(o.wx, P.wx + off); ASM_LD4(o.wy, P.wy + off); ASM_LD4(o.wys, P.wy + off - P.pitch);
    ASM_LD1(o.wxw, P.wx + off - 1);
}
// waits until at most N of this wave's vector-memory operations are outstanding; the registers of `o` are tied to the wait so that the
// compiler cannot move a use above it
template <int N>
__device__ __forceinline__ void own_wait(Own &o)
{
    asm volatile("s_waitcnt vmcnt(%9)" : "+v"(o.ru), "+v"(o.rv), "+v"(o.a1), "+v"(o.a4), "+v"(o.a2), "+v"(o.wx), "+v"(o.wy), "+v"(o.wys), "+v"(o.wxw) : "n"(N) : "memory");
}

template <int WAITN>
__global__ __launch_bounds__(512, 1) void k_pattern_b(Planes P)
{
    __shared__ __attribute__((aligned(16))) float s_o[2][2][(TY + 4) * COLS];
    __shared__ __attribute__((aligned(16))) float s_n[2][2][(TY + 2) * COLS];
    __shared__ __attribute__((aligned(16))) float s_ring[2][NOPS * NRING * 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gx = tid & 31, gy = tid >> 5;
    const int tiles_x = (P.w + TX - 1) / TX, tiles_y = (P.h + TY - 1) / TY, ntiles = tiles_x * tiles_y;
    const int step = gridDim.x;
    f4v acc = {0, 0, 0, 0};
    Own setA, setB;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    // prologue: the first tile's p tile, ring operands and own operands
    dma_p_tile<8>(P, s_o[0][0], s_o[0][1], (t % tiles_x) * TX, (t / tiles_x) * TY, lane, wv);
    dma_ring<8, 4>(P, s_ring[0], (t % tiles_x) * TX, (t / tiles_x) * TY, lane, wv);
    own_issue(P, setA, (t % tiles_x) * TX, (t / tiles_x) * TY, gx, gy);
    // one tile: `cur` was requested a whole tile ago, `nxt` is requested now; buf = which LDS set holds this tile
    auto tile = [&](Own &cur, Own &nxt, int tt, int buf) {
        const int tx0 = (tt % tiles_x) * TX, ty0 = (tt / tiles_x) * TY;
        const int tn = tt + step;
        if (tn < ntiles) {          // everything of the next tile is requested now
            const int nx0 = (tn % tiles_x) * TX, ny0 = (tn / tiles_x) * TY;
            dma_p_tile<8>(P, s_o[buf ^ 1][0], s_o[buf ^ 1][1], nx0, ny0, lane, wv);
            dma_ring<8, 4>(P, s_ring[buf ^ 1], nx0, ny0, lane, wv);
            own_issue(P, nxt, nx0, ny0, gx, gy);
            // in-order completion: once no more than what was requested since is outstanding, this tile's data has landed
            own_wait<WAITN>(cur);
        } else {
            own_wait<0>(cur);
        }
        __syncthreads();
        // ---- phase 1
        float *const so_u = s_o[buf][0], *const so_v = s_o[buf][1];
        float *const sn_u = s_n[buf][0], *const sn_v = s_n[buf][1];
        if (tid < NRING) {
            int rx, ry;
            if (tid < 34) { rx = tid - 1; ry = -1; }
            else if (tid < 68) { rx = tid - 35; ry = TY; }
            else if (tid < 84) { rx = -1; ry = tid - 68; }
            else { rx = TX / 4; ry = tid - 84; }
            f4v s = {0, 0, 0, 0};
#pragma unroll
            for (int op = 0; op < NOPS; op++) s += ld4(&s_ring[buf][(op * NRING + tid) * 4]);
            const f4v qu = stencil_reads(so_u, ry + 2, OFF + 4 * rx), qv = stencil_reads(so_v, ry + 2, OFF + 4 * rx);
            *(f4v *)&sn_u[(ry + 1) * COLS + OFF + 4 * rx] = qu + s;
            *(f4v *)&sn_v[(ry + 1) * COLS + OFF + 4 * rx] = qv + s;
        }
        {
            const f4v qu = stencil_reads(so_u, gy + 2, OFF + 4 * gx), qv = stencil_reads(so_v, gy + 2, OFF + 4 * gx);
            const f4v rnu = cur.ru + qu + cur.a1 + cur.a2 + cur.wx + cur.wys, rnv = cur.rv + qv + cur.a4 + cur.wy + cur.wxw;
            const f4v pnu = rnu * 0.5f + qu, pnv = rnv * 0.5f + qv;
            const int x0 = tx0 + 4 * gx, y = ty0 + gy;
            if (x0 < P.w && y < P.h) {
                const size_t off = (size_t)y * P.pitch + x0;
                st4nt(P.ro_u + off, rnu); st4nt(P.ro_v + off, rnv); st4nt(P.po_u + off, pnu); st4nt(P.po_v + off, pnv);
            }
            *(f4v *)&sn_u[(gy + 1) * COLS + OFF + 4 * gx] = pnu;
            *(f4v *)&sn_v[(gy + 1) * COLS + OFF + 4 * gx] = pnv;
            acc += rnu + rnv;
        }
        __syncthreads();
        // ---- phase 2
        acc += stencil_reads(sn_u, gy + 1, OFF + 4 * gx) + stencil_reads(sn_v, gy + 1, OFF + 4 * gx);
    };
    for (; t < ntiles; t += 2 * step) {
        tile(setA, setB, t, 0);
        if (t + step < ntiles) tile(setB, setA, t + step, 1);
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) P.sink[blockIdx.x] = acc.x;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// A: the production structure
// ---------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_pattern_a(Planes P)
{
    __shared__ __attribute__((aligned(16))) float s_o[2][(TY + 4) * COLS];
    __shared__ __attribute__((aligned(16))) float s_n[2][2][(TY + 2) * COLS];
    __shared__ __attribute__((aligned(16))) float s_ring[NOPS * NRING * 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (P.w + TX - 1) / TX, tiles_y = (P.h + TY - 1) / TY, ntiles = tiles_x * tiles_y;
    const int step = gridDim.x;
    f4v acc = {0, 0, 0, 0};
    int t = blockIdx.x;
    if (t >= ntiles) return;
    dma_p_tile<4>(P, s_o[0], s_o[1], (t % tiles_x) * TX, (t / tiles_x) * TY, lane, wv);
    dma_ring<4, 0>(P, s_ring, (t % tiles_x) * TX, (t / tiles_x) * TY, lane, wv);
    int par = 0;
    for (; t < ntiles; t += step, par ^= 1) {
        const int tx0 = (t % tiles_x) * TX, ty0 = (t / tiles_x) * TY;
        f4v o[2][8]; float ow[2];
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            const int gx = tid & 31, gy = (tid >> 5) + 8 * slot;
            const int x0 = clampi(tx0 + 4 * gx, 4, P.pitch - 4), y = clampi(ty0 + gy, 1, P.h - 1);
            const size_t off = (size_t)y * P.pitch + x0;
            o[slot][0] = ld4(P.ru + off); o[slot][1] = ld4(P.rv + off); o[slot][2] = ld4(P.a1 + off); o[slot][3] = ld4(P.a4 + off);
            o[slot][4] = ld4(P.a2 + off); o[slot][5] = ld4(P.wx + off); o[slot][6] = ld4(P.wy + off); o[slot][7] = ld4(P.wy + off - P.pitch);
            ow[slot] = P.wx[off - 1];
        }
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __syncthreads();
        float *const sn_u = s_n[par][0], *const sn_v = s_n[par][1];
        if (tid < NRING) {
            int rx, ry;
            if (tid < 34) { rx = tid - 1; ry = -1; }
            else if (tid < 68) { rx = tid - 35; ry = TY; }
            else if (tid < 84) { rx = -1; ry = tid - 68; }
            else { rx = TX / 4; ry = tid - 84; }
            f4v s = {0, 0, 0, 0};
#pragma unroll
            for (int op = 0; op < NOPS; op++) s += ld4(&s_ring[(op * NRING + tid) * 4]);
            const f4v qu = stencil_reads(s_o[0], ry + 2, OFF + 4 * rx), qv = stencil_reads(s_o[1], ry + 2, OFF + 4 * rx);
            *(f4v *)&sn_u[(ry + 1) * COLS + OFF + 4 * rx] = qu + s;
            *(f4v *)&sn_v[(ry + 1) * COLS + OFF + 4 * rx] = qv + s;
        }
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            const int gx = tid & 31, gy = (tid >> 5) + 8 * slot;
            const f4v qu = stencil_reads(s_o[0], gy + 2, OFF + 4 * gx), qv = stencil_reads(s_o[1], gy + 2, OFF + 4 * gx);
            const f4v rnu = o[slot][0] + qu + o[slot][2] + o[slot][4] + o[slot][5] + o[slot][7], rnv = o[slot][1] + qv + o[slot][3] + o[slot][6] + ow[slot];
            const f4v pnu = rnu * 0.5f + qu, pnv = rnv * 0.5f + qv;
            const int x0 = tx0 + 4 * gx, y = ty0 + gy;
            if (x0 < P.w && y < P.h) {
                const size_t off = (size_t)y * P.pitch + x0;
                st4nt(P.ro_u + off, rnu); st4nt(P.ro_v + off, rnv); st4nt(P.po_u + off, pnu); st4nt(P.po_v + off, pnv);
            }
            *(f4v *)&sn_u[(gy + 1) * COLS + OFF + 4 * gx] = pnu;
            *(f4v *)&sn_v[(gy + 1) * COLS + OFF + 4 * gx] = pnv;
            acc += rnu + rnv;
        }
        __syncthreads();
        const int tn = t + step;
        if (tn < ntiles) {
            dma_p_tile<4>(P, s_o[0], s_o[1], (tn % tiles_x) * TX, (tn / tiles_x) * TY, lane, wv);
            dma_ring<4, 0>(P, s_ring, (tn % tiles_x) * TX, (tn / tiles_x) * TY, lane, wv);
        }
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            const int gx = tid & 31, gy = (tid >> 5) + 8 * slot;
            acc += stencil_reads(sn_u, gy + 1, OFF + 4 * gx) + stencil_reads(sn_v, gy + 1, OFF + 4 * gx);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) P.sink[blockIdx.x] = acc.x;
}

template <typename F>
static double time_us(F launch, int reps = 15)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return t[t.size() / 2] * 1e3;
}

int main(int argc, char **argv)
{
    const int W = argc > 2 ? atoi(argv[1]) : 5000, H = argc > 2 ? atoi(argv[2]) : 5000;
    const int pitch = (W + 63) / 64 * 64;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("# %s, %d CUs; frame %d x %d, pitch %d; one launch moves %.3f GB of compulsory bytes (52 B/pixel)\n", prop.name, ncu, W, H, pitch, 52.0 * W * H / 1e9);
    const size_t plane = (size_t)pitch * (H + 2) + 64;
    float *arena; CK(hipMalloc(&arena, 14 * plane * sizeof(float))); CK(hipMemset(arena, 0, 14 * plane * sizeof(float)));
    double *sink; CK(hipMalloc(&sink, 4096 * sizeof(double)));
    Planes P;
    float *pl[13];
    for (int i = 0; i < 13; i++) pl[i] = arena + (size_t)i * plane + pitch;       // one row of slack in front (the row above row 0)
    P.ru = pl[0]; P.rv = pl[1]; P.pu = pl[2]; P.pv = pl[3]; P.a1 = pl[4]; P.a2 = pl[5]; P.a4 = pl[6]; P.wx = pl[7]; P.wy = pl[8];
    P.ro_u = pl[9]; P.ro_v = pl[10]; P.po_u = pl[11]; P.po_v = pl[12]; P.sink = sink; P.w = W; P.h = H; P.pitch = pitch;
    const double gb = 52.0 * W * H / 1e9;
    auto report = [&](const char *name, double us) { printf("%-52s %8.1f us  %6.0f GB/s  %.3f of 8 TB/s\n", name, us, gb / us * 1e6, gb / us * 1e6 / 8000.); fflush(stdout); };
    for (int rep = 0; rep < 2; rep++) {
        report("A  256 thr x 2 groups, 2 WG/CU (production shape)", time_us([&] { hipLaunchKernelGGL(k_pattern_a, dim3(2 * ncu), dim3(256), 0, 0, P); }));
        report("B  512 thr, 1 WG/CU, next tile requested first, vmcnt(16)", time_us([&] { hipLaunchKernelGGL((k_pattern_b<16>), dim3(ncu), dim3(512), 0, 0, P); }));
        report("B  the same, waiting for everything (vmcnt(0))", time_us([&] { hipLaunchKernelGGL((k_pattern_b<0>), dim3(ncu), dim3(512), 0, 0, P); }));
    }
    return 0;
}
