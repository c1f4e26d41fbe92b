// iter_sync.hip -- what would several PCG iterations of a streaming level in ONE launch cost?  (round 3)
//
// The finest levels pay ~10 us of fixed cost per launch plus ~2 us of kernel-to-kernel gap, 270 times per level and pyramid: 17 % of a
// 2000^2 launch.  The alternative is a persistent grid (512 workgroups, two per CU, all resident) that meets at a grid-wide exchange of
// its partial sums instead of at a kernel boundary.  The XCDs' L2s are not coherent with each other, so what one workgroup wrote with
// plain stores has to be written back (release fence at agent scope: buffer_wbl2) before the exchange and the reader's L2 invalidated
// (acquire fence: buffer_inv) after it.  This program measures exactly that, on a stand-in with the PCG launch's traffic and a result
// that proves the hand-off: per "iteration" every tile (128 x 16 pixels, round-robin over the grid as in k_pcg_fused_q_dma) reads nine
// planes and the rows above / below it of one of them (other workgroups' pixels, other XCDs), writes four, and leaves seven partial sums
// per workgroup that everybody folds before the next iteration.  The stencil is exact in float (small integers), so the planes after K
// iterations are compared bit for bit between
//   L  one launch per iteration (the production form), and
//   P  one launch for all K iterations: release fence, seven tagged granules per workgroup published, all 512 x 7 polled (the fold IS
//      the barrier, as in k_pcg_solve_mid), acquire fence.
// Every wait is bounded (a flag ends the launch): this cannot hang the GPU.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o iter_sync iter_sync.hip      Run: ./iter_sync [W H [K]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TX = 128, TY = 16, NWG = 512, NK = 7;
typedef float f4v __attribute__((ext_vector_type(4)));

struct Args {
    float *r[2][2], *p[2][2];            // [parity][u/v]
    const float *op[5];
    double *parts[2];                    // L: [parity][NK * NWG]
    unsigned long long *gran[2];         // P: tagged granules, two per double
    unsigned *abort_word;
    double *sink;                        // folded sums per iteration (to keep them alive and to compare)
    int w, h, pitch;
};

__device__ __forceinline__ f4v ld4(const float *p) { return *(const f4v *)p; }

// one iteration's tiles for this workgroup; returns the thread's partial sums
__device__ __forceinline__ void tiles(const Args &A, int k, double scale, double acc[NK])
{
    const int in = k & 1, out = in ^ 1;
    const int tiles_x = (A.w + TX - 1) / TX, tiles_y = (A.h + TY - 1) / TY, ntiles = tiles_x * tiles_y;
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx0 = (t % tiles_x) * TX, ty0 = (t / tiles_x) * TY;
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            const int gx = tid & 31, gy = (tid >> 5) + 8 * slot;
            const int x0 = tx0 + 4 * gx, y = ty0 + gy;
            if (x0 >= A.w || y >= A.h) continue;
            const size_t o = (size_t)y * A.pitch + x0;
            const size_t on = (size_t)(y > 0 ? y - 1 : y) * A.pitch + x0, os = (size_t)(y < A.h - 1 ? y + 1 : y) * A.pitch + x0;
            const f4v ru = ld4(A.r[in][0] + o), rv = ld4(A.r[in][1] + o), pu = ld4(A.p[in][0] + o), pv = ld4(A.p[in][1] + o);
            const f4v pn = ld4(A.p[in][0] + on), ps = ld4(A.p[in][0] + os), qn = ld4(A.p[in][1] + on), qs = ld4(A.p[in][1] + os);
            f4v c[5];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = ld4(A.op[j] + o);
            // small integers: exact in float for the iteration counts used here (values stay below 2^24)
            f4v npu, npv, nru, nrv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                npu[e] = fmodf(pn[e] + ps[e] + pu[e] + c[0][e], 4096.f);
                npv[e] = fmodf(qn[e] + qs[e] + pv[e] + c[1][e], 4096.f);
                nru[e] = fmodf(ru[e] + npu[e] + c[2][e] + c[3][e], 4096.f);
                nrv[e] = fmodf(rv[e] + npv[e] + c[4][e], 4096.f);
                acc[0] += (double)npu[e]; acc[1] += (double)npv[e]; acc[2] += (double)nru[e]; acc[3] += (double)nrv[e];
                acc[4] += (double)(npu[e] * 0.5f); acc[5] += (double)(nru[e] * 0.25f); acc[6] += scale;
            }
            *(f4v *)(A.p[out][0] + o) = npu; *(f4v *)(A.p[out][1] + o) = npv;
            *(f4v *)(A.r[out][0] + o) = nru; *(f4v *)(A.r[out][1] + o) = nrv;
        }
    }
}

__device__ __forceinline__ void block_sums(double acc[NK], double *s_red, double tot[NK])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NK; j++) {
        double v = acc[j];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
        if ((tid & 63) == 0) s_red[j * 4 + (tid >> 6)] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NK; j++) tot[j] = s_red[j * 4] + s_red[j * 4 + 1] + s_red[j * 4 + 2] + s_red[j * 4 + 3];
    __syncthreads();
}

constexpr int kPad = 70 * 1024;         // LDS that leaves room for two workgroups per CU, as the production kernel

// ---- L: one launch per iteration
__global__ __launch_bounds__(256, 2) void k_launch(Args A, int k)
{
    __shared__ double s_red[NK * 4], s_tot[NK];
    __shared__ char s_pad[kPad];
    ((volatile char *)s_pad)[threadIdx.x * 256] = 1;
    const int tid = threadIdx.x;
    double scale = 1.;
    if (k > 0) {        // fold the previous iteration's sums: every workgroup, same order
        double t[NK];
#pragma unroll
        for (int j = 0; j < NK; j++) t[j] = A.parts[(k + 1) & 1][j * NWG + tid] + A.parts[(k + 1) & 1][j * NWG + tid + 256];
        double tot[NK];
        block_sums(t, s_red, tot);
        scale = tot[6] > 0. ? 1. : 2.;
        if (blockIdx.x == 0 && tid == 0) for (int j = 0; j < NK; j++) A.sink[k * NK + j] = tot[j];
    }
    double acc[NK] = {0, 0, 0, 0, 0, 0, 0}, tot[NK];
    tiles(A, k, scale, acc);
    block_sums(acc, s_red, tot);
    if (tid == 0) for (int j = 0; j < NK; j++) A.parts[k & 1][j * NWG + blockIdx.x] = tot[j];
    (void)s_tot;
}

// ---- P: all iterations in one launch
__device__ __forceinline__ bool keep_waiting(const Args &A, unsigned &spins)
{
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) != 0u) return true;
    if (__hip_atomic_load(A.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if (spins > (1u << 22)) { __hip_atomic_store(A.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    return true;
}

template <int FENCE>     // 1: release / acquire fences at agent scope (what correctness needs); 0: none (what the exchange alone costs; WRONG results)
__global__ __launch_bounds__(256, 2) void k_persist(Args A, int K, unsigned seq)
{
    __shared__ double s_red[NK * 4];
    __shared__ int s_ok;
    __shared__ char s_pad[kPad];
    ((volatile char *)s_pad)[threadIdx.x * 256] = 1;
    const int tid = threadIdx.x;
    double scale = 1.;
    for (int k = 0; k < K; k++) {
        if (k > 0) {
            // poll the 512 x 7 sums of iteration k - 1 (two workgroups' worth per thread and kind): the data is the flag
            const unsigned tag = seq * 4096u + (unsigned)k;
            const unsigned long long *g = A.gran[(k + 1) & 1];
            double t[NK];
            bool ok = true;
            unsigned spins = 0;
#pragma unroll
            for (int j = 0; j < NK; j++) {
                double v2[2];
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int src = tid + 256 * half;
                    unsigned long long lo, hi;
                    for (;;) {
                        lo = __hip_atomic_load(&g[(j * NWG + src) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi = __hip_atomic_load(&g[(j * NWG + src) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
                        if (!keep_waiting(A, spins)) { ok = false; break; }
                    }
                    v2[half] = __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
                }
                t[j] = v2[0] + v2[1];
            }
            if (tid == 0) s_ok = 1;
            __syncthreads();
            if (!ok) s_ok = 0;
            __syncthreads();
            if (!s_ok) return;
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double tot[NK];
            block_sums(t, s_red, tot);
            scale = tot[6] > 0. ? 1. : 2.;
            if (blockIdx.x == 0 && tid == 0) for (int j = 0; j < NK; j++) A.sink[k * NK + j] = tot[j];
        }
        double acc[NK] = {0, 0, 0, 0, 0, 0, 0}, tot[NK];
        tiles(A, k, scale, acc);
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // this thread's stores written back before anything is published
        block_sums(acc, s_red, tot);                                         // (its barriers order every thread's fence before thread 0 .. 6's stores)
        if (tid < NK) {
            const unsigned tag = seq * 4096u + (unsigned)(k + 1);
            const unsigned long long bits = (unsigned long long)__double_as_longlong(tot[tid]);
            unsigned long long *g = A.gran[k & 1] + (size_t)(tid * NWG + blockIdx.x) * 2;
            __hip_atomic_store(&g[0], ((unsigned long long)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&g[1], ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv)
{
    const int W = argc > 2 ? atoi(argv[1]) : 2000, H = argc > 2 ? atoi(argv[2]) : 2000, K = argc > 3 ? atoi(argv[3]) : 60;
    const int pitch = (W + 63) / 64 * 64;
    const size_t n = (size_t)pitch * H;
    Args A;
    memset(&A, 0, sizeof A);
    A.w = W; A.h = H; A.pitch = pitch;
    std::vector<float> init(n);
    float *planes[13];
    for (int i = 0; i < 13; i++) CK(hipMalloc((void **)&planes[i], n * sizeof(float)));
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { A.r[i][j] = planes[i * 2 + j]; A.p[i][j] = planes[4 + i * 2 + j]; }
    for (int j = 0; j < 5; j++) {
        for (size_t i = 0; i < n; i++) init[i] = (float)((i * (j + 3) + j) % 7);
        CK(hipMemcpy(planes[8 + j], init.data(), n * sizeof(float), hipMemcpyHostToDevice));
        A.op[j] = planes[8 + j];
    }
    for (int i = 0; i < 2; i++) CK(hipMalloc((void **)&A.parts[i], NK * NWG * sizeof(double)));
    for (int i = 0; i < 2; i++) { CK(hipMalloc((void **)&A.gran[i], NK * NWG * 2 * sizeof(unsigned long long))); CK(hipMemset(A.gran[i], 0, NK * NWG * 2 * sizeof(unsigned long long))); }
    CK(hipMalloc((void **)&A.abort_word, 4)); CK(hipMemset(A.abort_word, 0, 4));
    CK(hipMalloc((void **)&A.sink, (size_t)(K + 1) * NK * sizeof(double)));
    auto reset = [&]() {
        for (int j = 0; j < 4; j++) {
            for (size_t i = 0; i < n; i++) init[i] = (float)((i * 5 + j) % 11);
            CK(hipMemcpy(planes[j < 2 ? j : 4 + (j - 2)], init.data(), n * sizeof(float), hipMemcpyHostToDevice));      // parity 0 of r and p
        }
        CK(hipMemset(A.sink, 0, (size_t)(K + 1) * NK * sizeof(double)));
    };
    auto fetch = [&](std::vector<float> &out, std::vector<double> &sums) {
        out.resize(4 * n); sums.resize((size_t)K * NK);
        const int fin = K & 1;
        for (int j = 0; j < 2; j++) {
            CK(hipMemcpy(out.data() + (size_t)j * n, A.r[fin][j], n * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(out.data() + (size_t)(2 + j) * n, A.p[fin][j], n * sizeof(float), hipMemcpyDeviceToHost));
        }
        CK(hipMemcpy(sums.data(), A.sink, (size_t)K * NK * sizeof(double), hipMemcpyDeviceToHost));
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> refp, gotp;
    std::vector<double> refs, gots;
    float ms;
    const double mb = 13.0 * 4 * (double)W * H / 1e6;
    // L
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        reset();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_launch, dim3(NWG), dim3(256), 0, 0, A, k);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    fetch(refp, refs);
    printf("%dx%d, %d iterations, 13 planes of traffic = %.0f MB per iteration\n", W, H, K, mb);
    printf("L  one launch per iteration:          %8.2f us per iteration (%.2f TB/s)\n", best * 1e3 / K, mb / (best * 1e3 / K));
    unsigned seq = 1;
    for (int fence = 1; fence >= 0; fence--) {
        best = 1e9f;
        bool aborted = false;
        for (int rep = 0; rep < 3; rep++) {
            reset();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (fence) hipLaunchKernelGGL(k_persist<1>, dim3(NWG), dim3(256), 0, 0, A, K, seq);
            else hipLaunchKernelGGL(k_persist<0>, dim3(NWG), dim3(256), 0, 0, A, K, seq);
            seq++;
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned ab = 0;
            CK(hipMemcpy(&ab, A.abort_word, 4, hipMemcpyDeviceToHost));
            if (ab) { aborted = true; CK(hipMemset(A.abort_word, 0, 4)); break; }
            if (ms < best) best = ms;
        }
        if (aborted) { printf("P  fence=%d: a wait timed out, launch abandoned\n", fence); continue; }
        fetch(gotp, gots);
        size_t badp = 0, bads = 0;
        for (size_t i = 0; i < gotp.size(); i++) badp += memcmp(&gotp[i], &refp[i], 4) != 0;
        for (size_t i = 0; i < gots.size(); i++) bads += memcmp(&gots[i], &refs[i], 8) != 0;
        printf("P  one launch, fences %s: %8.2f us per iteration (%.2f TB/s)   planes differing from L in %zu of %zu floats, folded sums in %zu of %zu\n",
               fence ? "release/acquire" : "NONE (wrong) ", best * 1e3 / K, mb / (best * 1e3 / K), badp, gotp.size(), bads, gots.size());
    }
    return 0;
}
