// HBM microbenchmark 3: does the LAYOUT of a tile matter?  The PCG kernels read 128 x 16 tiles of ~15 planes stored row
// by row (pitch 5056 floats): a tile is 16 segments of 512 B, 20 KB apart, in every plane.  This streams the same number
// of planes with the same workgroup shape (256 threads, two float4 per thread and plane, 512-workgroup persistent grid)
//   A  from row-major planes, tile by tile (what the kernels do),
//   B  from "tile-major" planes in which a tile's 8 KB are contiguous,
// and prints both rates.  No LDS, no barriers, all loads of a tile issued before the first use.
//   hipcc --offload-arch=gfx950 -O3 -o membench3 tools/micro/membench3.hip && ./membench3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Ptrs { float *p[16]; };

template <int R, int W, bool TILEMAJOR>
__global__ __launch_bounds__(256) void k_tiles(Ptrs P, int w, int h, int pitch)
{
    const int tiles_x = w / 128, tiles_y = h / 16, ntiles = tiles_x * tiles_y;
    const int gx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t o[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (TILEMAJOR) o[s] = (size_t)t * 2048 + (size_t)(gy + 8 * s) * 128 + 4 * gx;
            else o[s] = (size_t)((t / tiles_x) * 16 + gy + 8 * s) * pitch + (t % tiles_x) * 128 + 4 * gx;
        }
        float4 v[R][2];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int s = 0; s < 2; s++) v[r][s] = *reinterpret_cast<const float4 *>(P.p[r] + o[s]);
        float4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int s = 0; s < 2; s++) { acc[s].x += v[r][s].x; acc[s].y += v[r][s].y; acc[s].z += v[r][s].z; acc[s].w += v[r][s].w; }
#pragma unroll
        for (int q = 0; q < W; q++)
#pragma unroll
            for (int s = 0; s < 2; s++) { float4 ov = acc[s]; ov.x += q; *reinterpret_cast<float4 *>(P.p[R + q] + o[s]) = ov; }
    }
}

template <int R, int W, bool TM>
static double run(float *arena, int w, int h, int pitch, int grid)
{
    Ptrs P;
    const size_t plane = (size_t)pitch * h + 4096;
    for (int i = 0; i < R + W; i++) P.p[i] = arena + i * plane;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_tiles<R, W, TM>), dim3(grid), dim3(256), 0, 0, P, w, h, pitch);
    CK(hipEventRecord(a));
    const int reps = 8;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_tiles<R, W, TM>), dim3(grid), dim3(256), 0, 0, P, w, h, pitch);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double bytes = (double)(R + W) * (w / 128) * (h / 16) * 2048.0 * 4.0;
    return bytes / ms / 1e6;
}

int main()
{
    const int w = 4992, h = 4992, pitch = 5056;       // whole tiles only
    float *arena; const size_t fl = (size_t)16 * ((size_t)pitch * h + 4096);
    CK(hipMalloc(&arena, fl * 4)); CK(hipMemset(arena, 0, fl * 4));
    for (int grid : {512, 768, 1024}) {
        printf("grid %4d  10 reads + 5 writes: row-major tiles %5.0f GB/s, tile-major %5.0f GB/s\n", grid,
               run<10, 5, false>(arena, w, h, pitch, grid), run<10, 5, true>(arena, w, h, pitch, grid));
        printf("grid %4d   9 reads + 4 writes: row-major tiles %5.0f GB/s, tile-major %5.0f GB/s\n", grid,
               run<9, 4, false>(arena, w, h, pitch, grid), run<9, 4, true>(arena, w, h, pitch, grid));
    }
    return 0;
}
