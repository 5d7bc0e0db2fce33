// scratch: tile reads of 128-byte row segments, rows 4 KB apart (y tiles) vs 2 MB apart (z tiles), on a 1 GiB
// field carved from a 1 GiB or a 24 GiB allocation: is the z pattern bound by address translation?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k_tile(const double* __restrict__ f, double* __restrict__ out, long rstride,
                                               long ostride, int ntx, int ntiles)
{
    const int cy = threadIdx.x >> 3, cc = threadIdx.x & 7;
    double s = 0;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const double* p = f + (long)(tl / ntx) * ostride + (long)(tl % ntx) * 16;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double2 v = *reinterpret_cast<const double2*>(p + (long)(cy + 128 * i) * rstride + 2 * cc);
            s += v.x + v.y;
        }
    }
    if (s == 123.456) out[0] = s;
}
__global__ void __launch_bounds__(1024) k_tile_w(double* __restrict__ f, long rstride, long ostride, int ntx, int ntiles,
                                                 int rmw)
{
    const int cy = threadIdx.x >> 3, cc = threadIdx.x & 7;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        double* p = f + (long)(tl / ntx) * ostride + (long)(tl % ntx) * 16;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double2* q = reinterpret_cast<double2*>(p + (long)(cy + 128 * i) * rstride + 2 * cc);
            double2 v = make_double2(1.0, 2.0);
            if (rmw) { v = *q; v.x += 1.0; v.y += 2.0; }
            *q = v;
        }
    }
}
static void runw(const char* name, double* f, long rstride, long ostride, int rmw)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntx = 32, ntiles = 32 * 512;
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_tile_w, dim3(256), dim3(1024), 0, 0, f, rstride, ostride, ntx, ntiles, rmw);
    hipEventRecord(e0);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL(k_tile_w, dim3(256), dim3(1024), 0, 0, f, rstride, ostride, ntx, ntiles, rmw);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, (rmw ? 2 : 1) * 1073.7 / ms);
}
static void run(const char* name, const double* f, double* out, long rstride, long ostride)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntx = 32, ntiles = 32 * 512;
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_tile, dim3(256), dim3(1024), 0, 0, f, out, rstride, ostride, ntx, ntiles);
    hipEventRecord(e0);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL(k_tile, dim3(256), dim3(1024), 0, 0, f, out, rstride, ostride, ntx, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, 1073.7 / ms);
}
int main()
{
    double *a, *big, *out; hipMalloc(&out, 8);
    hipMalloc(&a, (size_t)1 << 30); hipMemset(a, 0, (size_t)1 << 30);
    run("1 GiB alloc, y tiles (rows 4 KB apart)", a, out, 512, 512L * 512);
    run("1 GiB alloc, z tiles (rows 2 MB apart)", a, out, 512L * 512, 512);
    runw("1 GiB alloc, y tiles, write", a, 512, 512L * 512, 0);
    runw("1 GiB alloc, z tiles, write", a, 512L * 512, 512, 0);
    runw("1 GiB alloc, y tiles, read-modify-write", a, 512, 512L * 512, 1);
    runw("1 GiB alloc, z tiles, read-modify-write", a, 512L * 512, 512, 1);
    if (hipMalloc(&big, (size_t)24 << 30) == hipSuccess) {
        hipMemset(big, 0, (size_t)24 << 30);
        const double* f = big + ((size_t)7 << 27);  // 7 GiB into the arena
        run("24 GiB arena, y tiles", f, out, 512, 512L * 512);
        run("24 GiB arena, z tiles", f, out, 512L * 512, 512);
    } else printf("24 GiB hipMalloc failed\n");
    return 0;
}
