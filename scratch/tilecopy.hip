// Streaming rate of the y-tile access pattern as a function of the segment width: a workgroup of NPEN waves owns
// NPEN x-adjacent y pencils of 512 rows (row segments of NPEN * 8 bytes, rows `pitch` doubles apart), reads the
// tile (4 x 16-byte loads per thread), writes it to the same place of another block; persistent over tiles.
//   hipcc -O2 --offload-arch=gfx950 scratch/tilecopy.hip -o scratch/tilecopy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
// nontemporal 16-byte accesses (round 4: a flat copy gains 10 % from them, scratch/copybench.hip)
template <int NTH> __device__ __forceinline__ double2 ldg(const double *p)
{
    if (NTH) { d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p)); return make_double2(v.x, v.y); }
    return *reinterpret_cast<const double2 *>(p);
}
template <int NTH> __device__ __forceinline__ void stg(double *p, double2 v)
{
    if (NTH) { d2v w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p)); }
    else *reinterpret_cast<double2 *>(p) = v;
}
template <int NPEN, int NT = NPEN * 64, int NTL = 0, int NTS = 0>
__global__ void __launch_bounds__(NT) k(const double *__restrict__ a, double *__restrict__ b, int ntx, int ntiles,
                                               long prow, long pplane)
{
    constexpr int CPR = NPEN / 2;              // double2 per row segment
    constexpr int RPI = NT / CPR;              // rows per load instruction
    constexpr int NL = 512 / RPI;
    const int cc = threadIdx.x % CPR, cy = threadIdx.x / CPR;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = (long)(tl / ntx) * pplane + (long)(tl % ntx) * NPEN;
        double2 v[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) v[i] = ldg<NTL>(a + off + (long)(cy + RPI * i) * prow + 2 * cc);
#pragma unroll
        for (int i = 0; i < NL; i++) stg<NTS>(b + off + (long)(cy + RPI * i) * prow + 2 * cc, v[i]);
    }
}
// the accumulating pattern of the transeq tile kernels: c (+)= a, tile by tile (R a, R c, W c = 3 streams)
template <int NPEN, int NT = NPEN * 64, int NTA = 0, int NTC = 0>
__global__ void __launch_bounds__(NT) kacc(const double *__restrict__ a, double *c, int ntx, int ntiles, long prow, long pplane)
{
    constexpr int CPR = NPEN / 2, RPI = NT / CPR, NL = 512 / RPI;
    const int cc = threadIdx.x % CPR, cy = threadIdx.x / CPR;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = (long)(tl / ntx) * pplane + (long)(tl % ntx) * NPEN;
        double2 v[NL], w[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) v[i] = ldg<NTA>(a + off + (long)(cy + RPI * i) * prow + 2 * cc);
#pragma unroll
        for (int i = 0; i < NL; i++) w[i] = ldg<NTC>(c + off + (long)(cy + RPI * i) * prow + 2 * cc);
#pragma unroll
        for (int i = 0; i < NL; i++)
            stg<NTC>(c + off + (long)(cy + RPI * i) * prow + 2 * cc, make_double2(v[i].x + w[i].x, v[i].y + w[i].y));
    }
}
int main()
{
    const int nx = 512, ny = 512, nz = 512, nxp = 528;
    const size_t n = (size_t)nxp * ny * nz;
    double *a, *b;
    (void)hipMalloc(&a, n * 8 + (1 << 20)); (void)hipMalloc(&b, n * 8 + (1 << 20)); (void)hipMemset(a, 0, n * 8);
    b += 528 * 3;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char *nm, auto f) {
        for (int i = 0; i < 2; i++) f();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; i++) f();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %7.3f ms %8.1f GB/s\n", nm, ms, 2.0 * nx * ny * nz * 8 / ms * 1e-6);
    };
    const long pxy = (long)nxp * ny;
    // y pencils: rows nxp apart, tiles stacked over z; z pencils: rows pxy apart, tiles stacked over y
    run("y, 16 pencils (128 B), 256 WGs x 1024", [&] { hipLaunchKernelGGL(k<16>, dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run("y,  8 pencils ( 64 B), 512 WGs x  512", [&] { hipLaunchKernelGGL(k<8>, dim3(512), dim3(512), 0, 0, a, b, nx / 8, nx / 8 * nz, (long)nxp, pxy); });
    run("y,  8 pencils ( 64 B), 1024 WGs x 512", [&] { hipLaunchKernelGGL(k<8>, dim3(1024), dim3(512), 0, 0, a, b, nx / 8, nx / 8 * nz, (long)nxp, pxy); });
    run("z, 16 pencils (128 B), 256 WGs x 1024", [&] { hipLaunchKernelGGL(k<16>, dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run("z,  8 pencils ( 64 B), 512 WGs x  512", [&] { hipLaunchKernelGGL(k<8>, dim3(512), dim3(512), 0, 0, a, b, nx / 8, nx / 8 * ny, pxy, (long)nxp); });
    run("y, 16 pencils (128 B), 512 WGs x 1024", [&] { hipLaunchKernelGGL(k<16>, dim3(512), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    auto run3 = [&](const char *nm, auto f) {
        for (int i = 0; i < 2; i++) f();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; i++) f();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %7.3f ms %8.1f GB/s\n", nm, ms, 3.0 * nx * ny * nz * 8 / ms * 1e-6);
    };
    run3("y acc, 16 pencils (128 B), 3 streams", [&] { hipLaunchKernelGGL((kacc<16>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run3("z acc, 16 pencils (128 B), 3 streams", [&] { hipLaunchKernelGGL((kacc<16>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run3("y acc, 32 pencils (256 B), 3 streams", [&] { hipLaunchKernelGGL((kacc<32, 1024>), dim3(256), dim3(1024), 0, 0, a, b, nx / 32, nx / 32 * nz, (long)nxp, pxy); });
    run3("z acc, 32 pencils (256 B), 3 streams", [&] { hipLaunchKernelGGL((kacc<32, 1024>), dim3(256), dim3(1024), 0, 0, a, b, nx / 32, nx / 32 * ny, pxy, (long)nxp); });
    run("y, 32 pencils (256 B), 256 WGs x 1024", [&] { hipLaunchKernelGGL((k<32, 1024>), dim3(256), dim3(1024), 0, 0, a, b, nx / 32, nx / 32 * nz, (long)nxp, pxy); });
    run("z, 32 pencils (256 B), 256 WGs x 1024", [&] { hipLaunchKernelGGL((k<32, 1024>), dim3(256), dim3(1024), 0, 0, a, b, nx / 32, nx / 32 * ny, pxy, (long)nxp); });
    run("y, 64 pencils (512 B), 256 WGs x 1024", [&] { hipLaunchKernelGGL((k<64, 1024>), dim3(256), dim3(1024), 0, 0, a, b, nx / 64, nx / 64 * nz, (long)nxp, pxy); });
    run("z, 64 pencils (512 B), 256 WGs x 1024", [&] { hipLaunchKernelGGL((k<64, 1024>), dim3(256), dim3(1024), 0, 0, a, b, nx / 64, nx / 64 * ny, pxy, (long)nxp); });
    // round 4: the same patterns with nontemporal accesses
    run("y, 16 pencils, nt loads", [&] { hipLaunchKernelGGL((k<16, 1024, 1, 0>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run("y, 16 pencils, nt stores", [&] { hipLaunchKernelGGL((k<16, 1024, 0, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run("y, 16 pencils, nt loads + stores", [&] { hipLaunchKernelGGL((k<16, 1024, 1, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run("z, 16 pencils, nt loads", [&] { hipLaunchKernelGGL((k<16, 1024, 1, 0>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run("z, 16 pencils, nt stores", [&] { hipLaunchKernelGGL((k<16, 1024, 0, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run("z, 16 pencils, nt loads + stores", [&] { hipLaunchKernelGGL((k<16, 1024, 1, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run("z, 32 pencils, nt loads + stores", [&] { hipLaunchKernelGGL((k<32, 1024, 1, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 32, nx / 32 * ny, pxy, (long)nxp); });
    run("y, 32 pencils, nt loads + stores", [&] { hipLaunchKernelGGL((k<32, 1024, 1, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 32, nx / 32 * nz, (long)nxp, pxy); });
    run3("y acc, 16 pencils, nt a", [&] { hipLaunchKernelGGL((kacc<16, 1024, 1, 0>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run3("y acc, 16 pencils, nt c", [&] { hipLaunchKernelGGL((kacc<16, 1024, 0, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run3("y acc, 16 pencils, nt a + c", [&] { hipLaunchKernelGGL((kacc<16, 1024, 1, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run3("z acc, 16 pencils, nt a", [&] { hipLaunchKernelGGL((kacc<16, 1024, 1, 0>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run3("z acc, 16 pencils, nt c", [&] { hipLaunchKernelGGL((kacc<16, 1024, 0, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    run3("z acc, 16 pencils, nt a + c", [&] { hipLaunchKernelGGL((kacc<16, 1024, 1, 1>), dim3(256), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * ny, pxy, (long)nxp); });
    // more workgroups than CUs (the tile kernels are persistent over 256)
    run("y, 16 pencils, 1024 WGs x 1024 (non-persistent-ish)", [&] { hipLaunchKernelGGL((k<16>), dim3(1024), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    run("y, 16 pencils, nt l+s, 512 WGs x 1024", [&] { hipLaunchKernelGGL((k<16, 1024, 1, 1>), dim3(512), dim3(1024), 0, 0, a, b, nx / 16, nx / 16 * nz, (long)nxp, pxy); });
    return 0;
}
