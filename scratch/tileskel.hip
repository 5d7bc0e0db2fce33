// Memory skeleton of k_ytile_transeq3 (round 4; scratch, not part of the product): per tile and component the workgroup
// reads the 16 x 512 tile of u_c (prefetched one component ahead), spends DELAY microseconds without touching memory
// (s_sleep: what the three solves take in the real kernel: 5 us), reads the old rhs_c rows, adds, stores rhs_c -- the
// same addresses, the same 1024-thread workgroups, one per CU, persistent over 8192 tiles; no LDS, no arithmetic.
// EARLY = 1: the old rhs rows are requested BEFORE the delay (what "requesting them after the first solve" did in the
// real kernel).  Answers: does the memory system deliver this access mix at the tile copy's rate when the requests
// are spread the way the real kernel spreads them, and how much of the delay hides?
//   hipcc -O2 --offload-arch=gfx950 scratch/tileskel.hip -o scratch/tileskel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
template <int NT> __device__ __forceinline__ double2 ldg(const double *p)
{
    if (NT) { d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p)); return make_double2(v.x, v.y); }
    return *reinterpret_cast<const double2 *>(p);
}
template <int NT> __device__ __forceinline__ void stg(double *p, double2 v)
{
    if (NT) { d2v w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p)); }
    else *reinterpret_cast<double2 *>(p) = v;
}
__device__ __forceinline__ void delay_us(int sleeps)
{
    for (int k = 0; k < sleeps; k++) __builtin_amdgcn_s_sleep(8);  // 8 x 64 clocks
}
// LDSX = 1: + the real kernel's trips through the LDS tile and their three barriers per component (to_tile ; barrier ;
// pick ... put ; barrier ; read back + add + store ; barrier), still without the solves
template <int EARLY, int NT, int LDSX = 0>
__global__ void __launch_bounds__(1024) k_skel(const double *u0, const double *u1, const double *u2, double *r0, double *r1,
                                               double *r2, int ntx, int ntiles, long prow, long pplane, int sleeps)
{
    constexpr int NL = 4, TP = 516;
    extern __shared__ double tile[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cc = threadIdx.x & 7, cy = threadIdx.x >> 3;
    auto off_of = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * 16; };
    auto load = [&](double2 (&v)[NL], const double *f, long off, bool nt) {
#pragma unroll
        for (int i = 0; i < NL; i++) v[i] = nt ? ldg<NT>(f + off + (long)(cy + 128 * i) * prow + 2 * cc) : ldg<0>(f + off + (long)(cy + 128 * i) * prow + 2 * cc);
    };
    double2 nxt[NL];
    if ((int)blockIdx.x < ntiles) load(nxt, u0, off_of(blockIdx.x), false);
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = off_of(tl);
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            double2 v[NL], old[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) v[i] = nxt[i];
            double b[8];
            if (LDSX) {
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    tile[(2 * cc) * TP + cy + 128 * i] = v[i].x;
                    tile[(2 * cc + 1) * TP + cy + 128 * i] = v[i].y;
                }
                __syncthreads();
                const double2 *src = reinterpret_cast<const double2 *>(tile + wave * TP + lane * 8);
#pragma unroll
                for (int m = 0; m < 4; m++) { const double2 t = src[m]; b[2 * m] = t.x; b[2 * m + 1] = t.y; }
            } else {
                __syncthreads();
            }
            const int tn = tl + gridDim.x;
            const double *nsrc = c == 0 ? u1 : (c == 1 ? u2 : u0);
            if (c < 2 || tn < ntiles) load(nxt, nsrc, c < 2 ? off : off_of(tn), false);
            double *o = c == 0 ? r0 : (c == 1 ? r1 : r2);
            if (EARLY) load(old, o, off, true);
            delay_us(sleeps);
            if (!EARLY) load(old, o, off, true);
            if (LDSX) {
                double2 *dst = reinterpret_cast<double2 *>(tile + wave * TP + lane * 8);
#pragma unroll
                for (int m = 0; m < 4; m++) dst[m] = make_double2(b[2 * m] * 0.5, b[2 * m + 1] * 0.5);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < NL; i++) v[i] = make_double2(tile[(2 * cc) * TP + cy + 128 * i], tile[(2 * cc + 1) * TP + cy + 128 * i]);
            } else {
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < NL; i++)
                stg<NT>(o + off + (long)(cy + 128 * i) * prow + 2 * cc, make_double2(v[i].x + old[i].x, v[i].y + old[i].y));
            __syncthreads();
        }
    }
}
// W = 32 pencils per tile: 256-byte row segments instead of 128-byte ones (would need compressed lane tables to fit the
// 132 KB tile; the question here is only what the memory system makes of the wider segments).  No LDS trips.
template <int NT>
__global__ void __launch_bounds__(1024) k_skel32(const double *u0, const double *u1, const double *u2, double *r0, double *r1,
                                                 double *r2, int ntx, int ntiles, long prow, long pplane, int sleeps)
{
    constexpr int NL = 8;
    const int cc = threadIdx.x & 15, cy = threadIdx.x >> 4;
    auto off_of = [&](int tl) { return (long)(tl / ntx) * pplane + (long)(tl % ntx) * 32; };
    auto load = [&](double2 (&v)[NL], const double *f, long off, bool nt) {
#pragma unroll
        for (int i = 0; i < NL; i++) v[i] = nt ? ldg<NT>(f + off + (long)(cy + 64 * i) * prow + 2 * cc) : ldg<0>(f + off + (long)(cy + 64 * i) * prow + 2 * cc);
    };
    double2 nxt[NL];
    if ((int)blockIdx.x < ntiles) load(nxt, u0, off_of(blockIdx.x), false);
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long off = off_of(tl);
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            double2 v[NL], old[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) v[i] = nxt[i];
            __syncthreads();
            const int tn = tl + gridDim.x;
            const double *nsrc = c == 0 ? u1 : (c == 1 ? u2 : u0);
            if (c < 2 || tn < ntiles) load(nxt, nsrc, c < 2 ? off : off_of(tn), false);
            double *o = c == 0 ? r0 : (c == 1 ? r1 : r2);
            delay_us(sleeps);
            load(old, o, off, true);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NL; i++)
                stg<NT>(o + off + (long)(cy + 64 * i) * prow + 2 * cc, make_double2(v[i].x + old[i].x, v[i].y + old[i].y));
            __syncthreads();
        }
    }
}
int main()
{
    const int nx = 512, ny = 512, nz = 512, nxp = 528;
    const size_t n = (size_t)nxp * ny * nz + (1 << 17);
    double *f[6];
    for (int k = 0; k < 6; k++) { (void)hipMalloc(&f[k], n * 8 + 4224 * 16); (void)hipMemset(f[k], 0, n * 8); f[k] += 528 * (k + 1); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long pxy = (long)nxp * ny;
    auto run = [&](const char *nm, auto fn) {
        for (int i = 0; i < 2; i++) fn();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 6; i++) fn();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 6;
        // 9 passes of 1.07 GB (u x 3, rhs read x 3, rhs write x 3)
        printf("%-64s %7.3f ms per launch  %6.2f us per tile-component  %7.1f GB/s\n", nm, ms, ms * 1e3 / (16384.0 * 3 / 256), 9.0 * nx * ny * nz * 8 / ms * 1e-6);
        fflush(stdout);
    };
    char nm[128];
    // one s_sleep(8) = 512 clocks = 0.21 us at 2.4 GHz: 0 / 12 / 24 sleeps = 0 / 2.5 / 5.1 us
    for (int dir = 0; dir < 2; dir++)
        for (int sl : {0, 12, 24, 36})
            for (int early = 0; early < 2; early++)
                for (int nt = 0; nt < 2; nt++) {
                    snprintf(nm, 128, "%s delay %4.1f us, rhs rows %s, rhs %s", dir ? "z" : "y", sl * 512 / 2400.0, early ? "BEFORE the delay" : "after the delay  ", nt ? "nontemporal" : "plain");
                    const long prow = dir ? pxy : nxp, pplane = dir ? nxp : pxy;
                    const int ntiles = nx / 16 * (dir ? ny : nz);
                    if (early && nt) run(nm, [&] { hipLaunchKernelGGL((k_skel<1, 1>), dim3(256), dim3(1024), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 16, ntiles, prow, pplane, sl); });
                    else if (early) run(nm, [&] { hipLaunchKernelGGL((k_skel<1, 0>), dim3(256), dim3(1024), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 16, ntiles, prow, pplane, sl); });
                    else if (nt) run(nm, [&] { hipLaunchKernelGGL((k_skel<0, 1>), dim3(256), dim3(1024), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 16, ntiles, prow, pplane, sl); });
                    else run(nm, [&] { hipLaunchKernelGGL((k_skel<0, 0>), dim3(256), dim3(1024), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 16, ntiles, prow, pplane, sl); });
                }
    // with the LDS trips and barriers of the real kernel
    const int lds = 16 * 516 * 8;
    (void)hipFuncSetAttribute((const void *)(k_skel<0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)(k_skel<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int dir = 0; dir < 2; dir++)
        for (int sl : {0, 12, 24, 30, 36})
            for (int early = 0; early < 2; early++) {
                snprintf(nm, 128, "%s + LDS tile trips, delay %4.1f us, rhs rows %s, nontemporal", dir ? "z" : "y", sl * 512 / 2400.0, early ? "BEFORE the delay" : "after the delay  ");
                const long prow = dir ? pxy : nxp, pplane = dir ? nxp : pxy;
                const int ntiles = nx / 16 * (dir ? ny : nz);
                if (early) run(nm, [&] { hipLaunchKernelGGL((k_skel<1, 1, 1>), dim3(256), dim3(1024), lds, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 16, ntiles, prow, pplane, sl); });
                else run(nm, [&] { hipLaunchKernelGGL((k_skel<0, 1, 1>), dim3(256), dim3(1024), lds, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 16, ntiles, prow, pplane, sl); });
            }
    // 32-pencil tiles (256-byte segments); the delay per tile-component doubled (twice the pencils per tile)
    for (int dir = 0; dir < 2; dir++)
        for (int sl : {0, 24, 48, 60})
            for (int nt = 0; nt < 2; nt++) {
                snprintf(nm, 128, "%s 32-pencil tiles, delay %4.1f us, rhs %s", dir ? "z" : "y", sl * 512 / 2400.0, nt ? "nontemporal" : "plain");
                const long prow = dir ? pxy : nxp, pplane = dir ? nxp : pxy;
                const int ntiles = nx / 32 * (dir ? ny : nz);
                if (nt) run(nm, [&] { hipLaunchKernelGGL((k_skel32<1>), dim3(256), dim3(1024), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 32, ntiles, prow, pplane, sl); });
                else run(nm, [&] { hipLaunchKernelGGL((k_skel32<0>), dim3(256), dim3(1024), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], nx / 32, ntiles, prow, pplane, sl); });
            }
    // x-slab order (round 4, the Infinity Cache question): for every column of 16 x-pencils the y pass and then the z pass
    // over the same [512 z][512 y][16 x] sub-volume of the six fields (201 MB; the cache holds 256 MB) -- does the z pass
    // find the y pass's lines?  64 launches of 512 tiles each instead of 2 of 16384.
    {
        auto slab = [&](int kx, int sl) {
            for (int xt = 0; xt < nx / 16; xt += 1) {
                const long xb = (long)xt * 16;
                for (int dir = 0; dir < 2; dir++) {
                    const long prow = dir ? pxy : nxp, pplane = dir ? nxp : pxy;
                    hipLaunchKernelGGL((k_skel<0, 0>), dim3(256), dim3(1024), 0, 0, f[0] + xb, f[1] + xb, f[2] + xb, f[3] + xb, f[4] + xb, f[5] + xb, 1, 512, prow, pplane, sl);
                }
            }
        };
        for (int sl : {0, 24}) {
            snprintf(nm, 128, "x-slab order: y then z per 16-pencil column, delay %4.1f us (BOTH directions: compare with the sum)", sl * 512 / 2400.0);
            for (int i = 0; i < 2; i++) slab(0, sl);
            (void)hipEventRecord(e0);
            for (int i = 0; i < 4; i++) slab(0, sl);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
            printf("%-100s %7.3f ms for y + z  %7.1f GB/s\n", nm, ms, 18.0 * nx * ny * nz * 8 / ms * 1e-6);
        }
    }
    return 0;
}