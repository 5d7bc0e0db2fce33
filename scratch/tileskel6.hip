// Round 5, second question: does the skeleton gain from TWO (or four) de-synchronised workgroups per CU?  (tileskel5.hip + NP, grid)
// Round 5: access-pattern questions about k_ytile_transeq3's memory skeleton (scratch, not part of the product).
// Base = scratch/tileskel.hip's k_skel<EARLY = 0, NT = 1, LDSX = 1>: per tile and component the workgroup takes the
// prefetched 16 x 512 tile of u_c through the LDS tile, requests the next field's rows, sleeps DELAY (the solves),
// requests the old rhs rows, adds, stores.  Questions ("next" item 1):
//   MAP   0 tile = block + k * grid (the product's order)
//         1 XCD-aware: the 32 workgroups of XCD k (block % 8 == k) own the x tiles [4k, 4k + 4) of 8 consecutive rows
//         2 XCD-aware: XCD k owns whole rows (other-direction index 8 it + k), its 32 workgroups the 32 x tiles of it
//         3 one contiguous run of tiles per workgroup
//         4 "other" index fastest: concurrently running workgroups share the x tile and differ in y (z tiles) / z (y tiles)
//   PPAD  extra rows in the plane pitch (z stride = nxp * (ny + PPAD) * 8 bytes): 528 * 512 * 8 is 66 x 32 KiB exactly
//   SPLIT the rhs read-modify-write replaced by "read NR other arrays, write a third" (no address is read and written):
//         y with NR = 0 (3R + 3W) and z with NR = 2 (9R + 3W) = the "rhs_y written, z sums" count of round 4, measured
//   hipcc -O2 --offload-arch=gfx950 scratch/tileskel5.hip -o scratch/tileskel5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldg_nt(const double *p)
{
    d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void stg_nt(double *p, double2 v)
{
    d2v w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p));
}
__device__ __forceinline__ void delay_us(int sleeps)
{
    for (int k = 0; k < sleeps; k++) __builtin_amdgcn_s_sleep(8);  // 8 x 64 clocks
}

struct Geo { int ntx, nother, ntiles; long prow, pplane; };

template <int MAP>
__device__ __forceinline__ int tile_of(int it, const Geo &g)
{
    int b = blockIdx.x; const int G = gridDim.x;
    if (MAP == 5) { const int g16 = b / 16, r16 = b % 16; b = g16 * 16 + 2 * (r16 % 8) + r16 / 8; return b + it * G; }  // pairs (2j, 2j+1) on blocks b, b+8
    if (MAP == 0) return b + it * G;
    if (MAP == 1) { const int x = b & 7, s = b >> 3; return (8 * it + (s >> 2)) * g.ntx + 4 * x + (s & 3); }
    if (MAP == 2) { const int x = b & 7, s = b >> 3; return (8 * it + x) * g.ntx + s; }
    if (MAP == 3) return b * (g.ntiles / G) + it;
    // MAP 4: other index fastest
    { const int id = b + it * G; return (id % g.nother) * g.ntx + id / g.nother; }
}

// NR: 1 = read-modify-write of r_c (the product), 0 = write only, 2 = read two other arrays (q_c, s_c) and write r_c
template <int MAP, int NR, int NP, int WPE>
__global__ void __launch_bounds__(64 * NP, WPE) k_skel(const double *u0, const double *u1, const double *u2, double *r0, double *r1,
                                               double *r2, const double *q0, const double *q1, const double *q2,
                                               const double *s0, const double *s1, const double *s2, Geo g, int sleeps)
{
    constexpr int NL = 4, TP = 516;
    extern __shared__ double tile[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cc = threadIdx.x % (NP / 2), cy = threadIdx.x / (NP / 2);
    auto off_of = [&](int tl) { return (long)(tl / g.ntx) * g.pplane + (long)(tl % g.ntx) * NP; };
    auto load = [&](double2 (&v)[NL], const double *f, long off, bool nt) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const double *p = f + off + (long)(cy + 128 * i) * g.prow + 2 * cc;
            v[i] = nt ? ldg_nt(p) : *reinterpret_cast<const double2 *>(p);
        }
    };
    const int nit = g.ntiles / gridDim.x;
    double2 nxt[NL];
    load(nxt, u0, off_of(tile_of<MAP>(0, g)), false);
    for (int it = 0; it < nit; it++) {
        const long off = off_of(tile_of<MAP>(it, g));
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            double2 v[NL], old[NL], old2[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) v[i] = nxt[i];
            double b[8];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                tile[(2 * cc) * TP + cy + 128 * i] = v[i].x;
                tile[(2 * cc + 1) * TP + cy + 128 * i] = v[i].y;
            }
            __syncthreads();
            const double2 *src = reinterpret_cast<const double2 *>(tile + wave * TP + lane * 8);
#pragma unroll
            for (int m = 0; m < 4; m++) { const double2 t = src[m]; b[2 * m] = t.x; b[2 * m + 1] = t.y; }
            const double *nsrc = c == 0 ? u1 : (c == 1 ? u2 : u0);
            if (c < 2 || it + 1 < nit) load(nxt, nsrc, c < 2 ? off : off_of(tile_of<MAP>(it + 1, g)), false);
            double *o = c == 0 ? r0 : (c == 1 ? r1 : r2);
            delay_us(sleeps);
            if (NR == 1) load(old, o, off, true);
            if (NR == 2) {
                load(old, c == 0 ? q0 : (c == 1 ? q1 : q2), off, true);
                load(old2, c == 0 ? s0 : (c == 1 ? s1 : s2), off, true);
            }
            double2 *dst = reinterpret_cast<double2 *>(tile + wave * TP + lane * 8);
#pragma unroll
            for (int m = 0; m < 4; m++) dst[m] = make_double2(b[2 * m] * 0.5, b[2 * m + 1] * 0.5);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NL; i++) v[i] = make_double2(tile[(2 * cc) * TP + cy + 128 * i], tile[(2 * cc + 1) * TP + cy + 128 * i]);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                double2 w = v[i];
                if (NR >= 1) { w.x += old[i].x; w.y += old[i].y; }
                if (NR == 2) { w.x += old2[i].x; w.y += old2[i].y; }
                stg_nt(o + off + (long)(cy + 128 * i) * g.prow + 2 * cc, w);
            }
            __syncthreads();
        }
    }
}

int main(int argc, char **argv)
{
    const int nx = 512, ny = 512, nz = 512, nxp = 528, maxpad = 8;
    const size_t n = (size_t)nxp * (ny + maxpad) * nz + (1 << 17);
    double *f[12];
    for (int k = 0; k < 12; k++) {
        if (hipMalloc(&f[k], n * 8 + 4224 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMemset(f[k], 0, n * 8);
        f[k] += 528 * (k % 6 + 1);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char *nm, double passes, auto fn) {
        for (int i = 0; i < 2; i++) fn();
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 4; i++) fn();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-86s %7.3f ms (best %7.3f)  %7.1f GB/s\n", nm, sum / 3, best, passes * nx * ny * nz * 8 / (sum / 3) * 1e-6);
        fflush(stdout);
    };
#define LAUNCH(MAP, NP, WPE, GRID) do { \
        const int lds = NP * 516 * 8; \
        (void)hipFuncSetAttribute((const void *)(k_skel<MAP, 1, NP, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_skel<MAP, 1, NP, WPE>, 64 * NP, lds); \
        snprintf(nm, 200, "%s NP %2d grid %4d (occupancy %d blocks/CU) map %d delay %3.1f us", dir ? "z" : "y", NP, GRID, occ, MAP, sl * 512 / 2400.0); \
        Geo g{nx / NP, dir ? ny : nz, nx / NP * (dir ? ny : nz), dir ? pxy : nxp, dir ? nxp : pxy}; \
        run(nm, 9.0, [&] { hipLaunchKernelGGL((k_skel<MAP, 1, NP, WPE>), dim3(GRID), dim3(64 * NP), lds, 0, f[0], f[1], f[2], f[3], f[4], f[5], \
            f[6], f[7], f[8], f[9], f[10], f[11], g, sl); }); } while (0)
    char nm[200];
    for (int rep = 0; rep < 2; rep++)
    for (int dir = 0; dir < 2; dir++) {
        const long pxy = (long)nxp * ny;
        for (int sl : {0, 24}) {
            LAUNCH(0, 16, 1, 256);
            LAUNCH(0, 16, 8, 512);
            LAUNCH(0, 8, 4, 512);
            LAUNCH(5, 8, 4, 512);
            LAUNCH(5, 8, 8, 1024);
            LAUNCH(5, 8, 4, 256);
        }
    }
    return 0;
}
