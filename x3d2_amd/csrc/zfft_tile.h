// z-first Poisson solve (000, 512^3 on one rank): the real-to-complex transform ALONG z done on the LDS tile of the
// kernel that produces the divergence's last z operator, the complex-to-real one on the tile of the kernel that
// consumes the pressure first (csrc/xscan.hip k_ytile_tds_pair<.., ZF>), and as stand-alone tile kernels
// (csrc/zfirst.hip).  The reference transforms x, then y, then z (2decomp&FFT / cuFFT behind
// src/backend/omp/poisson_fft.f90:89-97, 129-137; cuda :519, 568) -- the 3-D DFT is the same in any order, and with z
// first the y transforms, the spectral division (process_spectral_000, src/backend/omp/kernels/
// spectral_processing.f90:7-106) and the inverse y transforms are one pass: 6.5 passes over the spectrum instead of
// 10.5.
//
// Spectrum in this form: C[kz][y][x], kz = 0 .. 256 (the half axis is z instead of x), rows of `px` complex numbers.
// A tile = 16 x-adjacent z pencils of one y: two real pencils a, b go through ONE complex transform z = a + i b
// (a wave per pair, csrc/fft512_core.h):  A_k = (Z_k + conj Z_{512-k}) / 2,  B_k = (Z_k - conj Z_{512-k}) / (2 i).
// LDS `area`: 73728 B = the real tile [16][TP] (TP = 516), later the 8 waves' transform regions (576 real2_t each),
// later the transposed block T2[kz][16] (XOR-swizzled columns: conflict-free for the column writes of a wave and the
// row reads of the cooperative 256-byte stores).
#pragma once
#include "fft512_core.h"

#define ZF_AREA_DOUBLES 9216  // 8 x 576 real2_t
#define ZF_PEN 576

struct ZfArg {
    real2_t *c;         // C[257][ny][px]
    const real2_t *tw;  // W512^k, first half (fft512.hip)
    int ny;
    long px;
    int permn;  // > 0: row r < permn of the field sits at the 010 solver's interleaved position in C (enforce_periodicity_y's
                // order, src/backend/cuda/kernels/spectral_processing.f90:1062-1114): the channel's z-first solve
};

__device__ __forceinline__ int zf_t2(int m, int x) { return m * 16 + (x ^ (m & 15)); }

// area holds the real tile [16][TP] and every thread of the 16 waves has passed a barrier since it was written;
// crow = C + y * px + x0; on return the 257 x 16 modes are stored and the area is free again
#ifdef ZF_16WAVES
// Round 4 experiment, measured SLOWER (profiles/r04_zf16_waves.txt; off by default): EVERY wave transforms -- wave w takes the real pencil w as 256 complex numbers z_j = x_{2j} + i x_{2j+1}
// (fft256_wave) and splits:  E_k = (Z_k + conj Z_{256-k}) / 2,  O_k = (Z_k - conj Z_{256-k}) / (2 i),
// X_k = E_k + W512^k O_k (k = 0 .. 255),  X_256 = E_0 - O_0.  Four radix-4 stages exchange 24 KB per real pencil through
// LDS where the 512-point radix-8 form below exchanges 16 KB, and the transform phase is LDS-bandwidth bound.
template <int TP>
__device__ __forceinline__ void zf_forward(real_t *__restrict__ area, const real2_t *__restrict__ tws,
                                           real2_t *__restrict__ crow, long kzstride, int wave, int lane)
{
    real2_t a[4], X[4], X256 = make_real2(0.0, 0.0);
    real2_t *__restrict__ T2 = reinterpret_cast<real2_t *>(area);
    {
        const real2_t *__restrict__ pa = reinterpret_cast<const real2_t *>(area + wave * TP);
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = pa[lane + 64 * k];
    }
    __syncthreads();  // (the transform regions overlap other waves' pencils)
    {
        real2_t *__restrict__ pen = T2 + wave * FP256;
        fft256_wave<-1>(a, pen, tws, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) pen[lane + 64 * k] = a[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int idx = lane + 64 * k;
            const real2_t z = a[k], c = pen[(256 - idx) & 255];
            const real2_t E = make_real2(0.5 * (z.x + c.x), 0.5 * (z.y - c.y));
            const real2_t O = make_real2(0.5 * (z.y + c.y), -0.5 * (z.x - c.x));
            X[k] = cadd(E, cmul(twiddle<-1>(tws, idx), O));
            if (k == 0 && lane == 0) X256 = make_real2(E.x - O.x, 0.0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) T2[zf_t2(lane + 64 * k, wave)] = X[k];
    if (lane == 0) T2[zf_t2(256, wave)] = X256;
    __syncthreads();
    const int r = threadIdx.x >> 4, x = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int m = r + 64 * i;
        if (i < 4 || threadIdx.x < 16) crow[(long)m * kzstride + x] = T2[zf_t2(m, x)];
    }
    __syncthreads();
}
#else
template <int TP>
__device__ __forceinline__ void zf_forward(real_t *__restrict__ area, const real2_t *__restrict__ tws,
                                           real2_t *__restrict__ crow, long kzstride, int wave, int lane)
{
    real2_t a[8], A[5], B[5];
    real2_t *__restrict__ T2 = reinterpret_cast<real2_t *>(area);
    if (wave < 8) {
        const real_t *__restrict__ pa = area + (2 * wave) * TP, *__restrict__ pb = pa + TP;
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = make_real2(pa[lane + 64 * k], pb[lane + 64 * k]);
    }
    __syncthreads();  // (the transform regions overlap other waves' pencils)
    if (wave < 8) {
        real2_t *__restrict__ pen = T2 + wave * ZF_PEN;
#if !(ZF_EXP & 1)
        fft512_wave<-1>(a, pen, tws, lane);
#endif
#pragma unroll
        for (int k = 0; k < 8; k++) pen[lane + 64 * k] = a[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int idx = lane + 64 * k;
            if (k < 4 || lane == 0) {
                const real2_t x = a[k], y = pen[(512 - idx) & 511];
                A[k] = make_real2(0.5 * (x.x + y.x), 0.5 * (x.y - y.y));
                B[k] = make_real2(0.5 * (x.y + y.y), -0.5 * (x.x - y.x));
            }
        }
    }
    __syncthreads();
    if (wave < 8) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int idx = lane + 64 * k;
            if (k < 4 || lane == 0) {
                T2[zf_t2(idx, 2 * wave)] = A[k];
                T2[zf_t2(idx, 2 * wave + 1)] = B[k];
            }
        }
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, x = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int m = r + 64 * i;
        if (i < 4 || threadIdx.x < 16) crow[(long)m * kzstride + x] = T2[zf_t2(m, x)];
    }
    __syncthreads();
}

#endif

// the tile's 257 x 16 modes into registers (issued early: they are in flight while the previous tile is worked on)
struct ZfRows { real2_t v0, v1, v2, v3, v4; };
__device__ __forceinline__ ZfRows zf_inverse_load(const real2_t *__restrict__ crow, long kzstride)
{
    const int r = threadIdx.x >> 4, x = threadIdx.x & 15;
    const real2_t *__restrict__ p = crow + (long)r * kzstride + x;
    ZfRows v;
    v.v0 = p[0]; v.v1 = p[64 * kzstride]; v.v2 = p[128 * kzstride]; v.v3 = p[192 * kzstride];
    v.v4 = p[threadIdx.x < 16 ? 256 * kzstride : 0];  // (row 256: 16 threads; the others repeat their first load)
    return v;
}

// the area is free (a barrier since its last use); on return it holds the real tile [16][TP], barrier passed.
// Unnormalised inverse (e^{+i}); imaginary parts of the kz = 0 and kz = 256 planes belong to the pair's other pencil
// only through rounding noise of a Hermitian spectrum.
#ifdef ZF_16WAVES
// every wave: Z_k = E_k + i O_k with E_k = X_k + conj X_{256-k}, O_k = (X_k - conj X_{256-k}) W512^{-k} (the factor 2 of
// the 256-point transform folded in: the result is the unnormalised inverse, 512 x the pencil)
template <int TP>
__device__ __forceinline__ void zf_inverse(real_t *__restrict__ area, const real2_t *__restrict__ tws,
                                           const ZfRows &v, int wave, int lane)
{
    real2_t *__restrict__ T2 = reinterpret_cast<real2_t *>(area);
    const int r = threadIdx.x >> 4, x = threadIdx.x & 15;
    T2[zf_t2(r, x)] = v.v0;
    T2[zf_t2(r + 64, x)] = v.v1;
    T2[zf_t2(r + 128, x)] = v.v2;
    T2[zf_t2(r + 192, x)] = v.v3;
    if (threadIdx.x < 16) T2[zf_t2(256, x)] = v.v4;
    __syncthreads();
    real2_t a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int m = lane + 64 * k;
        const real2_t X = T2[zf_t2(m, wave)], C = T2[zf_t2(256 - m, wave)];
        const real2_t E = make_real2(X.x + C.x, X.y - C.y), D = make_real2(X.x - C.x, X.y + C.y);
        const real2_t O = cmul(D, twiddle<1>(tws, m));
        a[k] = make_real2(E.x - O.y, E.y + O.x);
    }
    __syncthreads();
    fft256_wave<1>(a, T2 + wave * FP256, tws, lane);
    __syncthreads();
    {
        real2_t *__restrict__ pa = reinterpret_cast<real2_t *>(area + wave * TP);
#pragma unroll
        for (int k = 0; k < 4; k++) pa[lane + 64 * k] = a[k];
    }
    __syncthreads();
}
#else
template <int TP>
__device__ __forceinline__ void zf_inverse(real_t *__restrict__ area, const real2_t *__restrict__ tws,
                                           const ZfRows &v, int wave, int lane)
{
    real2_t *__restrict__ T2 = reinterpret_cast<real2_t *>(area);
    const int r = threadIdx.x >> 4, x = threadIdx.x & 15;
    T2[zf_t2(r, x)] = v.v0;
    T2[zf_t2(r + 64, x)] = v.v1;
    T2[zf_t2(r + 128, x)] = v.v2;
    T2[zf_t2(r + 192, x)] = v.v3;
    if (threadIdx.x < 16) T2[zf_t2(256, x)] = v.v4;
    __syncthreads();
    real2_t a[8];
    if (wave < 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int m = lane + 64 * k, mm = m <= 256 ? m : 512 - m;
            real2_t A = T2[zf_t2(mm, 2 * wave)], B = T2[zf_t2(mm, 2 * wave + 1)];
            if (m > 256) { A.y = -A.y; B.y = -B.y; }
            a[k] = make_real2(A.x - B.y, A.y + B.x);
        }
    }
    __syncthreads();
#if !(ZF_EXP & 1)
    if (wave < 8) fft512_wave<1>(a, T2 + wave * ZF_PEN, tws, lane);
#endif
    __syncthreads();
    if (wave < 8) {
        real_t *__restrict__ pa = area + (2 * wave) * TP, *__restrict__ pb = pa + TP;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            pa[lane + 64 * k] = a[k].x;
            pb[lane + 64 * k] = a[k].y;
        }
    }
    __syncthreads();
}
#endif
