// 512-point complex FFT of one pencil per wave, on chip (kernel family K5f; used by csrc/fft512.hip, csrc/zfirst.hip
// and the z-transforming forms of the pair kernel in csrc/xscan.hip).  Unnormalised DFT, forward e^{-i}, like
// 2decomp&FFT / cuFFT behind the reference's fft_forward / fft_backward (src/backend/omp/poisson_fft.f90:89-97, 129-137).
#pragma once
#include "common.h"

#define FP 584  // LDS pitch per pencil in double2 (= 8 mod 16: the 8 modes of a row go to different banks)

__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// 8-point DFT, decimation in frequency, natural-order output.  S = -1 forward, +1 backward.
template <int S>
__device__ __forceinline__ void fft8(double2 (&a)[8])
{
    const double h = 0.70710678118654752440;
    double2 b[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        b[k] = cadd(a[k], a[k + 4]);
        b[k + 4] = csub(a[k], a[k + 4]);
    }
    // b[4+k] *= W8^k: W8^1 = (1 + S i)/sqrt2, W8^2 = S i, W8^3 = (-1 + S i)/sqrt2
    {
        double2 t = b[5];
        b[5] = make_double2(h * (t.x - S * t.y), h * (t.y + S * t.x));
        t = b[6];
        b[6] = make_double2(-S * t.y, S * t.x);
        t = b[7];
        b[7] = make_double2(h * (-t.x - S * t.y), h * (-t.y + S * t.x));
    }
#pragma unroll
    for (int o = 0; o < 8; o += 4) {
        const double2 c0 = cadd(b[o], b[o + 2]), c1 = cadd(b[o + 1], b[o + 3]), c2 = csub(b[o], b[o + 2]);
        const double2 d = csub(b[o + 1], b[o + 3]);
        const double2 c3 = make_double2(-S * d.y, S * d.x);  // * W4^1 = S i
        const int r = o ? 1 : 0;
        a[r] = cadd(c0, c1);
        a[r + 4] = csub(c0, c1);
        a[r + 2] = cadd(c2, c3);
        a[r + 6] = csub(c2, c3);
    }
}

// W512^e for the transform direction S; tw holds the first half, W^(e+256) = -W^e
template <int S>
__device__ __forceinline__ double2 twiddle(const double2 *__restrict__ tw, int e)
{
    double2 w = tw[e & 255];
    const double sg = (e & 256) ? -1.0 : 1.0;
    return make_double2(sg * w.x, (S > 0 ? -sg : sg) * w.y);
}

// one pencil per wave: in/out a[k] = point l + 64 k; pen = this wave's LDS region (FP double2)
template <int S>
__device__ __forceinline__ void fft512_wave(double2 (&a)[8], double2 *__restrict__ pen,
                                            const double2 *__restrict__ tw, int l)
{
    // pass A: over n1 (stride 64), twiddle W512^(l k1)
    wave_lds_fence();  // (the caller's reads of this region, e.g. pass C of a previous transform, come first)
    fft8<S>(a);
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) {
        a[k1] = cmul(a[k1], twiddle<S>(tw, l * k1));
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) pen[k1 * 72 + l] = a[k1];
    // no block barrier: the region belongs to this wave alone and a wave's LDS operations execute in order
    wave_lds_fence();
    // pass B: lane (k1 = l >> 3, b = l & 7) takes T1[k1][b + 8 a], twiddle W64^(b q1) = W512^(8 b q1)
    {
        const int k1 = l >> 3, b = l & 7;
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = pen[k1 * 72 + b + 8 * q];
        fft8<S>(a);
#pragma unroll
        for (int q1 = 1; q1 < 8; q1++) {
            a[q1] = cmul(a[q1], twiddle<S>(tw, 8 * b * q1));
        }
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) pen[(k1 * 8 + q1) * 9 + b] = a[q1];
    }
    // pass C: lane (k1 = l & 7, q1 = l >> 3) takes T2[k1][q1][b]; output X[k1 + 8 q1 + 64 q2] = X[l + 64 q2]
    wave_lds_fence();
    {
        const int k1 = l & 7, q1 = l >> 3;
#pragma unroll
        for (int b = 0; b < 8; b++) a[b] = pen[(k1 * 8 + q1) * 9 + b];
        fft8<S>(a);
    }
    wave_lds_fence();
}
