// 512-point complex FFT of one pencil per wave, on chip (kernel family K5f; used by csrc/fft512.hip, csrc/zfirst.hip
// and the z-transforming forms of the pair kernel in csrc/xscan.hip).  Unnormalised DFT, forward e^{-i}, like
// 2decomp&FFT / cuFFT behind the reference's fft_forward / fft_backward (src/backend/omp/poisson_fft.f90:89-97, 129-137).
#pragma once
#include "common.h"

#define FP 584  // LDS pitch per pencil in real2_t (= 8 mod 16: the 8 modes of a row go to different banks)

__device__ __forceinline__ real2_t cmul(real2_t a, real2_t b)
{
    return make_real2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ real2_t cadd(real2_t a, real2_t b) { return make_real2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ real2_t csub(real2_t a, real2_t b) { return make_real2(a.x - b.x, a.y - b.y); }

// 8-point DFT, decimation in frequency, natural-order output.  S = -1 forward, +1 backward.
template <int S>
__device__ __forceinline__ void fft8(real2_t (&a)[8])
{
    const real_t h = 0.70710678118654752440;
    real2_t b[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        b[k] = cadd(a[k], a[k + 4]);
        b[k + 4] = csub(a[k], a[k + 4]);
    }
    // b[4+k] *= W8^k: W8^1 = (1 + S i)/sqrt2, W8^2 = S i, W8^3 = (-1 + S i)/sqrt2
    {
        real2_t t = b[5];
        b[5] = make_real2(h * (t.x - S * t.y), h * (t.y + S * t.x));
        t = b[6];
        b[6] = make_real2(-S * t.y, S * t.x);
        t = b[7];
        b[7] = make_real2(h * (-t.x - S * t.y), h * (-t.y + S * t.x));
    }
#pragma unroll
    for (int o = 0; o < 8; o += 4) {
        const real2_t c0 = cadd(b[o], b[o + 2]), c1 = cadd(b[o + 1], b[o + 3]), c2 = csub(b[o], b[o + 2]);
        const real2_t d = csub(b[o + 1], b[o + 3]);
        const real2_t c3 = make_real2(-S * d.y, S * d.x);  // * W4^1 = S i
        const int r = o ? 1 : 0;
        a[r] = cadd(c0, c1);
        a[r + 4] = csub(c0, c1);
        a[r + 2] = cadd(c2, c3);
        a[r + 6] = csub(c2, c3);
    }
}

// W512^e for the transform direction S; tw holds the first half, W^(e+256) = -W^e
template <int S>
__device__ __forceinline__ real2_t twiddle(const real2_t *__restrict__ tw, int e)
{
    real2_t w = tw[e & 255];
    const real_t sg = (e & 256) ? -1.0 : 1.0;
    return make_real2(sg * w.x, (S > 0 ? -sg : sg) * w.y);
}

// one pencil per wave: in/out a[k] = point l + 64 k; pen = this wave's LDS region (FP real2_t)
template <int S>
__device__ __forceinline__ void fft512_wave(real2_t (&a)[8], real2_t *__restrict__ pen,
                                            const real2_t *__restrict__ tw, int l)
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


// ---- 256-point complex FFT of one pencil per wave, FOUR points per lane (round 4 experiment, -DZF_16WAVES, measured
// slower and off by default -- profiles/r04_zf16_waves.txt: the z transforms on the tile of the
// z operator pairs give every one of the 16 waves ONE real pencil -- 512 reals as 256 complex numbers -- instead of two
// real pencils to 8 of them).  in / out a[k] = point l + 64 k; pen = this wave's LDS region (FP256 real2_t).
// 256 = 4 x 4 x 4 x 4, decimation in frequency: f = f1 + 4 f2 + 16 f3 + 64 f4; every stage is a 4-point DFT in
// registers, the three exchanges go through the wave's own region (no block barrier: a wave's LDS operations execute in
// order).  Unnormalised, S = -1 forward / +1 backward like fft512_wave.
#define FP256 288

template <int S>
__device__ __forceinline__ void fft4(real2_t (&a)[4])
{
    const real2_t b0 = cadd(a[0], a[2]), b1 = csub(a[0], a[2]), b2 = cadd(a[1], a[3]), d = csub(a[1], a[3]);
    const real2_t b3 = make_real2(-S * d.y, S * d.x);  // * W4^1 = S i
    a[0] = cadd(b0, b2);
    a[2] = csub(b0, b2);
    a[1] = cadd(b1, b3);
    a[3] = csub(b1, b3);
}

template <int S>
__device__ __forceinline__ void fft256_wave(real2_t (&a)[4], real2_t *__restrict__ pen, const real2_t *__restrict__ tw,
                                            int l)
{
    wave_lds_fence();  // (the caller's accesses to this region come first)
    // stage 1: over the points l + 64 n_a; twiddle W256^(l f1) = W512^(2 l f1)
    fft4<S>(a);
#pragma unroll
    for (int f = 1; f < 4; f++) a[f] = cmul(a[f], twiddle<S>(tw, 2 * l * f));
#pragma unroll
    for (int f = 0; f < 4; f++) pen[f * 68 + l] = a[f];
    wave_lds_fence();
    {   // stage 2: lane (f1 = l >> 4, l0 = l & 15) takes the points l0 + 16 j; twiddle W64^(l0 f2) = W512^(8 l0 f2)
        const int f1 = l >> 4, l0 = l & 15;
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = pen[f1 * 68 + l0 + 16 * j];
        fft4<S>(a);
#pragma unroll
        for (int f = 1; f < 4; f++) a[f] = cmul(a[f], twiddle<S>(tw, 8 * l0 * f));
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 4; f++) pen[(f1 * 4 + f) * 17 + l0] = a[f];
    }
    wave_lds_fence();
    {   // stage 3: lane (f1 = l >> 4, f2 = (l >> 2) & 3, m0 = l & 3) takes the points m0 + 4 m; twiddle W16^(m0 f3) = W512^(32 m0 f3)
        const int r = l >> 2, m0 = l & 3;
#pragma unroll
        for (int m = 0; m < 4; m++) a[m] = pen[r * 17 + m0 + 4 * m];
        fft4<S>(a);
#pragma unroll
        for (int f = 1; f < 4; f++) a[f] = cmul(a[f], twiddle<S>(tw, 32 * m0 * f));
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 4; f++) pen[(r * 4 + f) * 4 + m0 + (r >> 2) * 4] = a[f];  // (+ 4 per f1: the rows of different f1 leave the same banks)
    }
    wave_lds_fence();
    {   // stage 4: lane l = f1 + 4 f2 + 16 f3 takes the four m0 of its (f1, f2, f3); output a[f4] = X[l + 64 f4]
        const int f1 = l & 3, f2 = (l >> 2) & 3, f3 = l >> 4;
        const int row = ((f1 * 4 + f2) * 4 + f3) * 4 + f1 * 4;
#pragma unroll
        for (int m = 0; m < 4; m++) a[m] = pen[row + m];
        fft4<S>(a);
    }
    wave_lds_fence();
}
