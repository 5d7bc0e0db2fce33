// 512-point complex FFTs along a strided axis of the spectral array c[nz][ny][nxs]
// (kernel family K5f), replacing rocFFT's strided column passes in the single-rank Poisson solve:
//   src/backend/omp/poisson_fft.f90:89-97, 129-137   fft_forward / fft_backward (unnormalised DFT,
//                                                    forward e^{-i}; 2decomp&FFT / cuFFT there)
//   src/backend/omp/kernels/spectral_processing.f90:7-106 process_spectral_000 (fused into the z pass)
//
// A workgroup (8 waves) owns 8 x-adjacent modes (128 contiguous bytes per row) x 512 points along
// the axis.  Rows are loaded cooperatively (8 rows x 128 B per wave instruction) into an LDS tile
// [mode][point]; wave w then transforms pencil w entirely on chip: lane l holds points l + 64 k,
// three radix-8 passes (512 = 8 x 8 x 8) with two in-LDS exchanges, natural-order output in the same
// register layout.  For the z axis the forward transform, the spectral division and the backward
// transform are done in ONE kernel: the spectrum is read once and written once (2 passes instead
// of 6 + the 1.5 of a separate process_spectral_000).
#include "common.h"

#include "fft512_core.h"

struct Spec000 {
    const real_t *waves, *ax, *bx, *ay, *by, *az, *bz;
    int nx, ny, nz;
    const real_t *rwT;  // [ny][nxs][nz]: -1 / waves (0 where waves < 1e-16), z fastest; null: use waves
};

// MODE 0: forward, MODE 1: backward, MODE 2: forward + process_spectral_000 + backward (z axis only)
// ZH (MODE 2 with sp.rwT): the z-first spectrum C[kz][y][x] (csrc/zfft_tile.h) -- the transformed axis is y, the rows'
// other index is kz (the half axis: never mirrored), the mode index is the FULL x axis (mirrored above nx / 2 like y and
// z are in the reference's kernel), sp.rwT = [kz][x][y]
template <int MODE, int NP, bool ZH = false>
__global__ void __launch_bounds__(64 * NP, 16 / NP)
    k_fft512(real2_t *c, const real2_t *__restrict__ twg, long stride_axis, long stride_other, int nxs, Spec000 sp,
             real2_t *xbuf, int ys, int ysc)
{
    extern __shared__ real2_t tile[];  // [NP][FP] + 256 twiddles
    real2_t *__restrict__ tws = tile + NP * FP;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * NP, m = tid & (NP - 1), r = tid / NP;
    const long base = (long)blockIdx.y * stride_other + i0;
    const bool valid = i0 + m < nxs;
    // slab-exchange addressing (multi-rank solver, y axis only): point t of the axis belongs to peer t / ys and,
    // inside that peer's share, to part (t % ys) / ysc: xbuf = [peer][part][other][ysc][nxs], the layout the z
    // all-to-all sends and receives part by part (sfft.hip; ysc = ys: one part).
    // MODE 0 stores there (forward y pass = pack), MODE 1 loads from there (backward y pass = unpack).
    auto xaddr = [&](int t) {
        const int ch = t / ys, tt = t - ch * ys, pt = tt / ysc, t3 = tt - pt * ysc;
        return ((((long)ch * (ys / ysc) + pt) * gridDim.y + blockIdx.y) * ysc + t3) * nxs + i0 + m;
    };
    // MODE 2: the wave numbers of this thread's 8 spectral points, requested before anything else (their
    // latency used to be exposed between the two transforms)
    real_t rw8[8];
    if (MODE == 2 && valid && !sp.rwT) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const real_t wv = sp.waves[((size_t)(it * 64 + r) * sp.ny + blockIdx.y) * nxs + i0 + m];
            rw8[it] = wv < 1.e-16 ? 0.0 : -1.0 / wv;
        }
    }
    // ---- cooperative load: NP * 16 contiguous bytes per row
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int z = it * 64 + r;
        const real2_t *__restrict__ src = (MODE == 1 && xbuf) ? xbuf + xaddr(z) : c + base + (long)z * stride_axis + m;
        tile[m * FP + z] = valid ? *src : make_real2(0.0, 0.0);
    }
    __syncthreads();
    real2_t *__restrict__ pen = tile + w * FP;
    real2_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = pen[l + 64 * k];
    // (no barrier: from here to the cooperative store a wave only touches its own pencil's region of the tile)
    if (MODE == 1) fft512_wave<1>(a, pen, tws, l);
    else fft512_wave<-1>(a, pen, tws, l);
    if (MODE == 2 && sp.rwT) {
        // spectral division in the transform's own register layout (lane l holds points l + 64 k of mode
        // i0 + w): the reciprocal wave numbers come from a z-fastest copy, no trip through the tile, no barrier
        // between the two transforms (src/backend/omp/kernels/spectral_processing.f90:36-99, same operation order)
        const int i = i0 + w, j = blockIdx.y;
        if (i < nxs) {
            // fixed over the pencil: the x mode and the rows' other index (y; ZH: kz); per point: the axis (z; ZH: y)
            const real_t a_o = ZH ? sp.az[j] : sp.ay[j], b_o = ZH ? sp.bz[j] : sp.by[j];
            const bool f_o = ZH ? false : (j + 1) > sp.ny / 2 + 1;
            const real_t axi = sp.ax[i], bxi = sp.bx[i];
            const bool fx = ZH && (i + 1) > sp.nx / 2 + 1;
            const real_t *__restrict__ a_ax = ZH ? sp.ay : sp.az, *__restrict__ b_ax = ZH ? sp.by : sp.bz;
            const int n_ax = ZH ? sp.ny : sp.nz;
            const real_t rn = 1.0 / sp.nx / sp.ny / sp.nz;
            const real_t *__restrict__ rwp = sp.rwT + ((size_t)j * nxs + i) * 512;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = l + 64 * kk;
                real_t div_r = a[kk].x * rn, div_c = a[kk].y * rn;
                const real_t a_k = a_ax[k], b_k = b_ax[k];
                const bool f_k = (k + 1) > n_ax / 2 + 1;
                const real_t azk = ZH ? a_o : a_k, bzk = ZH ? b_o : b_k, ayj = ZH ? a_k : a_o, byj = ZH ? b_k : b_o;
                const bool fz = ZH ? f_o : f_k, fy = ZH ? f_k : f_o;
                real_t tr, tc;
                tr = div_r; tc = div_c;
                div_r = tr * bzk + tc * azk; div_c = tc * bzk - tr * azk;
                if (fz) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
                if (fy) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * bxi + tc * axi; div_c = tc * bxi - tr * axi;
                if (fx) { div_r = -div_r; div_c = -div_c; }
                const real_t rw = rwp[k];
                div_r = div_r * rw; div_c = div_c * rw;
                tr = div_r; tc = div_c;
                div_r = tr * bzk - tc * azk; div_c = -tc * bzk - tr * azk;
                if (fz) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
                if (fy) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * bxi + tc * axi; div_c = -tc * bxi + tr * axi;
                if (fx) { div_r = -div_r; div_c = -div_c; }
                a[kk] = make_real2(div_r, div_c);
            }
        }
        fft512_wave<1>(a, pen, tws, l);
    } else if (MODE == 2) {
        // natural order back to the tile, spectral division in the row-cooperative layout (waves is
        // read as 64 contiguous bytes per row), then the backward transform
#pragma unroll
        for (int k = 0; k < 8; k++) pen[l + 64 * k] = a[k];
        __syncthreads();
        const int j = blockIdx.y, i = i0 + m;
        if (valid) {
            const real_t ayj = sp.ay[j], byj = sp.by[j], axi = sp.ax[i], bxi = sp.bx[i];
            // the reference divides by nx, ny, nz and by waves per element (8 FP64 divisions); here one
            // reciprocal of the product and one of waves: <= 2 ulp apart, far inside the parity tolerance
            const real_t rn = 1.0 / sp.nx / sp.ny / sp.nz;
            const bool fy = (j + 1) > sp.ny / 2 + 1;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int k = it * 64 + r;
                real2_t v = tile[m * FP + k];
                // src/backend/omp/kernels/spectral_processing.f90:36-99, same order as k_process_spectral_000
                real_t div_r = v.x * rn, div_c = v.y * rn;
                const real_t azk = sp.az[k], bzk = sp.bz[k];
                const bool fz = (k + 1) > sp.nz / 2 + 1;
                real_t tr, tc;
                tr = div_r; tc = div_c;
                div_r = tr * bzk + tc * azk; div_c = tc * bzk - tr * azk;
                if (fz) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
                if (fy) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * bxi + tc * axi; div_c = tc * bxi - tr * axi;
                const real_t rw = rw8[it];
                div_r = div_r * rw; div_c = div_c * rw;
                tr = div_r; tc = div_c;
                div_r = tr * bzk - tc * azk; div_c = -tc * bzk - tr * azk;
                if (fz) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
                if (fy) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * bxi + tc * axi; div_c = -tc * bxi + tr * axi;
                tile[m * FP + k] = make_real2(div_r, div_c);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = pen[l + 64 * k];
        __syncthreads();
        fft512_wave<1>(a, pen, tws, l);
    }
    // ---- natural order back to the tile, cooperative store
#pragma unroll
    for (int k = 0; k < 8; k++) pen[l + 64 * k] = a[k];
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int z = it * 64 + r;
            real2_t *__restrict__ dstp = (MODE == 0 && xbuf) ? xbuf + xaddr(z) : c + base + (long)z * stride_axis + m;
            *dstp = tile[m * FP + z];
        }
    }
}

// ---------------------------------------------------------------- z stage of the slab solver on N ranks
// After the all-to-all a rank holds, for its share of the (x, y) modes, all 512 N points along z as N chunks of
// 512 (chunk p = what peer p sent): R[g = 512 p + zl][W modes].  The transform of length M = 512 N is split as
//   Y_k1[zl] = sum_p x[zl + 512 p] W_N^(p k1)        N-point DFTs ACROSS the chunks (same zl, same mode)
//   Y'_k1[zl] = Y_k1[zl] W_M^(zl k1)                 twiddle
//   X[k1 + N k2] = sum_zl Y'_k1[zl] W_512^(zl k2)    N transforms of 512 points, one per wave (fft512_wave)
// so that one workgroup of 16 waves = N chunks x 16/N modes does the forward transform, process_spectral_000
// (src/backend/omp/kernels/spectral_processing.f90:7-106, same operation order as k_process_spectral_000_slab) and
// the inverse entirely on chip: the received array is read once and written once (2 passes) instead of transpose,
// rocFFT forward, division, rocFFT backward, transpose (10 passes; the reference issues cuFFTMp's z transforms
// and a separate spectral kernel, src/backend/cuda/poisson_fft.f90:519-568).
template <int S, int N>
__device__ __forceinline__ void dft_small(real2_t (&v)[8])
{
    if constexpr (N == 8) fft8<S>(v);
    else if constexpr (N == 4) {
        const real2_t c0 = cadd(v[0], v[2]), c1 = cadd(v[1], v[3]), c2 = csub(v[0], v[2]), d = csub(v[1], v[3]);
        const real2_t c3 = make_real2(-S * d.y, S * d.x);  // d * (S i)
        v[0] = cadd(c0, c1); v[2] = csub(c0, c1); v[1] = cadd(c2, c3); v[3] = csub(c2, c3);
    } else if constexpr (N == 2) {
        const real2_t t0 = cadd(v[0], v[1]), t1 = csub(v[0], v[1]);
        v[0] = t0; v[1] = t1;
    }
}

struct SpecSlab {
    const real_t *waves;  // this part's -1 / waves [W][nz], z fastest (0 where waves < 1e-16)
    const real_t *ax, *bx, *ay, *by, *az, *bz;
    int nx, ny, nz, nxs, yoff;
    int xoff;  // YL only
};

// WV waves = N chunks x WV / N modes per workgroup.  N = 1 (the DFTs across the chunks done by k_radix_peers, or a
// single rank): blockIdx.y = the chunk, whose points are the z modes chunk + nk * k2.
// YL (y slabs, z-first spectrum, csrc/sfftz.hip): the long axis is y (sp.ny points, sp.waves = [W][ny]), mode wl =
// (kz - yoff) * nxs + (kx - xoff) with kz on the half axis (never mirrored) and kx on the full x axis (mirrored above
// nx / 2 like y and z in the reference's kernel)
// PART 0: everything; 1: the forward transform only; 2: the inverse only; 3: the division only -- the three hooks of the
// reference's interface one by one; a transformed row (k1, k2) stands for the mode k1 + N k2 in all of them
template <int N, int WV, bool YL = false, int PART = 0>
__global__ void __launch_bounds__(64 * WV, 16 / WV)
    k_fft512_peers(real2_t *R, const real2_t *__restrict__ twg, long W, SpecSlab sp, int nk)
{
    constexpr int NM = WV / N;  // modes per workgroup
    extern __shared__ real2_t tile[];  // [WV pencils][FP] + 256 twiddles
    real2_t *__restrict__ tws = tile + WV * FP;
    R += (long)blockIdx.y * 512 * W;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // NM < 8: a workgroup's row segments are 64 / 32 bytes, 2 / 4 workgroups share every 128-byte line.  Workgroups
    // are dealt round-robin to the 8 XCDs (each with its own L2): number the mode groups so that the workgroups
    // that share lines are bid, bid + 8, ... -- same XCD, launched back to back -- instead of 4 different L2s
    long grp = blockIdx.x;
    if constexpr (NM < 8) {
        constexpr int SH = 8 / NM;
        const long bid = blockIdx.x, blk = bid / (8 * SH), in = bid % (8 * SH);
        grp = blk * (8 * SH) + (in % 8) * SH + in / 8;
    }
    const long w0 = grp * NM;
    const int m = tid % NM, r = tid / NM;  // r: row (point along z) of a pass of 64 N rows
    const bool valid = w0 + m < W;
    // ---- cooperative load: row rr = 512 p + zl of the received array, NM adjacent modes per row
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int rr = it * (64 * N) + r, p = rr >> 9, zl = rr & 511;
        tile[(p * NM + m) * FP + zl] = valid ? R[(long)rr * W + w0 + m] : make_real2(0.0, 0.0);
    }
    __syncthreads();
    // ---- N-point DFTs across the chunks + twiddle W_M^(zl k1); thread -> (zl, mode)
    if constexpr (N > 1 && (PART == 0 || PART == 1)) {
#pragma unroll
        for (int s_ = 0; s_ < 8 / N; s_++) {
            const int idx = tid + 64 * WV * s_, zl = idx & 511, mm = idx >> 9;
            real2_t v[8];
#pragma unroll
            for (int p = 0; p < N; p++) v[p] = tile[(p * NM + mm) * FP + zl];
            dft_small<-1, N>(v);
#pragma unroll
            for (int k1 = 1; k1 < N; k1++) {
                x3d_f64 sn, cs;  // (twiddles in FP64 whatever the real kind: one per point, outside the butterflies)
                sincospi(-2.0 * (x3d_f64)(zl * k1) / (512.0 * N), &sn, &cs);
                v[k1] = cmul(v[k1], make_real2((real_t)cs, (real_t)sn));
            }
#pragma unroll
            for (int k1 = 0; k1 < N; k1++) tile[(k1 * NM + mm) * FP + zl] = v[k1];
        }
        __syncthreads();
    }
    // ---- wave w: pencil (k1 = w / NM, mode w % NM): 512-point transform, spectral division, inverse
    real2_t *__restrict__ pen = tile + w * FP;
    real2_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = pen[l + 64 * k];
    if constexpr (PART == 0 || PART == 1) fft512_wave<-1>(a, pen, tws, l);
    if constexpr (PART == 0 || PART == 3) {
        const int k1 = w / NM;
        const long wl = w0 + w % NM;
        if (wl < W) {
            const int i = (int)(wl % sp.nxs) + (YL ? sp.xoff : 0), j = (int)(wl / sp.nxs) + sp.yoff;
            // fixed over the pencil: the x mode and the second mode index (y; YL: kz); per point: the long axis (z; YL: y)
            const real_t a_o = YL ? sp.az[j] : sp.ay[j], b_o = YL ? sp.bz[j] : sp.by[j];
            const bool f_o = YL ? false : (j + 1) > sp.ny / 2 + 1;
            const real_t axi = sp.ax[i], bxi = sp.bx[i];
            const bool fx = YL && (i + 1) > sp.nx / 2 + 1;
            const real_t *__restrict__ a_ax = YL ? sp.ay : sp.az, *__restrict__ b_ax = YL ? sp.by : sp.bz;
            const int n_ax = YL ? sp.ny : sp.nz;
            const real_t *__restrict__ wv = sp.waves + wl * n_ax;
            const real_t rn = 1.0 / sp.nx / sp.ny / sp.nz;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = (N == 1 ? (int)blockIdx.y : k1) + nk * (l + 64 * kk);  // this point's mode on the long axis
                real_t div_r = a[kk].x * rn, div_c = a[kk].y * rn;
                const real_t a_k = a_ax[k], b_k = b_ax[k];
                const bool f_k = (k + 1) > n_ax / 2 + 1;
                const real_t azk = YL ? a_o : a_k, bzk = YL ? b_o : b_k, ayj = YL ? a_k : a_o, byj = YL ? b_k : b_o;
                const bool fz = YL ? f_o : f_k, fy = YL ? f_k : f_o;
                real_t tr, tc;
                tr = div_r; tc = div_c;
                div_r = tr * bzk + tc * azk; div_c = tc * bzk - tr * azk;
                if (fz) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
                if (fy) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * bxi + tc * axi; div_c = tc * bxi - tr * axi;
                if (fx) { div_r = -div_r; div_c = -div_c; }
                const real_t rw = wv[k];  // (-1 / waves)
                div_r = div_r * rw; div_c = div_c * rw;
                tr = div_r; tc = div_c;
                div_r = tr * bzk - tc * azk; div_c = -tc * bzk - tr * azk;
                if (fz) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * byj + tc * ayj; div_c = tc * byj - tr * ayj;
                if (fy) { div_r = -div_r; div_c = -div_c; }
                tr = div_r; tc = div_c;
                div_r = tr * bxi + tc * axi; div_c = -tc * bxi + tr * axi;
                if (fx) { div_r = -div_r; div_c = -div_c; }
                a[kk] = make_real2(div_r, div_c);
            }
        }
    }
    if constexpr (PART == 0 || PART == 2) fft512_wave<1>(a, pen, tws, l);
#pragma unroll
    for (int k = 0; k < 8; k++) pen[l + 64 * k] = a[k];
    __syncthreads();
    // ---- inverse twiddle + inverse N-point DFTs across the chunks
    if constexpr (N > 1 && (PART == 0 || PART == 2)) {
#pragma unroll
        for (int s_ = 0; s_ < 8 / N; s_++) {
            const int idx = tid + 64 * WV * s_, zl = idx & 511, mm = idx >> 9;
            real2_t v[8];
#pragma unroll
            for (int k1 = 0; k1 < N; k1++) v[k1] = tile[(k1 * NM + mm) * FP + zl];
#pragma unroll
            for (int k1 = 1; k1 < N; k1++) {
                x3d_f64 sn, cs;
                sincospi(2.0 * (x3d_f64)(zl * k1) / (512.0 * N), &sn, &cs);
                v[k1] = cmul(v[k1], make_real2((real_t)cs, (real_t)sn));
            }
            dft_small<1, N>(v);
#pragma unroll
            for (int p = 0; p < N; p++) tile[(p * NM + mm) * FP + zl] = v[p];
        }
        __syncthreads();
    }
    if (valid) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int rr = it * (64 * N) + r, p = rr >> 9, zl = rr & 511;
            R[(long)rr * W + w0 + m] = tile[(p * NM + m) * FP + zl];
        }
    }
}

// the DFTs across the chunks as a pass of their own (N = 4, 8: with 16 / N modes per workgroup the fused kernel's row
// segments shrink to 64 / 32 bytes): one thread per (point, mode), in place; S = -1: DFT then twiddle, +1: the inverse
template <int S, int N>
__global__ void __launch_bounds__(256) k_radix_peers(real2_t *R, long W)
{
    const long w = (long)blockIdx.x * 256 + threadIdx.x;
    const int zl = blockIdx.y;
    if (w >= W) return;
    real2_t v[8];
#pragma unroll
    for (int p = 0; p < N; p++) v[p] = R[((long)p * 512 + zl) * W + w];
    if (S < 0) dft_small<-1, N>(v);
#pragma unroll
    for (int k1 = 1; k1 < N; k1++) {
        x3d_f64 sn, cs;
        sincospi(S * 2.0 * (x3d_f64)(zl * k1) / (512.0 * N), &sn, &cs);
        v[k1] = cmul(v[k1], make_real2((real_t)cs, (real_t)sn));
    }
    if (S > 0) dft_small<1, N>(v);
#pragma unroll
    for (int p = 0; p < N; p++) R[((long)p * 512 + zl) * W + w] = v[p];
}

static real2_t *g_tw = nullptr;  // W512^k = exp(-2 pi i k / 512), first half, shared by all plans

// R: one part of the received array [512 N][W]; waves: that part's [W][512 N]; returns *done = false when the
// sizes are not served (N not 1, 2, 4, 8)
int x3d_fft512_peers(x3d_backend *b, real2_t *R, long W, int npeers, const real_t *waves, const real_t *ab, int nx, int ny,
                     int nz, int nxs, int yoff, bool *done);

template <int MODE, int NP>
static int launch512(x3d_backend *b, real2_t *c, long stride_axis, long stride_other, int nxs, int nother,
                     const Spec000 &sp, real2_t *xbuf, int ys, int ysc)
{
    const int lds = sizeof(real2_t) * (NP * FP + 256);
    X3D_LDS_OPTIN(b, (k_fft512<MODE, NP>));
    dim3 grid((nxs + NP - 1) / NP, nother);
    hipLaunchKernelGGL((k_fft512<MODE, NP>), grid, dim3(64 * NP), lds, b->stream, c, g_tw, stride_axis,
                       stride_other, nxs, sp, xbuf, ys, ysc);
    X3D_HIP(hipGetLastError());
    return 0;
}

// the fused y pass of the z-first solve: C[nkz][512][px] (x: 512 modes), rwZ = [nkz][512 x][512 y]
int x3d_fft512_run_zh(x3d_backend *b, real2_t *c, long px, int kz0, int nkz, const real_t *rwZ, const real_t *ab, int nx,
                      int ny, int nz)
{
    X3D_REQUIRE(g_tw && nx == 512 && ny == 512 && nz == 512 && rwZ, "x3d_fft512_run_zh: 512^3 only");
    const real_t *ax = ab, *bx = ax + nx, *ay = bx + nx, *by = ay + ny, *az = by + ny, *bz = az + nz;
    // the planes kz0 .. kz0 + nkz - 1 (the kernel numbers its rows' other index from 0)
    c += (long)kz0 * ny * px;
    const Spec000 sp{nullptr, ax, bx, ay, by, az + kz0, bz + kz0, nx, ny, nz, rwZ + (size_t)kz0 * 512 * 512};
    static int np16 = -1;
    if (np16 < 0) { const char *e = getenv("X3D_ZFIRST_Y16"); np16 = (e && e[0] == '1') ? 1 : 0; }
    ProfScope ps(b, X3D_K_SPECTRAL, 1);
    if (np16) {
        const int lds = sizeof(real2_t) * (16 * FP + 256);
        X3D_LDS_OPTIN(b, (k_fft512<2, 16, true>));
        hipLaunchKernelGGL((k_fft512<2, 16, true>), dim3(512 / 16, nkz), dim3(1024), lds, b->stream, c, g_tw, px,
                           (long)ny * px, 512, sp, nullptr, 1, 1);
    } else {
        const int lds = sizeof(real2_t) * (8 * FP + 256);
        X3D_LDS_OPTIN(b, (k_fft512<2, 8, true>));
        hipLaunchKernelGGL((k_fft512<2, 8, true>), dim3(512 / 8, nkz), dim3(512), lds, b->stream, c, g_tw, px,
                           (long)ny * px, 512, sp, nullptr, 1, 1);
    }
    X3D_HIP(hipGetLastError());
    return 0;
}

const real2_t *x3d_fft512_twiddles() { return g_tw; }

int x3d_fft512_init()
{
    if (g_tw) return 0;
    std::vector<real2_t> h(256);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < 256; k++) {
        const long double t = -2.0L * pi * k / 512.0L;
        h[k] = make_real2((real_t)cosl(t), (real_t)sinl(t));
    }
    X3D_HIP(hipMalloc(&g_tw, sizeof(real2_t) * 256));
    X3D_HIP(hipMemcpy(g_tw, h.data(), sizeof(real2_t) * 256, hipMemcpyHostToDevice));
    return 0;
}

// axis: 1 = y (ny must be 512), 2 = z (nz must be 512); mode 0 fwd, 1 bwd, 2 fused z pass
int x3d_fft512_run_x(x3d_backend *b, real2_t *c, int nxs, int ny, int nz, int axis, int mode, const real_t *waves,
                     const real_t *ab, int nx, real2_t *xbuf, int ys, int ysc);
static const real_t *g_rwT = nullptr;  // set by x3d_fft512_set_rwT for the next fused z pass (poisson.hip)
void x3d_fft512_set_rwT(const real_t *rwT) { g_rwT = rwT; }

int x3d_fft512_run(x3d_backend *b, real2_t *c, int nxs, int ny, int nz, int axis, int mode, const real_t *waves,
                   const real_t *ab, int nx)
{
    return x3d_fft512_run_x(b, c, nxs, ny, nz, axis, mode, waves, ab, nx, nullptr, 1, 1);
}

// xbuf != null (y axis, mode 0 or 1): the far side of the pass is the slab-exchange buffer (see k_fft512)
int x3d_fft512_run_x(x3d_backend *b, real2_t *c, int nxs, int ny, int nz, int axis, int mode, const real_t *waves,
                     const real_t *ab, int nx, real2_t *xbuf, int ys, int ysc)
{
    const long sy = nxs, sz = (long)nxs * ny;
    const long stride_axis = axis == 1 ? sy : sz, stride_other = axis == 1 ? sz : sy;
    const int nother = axis == 1 ? nz : ny;
    X3D_REQUIRE((axis == 1 ? ny : nz) == 512, "x3d_fft512_run: axis length must be 512");
    X3D_REQUIRE(mode != 2 || axis == 2, "x3d_fft512_run: the fused pass is the z pass");
    X3D_REQUIRE(!xbuf || (axis == 1 && mode != 2 && ys > 0 && 512 % ys == 0 && ysc > 0 && ys % ysc == 0),
                "x3d_fft512_run: bad slab exchange");
    Spec000 sp{};
    if (mode == 2) {
        const real_t *ax = ab, *bx = ax + nx, *ay = bx + nx, *by = ay + ny, *az = by + ny, *bz = az + nz;
        sp = Spec000{waves, ax, bx, ay, by, az, bz, nx, ny, nz, g_rwT};
    }
    static int wide = -1;  // rows of 256 B (16 modes, one 16-wave workgroup per CU) instead of 128 B
    if (wide < 0) {
        const char *e = getenv("X3D_FFT512_WIDE");
        wide = e ? atoi(e) : 1;  // measured: y pass 0.53 vs 0.58 ms with 16 modes, z pass 0.90 vs 0.96 with 8
    }
    const bool w16 = (wide & axis) != 0;  // bit 0: y pass, bit 1: z pass
    ProfScope ps(b, mode == 2 ? X3D_K_SPECTRAL : X3D_K_FFT, axis);
#define GO(M_)                                                                                             \
    (w16 ? launch512<M_, 16>(b, c, stride_axis, stride_other, nxs, nother, sp, xbuf, ys, ysc)              \
         : launch512<M_, 8>(b, c, stride_axis, stride_other, nxs, nother, sp, xbuf, ys, ysc))
    if (mode == 0) return GO(0);
    if (mode == 1) return GO(1);
    return GO(2);
#undef GO
}

// ---------------------------------------------------------------- x pass, forward: real rows of 512 points
// Two real rows a, b are transformed as ONE complex row z = a + i b (a wave per pair, contiguous 4 KB loads):
//   A_k = (Z_k + conj(Z_{N-k})) / 2,   B_k = (Z_k - conj(Z_{N-k})) / (2 i),   k = 0 .. 256
// -- one kernel and half the butterflies of rocFFT's r2c pair (sbrr + r2c_even_post: 0.35 + 0.38 ms).
// Same unnormalised forward DFT (e^{-i}) as hipfftExecD2Z; results differ by rounding only.
__global__ void __launch_bounds__(512)
    k_r2c512(real2_t *__restrict__ c, const real_t *__restrict__ f, const real2_t *__restrict__ twg, long npairs,
             long frow, long crow)
{
    extern __shared__ real2_t tile[];  // [8][FP] + 256 twiddles
    real2_t *__restrict__ tws = tile + 8 * FP;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    __syncthreads();
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real2_t *__restrict__ pen = tile + w * FP;
    for (long pr = (long)blockIdx.x * 8 + w; pr < npairs; pr += (long)gridDim.x * 8) {
        const real_t *__restrict__ fa = f + 2 * pr * frow, *__restrict__ fb = fa + frow;
        real2_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = make_real2(fa[l + 64 * k], fb[l + 64 * k]);
        fft512_wave<-1>(a, pen, tws, l);
#pragma unroll
        for (int k = 0; k < 8; k++) pen[l + 64 * k] = a[k];  // (wave-private region, LDS operations in order)
        wave_lds_fence();
        real2_t *__restrict__ ca = c + 2 * pr * crow, *__restrict__ cb = ca + crow;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int idx = l + 64 * k;
            if (k < 4 || l == 0) {
                const real2_t x = a[k], y = pen[(512 - idx) & 511];
                ca[idx] = make_real2(0.5 * (x.x + y.x), 0.5 * (x.y - y.y));
                cb[idx] = make_real2(0.5 * (x.y + y.y), -0.5 * (x.x - y.x));
            }
        }
    }
}

int x3d_fft512_r2c(x3d_backend *b, real2_t *c, const real_t *f, long nrows, long frow, long crow)
{
    X3D_REQUIRE(g_tw && nrows % 2 == 0, "x3d_fft512_r2c: not initialised / odd number of rows");
    const int lds = sizeof(real2_t) * (8 * FP + 256);
    X3D_LDS_OPTIN(b, k_r2c512);
    const long npairs = nrows / 2;
    long blocks = (npairs + 7) / 8;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_r2c512, dim3((unsigned)blocks), dim3(512), lds, b->stream, c, f, g_tw, npairs, frow, crow);
    X3D_HIP(hipGetLastError());
    return 0;
}

template <bool YL, int PART = 0>
static int peers_run(x3d_backend *b, real2_t *R, long W, int npeers, const SpecSlab &sp)
{
    // X3D_SLAB_Z_SPLIT=1: the DFTs across the chunks as streaming passes (k_radix_peers) around the chunk-local kernel
    // -- 6 passes with 128-byte row segments instead of 2 with 64- / 32-byte ones.  Measured per solve at 512^3 per
    // rank (scratch/zstage_bench.py; transposes + rocFFT + division: 2.07 - 2.26 ms): 1 rank 0.76, 2 ranks 0.96,
    // 4 ranks 1.07 (split 1.62), 8 ranks 1.47 (split 1.86; 1.88 before the XCD-aware numbering of the mode groups)
    static int split_env = -2;
    if (split_env == -2) { const char *e = getenv("X3D_SLAB_Z_SPLIT"); split_env = e ? atoi(e) : -1; }
    const bool split = npeers > 1 && split_env > 0 && PART == 0;
#define LOCAL(NK_)                                                                                              \
    do {                                                                                                        \
        const int lds8 = sizeof(real2_t) * (8 * FP + 256);                                                      \
        X3D_LDS_OPTIN(b, (k_fft512_peers<1, 8, YL, PART>));                                                           \
        hipLaunchKernelGGL((k_fft512_peers<1, 8, YL, PART>), dim3((unsigned)((W + 7) / 8), NK_), dim3(512), lds8, b->stream, R, g_tw, W, \
                           sp, NK_);                                                                            \
    } while (0)
#define FUSED(N_)                                                                                               \
    do {                                                                                                        \
        const int lds16 = sizeof(real2_t) * (16 * FP + 256);                                                    \
        const long nm = 16 / N_, sh = nm < 8 ? 8 * (8 / nm) : 1;                                                \
        const long ng = ((W + nm - 1) / nm + sh - 1) / sh * sh; /* (whole blocks of the XCD numbering) */       \
        X3D_LDS_OPTIN(b, (k_fft512_peers<N_, 16, YL, PART>));                                                         \
        hipLaunchKernelGGL((k_fft512_peers<N_, 16, YL, PART>), dim3((unsigned)ng), dim3(1024), lds16, b->stream, R, g_tw, W, sp, N_); \
    } while (0)
#define SPLIT(N_)                                                                                               \
    do {                                                                                                        \
        const dim3 g((unsigned)((W + 255) / 256), 512);                                                         \
        hipLaunchKernelGGL((k_radix_peers<-1, N_>), g, dim3(256), 0, b->stream, R, W);                          \
        LOCAL(N_);                                                                                              \
        hipLaunchKernelGGL((k_radix_peers<1, N_>), g, dim3(256), 0, b->stream, R, W);                           \
    } while (0)
    if (npeers == 1) LOCAL(1);
    else if (npeers == 2) { if (split) SPLIT(2); else FUSED(2); }
    else if (npeers == 4) { if (split) SPLIT(4); else FUSED(4); }
    else { if (split) SPLIT(8); else FUSED(8); }
#undef SPLIT
#undef FUSED
#undef LOCAL
    X3D_HIP(hipGetLastError());
    return 0;
}

int x3d_fft512_peers(x3d_backend *b, real2_t *R, long W, int npeers, const real_t *waves, const real_t *ab, int nx, int ny,
                     int nz, int nxs, int yoff, bool *done)
{
    *done = false;
    if (!g_tw || nz != 512 * npeers || !(npeers == 1 || npeers == 2 || npeers == 4 || npeers == 8)) return 0;
    const real_t *ax = ab, *bx = ax + nx, *ay = bx + nx, *by = ay + ny, *az = by + ny, *bz = az + nz;
    const SpecSlab sp{waves, ax, bx, ay, by, az, bz, nx, ny, nz, nxs, yoff, 0};
    if (int rc = peers_run<false>(b, R, W, npeers, sp)) return rc;
    *done = true;
    return 0;
}

// y slabs with the z-first spectrum (csrc/sfftz.hip): R = one part of the received array [512 npeers (y)][W], W modes =
// [kzc][xs] with kz = kz0 + ..., kx = xoff + ...; rw = that part's [W][ny] reciprocal wave numbers (y fastest)
int x3d_fft512_peers_yl(x3d_backend *b, real2_t *R, long W, int npeers, const real_t *rw, const real_t *ab, int nx, int ny,
                        int nz, int xs, int xoff, int kz0, int part)
{
    X3D_REQUIRE(g_tw && ny == 512 * npeers && (npeers == 1 || npeers == 2 || npeers == 4 || npeers == 8),
                "x3d_fft512_peers_yl: %d chunks of 512 rows along y on 1, 2, 4 or 8 ranks", npeers);
    const real_t *ax = ab, *bx = ax + nx, *ay = bx + nx, *by = ay + ny, *az = by + ny, *bz = az + nz;
    const SpecSlab sp{rw, ax, bx, ay, by, az, bz, nx, ny, nz, xs, kz0, xoff};
    switch (part) {  // 0: forward + division + inverse; 1 / 2 / 3: forward / inverse / division alone (the hooks)
    case 0: return peers_run<true, 0>(b, R, W, npeers, sp);
    case 1: return peers_run<true, 1>(b, R, W, npeers, sp);
    case 2: return peers_run<true, 2>(b, R, W, npeers, sp);
    case 3: return peers_run<true, 3>(b, R, W, npeers, sp);
    }
    x3d_set_error("x3d_fft512_peers_yl: part must be 0 .. 3");
    return 2;
}
