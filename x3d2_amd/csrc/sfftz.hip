// All-periodic spectral Poisson solve (000) on Y SLABS, z-first: nproc_dir = [1, py, 1], 512 x 512 x 512 cells per rank
// (poisson_000, /root/reference/src/poisson_fft.f90:216-226; the reference's distributed solvers split differently --
// 2decomp&FFT pencils src/backend/omp/poisson_fft.f90:72-97, cuFFTMp z slabs src/backend/cuda/poisson_fft.f90:124-181 --
// and run their transposes and transforms one after the other).
//
// With y decomposed z stays whole on every rank, so the z transforms ride on the z operator pairs next to the solve
// exactly as on one rank (csrc/zfirst.hip, k_ytile_tds_pair<.., ZF>): the rank's spectrum C[kz][yl][x] (kz = 0 .. 256)
// appears without the divergence ever being stored.  Then
//   x   complex transform of the contiguous rows, stored straight into the exchange layout (k_c2c512_x_pack): peer r
//       gets the x modes [r xs, (r + 1) xs), xs = 512 / py, of every row
//   all-to-all inside the py ranks, in `parts` groups of kz planes
//   y   on the received chunks [peer p][yl][kzc][xs] = rows g = 512 p + yl of a [512 py][W] array: the py-point DFTs
//       across the chunks, 512-point transforms, process_spectral_000, and all of it back in ONE kernel
//       (k_fft512_peers<N, ., YL>, csrc/fft512.hip) -- a group's y stage runs beside the transfer of the next ones
//   all-to-all back, x inverse out of the exchange layout (k_c2c512_x_unpack), z inverse inside the gradient's z pair.
// 6.5 passes over the spectrum per solve and ONE all-to-all pair, like the z-slab solver (csrc/sfft.hip) -- which
// needs 10.5 passes because there the decomposed axis is the one the neighbouring operators work along.
// Exchange buffers: [part][peer][yl = 512][kzc_part][xs] complex numbers, a part's block contiguous.
//
// Round 5: BLOCKS of (y rows) x (kz planes).  The all-to-alls are the one term of a multi-GPU step that nothing hid
// (2 x 0.95 GB per solve over 7 links: 2 x 2.2 ms at 8 GPUs): the z pair in front of the solve had to
// finish for ALL tiles before the first group of kz planes could leave, and the z pair behind it could not start before
// the last group was back.  But a z pair works tile by tile -- a group of local y rows yields COMPLETE kz columns of those
// rows -- and in the exchange layout the rows yl0 .. yl1 of a (part, peer) chunk are one contiguous piece.  So the
// *_rows entry points below take a range of y rows: the z pair and the x transforms of rows group a + 1 run beside the
// transfers of group a, the y stage of kz group k starts when the LAST rows group's block of k has arrived (beside the
// transfers of the kz groups behind it), and the way back is rows-major (all kz groups of a rows group in one exchange
// group): the x transforms and the z pair of rows group a run when ITS blocks are in, beside the transfers of the rows
// groups behind it (poisson_fft.HipSlabPoissonFFTZ._pipelined; 8 emulated ranks 63.2 -> 54.1 ms per step,
// profiles/r05_yslab_pipeline_timeline.txt).  Same kernels on the same data: bit for bit the unsplit solve.
#include "zfft_tile.h"

#define SZ_PX 520
#define SZ_MAXPARTS 8

int x3d_fft512_init();
const real2_t *x3d_fft512_twiddles();
int x3d_fft512_peers_yl(x3d_backend *b, real2_t *R, long W, int npeers, const real_t *rw, const real_t *ab, int nx, int ny,
                        int nz, int xs, int xoff, int kz0, int part);
int x3d_ztile_fft_run(x3d_backend *b, real_t *f, const ZfArg &zf, bool fwd, int y0, int nyr);
int x3d_ytile_tds_pair_zf(x3d_backend *b, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                          const x3d_tdsops *ta, const x3d_tdsops *tb, const ZfArg &zf, bool *done, int y0, int nyr);

struct x3d_sfftz {
    x3d_backend *b;
    int py, ry, xs, xoff, parts;
    int kz0[SZ_MAXPARTS + 1];   // part m: planes kz0[m] .. kz0[m + 1] - 1
    long off[SZ_MAXPARTS + 1];  // part m's block in the exchange buffers (complex elements)
    real2_t *c;                 // C[257][512][SZ_PX]
    real_t *rw;                 // [257 * xs][512 py]: -1 / waves of this rank's modes, y fastest, part after part
    real_t *ab;                 // ax bx (512) ay by (512 py) az bz (512)
};

// rows (kz, yl) of C, kz in [k0, k1): forward transform along x, x mode kx -> peer kx / xs: S[(r * 512 + yl) * kzc + kz - k0][kx % xs]
template <bool PACK>
__global__ void __launch_bounds__(512)
    k_c2c512_x_xchg(real2_t *__restrict__ c, real2_t *__restrict__ sm, const real2_t *__restrict__ twg, int k0, int kzc,
                    int xs, int y0, int nyr)
{
    extern __shared__ real2_t zx[];  // [8][FP] + 256 twiddles
    real2_t *__restrict__ tws = zx + 8 * FP;
    if (threadIdx.x < 256) tws[threadIdx.x] = twg[threadIdx.x];
    __syncthreads();
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real2_t *__restrict__ pen = zx + w * FP;
    const long nrows = (long)kzc * nyr, step = (long)gridDim.x * 8;  // rows (kl, yl), yl in [y0, y0 + nyr)
    for (long row = (long)blockIdx.x * 8 + w; row < nrows; row += step) {
        const int kl = (int)(row / nyr), yl = y0 + (int)(row - (long)kl * nyr);
        real2_t *__restrict__ cr = c + ((long)(k0 + kl) * 512 + yl) * SZ_PX;
        real2_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int kx = l + 64 * k, r = kx / xs, xi = kx - r * xs;
            real2_t *__restrict__ sp = sm + (((long)r * 512 + yl) * kzc + kl) * xs + xi;
            if (PACK) a[k] = cr[kx]; else a[k] = *sp;
        }
        if (PACK) fft512_wave<-1>(a, pen, tws, l); else fft512_wave<1>(a, pen, tws, l);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int kx = l + 64 * k, r = kx / xs, xi = kx - r * xs;
            real2_t *__restrict__ sp = sm + (((long)r * 512 + yl) * kzc + kl) * xs + xi;
            if (PACK) *sp = a[k]; else cr[kx] = a[k];
        }
    }
}

extern "C" int x3d_sfftz_create(x3d_backend *b, x3d_sfftz **out, const int nglob[3], int py, int ry, int parts)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && nglob, "x3d_sfftz_create: null argument");
    X3D_REQUIRE(py == 1 || py == 2 || py == 4 || py == 8, "x3d_sfftz_create: 1, 2, 4 or 8 ranks along y");
    X3D_REQUIRE(ry >= 0 && ry < py, "x3d_sfftz_create: bad rank");
    X3D_REQUIRE(nglob[0] == 512 && nglob[1] == 512 * py && nglob[2] == 512 && b->nx == 512 && b->ny == 512 && b->nz == 512,
                "x3d_sfftz_create: 512 x 512 x 512 cells per rank");
    if (parts <= 0) parts = py > 1 ? 4 : 1;
    X3D_REQUIRE(parts <= SZ_MAXPARTS, "x3d_sfftz_create: at most %d parts", SZ_MAXPARTS);
    if (int rc = x3d_fft512_init()) return rc;
    x3d_sfftz *p = new x3d_sfftz();
    memset(p, 0, sizeof *p);
    p->b = b; p->py = py; p->ry = ry; p->xs = 512 / py; p->xoff = ry * p->xs; p->parts = parts;
    for (int m = 0; m <= parts; m++) {
        p->kz0[m] = (int)((long)257 * m / parts);
        p->off[m] = (long)p->kz0[m] * 512 * 512;  // (py peers x 512 rows x kzc x xs = 512 x 512 per plane)
    }
    X3D_HIP(hipMalloc(&p->c, sizeof(real2_t) * 257 * 512 * SZ_PX));
    X3D_HIP(hipMemset(p->c, 0, sizeof(real2_t) * 257 * 512 * SZ_PX));
    X3D_HIP(hipMalloc(&p->rw, sizeof(real_t) * 257 * p->xs * 512 * py));
    X3D_HIP(hipMalloc(&p->ab, sizeof(real_t) * 2 * (512 + 512 * py + 512)));
    *out = p;
    return 0;
}

extern "C" int x3d_sfftz_destroy(x3d_sfftz *p)
{
    X3D_RANGE(__func__);
    if (!p) return 0;
    hipFree(p->c); hipFree(p->rw); hipFree(p->ab);
    delete p;
    return 0;
}

// this solver's spectrum C[257][512][px] (for x3d_poisson_create_proxy: the hooks of a host that runs the middle itself)
extern "C" int x3d_sfftz_spectrum(const x3d_sfftz *p, real_t **c, int *ny, long *px)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && c && ny && px, "null argument");
    *c = reinterpret_cast<real_t *>(p->c); *ny = 512; *px = SZ_PX;
    return 0;
}

// out = {parts, xs, xoff, complex elements of the exchange buffers, kz0[0 .. parts]}
extern "C" int x3d_sfftz_sizes(const x3d_sfftz *p, long out[16])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && out, "null argument");
    out[0] = p->parts; out[1] = p->xs; out[2] = p->xoff; out[3] = (long)257 * 512 * 512;
    for (int m = 0; m <= p->parts; m++) out[4 + m] = p->kz0[m];
    return 0;
}

// rw: -1 / waves (0 where waves < 1e-16) of this rank's modes, [kz = 0 .. 256][x = xoff .. xoff + xs)[y = 0 .. 512 py),
// y fastest (a part's modes are then contiguous: W = kzc * xs rows of ny); ax .. bz: global lengths
extern "C" int x3d_sfftz_set_waves(x3d_sfftz *p, const real_t *rw, const real_t *ax, const real_t *bx, const real_t *ay,
                                   const real_t *by, const real_t *az, const real_t *bz)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && rw && ax && bx && ay && by && az && bz, "null argument");
    X3D_HIP(hipMemcpy(p->rw, rw, sizeof(real_t) * 257 * p->xs * 512 * p->py, hipMemcpyHostToDevice));
    real_t *d = p->ab;
    const real_t *src[6] = {ax, bx, ay, by, az, bz};
    const int len[6] = {512, 512, 512 * p->py, 512 * p->py, 512, 512};
    for (int i = 0; i < 6; i++) {
        X3D_HIP(hipMemcpy(d, src[i], sizeof(real_t) * len[i], hipMemcpyHostToDevice));
        d += len[i];
    }
    return 0;
}

static ZfArg zfarg(const x3d_sfftz *p) { return ZfArg{p->c, x3d_fft512_twiddles(), 512, (long)SZ_PX}; }

// the z pairs next to the solve (as x3d_tds_pair_zfirst); *done = 0: these operators are not served, nothing was done
extern "C" int x3d_sfftz_tds_pair(x3d_sfftz *p, int mode, real_t *out1, real_t *out2, const real_t *in1, const real_t *in2,
                                  const x3d_tdsops *ta, const x3d_tdsops *tb, int *done)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && ta && tb && done, "x3d_sfftz_tds_pair: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_sfftz_tds_pair: mode must be 0 or 1");
    X3D_REQUIRE(mode == 0 ? (in1 && in2) : (out1 && out2 && out1 != out2), "x3d_sfftz_tds_pair: null argument");
    X3D_LAZY_SYNC(p->b);
    X3D_LAZY_EAGER(p->b);
    bool ok = false;
    *done = 0;
    if (int rc = x3d_ytile_tds_pair_zf(p->b, mode, out1, out2, in1, in2, ta, tb, zfarg(p), &ok, 0, -1)) return rc;
    *done = ok ? 1 : 0;
    return 0;
}
// ... for the tiles of the local y rows [y0, y0 + nyr) only
extern "C" int x3d_sfftz_tds_pair_rows(x3d_sfftz *p, int mode, real_t *out1, real_t *out2, const real_t *in1,
                                       const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb, int y0, int nyr, int *done)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && ta && tb && done, "x3d_sfftz_tds_pair_rows: null argument");
    X3D_REQUIRE(mode == 0 || mode == 1, "x3d_sfftz_tds_pair_rows: mode must be 0 or 1");
    X3D_REQUIRE(mode == 0 ? (in1 && in2) : (out1 && out2 && out1 != out2), "x3d_sfftz_tds_pair_rows: null argument");
    X3D_REQUIRE(y0 >= 0 && nyr >= 0 && y0 + nyr <= 512, "x3d_sfftz_tds_pair_rows: rows [%d, %d) of 512", y0, y0 + nyr);
    X3D_LAZY_SYNC(p->b);
    X3D_LAZY_EAGER(p->b);
    bool ok = false;
    *done = 0;
    if (int rc = x3d_ytile_tds_pair_zf(p->b, mode, out1, out2, in1, in2, ta, tb, zfarg(p), &ok, y0, nyr)) return rc;
    *done = ok ? 1 : 0;
    return 0;
}

// the z transform of a field in memory (the hooks' form; the solver's fused driver uses x3d_sfftz_tds_pair)
extern "C" int x3d_sfftz_z(x3d_sfftz *p, real_t *f, int inverse)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "null argument");
    X3D_LAZY_SYNC(p->b);
    return x3d_ztile_fft_run(p->b, f, zfarg(p), !inverse, 0, -1);
}
// the same for the reference's hooks under deferred execution (the Fortran shim, round 6): fft_forward's input / fft_backward's
// output is a HANDLE -- the queue is flushed and the buffer that holds the field used (inverse: the whole block is written,
// a shared buffer is not copied first) instead of restoring "every block holds its own data" (x3d_sfftz_z's X3D_LAZY_SYNC
// would materialise the reference's p_temp, a reordered alias of div_u: a 1 GB copy per solve)
extern "C" int x3d_sfftz_z_field(x3d_sfftz *p, real_t *f, int inverse)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "null argument");
    if (inverse) { X3D_LAZY_OUT(p->b, f, true); }
    else {
        const real_t *fi = f;
        X3D_LAZY_IN(p->b, fi);
        f = const_cast<real_t *>(fi);
    }
    return x3d_ztile_fft_run(p->b, f, zfarg(p), !inverse, 0, -1);
}
extern "C" int x3d_sfftz_z_rows(x3d_sfftz *p, real_t *f, int inverse, int y0, int nyr)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(p && f, "null argument");
    X3D_REQUIRE(y0 >= 0 && nyr >= 0 && y0 + nyr <= 512, "x3d_sfftz_z_rows: rows [%d, %d) of 512", y0, y0 + nyr);
    X3D_LAZY_SYNC(p->b);
    return x3d_ztile_fft_run(p->b, f, zfarg(p), !inverse, y0, nyr);
}

#define SZ_PART(p, m, name) X3D_REQUIRE((p) && (m) >= 0 && (m) < (p)->parts, name ": part %d of %d", (m), (p) ? (p)->parts : 0)

template <bool PACK>
static int x_xchg(x3d_sfftz *p, real_t *buf, int m, int y0 = 0, int nyr = 512)
{
    const int lds = sizeof(real2_t) * (8 * FP + 256);
    X3D_LDS_OPTIN(p->b, (k_c2c512_x_xchg<PACK>));
    const int kzc = p->kz0[m + 1] - p->kz0[m];
    if (nyr == 0) return 0;
    long blocks = ((long)kzc * nyr + 7) / 8;
    if (blocks > 2048) blocks = 2048;
    ProfScope ps(p->b, X3D_K_FFT, PACK ? 1 : 2);
    hipLaunchKernelGGL((k_c2c512_x_xchg<PACK>), dim3((unsigned)blocks), dim3(512), lds, p->b->stream, p->c,
                       (real2_t *)buf + p->off[m], x3d_fft512_twiddles(), p->kz0[m], kzc, p->xs, y0, nyr);
    X3D_HIP(hipGetLastError());
    return 0;
}
// part m of the spectrum: x forward, into sendbuf's part block
extern "C" int x3d_sfftz_x_forward(x3d_sfftz *p, real_t *sendbuf, int m)
{
    X3D_RANGE(__func__);
    SZ_PART(p, m, "x3d_sfftz_x_forward");
    X3D_REQUIRE(sendbuf, "null argument");
    return x_xchg<true>(p, sendbuf, m);
}
// part m received: y forward + process_spectral_000 + y inverse, in place (what = 0); what = 1 / 2 / 3: the forward
// transform / the inverse / the division alone, for the hooks of the reference's interface called one by one
extern "C" int x3d_sfftz_y_stage(x3d_sfftz *p, real_t *recvbuf, int m, int what)
{
    X3D_RANGE(__func__);
    SZ_PART(p, m, "x3d_sfftz_y_stage");
    X3D_REQUIRE(recvbuf && what >= 0 && what <= 3, "x3d_sfftz_y_stage: bad argument");
    const int kzc = p->kz0[m + 1] - p->kz0[m];
    const long W = (long)kzc * p->xs;
    ProfScope ps(p->b, X3D_K_SPECTRAL, 1);
    return x3d_fft512_peers_yl(p->b, (real2_t *)recvbuf + p->off[m], W, p->py,
                               p->rw + (long)p->kz0[m] * p->xs * 512 * p->py, p->ab, 512, 512 * p->py, 512, p->xs, p->xoff,
                               p->kz0[m], what);
}
// part m back in buf's part block: x inverse, into the spectrum
extern "C" int x3d_sfftz_x_backward(x3d_sfftz *p, const real_t *buf, int m)
{
    X3D_RANGE(__func__);
    SZ_PART(p, m, "x3d_sfftz_x_backward");
    X3D_REQUIRE(buf, "null argument");
    return x_xchg<false>(p, const_cast<real_t *>(buf), m);
}

// block (rows [y0, y0 + nyr), part m): x forward into sendbuf / x inverse out of buf -- the rows' piece of every (part, peer)
// chunk is contiguous: complex elements [off(m) + (r 512 + y0) kzc xs, + nyr kzc xs) for peer r
#define SZ_ROWS(y0, nyr, name) X3D_REQUIRE((y0) >= 0 && (nyr) >= 0 && (y0) + (nyr) <= 512, name ": rows [%d, %d) of 512", (y0), (y0) + (nyr))
extern "C" int x3d_sfftz_x_forward_rows(x3d_sfftz *p, real_t *sendbuf, int m, int y0, int nyr)
{
    X3D_RANGE(__func__);
    SZ_PART(p, m, "x3d_sfftz_x_forward_rows");
    SZ_ROWS(y0, nyr, "x3d_sfftz_x_forward_rows");
    X3D_REQUIRE(sendbuf, "null argument");
    return x_xchg<true>(p, sendbuf, m, y0, nyr);
}
extern "C" int x3d_sfftz_x_backward_rows(x3d_sfftz *p, const real_t *buf, int m, int y0, int nyr)
{
    X3D_RANGE(__func__);
    SZ_PART(p, m, "x3d_sfftz_x_backward_rows");
    SZ_ROWS(y0, nyr, "x3d_sfftz_x_backward_rows");
    X3D_REQUIRE(buf, "null argument");
    return x_xchg<false>(p, const_cast<real_t *>(buf), m, y0, nyr);
}
