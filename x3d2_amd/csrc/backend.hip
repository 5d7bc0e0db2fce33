// Backend context, block storage, tdsops device tables, BLAS-1 field ops,
// reductions, face setters and host<->device field copies.
//
// Reference behaviour mirrored here (paths under /root/reference):
//   src/backend/omp/backend.f90:66-114 (constructor), :529-614 (vec ops),
//   :651-712 (scalar_product), :739-810 (field_max_mean), :812-872
//   (slice_max_sum), :874-901 (scale/shift), :903-1021 (face setters),
//   :1023-1066 (volume integral), src/backend/backend.f90:402-466
//   (get/set_field_data), src/allocator.f90:64-93 (padding).
#include <mutex>

#include <algorithm>
#include <vector>

#include "common.h"

#include <unordered_map>
#include <unordered_set>

#include <cmath>

static thread_local char g_err[512] = "";

void x3d_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *x3d_last_error(void) { return g_err; }
extern "C" int x3d_abi_version(void) { return 1; }
extern "C" int x3d_real_bytes(void) { return (int)sizeof(real_t); }

PencilGeom x3d_geom(const x3d_backend *b, int dir)
{
    PencilGeom g;
    const long nxp = b->nxp, nyp = b->nyp;
    switch (dir) {
    case X3D_DIR_X:  // pencil (y,z): lanes run over y
        g.np = b->ny * b->nz; g.dim0 = b->ny; g.s0 = nxp; g.s1 = nxp * nyp; g.rs = 1; break;
    case X3D_DIR_Y:  // pencil (x,z): lanes run over x
        g.np = b->nx * b->nz; g.dim0 = b->nx; g.s0 = 1; g.s1 = nxp * nyp; g.rs = nxp; break;
    default:         // DIR_Z, pencil (x,y)
        g.np = b->nx * b->ny; g.dim0 = b->nx; g.s0 = 1; g.s1 = nxp; g.rs = nxp * nyp; break;
    }
    return g;
}

extern "C" int x3d_backend_create(x3d_backend **out, const int dims_vert[3], int device, void *stream)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(out && dims_vert, "x3d_backend_create: null argument");
    X3D_REQUIRE(dims_vert[0] > 0 && dims_vert[1] > 0 && dims_vert[2] > 0,
                "x3d_backend_create: dims must be positive");
    int ndev = 0;
    X3D_HIP(hipGetDeviceCount(&ndev));
    X3D_REQUIRE(ndev > 0, "x3d_backend_create: no HIP device visible (the HIP backend has no CPU path)");
    X3D_REQUIRE(device >= 0 && device < ndev, "x3d_backend_create: device %d out of range", device);
    X3D_HIP(hipSetDevice(device));
    x3d_backend *b = new x3d_backend();
    memset(b, 0, sizeof *b);
    b->device = device;
    b->prof_mask = ~0u;
    b->stream = (hipStream_t)stream;
    b->nx = dims_vert[0]; b->ny = dims_vert[1]; b->nz = dims_vert[2];
    b->nxp = (b->nx + 15) / 16 * 16;
    {
        // Row pitches that are multiples of 512 B make the y stride (nxp * 8 B) and above all the z stride
        // (nxp * nyp * 8 B = 2 MiB at 512^3) powers of two: the 512 row segments of a z tile then alias in the
        // memory channels.  One more 128-byte segment per row breaks that (512^3, same box, scratch/tile_bench.py):
        // z operator pairs 0.91 -> 0.71 and 0.84 -> 0.68 ms, transeq_z 2.48 -> 2.29, y pairs 0.67 -> 0.64; the
        // contiguous-row x kernels lose a little in isolation (k_xscan_tds_lin 0.81 -> 0.89 ms; two segments:
        // 0.83), but the full step is the same for 16 / 32 / 48 / 80 doubles of padding: 49.0-49.6 ms against
        // 51.3 without (profiles/README.md).  3 % more memory.  X3D_PAD_X=<doubles> overrides (0: none).
        const char *e = getenv("X3D_PAD_X");
        if (e) b->nxp += (atoi(e) + 15) / 16 * 16;
        else if (b->nxp % 64 == 0 && b->nxp >= 256) b->nxp += 16;
    }
    b->nyp = b->ny;
    b->nzp = b->nz;
    b->nblock = (size_t)b->nxp * b->nyp * b->nzp;
    int npmax = b->ny * b->nz;
    if (b->nx * b->nz > npmax) npmax = b->nx * b->nz;
    if (b->nx * b->ny > npmax) npmax = b->nx * b->ny;
    X3D_HIP(hipMalloc(&b->send_s, sizeof(real_t) * 3 * (size_t)npmax));
    X3D_HIP(hipMalloc(&b->send_e, sizeof(real_t) * 3 * (size_t)npmax));
    for (int i = 0; i < 3; i++)
        X3D_HIP(hipMalloc(&b->scratch[i], sizeof(real_t) * (b->nblock + 64 * (size_t)b->nxp)));
    b->red_cap = 4096;
    X3D_HIP(hipMalloc(&b->red_buf, sizeof(real_t) * 2 * b->red_cap));
    X3D_HIP(hipHostMalloc(&b->red_host, sizeof(real_t) * 2 * b->red_cap));
    X3D_HIP(hipMalloc(&b->epi_dev, 512));
    b->lds_optin = new std::unordered_set<const void *>();
    X3D_HIP(hipEventCreate(&b->ev0));
    X3D_HIP(hipEventCreate(&b->ev1));
    *out = b;
    return 0;
}

extern "C" int x3d_backend_create_like(x3d_backend **out, const x3d_backend *like, const int dims_vert[3])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(like, "x3d_backend_create_like: null argument");
    return x3d_backend_create(out, dims_vert, like->device, (void *)like->stream);
}

extern "C" int x3d_backend_destroy(x3d_backend *b)
{
    X3D_RANGE(__func__);
    if (!b) return 0;
    if (b->ipc_maps) {  // peers' buffers still mapped: unmap them before this rank's own memory goes
        auto *v = static_cast<std::vector<void *> *>(b->ipc_maps);
        for (void *p : *v) (void)hipIpcCloseMemHandle(p);
        delete v;
    }
    hipFree(b->send_s); hipFree(b->send_e);
    hipFree(b->scratch[0]); hipFree(b->scratch[1]); hipFree(b->scratch[2]);
    x3d_prof_enable_c(b, 0);
    hipFree(b->red_buf); hipHostFree(b->red_host); hipFree(b->epi_dev);
    hipEventDestroy(b->ev0); hipEventDestroy(b->ev1);
    delete static_cast<std::unordered_set<const void *> *>(b->lds_optin);
    x3d_lazy_destroy(b);
    delete b;
    return 0;
}

int x3d_lds_optin(x3d_backend *b, const void *kernel)
{
    auto *seen = static_cast<std::unordered_set<const void *> *>(b->lds_optin);
    if (seen->count(kernel)) return 0;
    X3D_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    seen->insert(kernel);
    return 0;
}

// which = 0: launches of the three-components-in-one transeq kernels since creation; 1: those of them that
// also applied a pending velocity correction
extern "C" long x3d_backend_counter(const x3d_backend *b, int which)
{
    X3D_RANGE(__func__);
    if (!b) return -1;
    return which == 0 ? b->n_tq3 : (which == 1 ? b->n_upd : (which == 2 ? b->n_halo : -1));
}

extern "C" int x3d_backend_set_stream(x3d_backend *b, void *stream)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "null backend");
    b->stream = (hipStream_t)stream;
    return 0;
}

// CUs the persistent kernels leave free for the kernels of other streams (RCCL's send / recv kernels of an exchange that
// is to run beside them; common.h, comm_reserve): 0 = none (one rank), a multi-rank driver sets 8 (one per XCD)
extern "C" int x3d_backend_set_comm_reserve(x3d_backend *b, int ncus)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "null backend");
    X3D_REQUIRE(ncus >= 0 && ncus < X3D_NCU / 2, "x3d_backend_set_comm_reserve: 0 .. %d CUs", X3D_NCU / 2 - 1);
    b->comm_reserve = ncus;
    return 0;
}

// a decomposed direction that is periodic over ALL its ranks (the mesh knows; an operator with BC_HALO ends does not): the
// single-pass HALO kernels may then use the open-ended circulant solve, whose boundary values mean something else than
// du_1 / X_n -- every rank of the ring makes the same choice because every rank is told the same.  Default 0: the table form
extern "C" int x3d_backend_set_ring(x3d_backend *b, int dir, int periodic_over_all_ranks)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && x3d_dir_ok(dir), "x3d_backend_set_ring: null backend or bad direction");
    b->ring[dir] = periodic_over_all_ranks ? 1 : 0;
    return 0;
}

extern "C" size_t x3d_block_elems(const x3d_backend *b) { return b ? b->nblock : 0; }

extern "C" int x3d_padded_dims(const x3d_backend *b, int d[3])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && d, "null argument");
    d[0] = b->nxp; d[1] = b->nyp; d[2] = b->nzp;
    return 0;
}

extern "C" int x3d_device_sync(x3d_backend *b)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_FLUSH(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b, "null backend");
    X3D_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

// Blocks start 4224 B further into their allocation than the previous one (modulo 16): the kernels stream several
// blocks at the same relative offset at once, and with every block on a 2 MiB boundary those streams meet in the
// same memory channels (x3d2_amd/field.py has the measurement: -2 % per step at 512^3)
// (process-wide table: several backends -- the twin backends of Poisson 100 / 110, one per host thread -- may
// allocate and free at the same time)
static std::unordered_map<real_t *, void *> g_block_base;
static int g_block_count = 0;
static std::mutex g_block_mutex;

extern "C" int x3d_block_alloc(x3d_backend *b, real_t **out)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out, "null argument");
    const size_t st = 528;
    void *base = nullptr;
    X3D_HIP(hipSetDevice(b->device));
    X3D_HIP(hipMalloc(&base, sizeof(real_t) * (b->nblock + 16 * st)));
    std::lock_guard<std::mutex> lock(g_block_mutex);
    *out = static_cast<real_t *>(base) + (size_t)(g_block_count++ % 16) * st;
    g_block_base[*out] = base;
    x3d_lazy_register(b, *out);  // (a handle of the deferred-execution layer, should the caller switch it on)
    return 0;
}

extern "C" int x3d_block_free(x3d_backend *b, real_t *p)
{
    X3D_RANGE(__func__);
    if (b) { X3D_LAZY_SYNC(b); x3d_lazy_unregister(b, p); }
    (void)b;
    void *base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_block_mutex);
        auto it = g_block_base.find(p);
        X3D_REQUIRE(it != g_block_base.end(), "x3d_block_free: not a block of x3d_block_alloc");
        base = it->second;
        g_block_base.erase(it);
    }
    X3D_HIP(hipFree(base));
    return 0;
}

// exchange buffers + host staging for callers whose MPI is not GPU-aware (the Fortran shim on more than one rank:
// sendrecv_fields, src/backend/cuda/sendrecv.f90:13-42, through host memory)
extern "C" int x3d_device_alloc(x3d_backend *b, real_t **out, long n)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out && n > 0, "x3d_device_alloc: bad argument");
    X3D_HIP(hipMalloc(reinterpret_cast<void **>(out), sizeof(real_t) * (size_t)n));
    X3D_HIP(hipMemsetAsync(*out, 0, sizeof(real_t) * (size_t)n, b->stream));
    return 0;
}
extern "C" int x3d_device_free(x3d_backend *b, real_t *p)
{
    X3D_RANGE(__func__);
    (void)b;
    X3D_HIP(hipFree(p));
    return 0;
}
// ---- device-to-device exchanges between the ranks of one node (round 4).  The reference's GPU backend hands device
// buffers to a GPU-aware MPI (src/backend/cuda/sendrecv.f90:13-42); where the MPI at hand is not GPU-aware the same
// thing is done here with HIP's inter-process memory handles: a rank exports its exchange buffers once, its neighbours
// map them, and an exchange is a device-to-device copy on the PULLING rank's stream -- over xGMI between two GPUs, inside
// HBM when ranks share a GPU; only the 64-byte handles and empty "ready" messages go through MPI.
extern "C" int x3d_device_count(int *n)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(n, "x3d_device_count: null argument");
    X3D_HIP(hipGetDeviceCount(n));
    return 0;
}
extern "C" int x3d_ipc_export(x3d_backend *b, const real_t *dev, unsigned char handle[64])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && dev && handle, "x3d_ipc_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    X3D_HIP(hipIpcGetMemHandle(&h, const_cast<real_t *>(dev)));
    memcpy(handle, &h, 64);
    return 0;
}
extern "C" int x3d_ipc_open(x3d_backend *b, const unsigned char handle[64], real_t **dev)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && dev && handle, "x3d_ipc_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    void *p = nullptr;
    X3D_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *dev = static_cast<real_t *>(p);
    if (!b->ipc_maps) b->ipc_maps = new std::vector<void *>();
    static_cast<std::vector<void *> *>(b->ipc_maps)->push_back(p);  // (unmapped by x3d_backend_destroy at the latest)
    return 0;
}
extern "C" int x3d_ipc_close(x3d_backend *b, real_t *dev)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && dev, "x3d_ipc_close: null argument");
    X3D_HIP(hipIpcCloseMemHandle(dev));
    if (b->ipc_maps) {
        auto *v = static_cast<std::vector<void *> *>(b->ipc_maps);
        v->erase(std::remove(v->begin(), v->end(), static_cast<void *>(dev)), v->end());
    }
    return 0;
}
// n doubles device to device (own or mapped memory), ordered on the backend's stream like a kernel; returns at once
extern "C" int x3d_copy_device(x3d_backend *b, real_t *dst, const real_t *src, long n)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && dst && src && n >= 0, "x3d_copy_device: bad argument");
    if (n == 0) return 0;
    X3D_HIP(hipMemcpyAsync(dst, src, sizeof(real_t) * (size_t)n, hipMemcpyDeviceToDevice, b->stream));
    return 0;
}

// ordered behind the kernels queued on the backend's stream; returns when the copy is complete
extern "C" int x3d_copy_to_host(x3d_backend *b, real_t *host, const real_t *dev, long n)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && host && dev && n >= 0, "x3d_copy_to_host: bad argument");
    X3D_HIP(hipMemcpyAsync(host, dev, sizeof(real_t) * (size_t)n, hipMemcpyDeviceToHost, b->stream));
    X3D_HIP(hipStreamSynchronize(b->stream));
    return 0;
}
extern "C" int x3d_copy_to_device(x3d_backend *b, real_t *dev, const real_t *host, long n)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && host && dev && n >= 0, "x3d_copy_to_device: bad argument");
    X3D_HIP(hipMemcpyAsync(dev, host, sizeof(real_t) * (size_t)n, hipMemcpyHostToDevice, b->stream));
    X3D_HIP(hipStreamSynchronize(b->stream));  // (the host array may be reused at once)
    return 0;
}

// ---------------------------------------------------------------- x <-> y transposed copy between two block layouts
// Poisson 100 (x non-periodic): the reference transposes x <-> y and runs its 010 machinery on the transposed
// problem (memcpy3D_with_transpose / _back, src/backend/cuda/kernels/spectral_processing.f90:30-76, called by
// fft_forward_100 / fft_backward_100, src/backend/cuda/poisson_fft.f90:482-616).  dst belongs to a backend of
// the transposed dims: dst(y, x, z) = src(x, y, z) for x < nx, y < ny, z < nz; 32 x 32 tiles through LDS.
// dst[c][a][b] = src[c][b][a] in terms of strides: a is the fast axis of src, b the fast axis of dst, c a batch
__global__ void __launch_bounds__(256) k_transpose_ab(real_t *__restrict__ dst, const real_t *__restrict__ src, int na,
                                                      int nb, long s_b, long s_c, long d_a, long d_c)
{
    __shared__ real_t t[32][33];
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32, c = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (a0 + tx < na && b0 + r < nb) t[r][tx] = src[(long)c * s_c + (long)(b0 + r) * s_b + a0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (b0 + tx < nb && a0 + r < na) dst[(long)c * d_c + (long)(a0 + r) * d_a + b0 + tx] = t[tx][r];
}
static int transpose_launch(x3d_backend *bs, real_t *dst, const real_t *src, int na, int nb, int nc, long s_b, long s_c,
                            long d_a, long d_c)
{
    ProfScope ps(bs, X3D_K_COPY);
    hipLaunchKernelGGL(k_transpose_ab, dim3((na + 31) / 32, (nb + 31) / 32, nc), dim3(256), 0, bs->stream, dst, src, na, nb,
                       s_b, s_c, d_a, d_c);
    X3D_HIP(hipGetLastError());
    return 0;
}
extern "C" int x3d_transpose_xy(x3d_backend *bs, x3d_backend *bd, real_t *dst, const real_t *src, int nx, int ny, int nz)
{
    X3D_RANGE(__func__);
    if (bs) X3D_LAZY_SYNC(bs);
    if (bd) X3D_LAZY_SYNC(bd);
    X3D_LAZY_EAGER(bs);
    LazyScope lazy_scope_d_(bd);
    X3D_REQUIRE(bs && bd && dst && src, "x3d_transpose_xy: null argument");
    X3D_REQUIRE(nx <= bs->nxp && ny <= bs->nyp && nz <= bs->nzp && ny <= bd->nxp && nx <= bd->nyp && nz <= bd->nzp,
                "x3d_transpose_xy: dims exceed the blocks");
    X3D_REQUIRE(bs->stream == bd->stream && bs->device == bd->device, "x3d_transpose_xy: the two backends must share a device and a stream");
    return transpose_launch(bs, dst, src, nx, ny, nz, (long)bs->nxp, (long)bs->nxp * bs->nyp, (long)bd->nxp,
                            (long)bd->nxp * bd->nyp);
}
// Poisson 110: the reference moves z to the front (transpose_xyz_to_zxy / _zxy_to_xyz,
// src/backend/cuda/kernels/spectral_processing.f90:78-125, called by fft_forward_110 / fft_backward_110) so that
// the R2C runs along the periodic z: dst(z, x, y) = src(x, y, z); dst belongs to a backend of dims (nz, nx, ny)
extern "C" int x3d_transpose_xyz_zxy(x3d_backend *bs, x3d_backend *bd, real_t *dst, const real_t *src, int nx, int ny,
                                     int nz)
{
    X3D_RANGE(__func__);
    if (bs) X3D_LAZY_SYNC(bs);
    if (bd) X3D_LAZY_SYNC(bd);
    X3D_LAZY_EAGER(bs);
    LazyScope lazy_scope_d_(bd);
    X3D_REQUIRE(bs && bd && dst && src, "x3d_transpose_xyz_zxy: null argument");
    X3D_REQUIRE(nx <= bs->nxp && ny <= bs->nyp && nz <= bs->nzp && nz <= bd->nxp && nx <= bd->nyp && ny <= bd->nzp,
                "x3d_transpose_xyz_zxy: dims exceed the blocks");
    X3D_REQUIRE(bs->stream == bd->stream && bs->device == bd->device, "x3d_transpose_xyz_zxy: the two backends must share a device and a stream");
    // a = x (fast in src), b = z (fast in dst), batch = y
    return transpose_launch(bs, dst, src, nx, nz, ny, (long)bs->nxp * bs->nyp, (long)bs->nxp, (long)bd->nxp,
                            (long)bd->nxp * bd->nyp);
}
// and back: dst(x, y, z) = src(z, x, y); src belongs to the backend of dims (nz, nx, ny)
extern "C" int x3d_transpose_zxy_xyz(x3d_backend *bs, x3d_backend *bd, real_t *dst, const real_t *src, int nx, int ny,
                                     int nz)
{
    X3D_RANGE(__func__);
    if (bs) X3D_LAZY_SYNC(bs);
    if (bd) X3D_LAZY_SYNC(bd);
    X3D_LAZY_EAGER(bs);
    LazyScope lazy_scope_d_(bd);
    X3D_REQUIRE(bs && bd && dst && src, "x3d_transpose_zxy_xyz: null argument");
    X3D_REQUIRE(nx <= bd->nxp && ny <= bd->nyp && nz <= bd->nzp && nz <= bs->nxp && nx <= bs->nyp && ny <= bs->nzp,
                "x3d_transpose_zxy_xyz: dims exceed the blocks");
    X3D_REQUIRE(bs->stream == bd->stream && bs->device == bd->device, "x3d_transpose_zxy_xyz: the two backends must share a device and a stream");
    // a = z (fast in src), b = x (fast in dst), batch = y
    return transpose_launch(bs, dst, src, nz, nx, ny, (long)bs->nxp, (long)bs->nxp * bs->nyp, (long)bd->nxp * bd->nyp,
                            (long)bd->nxp);
}

// ---------------------------------------------------------------- BLAS-1
// Whole padded blocks, like the reference (src/backend/omp/backend.f90:545-557):
// streaming, 16 B per lane, grid-stride over at most 2048 workgroups.
template <class F>
__global__ void __launch_bounds__(256) k_map2(real2_t *__restrict__ y, const real2_t *__restrict__ x,
                                              size_t n2, F f)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += st) {
        real2_t a = x[i], c = y[i];
        c.x = f(a.x, c.x);
        c.y = f(a.y, c.y);
        y[i] = c;
    }
}

template <class F>
__global__ void __launch_bounds__(256) k_map1(real2_t *__restrict__ y, size_t n2, F f)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += st) {
        real2_t c = y[i];
        c.x = f(c.x);
        c.y = f(c.y);
        y[i] = c;
    }
}

static inline int stream_grid(size_t n2)
{
    size_t g = (n2 + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g ? g : 1));
}

struct OpCopy { __device__ real_t operator()(real_t x, real_t) const { return x; } };
struct OpAxpby { real_t a, b; __device__ real_t operator()(real_t x, real_t y) const { return a * x + b * y; } };
struct OpMul { __device__ real_t operator()(real_t x, real_t y) const { return y * x; } };
struct OpAdd { __device__ real_t operator()(real_t x, real_t y) const { return y + x; } };
struct OpScale { real_t a; __device__ real_t operator()(real_t y) const { return a * y; } };
struct OpShift { real_t a; __device__ real_t operator()(real_t y) const { return y + a; } };
struct OpFill { real_t a; __device__ real_t operator()(real_t) const { return a; } };

extern "C" int x3d_veccopy(x3d_backend *b, real_t *dst, const real_t *src)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && dst && src, "x3d_veccopy: null argument");
    if (x3d_lazy_active(b)) return x3d_lazy_copy(b, dst, src);  // (recorded only after the eager path's checks)
    ProfScope ps(b, X3D_K_COPY);
    X3D_HIP(hipMemcpyAsync(dst, src, sizeof(real_t) * b->nblock, hipMemcpyDeviceToDevice, b->stream));
    return 0;
}

extern "C" int x3d_vecadd(x3d_backend *b, real_t a, const real_t *x, real_t bb, real_t *y)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && x && y, "x3d_vecadd: null argument");
    if (x3d_lazy_active(b)) return x3d_lazy_vecadd(b, a, x, bb, y);
    ProfScope ps(b, X3D_K_BLAS1);
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_map2<OpAxpby>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)y,
                       (const real2_t *)x, n2, OpAxpby{a, bb});
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_vecmult(x3d_backend *b, real_t *y, const real_t *x)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && x && y, "x3d_vecmult: null argument");
    if (x3d_lazy_active(b)) return x3d_lazy_unary(b, 0, y, x, 0.0);
    ProfScope ps(b, X3D_K_BLAS1);
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_map2<OpMul>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)y,
                       (const real2_t *)x, n2, OpMul{});
    X3D_HIP(hipGetLastError());
    return 0;
}

// compute_vorticity / compute_qcriterion (src/backend/omp/backend.f90:616-649): pointwise functions of the nine
// velocity gradients, whole padded blocks like the reference.  g = {dudx, dudy, dudz, dvdx, dvdy, dvdz, dwdx, dwdy, dwdz}
struct Grad9 { const real2_t *g[9]; };
template <bool QCRIT>
__global__ void __launch_bounds__(256) k_from_gradients(real2_t *__restrict__ out, Grad9 G, size_t n2)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += st) {
        real2_t v[9];
#pragma unroll
        for (int m = 0; m < 9; m++) v[m] = G.g[m][i];
        auto f = [](real_t dudx, real_t dudy, real_t dudz, real_t dvdx, real_t dvdy, real_t dvdz, real_t dwdx,
                    real_t dwdy, real_t dwdz) -> real_t {
            if (QCRIT)
                return -0.5 * (dudx * dudx + dvdy * dvdy + dwdz * dwdz) - dudy * dvdx - dudz * dwdx - dvdz * dwdy;
            return sqrt((dwdy - dvdz) * (dwdy - dvdz) + (dudz - dwdx) * (dudz - dwdx) +
                        (dvdx - dudy) * (dvdx - dudy));
        };
        out[i] = make_real2(f(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x, v[8].x),
                              f(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y, v[8].y));
    }
}

static int from_gradients(x3d_backend *b, real_t *out, const real_t *const g[9], bool qcrit)
{
    X3D_REQUIRE(b && out && g, "derive_field_from_gradients: null argument");
    Grad9 G;
    for (int m = 0; m < 9; m++) {
        X3D_REQUIRE(g[m] && g[m] != out, "derive_field_from_gradients: bad gradient block %d", m);
        G.g[m] = (const real2_t *)g[m];
    }
    ProfScope ps(b, X3D_K_BLAS1);
    const size_t n2 = b->nblock / 2;
    if (qcrit) hipLaunchKernelGGL(k_from_gradients<true>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)out, G, n2);
    else hipLaunchKernelGGL(k_from_gradients<false>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)out, G, n2);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_compute_vorticity(x3d_backend *b, real_t *out, const real_t *const grads[9])
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    return from_gradients(b, out, grads, false);
}

extern "C" int x3d_compute_qcriterion(x3d_backend *b, real_t *out, const real_t *const grads[9])
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    return from_gradients(b, out, grads, true);
}

extern "C" int x3d_field_scale(x3d_backend *b, real_t *f, real_t a)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_field_scale: null argument");
    if (x3d_lazy_active(b)) return x3d_lazy_unary(b, 1, f, nullptr, a);
    ProfScope ps(b, X3D_K_BLAS1);
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_map1<OpScale>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)f, n2,
                       OpScale{a});
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_field_shift(x3d_backend *b, real_t *f, real_t a)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_field_shift: null argument");
    if (x3d_lazy_active(b)) return x3d_lazy_unary(b, 2, f, nullptr, a);
    ProfScope ps(b, X3D_K_BLAS1);
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_map1<OpShift>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)f, n2,
                       OpShift{a});
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_block_fill(x3d_backend *b, real_t *f, real_t c)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_block_fill: null argument");
    if (x3d_lazy_active(b)) return x3d_lazy_unary(b, 3, f, nullptr, c);
    ProfScope ps(b, X3D_K_BLAS1);
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_map1<OpFill>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)f, n2,
                       OpFill{c});
    X3D_HIP(hipGetLastError());
    return 0;
}

// reorder: every DIR_* tag shares one physical layout -> a device copy.
// (reference: src/backend/omp/backend.f90:393-452; codes src/common.f90:23-26)
extern "C" int x3d_reorder(x3d_backend *b, real_t *u_, const real_t *u, int rdr)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && u_ && u, "x3d_reorder: null argument");
    int from = rdr / 10, to = rdr % 10;
    X3D_REQUIRE(from >= 1 && from <= 4 && to >= 1 && to <= 4 && from != to,
                "x3d_reorder: invalid reorder code %d", rdr);
    if (x3d_lazy_active(b)) return x3d_lazy_copy(b, u_, u);
    if (u_ == u) return 0;
    ProfScope ps(b, X3D_K_COPY);
    X3D_HIP(hipMemcpyAsync(u_, u, sizeof(real_t) * b->nblock, hipMemcpyDeviceToDevice, b->stream));
    return 0;
}

// sum_yintox / sum_zintox: u += u_ (src/backend/omp/backend.f90:454-527)
extern "C" int x3d_sum_intox(x3d_backend *b, real_t *u, const real_t *u_, int dir_from)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && u && u_, "x3d_sum_intox: null argument");
    X3D_REQUIRE(dir_from == X3D_DIR_Y || dir_from == X3D_DIR_Z, "x3d_sum_intox: dir must be Y or Z");
    if (x3d_lazy_active(b)) return x3d_lazy_sum(b, u, u_, dir_from);
    ProfScope ps(b, X3D_K_BLAS1);
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_map2<OpAdd>, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)u,
                       (const real2_t *)u_, n2, OpAdd{});
    X3D_HIP(hipGetLastError());
    return 0;
}

// y = base + sum_i c[i] * x[i]  (one pass instead of a veccopy/vecadd chain)
struct LinArgs {
    const real2_t *x[5];
    real_t c[5];
    int n;
};

__global__ void __launch_bounds__(256) k_lincomb(real2_t *__restrict__ y, const real2_t *base, size_t n2,
                                                 LinArgs a)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += st) {
        // streamed once: nontemporal (real2_t has no builtin overload: two 8-byte accesses fuse back to 16)
        const real_t *bp = reinterpret_cast<const real_t *>(base + i);
        real2_t r = make_real2(__builtin_nontemporal_load(bp), __builtin_nontemporal_load(bp + 1));
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (k < a.n) {
                const real_t *xp = reinterpret_cast<const real_t *>(a.x[k] + i);
                const real_t vx = __builtin_nontemporal_load(xp), vy = __builtin_nontemporal_load(xp + 1);
                r.x = a.c[k] * vx + r.x;
                r.y = a.c[k] * vy + r.y;
            }
        real_t *yp = reinterpret_cast<real_t *>(y + i);
        __builtin_nontemporal_store(r.x, yp);
        __builtin_nontemporal_store(r.y, yp + 1);
    }
}

extern "C" int x3d_lincomb(x3d_backend *b, real_t *y, const real_t *base, int nterm, const real_t *c,
                           const real_t *const *x)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_SYNC(b);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && y && base && c && x, "x3d_lincomb: null argument");
    X3D_REQUIRE(nterm >= 1 && nterm <= 5, "x3d_lincomb: nterm must be 1..5");
    ProfScope ps(b, X3D_K_BLAS1);
    LinArgs a;
    a.n = nterm;
    for (int k = 0; k < 5; k++) {
        a.x[k] = (const real2_t *)(k < nterm ? x[k] : x[0]);
        a.c[k] = k < nterm ? c[k] : 0.0;
    }
    size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_lincomb, dim3(stream_grid(n2)), dim3(256), 0, b->stream, (real2_t *)y,
                       (const real2_t *)base, n2, a);
    X3D_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- reductions
// Two-stage, deterministic: per-workgroup partials in a fixed order, final
// sum on the host in index order (no atomics -> bitwise reproducible).
enum { RED_DOT = 0, RED_ABS = 1, RED_SUM = 2 };

template <int MODE>
__global__ void __launch_bounds__(256) k_reduce(const real_t *__restrict__ x, const real_t *__restrict__ y,
                                                int nx, int ny, int nz, long nxp, long nyp,
                                                real_t *__restrict__ part_sum,
                                                real_t *__restrict__ part_max)
{
    __shared__ real_t ssum[4], smax[4];
    real_t s = 0.0, m = 0.0;
    const long nrow = (long)ny * nz;
    for (long r = blockIdx.x; r < nrow; r += gridDim.x) {
        const long j = r % ny, k = r / ny;
        const long off = nxp * (j + nyp * k);
        for (int i = threadIdx.x; i < nx; i += blockDim.x) {
            real_t v = x[off + i];
            if (MODE == RED_DOT) s += v * y[off + i];
            else if (MODE == RED_ABS) { v = fabs(v); s += v; m = fmax(m, v); }
            else s += v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        m = fmax(m, __shfl_down(m, o));
    }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) { ssum[wv] = s; smax[wv] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        part_max[blockIdx.x] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}

template <int MODE>
static int run_reduce(x3d_backend *b, const real_t *x, const real_t *y, const int dims[3], real_t *sum,
                      real_t *mx)
{
    X3D_REQUIRE(dims[0] > 0 && dims[0] <= b->nxp && dims[1] > 0 && dims[1] <= b->nyp && dims[2] > 0 &&
                    dims[2] <= b->nzp,
                "reduction: dims (%d,%d,%d) outside the block", dims[0], dims[1], dims[2]);
    long nrow = (long)dims[1] * dims[2];
    int grid = (int)(nrow < 2048 ? nrow : 2048);
    ProfScope ps(b, X3D_K_REDUCE);
    hipLaunchKernelGGL(k_reduce<MODE>, dim3(grid), dim3(256), 0, b->stream, x, y, dims[0], dims[1], dims[2],
                       (long)b->nxp, (long)b->nyp, b->red_buf, b->red_buf + b->red_cap);
    X3D_HIP(hipGetLastError());
    X3D_HIP(hipMemcpyAsync(b->red_host, b->red_buf, sizeof(real_t) * 2 * b->red_cap, hipMemcpyDeviceToHost,
                           b->stream));
    X3D_HIP(hipStreamSynchronize(b->stream));
    real_t s = 0.0, m = 0.0;
    for (int i = 0; i < grid; i++) {
        s += b->red_host[i];
        m = fmax(m, b->red_host[b->red_cap + i]);
    }
    if (sum) *sum = s;
    if (mx) *mx = m;
    return 0;
}

extern "C" int x3d_scalar_product(x3d_backend *b, const real_t *x, const real_t *y, const int dims[3],
                                  real_t *out)
{
    X3D_RANGE(__func__);
    if (b) { X3D_LAZY_IN(b, x); X3D_LAZY_IN(b, y); }
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && x && y && dims && out, "x3d_scalar_product: null argument");
    return run_reduce<RED_DOT>(b, x, y, dims, out, nullptr);
}

extern "C" int x3d_field_max_sum(x3d_backend *b, const real_t *f, const int dims[3], real_t *max_abs,
                                 real_t *sum_abs)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_IN(b, f);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && dims && max_abs && sum_abs, "x3d_field_max_sum: null argument");
    return run_reduce<RED_ABS>(b, f, f, dims, sum_abs, max_abs);
}

extern "C" int x3d_field_volume_integral(x3d_backend *b, const real_t *f, const int dims[3], real_t *out)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_IN(b, f);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && dims && out, "x3d_field_volume_integral: null argument");
    return run_reduce<RED_SUM>(b, f, f, dims, out, nullptr);
}

// ---------------------------------------------------------------- channel case, per sub-step, without the host
// define_BC_channel (src/case/channel.f90:59-130) does, per sub-step: ub = field_volume_integral(u) / ncell
// (+ MPI_Allreduce), field_shift(u, 2/3 - ub), and -- on the host -- 6 random_number planes scaled to
// noise * (2 r - 1), uploaded as 3 full blocks.  On one rank nothing of that needs the host:
//   x3d_field_shift_to_mean: the partial sums stay on the device, one thread adds them in the order the host
//     would (bit-identical to x3d_field_volume_integral + x3d_field_shift), the shift kernel reads the result;
//   x3d_wall_noise: the two wall planes are generated in place by a counter-based generator.
__global__ void k_finish_shift(const real_t *__restrict__ part, int n, real_t ncell, real_t target,
                               real_t *__restrict__ out)
{
    // (the partials come in through LDS with all lanes: one thread chasing 2048 dependent global loads took 0.12 ms)
    __shared__ real_t sp[2048];
    for (int i = threadIdx.x; i < n; i += blockDim.x) sp[i] = part[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        real_t s = 0.0;
        for (int i = 0; i < n; i++) s += sp[i];    // the order of run_reduce's host loop
        out[0] = target - s / ncell;               // can = 2/3 - ub, src/case/channel.f90:70-77
        out[1] = s;
    }
}

__global__ void __launch_bounds__(256) k_shift_dev(real2_t *__restrict__ f, size_t n2, const real_t *__restrict__ a)
{
    const real_t sh = a[0];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        real2_t v = f[i];
        v.x += sh; v.y += sh;
        f[i] = v;
    }
}

// the two halves of x3d_field_shift_to_mean: *shift = device address of target - volume_integral(f) / ncell (valid
// until the backend's next reduction); f += that device scalar
// (the entry points below translate their field ONCE -- a translated pointer must not be translated again: the buffer that
//  holds a handle's data is usually another handle's own block -- and run at once: X3D_LAZY_IN / _OUT, no copies; a queue
//  flushed through X3D_LAZY_SYNC here cost the channel case five block copies per sub-step, round 4)
static int mean_shift_impl(x3d_backend *b, const real_t *f, const int dims[3], real_t ncell, real_t target,
                           const real_t **shift)
{
    X3D_REQUIRE(b && f && dims && ncell > 0.0 && shift, "x3d_field_mean_shift: bad argument");
    X3D_REQUIRE(dims[0] > 0 && dims[0] <= b->nxp && dims[1] > 0 && dims[1] <= b->nyp && dims[2] > 0 && dims[2] <= b->nzp,
                "x3d_field_mean_shift: dims outside the block");
    const long nrow = (long)dims[1] * dims[2];
    const int grid = (int)(nrow < 2048 ? nrow : 2048);
    ProfScope ps(b, X3D_K_REDUCE);
    hipLaunchKernelGGL(k_reduce<RED_SUM>, dim3(grid), dim3(256), 0, b->stream, f, f, dims[0], dims[1], dims[2],
                       (long)b->nxp, (long)b->nyp, b->red_buf, b->red_buf + b->red_cap);
    hipLaunchKernelGGL(k_finish_shift, dim3(1), dim3(256), 0, b->stream, (const real_t *)b->red_buf, grid, ncell, target,
                       b->red_buf + 2 * b->red_cap - 2);
    X3D_HIP(hipGetLastError());
    *shift = b->red_buf + 2 * b->red_cap - 2;
    return 0;
}

// the second stage alone: `nparts` partial sums already sit in b->red_buf (k_xwide_tds_lin)
int x3d_finish_mean_shift(x3d_backend *b, int nparts, real_t ncell, real_t target, const real_t **shift)
{
    X3D_REQUIRE(b && shift && nparts > 0 && nparts <= 2048 && nparts <= b->red_cap, "x3d_finish_mean_shift: %d partials", nparts);
    ProfScope ps(b, X3D_K_REDUCE);
    hipLaunchKernelGGL(k_finish_shift, dim3(1), dim3(256), 0, b->stream, (const real_t *)b->red_buf, nparts, ncell, target,
                       b->red_buf + 2 * b->red_cap - 2);
    X3D_HIP(hipGetLastError());
    *shift = b->red_buf + 2 * b->red_cap - 2;
    return 0;
}

extern "C" int x3d_field_mean_shift(x3d_backend *b, const real_t *f, const int dims[3], real_t ncell, real_t target,
                                    const real_t **shift)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_field_mean_shift: bad argument");
    X3D_LAZY_IN(b, f);
    X3D_LAZY_EAGER(b);
    return mean_shift_impl(b, f, dims, ncell, target, shift);
}

static int shift_by_impl(x3d_backend *b, real_t *f, const real_t *shift)
{
    X3D_REQUIRE(b && f && shift, "x3d_field_shift_by: null argument");
    ProfScope ps(b, X3D_K_BLAS1);
    const size_t n2 = b->nblock / 2;
    hipLaunchKernelGGL(k_shift_dev, dim3(2048), dim3(256), 0, b->stream, (real2_t *)f, n2, shift);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_field_shift_by(x3d_backend *b, real_t *f, const real_t *shift)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_field_shift_by: null argument");
    X3D_LAZY_OUT(b, f, false);
    X3D_LAZY_EAGER(b);
    return shift_by_impl(b, f, shift);
}

extern "C" int x3d_field_shift_to_mean(x3d_backend *b, real_t *f, const int dims[3], real_t ncell, real_t target)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_field_shift_to_mean: bad argument");
    X3D_LAZY_OUT(b, f, false);
    X3D_LAZY_EAGER(b);
    const real_t *shift = nullptr;
    if (int rc = mean_shift_impl(b, f, dims, ncell, target, &shift)) return rc;
    return shift_by_impl(b, f, shift);
}

// splitmix64 of (seed, counter): the value depends on its inputs only, not on the launch geometry
__host__ __device__ inline unsigned long long x3d_mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// planes y = 1 and y = ny of f <- amp * (2 r - 1), r uniform in [0, 1) with 53 random bits:
// r(face, i, k) = mix64(mix64(seed + draw) + (face * nz + k) * nx + i) >> 11) * 2^-53
__global__ void __launch_bounds__(256) k_wall_noise(real_t *__restrict__ f, int nx, int ny, int nz, long nxp, long nyp,
                                                    real_t amp, unsigned long long key)
{
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= 2L * nx * nz) return;
    const int face = (int)(q / ((long)nx * nz));
    const long r = q - (long)face * nx * nz;
    const int i = (int)(r % nx), k = (int)(r / nx);
    const real_t u01 = (real_t)(x3d_mix64(key + (unsigned long long)q) >> 11) * 0x1.0p-53;
    f[i + nxp * ((face ? ny - 1 : 0) + nyp * (long)k)] = amp * (2.0 * u01 - 1.0);
}

extern "C" int x3d_wall_noise(x3d_backend *b, real_t *f, const int dims[3], real_t amp, unsigned long long seed,
                              unsigned long long draw)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f && dims, "x3d_wall_noise: null argument");
    X3D_LAZY_OUT(b, f, false);  // (two planes are written: the rest of the block keeps its contents)
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(dims[0] > 0 && dims[0] <= b->nxp && dims[1] > 1 && dims[1] <= b->nyp && dims[2] > 0 && dims[2] <= b->nzp,
                "x3d_wall_noise: dims outside the block");
    const long n = 2L * dims[0] * dims[2];
    ProfScope ps(b, X3D_K_COPY);
    hipLaunchKernelGGL(k_wall_noise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, f, dims[0], dims[1],
                       dims[2], (long)b->nxp, (long)b->nyp, amp, x3d_mix64(seed + draw));
    X3D_HIP(hipGetLastError());
    return 0;
}

// slice_max_sum: signed max and sum over the plane i_slice of direction dir
__global__ void __launch_bounds__(256) k_slice(const real_t *__restrict__ f, int n0, int n1, long s0, long s1,
                                               long off, real_t *__restrict__ part_sum,
                                               real_t *__restrict__ part_max)
{
    __shared__ real_t ssum[4], smax[4];
    real_t s = 0.0, m = -HUGE_VAL;
    const long n = (long)n0 * n1;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        real_t v = f[off + (q % n0) * s0 + (q / n0) * s1];
        s += v;
        m = fmax(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        m = fmax(m, __shfl_down(m, o));
    }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) { ssum[wv] = s; smax[wv] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        part_max[blockIdx.x] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}

extern "C" int x3d_slice_max_sum(x3d_backend *b, const real_t *f, const int dims[3], int dir, int i_slice,
                                 real_t *max_val, real_t *sum_val)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_IN(b, f);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && dims && max_val && sum_val, "x3d_slice_max_sum: null argument");
    X3D_REQUIRE(x3d_dir_ok(dir), "slice_max_sum does not support DIR_C fields!");
    X3D_REQUIRE(i_slice >= 1 && i_slice <= dims[dir - 1], "slice_max_sum: i_slice out of range");
    const long nxp = b->nxp, nyp = b->nyp;
    int n0, n1; long s0, s1, off;
    if (dir == X3D_DIR_X) { n0 = dims[1]; n1 = dims[2]; s0 = nxp; s1 = nxp * nyp; off = i_slice - 1; }
    else if (dir == X3D_DIR_Y) { n0 = dims[0]; n1 = dims[2]; s0 = 1; s1 = nxp * nyp; off = nxp * (i_slice - 1); }
    else { n0 = dims[0]; n1 = dims[1]; s0 = 1; s1 = nxp; off = nxp * nyp * (i_slice - 1); }
    long n = (long)n0 * n1;
    int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_slice, dim3(grid), dim3(256), 0, b->stream, f, n0, n1, s0, s1, off, b->red_buf,
                       b->red_buf + b->red_cap);
    X3D_HIP(hipGetLastError());
    X3D_HIP(hipMemcpyAsync(b->red_host, b->red_buf, sizeof(real_t) * 2 * b->red_cap, hipMemcpyDeviceToHost,
                           b->stream));
    X3D_HIP(hipStreamSynchronize(b->stream));
    real_t s = 0.0, m = -HUGE_VAL;
    for (int i = 0; i < grid; i++) {
        s += b->red_host[i];
        m = fmax(m, b->red_host[b->red_cap + i]);
    }
    *max_val = m;
    *sum_val = s;
    return 0;
}

// ---------------------------------------------------------------- faces
// field_set_face(Y_FACE): plane y=1 <- c_start, plane y=ny <- c_end
// (src/backend/omp/backend.f90:903-952; only Y_FACE is supported there).
__global__ void k_set_face_y(real_t *__restrict__ f, const real_t *__restrict__ src, int nx, int ny, int nz,
                             long nxp, long nyp, real_t c_start, real_t c_end, int from_field)
{
    long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= (long)nx * nz) return;
    long i = q % nx, k = q / nx;
    long lo = i + nxp * (0 + nyp * k), hi = i + nxp * ((ny - 1) + nyp * k);
    f[lo] = from_field ? src[lo] : c_start;
    f[hi] = from_field ? src[hi] : c_end;
}

// field_set_face_from_field(X_FACE): inflow plane from f_start, convective
// outflow at x = nx (src/backend/omp/backend.f90:978-1003)
__global__ void k_set_face_x_from(real_t *__restrict__ f, const real_t *__restrict__ src, int nx, int ny,
                                  int nz, long nxp, long nyp, real_t c_end, real_t frd)
{
    long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= (long)ny * nz) return;
    long j = q % ny, k = q / ny;
    long row = nxp * (j + nyp * k);
    f[row] = src[row];
    real_t fd = f[row + nx - 1], fd1 = f[row + nx - 2];
    f[row + nx - 1] = fd - c_end * (fd - fd1) + frd;
}

extern "C" int x3d_field_set_face(x3d_backend *b, real_t *f, const int dims[3], real_t c_start, real_t c_end,
                                  int face)
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_OUT(b, f, false);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && dims, "x3d_field_set_face: null argument");
    X3D_REQUIRE(face != X3D_X_FACE, "Setting X_FACE is not yet supported.");
    X3D_REQUIRE(face != X3D_Z_FACE, "Setting Z_FACE is not yet supported.");
    X3D_REQUIRE(face == X3D_Y_FACE, "face is undefined.");
    long n = (long)dims[0] * dims[2];
    hipLaunchKernelGGL(k_set_face_y, dim3((n + 255) / 256), dim3(256), 0, b->stream, f, (const real_t *)nullptr,
                       dims[0], dims[1], dims[2], (long)b->nxp, (long)b->nyp, c_start, c_end, 0);
    X3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int x3d_field_set_face_from_field(x3d_backend *b, real_t *f, const real_t *f_start, const int dims[3],
                                             real_t c_end, int face, real_t flow_rate_diff)
{
    X3D_RANGE(__func__);
    if (b && x3d_lazy_active(b) && face == X3D_Y_FACE) {  // recorded (the RK stage before it and the x operator behind it fuse)
        X3D_REQUIRE(f && f_start && dims, "x3d_field_set_face_from_field: null argument");
        X3D_REQUIRE(dims[0] > 0 && dims[0] <= b->nxp && dims[1] > 0 && dims[1] <= b->nyp && dims[2] > 0 && dims[2] <= b->nzp,
                    "x3d_field_set_face_from_field: dims outside the block");
        return x3d_lazy_setface(b, f, f_start, dims);
    }
    if (b) { X3D_LAZY_IN(b, f_start); X3D_LAZY_OUT(b, f, false); }
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && f_start && dims, "x3d_field_set_face_from_field: null argument");
    if (face == X3D_Y_FACE) {
        long n = (long)dims[0] * dims[2];
        hipLaunchKernelGGL(k_set_face_y, dim3((n + 255) / 256), dim3(256), 0, b->stream, f, f_start, dims[0],
                           dims[1], dims[2], (long)b->nxp, (long)b->nyp, 0.0, 0.0, 1);
    } else if (face == X3D_X_FACE) {
        long n = (long)dims[1] * dims[2];
        hipLaunchKernelGGL(k_set_face_x_from, dim3((n + 255) / 256), dim3(256), 0, b->stream, f, f_start,
                           dims[0], dims[1], dims[2], (long)b->nxp, (long)b->nyp, c_end, flow_rate_diff);
    } else {
        X3D_REQUIRE(false, "field_set_face_from_field: only X_FACE and Y_FACE supported.");
    }
    X3D_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- host <-> field
extern "C" int x3d_set_field_data(x3d_backend *b, real_t *f, const real_t *host, const int dims[3])
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_OUT(b, f, false);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(dims, "x3d_set_field_data: null argument");
    return x3d_set_field_data_pitched(b, f, host, dims[0], dims[1], dims);
}

extern "C" int x3d_get_field_data(x3d_backend *b, real_t *host, const real_t *f, const int dims[3])
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_IN(b, f);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(dims, "x3d_get_field_data: null argument");
    return x3d_get_field_data_pitched(b, host, f, dims[0], dims[1], dims);
}

extern "C" int x3d_set_field_data_pitched(x3d_backend *b, real_t *f, const real_t *host, int hx, int hy,
                                          const int dims[3])
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_OUT(b, f, false);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && host && dims, "x3d_set_field_data: null argument");
    X3D_REQUIRE(dims[0] <= b->nxp && dims[1] <= b->nyp && dims[2] <= b->nzp && dims[0] <= hx && dims[1] <= hy,
                "x3d_set_field_data: dims exceed the block");
    hipMemcpy3DParms p;
    memset(&p, 0, sizeof p);
    p.srcPtr = make_hipPitchedPtr((void *)host, sizeof(real_t) * hx, hx, hy);
    p.dstPtr = make_hipPitchedPtr((void *)f, sizeof(real_t) * b->nxp, b->nxp, b->nyp);
    p.extent = make_hipExtent(sizeof(real_t) * dims[0], dims[1], dims[2]);
    p.kind = hipMemcpyHostToDevice;
    X3D_HIP(hipMemcpy3DAsync(&p, b->stream));
    X3D_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int x3d_get_field_data_pitched(x3d_backend *b, real_t *host, const real_t *f, int hx, int hy,
                                          const int dims[3])
{
    X3D_RANGE(__func__);
    if (b) X3D_LAZY_IN(b, f);
    X3D_LAZY_EAGER(b);
    X3D_REQUIRE(b && f && host && dims, "x3d_get_field_data: null argument");
    X3D_REQUIRE(dims[0] <= b->nxp && dims[1] <= b->nyp && dims[2] <= b->nzp && dims[0] <= hx && dims[1] <= hy,
                "x3d_get_field_data: dims exceed the block");
    hipMemcpy3DParms p;
    memset(&p, 0, sizeof p);
    p.srcPtr = make_hipPitchedPtr((void *)f, sizeof(real_t) * b->nxp, b->nxp, b->nyp);
    p.dstPtr = make_hipPitchedPtr((void *)host, sizeof(real_t) * hx, hx, hy);
    p.extent = make_hipExtent(sizeof(real_t) * dims[0], dims[1], dims[2]);
    p.kind = hipMemcpyDeviceToHost;
    X3D_HIP(hipMemcpy3DAsync(&p, b->stream));
    X3D_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

// ---------------------------------------------------------------- timing
extern "C" int x3d_timer_start(x3d_backend *b)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "null backend");
    X3D_HIP(hipEventRecord(b->ev0, b->stream));
    return 0;
}

extern "C" int x3d_timer_stop_ms(x3d_backend *b, float *ms)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && ms, "null argument");
    X3D_HIP(hipEventRecord(b->ev1, b->stream));
    X3D_HIP(hipEventSynchronize(b->ev1));
    X3D_HIP(hipEventElapsedTime(ms, b->ev0, b->ev1));
    return 0;
}
