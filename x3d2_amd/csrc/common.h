// Internal definitions shared by the HIP translation units of libx3d2_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/x3d2_hip.h"

// the real kind of every field, table, scalar and transform (x3d_real of the public header): FP64, or FP32 with
// -DX3D_SINGLE_PREC (the reference's -DSINGLE_PREC, src/common.f90:6-12)
typedef x3d_real real_t;
#ifdef X3D_SINGLE_PREC
typedef float2 real2_t;
#define make_real2 make_float2
#else
typedef double2 real2_t;
#define make_real2 make_double2
#endif
// hipFFT's names for the real kind's transforms (the Poisson solvers' rocFFT plans)
#ifdef X3D_SINGLE_PREC
#define X3D_FFT_R2C HIPFFT_R2C
#define X3D_FFT_C2R HIPFFT_C2R
#define X3D_FFT_C2C HIPFFT_C2C
#define x3d_fftExecR2C hipfftExecR2C
#define x3d_fftExecC2R hipfftExecC2R
#define x3d_fftExecC2C hipfftExecC2C
#define x3d_fft_real hipfftReal
#define x3d_fft_cplx hipfftComplex
#else
#define X3D_FFT_R2C HIPFFT_D2Z
#define X3D_FFT_C2R HIPFFT_Z2D
#define X3D_FFT_C2C HIPFFT_Z2Z
#define x3d_fftExecR2C hipfftExecD2Z
#define x3d_fftExecC2R hipfftExecZ2D
#define x3d_fftExecC2C hipfftExecZ2Z
#define x3d_fft_real hipfftDoubleReal
#define x3d_fft_cplx hipfftDoubleComplex
#endif
typedef double x3d_f64;                // (where FP64 is meant whatever the real kind: twiddle generation, timers)
#define X3D_RB ((int)sizeof(real_t))  // bytes per real

#define X3D_NH 4

void x3d_set_error(const char *fmt, ...);

#define X3D_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            x3d_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                          __LINE__);                                                           \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

#define X3D_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            x3d_set_error(__VA_ARGS__);                                                        \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)

// a * b + c with ONE rounding, spelled out.  -ffp-contract=fast leaves it to the compiler WHICH product of
// `x * y + u * v` it fuses (the operand order of the addition decides, and that order depends on the code around it): two
// kernels that must give the same bits from the same source expression spell the choice out where both terms are products
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Between the store and the load phase of a WAVE-PRIVATE exchange through LDS: a wave's LDS operations execute
// in order, so no s_barrier is needed, but the compiler must not move one lane's load over another lane's store
// (plain C++ semantics would allow it).  Emits no instruction.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Streaming accesses (round 4).  A flat copy on this chip reaches 6.2 TB/s only with NONTEMPORAL 16-byte loads and
// stores (5.6 with plain ones: profiles/r04_copy_ceiling.txt); every field row of the derivative kernels is read once
// and written once per launch, so their global accesses carry the hint.  The builtin has no overload for HIP's real2_t
// struct -- through an ext_vector_type the access stays ONE global_load_dwordx4 ... nt (two 8-byte builtins do not
// always fuse back).  -DX3D_NO_NT: plain accesses (A/B builds).
typedef real_t x3d_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ real2_t ldg_stream(const real2_t *p)
{
#ifndef X3D_NO_NT
    const x3d_d2v v = __builtin_nontemporal_load(reinterpret_cast<const x3d_d2v *>(p));
    return make_real2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stg_stream(real2_t *p, real2_t v)
{
#ifndef X3D_NO_NT
    const x3d_d2v w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<x3d_d2v *>(p));
#else
    *p = v;
#endif
}

// Store acknowledgements and loop-carried prefetches (round 4).  On gfx950 loads and stores share ONE in-order counter
// (vmcnt).  A persistent kernel that prefetches its next input before it stores its current result has, at the top of
// the next iteration, [prefetch loads][later loads][stores] outstanding, and needs only the prefetch -- the hardware
// can wait for exactly that (s_waitcnt vmcnt(number of later operations)).  The compiler's wait-count pass however
// merges the loop's two entries: on the way in from the prologue the prefetch loads are the LAST operations, so the
// merged state says "wait for everything", and every iteration starts by waiting for the previous iteration's stores
// to be acknowledged by memory (seen in the ISA as s_waitcnt vmcnt(0) at the loop top).  Issuing as many (tiny) stores
// in the prologue as the loop body issues operations behind its prefetch makes both entries look alike: the wait
// becomes vmcnt(n) and the stores drain behind the next iteration's work.  Performance only: the values land in a
// sink nobody reads.
#ifndef X3D_NO_VMCNT_PAD
static __device__ __attribute__((used)) float g_vmcnt_sink[32 * 1024];
template <int N>
__device__ __forceinline__ void vmcnt_pad_stores()
{
#pragma unroll
    for (int k = 0; k < N; k++) __builtin_nontemporal_store(0.0f, &g_vmcnt_sink[k * 1024 + (threadIdx.x & 1023)]);
}
#else
template <int N> __device__ __forceinline__ void vmcnt_pad_stores() {}
#endif

// Pencil enumeration for one direction of the Cartesian-pitched block:
// pencil p -> base = (p % dim0) * s0 + (p / dim0) * s1, rows advance by rs.
struct PencilGeom {
    int np;    // number of pencils (real, unpadded cross-section)
    int dim0;  // extent of the fastest cross-section index
    long s0, s1, rs;
};

struct x3d_backend {
    int device;
    hipStream_t stream;
    int nx, ny, nz;     // local vertex dims
    int ring[4];        // [dir]: the decomposed direction is PERIODIC over all its ranks (x3d_backend_set_ring): the HALO forms
                        // may then run the open-ended circulant solve -- every rank of the ring must make the same choice
    int nxp, nyp, nzp;  // pitched dims
    size_t nblock;      // elements per block
    // boundary-value exchange buffers for the local (non-decomposed) forms:
    // [3 ops][npencil_max] each
    real_t *send_s, *send_e;
    // two scratch blocks for the transeq intermediates (dud, d2u):
    // transeq_dist_component gets the same two from the pool,
    // src/backend/omp/backend.f90:319-320
    real_t *scratch[3];  // + slack of 64 rows: x-direction kernels keep them wave-transposed
    real_t *red_buf;  // reduction partials (device)
    real_t *red_host; // pinned host landing zone
    int red_cap;
    long n_upd;       // ... of those, launches that also applied the pending velocity correction (UPD form)
    long n_tq3;       // launches of the three-components-in-one transeq kernels (bench.py prices them at 48 B/DoF)
    long n_halo;      // launches of the HALO forms of the tile kernels (a decomposed direction in one pass)
    void *epi_dev;    // 512-byte device slot for the RK-stage descriptions of k_ytile_transeq<EPI> / k_ytile_transeq3<EPI> (xscan.hip)
    hipEvent_t ev0, ev1;
    void *ipc_maps;         // peers' buffers mapped by x3d_ipc_open and not closed yet (std::vector<void *>; closed by x3d_backend_destroy)
    struct x3d_prof *prof;  // per-kernel HIP-event timers (prof.hip), null until enabled
    unsigned prof_mask;     // kernel classes that are timed while the timers are on (bit = X3D_K_*; x3d_prof_select)
    int pair_yperm;          // > 0 during x3d_tds_solve_pair_yperm: the z pair kernels permute that many y rows
    void *lds_optin;        // kernels of this backend's device whose dynamic-LDS limit has been raised (backend.hip)
    struct x3d_lazy *lazy;  // deferred execution of the op-granular call sequence (lazy.hip), null until used
    // round 5: CUs the PERSISTENT kernels leave free (x3d_backend_set_comm_reserve).  The tile / scan kernels launch one
    // workgroup per CU that stays for the whole launch and owns all of its CU's registers and LDS: a kernel of another
    // stream -- RCCL's send / recv kernels of an exchange meant to run beside them -- finds no CU to start on until the
    // launch ENDS (measured with a one-wave kernel on the communication stream: none of its time hidden,
    // profiles/r05_yslab_pipeline_timeline.txt).  With fewer workgroups than CUs some CUs stay empty and the exchange
    // starts at once; the kernels are memory-bound, 248 CUs stream what 256 do.  0 on one rank.
    int comm_reserve;
};
// the RK / AB stage of one variable as the epilogue of a tile kernel (xscan.hip, k_ytile_transeq<EPI> / k_ytile_transeq3<EPI>):
// d = x[ipend] + component;  [store: x[ipend] = d;]  y = base + sum_k c[k] (k == ipend ? d : x[k])
struct TileEpi {
    real_t *y;
    const real_t *base;
    const real_t *x[5];
    real_t c[5];
    int n, ipend, store;
};
#define X3D_NCU 256  // MI355X
static inline int x3d_persistent_blocks(const x3d_backend *b, long want)
{
    const long cap = X3D_NCU - (b->comm_reserve > 0 && b->comm_reserve < X3D_NCU / 2 ? b->comm_reserve : 0);
    return (int)(want > cap ? cap : want);
}

// ---- deferred execution (lazy.hip).  While the mode is on, block addresses are HANDLES: an entry point either records
// its call (x3d_lazy_active + x3d_lazy_<op>), or runs at once on the buffers that hold the handles' data
// (X3D_LAZY_IN / X3D_LAZY_OUT: flush the queue, translate), or first restores the identity map (X3D_LAZY_SYNC).
bool x3d_lazy_active(const x3d_backend *b);
void x3d_lazy_destroy(x3d_backend *b);
void x3d_lazy_register(x3d_backend *b, real_t *h);
void x3d_lazy_unregister(x3d_backend *b, real_t *h);
int x3d_lazy_flush_c(x3d_backend *b);
int x3d_lazy_sync_c(x3d_backend *b);
int x3d_lazy_in(x3d_backend *b, const real_t *h, const real_t **out);
int x3d_lazy_out(x3d_backend *b, real_t *h, bool full, real_t **out);
int x3d_lazy_transeq(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u, const real_t *v,
                     const real_t *w, real_t nu, const x3d_tdsops *t0, const x3d_tdsops *t1, const x3d_tdsops *t2,
                     const x3d_tdsops *t3);
int x3d_lazy_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir);
int x3d_lazy_species(x3d_backend *b, int dir, real_t *dspec, const real_t *uvw, const real_t *spec, real_t nu,
                     const x3d_tdsops *t0, const x3d_tdsops *t1, const x3d_tdsops *t2, int accumulate);
int x3d_lazy_copy(x3d_backend *b, real_t *dst, const real_t *src);
int x3d_lazy_sum(x3d_backend *b, real_t *u, const real_t *u_, int dir);
int x3d_lazy_vecadd(x3d_backend *b, real_t a, const real_t *x, real_t bb, real_t *y);
int x3d_lazy_unary(x3d_backend *b, int kind, real_t *f, const real_t *x, real_t a);  // 0 vecmult, 1 scale, 2 shift, 3 fill
int x3d_lazy_fft(x3d_backend *b, int which, void *poisson, real_t *f);               // 0 forward, 1 postprocess_000, 2 backward
int x3d_lazy_setface(x3d_backend *b, real_t *f, const real_t *f_start, const int dims[3]);  // field_set_face_from_field(Y_FACE)
// an entry point that runs at once while the mode is on: nothing it calls may be recorded (or have its -- already
// translated -- pointers translated again) until it returns
bool x3d_lazy_set_executing(x3d_backend *b, bool on);  // returns the previous state
struct LazyScope {
    x3d_backend *b;
    bool prev;
    explicit LazyScope(x3d_backend *b_) : b(b_ && b_->lazy ? b_ : nullptr), prev(false)
    {
        if (b) prev = x3d_lazy_set_executing(b, true);
    }
    ~LazyScope() { if (b) x3d_lazy_set_executing(b, prev); }
    LazyScope(const LazyScope &) = delete;
    LazyScope &operator=(const LazyScope &) = delete;
};
#define X3D_LAZY_EAGER(b) LazyScope lazy_scope_((b))
#define X3D_LAZY_SYNC(b)                                                                       \
    do {                                                                                       \
        if ((b)->lazy)                                                                         \
            if (int rc_ = x3d_lazy_sync_c(b)) return rc_;                                      \
    } while (0)
#define X3D_LAZY_FLUSH(b)                                                                      \
    do {                                                                                       \
        if ((b)->lazy)                                                                         \
            if (int rc_ = x3d_lazy_flush_c(b)) return rc_;                                     \
    } while (0)
#define X3D_LAZY_IN(b, ptr)                                                                    \
    do {                                                                                       \
        if ((b)->lazy) {                                                                       \
            const real_t *q_ = nullptr;                                                        \
            if (int rc_ = x3d_lazy_in((b), (ptr), &q_)) return rc_;                            \
            (ptr) = q_;                                                                        \
        }                                                                                      \
    } while (0)
#define X3D_LAZY_OUT(b, ptr, full)                                                             \
    do {                                                                                       \
        if ((b)->lazy) {                                                                       \
            real_t *q_ = nullptr;                                                              \
            if (int rc_ = x3d_lazy_out((b), (ptr), (full), &q_)) return rc_;                   \
            (ptr) = q_;                                                                        \
        }                                                                                      \
    } while (0)

// > 64 KB of dynamic LDS needs an opt-in per kernel and DEVICE: raise the limit to the CU's 160 KB once per
// (backend, kernel) -- not a process-wide flag, and not the first call's size
int x3d_lds_optin(x3d_backend *b, const void *kernel);
#define X3D_LDS_OPTIN(b, kernel)                                                               \
    do {                                                                                       \
        if (int rc_ = x3d_lds_optin((b), (const void *)(kernel))) return rc_;                  \
    } while (0)

// kernel classes timed by the profiler; index = kind * 4 + dir (dir 0 = n/a)
// (X3D_K_* are declared in include/x3d2_hip.h)
#define X3D_K_NKINDS 10
void x3d_prof_begin(x3d_backend *b, int kind, int dir);
void x3d_prof_end(x3d_backend *b);
int x3d_prof_enable_c(x3d_backend *b, int on);
// ---- roctx ranges (round 6): every entry point of the C ABI -- one per deferred procedure of base_backend_t / poisson_fft_t
// plus the fused forms -- and every operation the deferred-execution layer issues pushes a named range, so that a
// `rocprofv3 --marker-trace --kernel-trace` timeline attributes kernels to the reference operation they serve (the reference
// has wall-clock prints only, src/case/base_case.f90:258-303).  Off unless X3D_ROCTX=1; the roctx library is opened at first
// use (no link-time dependency): librocprofiler-sdk-roctx.so, else libroctx64.so.  prof.hip
bool x3d_roctx_on();
void x3d_roctx_push(const char *name);
void x3d_roctx_pop();
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(x3d_roctx_on()) { if (on) x3d_roctx_push(name); }
    ~RoctxRange() { if (on) x3d_roctx_pop(); }
};
#define X3D_RANGE(name) RoctxRange x3d_range_scope_(name)

struct ProfScope {
    x3d_backend *b;
    bool on;  // false: the caller times a group of launches as one (scopes do not nest)
    ProfScope(x3d_backend *b_, int kind, int dir = 0, bool on_ = true)
        : b(b_), on(on_ && b_->prof && ((b_->prof_mask >> kind) & 1u))
    {
        if (on) x3d_prof_begin(b, kind, dir);
    }
    ~ProfScope() { if (on) x3d_prof_end(b); }
};

// Device-side view of one tdsops_t: row tables prepared on the host from the
// reference's arrays (see tdsops.hip).  All table pointers are device arrays
// indexed by the 1-based row j (entry 0 unused).
struct TdsTab {
    int n_tds, n_rhs;
    int chunk;          // rows per wave of the on-chip chunk-parallel solve (32 or 64)
    // Row tables are interleaved so that one wide scalar load fetches everything a
    // sweep needs for row j (1-based; record 0 unused):
    //   RF[4*j + {0,1,2,3}] = F, A, W, PF     forward sweep
    //   RB[8*j + {0..7}]    = Bw, Sa, Sc, St, Stc, QB, PF16, QB16   backward sweep / substitution
    // F  forward multiplier  (rows 1,2: dist_af; >=3: dist_fw)
    // A  forward coupling    (rows 1,2: 0; bulk: dist_af(5); else dist_af(j))
    // W  weights of d_k in du_2 (backward chain), see tds.hip
    // PF/QB chunk-local carry multipliers (onchip.hip)
    const real_t *RF, *RB;
    const real_t *TL;   // lane tables of the wave-per-pencil x kernels (xscan.hip): [entry][64 lanes], or null
    int Q;              // rows per lane there (4, 8; 16: only the compressed form is used), 0 if unavailable
    int bulk_only;      // 1: start/end stencils equal the bulk stencil (periodic / BC_HALO both ends)
    const real_t *Cs;   // [4][9] start stencils, then [4][9] end stencils, then [9] bulk
    real_t last_r;      // dist_fw(1)
    real_t bw1;         // dist_bw(1)
    real_t rs_s, rs_e;  // 1/(1 - sa(1)^2), 1/(1 - sc(n)^2)
    real_t sa1, scn;
};
#define T_F(t, j) ((t).RF[4 * (j) + 0])
#define T_A(t, j) ((t).RF[4 * (j) + 1])
#define T_W(t, j) ((t).RF[4 * (j) + 2])
#define T_PF(t, j) ((t).RF[4 * (j) + 3])
#define T_BW(t, j) ((t).RB[8 * (j) + 0])
#define T_SA(t, j) ((t).RB[8 * (j) + 1])
#define T_SC(t, j) ((t).RB[8 * (j) + 2])
#define T_ST(t, j) ((t).RB[8 * (j) + 3])
#define T_STC(t, j) ((t).RB[8 * (j) + 4])
#define T_QB(t, j) ((t).RB[8 * (j) + 5])
#define T_PF16(t, j) ((t).RB[8 * (j) + 6])  // carry multipliers for 16-row chunks
#define T_QB16(t, j) ((t).RB[8 * (j) + 7])

// A periodic operator on a uniform grid is the circulant system (alpha, 1, alpha) x = r: with rho the root of
// rho^2 - rho / alpha + 1 = 0 inside the unit circle it factors as (alpha / rho) (1 + rho z^-1) (1 + rho z), i.e. two
// constant-coefficient first-order recurrences (xscan_core.h, circ_solve).  Filled by x3d_tdsops_create (tds.hip).
struct CircOp {
    real_t c[9];   // the bulk stencil times rho / alpha
    real_t nr;     // -rho
    real_t pf[8];  // (-rho)^(q + 1): what the value carried into a lane adds to its row q (Q rows per lane, Q <= 8)
    real_t mu[4];  // mu, mu^2, mu^4, mu^8 with mu = (-rho)^Q: the lane-to-lane multiplier of the scans
    real_t phi0;   // -rho / (1 - rho^2): what a unit forward carry into a pencil adds to the solution's first row (HALO form)
};
struct Circ4 { CircOp o[4]; };  // der1st, der2nd, op_s, op_i of a transeq_x launch
struct x3d_tdsops {
    x3d_backend *b;
    int n_tds, n_rhs, move, periodic;
    real_t *dev;  // one allocation holding all tables
    TdsTab tab;
    real_t coeffs[9];  // host copy of the bulk stencil (passed by value to the scan kernels)
    unsigned long long tl_hash;  // FNV-1a of the lane tables: equal operators can share them in LDS (xscan.hip, K3y)
    const real_t *tlc;           // compressed lane tables (xscan_core.h, LTC_*; xwide.hip), or null
    const real_t *tl5;           // lane tables for 5 rows per lane (257..320-row pencils, ygen.hip), or null
    int direct;                  // 1: non-periodic on one rank and the plain Thomas factors reproduce the reference's sweeps
    int circ_ok;                 // 1: periodic, uniform grid, and circ (for tab.Q rows per lane) reproduces the reference's sweeps
    CircOp circ;
    int circ_open_ok;            // 1: the open-ended (HALO) circulant form is on offer: as circ_ok, for periodic operators and for
                                 //    operators whose two ends are neighbour ranks (BC_HALO: the same rows)
    TdsTab tabc;                 // circ_open_ok: the strip corrections of the circulant HALO form in the strip kernels' table layout
    int halo_ws_c, halo_we_c;    // ... and the rows they reach (as halo_ws / halo_we)
    const real_t *td5, *td8h;    // DIRECT lane tables (tds.hip, ygen.hip): 5 rows per lane; 8 rows per lane of a half-wave + row 257
    int narrow_all;              // 1: no stencil of the operator (bulk, start rows, end rows) reaches beyond 2 rows
    int uniform;                 // 1: stretch == 1 and stretch_correct == 0 on every row (a uniform grid): kernels may skip
                                 //    the ST / STC lane-table reads and their multiplications (x * 1.0, + nu * x * 0.0)
    int halo_ws, halo_we;        // rows 1..ws / n-we+1..n: where |dist_sa| / |dist_sc| >= 2^-60 (xscan.hip, *_halo_fix)
    struct x3d_penta *penta;     // compact10_penta: the pentadiagonal LU tables (penta.hip), else null
};
void x3d_penta_free(x3d_tdsops *t);

PencilGeom x3d_geom(const x3d_backend *b, int dir);

// decomposed direction through the tile kernels (xscan.hip, HALO forms)
struct TileHalo {
    const real_t *recv;  // halo rows [side 2][field nf][4][hnp] (side 0: rows -3..0 from prev, 1: n+1..n+4 from next)
    real_t *bsend;       // boundary values out [side 2][nb][np]: side 0 = du_1 (goes to prev), 1 = X_n (to next)
    int np, nf, nb;      // pencils of the direction, fields, operators per pencil
    // a halo row is a plane of pencils: pencil (x, o) sits at o * hp + x.  y pencils: packed (hp = nx); z pencils:
    // the block's own plane layout (hp = nxp, hnp = nxp * nyp) -- four halo rows are four consecutive xy planes of
    // the neighbour's block and travel straight out of it, without a pack kernel
    int hp;
    long hnp;
};
void x3d_halo_layout(const x3d_backend *b, int dir, int *hp, long *hnp);



static inline int x3d_dir_ok(int dir) { return dir >= X3D_DIR_X && dir <= X3D_DIR_Z; }
