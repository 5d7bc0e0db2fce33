// Deferred execution of the reference's OPERATOR-GRANULAR call sequence: fusion INSIDE the library.
//
// The unchanged solver.f90 / time_integrator.f90 / vector_calculus.f90 of the reference issue, per sub-step, 16 reorder,
// 6 sum_{y,z}intox, ~20 veccopy / vecadd, 16 tds_solve, 3 transeq_* (/root/reference/src/solver.f90:291-389, 693-739;
// src/time_integrator.f90:166-282; src/vector_calculus.f90:142-332).  Executed one by one those are real copies and
// axpys: 109.6 ms per step at 512^3 against 48 for the fused driver (profiles/README.md).  With x3d_lazy_enable(b, 1)
// the entry points of base_backend_t only RECORD their call; the queue runs when a result has to be visible to the host
// (reductions, get / set_field_data, x3d_device_sync) or to an entry point that is not queued.  Before it runs, a
// peephole pass rewrites the reference's fixed sequences onto the fused kernels the library already has:
//
//   transeq_<d>(du_d, dv_d, dw_d; ...) ; sum_<d>intox(du, du_d) x 3        -> x3d_transeq_acc(du, dv, dw; ..., accumulate)
//   tds_solve(a; i1; A) ; tds_solve(b; i2; B) ; vecadd(1, b, 1, a)          -> x3d_tds_solve_pair(mode 0)
//   tds_solve(a; i; A) ; tds_solve(b; i; B)                                 -> x3d_tds_solve_pair(mode 1)
//   tds_solve(g; p; A) ; vecadd(s, g, 1, u)                                 -> x3d_tds_solve_acc(u; p; A; scale s)
//   veccopy(y, b) ; vecadd(c1, x1, 1, y) ; vecadd(c2, x2, 1, y) ...         -> x3d_lincomb(y = b + c1 x1 + c2 x2 ...)
//   lincomb(y) ; tds_solve(du; y; A) along x                                -> x3d_tds_solve_lincomb
//   fft_forward(f) ; fft_postprocess_000 ; fft_backward(f)                  -> x3d_poisson_solve_000
//   u += s A(gu) ; v += s B(gv) ; w += s B(gw) ; transeq_x(...; u, v, w)    -> x3d_transeq_x_update
//   pair_z(mode 0) -> d ; reorder ; solve_000 ; reorder ; pair_z(mode 1)     -> the z-first solve (csrc/zfirst.hip)
//   transeq_acc_z(du, dv, dw) ; [olds = du] ; lincomb(u), lincomb(v), lincomb(w) that read du, dv, dw
//                                                                             -> x3d_transeq_lincomb3 (the RK stage in the z launch)
// every one of which is the same arithmetic in the same order as the calls it replaces (tests/test_hip_lazy.py
// compares bit for bit).  A temporary is dropped only when the queue shows it dead: overwritten, or released to the
// allocator (x3d_block_discard, which the shim's release_block issues) before anything reads it.
//
// reorder and veccopy move nothing: all four DIR tags share one device layout, so the destination becomes an ALIAS
// of the source's buffer.  Block addresses are therefore HANDLES while the mode is on: the layer owns the map handle ->
// physical buffer (copy on write: an operation that overwrites a handle whose buffer is shared gets a free buffer; an
// in-place update of a shared buffer runs out of place -- veccopy(olds, curr) followed by the stage's update of curr is
// a buffer swap, as in the fused driver, not a copy).  Entry points outside the queue see the identity map again
// (x3d_lazy_sync: flush + move every handle's data home).
#include <algorithm>
#include <unordered_map>
#include <vector>

#include <unistd.h>

#include "common.h"

enum LKind {
    L_DEAD = 0, L_TRANSEQ, L_TRANSEQ_ACC, L_TDS, L_TDS_ACC, L_PAIR, L_TDS_LIN, L_COPY, L_SUM, L_VECADD, L_LINCOMB, L_VECMULT,
    L_SCALE, L_SHIFT, L_FILL, L_DISCARD, L_FFT_FWD, L_FFT_POST000, L_FFT_BWD, L_SOLVE000, L_TRANSEQ_UPD, L_SPECIES, L_SPECIES_ACC, L_ZFIRST,
    L_FFT_POST010, L_SOLVE010R, L_BIND, L_SETFACE, L_TRANSEQ_STAGE
};

struct LOp {
    int kind = L_DEAD;
    int dir = 0, mode = 0, nterm = 0;
    real_t *o[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // handles written (or updated in place)
    const real_t *in[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const x3d_tdsops *t[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    real_t s[6] = {0, 0, 0, 0, 0, 0};
    void *obj = nullptr;  // x3d_poisson* of the FFT hooks
    int dims[3] = {0, 0, 0};  // L_SETFACE: the field's extent
    // L_TRANSEQ_STAGE (rule 10): o[0..2] the derivative blocks (updated), o[3..5] the stage's results, in[0..2] u, v, w;
    // per variable c: xin[c] the base, xin[3 + 5 c + k] / xs[5 c + k] term k of xn[c], xp[c] the term that IS o[c];
    // mode bit c: o[c] is released behind the launch (its release stood between the combinations); post[c]: see below
    const real_t *xin[18] = {};
    real_t xs[15] = {};
    int xn[3] = {0, 0, 0}, xp[3] = {0, 0, 0};
    real_t *post[3] = {nullptr, nullptr, nullptr};  // handle that becomes an alias of o[c] behind the launch (an AB history save)
};

enum { ST_QUEUED = 0, ST_EXECUTED, ST_ALIAS, ST_TRANSEQ_ACC, ST_PAIR, ST_TDS_ACC, ST_LINCOMB, ST_TDS_LIN, ST_SOLVE000, ST_OOP,
       ST_MATERIALISE, ST_NORMALISE_COPIES, ST_FLUSHES, ST_DROPPED, ST_TRANSEQ_UPD, ST_POOL_SLOT, ST_ZFIRST, ST_DECLINED,
       ST_TRANSEQ_STAGE, ST_N };
// ST_DECLINED: operations of the recorded sequence that no rewrite absorbed although the reference's fixed sequences
// always offer one -- a sum_{y,z}intox launched by itself (rule 1 did not find its transeq_<d> + three sums), a transeq_y / z
// launched without its sums, a reorder / veccopy that had to move data (materialised).  Not wrong, only slow (the
// call-by-call speed for that piece): x3d_lazy_stats reports the count and the process says so ONCE when it ends.

struct x3d_lazy {
    bool on = false, executing = false;
    bool keep_zero_terms = false;
    unsigned rules = ~0u;  // X3D_LAZY_RULES: bit k = rewrite (k + 1) of optimise() allowed (A/B runs, tests)
    std::vector<LOp> q;
    std::unordered_map<const real_t *, real_t *> phys;  // registered handle -> physical buffer (nullptr: none yet)
    std::unordered_map<const real_t *, int> users;      // physical buffer -> handles mapped to it
    std::vector<real_t *> handles;                      // registration order
    std::vector<real_t *> pool;                         // extra physical buffers owned by the layer
    std::unordered_map<const real_t *, real_t *> bind;  // handle -> the buffer an earlier operation filled for its NEXT life (L_BIND)
    // decomposed directions whose transeq the HOST executes when the queue runs (x3d_lazy_set_dist_transeq): bit d = direction d
    unsigned dist_mask = 0;
    x3d_dist_transeq_fn dist_fn = nullptr;
    void *dist_user = nullptr;
    unsigned dist_tds_mask = 0;  // ... and whose tds_solve / operator pairs it executes (x3d_lazy_set_dist_tds)
    x3d_dist_tds_fn dist_tds_fn = nullptr;
    void *dist_tds_user = nullptr;
    long stats[ST_N] = {};
};

static x3d_lazy *lazy_of(x3d_backend *b)
{
    if (!b->lazy) b->lazy = new x3d_lazy();
    return b->lazy;
}

bool x3d_lazy_active(const x3d_backend *b) { return b->lazy && b->lazy->on && !b->lazy->executing; }

bool x3d_lazy_set_executing(x3d_backend *b, bool on)
{
    if (!b->lazy) return false;
    const bool prev = b->lazy->executing;
    b->lazy->executing = on;
    return prev;
}

static std::vector<x3d_lazy *> g_reported;  // X3D_LAZY_REPORT (below)
static std::vector<const x3d_backend *> g_reported_b;

void x3d_lazy_destroy(x3d_backend *b)
{
    if (!b->lazy) return;
    for (size_t k = 0; k < g_reported.size(); k++)
        if (g_reported[k] == b->lazy) { g_reported.erase(g_reported.begin() + k); g_reported_b.erase(g_reported_b.begin() + k); break; }
    for (real_t *p : b->lazy->pool) hipFree(p);
    delete b->lazy;
    b->lazy = nullptr;
}

void x3d_lazy_register(x3d_backend *b, real_t *h)
{
    x3d_lazy *L = lazy_of(b);
    if (L->phys.count(h)) return;
    L->phys[h] = h;
    L->users[h] = 1;
    L->handles.push_back(h);
}

void x3d_lazy_unregister(x3d_backend *b, real_t *h)
{
    if (!b->lazy) return;
    x3d_lazy *L = b->lazy;
    auto it = L->phys.find(h);
    if (it == L->phys.end()) return;
    if (it->second) L->users[it->second]--;
    L->phys.erase(it);
    L->handles.erase(std::remove(L->handles.begin(), L->handles.end(), h), L->handles.end());
}

// ---------------------------------------------------------------- physical buffers
static bool registered(x3d_lazy *L, const real_t *h) { return L->phys.find(h) != L->phys.end(); }

static int find_free(x3d_backend *b, real_t *prefer, real_t **out)
{
    x3d_lazy *L = b->lazy;
    if (prefer && L->users[prefer] == 0) { *out = prefer; return 0; }
    for (real_t *p : L->pool)
        if (L->users[p] == 0) { *out = p; return 0; }
    for (real_t *h : L->handles)
        if (L->users[h] == 0) { *out = h; return 0; }
    real_t *p = nullptr;
    X3D_HIP(hipMalloc(reinterpret_cast<void **>(&p), sizeof(real_t) * b->nblock));
    X3D_HIP(hipMemsetAsync(p, 0, sizeof(real_t) * b->nblock, b->stream));
    L->pool.push_back(p);
    L->users[p] = 0;
    *out = p;
    return 0;
}

static int copy_block(x3d_backend *b, real_t *dst, const real_t *src)
{
    X3D_HIP(hipMemcpyAsync(dst, src, sizeof(real_t) * b->nblock, hipMemcpyDeviceToDevice, b->stream));
    return 0;
}

// the buffer that holds h's data (a handle nothing was written to yet gets a free buffer: its contents are undefined anyway)
static int resolve_in(x3d_backend *b, const real_t *h, const real_t **out)
{
    x3d_lazy *L = b->lazy;
    auto it = L->phys.find(h);
    if (it == L->phys.end()) { *out = h; return 0; }
    if (!it->second) {
        real_t *p = nullptr;
        if (int rc = find_free(b, const_cast<real_t *>(h), &p)) return rc;
        it->second = p;
        L->users[p] = 1;
    }
    *out = it->second;
    return 0;
}

// a buffer h may be written to: its own unless that is shared (then a free one; full = false keeps the contents)
static int prepare_out(x3d_backend *b, real_t *h, bool full, real_t **out)
{
    x3d_lazy *L = b->lazy;
    auto it = L->phys.find(h);
    if (it == L->phys.end()) { *out = h; return 0; }
    real_t *p = it->second;
    if (p && L->users[p] == 1) { *out = p; return 0; }
    real_t *q = nullptr;
    if (p) L->users[p]--;
    if (int rc = find_free(b, h, &q)) return rc;
    it->second = q;
    L->users[q] = 1;
    if (p && !full) {
        L->stats[ST_MATERIALISE]++;
        if (int rc = copy_block(b, q, p)) return rc;
    }
    *out = q;
    return 0;
}

static void drop(x3d_lazy *L, real_t *h)
{
    auto it = L->phys.find(h);
    if (it == L->phys.end() || !it->second) return;
    L->users[it->second]--;
    it->second = nullptr;
}

// every registered handle's data back in its own buffer, nothing shared
static int normalise(x3d_backend *b)
{
    x3d_lazy *L = b->lazy;
    for (real_t *h : L->handles) {
        real_t *p = L->phys[h];
        if (p == h && L->users[h] == 1) continue;
        if (p == h) continue;  // (shared with handles that alias h's own buffer: they move when their turn comes)
        if (L->users[h] > 0) {
            // h's own buffer holds somebody else's data: move it out of the way
            real_t *q = nullptr;
            bool found = false;
            for (real_t *c : L->pool)
                if (L->users[c] == 0) { q = c; found = true; break; }
            if (!found) {
                X3D_HIP(hipMalloc(reinterpret_cast<void **>(&q), sizeof(real_t) * b->nblock));
                L->pool.push_back(q);
                L->users[q] = 0;
            }
            if (int rc = copy_block(b, q, h)) return rc;
            L->stats[ST_NORMALISE_COPIES]++;
            for (real_t *g : L->handles)
                if (g != h && L->phys[g] == h) { L->phys[g] = q; L->users[q]++; L->users[h]--; }
        }
        if (p) {
            if (int rc = copy_block(b, h, p)) return rc;
            L->stats[ST_NORMALISE_COPIES]++;
            L->users[p]--;
        }
        L->phys[h] = h;
        L->users[h] = 1;
    }
    // handles that still alias another handle's own buffer (the owner is home now): give them their own copy
    for (real_t *h : L->handles) {
        real_t *p = L->phys[h];
        if (p == h) continue;
        if (L->users[h] > 0) { x3d_set_error("x3d_lazy: normalise left buffer %p occupied", (void *)h); return 2; }
        if (p) {
            if (int rc = copy_block(b, h, p)) return rc;
            L->stats[ST_NORMALISE_COPIES]++;
            L->users[p]--;
        }
        L->phys[h] = h;
        L->users[h] = 1;
    }
    return 0;
}

// ---------------------------------------------------------------- what an operation touches
enum { A_R = 1, A_W = 2, A_M = 4 };  // read, overwritten completely, updated in place

static int nin(const LOp &op)
{
    switch (op.kind) {
    case L_TRANSEQ: case L_TRANSEQ_ACC: case L_TRANSEQ_UPD: case L_TRANSEQ_STAGE: return 3;  // (UPD: the three gradients; u, v, w are outputs 3..5; STAGE: + xin, see touch)
    case L_TDS: case L_TDS_ACC: case L_COPY: case L_SUM: case L_VECADD: case L_VECMULT: case L_SETFACE: return 1;
    case L_SPECIES: case L_SPECIES_ACC: return 2;  // uvw, spec
    case L_ZFIRST: return 2;
    case L_PAIR: return op.mode == 0 ? 2 : 1;
    case L_LINCOMB: return 1 + op.nterm;  // base, x...
    case L_TDS_LIN: return 1 + op.nterm + ((op.mode & 2) ? 1 : 0);  // base, x..., the wall field (mode & 2)
    default: return 0;
    }
}
static int nout(const LOp &op)
{
    switch (op.kind) {
    case L_TRANSEQ: case L_TRANSEQ_ACC: return 3;
    case L_TRANSEQ_UPD: return 6;  // du, dv, dw written; u, v, w updated in place (and read)
    case L_TRANSEQ_STAGE: return 6;  // du, dv, dw updated; the three results written
    case L_PAIR: return op.mode == 0 ? 1 : 2;
    case L_TDS_LIN: return 2;  // du, y
    case L_ZFIRST: return 2;
    case L_DEAD: case L_FFT_POST000: case L_FFT_POST010: return 0;
    default: return 1;
    }
}
static bool out_is_update(const LOp &op, int slot = 0)
{
    if (op.kind == L_TRANSEQ_UPD) return slot >= 3;
    if (op.kind == L_TRANSEQ_STAGE) return slot < 3;
    switch (op.kind) {
    case L_SPECIES_ACC:
    case L_TRANSEQ_ACC: case L_TDS_ACC: case L_SUM: case L_VECADD: case L_VECMULT: case L_SCALE: case L_SHIFT: case L_FFT_FWD:
    case L_FFT_BWD: case L_SOLVE000: case L_SOLVE010R: case L_SETFACE:
        return true;  // (the FFT hooks: forward only reads f, backward writes the real extent of f -- keep the contents)
    default: return false;
    }
}
static int touch(const LOp &op, const real_t *h)
{
    int m = 0;
    for (int k = 0; k < nin(op); k++)
        if (op.in[k] == h) m |= A_R;
    for (int k = 0; k < nout(op); k++)
        if (op.o[k] == h) m |= out_is_update(op, k) ? A_M : A_W;
    if (op.kind == L_TRANSEQ_STAGE)
        for (int c = 0; c < 3; c++) {
            if (op.xin[c] == h) m |= A_R;
            for (int k = 0; k < op.xn[c]; k++)
                if (op.xin[3 + 5 * c + k] == h) m |= A_R;
            if (op.post[c] && op.post[c] == h) m |= A_W;
        }
    return m;
}
// no operation strictly between lo and hi touches any of `quiet`, none writes any of `stable`
static bool range_clear(const std::vector<LOp> &q, int lo, int hi, std::initializer_list<const real_t *> quiet,
                        const std::vector<const real_t *> &stable)
{
    for (int k = lo + 1; k < hi; k++) {
        const LOp &op = q[k];
        if (op.kind == L_DEAD) continue;
        if (op.kind == L_FFT_POST000 || op.kind == L_FFT_POST010) continue;
        for (const real_t *h : quiet)
            if (h && touch(op, h)) return false;
        for (const real_t *h : stable)
            if (h && (touch(op, h) & (A_W | A_M))) return false;
    }
    return true;
}
// h's contents are not needed after position p: the next operation that touches it overwrites or releases it
static bool dead_after(const std::vector<LOp> &q, int p, const real_t *h)
{
    for (int k = p + 1; k < (int)q.size(); k++) {
        const int m = touch(q[k], h);
        if (!m) continue;
        return m == A_W;
    }
    return false;
}
static int last_touch_before(const std::vector<LOp> &q, int p, const real_t *h)
{
    for (int k = p - 1; k >= 0; k--)
        if (touch(q[k], h)) return k;
    return -1;
}
static int first_touch_after(const std::vector<LOp> &q, int p, const real_t *h)
{
    for (int k = p + 1; k < (int)q.size(); k++)
        if (touch(q[k], h)) return k;
    return -1;
}
static std::vector<const real_t *> inputs_of(const LOp &op)
{
    std::vector<const real_t *> v;
    for (int k = 0; k < nin(op); k++) v.push_back(op.in[k]);
    if (op.kind == L_TRANSEQ_STAGE)
        for (int c = 0; c < 3; c++) {
            v.push_back(op.xin[c]);
            for (int k = 0; k < op.xn[c]; k++) v.push_back(op.xin[3 + 5 * c + k]);
        }
    return v;
}

// ---------------------------------------------------------------- the peephole pass
struct x3d_poisson;
bool x3d_zfirst_on_offer(x3d_poisson *p);
extern "C" int x3d_transeq_stage_ok(x3d_backend *b, int dir, const x3d_tdsops *der1st, const x3d_tdsops *der1st_sym,
                                    const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym);
extern "C" int x3d_transeq_lincomb3(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u,
                                    const real_t *v, const real_t *w, real_t nu, const x3d_tdsops *der1st,
                                    const x3d_tdsops *der1st_sym, const x3d_tdsops *der2nd, const x3d_tdsops *der2nd_sym,
                                    real_t *const y[3], const real_t *const base[3], const int nterm[3], const real_t *c,
                                    real_t *const *x, const int ipend[3], const int store[3], int *done);
bool x3d_zfirst_pairs_ok(const x3d_backend *b, const x3d_tdsops *ta, const x3d_tdsops *tb);

static void optimise(x3d_backend *b)
{
    x3d_lazy *L = b->lazy;
    std::vector<LOp> &q = L->q;
    const int n = (int)q.size();
    // (0) a release takes effect right behind the last operation that touches the block (nothing else sees it): buffers
    // become free -- and aliases exclusive -- as early as the program allows
    for (int p = 0; p < n; p++) {
        if (q[p].kind != L_DISCARD) continue;
        const int k = last_touch_before(q, p, q[p].o[0]);
        if (k + 1 >= p) continue;
        const LOp d = q[p];
        for (int m = p; m > k + 1; m--) q[m] = q[m - 1];
        q[k + 1] = d;
    }
    // (1) transeq_<d> + the three sum_<d>intox of its outputs
    for (int p = 0; p < n && (L->rules & 1u); p++) {
        if (q[p].kind != L_TRANSEQ) continue;
        int ps[3];
        real_t *acc[3];
        bool ok = true;
        for (int c = 0; c < 3 && ok; c++) {
            const int k = first_touch_after(q, p, q[p].o[c]);
            ok = k > 0 && q[k].kind == L_SUM && q[k].in[0] == q[p].o[c] && q[k].dir == q[p].dir && dead_after(q, k, q[p].o[c]);
            if (!ok) break;
            ps[c] = k;
            acc[c] = q[k].o[0];
            for (int d = 0; d < 3; d++) ok = ok && acc[c] != q[p].in[d] && acc[c] != q[p].o[d];
            for (int d = 0; d < c; d++) ok = ok && acc[c] != acc[d];
            ok = ok && range_clear(q, p, k, {acc[c]}, {});
        }
        if (!ok) continue;
        for (int c = 0; c < 3; c++) { q[ps[c]].kind = L_DEAD; q[p].o[c] = acc[c]; }
        q[p].kind = L_TRANSEQ_ACC;
    }
    // (1b) transeq_species(dspec_d; ...) ; sum_<d>intox(r, dspec_d)  (src/solver.f90:507-601)
    for (int p = 0; p < n && (L->rules & 1u); p++) {
        if (q[p].kind != L_SPECIES) continue;
        const int k = first_touch_after(q, p, q[p].o[0]);
        if (k < 0 || q[k].kind != L_SUM || q[k].in[0] != q[p].o[0] || q[k].dir != q[p].dir || !dead_after(q, k, q[p].o[0])) continue;
        real_t *acc = q[k].o[0];
        if (acc == q[p].in[0] || acc == q[p].in[1] || !range_clear(q, p, k, {acc}, {})) continue;
        q[p].kind = L_SPECIES_ACC; q[p].o[0] = acc;
        q[k].kind = L_DEAD;
    }
    // (2) a = A(i1) ; b = B(i2) ; a += b   (divergence_v2c: interpl + stagder of the same direction)
    // fused where the EARLIER solve stands if the later one may move up there, else where the later one stands
    for (int p = 0; p < n && (L->rules & 2u); p++) {
        if (q[p].kind != L_VECADD || q[p].s[0] != 1.0 || q[p].s[1] != 1.0) continue;
        const real_t *X = q[p].in[0];
        real_t *Y = q[p].o[0];
        const int px = last_touch_before(q, p, X), py = last_touch_before(q, p, Y);
        if (px < 0 || py < 0 || q[px].kind != L_TDS || q[py].kind != L_TDS || q[px].o[0] != X || q[py].o[0] != Y) continue;
        if (q[px].dir != q[py].dir || q[px].dir == X3D_DIR_X || !dead_after(q, p, X)) continue;
        const int lo = std::min(px, py), hi = std::max(px, py);
        if (!range_clear(q, hi, p, {X, Y}, {})) continue;
        const LOp &early = q[lo], &late = q[hi];
        int at = -1;
        if (range_clear(q, lo, hi, {late.o[0]}, {late.in[0]})) at = lo;
        else if (range_clear(q, lo, hi, {early.o[0]}, {early.in[0]})) at = hi;
        if (at < 0) continue;
        const real_t *i1 = q[py].in[0], *i2 = q[px].in[0];
        if (Y == i1 || Y == i2 || X == i1 || X == i2) continue;
        LOp f;
        f.kind = L_PAIR; f.mode = 0; f.dir = q[py].dir;
        f.o[0] = Y; f.in[0] = i1; f.in[1] = i2; f.t[0] = q[py].t[0]; f.t[1] = q[px].t[0];
        q[lo].kind = L_DEAD; q[hi].kind = L_DEAD; q[p].kind = L_DEAD;
        q[at] = f;
    }
    // (3) a = A(i) ; b = B(i)   (gradient_c2v: interpl and stagder of the same field), the second right behind the first
    // (releases of other blocks aside)
    for (int p = 0; p < n && (L->rules & 4u); p++) {
        if (q[p].kind != L_TDS || q[p].dir == X3D_DIR_X) continue;
        for (int k = p + 1; k < n; k++) {
            if (q[k].kind == L_DEAD) continue;
            if (q[k].kind == L_DISCARD && q[k].o[0] != q[p].o[0] && q[k].o[0] != q[p].in[0]) continue;
            // (the first solve must not run in place: the second one reads the same input)
            if (q[k].kind == L_TDS && q[k].dir == q[p].dir && q[k].in[0] == q[p].in[0] && q[k].o[0] != q[p].o[0] &&
                q[k].o[0] != q[p].in[0] && q[p].o[0] != q[p].in[0] && range_clear(q, p, k, {q[k].o[0]}, {q[p].in[0]})) {
                LOp f;
                f.kind = L_PAIR; f.mode = 1; f.dir = q[p].dir;
                f.o[0] = q[p].o[0]; f.o[1] = q[k].o[0]; f.in[0] = q[p].in[0]; f.t[0] = q[p].t[0]; f.t[1] = q[k].t[0];
                q[k].kind = L_DEAD;
                q[p] = f;
            }
            break;
        }
    }
    // (4) g = A(p) ; u += s g   (the velocity correction behind gradient_c2v, src/solver.f90:731-733)
    for (int p = 0; p < n && (L->rules & 8u); p++) {
        if (q[p].kind != L_VECADD || q[p].s[1] != 1.0) continue;
        const real_t *G = q[p].in[0];
        real_t *U = q[p].o[0];
        const int pg = last_touch_before(q, p, G);
        if (pg < 0 || q[pg].kind != L_TDS || q[pg].o[0] != G || U == q[pg].in[0] || U == G || G == q[pg].in[0] ||
            !dead_after(q, p, G))
            continue;
        if ((L->dist_tds_mask >> q[pg].dir) & 1u) continue;  // (a decomposed direction's solve runs on the host's side: no accumulating form there)
        if (!range_clear(q, pg, p, {G, U}, {})) continue;
        q[pg].kind = L_TDS_ACC; q[pg].o[0] = U; q[pg].s[0] = q[p].s[0];
        q[p].kind = L_DEAD;
    }
    // (8) u += s A(gu) ; v += s B(gv) ; w += s B(gw) ; transeq_x(du, dv, dw; u, v, w): the velocity correction of the
    // pressure step inside the next sub-step's transeq_x kernel (x3d_transeq_x_update), fused where the FIRST of the
    // three solves stands (their inputs are released right behind them; the outputs of transeq_x are fresh blocks
    // whose release, if it falls in between, the overwrite makes redundant)
    for (int p = 0; p < n && (L->rules & 128u); p++) {
        if (q[p].kind != L_TRANSEQ || q[p].dir != X3D_DIR_X) continue;
        int ks[3];
        bool ok = true;
        for (int c = 0; c < 3 && ok; c++) {
            const real_t *f = q[p].in[c];
            const int k = last_touch_before(q, p, f);
            ok = k >= 0 && q[k].kind == L_TDS_ACC && q[k].o[0] == f && q[k].dir == X3D_DIR_X;
            ks[c] = k;
        }
        if (!ok || q[ks[1]].t[0] != q[ks[2]].t[0] || q[ks[0]].s[0] != q[ks[1]].s[0] || q[ks[0]].s[0] != q[ks[2]].s[0]) continue;
        const int k0 = std::min(ks[0], std::min(ks[1], ks[2])), k1 = std::max(ks[0], std::max(ks[1], ks[2]));
        const real_t *g[3] = {q[ks[0]].in[0], q[ks[1]].in[0], q[ks[2]].in[0]};
        // the later solves move up to k0: their gradients must not be written in between
        (void)k1;
        bool stable = true;
        for (int c = 0; c < 3; c++) stable = stable && range_clear(q, k0, ks[c], {}, {g[c]});
        if (!stable) continue;
        // transeq_x moves up to k0: nothing between may touch its outputs (a bare release aside) or u, v, w
        bool clear = true;
        for (int m = k0 + 1; m < p && clear; m++) {
            const LOp &op = q[m];
            if (op.kind == L_DEAD || m == ks[0] || m == ks[1] || m == ks[2]) continue;
            for (int c = 0; c < 3; c++) {
                if (touch(op, q[p].in[c])) clear = false;
                if (touch(op, q[p].o[c]) && op.kind != L_DISCARD) clear = false;
            }
        }
        for (int c = 0; c < 3; c++)
            for (int d = 0; d < 3; d++) clear = clear && q[p].o[c] != g[d] && q[p].o[c] != q[p].in[d] && q[p].in[c] != g[d];
        if (!clear) continue;
        LOp f;
        f.kind = L_TRANSEQ_UPD; f.dir = X3D_DIR_X;
        for (int c = 0; c < 3; c++) { f.o[c] = q[p].o[c]; f.o[3 + c] = const_cast<real_t *>(q[p].in[c]); f.in[c] = g[c]; }
        for (int c = 0; c < 4; c++) f.t[c] = q[p].t[c];
        f.t[4] = q[ks[0]].t[0]; f.t[5] = q[ks[1]].t[0];
        f.s[0] = q[p].s[0]; f.s[1] = q[ks[0]].s[0];
        for (int m = k0 + 1; m < p; m++)  // releases of the output blocks in between: the overwrite replaces them
            if (q[m].kind == L_DISCARD && (q[m].o[0] == f.o[0] || q[m].o[0] == f.o[1] || q[m].o[0] == f.o[2])) q[m].kind = L_DEAD;
        q[ks[0]].kind = L_DEAD; q[ks[1]].kind = L_DEAD; q[ks[2]].kind = L_DEAD; q[p].kind = L_DEAD;
        q[k0] = f;
    }
    // (5) y = 1 y + a x  ->  lincomb; chains (and a leading veccopy) merge into one
    for (int p = 0; p < n; p++) {
        if (q[p].kind != L_VECADD || q[p].s[1] != 1.0) continue;
        LOp f;
        f.kind = L_LINCOMB; f.o[0] = q[p].o[0]; f.in[0] = q[p].o[0]; f.nterm = 1; f.in[1] = q[p].in[0]; f.s[0] = q[p].s[0];
        q[p] = f;
    }
    for (int p = 0; p < n; p++) {
        if (q[p].kind != L_LINCOMB) continue;
        int merged_into = -1;
        while (q[p].in[0] == q[p].o[0] && (L->rules & 16u)) {  // base == y: whatever wrote y last may be folded in
            real_t *y = q[p].o[0];
            const int k = last_touch_before(q, p, y);
            if (k < 0) break;
            const LOp &e = q[k];
            int ne = 0;
            if (e.kind == L_COPY && e.o[0] == y) ne = 0;
            else if (e.kind == L_LINCOMB && e.o[0] == y) ne = e.nterm;
            else break;
            if (ne + q[p].nterm > 5) break;
            bool selfref = false;  // (a term of the later one that reads y would see the earlier result)
            for (int c = 1; c <= q[p].nterm; c++) selfref = selfref || q[p].in[c] == y;
            if (selfref) break;
            // the later combination moves UP to the earlier one (nothing in between touches y or writes its terms)
            std::vector<const real_t *> later_terms;
            for (int c = 1; c <= q[p].nterm; c++) later_terms.push_back(q[p].in[c]);
            if (!range_clear(q, k, p, {y}, later_terms)) break;
            LOp f;
            f.kind = L_LINCOMB; f.o[0] = y; f.in[0] = e.in[0]; f.nterm = ne + q[p].nterm;
            for (int c = 0; c < ne; c++) { f.in[1 + c] = e.in[1 + c]; f.s[c] = e.s[c]; }
            for (int c = 0; c < q[p].nterm; c++) { f.in[1 + ne + c] = q[p].in[1 + c]; f.s[ne + c] = q[p].s[c]; }
            q[p].kind = L_DEAD;
            q[k] = f;
            merged_into = k;
            break;
        }
        (void)merged_into;
    }
    for (int p = 0; p < n; p++) {
        if (q[p].kind != L_LINCOMB) continue;
        if (!L->keep_zero_terms) {  // y + 0 x = y (up to the sign of a zero)
            LOp &f = q[p];
            int m = 0;
            for (int c = 0; c < f.nterm; c++)
                if (f.s[c] != 0.0) { f.s[m] = f.s[c]; f.in[1 + m] = f.in[1 + c]; m++; }
            f.nterm = m;
            if (m == 0) {
                if (f.in[0] == f.o[0]) f.kind = L_DEAD;
                else { f.kind = L_COPY; }
            }
        }
    }
    // (10) the RK stage of u, v, w inside the LAST transeq launch of the sub-step (src/time_integrator.f90:166-231 behind
    // src/solver.f90:291-389): rhs_c += transeq_z(...) ; [X_c = rhs_c ; release rhs_c] ; y_c = base_c + sum s x, one x being
    // rhs_c (or its alias X_c), for c = u, v, w -> ONE launch of the three-component tile kernel that forms d_c = rhs_c +
    // component, stores it where a later stage reads it and writes y_c (x3d_transeq_lincomb3).  Fused where the LAST of the
    // three combinations stands: the aliases and the saves of the old velocity in between have run by then.  The launch
    // reads u, v, w themselves (the reordered copies it was recorded with are released right behind the transeq)
    for (int p = 0; p < n && (L->rules & 512u); p++) {
        if (q[p].kind != L_TRANSEQ_ACC || q[p].dir == X3D_DIR_X) continue;
        if ((L->dist_mask >> q[p].dir) & 1u) continue;  // (a decomposed direction runs on the host's side: x3d_lazy_set_dist_transeq)
        if (!x3d_transeq_stage_ok(b, q[p].dir, q[p].t[0], q[p].t[1], q[p].t[2], q[p].t[3])) continue;
        const real_t *r[3];
        int kc[3], ip[3] = {0, 0, 0}, K = -1;
        bool ok = true;
        std::vector<int> rel[3];  // releases of r_c behind its combination
        int cp[3] = {-1, -1, -1};  // a save X = r_c behind its combination
        for (int c = 0; c < 3 && ok; c++) {
            const real_t *du = q[p].o[c];
            r[c] = du;
            kc[c] = -1;
            for (int m = p + 1; m < n && ok && kc[c] < 0; m++) {
                const LOp &op = q[m];
                if (op.kind == L_DEAD) continue;
                const int t = touch(op, r[c]) | (r[c] != du ? touch(op, du) : 0);
                if (!t) continue;
                if (op.kind == L_COPY && r[c] == du && op.in[0] == du && op.o[0] != du && dead_after(q, m, du)) { r[c] = op.o[0]; continue; }
                if (op.kind == L_DISCARD && op.o[0] == du && r[c] != du) continue;
                if (op.kind == L_LINCOMB && op.nterm >= 1 && op.nterm <= 5 && (r[c] == du || !touch(op, du))) {
                    int hits = 0;
                    for (int k = 1; k <= op.nterm; k++)
                        if (op.in[k] == r[c]) { hits++; ip[c] = k - 1; }
                    ok = hits == 1 && op.in[0] != r[c] && op.o[0] != r[c];
                    kc[c] = m;
                } else ok = false;
            }
            ok = ok && kc[c] > 0;
            K = std::max(K, kc[c]);
        }
        if (!ok) continue;
        // u, v, w behind the reordered copies the transeq was recorded with
        const real_t *src[3];
        for (int c = 0; c < 3 && ok; c++) {
            const real_t *h = q[p].in[c];
            const int k = last_touch_before(q, p, h);
            src[c] = h;
            if (k >= 0 && q[k].kind == L_COPY && q[k].o[0] == h && q[k].in[0] != h && range_clear(q, k, p, {}, {q[k].in[0]}))
                src[c] = q[k].in[0];
            for (int m = p + 1; m < K && ok; m++) {  // nothing but the three combinations writes it before the launch
                if (q[m].kind == L_DEAD || m == kc[0] || m == kc[1] || m == kc[2]) continue;
                if (touch(q[m], src[c]) & (A_W | A_M)) ok = false;
            }
        }
        for (int c = 0; c < 3 && ok; c++) {
            const LOp &lc = q[kc[c]];
            real_t *y = lc.o[0];
            // the combination moves down to K: its result untouched, its other inputs unwritten on the way; r_c may only be
            // released there (the launch releases it instead)
            std::vector<const real_t *> stable;
            for (int k = 0; k <= lc.nterm; k++)
                if (lc.in[k] != r[c] && lc.in[k] != y) stable.push_back(lc.in[k]);
            for (int m = kc[c] + 1; m < K && ok; m++) {
                const LOp &op = q[m];
                if (op.kind == L_DEAD) continue;
                if (touch(op, r[c])) {
                    // (an Adams-Bashforth history save olds = rhs_c behind the combination -- olds' previous contents being
                    //  one of the combination's terms: the save joins the launch and is replayed behind it)
                    if (op.kind == L_COPY && op.in[0] == r[c] && op.o[0] != r[c] && op.o[0] != y && rel[c].empty() && cp[c] < 0) {
                        cp[c] = m;
                        continue;
                    }
                    if (op.kind == L_DISCARD) rel[c].push_back(m); else ok = false;
                }
                if (cp[c] >= 0 && touch(op, q[cp[c]].o[0])) ok = false;  // (the saved copy's handle: quiet until the launch)
                if (m == kc[0] || m == kc[1] || m == kc[2]) {
                    // (another variable's combination: it joins the launch -- it must not read or write this one's result)
                    if (touch(op, y)) ok = false;
                    continue;
                }
                if (touch(op, y)) ok = false;
                for (const real_t *h : stable)
                    if (touch(op, h) & (A_W | A_M)) ok = false;
            }
        }
        // the launch moves down from p: nothing else reads or writes rhs_c on the way (the scans above covered p..K); a result
        // is no other variable's input, derivative block or result
        for (int c = 0; c < 3 && ok; c++) {
            real_t *y = q[kc[c]].o[0];
            for (int d = 0; d < 3; d++) {
                ok = ok && y != r[d] && y != q[p].o[d] && (y != src[d] || d == c);
                ok = ok && (d == c || y != q[kc[d]].o[0]);
                for (int k = 0; k <= q[kc[d]].nterm && d != c; k++) ok = ok && q[kc[d]].in[k] != y;
            }
        }
        for (int c = 0; c < 3 && ok; c++)
            for (int d = 0; d < c; d++) ok = ok && r[c] != r[d];
        for (int c = 0; c < 3 && ok; c++) {  // a save's handle: not a result, a derivative block, u, v, w or another save's
            if (cp[c] < 0) continue;
            const real_t *X = q[cp[c]].o[0];
            for (int d = 0; d < 3; d++)
                ok = ok && X != q[kc[d]].o[0] && X != r[d] && X != src[d] && (d == c || cp[d] < 0 || X != q[cp[d]].o[0]);
        }
        if (!ok) continue;
        LOp f;
        f.kind = L_TRANSEQ_STAGE; f.dir = q[p].dir; f.s[0] = q[p].s[0];
        for (int k = 0; k < 4; k++) f.t[k] = q[p].t[k];
        for (int c = 0; c < 3; c++) {
            const LOp &lc = q[kc[c]];
            f.o[c] = const_cast<real_t *>(r[c]);
            f.o[3 + c] = lc.o[0];
            f.in[c] = src[c];
            f.xin[c] = lc.in[0];
            f.xn[c] = lc.nterm;
            for (int k = 0; k < lc.nterm; k++) {
                f.xin[3 + 5 * c + k] = lc.in[1 + k];
                f.xs[5 * c + k] = lc.s[k];
            }
            f.xp[c] = ip[c];
            f.xin[3 + 5 * c + ip[c]] = r[c];  // (the term that is the derivative block, under the name it has at K)
            if (!rel[c].empty()) f.mode |= 1 << c;
            if (cp[c] >= 0) f.post[c] = q[cp[c]].o[0];
        }
        for (int c = 0; c < 3; c++) {
            for (int m : rel[c]) q[m].kind = L_DEAD;
            if (cp[c] >= 0) q[cp[c]].kind = L_DEAD;
            q[kc[c]].kind = L_DEAD;
        }
        q[p].kind = L_DEAD;
        q[K] = f;
    }
    // (6) y = lincomb ; du = A(y) along x: the stage is the operator's prologue.  Fused where the solve stands if the
    // combination may move down there (its terms are not overwritten or released on the way), else where the
    // combination stands if the solve may move up (its output is not in use in between)
    for (int p = 0; p < n && (L->rules & 32u); p++) {
        if (q[p].kind != L_LINCOMB) continue;
        real_t *y = q[p].o[0];
        int k = first_touch_after(q, p, y);
        // ... with the y faces of y stamped from a wall field in between (field_set_face_from_field(Y_FACE): the channel
        // case's apply_BC, src/case/channel.f90:214-231): x3d_tds_solve_lincomb_wall does all three.  The stamping is taken
        // out of the queue for the checks below and dies with the rewrite.
        int ksf = -1;
        const real_t *wall = nullptr;
        if (k >= 0 && q[k].kind == L_SETFACE && q[k].o[0] == y && q[p].nterm <= 5 && q[k].in[0] != y &&
            q[k].dims[0] == b->nx && q[k].dims[1] == b->ny && q[k].dims[2] == b->nz) {  // (the whole vertex block's faces)
            ksf = k;
            wall = q[k].in[0];
            k = first_touch_after(q, ksf, y);
        }
        if (k < 0 || q[k].kind != L_TDS || q[k].dir != X3D_DIR_X || q[k].in[0] != y) continue;
        real_t *du = q[k].o[0];
        bool ok = du != y && du != wall;
        for (int c = 0; c <= q[p].nterm; c++) ok = ok && du != q[p].in[c];
        if (!ok) continue;
        LOp sf;
        if (ksf >= 0) { sf = q[ksf]; q[ksf].kind = L_DEAD; }
        auto undo_sf = [&]() { if (ksf >= 0) q[ksf] = sf; };
        std::vector<const real_t *> stable = inputs_of(q[p]);
        if (wall) stable.push_back(wall);
        int at = -1;
        if (range_clear(q, p, k, {y}, stable)) at = k;
        else if (range_clear(q, p, k, {y, du}, wall ? std::vector<const real_t *>{wall} : std::vector<const real_t *>{})) at = p;
        if (at < 0) {
            // neither move is possible when the blocks of the three stages change hands in a ring (the allocator gives the
            // solve of u the block v's stage has just released, and so on: RK3's last stage, once per step): the handle du
            // is still in use under its previous life between p and k.  The solve starts a NEW life of it (it overwrites
            // the whole block), so the fused kernel may run where the combination stood and write into a free buffer of
            // the layer; an L_BIND where the solve stood hands that buffer to the handle.
            bool busy = !registered(L, du) ||
                        !range_clear(q, p, k, {y}, wall ? std::vector<const real_t *>{wall} : std::vector<const real_t *>{});
            // one pending buffer per handle, in the WHOLE queue: a fused solve of du before p whose bind lies behind k would
            // enclose this pair -- its buffer would be overwritten in L->bind and leak
            for (int m = 0; m < n && !busy; m++)
                busy = m != p && m != k &&
                       ((q[m].kind == L_BIND && q[m].o[0] == du) || (q[m].kind == L_TDS_LIN && (q[m].mode & 1) && q[m].obj == du));
            busy = busy || L->bind.count(du) != 0;
            if (busy) { undo_sf(); continue; }
            LOp f = q[p];
            f.kind = L_TDS_LIN; f.dir = X3D_DIR_X; f.o[0] = nullptr; f.obj = du; f.mode = 1 | (wall ? 2 : 0); f.o[1] = y;
            f.t[0] = q[k].t[0];
            if (wall) f.in[1 + f.nterm] = wall;
            q[p] = f;
            LOp g;
            g.kind = L_BIND; g.o[0] = du;
            q[k] = g;
            continue;
        }
        LOp f = q[p];
        f.kind = L_TDS_LIN; f.dir = X3D_DIR_X; f.o[0] = du; f.o[1] = y; f.t[0] = q[k].t[0];
        f.mode = wall ? 2 : 0;
        if (wall) f.in[1 + f.nterm] = wall;
        q[p].kind = L_DEAD; q[k].kind = L_DEAD;
        q[at] = f;
    }
    // (7) fft_forward(f) ; fft_postprocess_000 ; fft_backward(f) -> the solver's one-call form (z passes fused);
    //     with fft_postprocess_010 between them (poisson_010's middle, src/poisson_fft.f90:228-242) -> x3d_poisson_solve_010_rows
    //     (256 cells along a stretched y: y transformed last, inside the kernels that post-process, csrc/y010.hip)
    for (int p = 0; p + 2 < n && (L->rules & 64u); p++) {
        if (q[p].kind != L_FFT_FWD) continue;
        int a = p + 1;
        while (a < n && q[a].kind == L_DEAD) a++;
        int c = a + 1;
        while (c < n && q[c].kind == L_DEAD) c++;
        if (c >= n || (q[a].kind != L_FFT_POST000 && q[a].kind != L_FFT_POST010) || q[c].kind != L_FFT_BWD) continue;
        if (q[a].obj != q[p].obj || q[c].obj != q[p].obj || q[c].o[0] != q[p].o[0]) continue;
        q[p].kind = q[a].kind == L_FFT_POST000 ? L_SOLVE000 : L_SOLVE010R;
        q[a].kind = L_DEAD; q[c].kind = L_DEAD;
    }
    // (9) d = A(i1) + B(i2) along z ; [p_temp = d] ; solve_000(p_temp) ; [pressure = p_temp] ; o1 = A'(pressure),
    // o2 = B'(pressure) along z (pressure_correction, src/solver.f90:693-739 with poisson_fft :653-678): the z-first
    // solve -- the divergence and the pressure are never stored.  Fused where the first pair stands (o1, o2 are fresh
    // blocks); the fields in between must be dead behind their one reader
    for (int p = 0; p < n && (L->rules & 256u); p++) {
        if (q[p].kind != L_SOLVE000) continue;
        x3d_poisson *pf = (x3d_poisson *)q[p].obj;
        if (!x3d_zfirst_on_offer(pf)) continue;
        // backwards: whatever wrote the solve's block, through at most one alias
        real_t *h1 = q[p].o[0];
        int kw = last_touch_before(q, p, h1), c1 = -1;
        if (kw >= 0 && q[kw].kind == L_COPY && q[kw].o[0] == h1) { c1 = kw; kw = last_touch_before(q, c1, q[c1].in[0]); }
        const real_t *h0 = c1 >= 0 ? q[c1].in[0] : h1;
        if (kw < 0 || q[kw].kind != L_PAIR || q[kw].mode != 0 || q[kw].dir != X3D_DIR_Z || q[kw].o[0] != h0) continue;
        // forwards: the one reader of the result, through at most one alias
        int kr = first_touch_after(q, p, h1), c2 = -1;
        const real_t *h2 = h1;
        if (kr >= 0 && q[kr].kind == L_COPY && q[kr].in[0] == h1) { c2 = kr; h2 = q[c2].o[0]; kr = first_touch_after(q, c2, h2); }
        if (kr < 0 || q[kr].kind != L_PAIR || q[kr].mode != 1 || q[kr].dir != X3D_DIR_Z || q[kr].in[0] != h2) continue;
        if (!x3d_zfirst_pairs_ok(b, q[kw].t[0], q[kw].t[1]) || !x3d_zfirst_pairs_ok(b, q[kr].t[0], q[kr].t[1])) continue;
        // every intermediate dies with its reader, nothing else looks at them
        bool ok = dead_after(q, kr, h2) && (c2 < 0 || dead_after(q, c2, h1)) && (c1 < 0 || dead_after(q, c1, h0));
        for (int m = kw + 1; m < kr && ok; m++) {
            if (m == c1 || m == p || m == c2 || q[m].kind == L_DEAD) continue;
            const bool inter = touch(q[m], h0) || touch(q[m], h1) || touch(q[m], h2);
            if (inter && q[m].kind != L_DISCARD) ok = false;  // (their releases, hoisted behind the one reader, aside)
        }
        // the gradient's pair moves up to the divergence's: its outputs must be untouched in between (releases aside)
        real_t *o1 = q[kr].o[0], *o2 = q[kr].o[1];
        for (int m = kw + 1; m < kr && ok; m++) {
            if (q[m].kind == L_DEAD || q[m].kind == L_DISCARD) continue;
            if (touch(q[m], o1) || touch(q[m], o2)) ok = false;
        }
        ok = ok && o1 != q[kw].in[0] && o1 != q[kw].in[1] && o2 != q[kw].in[0] && o2 != q[kw].in[1] && o1 != o2;
        if (!ok) continue;
        LOp f;
        f.kind = L_ZFIRST; f.dir = X3D_DIR_Z; f.obj = pf;
        f.in[0] = q[kw].in[0]; f.in[1] = q[kw].in[1]; f.o[0] = o1; f.o[1] = o2;
        f.t[0] = q[kw].t[0]; f.t[1] = q[kw].t[1]; f.t[2] = q[kr].t[0]; f.t[3] = q[kr].t[1];
        for (int m = kw + 1; m < kr; m++)  // releases of the output blocks in between: the overwrite replaces them
            if (q[m].kind == L_DISCARD && (q[m].o[0] == o1 || q[m].o[0] == o2)) q[m].kind = L_DEAD;
        q[p].kind = L_DEAD; q[kr].kind = L_DEAD;
        if (c1 >= 0) q[c1].kind = L_DEAD;
        if (c2 >= 0) q[c2].kind = L_DEAD;
        q[kw] = f;
    }
}

// ---------------------------------------------------------------- execution
extern "C" int x3d_poisson_fft_forward(x3d_poisson *p, const real_t *f_in);
extern "C" int x3d_poisson_postprocess_000(x3d_poisson *p);
extern "C" int x3d_poisson_fft_backward(x3d_poisson *p, real_t *f_out);
extern "C" int x3d_poisson_solve_000(x3d_poisson *p, real_t *f);
extern "C" int x3d_poisson_zfirst_middle(x3d_poisson *p);
extern "C" int x3d_tds_pair_zfirst(x3d_backend *b, x3d_poisson *poisson, int mode, real_t *out1, real_t *out2,
                                   const real_t *in1, const real_t *in2, const x3d_tdsops *ta, const x3d_tdsops *tb,
                                   int *done);

// (the order of enum LKind: the roctx range of an operation the queue issues says which rewritten form it is)
static const char *const LKIND_NAME[] = {
    "lazy:dead", "lazy:transeq", "lazy:transeq+3sums (transeq_acc)", "lazy:tds_solve", "lazy:tds_solve+vecadd (tds_acc)",
    "lazy:tds pair", "lazy:lincomb+tds_solve (tds_lin)", "lazy:copy/alias", "lazy:sum_intox (declined)", "lazy:vecadd", "lazy:lincomb",
    "lazy:vecmult", "lazy:scale", "lazy:shift", "lazy:fill", "lazy:discard", "lazy:fft_forward", "lazy:fft_postprocess_000",
    "lazy:fft_backward", "lazy:poisson_000 (one solve)", "lazy:velocity correction+transeq_x", "lazy:transeq_species",
    "lazy:transeq_species_acc", "lazy:z-first solve", "lazy:fft_postprocess_010", "lazy:poisson_010 rows", "lazy:bind", "lazy:set_face",
    "lazy:transeq_z+RK stage"};
static_assert(sizeof(LKIND_NAME) / sizeof(LKIND_NAME[0]) == L_TRANSEQ_STAGE + 1, "LKIND_NAME follows enum LKind");

static int exec(x3d_backend *b, const LOp &op)
{
    X3D_RANGE(LKIND_NAME[op.kind]);
    x3d_lazy *L = b->lazy;
    const real_t *in[7] = {};
    real_t *o[6] = {};
    if (op.kind == L_COPY) {
        // dst becomes an alias of src's buffer
        if (!registered(L, op.o[0]) || !registered(L, op.in[0])) {
            if (int rc = resolve_in(b, op.in[0], &in[0])) return rc;
            if (int rc = prepare_out(b, op.o[0], true, &o[0])) return rc;
            return copy_block(b, o[0], in[0]);
        }
        if (op.o[0] == op.in[0]) return 0;
        if (int rc = resolve_in(b, op.in[0], &in[0])) return rc;
        drop(L, op.o[0]);
        L->phys[op.o[0]] = const_cast<real_t *>(in[0]);
        L->users[in[0]]++;
        L->stats[ST_ALIAS]++;
        return 0;
    }
    if (op.kind == L_DISCARD) { drop(L, op.o[0]); return 0; }
    if (op.kind == L_BIND) {  // the handle's new life starts in the buffer a fused kernel filled earlier (rule 6)
        auto it = L->bind.find(op.o[0]);
        if (it == L->bind.end()) { x3d_set_error("x3d_lazy: L_BIND without a pending buffer"); return 2; }
        drop(L, op.o[0]);
        L->phys[op.o[0]] = it->second;  // (users[buffer] is 1 since the kernel took it)
        L->bind.erase(it);
        return 0;
    }
    if (op.kind == L_TRANSEQ_STAGE) {
        const real_t *f[3], *base[3];
        real_t *rr[3], *y[3], *x[15] = {};
        for (int c = 0; c < 3; c++) {
            if (int rc = resolve_in(b, op.in[c], &f[c])) return rc;
            if (int rc = resolve_in(b, op.xin[c], &base[c])) return rc;
            for (int k = 0; k < op.xn[c]; k++) {
                const real_t *t = nullptr;
                if (int rc = resolve_in(b, op.xin[3 + 5 * c + k], &t)) return rc;
                x[5 * c + k] = const_cast<real_t *>(t);  // (read only: the C ABI's term lists are not const-qualified)
            }
        }
        for (int c = 0; c < 3; c++) {
            if (int rc = prepare_out(b, op.o[c], false, &rr[c])) return rc;
            x[5 * c + op.xp[c]] = rr[c];
        }
        for (int c = 0; c < 3; c++) {
            real_t *before = registered(L, op.o[3 + c]) ? L->phys[op.o[3 + c]] : nullptr;
            if (int rc = prepare_out(b, op.o[3 + c], true, &y[c])) return rc;
            if (before && before != y[c]) L->stats[ST_OOP]++;
        }
        int store[3], done = 0;
        for (int c = 0; c < 3; c++) store[c] = (op.post[c] || !((op.mode >> c) & 1)) ? 1 : 0;
        // (a result in a buffer the launch reads for another purpose -- only its own variable may be updated in place --
        //  cannot happen with handles kept apart by the rule, but the call-by-call form below is always right)
        bool clean = true;
        for (int c = 0; c < 3; c++)
            for (int d = 0; d < 3; d++) {
                clean = clean && y[c] != rr[d] && (y[c] != f[d] || d == c) && (d == c || (y[c] != y[d] && y[c] != base[d]));
                for (int k = 0; k < op.xn[d]; k++) clean = clean && y[c] != x[5 * d + k];
            }
        L->stats[ST_EXECUTED]++;
        if (clean) {
            if (int rc = x3d_transeq_lincomb3(b, op.dir, rr[0], rr[1], rr[2], f[0], f[1], f[2], op.s[0], op.t[0], op.t[1], op.t[2],
                                              op.t[3], y, base, op.xn, op.xs, x, op.xp, store, &done))
                return rc;
        }
        if (done) L->stats[ST_TRANSEQ_STAGE]++;
        else {
            L->stats[ST_TRANSEQ_ACC]++;
            L->stats[ST_LINCOMB] += 3;
            if (int rc = x3d_transeq_acc(b, op.dir, rr[0], rr[1], rr[2], f[0], f[1], f[2], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3], 1))
                return rc;
            for (int c = 0; c < 3; c++)
                if (int rc = x3d_lincomb(b, y[c], base[c], op.xn[c], &op.xs[5 * c], &x[5 * c])) return rc;
        }
        for (int c = 0; c < 3; c++) {
            if (op.post[c]) {  // the history save that stood behind the combination
                LOp cpy;
                cpy.kind = L_COPY; cpy.o[0] = op.post[c]; cpy.in[0] = op.o[c];
                if (int rc = exec(b, cpy)) return rc;
            }
            if ((op.mode >> c) & 1) drop(L, op.o[c]);
        }
        return 0;
    }
    for (int k = 0; k < nin(op); k++)
        if (int rc = resolve_in(b, op.in[k], &in[k])) return rc;
    for (int k = 0; k < nout(op); k++) {
        if (op.kind == L_TDS_LIN && (op.mode & 1) && k == 0) {  // du goes to a free buffer, bound to its handle later
            real_t *f = nullptr;
            if (int rc = find_free(b, nullptr, &f)) return rc;
            L->users[f] = 1;
            L->bind[static_cast<real_t *>(op.obj)] = f;
            o[0] = f;
            continue;
        }
        real_t *before = registered(L, op.o[k]) ? L->phys[op.o[k]] : nullptr;
        if (int rc = prepare_out(b, op.o[k], !out_is_update(op, k), &o[k])) return rc;
        if (before && before != o[k]) L->stats[ST_OOP]++;
    }
    L->stats[ST_EXECUTED]++;
    switch (op.kind) {
    case L_TRANSEQ_ACC: L->stats[ST_TRANSEQ_ACC]++; break;
    case L_PAIR: L->stats[ST_PAIR]++; break;
    case L_TDS_ACC: L->stats[ST_TDS_ACC]++; break;
    case L_LINCOMB: L->stats[ST_LINCOMB]++; break;
    case L_TDS_LIN: L->stats[ST_TDS_LIN]++; break;
    case L_SOLVE000: case L_SOLVE010R: L->stats[ST_SOLVE000]++; break;
    case L_TRANSEQ_UPD: L->stats[ST_TRANSEQ_UPD]++; break;
    case L_ZFIRST: L->stats[ST_ZFIRST]++; break;
    case L_SUM: L->stats[ST_DECLINED]++; break;
    case L_TRANSEQ: if (op.dir != X3D_DIR_X) L->stats[ST_DECLINED]++; break;
    default: break;
    }
    switch (op.kind) {
    case L_TRANSEQ:
        if (L->dist_fn && ((L->dist_mask >> op.dir) & 1u))
            return L->dist_fn(L->dist_user, op.dir, o[0], o[1], o[2], in[0], in[1], in[2], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3], 0);
        return x3d_transeq(b, op.dir, o[0], o[1], o[2], in[0], in[1], in[2], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3]);
    case L_TRANSEQ_ACC:
        // a decomposed direction: the host's pack -> exchange -> tile kernel -> exchange -> strip correction, on the buffers the
        // queue resolved; every rank runs the same queue at the same call of the program, so the exchanges meet
        if (L->dist_fn && ((L->dist_mask >> op.dir) & 1u))
            return L->dist_fn(L->dist_user, op.dir, o[0], o[1], o[2], in[0], in[1], in[2], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3], 1);
        return x3d_transeq_acc(b, op.dir, o[0], o[1], o[2], in[0], in[1], in[2], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3], 1);
    case L_TDS:
        if (L->dist_tds_fn && ((L->dist_tds_mask >> op.dir) & 1u))
            return L->dist_tds_fn(L->dist_tds_user, op.dir, 2, o[0], nullptr, in[0], nullptr, op.t[0], nullptr);
        return x3d_tds_solve(b, o[0], in[0], op.t[0], op.dir);
    case L_TDS_ACC: return x3d_tds_solve_acc(b, o[0], in[0], op.t[0], op.dir, 1, op.s[0]);
    case L_PAIR:
        if (L->dist_tds_fn && ((L->dist_tds_mask >> op.dir) & 1u))
            return L->dist_tds_fn(L->dist_tds_user, op.dir, op.mode, o[0], op.mode == 1 ? o[1] : nullptr, in[0],
                                  op.mode == 0 ? in[1] : nullptr, op.t[0], op.t[1]);
        return x3d_tds_solve_pair(b, op.dir, op.mode, o[0], o[1], in[0], in[1], op.t[0], op.t[1]);
    case L_TDS_LIN:
        if (op.mode & 2) return x3d_tds_solve_lincomb_wall(b, op.dir, o[0], op.t[0], o[1], in[0], op.nterm, op.s, &in[1], in[1 + op.nterm]);
        return x3d_tds_solve_lincomb(b, op.dir, o[0], op.t[0], o[1], in[0], op.nterm, op.s, &in[1]);
    case L_SETFACE: return x3d_field_set_face_from_field(b, o[0], in[0], op.dims, 0.0, X3D_Y_FACE, 0.0);
    case L_SUM: return x3d_sum_intox(b, o[0], in[0], op.dir);
    case L_VECADD: return x3d_vecadd(b, op.s[0], in[0], op.s[1], o[0]);
    case L_LINCOMB: return x3d_lincomb(b, o[0], in[0], op.nterm, op.s, &in[1]);
    case L_VECMULT: return x3d_vecmult(b, o[0], in[0]);
    case L_SCALE: return x3d_field_scale(b, o[0], op.s[0]);
    case L_SHIFT: return x3d_field_shift(b, o[0], op.s[0]);
    case L_FILL: return x3d_block_fill(b, o[0], op.s[0]);
    case L_FFT_FWD: return x3d_poisson_fft_forward((x3d_poisson *)op.obj, o[0]);
    case L_FFT_POST000: return x3d_poisson_postprocess_000((x3d_poisson *)op.obj);
    case L_FFT_BWD: return x3d_poisson_fft_backward((x3d_poisson *)op.obj, o[0]);
    case L_SOLVE000: return x3d_poisson_solve_000((x3d_poisson *)op.obj, o[0]);
    case L_FFT_POST010: return x3d_poisson_postprocess_010((x3d_poisson *)op.obj);
    case L_SOLVE010R: return x3d_poisson_solve_010_rows((x3d_poisson *)op.obj, o[0]);
    case L_SPECIES: case L_SPECIES_ACC:
        return x3d_transeq_species(b, op.dir, o[0], in[0], in[1], op.s[0], op.t[0], op.t[1], op.t[2], op.kind == L_SPECIES_ACC);
    case L_TRANSEQ_UPD: {
        int done = 0;
        if (int rc = x3d_transeq_x_update(b, o[0], o[1], o[2], o[3], o[4], o[5], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3],
                                          in[0], in[1], in[2], op.t[4], op.t[5], op.s[1], &done))
            return rc;
        if (done) return 0;
        // these pencils are not served by the kernel: the calls it stands for, one after the other
        L->stats[ST_TRANSEQ_UPD]--;
        L->stats[ST_TDS_ACC] += 3;
        if (int rc = x3d_tds_solve_acc(b, o[3], in[0], op.t[4], X3D_DIR_X, 1, op.s[1])) return rc;
        if (int rc = x3d_tds_solve_acc(b, o[4], in[1], op.t[5], X3D_DIR_X, 1, op.s[1])) return rc;
        if (int rc = x3d_tds_solve_acc(b, o[5], in[2], op.t[5], X3D_DIR_X, 1, op.s[1])) return rc;
        return x3d_transeq(b, X3D_DIR_X, o[0], o[1], o[2], o[3], o[4], o[5], op.s[0], op.t[0], op.t[1], op.t[2], op.t[3]);
    }
    case L_ZFIRST: {
        x3d_poisson *pf = (x3d_poisson *)op.obj;
        int done = 0;
        if (int rc = x3d_tds_pair_zfirst(b, pf, 0, nullptr, nullptr, in[0], in[1], op.t[0], op.t[1], &done)) return rc;
        if (done) {
            if (int rc = x3d_poisson_zfirst_middle(pf)) return rc;
            if (int rc = x3d_tds_pair_zfirst(b, pf, 1, o[0], o[1], nullptr, nullptr, op.t[2], op.t[3], &done)) return rc;
            if (done) return 0;
            x3d_set_error("x3d_lazy: the z-first pair served the divergence but not the gradient");
            return 2;
        }
        // not served after all: the calls it stands for, through a scratch buffer
        L->stats[ST_ZFIRST]--;
        L->stats[ST_PAIR] += 2;
        L->stats[ST_SOLVE000]++;
        real_t *d = nullptr;
        if (int rc = find_free(b, nullptr, &d)) return rc;
        L->users[d] = 1;
        int rc = x3d_tds_solve_pair(b, X3D_DIR_Z, 0, d, nullptr, in[0], in[1], op.t[0], op.t[1]);
        if (!rc) rc = x3d_poisson_solve_000(pf, d);
        if (!rc) rc = x3d_tds_solve_pair(b, X3D_DIR_Z, 1, o[0], o[1], d, nullptr, op.t[2], op.t[3]);
        L->users[d] = 0;
        return rc;
    }
    default: x3d_set_error("x3d_lazy: unknown operation %d in the queue", op.kind); return 2;
    }
}

// X3D_LAZY_DUMP=1: the queue before and after the rewrite, on stderr (handles numbered in order of appearance)
static void dump(const x3d_lazy *L, const char *title)
{
    static const char *names[] = {"dead", "transeq", "transeq_acc", "tds", "tds_acc", "pair", "tds_lin", "copy", "sum", "vecadd",
                                  "lincomb", "vecmult", "scale", "shift", "fill", "discard", "fft_fwd", "fft_post000", "fft_bwd",
                                  "solve000", "transeq_upd", "species", "species_acc", "zfirst", "fft_post010", "solve010_rows",
                                  "bind", "setface", "transeq_stage"};
    std::unordered_map<const real_t *, int> id;
    auto nm = [&](const real_t *h) { if (!h) return -1; auto it = id.find(h); if (it == id.end()) it = id.emplace(h, (int)id.size()).first; return it->second; };
    fprintf(stderr, "---- %s (%zu operations)\n", title, L->q.size());
    int k = 0;
    for (const LOp &op : L->q) {
        if (op.kind == L_DEAD) { k++; continue; }
        fprintf(stderr, "%4d %-12s dir %d mode %d out", k++, names[op.kind], op.dir, op.mode);
        for (int c = 0; c < nout(op); c++) fprintf(stderr, " %d", nm(op.o[c]));
        fprintf(stderr, " | in");
        for (int c = 0; c < nin(op); c++) fprintf(stderr, " %d", nm(op.in[c]));
        if (op.kind == L_LINCOMB || op.kind == L_TDS_LIN || op.kind == L_VECADD) {
            fprintf(stderr, " | c");
            for (int c = 0; c < (op.kind == L_VECADD ? 2 : op.nterm); c++) fprintf(stderr, " %g", op.s[c]);
        }
        fprintf(stderr, "\n");
    }
}

int x3d_lazy_flush_c(x3d_backend *b)
{
    x3d_lazy *L = b->lazy;
    if (!L || L->executing || L->q.empty()) return 0;
    L->stats[ST_FLUSHES]++;
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("X3D_LAZY_DUMP"); dbg = e && e[0] == '1'; }
    if (dbg) dump(L, "recorded");
    optimise(b);
    if (dbg) dump(L, "rewritten");
    L->executing = true;
    int rc = 0;
    for (const LOp &op : L->q) {
        if (op.kind == L_DEAD) { L->stats[ST_DROPPED]++; continue; }
        rc = exec(b, op);
        if (rc) break;
    }
    L->executing = false;
    L->q.clear();
    // a fused solve whose L_BIND never ran (an error between the two): its buffer goes back to the pool, the handle keeps
    // its old data -- nothing stays pinned behind a failed flush
    for (auto &kv : L->bind)
        if (kv.second) L->users[kv.second] = 0;
    L->bind.clear();
    return rc;
}

// flush, then the buffer a queued-mode caller must hand to an entry point that runs at once
int x3d_lazy_in(x3d_backend *b, const real_t *h, const real_t **out)
{
    *out = h;
    if (!b->lazy || !b->lazy->on || b->lazy->executing) return 0;
    if (int rc = x3d_lazy_flush_c(b)) return rc;
    return resolve_in(b, h, out);
}
int x3d_lazy_out(x3d_backend *b, real_t *h, bool full, real_t **out)
{
    *out = h;
    if (!b->lazy || !b->lazy->on || b->lazy->executing) return 0;
    if (int rc = x3d_lazy_flush_c(b)) return rc;
    return prepare_out(b, h, full, out);
}

int x3d_lazy_sync_c(x3d_backend *b)
{
    if (!b->lazy || !b->lazy->on || b->lazy->executing) return 0;
    if (int rc = x3d_lazy_flush_c(b)) return rc;
    return normalise(b);
}

static int push(x3d_backend *b, const LOp &op)
{
    x3d_lazy *L = b->lazy;
    L->q.push_back(op);
    L->stats[ST_QUEUED]++;
    if (L->q.size() >= 4096) return x3d_lazy_flush_c(b);  // (never reached by a time step: a bound, not a window)
    return 0;
}

// ---------------------------------------------------------------- recording (called by the entry points while the mode is on)
int x3d_lazy_transeq(x3d_backend *b, int dir, real_t *du, real_t *dv, real_t *dw, const real_t *u, const real_t *v,
                     const real_t *w, real_t nu, const x3d_tdsops *t0, const x3d_tdsops *t1, const x3d_tdsops *t2,
                     const x3d_tdsops *t3)
{
    // transeq_x opens a sub-step (src/solver.f90:320): no rewrite reaches back across it, and everything the previous
    // sub-step released has been seen -- what has been recorded runs now, the device works on it while the host records on
    LOp op;
    op.kind = L_TRANSEQ; op.dir = dir;
    op.o[0] = du; op.o[1] = dv; op.o[2] = dw; op.in[0] = u; op.in[1] = v; op.in[2] = w;
    op.s[0] = nu; op.t[0] = t0; op.t[1] = t1; op.t[2] = t2; op.t[3] = t3;
    if (int rc = push(b, op)) return rc;
    // (transeq_x itself belongs to the window it closes: the velocity correction left by the pressure step folds into it)
    return dir == X3D_DIR_X ? x3d_lazy_flush_c(b) : 0;
}
int x3d_lazy_species(x3d_backend *b, int dir, real_t *dspec, const real_t *uvw, const real_t *spec, real_t nu,
                     const x3d_tdsops *t0, const x3d_tdsops *t1, const x3d_tdsops *t2, int accumulate)
{
    LOp op;
    op.kind = accumulate ? L_SPECIES_ACC : L_SPECIES; op.dir = dir;
    op.o[0] = dspec; op.in[0] = uvw; op.in[1] = spec; op.s[0] = nu; op.t[0] = t0; op.t[1] = t1; op.t[2] = t2;
    return push(b, op);
}
int x3d_lazy_tds(x3d_backend *b, real_t *du, const real_t *u, const x3d_tdsops *t, int dir)
{
    LOp op;
    op.kind = L_TDS; op.dir = dir; op.o[0] = du; op.in[0] = u; op.t[0] = t;
    return push(b, op);
}
int x3d_lazy_copy(x3d_backend *b, real_t *dst, const real_t *src)
{
    LOp op;
    op.kind = L_COPY; op.o[0] = dst; op.in[0] = src;
    return push(b, op);
}
int x3d_lazy_sum(x3d_backend *b, real_t *u, const real_t *u_, int dir)
{
    LOp op;
    op.kind = L_SUM; op.dir = dir; op.o[0] = u; op.in[0] = u_;
    return push(b, op);
}
int x3d_lazy_vecadd(x3d_backend *b, real_t a, const real_t *x, real_t bb, real_t *y)
{
    LOp op;
    op.kind = L_VECADD; op.o[0] = y; op.in[0] = x; op.s[0] = a; op.s[1] = bb;
    return push(b, op);
}
int x3d_lazy_unary(x3d_backend *b, int kind, real_t *f, const real_t *x, real_t a)
{
    LOp op;
    op.kind = kind == 0 ? L_VECMULT : kind == 1 ? L_SCALE : kind == 2 ? L_SHIFT : L_FILL;
    op.o[0] = f; op.in[0] = x; op.s[0] = a;
    return push(b, op);
}
int x3d_lazy_setface(x3d_backend *b, real_t *f, const real_t *f_start, const int dims[3])
{
    LOp op;
    op.kind = L_SETFACE;
    op.o[0] = f; op.in[0] = f_start;
    for (int k = 0; k < 3; k++) op.dims[k] = dims[k];
    return push(b, op);
}
int x3d_lazy_fft(x3d_backend *b, int which, void *poisson, real_t *f)
{
    LOp op;
    op.kind = which == 0 ? L_FFT_FWD : which == 1 ? L_FFT_POST000 : which == 3 ? L_FFT_POST010 : L_FFT_BWD;
    op.obj = poisson; op.o[0] = f;
    return push(b, op);
}

// ---------------------------------------------------------------- C ABI
// X3D_LAZY_REPORT=1: the counters of x3d_lazy_stats on stderr when the process ends (a host side that never asks for
// them -- the Fortran shim under mpirun -- still shows whether the rewrites engaged)
static bool g_report_all = false;
static void report_at_exit()
{
    static const char *nm[ST_N] = {"recorded", "launches", "aliases", "transeq_acc", "pairs", "tds_acc", "lincombs", "tds_lincomb",
                                   "solve000", "out_of_place", "materialised", "sync_copies", "flushes", "dropped",
                                   "transeq_x_update", "pool_slot", "zfirst", "declined", "transeq_stage"};
    for (size_t r = 0; r < g_reported.size(); r++) {
        const x3d_lazy *L = g_reported[r];
        char line[1024];
        const long slow = L->stats[ST_DECLINED] + L->stats[ST_MATERIALISE];
        if (slow > 0) {  // whatever X3D_LAZY_REPORT says: a sequence the rewrites did not recognise must not pass silently
            int n = snprintf(line, sizeof line, "x3d_lazy WARNING pid %d: %ld operation(s) of the recorded call sequence ran unfused "
                             "(%ld sum_intox / transeq without a matching rewrite, %ld reorder / veccopy as real copies): results are "
                             "the call-by-call ones, only slower -- X3D_LAZY_REPORT=1 prints every counter\n", (int)getpid(), slow,
                             L->stats[ST_DECLINED], L->stats[ST_MATERIALISE]);
            if (write(2, line, (size_t)n) < 0) {}
        }
        if (!g_report_all) continue;
        int n = snprintf(line, sizeof line, "x3d_lazy_report pid %d: three_in_one=%ld halo_forms=%ld", (int)getpid(),
                         g_reported_b[r]->n_tq3, g_reported_b[r]->n_halo);
        for (int k = 0; k < ST_N && n < (int)sizeof line - 40; k++) n += snprintf(line + n, sizeof line - n, " %s=%ld", nm[k], L->stats[k]);
        n += snprintf(line + n, sizeof line - n, "\n");
        if (write(2, line, (size_t)n) < 0) {}  // (one write per rank: the ranks of an mpirun share the stream)
    }
}

extern "C" int x3d_lazy_enable(x3d_backend *b, int on)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "x3d_lazy_enable: null backend");
    x3d_lazy *L = lazy_of(b);
    if (on && std::find(g_reported.begin(), g_reported.end(), L) == g_reported.end()) {
        const char *rep = getenv("X3D_LAZY_REPORT");
        g_report_all = rep && rep[0] == '1';
        if (g_reported.empty()) atexit(report_at_exit);
        g_reported.push_back(L);
        g_reported_b.push_back(b);
    }
    if (!on && L->on) {
        if (int rc = x3d_lazy_sync_c(b)) return rc;
    }
    L->on = on != 0;
    const char *e = getenv("X3D_LAZY_KEEP_ZERO_TERMS");
    L->keep_zero_terms = e && e[0] == '1';
    const char *r = getenv("X3D_LAZY_RULES");
    L->rules = r ? (unsigned)strtoul(r, nullptr, 0) : ~0u;
    return 0;
}

// A direction decomposed over ranks: its transeq can still be RECORDED (x3d_transeq while the mode is on) -- and fold its
// three sum_<d>intox into the accumulating form like a local one -- if the host says how to run it: when the queue
// executes the operation it calls fn with the buffers that hold the handles' data (accumulate = 1: du, dv, dw are
// added to).  Inside fn the library's entry points run at once, on those buffers.  dir_mask: bit d = direction d.
extern "C" int x3d_lazy_set_dist_transeq(x3d_backend *b, unsigned dir_mask, x3d_dist_transeq_fn fn, void *user)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "x3d_lazy_set_dist_transeq: null backend");
    X3D_REQUIRE(!(dir_mask & ~((1u << X3D_DIR_Y) | (1u << X3D_DIR_Z))), "x3d_lazy_set_dist_transeq: y and z only");
    x3d_lazy *L = lazy_of(b);
    if (int rc = x3d_lazy_flush_c(b)) return rc;
    L->dist_mask = fn ? dir_mask : 0u;
    L->dist_fn = fn;
    L->dist_user = user;
    return 0;
}

// ... and the same for tds_solve along a decomposed direction: recorded (x3d_tds_solve while the mode is on), paired by the
// rewrites like local solves (a = A(i1) ; b = B(i2) ; a += b -> mode 0; a = A(i) ; b = B(i) -> mode 1), executed by fn:
// mode 2: out1 = ta(in1); 0: out1 = ta(in1) + tb(in2); 1: out1 = ta(in1), out2 = tb(in1)
extern "C" int x3d_lazy_set_dist_tds(x3d_backend *b, unsigned dir_mask, x3d_dist_tds_fn fn, void *user)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "x3d_lazy_set_dist_tds: null backend");
    X3D_REQUIRE(!(dir_mask & ~((1u << X3D_DIR_Y) | (1u << X3D_DIR_Z))), "x3d_lazy_set_dist_tds: y and z only");
    x3d_lazy *L = lazy_of(b);
    if (int rc = x3d_lazy_flush_c(b)) return rc;
    L->dist_tds_mask = fn ? dir_mask : 0u;
    L->dist_tds_fn = fn;
    L->dist_tds_user = user;
    return 0;
}

extern "C" int x3d_lazy_flush(x3d_backend *b)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "x3d_lazy_flush: null backend");
    return x3d_lazy_flush_c(b);
}

extern "C" int x3d_lazy_sync(x3d_backend *b)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "x3d_lazy_sync: null backend");
    return x3d_lazy_sync_c(b);
}

extern "C" int x3d_lazy_register_block(x3d_backend *b, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_lazy_register_block: null argument");
    x3d_lazy_register(b, f);
    return 0;
}

// before memory registered above goes back to its owner: every handle's data home, then the layer forgets the block
// (it would otherwise keep treating the buffer as reusable storage)
extern "C" int x3d_lazy_unregister_block(x3d_backend *b, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_lazy_unregister_block: null argument");
    if (!b->lazy) return 0;
    if (b->lazy->on) {
        if (int rc = x3d_lazy_sync_c(b)) return rc;
    }
    x3d_lazy_unregister(b, f);
    return 0;
}

// allocator%release_block (src/allocator.f90:160-168): the block's contents are dead until it is written again
extern "C" int x3d_block_discard(x3d_backend *b, real_t *f)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && f, "x3d_block_discard: null argument");
    if (!x3d_lazy_active(b)) return 0;
    LOp op;
    op.kind = L_DISCARD; op.o[0] = f;
    return push(b, op);
}

extern "C" int x3d_lazy_stats(x3d_backend *b, long out[24])
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && out, "x3d_lazy_stats: null argument");
    for (int k = 0; k < 24; k++) out[k] = 0;
    if (!b->lazy) return 0;
    for (int k = 0; k < ST_N; k++) out[k] = b->lazy->stats[k];
    out[15] = (long)b->lazy->pool.size();
    return 0;
}
