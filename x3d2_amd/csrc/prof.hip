// Per-kernel timing with HIP events on the stream the kernels are launched on
// (bench.py's roofline numbers come from here; rocprofv3 --kernel-trace is the
// cross-check committed under profiles/).
#include "common.h"

#include <dlfcn.h>

// ---- roctx ranges, resolved at run time (common.h, X3D_RANGE)
static struct {
    int state;  // 0: not tried, 1: on, -1: off
    int (*push)(const char *);
    int (*pop)();
} g_roctx;

bool x3d_roctx_on()
{
    if (g_roctx.state == 0) {
        g_roctx.state = -1;
        const char *e = getenv("X3D_ROCTX");
        if (e && e[0] == '1') {
            const char *libs[3] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so"};
            for (const char *l : libs) {
                void *h = dlopen(l, RTLD_NOW | RTLD_GLOBAL);
                if (!h) continue;
                g_roctx.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (g_roctx.push && g_roctx.pop) { g_roctx.state = 1; break; }
            }
            if (g_roctx.state != 1) fprintf(stderr, "x3d2_hip: X3D_ROCTX=1 but no roctx library could be opened: no ranges\n");
        }
    }
    return g_roctx.state == 1;
}
void x3d_roctx_push(const char *name) { g_roctx.push(name); }
void x3d_roctx_pop() { g_roctx.pop(); }

struct x3d_prof {
    static const int POOL = 2048;
    hipEvent_t e0[POOL], e1[POOL];
    int id[POOL];
    int used;
    double total_ms[X3D_K_NKINDS * 4];
    long count[X3D_K_NKINDS * 4];
};

static void prof_drain(x3d_backend *b)
{
    x3d_prof *p = b->prof;
    if (!p->used) return;
    hipEventSynchronize(p->e1[p->used - 1]);
    for (int i = 0; i < p->used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->e0[i], p->e1[i]) == hipSuccess) {
            p->total_ms[p->id[i]] += ms;
            p->count[p->id[i]]++;
        }
    }
    p->used = 0;
}

void x3d_prof_begin(x3d_backend *b, int kind, int dir)
{
    x3d_prof *p = b->prof;
    if (p->used == x3d_prof::POOL) prof_drain(b);
    p->id[p->used] = kind * 4 + (dir & 3);
    hipEventRecord(p->e0[p->used], b->stream);
}

void x3d_prof_end(x3d_backend *b)
{
    x3d_prof *p = b->prof;
    hipEventRecord(p->e1[p->used], b->stream);
    p->used++;
}

extern "C" int x3d_prof_enable(x3d_backend *b, int on)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "null backend");
    if (on && !b->prof) {
        x3d_prof *p = new x3d_prof();
        memset(p, 0, sizeof *p);
        for (int i = 0; i < x3d_prof::POOL; i++) {
            X3D_HIP(hipEventCreate(&p->e0[i]));
            X3D_HIP(hipEventCreate(&p->e1[i]));
        }
        b->prof = p;
    } else if (!on && b->prof) {
        prof_drain(b);
        for (int i = 0; i < x3d_prof::POOL; i++) { hipEventDestroy(b->prof->e0[i]); hipEventDestroy(b->prof->e1[i]); }
        delete b->prof;
        b->prof = nullptr;
    }
    return 0;
}

// which kernel classes are timed while the timers are on: bit k = class X3D_K_* (default: all).  A benchmark that
// only needs the dominant class keeps the event records of the others out of its timed region.
extern "C" int x3d_prof_select(x3d_backend *b, unsigned mask)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "null backend");
    if (b->prof) prof_drain(b);
    b->prof_mask = mask;
    return 0;
}

extern "C" int x3d_prof_reset(x3d_backend *b)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b, "null backend");
    if (!b->prof) return 0;
    prof_drain(b);
    memset(b->prof->total_ms, 0, sizeof b->prof->total_ms);
    memset(b->prof->count, 0, sizeof b->prof->count);
    return 0;
}

// kind/dir -> launches and summed device time; dir = 0 sums over directions
extern "C" int x3d_prof_get(x3d_backend *b, int kind, int dir, long *count, double *total_ms)
{
    X3D_RANGE(__func__);
    X3D_REQUIRE(b && count && total_ms, "null argument");
    X3D_REQUIRE(kind >= 0 && kind < X3D_K_NKINDS && dir >= 0 && dir <= 3, "bad kind/dir");
    *count = 0; *total_ms = 0.0;
    if (!b->prof) return 0;
    prof_drain(b);
    for (int d = 0; d < 4; d++)
        if (dir == 0 || d == dir) { *count += b->prof->count[kind * 4 + d]; *total_ms += b->prof->total_ms[kind * 4 + d]; }
    return 0;
}

int x3d_prof_enable_c(x3d_backend *b, int on) { return x3d_prof_enable(b, on); }
