"""hip_backend_t: the MI355X backend behind x3d2's operator interface.

Mirrors the reference's abstract `base_backend_t`
(/root/reference/src/backend/backend.f90:13-62) operation for operation --
same names, argument order and meaning, same `data_loc` bookkeeping and the
same precondition failures (raised instead of `error stop`) as the OpenMP
implementation (src/backend/omp/backend.f90).  All arithmetic on field data
happens in libx3d2_hip.so through the C ABI (include/x3d2_hip.h); this class
only sequences calls and does the neighbour exchanges.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .common import (CELL, DIR_C, DIR_X, DIR_Y, DIR_Z, N_HALO, NULL_LOC, VERT, X_FACE, Y_FACE, Z_FACE, X3dError,
                     get_rdr_from_dirs, move_data_loc)
from .field import Allocator
from .parallel import Comm
from .tdsops import Tdsops

VP = ctypes.c_void_p


def _dp(a):
    """pointer to `a` in the library's real kind (a float64 host array is converted for the FP32 flavour; the pointer
    object keeps the converted array alive for the call)"""
    if a.dtype != np.dtype(_lib.NP_REAL):
        a = np.ascontiguousarray(a, dtype=_lib.NP_REAL)
    return a.ctypes.data_as(_lib.c_double_p)


class HipBackend:
    n_halo = N_HALO  # src/backend/backend.f90:28-29

    def __init__(self, mesh, device=None, comm=None, lazy=None):
        """lazy (default: X3D_LAZY=1): the library's deferred execution (csrc/lazy.hip) -- the op-granular calls below
        are recorded, rewritten onto the fused kernels and run when a result has to be visible.  This is what the
        Fortran shim switches on for the unchanged solver.f90; one rank, op-granular driver (SolverConfig(fused=False))."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise X3dError("x3d2_amd: no HIP device available -- the HIP backend has no CPU path")
        self.mesh = mesh
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        torch.cuda.set_device(self.device)
        self.comm = comm if comm is not None else Comm()
        self.stream = torch.cuda.current_stream(self.device)
        h = VP()
        dims = _lib.ints(*mesh.vert_dims)
        _lib.check(self.lib.x3d_backend_create(ctypes.byref(h), dims, self.device.index,
                                               VP(self.stream.cuda_stream)))
        self.h = h
        self.nblock = int(self.lib.x3d_block_elems(h))
        pd = _lib.ints(0, 0, 0)
        _lib.check(self.lib.x3d_padded_dims(h, pd))
        self.padded_dims = tuple(pd)
        self.allocator = Allocator(self.nblock, self.device)
        # several ranks with exchanges that are meant to run BESIDE kernels (the overlapped path of parallel.Comm): the
        # persistent tile / scan kernels can leave X3D_COMM_RESERVE_CUS CUs free for the kernels of other streams
        # (csrc/common.h, comm_reserve)
        c = self.comm
        multi = (c.size > 1 and not c.host_staged) or getattr(c, "self_via_nccl", False)
        # (default 0: with the emulated exchanges of bench.py --virtual-ranks -- one-wave kernels and device copies on the
        #  communication stream -- 8 or 16 reserved CUs changed nothing, profiles/r05_yslab_pipeline_timeline.txt; kept as
        #  a switch for the first run over real links, where RCCL's kernels want more room)
        del multi
        self.comm_reserve = int(os.environ.get("X3D_COMM_RESERVE_CUS", "0"))
        _lib.check(self.lib.x3d_backend_set_comm_reserve(h, self.comm_reserve))
        # a decomposed direction that is periodic over all its ranks: its single-pass HALO kernels may take the open-ended
        # circulant solve (x3d_backend_set_ring; every rank holds the same mesh, so every rank says the same)
        emul = os.environ.get("X3D_EMULATE_DECOMP", "").lower()
        for d in (2, 3):
            split = int(mesh.nproc_dir[d - 1]) > 1 or "xyz"[d - 1] in emul
            _lib.check(self.lib.x3d_backend_set_ring(h, d, int(split and bool(mesh.periodic_BC[d - 1]))))
        self.lazy = (os.environ.get("X3D_LAZY") == "1") if lazy is None else bool(lazy)
        self.red_epoch = 0  # bumped by every call that uses the library's reduction buffer (Solver.take_mean_shift)
        if self.lazy:
            # several ranks (round 4): the distributed entry points flush the queue and run at once on the buffers that
            # hold their handles' data; the local directions keep their rewrites.  (One process standing in for several
            # ranks, X3D_EMULATE_DECOMP, exchanges by slicing block memory directly: not with handles.)
            if os.environ.get("X3D_EMULATE_DECOMP"):
                raise X3dError("deferred execution (lazy) does not serve X3D_EMULATE_DECOMP")
            _lib.check(self.lib.x3d_lazy_enable(h, 1))
            self.allocator.lazy = (self.lib, h)
        self.poisson_fft = None
        self._halo = {}
        self._tdsops = []
        self.before_read = []  # callables run before field data leaves the device (Solver.flush_grad)
        self._emulate = os.environ.get("X3D_EMULATE_DECOMP", "").lower().replace("x", "")
        self.halo_launches = 0  # single-pass launches on decomposed directions (tests assert the path engaged)

    def __del__(self):
        try:
            self.allocator.destroy()  # (deferred execution: the layer forgets the blocks torch owns before they go)
        except Exception:
            pass
        try:
            for t in self._tdsops:
                self.lib.x3d_tdsops_destroy(t)
            self.lib.x3d_backend_destroy(self.h)
        except Exception:
            pass

    # ------------------------------------------------------------ helpers
    def _dims(self, data_loc):
        return _lib.ints(*self.mesh.get_dims(data_loc))

    def _decomposed(self, direction):
        """X3D_EMULATE_DECOMP=z (or yz): treat these directions as decomposed although this rank owns them whole
        (its neighbours are itself, every exchange a device copy): the code path and the kernels of an N > 1 run
        in ONE process, which is how their local cost is measured on a one-GPU box"""
        if "xyz"[direction - 1] in getattr(self, "_emulate", ""):
            return True
        return int(self.mesh.nproc_dir[direction - 1]) > 1

    def _buffers(self, direction, rows, tag):
        """exchange buffers [rows][npencil], cached per (dir, rows, tag)
        (the reference sizes them once in the constructor,
        src/backend/omp/backend.f90:84-112)"""
        key = (direction, rows, tag)
        if key not in self._halo:
            npn = self.lib.x3d_npencils(self.h, direction)
            self._halo[key] = tuple(torch.zeros(rows * npn, dtype=_lib.torch_real(), device=self.device)
                                    for _ in range(4))
        return self._halo[key]

    def sync(self):
        _lib.check(self.lib.x3d_device_sync(self.h))

    LAZY_STATS = ("recorded", "launched", "aliases", "transeq_acc", "pairs", "tds_acc", "lincombs", "tds_lincomb",
                  "solve_000", "out_of_place", "materialised", "sync_copies", "flushes", "dropped", "transeq_upd",
                  "extra_buffers", "zfirst", "declined", "transeq_stage")

    def lazy_stats(self):
        """counters of the deferred-execution layer (x3d_lazy_stats)"""
        out = (ctypes.c_long * 24)()
        _lib.check(self.lib.x3d_lazy_stats(self.h, out))
        return dict(zip(self.LAZY_STATS, [int(v) for v in out]))

    # ------------------------------------------------------------ per-kernel timers
    KINDS = {"transeq_fwd": 0, "transeq_bwd": 1, "tds_fwd": 2, "tds_bwd": 3, "blas1": 4, "copy": 5,
             "reduce": 6, "fft": 7, "spectral": 8, "pack": 9}

    def prof_enable(self, on=True):
        _lib.check(self.lib.x3d_prof_enable(self.h, int(on)))

    def prof_select(self, kinds=None):
        """time only these kernel classes while the timers are on (None: all)"""
        mask = 0xFFFFFFFF if kinds is None else sum(1 << self.KINDS[k] for k in kinds)
        _lib.check(self.lib.x3d_prof_select(self.h, mask))

    def prof_reset(self):
        _lib.check(self.lib.x3d_prof_reset(self.h))

    def prof_get(self, kind, direction=0):
        """(launch count, summed device ms) of one kernel class"""
        n, ms = ctypes.c_long(), ctypes.c_double()  # (timers: FP64 whatever the real kind)
        _lib.check(self.lib.x3d_prof_get(self.h, self.KINDS[kind], direction, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    # ------------------------------------------------------------ alloc_tdsops
    def alloc_tdsops(self, n_tds, delta, operation, scheme, bc_start, bc_end, stretch=None,
                     stretch_correct=None, n_halo=None, from_to=None, sym=None, c_nu=None, nu0_nu=None):
        """src/backend/backend.f90:352-372: host factory + device copy"""
        t = Tdsops(n_tds, delta, operation, scheme, bc_start, bc_end, stretch, stretch_correct, n_halo,
                   from_to, sym, c_nu, nu0_nu)
        h = VP()
        cs = np.ascontiguousarray(t.coeffs_s.reshape(-1))
        ce = np.ascontiguousarray(t.coeffs_e.reshape(-1))
        _lib.check(self.lib.x3d_tdsops_create(
            self.h, ctypes.byref(h), t.n_tds, t.n_rhs, t.move, int(t.periodic), _dp(t.coeffs), _dp(cs),
            _dp(ce), _dp(t.dist_fw), _dp(t.dist_bw), _dp(t.dist_sa), _dp(t.dist_sc), _dp(t.dist_af),
            _dp(t.stretch), _dp(t.stretch_correct)))
        t.handle = h
        self._tdsops.append(h)
        if t.pentadiag:  # compact10_penta: the pentadiagonal LU tables + what the ghost rows are
            if t.bc_start != t.bc_end:
                raise X3dError("compact10_penta: the same boundary condition is needed at both ends")
            kind = {0: 1, 1: 2 if t.sym else 3, 2: 4}.get(t.bc_start)
            if kind is None:
                raise X3dError("compact10_penta: periodic, Neumann or Dirichlet ends (not BC_HALO: the solve is rank-local)")
            _lib.check(self.lib.x3d_tdsops_set_penta(h, t.alpha, t.beta, t.beta_lhs_s, _dp(t.dist_fw), _dp(t.dist_af),
                                                     _dp(t.dist_sa), _dp(t.dist_bw), _dp(cs), _dp(ce), kind))
        return t

    # ------------------------------------------------------------ transeq
    def transeq_x(self, du, dv, dw, u, v, w, nu, dirps):
        self._transeq(DIR_X, du, dv, dw, u, v, w, nu, dirps)

    def transeq_y(self, du, dv, dw, u, v, w, nu, dirps):
        self._transeq(DIR_Y, du, dv, dw, u, v, w, nu, dirps)

    def transeq_z(self, du, dv, dw, u, v, w, nu, dirps):
        self._transeq(DIR_Z, du, dv, dw, u, v, w, nu, dirps)

    def _transeq(self, direction, du, dv, dw, u, v, w, nu, dirps):
        if dirps.dir != direction:
            raise X3dError("transeq: dirps%dir does not match the call")
        # mesh%get_n error-stops on NULL_LOC (src/mesh.f90:274-305); it is
        # evaluated by transeq_halo_exchange (src/backend/omp/backend.f90:272)
        self.mesh.get_n(direction, u.data_loc)
        if not self._decomposed(direction):
            _lib.check(self.lib.x3d_transeq(self.h, direction, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr,
                                            float(nu), dirps.der1st.handle, dirps.der1st_sym.handle,
                                            dirps.der2nd.handle, dirps.der2nd_sym.handle))
        else:
            self._transeq_dist(direction, du, dv, dw, u, v, w, nu, dirps)
        for r in (du, dv, dw):
            r.set_data_loc(u.data_loc)  # :333

    def _transeq_dist(self, direction, du, dv, dw, u, v, w, nu, dirps, accumulate=False):
        """transeq_omp_dist with the permutation of :145-184; accumulate: d* += result (fused driver)"""
        if self.halo_tile_ok(direction, dirps):  # single-pass kernels + boundary-strip correction
            h = self.transeq_halo_begin(direction, u, v, w)
            hb = self.transeq_halo_main(direction, du, dv, dw, u, v, w, nu, dirps, accumulate, h)
            self.transeq_halo_finish(direction, du, dv, dw, u, v, w, nu, dirps, hb)
            return
        if direction == DIR_X:
            rhs, fld = (du, dv, dw), (u, v, w)
        elif direction == DIR_Y:
            rhs, fld = (dv, du, dw), (v, u, w)
        else:
            rhs, fld = (dw, du, dv), (w, u, v)
        d = direction - 1
        prev, nxt = int(self.mesh.pprev[d]), int(self.mesh.pnext[d])
        n = self.mesh.get_n(direction, u.data_loc)
        halos = []
        pairs = []
        for i, f in enumerate(fld):  # transeq_halo_exchange, :264-297
            ss, se, rs, re = self._buffers(direction, N_HALO, "u%d" % i)
            _lib.check(self.lib.x3d_pack_halos(self.h, ss.data_ptr(), se.data_ptr(), f.ptr, n, direction))
            pairs.append((ss, se, rs, re))
            halos.append((rs, re))
        self.comm.sendrecv(pairs, prev, nxt)  # stream-ordered (RCCL) or self-synchronising (host staged)
        ops = [(dirps.der1st, dirps.der1st_sym, dirps.der2nd),
               (dirps.der1st_sym, dirps.der1st, dirps.der2nd_sym),
               (dirps.der1st_sym, dirps.der1st, dirps.der2nd_sym)]
        bs, be, brs, bre = self._buffers(direction, 3, "b")
        for i in range(3):
            t_du, t_dud, t_d2u = ops[i]
            _lib.check(self.lib.x3d_transeq_dist_fwd(
                self.h, direction, rhs[i].ptr, bs.data_ptr(), be.data_ptr(), fld[i].ptr,
                halos[i][0].data_ptr(), halos[i][1].data_ptr(), fld[0].ptr, halos[0][0].data_ptr(),
                halos[0][1].data_ptr(), t_du.handle, t_dud.handle, t_d2u.handle))
            self.comm.sendrecv([(bs, be, brs, bre)], prev, nxt)
            _lib.check(self.lib.x3d_transeq_dist_bwd_acc(
                self.h, direction, rhs[i].ptr, bs.data_ptr(), brs.data_ptr(), bre.data_ptr(), fld[0].ptr,
                float(nu), t_du.handle, t_dud.handle, t_d2u.handle, int(bool(accumulate))))

    def transeq_species(self, dspec, uvw, spec, nu, dirps, sync=True, accumulate=False, direction=None):
        """src/backend/omp/backend.f90:186-233: convection-diffusion of one transported scalar along
        dirps%dir.  `sync` (halo exchange of the momentum field needed) is honoured implicitly: the halos of
        uvw are packed whenever the direction is decomposed.  direction / accumulate: fused-driver form on
        blocks of any tag."""
        d = dirps.dir if direction is None else direction
        self.mesh.get_n(d, spec.data_loc)
        if not self._decomposed(d):
            _lib.check(self.lib.x3d_transeq_species(self.h, d, dspec.ptr, uvw.ptr, spec.ptr, float(nu),
                                                    dirps.der1st.handle, dirps.der1st_sym.handle,
                                                    dirps.der2nd.handle, int(accumulate)))
        else:
            out = dspec if not accumulate else self.allocator.get_block(DIR_X)
            i = d - 1
            prev, nxt = int(self.mesh.pprev[i]), int(self.mesh.pnext[i])
            n = self.mesh.get_n(d, spec.data_loc)
            pairs, halos = [], []
            for k, f in enumerate((spec, uvw)):
                ss, se, rs, re = self._buffers(d, N_HALO, "u%d" % k)
                _lib.check(self.lib.x3d_pack_halos(self.h, ss.data_ptr(), se.data_ptr(), f.ptr, n, d))
                pairs.append((ss, se, rs, re))
                halos.append((rs, re))
            self.comm.sendrecv(pairs, prev, nxt)
            bs, be, brs, bre = self._buffers(d, 3, "b")
            t = (dirps.der1st, dirps.der1st_sym, dirps.der2nd)
            _lib.check(self.lib.x3d_transeq_dist_fwd(
                self.h, d, out.ptr, bs.data_ptr(), be.data_ptr(), spec.ptr, halos[0][0].data_ptr(),
                halos[0][1].data_ptr(), uvw.ptr, halos[1][0].data_ptr(), halos[1][1].data_ptr(), t[0].handle,
                t[1].handle, t[2].handle))
            self.comm.sendrecv([(bs, be, brs, bre)], prev, nxt)
            _lib.check(self.lib.x3d_transeq_dist_bwd(
                self.h, d, out.ptr, bs.data_ptr(), brs.data_ptr(), bre.data_ptr(), uvw.ptr, float(nu),
                t[0].handle, t[1].handle, t[2].handle))
            if accumulate:
                _lib.check(self.lib.x3d_vecadd(self.h, 1.0, out.ptr, 1.0, dspec.ptr))
                self.allocator.release_block(out)
        if not accumulate:
            dspec.set_data_loc(spec.data_loc)

    # ------------------------------------------------------------ decomposed directions, single pass
    # exec_dist_tds_compact / exec_dist_transeq_compact (src/backend/omp/exec_dist.f90:16-186) on the tile
    # kernels' HALO forms (csrc/xscan.hip): begin = pack the 4 + 4 boundary rows and start their exchange,
    # main = the whole local solve in one kernel (waits for the rows; starts the exchange of its boundary
    # values), finish = the contribution of the neighbours' boundary values on the boundary strips.  The
    # handles let the caller put independent kernels between the three calls (parallel.Comm.isendrecv).
    def _halo_buffers(self, direction, nf, nb, tag):
        key = ("halo", direction, nf, nb, tag)
        if key not in self._halo:
            npn = self.lib.x3d_npencils(self.h, direction)
            hrow = int(self.lib.x3d_halo_row_size(self.h, direction))
            z = lambda n: torch.zeros(n, dtype=_lib.torch_real(), device=self.device)
            # (z halos leave straight from the fields' blocks: no send buffer)
            hs = z(0) if self._halo_direct(direction) else z(2 * nf * N_HALO * hrow)
            self._halo[key] = (hs, z(2 * nf * N_HALO * hrow), z(2 * nb * npn), z(2 * nb * npn))
        return self._halo[key]

    def _halo_direct(self, direction):
        """a z halo row is one xy plane in the block's own layout (x3d_halo_row_size): rows 1..4 and n-3..n of a
        field are contiguous pieces of its block and are sent from there (X3D_PACK_Z_HALOS=1: through the pack
        kernel like y, for A/B runs)"""
        # (deferred execution: a block address is a handle, its rows may live in another buffer -- always pack)
        return direction == DIR_Z and os.environ.get("X3D_PACK_Z_HALOS") != "1" and not self.lazy

    def _halo_exchange(self, direction, fields, n, hs, hr):
        """start the exchange of the boundary rows 1..4 / n-3..n of `fields` with the two neighbours of the
        direction; they arrive in hr[side][field][4][row]"""
        nf = len(fields)
        if not self._halo_direct(direction):
            ptrs = (VP * nf)(*[f.ptr for f in fields])
            _lib.check(self.lib.x3d_pack_halos_multi(self.h, hs.data_ptr(), ptrs, nf, n, direction))
            return self.comm.isendrecv([self._halves(hs) + self._halves(hr)], *self._neighbours(direction))
        hrow = int(self.lib.x3d_halo_row_size(self.h, direction))
        m = N_HALO * hrow
        pairs = [(f.data[:m], f.data[(n - N_HALO) * hrow:n * hrow], hr[i * m:(i + 1) * m],
                  hr[(nf + i) * m:(nf + i + 1) * m]) for i, f in enumerate(fields)]
        return self.comm.isendrecv(pairs, *self._neighbours(direction))

    @staticmethod
    def _halves(t):
        h = t.numel() // 2
        return t[:h], t[h:]

    def _neighbours(self, direction):
        d = direction - 1
        return int(self.mesh.pprev[d]), int(self.mesh.pnext[d])

    @staticmethod
    def _component_order(direction, a, b, c):
        """the advecting component first (src/backend/omp/backend.f90:145-184)"""
        return (a, b, c) if direction == DIR_X else ((b, a, c) if direction == DIR_Y else (c, a, b))

    def halo_tile_ok(self, direction, dirps):
        """the single-pass kernels take this decomposed direction's transeq (probe: a launch over zero planes)"""
        if direction == DIR_X or not self._decomposed(direction) or os.environ.get("X3D_NO_HALO_TILE") == "1":
            return False
        key = ("tq_ok", direction, id(dirps))
        if key not in self._halo:
            hs, hr, bs, br = self._halo_buffers(direction, 3, 9, "tq")
            flag = ctypes.c_int(0)
            p = hr.data_ptr()  # (any valid device address: nothing is launched)
            _lib.check(self.lib.x3d_transeq_tile(self.h, direction, p, p + 8, p + 16, p + 24, p + 32, p + 40, 0.0,
                                                 dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                                 dirps.der2nd_sym.handle, 1, hr.data_ptr(), bs.data_ptr(), 0, 0,
                                                 ctypes.byref(flag)))
            self._halo[key] = bool(flag.value)
        return self._halo[key]

    def transeq_halo_begin(self, direction, u, v, w):
        f = self._component_order(direction, u, v, w)
        hs, hr, _, _ = self._halo_buffers(direction, 3, 9, "tq")
        return self._halo_exchange(direction, f, self.mesh.get_n(direction, u.data_loc), hs, hr)

    def transeq_halo_main(self, direction, du, dv, dw, u, v, w, nu, dirps, accumulate, handle):
        _, hr, bs, br = self._halo_buffers(direction, 3, 9, "tq")
        handle.wait()
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_tile(self.h, direction, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr, float(nu),
                                             dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                             dirps.der2nd_sym.handle, int(bool(accumulate)), hr.data_ptr(),
                                             bs.data_ptr(), 0, -1, ctypes.byref(flag)))
        if not flag.value:
            raise X3dError("transeq_halo_main: tile kernel refused (halo_tile_ok was not consulted)")
        self.halo_launches += 1
        return self.comm.isendrecv([self._halves(bs) + self._halves(br)], *self._neighbours(direction))

    def transeq_halo_finish(self, direction, du, dv, dw, u, v, w, nu, dirps, handle):
        _, _, _, br = self._halo_buffers(direction, 3, 9, "tq")
        handle.wait()
        _lib.check(self.lib.x3d_transeq_halo_fix(self.h, direction, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr,
                                                 float(nu), dirps.der1st.handle, dirps.der2nd.handle, br.data_ptr()))

    def transeq_planes(self, direction, du, dv, dw, u, v, w, nu, dirps, accumulate, other0, nother):
        """transeq_<dir> of a LOCAL y / z direction over the planes [other0, other0 + nother) of the coordinate
        its tiles are stacked along (z for y pencils); False: these pencils are not served by the tile kernel"""
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_tile(self.h, direction, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr, float(nu),
                                             dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                             dirps.der2nd_sym.handle, int(bool(accumulate)), None, None, int(other0),
                                             int(nother), ctypes.byref(flag)))
        return bool(flag.value)

    # jobs: (mode, out1, out2, in1, in2, ta, tb) with the modes of x3d_tds_pair_tile
    def _job_sizes(self, job):
        return (2 if job[0] == 0 else 1), (1 if job[0] == 2 else 2)

    def tds_tile_ok(self, direction, job, halo, yperm=0):
        """the tile kernel takes this job (probe: zero planes); yperm > 0: with that many y rows interleaved on
        the way (the halo form of tds_pair_yperm: z pairs next to the 010 Poisson solve on z slabs)"""
        mode, out1, out2, in1, in2, ta, tb = job
        key = ("tds_ok", direction, mode, id(ta), id(tb), bool(halo), int(yperm))
        if key not in self._halo:
            if os.environ.get("X3D_NO_HALO_TILE") == "1" and halo:
                self._halo[key] = False
            elif yperm:
                ok = (halo and direction == DIR_Z and mode in (0, 1) and os.environ.get("X3D_NO_YPERM") != "1"
                      and self.tds_tile_ok(direction, job, True))
                if ok:  # (the probe of the interleaving form needs whole blocks: ask the launcher's own conditions)
                    ok = 0 < yperm <= int(self.mesh.vert_dims[1]) and yperm % 2 == 0
                self._halo[key] = bool(ok)
            else:
                flag = ctypes.c_int(0)
                hp = bp = None
                if halo:
                    nf, nb = self._job_sizes(job)
                    _, hr, bs, _ = self._halo_buffers(direction, nf, nb, "job0")
                    hp, bp = hr.data_ptr(), bs.data_ptr()
                _lib.check(self.lib.x3d_tds_pair_tile(
                    self.h, direction, mode, out1.ptr, out2.ptr if out2 is not None else None, in1.ptr,
                    in2.ptr if in2 is not None else None, ta.handle, tb.handle if tb is not None else None, hp, bp, 0, 0,
                    ctypes.byref(flag)))
                self._halo[key] = bool(flag.value)
        return self._halo[key]

    def tds_tile_planes(self, direction, job, other0, nother):
        """one job of a LOCAL direction over a range of planes (tile kernel)"""
        mode, out1, out2, in1, in2, ta, tb = job
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_tds_pair_tile(
            self.h, direction, mode, out1.ptr, out2.ptr if out2 is not None else None, in1.ptr,
            in2.ptr if in2 is not None else None, ta.handle, tb.handle if tb is not None else None, None, None,
            int(other0), int(nother), ctypes.byref(flag)))
        if not flag.value:
            raise X3dError("tds_tile_planes: tile kernel refused (tds_tile_ok was not consulted)")

    def tds_job_local(self, direction, job):
        """one job of a local direction with the default kernels (pair kernel / K1e / two-sweep)"""
        mode, out1, out2, in1, in2, ta, tb = job
        if mode == 2:
            self.tds_apply(out1, in1, ta, direction)
        else:
            self.tds_pair(mode, out1, out2, in1, in2, ta, tb, direction)

    def tds_halo_begin(self, direction, job, k):
        mode, out1, out2, in1, in2, ta, tb = job
        nf, nb = self._job_sizes(job)
        hs, hr, _, _ = self._halo_buffers(direction, nf, nb, "job%d" % k)
        return self._halo_exchange(direction, (in1, in2)[:nf], ta.n_tds, hs, hr)

    def tds_halo_main(self, direction, job, k, handle, yperm=0):
        mode, out1, out2, in1, in2, ta, tb = job
        nf, nb = self._job_sizes(job)
        _, hr, bs, br = self._halo_buffers(direction, nf, nb, "job%d" % k)
        handle.wait()
        flag = ctypes.c_int(0)
        if yperm:
            _lib.check(self.lib.x3d_tds_pair_tile_yperm(
                self.h, mode, out1.ptr, out2.ptr if out2 is not None else None, in1.ptr,
                in2.ptr if in2 is not None else None, ta.handle, tb.handle, hr.data_ptr(), bs.data_ptr(), int(yperm),
                ctypes.byref(flag)))
        else:
            _lib.check(self.lib.x3d_tds_pair_tile(
                self.h, direction, mode, out1.ptr, out2.ptr if out2 is not None else None, in1.ptr,
                in2.ptr if in2 is not None else None, ta.handle, tb.handle if tb is not None else None, hr.data_ptr(),
                bs.data_ptr(), 0, -1, ctypes.byref(flag)))
        if not flag.value:
            raise X3dError("tds_halo_main: tile kernel refused (tds_tile_ok was not consulted)")
        self.halo_launches += 1
        return self.comm.isendrecv([self._halves(bs) + self._halves(br)], *self._neighbours(direction))

    def tds_halo_finish(self, direction, job, k, handle, yperm=0):
        mode, out1, out2, in1, in2, ta, tb = job
        nf, nb = self._job_sizes(job)
        _, _, _, br = self._halo_buffers(direction, nf, nb, "job%d" % k)
        handle.wait()
        if yperm:
            _lib.check(self.lib.x3d_tds_pair_halo_fix_yperm(self.h, mode, out1.ptr,
                                                            out2.ptr if out2 is not None else None, ta.handle,
                                                            tb.handle, br.data_ptr(), int(yperm)))
            return
        _lib.check(self.lib.x3d_tds_pair_halo_fix(self.h, direction, mode, out1.ptr,
                                                  out2.ptr if out2 is not None else None, ta.handle,
                                                  tb.handle if tb is not None else None, br.data_ptr()))

    def tds_jobs(self, direction, jobs, lead=None, after_lead=None, between=None, yperm=0):
        """run tds_solve jobs of one direction.
        Local direction: lead = list of (other0, nother) plane ranges to do FIRST, after which after_lead() is
        called (it starts the next direction's halo exchange on those planes) and the remaining planes follow.
        Decomposed direction: pack + exchange the boundary rows, single-pass kernels, exchange the boundary
        values, between() (kernels that do not depend on this direction's strips), strip corrections.
        Falls back to the default kernels (tds_pair / tds_apply: two-sweep DistD2 where decomposed)."""
        if not self._decomposed(direction):
            nall = int(self.mesh.vert_dims[2] if direction == DIR_Y else self.mesh.vert_dims[1])
            if lead and direction != DIR_X and all(self.tds_tile_ok(direction, j, False) for j in jobs):
                rest, pos = [], 0
                for o0, no in sorted(lead):
                    if o0 > pos:
                        rest.append((pos, o0 - pos))
                    pos = o0 + no
                if pos < nall:
                    rest.append((pos, nall - pos))
                for rng in lead:
                    for j in jobs:
                        self.tds_tile_planes(direction, j, *rng)
                if after_lead:
                    after_lead()
                for rng in rest:
                    for j in jobs:
                        self.tds_tile_planes(direction, j, *rng)
            else:
                for j in jobs:
                    self.tds_job_local(direction, j)
                if after_lead:
                    after_lead()
            if between:
                between()
            return
        if direction != DIR_X and all(self.tds_tile_ok(direction, j, True, yperm=yperm) for j in jobs):
            hs = [self.tds_halo_begin(direction, j, k) for k, j in enumerate(jobs)]
            hb = [self.tds_halo_main(direction, j, k, hs[k], yperm=yperm) for k, j in enumerate(jobs)]
            if after_lead:
                after_lead()
            if between:
                between()
            for k, j in enumerate(jobs):
                self.tds_halo_finish(direction, j, k, hb[k], yperm=yperm)
            return
        if yperm:
            raise X3dError("tds_jobs: interleaved rows need the halo form of the z pairs (tds_tile_ok)")
        for j in jobs:  # two-sweep DistD2 with its own exchanges
            self.tds_job_local(direction, j)
        if after_lead:
            after_lead()
        if between:
            between()

    def halo_strip_rows(self, *ops):
        """(ws, we): rows 1..ws and n-we+1..n are touched by the strip corrections of these operators"""
        out = (ctypes.c_int * 2)()
        ws = we = 0
        for t in ops:
            _lib.check(self.lib.x3d_tdsops_halo_rows(t.handle, out))
            ws, we = max(ws, out[0]), max(we, out[1])
        return ws, we

    # ------------------------------------------------------------ fused-driver forms
    def transeq_dir(self, direction, du, dv, dw, u, v, w, nu, dirps, accumulate=False):
        """transeq_<dir> on blocks of ANY tag (all tags share one device layout);
        accumulate: d{u,v,w} += result -- folds reorder + sum_{y,z}intox of
        transeq_default (src/solver.f90:320-377) into the derivative pass."""
        if not self._decomposed(direction):
            _lib.check(self.lib.x3d_transeq_acc(self.h, direction, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr,
                                                float(nu), dirps.der1st.handle, dirps.der1st_sym.handle,
                                                dirps.der2nd.handle, dirps.der2nd_sym.handle, int(accumulate)))
            return
        self._transeq_dist(direction, du, dv, dw, u, v, w, nu, dirps, accumulate=accumulate)

    def transeq_x_update(self, du, dv, dw, u, v, w, nu, dirps, grads, op_u, op_vw, scale):
        """transeq_x with the pending correction u,v,w += scale * tds_solve(grads) applied inside the kernel
        (csrc/xscan.hip, k_xscan_transeq2x3<UPD>); False: not applicable, nothing was done"""
        if self._decomposed(DIR_X):
            return False  # (the kernel closes each pencil with the periodic self-exchange: local pencils only)
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_x_update(self.h, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr, float(nu),
                                                 dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                                 dirps.der2nd_sym.handle, grads[0].ptr, grads[1].ptr, grads[2].ptr,
                                                 op_u.handle, op_vw.handle, float(scale), ctypes.byref(flag)))
        return bool(flag.value)

    def transeq_x_update_rot(self, du, dv, dw, u, v, w, nu, dirps, grads, op_u, op_vw, scale, omega, u_shift=None):
        """transeq_x_update and transeq_x_rot in one launch (csrc/xwide.hip, k_xwide_transeq3_upd): pending correction,
        bulk-velocity shift, transeq_x, rotation forcing; False: not served, nothing was done"""
        if self._decomposed(DIR_X):
            return False
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_x_update_rot(self.h, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr, float(nu),
                                                     dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                                     dirps.der2nd_sym.handle, grads[0].ptr, grads[1].ptr, grads[2].ptr,
                                                     op_u.handle, op_vw.handle, float(scale), float(omega), u_shift,
                                                     ctypes.byref(flag)))
        return bool(flag.value)

    def transeq_x_rot(self, du, dv, dw, u, v, w, nu, dirps, omega, u_shift=None):
        """transeq_x with the channel's rotation forcing (du -= omega v, dv += omega u, src/case/channel.f90:191-207)
        applied inside the kernel (csrc/xwide.hip, k_xwide_transeq3<ROT>) and, with u_shift (field_mean_shift), the
        bulk-velocity shift u += *u_shift done on the way; False: not served, nothing was done"""
        if self._decomposed(DIR_X):
            return False
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_x_rot(self.h, du.ptr, dv.ptr, dw.ptr, u.ptr, v.ptr, w.ptr, float(nu),
                                              dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                              dirps.der2nd_sym.handle, float(omega), u_shift, ctypes.byref(flag)))
        return bool(flag.value)

    def transeq_dir_defer(self, direction, pend, u, v, w, nu, dirps):
        """transeq_dir(accumulate=True) with the accumulation left pending (csrc/viax.hip): the results stay in
        the blocks `pend` (pencil layout) until lincomb_pending / pending_flush.  False: not applicable here,
        nothing was done."""
        if self._decomposed(direction):
            return False
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_defer(self.h, direction, pend[0].ptr, pend[1].ptr, pend[2].ptr, u.ptr, v.ptr,
                                              w.ptr, float(nu), dirps.der1st.handle, dirps.der1st_sym.handle,
                                              dirps.der2nd.handle, dirps.der2nd_sym.handle, ctypes.byref(flag)))
        return bool(flag.value)

    def transeq_stage_ok(self, direction, dirps):
        """the tile kernel takes this direction's pencils: the last direction of transeq can be computed
        inside the RK / AB stage (transeq_lincomb)"""
        if self._decomposed(direction):
            return False
        return bool(self.lib.x3d_transeq_stage_ok(self.h, direction, dirps.der1st.handle, dirps.der1st_sym.handle,
                                                  dirps.der2nd.handle, dirps.der2nd_sym.handle))

    def transeq_lincomb(self, direction, kind, u_ptr, conv_ptr, nu, dirps, y, base, coeffs, xs, ipend, store):
        """lincomb with term xs[ipend] completed by one transeq component computed in the same kernel"""
        n = len(xs)
        c = (_lib.REAL * n)(*[float(v) for v in coeffs])
        p = (VP * n)(*[x.ptr for x in xs])
        _lib.check(self.lib.x3d_transeq_lincomb(self.h, direction, int(kind), u_ptr, conv_ptr, float(nu),
                                                dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle,
                                                dirps.der2nd_sym.handle, y.ptr, base.ptr, n, c, p, int(ipend),
                                                int(bool(store))))
        self.rk_fused_passes = getattr(self, "rk_fused_passes", 0) + n + 1
        self.rk_fused_launches = getattr(self, "rk_fused_launches", 0) + 1

    def transeq_lincomb3(self, direction, rhs, vel, nu, dirps, specs):
        """transeq_<direction> accumulated into rhs = (du, dv, dw) AND the stage's linear combination of every variable in
        the same launch (csrc/xscan.hip k_ytile_transeq3<EPI>): specs[i] = (y, base, coeffs, fields, store) with rhs[i] one of
        `fields`.  False: these pencils are not served (nothing was done)"""
        y = (VP * 3)(*[sp[0].ptr for sp in specs])
        base = (VP * 3)(*[sp[1].ptr for sp in specs])
        nterm = _lib.ints(*[len(sp[3]) for sp in specs])
        c = (_lib.REAL * 15)()
        x = (VP * 15)()
        ipend, store = [], []
        for i, (_, _, coeffs, fields, st) in enumerate(specs):
            ptrs = [f.ptr for f in fields]
            ipend.append(ptrs.index(rhs[i].ptr))
            store.append(int(bool(st)))
            for k, (cv, pv) in enumerate(zip(coeffs, ptrs)):
                c[5 * i + k] = float(cv)
                x[5 * i + k] = pv
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_transeq_lincomb3(
            self.h, direction, rhs[0].ptr, rhs[1].ptr, rhs[2].ptr, vel[0].ptr, vel[1].ptr, vel[2].ptr, float(nu),
            dirps.der1st.handle, dirps.der1st_sym.handle, dirps.der2nd.handle, dirps.der2nd_sym.handle, y, base, nterm, c, x,
            _lib.ints(*ipend), _lib.ints(*store), ctypes.byref(flag)))
        if flag.value:
            self.rk_fused_passes = getattr(self, "rk_fused_passes", 0) + sum(len(sp[3]) + 1 for sp in specs)
            self.rk_fused_launches = getattr(self, "rk_fused_launches", 0) + 1
            self.rk_in_tile3_passes = getattr(self, "rk_in_tile3_passes", 0) + sum(len(sp[3]) + 1 for sp in specs)
            self.rk_in_tile3_launches = getattr(self, "rk_in_tile3_launches", 0) + 1
        return bool(flag.value)

    def transeq_component_acc(self, direction, kind, rhs_ptr, u_ptr, conv_ptr, nu, dirps):
        """rhs += one transeq component (kind as in transeq_lincomb)"""
        ops = (dirps.der1st, dirps.der1st_sym, dirps.der2nd) if kind == 0 else \
              (dirps.der1st_sym, dirps.der1st, dirps.der2nd_sym)
        _lib.check(self.lib.x3d_transeq_species(self.h, direction, rhs_ptr, conv_ptr, u_ptr, float(nu), ops[0].handle,
                                                ops[1].handle, ops[2].handle, 1))

    def pending_flush(self, direction, r_ptr, pend):
        _lib.check(self.lib.x3d_pending_flush(self.h, direction, r_ptr, pend.ptr))

    def lincomb_pending(self, y, base, coeffs, xs, ipend, pend, direction, store):
        """lincomb with term xs[ipend] completed on the fly from its pending transeq component"""
        n = len(xs)
        c = (_lib.REAL * n)(*[float(v) for v in coeffs])
        p = (VP * n)(*[x.ptr for x in xs])
        _lib.check(self.lib.x3d_lincomb_pending(self.h, direction, y.ptr, base.ptr, n, c, p, int(ipend), pend.ptr,
                                                int(bool(store))))
        # field passes the linear combination itself needs (base + the other terms + y): bench.py's roofline
        self.rk_fused_passes = getattr(self, "rk_fused_passes", 0) + n + 1
        self.rk_fused_launches = getattr(self, "rk_fused_launches", 0) + 1

    def tds_lincomb(self, du, tdsops, direction, y, base, coeffs, xs, wall=None, mean_target=None):
        """y = base + sum c_i x_i (lincomb) and du = tds_solve(y) in one kernel where the pencils allow
        (csrc/xscan.hip k_xscan_tds_lin, csrc/xwide.hip k_xwide_tds_lin: y is not read back).  wall: the y faces of y
        take that field's values before the operator acts (field_set_face_from_field(y, wall, 0, Y_FACE)).
        mean_target (one rank): also field_mean_shift(y, mean_target) -- its device scalar is returned -- with the
        integral taken inside the kernel where it can be (x3d_tds_solve_lincomb_wall_mean)"""
        if self._decomposed(direction):
            self.lincomb(y, base, coeffs, xs)
            if wall is not None:
                self.field_set_face_from_field(y, wall, 0.0, Y_FACE)
            self.tds_apply(du, y, tdsops, direction)
            return self.field_mean_shift(y, mean_target) if mean_target is not None else None
        n = len(xs)
        c = (_lib.REAL * n)(*[float(v) for v in coeffs])
        p = (VP * n)(*[x.ptr for x in xs])
        if mean_target is not None and self.comm.size == 1:
            if y.data_loc == NULL_LOC:
                raise X3dError("You must set the data_loc before calling volume integral.")
            out = VP()
            ncell = float(np.prod(self.mesh.get_global_dims(CELL)))
            _lib.check(self.lib.x3d_tds_solve_lincomb_wall_mean(
                self.h, direction, du.ptr, tdsops.handle, y.ptr, base.ptr, n, c, p, wall.ptr if wall is not None else None,
                self._dims(y.data_loc), ncell, float(mean_target), ctypes.byref(out)))
            return out
        if wall is None:
            _lib.check(self.lib.x3d_tds_solve_lincomb(self.h, direction, du.ptr, tdsops.handle, y.ptr, base.ptr, n, c, p))
        else:
            _lib.check(self.lib.x3d_tds_solve_lincomb_wall(self.h, direction, du.ptr, tdsops.handle, y.ptr, base.ptr,
                                                           n, c, p, wall.ptr))

    def tds_pair(self, mode, out1, out2, in1, in2, t_a, t_b, direction):
        """two tds_solve's that share an output (mode 0: out1 = A(in1) + B(in2)) or an input
        (mode 1: out1 = A(in1), out2 = B(in1)): one kernel where the pencils allow (csrc/xscan.hip)"""
        if self._decomposed(direction):
            self.tds_apply(out1, in1, t_a, direction)
            if mode == 0:
                self.tds_apply(out1, in2, t_b, direction, accumulate=True)
            else:
                self.tds_apply(out2, in1, t_b, direction)
            return
        _lib.check(self.lib.x3d_tds_solve_pair(self.h, direction, int(mode), out1.ptr,
                                               out2.ptr if out2 is not None else None, in1.ptr,
                                               in2.ptr if in2 is not None else None, t_a.handle, t_b.handle))

    def tds_pair_yperm(self, mode, out1, out2, in1, in2, t_a, t_b, ny):
        """tds_pair along z next to the 010 Poisson solve, doing the solver's interleave of the first ny y rows on
        the way (mode 0: on out1, = enforce_periodicity_y of the result; mode 1: on in1, = undo_periodicity_y
        before the operators); False: not served for these pencils, nothing was done"""
        if self._decomposed(DIR_Z) or os.environ.get("X3D_NO_YPERM") == "1":
            return False
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_tds_solve_pair_yperm(self.h, int(mode), out1.ptr,
                                                     out2.ptr if out2 is not None else None, in1.ptr,
                                                     in2.ptr if in2 is not None else None, t_a.handle, t_b.handle,
                                                     int(ny), ctypes.byref(flag)))
        return bool(flag.value)

    def tds_pair_zfirst(self, mode, out1, out2, in1, in2, t_a, t_b):
        """tds_pair along z next to the z-first 000 Poisson solve: mode 0 leaves its result A(in1) + B(in2) in the
        solver's spectrum, z-transformed (out1, out2 unused); mode 1 takes its input p from there (in1, in2 unused);
        False: not on offer for these pencils, nothing was done"""
        if self._decomposed(DIR_Z) or self.poisson_fft is None or not hasattr(self.poisson_fft, "zfirst_ok"):
            return False
        if hasattr(self.poisson_fft, "zfirst_pair"):  # (the y-slab solver keeps the spectrum: csrc/sfftz.hip)
            return self.poisson_fft.zfirst_pair(mode, out1, out2, in1, in2, t_a, t_b)
        flag = ctypes.c_int(0)
        ptr = lambda f: f.ptr if f is not None else None
        _lib.check(self.lib.x3d_tds_pair_zfirst(self.h, self.poisson_fft.h, int(mode), ptr(out1), ptr(out2), ptr(in1),
                                                ptr(in2), t_a.handle, t_b.handle, ctypes.byref(flag)))
        return bool(flag.value)

    def zfirst_pairs_ok(self, t_a, t_b):
        """probe (nothing is launched): the z-transforming pair kernels take these two operators on this backend's blocks"""
        flag = ctypes.c_int(0)
        _lib.check(self.lib.x3d_tds_pair_zfirst_ok(self.h, t_a.handle, t_b.handle, ctypes.byref(flag)))
        return bool(flag.value)

    def tds_apply_mean(self, du, u, tdsops, direction, mean_target):
        """tds_apply(du, u) and field_mean_shift(u, mean_target) -- its device scalar is returned -- with the integral taken
        by the operator's kernel where it can be (x3d_tds_solve_mean: 1024-row x pencils); one rank, local direction"""
        if u.data_loc == NULL_LOC:
            raise X3dError("You must set the data_loc before calling volume integral.")
        out = VP()
        ncell = float(np.prod(self.mesh.get_global_dims(CELL)))
        _lib.check(self.lib.x3d_tds_solve_mean(self.h, du.ptr, u.ptr, tdsops.handle, direction, self._dims(u.data_loc), ncell,
                                               float(mean_target), ctypes.byref(out)))
        return out

    def tds_apply(self, du, u, tdsops, direction, accumulate=False, scale=1.0):
        """tds_solve with an explicit direction; accumulate: du += scale * result"""
        if not self._decomposed(direction):
            _lib.check(self.lib.x3d_tds_solve_acc(self.h, du.ptr, u.ptr, tdsops.handle, direction,
                                                  int(accumulate), float(scale)))
            return
        self._tds_dist(du, u, tdsops, direction, accumulate=accumulate, scale=scale)

    # ------------------------------------------------------------ tds_solve
    def tds_solve(self, du, u, tdsops):
        """src/backend/omp/backend.f90:340-391"""
        if u.dir != du.dir:
            raise X3dError("DIR mismatch between fields in tds_solve.")
        if u.data_loc != NULL_LOC:
            du.set_data_loc(move_data_loc(u.data_loc, u.dir, tdsops.move))
        direction = u.dir
        if direction == DIR_C:
            raise X3dError("tds_solve needs a directional field")
        if tdsops.pentadiag:
            self.tds_penta_solve(du, u, tdsops, direction)
            return
        if not self._decomposed(direction):
            _lib.check(self.lib.x3d_tds_solve(self.h, du.ptr, u.ptr, tdsops.handle, direction))
            return
        self._tds_dist(du, u, tdsops, direction)

    def tds_penta_solve(self, du, u, tdsops, direction, u_s=None, u_e=None):
        """exec_dist_penta_compact / exec_dist_penta_periodic (src/backend/omp/exec_dist.f90:188-241): the
        compact10_penta first derivative, a rank-local solve.  u_s / u_e: ghost rows [4][npencil] (device tensors)
        or None: formed in the kernel from the operator's boundary conditions."""
        if self._decomposed(direction):
            raise X3dError("compact10_penta is a rank-local solve: the direction must not be decomposed")
        _lib.check(self.lib.x3d_tds_penta_solve(self.h, du.ptr, u.ptr, tdsops.handle, direction,
                                                u_s.data_ptr() if u_s is not None else None,
                                                u_e.data_ptr() if u_e is not None else None))

    def _tds_dist(self, du, u, tdsops, direction, accumulate=False, scale=1.0):
        """tds_solve_dist (src/backend/omp/backend.f90:361-391) + exec_dist_tds_compact;
        accumulate: du += scale * result (fused driver)"""
        if not accumulate and direction != DIR_X:
            job = (2, du, None, u, None, tdsops, None)
            if self.tds_tile_ok(direction, job, True):  # single-pass kernel + boundary-strip correction
                h = self.tds_halo_begin(direction, job, 0)
                hb = self.tds_halo_main(direction, job, 0, h)
                self.tds_halo_finish(direction, job, 0, hb)
                return
        d = direction - 1
        prev, nxt = int(self.mesh.pprev[d]), int(self.mesh.pnext[d])
        ss, se, rs, re = self._buffers(direction, N_HALO, "u0")
        _lib.check(self.lib.x3d_pack_halos(self.h, ss.data_ptr(), se.data_ptr(), u.ptr, tdsops.n_tds,
                                           direction))
        self.comm.sendrecv([(ss, se, rs, re)], prev, nxt)
        bs, be, brs, bre = self._buffers(direction, 1, "b1")
        _lib.check(self.lib.x3d_tds_dist_fwd(self.h, du.ptr, bs.data_ptr(), be.data_ptr(), u.ptr,
                                             rs.data_ptr(), re.data_ptr(), tdsops.handle, direction))
        self.comm.sendrecv([(bs, be, brs, bre)], prev, nxt)
        _lib.check(self.lib.x3d_tds_dist_bwd_acc(self.h, du.ptr, bs.data_ptr(), brs.data_ptr(), bre.data_ptr(),
                                                 tdsops.handle, direction, int(bool(accumulate)), float(scale)))

    # ------------------------------------------------------------ reorder / sums
    def reorder(self, u_, u, direction):
        """src/backend/omp/backend.f90:393-452"""
        dir_from, dir_to = direction // 10, direction % 10
        if u.dir != dir_from or u_.dir != dir_to:
            raise X3dError("reorder: field directions do not match the reorder code")
        _lib.check(self.lib.x3d_reorder(self.h, u_.ptr, u.ptr, int(direction)))
        u_.set_data_loc(u.data_loc)  # :449-450

    def sum_yintox(self, u, u_):
        if u.dir != DIR_X or u_.dir != DIR_Y:
            raise X3dError("sum_yintox: u must be DIR_X and u_ DIR_Y")
        _lib.check(self.lib.x3d_sum_intox(self.h, u.ptr, u_.ptr, DIR_Y))

    def sum_zintox(self, u, u_):
        if u.dir != DIR_X or u_.dir != DIR_Z:
            raise X3dError("sum_zintox: u must be DIR_X and u_ DIR_Z")
        _lib.check(self.lib.x3d_sum_intox(self.h, u.ptr, u_.ptr, DIR_Z))

    # ------------------------------------------------------------ BLAS-1
    def veccopy(self, dst, src):
        if src.dir != dst.dir:
            raise X3dError("Called vector copy with incompatible fields")
        if dst.dir == DIR_C:
            raise X3dError("veccopy does not support DIR_C fields")
        _lib.check(self.lib.x3d_veccopy(self.h, dst.ptr, src.ptr))

    def vecadd(self, a, x, b, y):
        if x.dir != y.dir:
            raise X3dError("Called vector add with incompatible fields")
        if y.dir == DIR_C:
            raise X3dError("vecadd does not support DIR_C fields")
        _lib.check(self.lib.x3d_vecadd(self.h, float(a), x.ptr, float(b), y.ptr))

    def vecmult(self, y, x):
        if x.dir != y.dir:
            raise X3dError("Called vector multiply with incompatible fields")
        if y.dir == DIR_C:
            raise X3dError("vecmult does not support DIR_C fields")
        _lib.check(self.lib.x3d_vecmult(self.h, y.ptr, x.ptr))

    def field_scale(self, f, a):
        _lib.check(self.lib.x3d_field_scale(self.h, f.ptr, float(a)))

    def field_shift(self, f, a):
        _lib.check(self.lib.x3d_field_shift(self.h, f.ptr, float(a)))

    def _from_gradients(self, fn, field_out, grads):
        if len(grads) != 9:
            raise X3dError("nine gradient fields expected")
        p = (VP * 9)(*[g.ptr for g in grads])
        _lib.check(fn(self.h, field_out.ptr, p))

    def compute_vorticity(self, field_out, dudx, dudy, dudz, dvdx, dvdy, dvdz, dwdx, dwdy, dwdz):
        """|curl u| from the nine velocity gradients (src/backend/omp/backend.f90:616-630)"""
        self._from_gradients(self.lib.x3d_compute_vorticity, field_out,
                             (dudx, dudy, dudz, dvdx, dvdy, dvdz, dwdx, dwdy, dwdz))

    def compute_qcriterion(self, field_out, dudx, dudy, dudz, dvdx, dvdy, dvdz, dwdx, dwdy, dwdz):
        """Q-criterion (src/backend/omp/backend.f90:632-649)"""
        self._from_gradients(self.lib.x3d_compute_qcriterion, field_out,
                             (dudx, dudy, dudz, dvdx, dvdy, dvdz, dwdx, dwdy, dwdz))

    def lincomb(self, y, base, coeffs, xs):
        """extension: y = base + sum c_i x_i in one pass (time-integrator fusion)"""
        n = len(xs)
        c = (_lib.REAL * n)(*[float(v) for v in coeffs])
        p = (VP * n)(*[x.ptr for x in xs])
        _lib.check(self.lib.x3d_lincomb(self.h, y.ptr, base.ptr, n, c, p))

    # ------------------------------------------------------------ reductions
    def scalar_product(self, x, y):
        """src/backend/omp/backend.f90:651-712"""
        if x.data_loc == NULL_LOC or y.data_loc == NULL_LOC:
            raise X3dError("You must set the data_loc before calling scalar product")
        if x.data_loc != y.data_loc:
            raise X3dError("Called scalar product with incompatible fields")
        out = _lib.REAL()
        self.red_epoch += 1
        _lib.check(self.lib.x3d_scalar_product(self.h, x.ptr, y.ptr, self._dims(x.data_loc), ctypes.byref(out)))
        return self.comm.allreduce(out.value, "sum")

    def field_max_mean(self, f, enforced_data_loc=None):
        """returns (max_val, mean_val), src/backend/omp/backend.f90:739-810"""
        if f.data_loc == NULL_LOC and enforced_data_loc is None:
            raise X3dError("The input field to field_max_mean does not have a valid f%data_loc.")
        loc = f.data_loc if enforced_data_loc is None else enforced_data_loc
        if f.dir == DIR_C:
            raise X3dError("field_max_mean does not support DIR_C fields!")
        mx, sm = _lib.REAL(), _lib.REAL()
        self.red_epoch += 1
        _lib.check(self.lib.x3d_field_max_sum(self.h, f.ptr, self._dims(loc), ctypes.byref(mx), ctypes.byref(sm)))
        nglob = float(np.prod(self.mesh.get_global_dims(loc)))
        return self.comm.allreduce(mx.value, "max"), self.comm.allreduce(sm.value / nglob, "sum")

    def slice_max_sum(self, f, i_slice, enforced_data_loc=None):
        """rank-local (max, sum); the caller reduces (src/backend/omp/backend.f90:812-872)"""
        if f.data_loc == NULL_LOC and enforced_data_loc is None:
            raise X3dError("The input field to slice_max_sum does not have a valid f%data_loc.")
        loc = f.data_loc if enforced_data_loc is None else enforced_data_loc
        mx, sm = _lib.REAL(), _lib.REAL()
        self.red_epoch += 1
        _lib.check(self.lib.x3d_slice_max_sum(self.h, f.ptr, self._dims(loc), f.dir, int(i_slice),
                                              ctypes.byref(mx), ctypes.byref(sm)))
        return mx.value, sm.value

    def field_shift_to_mean(self, f, target):
        """f += target - volume_integral(f) / ncell_global: define_BC_channel's bulk-velocity correction
        (src/case/channel.f90:70-77).  One rank: reduction, difference and shift stay on the device."""
        if f.data_loc == NULL_LOC:
            raise X3dError("You must set the data_loc before calling volume integral.")
        if f.dir != DIR_X:
            raise X3dError("Volume integral can only be called on DIR_X fields.")
        ncell = float(np.prod(self.mesh.get_global_dims(CELL)))
        if self.comm.size > 1:
            self.field_shift(f, target - self.field_volume_integral(f) / ncell)
            return
        self.red_epoch += 1
        _lib.check(self.lib.x3d_field_shift_to_mean(self.h, f.ptr, self._dims(f.data_loc), ncell, float(target)))

    def field_mean_shift(self, f, target):
        """first half of field_shift_to_mean: the device address of target - volume_integral(f) / ncell_global (valid
        until this backend's next reduction), for field_shift_by / transeq_x_rot; None: several ranks (host path)"""
        if f.data_loc == NULL_LOC:
            raise X3dError("You must set the data_loc before calling volume integral.")
        if f.dir != DIR_X:
            raise X3dError("Volume integral can only be called on DIR_X fields.")
        if self.comm.size > 1:
            return None
        ncell = float(np.prod(self.mesh.get_global_dims(CELL)))
        out = VP()
        self.red_epoch += 1
        _lib.check(self.lib.x3d_field_mean_shift(self.h, f.ptr, self._dims(f.data_loc), ncell, float(target),
                                                 ctypes.byref(out)))
        return out

    def field_shift_by(self, f, shift):
        """f += the device scalar `shift` (field_mean_shift)"""
        _lib.check(self.lib.x3d_field_shift_by(self.h, f.ptr, shift))

    def wall_noise(self, f, amp, seed, draw):
        """planes y = 1 and y = ny of the (VERT) wall field f <- amp * (2 r - 1), generated on the device"""
        _lib.check(self.lib.x3d_wall_noise(self.h, f.ptr, self._dims(VERT), float(amp), int(seed) & (2 ** 64 - 1),
                                           int(draw) & (2 ** 64 - 1)))

    def field_volume_integral(self, f):
        if f.data_loc == NULL_LOC:
            raise X3dError("You must set the data_loc before calling volume integral.")
        if f.dir != DIR_X:
            raise X3dError("Volume integral can only be called on DIR_X fields.")
        out = _lib.REAL()
        self.red_epoch += 1
        _lib.check(self.lib.x3d_field_volume_integral(self.h, f.ptr, self._dims(f.data_loc), ctypes.byref(out)))
        return self.comm.allreduce(out.value, "sum")

    # ------------------------------------------------------------ faces
    def field_set_face(self, f, c_start, c_end, face):
        if f.dir != DIR_X:
            raise X3dError("Setting a field face is only supported for DIR_X fields.")
        if f.data_loc == NULL_LOC:
            raise X3dError("field_set_face require a valid data_loc.")
        _lib.check(self.lib.x3d_field_set_face(self.h, f.ptr, self._dims(f.data_loc), float(c_start),
                                               float(c_end), int(face)))

    def field_set_face_from_field(self, f, f_start, c_end, face, flow_rate_diff=0.0):
        if f.dir != DIR_X:
            raise X3dError("field_set_face_from_field: only supported for DIR_X fields.")
        if f.data_loc == NULL_LOC:
            raise X3dError("field_set_face_from_field: requires a valid data_loc.")
        _lib.check(self.lib.x3d_field_set_face_from_field(self.h, f.ptr, f_start.ptr, self._dims(f.data_loc),
                                                          float(c_end), int(face), float(flow_rate_diff)))

    # ------------------------------------------------------------ host <-> field
    def set_field_data(self, f, data, loc=None):
        """src/backend/backend.f90:436-466.  data: numpy [nz, ny, nx] (x fastest);
        the extent is that of loc (default: f%data_loc, VERT if unset)."""
        loc = (f.data_loc if f.data_loc != NULL_LOC else 0) if loc is None else loc
        nx, ny, nz = self.mesh.get_dims(loc)
        a = np.ascontiguousarray(data, dtype=_lib.NP_REAL)
        if a.shape != (nz, ny, nx):
            raise X3dError(f"set_field_data: array shape {a.shape} != {(nz, ny, nx)}")
        _lib.check(self.lib.x3d_set_field_data(self.h, f.ptr, _dp(a), _lib.ints(nx, ny, nz)))

    def get_field_data(self, f, loc=None):
        for hook in self.before_read:
            hook()
        loc = (f.data_loc if f.data_loc != NULL_LOC else 0) if loc is None else loc
        nx, ny, nz = self.mesh.get_dims(loc)
        out = np.empty((nz, ny, nx), dtype=_lib.NP_REAL)
        _lib.check(self.lib.x3d_get_field_data(self.h, _dp(out), f.ptr, _lib.ints(nx, ny, nz)))
        return out

    # ------------------------------------------------------------ Poisson
    def init_poisson_fft(self, mesh, xdirps, ydirps, zdirps, lowmem=None):
        from .poisson_fft import make_poisson_fft
        self.poisson_fft = make_poisson_fft(self, mesh, xdirps, ydirps, zdirps)
        return self.poisson_fft
