"""The pencil Poisson solver's local shortcuts (csrc/pfft.hip), which only the Fortran shim drives
(fortran/m_hip_backend.f90:440-506): x3d_pfft_transpose_local -- the transposition along an UNDIVIDED direction in one
kernel -- and x3d_pfft_own_chunk -- a rank's own chunk unpacked straight out of its send buffer.  All ranks of a
[1, py, pz] layout are emulated in ONE process (one x3d_pfft handle per rank, the exchanges are device copies between
the ranks' buffers, chunk by chunk as the header lays them out) and the whole solve poisson_000
(/root/reference/src/poisson_fft.f90:216-226) is run twice: pack -> exchange -> unpack everywhere, and with every
shortcut the layout allows.  Same kernels on the same numbers otherwise: the pressure must agree BIT FOR BIT, on every
rank; the reference route is pinned against numpy's DFT."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NG = (32, 24, 16)  # global cells: nxs = 17 modes split unevenly over py (9 + 8), ny over pz


def share(n, p, r):
    return n // p + (1 if r < n % p else 0)


def share_off(n, p, r):
    return sum(share(n, p, q) for q in range(r))


class Ranks:
    """every rank of a [1, py, pz] decomposition of NG, in one process"""

    def __init__(self, py, pz, parts=1):
        import torch
        from x3d2_amd import Mesh, _lib
        from x3d2_amd.backend import HipBackend
        from x3d2_amd.common import CELL, DIR_C
        from x3d2_amd.poisson_fft import HipPoissonFFT
        from x3d2_amd.solver import Solver, SolverConfig
        self.torch, self.lib, self._lib = torch, _lib.load(), _lib
        self.py, self.pz = py, pz
        nx, ny, nz = NG
        self.yl, self.zl = ny // py, nz // pz
        self.nxs = nx // 2 + 1
        loc = (nx, self.yl, self.zl)
        # one context with the rank-local block dims serves every emulated rank (same slab shape everywhere)
        mesh = Mesh(loc, (1, 1, 1), (2 * np.pi,) * 3, ("periodic",) * 2, ("periodic",) * 2, ("periodic",) * 2)
        self.backend = HipBackend(mesh)
        # wave numbers of the GLOBAL problem: a single-rank solver object of the global size does the host set-up
        gmesh = Mesh(NG, (1, 1, 1), (2 * np.pi,) * 3, ("periodic",) * 2, ("periodic",) * 2, ("periodic",) * 2)
        self.gsolver = Solver(HipBackend(gmesh), gmesh, SolverConfig(poisson_solver_type="FFT"))
        gp = self.gsolver.backend.poisson_fft
        assert type(gp) is HipPoissonFFT
        VP = ctypes.c_void_p
        self.h, self.keep, self.f, self.sz = {}, [], {}, {}
        al = self.backend.allocator
        for rz in range(pz):
            for ry in range(py):
                h = VP()
                _lib.check(self.lib.x3d_pfft_create_parts(self.backend.h, ctypes.byref(h), _lib.ints(*NG), py, pz, ry, rz, parts))
                sz = (ctypes.c_long * 8)()
                _lib.check(self.lib.x3d_pfft_sizes(h, sz))
                xs, xoff, ys, yoff, yl, zl, nxs, nmax = [int(v) for v in sz]
                assert (yl, zl, nxs) == (self.yl, self.zl, self.nxs)
                full = gp.waves_block(slice(xoff, xoff + xs), slice(yoff, yoff + ys))  # [z, y, x]
                wl = np.ascontiguousarray(np.transpose(full, (2, 1, 0)))
                keep = [np.ascontiguousarray(a, dtype=_lib.NP_REAL) for a in (wl, gp.ax, gp.bx, gp.ay, gp.by, gp.az, gp.bz)]
                self.keep.append(keep)
                _lib.check(self.lib.x3d_pfft_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in keep]))
                self.h[(ry, rz)], self.sz[(ry, rz)] = h, (xs, xoff, ys, yoff, nmax)
                blk = al.get_block(DIR_C, CELL)
                self.f[(ry, rz)] = blk
        self.nmax = max(v[4] for v in self.sz.values())
        z = lambda: {k: torch.full((2 * self.nmax,), float("nan"), dtype=_lib.torch_real(), device=self.backend.device)
                     for k in self.h}
        self.send, self.recv = z(), z()

    def close(self):
        for h in self.h.values():
            self.lib.x3d_pfft_destroy(h)
        for b in self.f.values():
            self.backend.allocator.release_block(b)

    def set_rhs(self, g):
        """g: the global right-hand side [nz, ny, nx]"""
        from x3d2_amd.common import CELL
        for (ry, rz), blk in self.f.items():
            self.backend.set_field_data(blk, g[rz * self.zl:(rz + 1) * self.zl, ry * self.yl:(ry + 1) * self.yl, :], CELL)

    def get(self):
        from x3d2_amd.common import CELL
        out = np.empty((NG[2], NG[1], NG[0]))
        for (ry, rz), blk in self.f.items():
            out[rz * self.zl:(rz + 1) * self.zl, ry * self.yl:(ry + 1) * self.yl, :] = self.backend.get_field_data(blk, CELL)
        return out

    def call(self, name, *args):
        self._lib.check(getattr(self.lib, name)(*args))

    # ---- the four exchanges, chunk by chunk (complex elements -> 2 reals); skip_own: the own chunk is NOT delivered
    def xchg_xy(self, back, skip_own):
        for rz in range(self.pz):
            for ry in range(self.py):          # receiver
                xs, _, _, _, _ = self.sz[(ry, rz)]
                for q in range(self.py):       # sender
                    if skip_own and q == ry:
                        continue
                    xs_q = self.sz[(q, rz)][0]
                    if not back:
                        # sender q: chunk for ry at xoff_ry * yl * zl, xs_ry * yl * zl long; receiver: slot q
                        n = xs * self.yl * self.zl
                        so = self.sz[(ry, rz)][1] * self.yl * self.zl
                        ro = q * xs * self.yl * self.zl
                    else:
                        # sender q packed the receive layout: chunk for ry is slot ry of ITS layout (xs_q columns)
                        n = xs_q * self.yl * self.zl
                        so = ry * xs_q * self.yl * self.zl
                        ro = self.sz[(q, rz)][1] * self.yl * self.zl
                    self.recv[(ry, rz)][2 * ro:2 * (ro + n)].copy_(self.send[(q, rz)][2 * so:2 * (so + n)])

    def xchg_yz(self, back):
        for ry in range(self.py):
            for rz in range(self.pz):          # receiver
                xs, _, ys, yoff, _ = self.sz[(ry, rz)]
                for q in range(self.pz):       # sender
                    ys_q, yoff_q = self.sz[(ry, q)][2], self.sz[(ry, q)][3]
                    if not back:
                        n = ys * xs * self.zl
                        so, ro = yoff * xs * self.zl, q * ys * xs * self.zl
                    else:
                        n = ys_q * xs * self.zl
                        so, ro = rz * ys_q * xs * self.zl, yoff_q * xs * self.zl
                    self.recv[(ry, rz)][2 * ro:2 * (ro + n)].copy_(self.send[(ry, q)][2 * so:2 * (so + n)])

    def solve(self, shortcuts):
        """poisson_000 on every rank; shortcuts: transpose_local where a direction is undivided, own_chunk where the
        x-y exchange is real"""
        P = lambda t: t.data_ptr()
        loc_y, loc_z = shortcuts and self.py == 1, shortcuts and self.pz == 1
        own = shortcuts and self.py > 1
        nan = float("nan")
        for b in list(self.send.values()) + list(self.recv.values()):
            b.fill_(nan)
        R = list(self.h.items())
        for k, h in R:
            self.call("x3d_pfft_fwd_x", h, self.f[k].ptr)
        if loc_y:
            for k, h in R:
                self.call("x3d_pfft_transpose_local", h, 0)
        else:
            for k, h in R:
                self.call("x3d_pfft_pack_xy", h, P(self.send[k]))
            self.xchg_xy(False, own)
            for k, h in R:
                if own:
                    self.call("x3d_pfft_own_chunk", h, P(self.send[k]), k[0])
                self.call("x3d_pfft_unpack_xy", h, P(self.recv[k]))
        for k, h in R:
            self.call("x3d_pfft_fft_y", h, 0)
        if loc_z:
            for k, h in R:
                self.call("x3d_pfft_transpose_local", h, 1)
        else:
            for k, h in R:
                self.call("x3d_pfft_pack_yz", h, P(self.send[k]))
            self.xchg_yz(False)
            for k, h in R:
                self.call("x3d_pfft_unpack_yz", h, P(self.recv[k]))
        for k, h in R:
            self.call("x3d_pfft_fft_z", h, 0)
            self.call("x3d_pfft_postprocess_000", h)
            self.call("x3d_pfft_fft_z", h, 1)
        if loc_z:
            for k, h in R:
                self.call("x3d_pfft_transpose_local", h, 2)
        else:
            for k, h in R:
                self.call("x3d_pfft_pack_zy", h, P(self.send[k]))
            self.xchg_yz(True)
            for k, h in R:
                self.call("x3d_pfft_unpack_zy", h, P(self.recv[k]))
        for k, h in R:
            self.call("x3d_pfft_fft_y", h, 1)
        if loc_y:
            for k, h in R:
                self.call("x3d_pfft_transpose_local", h, 3)
        else:
            for b in self.recv.values():
                b.fill_(nan)  # (an own chunk that is NOT taken from the send buffer would now read NaNs)
            for k, h in R:
                self.call("x3d_pfft_pack_yx", h, P(self.send[k]))
            self.xchg_xy(True, own)
            for k, h in R:
                if own:
                    self.call("x3d_pfft_own_chunk", h, P(self.send[k]), k[0])
                self.call("x3d_pfft_unpack_yx", h, P(self.recv[k]))
        for k, h in R:
            self.call("x3d_pfft_bwd_x", h, self.f[k].ptr)
        self.backend.sync()
        return self.get()


@pytest.mark.parametrize("py,pz", [(1, 1), (1, 2), (2, 1), (1, 4), (3, 1), (2, 2)])
def test_pencil_solver_local_transposes_and_own_chunk_bit_for_bit(py, pz):
    rng = np.random.default_rng(100 * py + pz)
    g = rng.standard_normal((NG[2], NG[1], NG[0]))
    g -= g.mean()
    r = Ranks(py, pz)
    try:
        r.set_rhs(g)
        plain = r.solve(shortcuts=False)
        r.set_rhs(g)
        short = r.solve(shortcuts=True)
        assert np.all(np.isfinite(plain)) and np.all(np.isfinite(short))
        assert np.array_equal(plain, short)
        # the reference route itself: the single-rank solver of the global problem (pinned to numpy's DFT and the
        # reference's vectors in test_hip_parity.py) -- different transform sizes per kernel, round-off apart
        from x3d2_amd.common import CELL, DIR_C
        gs = r.gsolver
        gb, gal = gs.backend, gs.backend.allocator
        p, t = gal.get_block(DIR_C, CELL), gal.get_block(DIR_C)
        gb.set_field_data(p, g, CELL)
        gb.poisson_fft.solve_poisson(p, t)
        one = gb.get_field_data(p, CELL)
        gal.release_block(p); gal.release_block(t)
        assert np.max(np.abs(plain - one)) < 1e-12 * max(np.max(np.abs(one)), 1.0)
    finally:
        r.close()


def test_group_entry_points_refuse_a_pending_own_chunk():
    """x3d_pfft_own_chunk points at a whole-solve send buffer; a group's unpack (x3d_pfft_*_part) indexes a group's piece
    and would read another group's planes: refused loudly, the flag stays until a whole-solve unpack consumes it"""
    from x3d2_amd.common import X3dError
    r = Ranks(2, 1, parts=2)
    try:
        h = r.h[(0, 0)]
        r.call("x3d_pfft_own_chunk", h, r.send[(0, 0)].data_ptr(), 0)
        with pytest.raises(X3dError):
            r.call("x3d_pfft_fwd_b_part", h, r.recv[(0, 0)].data_ptr(), r.send[(0, 0)].data_ptr(), 0)
        with pytest.raises(X3dError):
            r.call("x3d_pfft_own_chunk", h, r.send[(0, 0)].data_ptr(), 2)  # not a rank of the y group
        r.send[(0, 0)].zero_(); r.recv[(0, 0)].zero_()
        r.call("x3d_pfft_unpack_xy", h, r.recv[(0, 0)].data_ptr())  # consumes the flag
        r.call("x3d_pfft_fwd_b_part", h, r.recv[(0, 0)].data_ptr(), r.send[(0, 0)].data_ptr(), 0)
    finally:
        r.close()
