"""poisson_fft_t mirror: host-side set-up (modified wave numbers, spectral
equivalence constants, BC dispatch) as in the reference's abstract class
(/root/reference/src/poisson_fft.f90:120-273, 654-882) plus the hooks the HIP
backend supplies (fft_forward / fft_postprocess_000 / fft_backward through
rocFFT; reference CPU hooks: src/backend/omp/poisson_fft.f90:89-137)."""
import ctypes
import math

import numpy as np

from . import _lib
from .common import CELL, X3dError

VP = ctypes.c_void_p


def wave_numbers(n, L, d, periodic, c_a, c_b, c_alpha):
    """src/poisson_fft.f90:833-882.  Returns a, b, k, e, k2 (k, e, k2 are
    complex with equal parts in the reference; the common value is kept)."""
    pi = 4.0 * math.atan(1.0)
    i = np.arange(1, n + 1, dtype=np.float64)
    if periodic:
        a, b = np.sin((i - 1) * pi / n), np.cos((i - 1) * pi / n)
    else:
        a, b = np.sin((i - 1) * pi / 2 / n), np.cos((i - 1) * pi / 2 / n)
    k, e, k2 = np.zeros(n), np.zeros(n), np.zeros(n)

    def mod_wave(w):
        wp = c_a * 2 * d * np.sin(0.5 * w) + c_b * 2 * d * np.sin(1.5 * w)
        return wp / (1.0 + 2 * c_alpha * np.cos(w))

    if periodic:
        h = n // 2 + 1
        w = 2 * pi * (i[:h] - 1) / n
        wp = mod_wave(w)
        k[:h], e[:h], k2[:h] = n * wp / L, n * w / L, (n * wp / L) ** 2
        for j in range(n // 2 + 2, n + 1):  # mirror: k(i) = k(n-i+2)
            k[j - 1], e[j - 1], k2[j - 1] = k[n - j + 1], e[n - j + 1], k2[n - j + 1]
    else:
        w = pi * (i - 1) / n
        wp = mod_wave(w)
        k[:], e[:], k2[:] = n * wp / L, n * w / L, (n * wp / L) ** 2
    return a, b, k, e, k2


def make_poisson_fft(backend, mesh, xdirps, ydirps, zdirps):
    """init_poisson_fft: single-rank 3-D rocFFT plan, or the pencil-decomposed
    solver when the domain is split over ranks"""
    import os
    if mesh.nproc > 1 or os.environ.get("X3D_FORCE_PENCIL_FFT") == "1":
        return HipPencilPoissonFFT(backend, mesh, xdirps, ydirps, zdirps)
    return HipPoissonFFT(backend, mesh, xdirps, ydirps, zdirps)


class HipPoissonFFT:
    def __init__(self, backend, mesh, xdirps, ydirps, zdirps):
        self.backend, self.mesh = backend, mesh
        if int(mesh.nproc_dir[0]) != 1:
            print("nproc_dir in x-dir must be 1")  # :131
        self.nx_glob, self.ny_glob, self.nz_glob = mesh.get_global_dims(CELL)
        self.nx_loc, self.ny_loc, self.nz_loc = mesh.get_dims(CELL)
        self.periodic_x, self.periodic_y, self.periodic_z = mesh.periodic_BC
        if mesh.stretched[0] or mesh.stretched[2]:
            raise X3dError("FFT based Poisson solver does not support stretching in x- or z-directions!")
        if not (self.periodic_x and self.periodic_y and self.periodic_z):
            raise X3dError("HIP Poisson solver: only the all-periodic (000) case is implemented in this round")
        self.nx_spec, self.ny_spec, self.nz_spec = self.nx_glob // 2 + 1, self.ny_glob, self.nz_glob
        self.sp_st = (0, 0, 0)
        self._waves_set(mesh, xdirps, ydirps, zdirps)
        self._create()

    def _create(self):
        backend, mesh = self.backend, self.mesh
        if mesh.nproc > 1:
            raise X3dError("HipPoissonFFT is the single-rank solver; use make_poisson_fft")
        h = VP()
        dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(_lib.c_double_p)
        self._keep = [np.ascontiguousarray(x, dtype=np.float64) for x in
                      (self.waves, self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_poisson_create(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob),
            *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self.h = h
        self.poisson = self.poisson_000

    def __del__(self):
        try:
            self.backend.lib.x3d_poisson_destroy(self.h)
        except Exception:
            pass

    def _waves_set(self, mesh, xd, yd, zd):
        """src/poisson_fft.f90:654-831, 000/010 branch (:781-818)"""
        sx, sy, sz = xd.stagder_v2p, yd.stagder_v2p, zd.stagder_v2p
        self.ax, self.bx, self.kx, exs, k2x = wave_numbers(self.nx_glob, mesh.L[0], mesh.d[0],
                                                           self.periodic_x, sx.a, sx.b, sx.alpha)
        self.ay, self.by, self.ky, eys, k2y = wave_numbers(self.ny_glob, mesh.L[1], mesh.d[1],
                                                           self.periodic_y, sy.a, sy.b, sy.alpha)
        self.az, self.bz, self.kz, ezs, k2z = wave_numbers(self.nz_glob, mesh.L[2], mesh.d[2],
                                                           self.periodic_z, sz.a, sz.b, sz.alpha)
        self.k2x, self.k2y, self.k2z = k2x, k2y, k2z

        def transfer(t, e, d):
            r = e * d
            tt = 2 * (t.a * np.cos(r * 0.5) + t.b * np.cos(r * 1.5) + t.c * np.cos(r * 2.5)
                      + t.d * np.cos(r * 3.5))
            return tt / (1.0 + 2 * t.alpha * np.cos(r))

        self._t1d = (transfer(xd.interpl_v2p, exs[:self.nx_spec], mesh.d[0]),
                     transfer(yd.interpl_v2p, eys, mesh.d[1]),
                     transfer(zd.interpl_v2p, ezs, mesh.d[2]),
                     k2x[:self.nx_spec], k2y, k2z)
        self._waves_cache = None

    def waves_block(self, xsl=slice(None), ysl=slice(None), zsl=slice(None)):
        """waves(ix, iy, iz) for index ranges, returned as [z, y, x]
        (real part == imaginary part in the reference, :781-818)"""
        tx, ty, tz, kx2, ky2, kz2 = self._t1d
        tx, kx2 = tx[xsl][None, None, :], kx2[xsl][None, None, :]
        ty, ky2 = ty[ysl][None, :, None], ky2[ysl][None, :, None]
        tz, kz2 = tz[zsl][:, None, None], kz2[zsl][:, None, None]
        return kx2 * (ty * tz) ** 2 + ky2 * (tx * tz) ** 2 + kz2 * (tx * ty) ** 2

    @property
    def waves(self):
        if self._waves_cache is None:
            self._waves_cache = self.waves_block()
        return self._waves_cache

    # ---- hooks (src/poisson_fft.f90:45-62)
    def fft_forward(self, f_in):
        _lib.check(self.backend.lib.x3d_poisson_fft_forward(self.h, f_in.ptr))

    def fft_postprocess_000(self):
        _lib.check(self.backend.lib.x3d_poisson_postprocess_000(self.h))

    def fft_backward(self, f_out):
        _lib.check(self.backend.lib.x3d_poisson_fft_backward(self.h, f_out.ptr))

    def poisson_000(self, f, temp):  # :216-226
        self.fft_forward(f)
        self.fft_postprocess_000()
        self.fft_backward(f)

    def solve_poisson(self, f, temp):  # :206-214
        self.poisson(f, temp)

    # ---- test hooks
    def get_spectral(self):
        out = np.empty((self.nz_spec, self.ny_spec, self.nx_spec), dtype=np.complex128)
        _lib.check(self.backend.lib.x3d_poisson_get_spectral(
            self.h, out.view(np.float64).ctypes.data_as(_lib.c_double_p)))
        return out

    def set_spectral(self, c):
        c = np.ascontiguousarray(c, dtype=np.complex128)
        _lib.check(self.backend.lib.x3d_poisson_set_spectral(
            self.h, c.view(np.float64).ctypes.data_as(_lib.c_double_p)))


class HipPencilPoissonFFT(HipPoissonFFT):
    """000 solver over a [1, py, pz] decomposition: local rocFFT stages in
    libx3d2_hip.so (csrc/pfft.hip), two pencil transposes per direction as
    packed point-to-point exchanges inside the py and pz rank groups (the
    reference's CPU backend gets the same from 2decomp&FFT,
    src/backend/omp/poisson_fft.f90:72-97)."""

    def _create(self):
        import torch
        backend, mesh = self.backend, self.mesh
        if int(mesh.nproc_dir[0]) != 1:
            raise X3dError("FFT Poisson solver: nproc_dir in x-dir must be 1")
        self.py, self.pz = int(mesh.nproc_dir[1]), int(mesh.nproc_dir[2])
        self.ry, self.rz = int(mesh.nrank_dir[1]), int(mesh.nrank_dir[2])
        h = VP()
        _lib.check(backend.lib.x3d_pfft_create(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob), self.py, self.pz,
            self.ry, self.rz))
        self.h = h
        sz = (ctypes.c_long * 8)()
        _lib.check(backend.lib.x3d_pfft_sizes(h, sz))
        self.xs, self.xoff, self.ys, self.yoff, self.yl, self.zl, nxs, nmax = [int(v) for v in sz]
        # this rank's spectral block, z fastest: waves[x, y, z]
        xsl, ysl = slice(self.xoff, self.xoff + self.xs), slice(self.yoff, self.yoff + self.ys)
        full = self.waves_block(xsl, ysl)                   # [z, y, x] of this rank's modes only
        wl = np.ascontiguousarray(np.transpose(full, (2, 1, 0)))  # [x, y, z]
        self._keep = [np.ascontiguousarray(a, dtype=np.float64) for a in
                      (wl, self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_pfft_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self.sendbuf = torch.zeros(2 * nmax, dtype=torch.float64, device=backend.device)
        self.recvbuf = torch.zeros(2 * nmax, dtype=torch.float64, device=backend.device)

        def share(n, p, r):
            return n // p + (1 if r < n % p else 0)

        npy = self.py
        self.peers_y = [r + npy * self.rz for r in range(self.py)]   # rank = ry + py*rz (x undivided)
        self.peers_z = [self.ry + npy * r for r in range(self.pz)]
        x_sh = [share(nxs, self.py, r) for r in range(self.py)]
        y_sh = [share(self.ny_glob, self.pz, r) for r in range(self.pz)]
        c = 2  # doubles per complex element
        self.cnt_xy_send = [c * x * self.yl * self.zl for x in x_sh]
        self.cnt_xy_recv = [c * self.xs * self.yl * self.zl] * self.py
        self.cnt_yz_send = [c * y * self.xs * self.zl for y in y_sh]
        self.cnt_yz_recv = [c * self.ys * self.xs * self.zl] * self.pz
        self.poisson = self.poisson_000

    def __del__(self):
        try:
            self.backend.lib.x3d_pfft_destroy(self.h)
        except Exception:
            pass

    def _xchg(self, send_counts, recv_counts, peers):
        # stream-ordered: RCCL ops wait on the current stream, host staging (gloo) synchronises itself
        self.backend.comm.alltoall(self.sendbuf, send_counts, self.recvbuf, recv_counts, peers)

    def fft_forward(self, f_in):
        lib, h, sb, rb = self.backend.lib, self.h, self.sendbuf.data_ptr(), self.recvbuf.data_ptr()
        _lib.check(lib.x3d_pfft_fwd_x(h, f_in.ptr))
        _lib.check(lib.x3d_pfft_pack_xy(h, sb))
        self._xchg(self.cnt_xy_send, self.cnt_xy_recv, self.peers_y)
        _lib.check(lib.x3d_pfft_unpack_xy(h, rb))
        _lib.check(lib.x3d_pfft_fft_y(h, 0))
        _lib.check(lib.x3d_pfft_pack_yz(h, sb))
        self._xchg(self.cnt_yz_send, self.cnt_yz_recv, self.peers_z)
        _lib.check(lib.x3d_pfft_unpack_yz(h, rb))
        _lib.check(lib.x3d_pfft_fft_z(h, 0))

    def fft_postprocess_000(self):
        _lib.check(self.backend.lib.x3d_pfft_postprocess_000(self.h))

    def fft_backward(self, f_out):
        lib, h, sb, rb = self.backend.lib, self.h, self.sendbuf.data_ptr(), self.recvbuf.data_ptr()
        _lib.check(lib.x3d_pfft_fft_z(h, 1))
        _lib.check(lib.x3d_pfft_pack_zy(h, sb))
        self._xchg(self.cnt_yz_recv, self.cnt_yz_send, self.peers_z)
        _lib.check(lib.x3d_pfft_unpack_zy(h, rb))
        _lib.check(lib.x3d_pfft_fft_y(h, 1))
        _lib.check(lib.x3d_pfft_pack_yx(h, sb))
        self._xchg(self.cnt_xy_recv, self.cnt_xy_send, self.peers_y)
        _lib.check(lib.x3d_pfft_unpack_yx(h, rb))
        _lib.check(lib.x3d_pfft_bwd_x(h, f_out.ptr))

    def get_spectral(self):
        raise X3dError("get_spectral: single-rank test hook")

    def set_spectral(self, c):
        raise X3dError("set_spectral: single-rank test hook")
