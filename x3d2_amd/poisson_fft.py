"""poisson_fft_t mirror: host-side set-up (modified wave numbers, spectral
equivalence constants, BC dispatch) as in the reference's abstract class
(/root/reference/src/poisson_fft.f90:120-273, 654-882) plus the hooks the HIP
backend supplies (fft_forward / fft_postprocess_000 / fft_backward through
rocFFT; reference CPU hooks: src/backend/omp/poisson_fft.f90:89-137)."""
import ctypes
import math
import os

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


def stretching_matrix(pf, mesh, xd, yd, zd, eys, exs, ezs, xsl=None):
    """src/poisson_fft.f90:275-652: pentadiagonal spectral operators for a stretched y
    (JCP 228 (2009) 5989, Sec. 5).  Real and imaginary copies of the reference are equal
    (every wave number is cmplx(1,1)*x), one copy is kept.  Layout [5][nz][n][nx_spec].
    Entries the reference leaves unset or reads past ky(ny) for (never used by the solve)
    are zero here.  xsl: only these x modes (a rank of the slab solver builds its own columns)."""
    pi = 4 * np.arctan(1.0)
    nxs, nys, nzs = pf.nx_spec, pf.ny_spec, pf.nz_spec
    xsl = slice(0, nxs) if xsl is None else xsl

    def transfer(t, e, d, n):
        tmp = e[:n] * d
        it = t.interpl_v2p
        return 2 * (it.a * np.cos(tmp * 0.5) + it.b * np.cos(tmp * 1.5) + it.c * np.cos(tmp * 2.5)
                    + it.d * np.cos(tmp * 3.5)) / (1.0 + 2 * it.alpha * np.cos(tmp))

    tx = transfer(xd, exs, mesh.d[0], nxs)[xsl]
    ty = transfer(yd, eys, mesh.d[1], nys)
    tz = transfer(zd, ezs, mesh.d[2], nzs)
    pf.trans_x, pf.trans_y, pf.trans_z = tx, ty, tz
    kyp = np.concatenate([pf.ky, np.zeros(4)])
    TX, TZ = tx[None, None, :], tz[:, None, None]
    KX, KZ = pf.kx[None, None, :nxs][:, :, xsl], pf.kz[:nzs, None, None]
    k2x_sel = pf.k2x[:nxs][xsl]
    nxs = len(tx)

    def km(iy):  # get_km(ix, iy, iz), iy 1-based array -> [nz, len(iy), nx]
        return (TX * kyp[np.asarray(iy) - 1][None, :, None]) * TZ

    def tyv(iy):
        return ty[np.asarray(iy) - 1][None, :, None]

    kind = mesh.stretching[1]
    L, beta, alpha = mesh.L[1], mesh.beta[1], mesh.alpha[1]
    a0 = (alpha / pi + 1.0 / (2 * pi * beta)) * L
    if kind == "bottom":
        pf.stretched_y_sym = False
        a1 = -1.0 / (4 * pi * beta) * L
        n = nys
        a = np.zeros((5, nzs, n, nxs))
        iy = np.arange(1, n + 1)
        K = km(iy)
        kma = km(np.clip(iy - 1, 1, None)) + km(iy + 1)
        kma[:, 0, :] = km([2])[:, 0, :]
        kma[:, n - 1, :] = km([n - 1])[:, 0, :]
        a[2] = -((KX * tyv(iy)) * TZ) ** 2 - ((KZ * tyv(iy)) * TX) ** 2 - a0 ** 2 * K ** 2 \
            - (a1 ** 2 * K) * kma
        a[3] = ((a0 * a1) * km(iy + 1)) * (K + km(iy + 1))
        a[4, :, :n - 2] = (-(a1 * a1 * km(iy + 1)) * km(iy + 2))[:, :n - 2]
        a[1, :, 1:] = (((a0 * a1) * km(np.clip(iy - 1, 1, None))) * (K + km(np.clip(iy - 1, 1, None))))[:, 1:]
        a[0, :, 2:] = (-(a1 * a1 * km(np.clip(iy - 1, 1, None))) * km(np.clip(iy - 2, 1, None)))[:, 2:]
        if xsl.start in (0, None) and nxs > 0:
            a[2, 0, 0, 0] = 1.0; a[3, 0, 0, 0] = 0.0; a[4, 0, 0, 0] = 0.0
        pf.a_full = a
        return
    pf.stretched_y_sym = True
    a1 = {"centred": 1.0, "top-bottom": -1.0}.get(kind, 0.0) / (4 * pi * beta) * L
    n = nys // 2
    j = np.arange(1, n + 1)
    out = {}
    for name, iyv in (("odd", 2 * j - 1), ("even", 2 * j)):
        ev = name == "even"
        a = np.zeros((5, nzs, n, nxs))
        K = km(iyv)
        Kp2, Kp4 = km(iyv + 2), km(iyv + 4)
        Km2, Km4 = km(np.clip(iyv - 2, 1, None)), km(np.clip(iyv - 4, 1, None))
        # diagonal
        c1 = np.full(n, a0 * a0); c2 = np.full(n, a1 * a1)
        kma = Km2 + Kp2
        kma[:, 0, :] = km([4 if ev else 3])[:, 0, :]
        kma[:, n - 1, :] = Km2[:, n - 1, :]
        if ev:
            c1[0] = a0 * a0 - a1 * a1
            c1[n - 1] = (a0 + a1) * (a0 + a1)
        C1, C2 = c1[None, :, None], c2[None, :, None]
        a[2] = -((KX * tyv(iyv)) * TZ) ** 2 - ((KZ * tyv(iyv)) * TX) ** 2 - C1 * K ** 2 - (C2 * K) * kma
        # diagonal + 1
        c1 = np.full(n, a0 * a1); c2 = np.full(n, a0 * a1)
        if ev:
            if n >= 2:
                c1[n - 2] = a0 * a1; c2[n - 2] = (a0 + a1) * a1
            c1[n - 1] = 0.0; c2[n - 1] = 0.0
            c1[0] = a0 * a1 - a1 * a1; c2[0] = a0 * a1
        else:
            c1[0] = 2 * a0 * a1; c2[0] = 2 * a0 * a1
        C1, C2 = c1[None, :, None], c2[None, :, None]
        a[3] = C1 * (K * Kp2) + C2 * Kp2 ** 2
        # diagonal + 2
        c1 = np.full(n, a1 * a1)
        if not ev:
            c1[0] = 2 * a1 * a1
        a[4, :, :max(n - 2, 0)] = (-((c1[None, :, None] * Kp2) * Kp4))[:, :max(n - 2, 0)]
        # diagonal - 1
        c1 = np.full(n, a0 * a1); c2 = np.full(n, a0 * a1)
        if ev:
            c1[n - 1] = (a0 + a1) * a1; c2[n - 1] = a0 * a1
            if n >= 2:
                c1[1] = a0 * a1; c2[1] = (a0 + a1) * a1
        C1, C2 = c1[None, :, None], c2[None, :, None]
        a[1, :, 1:] = (C1 * (K * Km2) + C2 * Km2 ** 2)[:, 1:]
        # diagonal - 2
        a[0, :, 2:] = (-(((a1 * a1) * Km2) * Km4))[:, 2:]
        out[name] = a
    zero = (k2x_sel[None, :] < 1e-15) & (pf.k2z[:nzs][:, None] < 1e-15)  # [nz, nx]
    ao = out["odd"]
    ao[2, :, 0, :][zero] = 1.0
    ao[3, :, 0, :][zero] = 0.0
    ao[4, :, 0, :][zero] = 0.0
    pf.a_odd, pf.a_even = out["odd"], out["even"]


def stretching_matrix_zfirst(pf, mesh, xd, yd, zd, eys, exs, ezs):
    """the same operators for the z-first form of the 010 solve (csrc/zfirst.hip, round 6): EVERY x mode -- the wave numbers
    above nx / 2 are the mirrored ones wave_numbers() builds, as the reference's are along z (src/poisson_fft.f90:833-882)
    -- and the z modes 0 .. nz / 2 only: [5][nz/2+1][n][nx].  stretching_matrix() itself with the mode ranges swapped;
    returns (a0, a1) for x3d_poisson_set_stretching_zfirst, pf's own matrices stay as they are."""
    names = ("a_odd", "a_even", "a_full", "trans_x", "trans_y", "trans_z", "stretched_y_sym", "nx_spec", "nz_spec")
    keep = {k: getattr(pf, k, None) for k in names}
    had = {k: hasattr(pf, k) for k in names}
    try:
        pf.nx_spec, pf.nz_spec = pf.nx_glob, pf.nz_glob // 2 + 1
        stretching_matrix(pf, mesh, xd, yd, zd, eys, exs, ezs)
        out = (pf.a_odd, pf.a_even) if pf.stretched_y_sym else (pf.a_full, pf.a_full)
    finally:
        for k in names:
            if had[k]:
                setattr(pf, k, keep[k])
            elif hasattr(pf, k):
                delattr(pf, k)
    return out


def make_poisson_fft(backend, mesh, xdirps, ydirps, zdirps):
    """init_poisson_fft: single-rank 3-D rocFFT plan, or the pencil-decomposed
    solver when the domain is split over ranks"""
    force = os.environ.get("X3D_FORCE_PENCIL_FFT")  # "1": generic pencil solver, "slab": z slabs, "yslab": y slabs
    if mesh.nproc > 1 or force in ("1", "slab", "yslab"):
        ny = int(mesh.get_global_dims(CELL)[1])
        pz = int(mesh.nproc_dir[2])
        py = int(mesh.nproc_dir[1])
        yslab_ok = (pz == 1 and py in (1, 2, 4, 8) and all(mesh.periodic_BC) and force in (None, "yslab")
                    and tuple(int(v) for v in mesh.get_dims(CELL)) == (512, 512, 512)
                    and os.environ.get("X3D_NO_YSLAB_FFT") != "1")
        if yslab_ok and (py > 1 or force == "yslab"):
            # y slabs of 512^3 cells: z is whole on every rank, the z-first solve applies (csrc/sfftz.hip)
            return HipSlabPoissonFFTZ(backend, mesh, xdirps, ydirps, zdirps)
        if tuple(bool(x) for x in mesh.periodic_BC) == (True, False, True):
            # non-periodic y on z slabs (the channel case): the x modes are split over the ranks, y stays whole
            return HipSlabPoissonFFT010(backend, mesh, xdirps, ydirps, zdirps)
        slab_ok = (int(mesh.nproc_dir[1]) == 1 and ny == 512 and 512 % pz == 0 and all(mesh.periodic_BC)
                   and os.environ.get("X3D_NO_SLAB_FFT") != "1")
        if slab_ok and force != "1":
            return HipSlabPoissonFFT(backend, mesh, xdirps, ydirps, zdirps)
        return HipPencilPoissonFFT(backend, mesh, xdirps, ydirps, zdirps)
    per = tuple(bool(x) for x in mesh.periodic_BC)
    if per == (False, True, True):
        return HipPoissonFFT100(backend, mesh, xdirps, ydirps, zdirps)
    if per == (False, False, True):
        return HipPoissonFFT110(backend, mesh, xdirps, ydirps, zdirps)
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
        # BC dispatch, src/poisson_fft.f90:171-203
        if self.periodic_x and self.periodic_y and self.periodic_z:
            self.case = "000"
        elif self.periodic_x and (not self.periodic_y) and self.periodic_z:
            if mesh.nproc > 1 and not isinstance(self, HipSlabPoissonFFT010):
                raise X3dError("Multiple ranks are not yet supported for non-periodic BCs!")
            self.case = "010"
        elif (not self.periodic_x) and self.periodic_y and self.periodic_z:
            if mesh.nproc > 1:
                raise X3dError("Multiple ranks are not yet supported for non-periodic BCs!")
            if type(self) is HipPoissonFFT:
                raise X3dError("Poisson 100: use make_poisson_fft (HipPoissonFFT100)")
            self.case = "100"
        elif (not self.periodic_x) and (not self.periodic_y) and self.periodic_z:
            if mesh.nproc > 1:
                raise X3dError("Multiple ranks are not yet supported for non-periodic BCs!")
            if type(self) is HipPoissonFFT:
                raise X3dError("Poisson 110: use make_poisson_fft (HipPoissonFFT110)")
            if mesh.stretched[1]:
                raise X3dError("Poisson 110: uniform grids only (as in the reference)")
            self.case = "110"
        else:
            raise X3dError("Requested BCs are not supported in FFT-based Poisson solver!")
        self.nx_spec, self.ny_spec, self.nz_spec = self.nx_glob // 2 + 1, self.ny_glob, self.nz_glob
        self.sp_st = (0, 0, 0)
        self.stretched_y = False
        self._waves_set(mesh, xdirps, ydirps, zdirps)
        if self.case == "010" and mesh.stretched[1]:
            self.stretched_y = True
            if not isinstance(self, HipSlabPoissonFFT010):  # (the slab solver builds its own columns only)
                stretching_matrix(self, mesh, xdirps, ydirps, zdirps, *self._es)
        self._dirps = (xdirps, ydirps, zdirps)
        self._create()
        self._dirps = None

    def _create(self):
        backend, mesh = self.backend, self.mesh
        if mesh.nproc > 1:
            raise X3dError("HipPoissonFFT is the single-rank solver; use make_poisson_fft")
        h = VP()
        dp = lambda a: np.ascontiguousarray(a, dtype=_lib.NP_REAL).ctypes.data_as(_lib.c_double_p)
        self._keep = [np.ascontiguousarray(x, dtype=_lib.NP_REAL) for x in
                      (self.waves, self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_poisson_create(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob),
            *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self.h = h
        self.poisson = self.poisson_000 if self.case == "000" else self.poisson_010
        if self.stretched_y:
            if self.stretched_y_sym:
                a0, a1 = self.a_odd, self.a_even
            else:
                a0, a1 = self.a_full, self.a_full
            _lib.check(backend.lib.x3d_poisson_set_stretching(self.h, int(self.stretched_y_sym), dp(a0), dp(a1)))
            if not getattr(self, "keep_matrices", False):  # GBs at production sizes; the device holds the factors
                self.a_odd = self.a_even = self.a_full = None
            # the channel's sizes on one rank: the operators once more in the z-first layout (x3d_poisson_zfirst_ok then
            # offers the solve with its z transforms on the tiles of the neighbouring z operator pairs, csrc/zfirst.hip)
            if (type(self) is HipPoissonFFT and (self.nx_glob, self.ny_glob, self.nz_glob) == (1024, 256, 512)
                    and os.environ.get("X3D_NO_ZFIRST010") != "1" and os.environ.get("X3D_NO_ZFIRST") != "1"
                    and not getattr(backend, "lazy", False)):
                z0, z1 = stretching_matrix_zfirst(self, mesh, *self._dirps, *self._es)
                _lib.check(backend.lib.x3d_poisson_set_stretching_zfirst(self.h, int(self.stretched_y_sym), dp(z0), dp(z1)))
                del z0, z1

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
        self._es = (eys, exs, ezs)

        def transfer(t, e, d):
            r = e * d
            tt = 2 * (t.a * np.cos(r * 0.5) + t.b * np.cos(r * 1.5) + t.c * np.cos(r * 2.5)
                      + t.d * np.cos(r * 3.5))
            return tt / (1.0 + 2 * t.alpha * np.cos(r))

        self._t1d_full = (transfer(xd.interpl_v2p, exs, mesh.d[0]), transfer(yd.interpl_v2p, eys, mesh.d[1]),
                          transfer(zd.interpl_v2p, ezs, mesh.d[2]), k2x, k2y, k2z)
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
        if type(self) is HipPoissonFFT and not getattr(self.backend, "lazy", False):
            # one C call: forward ; postprocess_000 ; backward (the library fuses the z passes when it can)
            # (deferred execution: the three hooks are recorded like the Fortran shim's and merged by the queue)
            _lib.check(self.backend.lib.x3d_poisson_solve_000(self.h, f.ptr))
            return
        self.fft_forward(f)
        self.fft_postprocess_000()
        self.fft_backward(f)

    def fft_postprocess_010(self):
        _lib.check(self.backend.lib.x3d_poisson_postprocess_010(self.h))

    def enforce_periodicity_y(self, f_out, f_in):
        _lib.check(self.backend.lib.x3d_poisson_enforce_periodicity_y(self.h, f_out.ptr, f_in.ptr))

    def undo_periodicity_y(self, f_out, f_in):
        _lib.check(self.backend.lib.x3d_poisson_undo_periodicity_y(self.h, f_out.ptr, f_in.ptr))

    def poisson_010(self, f, temp):  # :228-242
        if temp is None:
            raise X3dError("poisson_010 needs a scratch block")
        self.enforce_periodicity_y(temp, f)
        # fft_forward_010 => fft_forward (src/backend/cuda/poisson_fft.f90:79-86) ; fft_postprocess_010 ; fft_backward
        self.solve_interleaved(temp)
        self.undo_periodicity_y(f, temp)

    def solve_poisson(self, f, temp):  # :206-214
        self.poisson(f, temp)

    def zfirst_ok(self):
        """the z-first form of poisson_000 is on offer (512^3 on one rank, csrc/zfirst.hip): the z operator pairs next
        to the solve transform along z on their tiles (HipBackend.tds_pair_zfirst), zfirst_middle() does the rest"""
        # (round 6: also the channel's 010 solve at 1024 x 256 x 512 cells -- the library says which)
        if type(self) is not HipPoissonFFT or self.case not in ("000", "010") or getattr(self.backend, "lazy", False):
            return False
        ok = ctypes.c_int(0)
        _lib.check(self.backend.lib.x3d_poisson_zfirst_ok(self.h, ctypes.byref(ok)))
        return bool(ok.value)

    def zfirst_middle(self):
        _lib.check(self.backend.lib.x3d_poisson_zfirst_middle(self.h))

    def zfirst_forward(self, f):
        _lib.check(self.backend.lib.x3d_poisson_zfirst_forward(self.h, f.ptr))

    def zfirst_backward(self, f):
        _lib.check(self.backend.lib.x3d_poisson_zfirst_backward(self.h, f.ptr))

    def solve_zfirst(self, f):
        """poisson_000 through the z-first stages with the z transforms as kernels of their own, in place"""
        _lib.check(self.backend.lib.x3d_poisson_solve_000_zfirst(self.h, f.ptr))

    def interleaved_rows(self):
        """> 0: poisson_010 without its two row-interleaving copies is on offer (solve_interleaved): the caller's z
        operators next to the solve write / read that many y rows at their interleaved positions themselves
        (HipBackend.tds_pair_yperm)"""
        if type(self) is HipPoissonFFT and self.case == "010" and self.ny_glob % 2 == 0:
            return self.ny_glob
        return 0

    def solve_interleaved(self, f):
        """poisson_010 (:228-242) on a field whose y rows are already in enforce_periodicity_y's order; the result
        is left in that order (undo_periodicity_y not applied), in place.  One library call: 256 cells along a
        stretched y run x ; z ; ONE pass for the y transform, fft_postprocess_010 and the inverse y transform
        (csrc/y010.hip); every other size the three steps below"""
        if getattr(self.backend, "lazy", False):
            self.fft_forward(f)
            self.fft_postprocess_010()
            self.fft_backward(f)
            return
        _lib.check(self.backend.lib.x3d_poisson_solve_010_rows(self.h, f.ptr))

    def solve_interleaved_zfirst(self, f):
        """solve_interleaved through the z-first stages with the z transforms as kernels of their own (test hook: the fused
        driver has them inside its z operator pairs, Solver._zfirst_solve)"""
        _lib.check(self.backend.lib.x3d_poisson_solve_010_rows_zfirst(self.h, f.ptr))

    # ---- test hooks
    def get_spectral(self):
        out = np.empty((self.nz_spec, self.ny_spec, self.nx_spec), dtype=np.complex64 if _lib.SINGLE else np.complex128)
        _lib.check(self.backend.lib.x3d_poisson_get_spectral(
            self.h, out.view(_lib.NP_REAL).ctypes.data_as(_lib.c_double_p)))
        return out

    def set_spectral(self, c):
        c = np.ascontiguousarray(c, dtype=np.complex64 if _lib.SINGLE else np.complex128)
        _lib.check(self.backend.lib.x3d_poisson_set_spectral(
            self.h, c.view(_lib.NP_REAL).ctypes.data_as(_lib.c_double_p)))


class HipPoissonFFT100(HipPoissonFFT):
    """x non-periodic, y and z periodic.  The reference's poisson_100 (src/poisson_fft.f90:244-256) is its 010
    solve on the x <-> y transposed problem: enforce_periodicity_x ; fft_forward_100 (transposed copy + R2C along
    y) ; fft_postprocess_100 = process_spectral_010 with nx <-> ny, ax,bx <-> ay,by swapped and waves indexed
    (y mode, x mode, z mode) (src/backend/cuda/poisson_fft.f90:482-616, 781-820; waves_set :735-779) ;
    fft_backward_100 ; undo_periodicity_x.  Here literally: a second library backend of the transposed vertex
    dims on the same stream, an x3d_poisson on it built from the swapped arrays, and x3d_transpose_xy between the
    two block layouts (the even / odd interleave along x is the inner solver's enforce_periodicity_y)."""

    def _create(self):
        backend, mesh = self.backend, self.mesh
        lib = backend.lib
        if mesh.nproc > 1:
            raise X3dError("Multiple ranks are not yet supported for non-periodic BCs!")
        nx, ny, nz = self.nx_glob, self.ny_glob, self.nz_glob
        tb = VP()
        vd = [int(v) for v in mesh.vert_dims]
        _lib.check(lib.x3d_backend_create(ctypes.byref(tb), _lib.ints(vd[1], vd[0], vd[2]), backend.device.index,
                                          VP(backend.stream.cuda_stream)))
        self.tb = tb
        self.t1, self.t2 = VP(), VP()
        _lib.check(lib.x3d_block_alloc(tb, ctypes.byref(self.t1)))
        _lib.check(lib.x3d_block_alloc(tb, ctypes.byref(self.t2)))
        # waves'(y mode <= ny/2, x mode, z mode) as [z][x][y]: the 100 branch of waves_set
        tx, ty, tz, kx2, ky2, kz2 = self._t1d_full
        nys = ny // 2 + 1
        TX, KX = tx[None, :, None], kx2[None, :, None]
        TY, KY = ty[:nys][None, None, :], ky2[:nys][None, None, :]
        TZ, KZ = tz[:, None, None], kz2[:, None, None]
        self.waves100 = KX * (TY * TZ) ** 2 + KY * (TX * TZ) ** 2 + KZ * (TX * TY) ** 2
        self._keep = [np.ascontiguousarray(x, dtype=_lib.NP_REAL) for x in
                      (self.waves100, self.ay, self.by, self.ax, self.bx, self.az, self.bz)]
        h = VP()
        _lib.check(lib.x3d_poisson_create(tb, ctypes.byref(h), _lib.ints(ny, nx, nz),
                                          *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self.h = h
        self.poisson = self.poisson_100

    def __del__(self):
        try:
            lib = self.backend.lib
            lib.x3d_poisson_destroy(self.h)
            lib.x3d_block_free(self.tb, self.t1)
            lib.x3d_block_free(self.tb, self.t2)
            lib.x3d_backend_destroy(self.tb)
        except Exception:
            pass

    def poisson_100(self, f, temp):
        lib, b = self.backend.lib, self.backend
        nx, ny, nz = self.nx_glob, self.ny_glob, self.nz_glob
        _lib.check(lib.x3d_transpose_xy(b.h, self.tb, self.t1, f.ptr, nx, ny, nz))
        _lib.check(lib.x3d_poisson_enforce_periodicity_y(self.h, self.t2, self.t1))
        _lib.check(lib.x3d_poisson_fft_forward(self.h, self.t2))
        _lib.check(lib.x3d_poisson_postprocess_010(self.h))
        _lib.check(lib.x3d_poisson_fft_backward(self.h, self.t2))
        _lib.check(lib.x3d_poisson_undo_periodicity_y(self.h, self.t1, self.t2))
        _lib.check(lib.x3d_transpose_xy(self.tb, b.h, f.ptr, self.t1, ny, nx, nz))

    def _unsupported(self, *a):
        raise X3dError("Poisson 100: only solve_poisson is provided (the hooks act on the transposed problem)")

    fft_forward = fft_backward = fft_postprocess_000 = fft_postprocess_010 = _unsupported
    enforce_periodicity_y = undo_periodicity_y = get_spectral = set_spectral = _unsupported


class HipPoissonFFT110(HipPoissonFFT100):
    """x and y non-periodic, z periodic.  The reference's poisson_110 (src/poisson_fft.f90:258-273; CUDA backend
    only): enforce_periodicity_xy ; fft_forward_110 = transposed copy to (nz, nx, ny) + R2C along z ;
    fft_postprocess_110 = seven kernels on the spectrum (nz/2+1, nx, ny) ; fft_backward_110 ; undo_periodicity_xy
    (src/backend/cuda/poisson_fft.f90:401-480, 926-989; waves_set :690-733).  Here, as for 100, a twin library
    backend -- of vertex dims (nz, nx, ny) -- with an x3d_poisson whose x is the reference's z, whose y its x and
    whose z its y; the interleaves along x and y act on the twin's y and z, the seven kernels are
    x3d_poisson_postprocess_011."""

    def _create(self):
        backend, mesh = self.backend, self.mesh
        lib = backend.lib
        nx, ny, nz = self.nx_glob, self.ny_glob, self.nz_glob
        tb = VP()
        vd = [int(v) for v in mesh.vert_dims]
        _lib.check(lib.x3d_backend_create(ctypes.byref(tb), _lib.ints(vd[2], vd[0], vd[1]), backend.device.index,
                                          VP(backend.stream.cuda_stream)))
        self.tb = tb
        self.t1, self.t2 = VP(), VP()
        _lib.check(lib.x3d_block_alloc(tb, ctypes.byref(self.t1)))
        _lib.check(lib.x3d_block_alloc(tb, ctypes.byref(self.t2)))
        # waves(z mode <= nz/2, x mode, y mode) as [y][x][z]: the 110 branch of waves_set
        tx, ty, tz, kx2, ky2, kz2 = self._t1d_full
        nzs = nz // 2 + 1
        TX, KX = tx[None, :, None], kx2[None, :, None]
        TY, KY = ty[:, None, None], ky2[:, None, None]
        TZ, KZ = tz[:nzs][None, None, :], kz2[:nzs][None, None, :]
        self.waves110 = KX * (TY * TZ) ** 2 + KY * (TX * TZ) ** 2 + KZ * (TX * TY) ** 2
        self._keep = [np.ascontiguousarray(x, dtype=_lib.NP_REAL) for x in
                      (self.waves110, self.az, self.bz, self.ax, self.bx, self.ay, self.by)]
        h = VP()
        _lib.check(lib.x3d_poisson_create(tb, ctypes.byref(h), _lib.ints(nz, nx, ny),
                                          *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self.h = h
        self.poisson = self.poisson_110

    def poisson_110(self, f, temp):
        lib, b, h, t1, t2 = self.backend.lib, self.backend, self.h, self.t1, self.t2
        nx, ny, nz = self.nx_glob, self.ny_glob, self.nz_glob
        _lib.check(lib.x3d_transpose_xyz_zxy(b.h, self.tb, t1, f.ptr, nx, ny, nz))
        _lib.check(lib.x3d_poisson_enforce_periodicity_y(h, t2, t1))  # along x
        _lib.check(lib.x3d_poisson_enforce_periodicity_z(h, t1, t2))  # along y
        _lib.check(lib.x3d_poisson_fft_forward(h, t1))
        _lib.check(lib.x3d_poisson_postprocess_011(h))
        _lib.check(lib.x3d_poisson_fft_backward(h, t1))
        _lib.check(lib.x3d_poisson_undo_periodicity_z(h, t2, t1))
        _lib.check(lib.x3d_poisson_undo_periodicity_y(h, t1, t2))
        _lib.check(lib.x3d_transpose_zxy_xyz(self.tb, b.h, f.ptr, t1, nx, ny, nz))


class HipPencilPoissonFFT(HipPoissonFFT):
    """000 solver over a [1, py, pz] decomposition: local rocFFT stages in
    libx3d2_hip.so (csrc/pfft.hip), two pencil transposes per direction as
    packed point-to-point exchanges inside the py and pz rank groups (the
    reference's CPU backend gets the same from 2decomp&FFT,
    src/backend/omp/poisson_fft.f90:72-97).

    Overlap (poisson_000): the local z planes go through the stages before the z transform in `parts` groups
    (X3D_PENCIL_PARTS, default: the library's choice, 4 where zl allows; 1 = the stages one after the other): the
    transfers of a group -- xy exchange inside the py ranks, yz exchange inside the pz ranks, started on the
    communication stream (parallel.Comm.ialltoallv) -- run beside the transforms and packing of the others.  The
    hooks fft_forward / fft_postprocess_000 / fft_backward keep the reference's blocking order."""

    def _create(self):
        import torch
        backend, mesh = self.backend, self.mesh
        if int(mesh.nproc_dir[0]) != 1:
            raise X3dError("FFT Poisson solver: nproc_dir in x-dir must be 1")
        self.py, self.pz = int(mesh.nproc_dir[1]), int(mesh.nproc_dir[2])
        self.ry, self.rz = int(mesh.nrank_dir[1]), int(mesh.nrank_dir[2])
        h = VP()
        _lib.check(backend.lib.x3d_pfft_create_parts(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob), self.py, self.pz,
            self.ry, self.rz, int(os.environ.get("X3D_PENCIL_PARTS", "0"))))
        self.h = h
        sz = (ctypes.c_long * 8)()
        _lib.check(backend.lib.x3d_pfft_sizes(h, sz))
        self.xs, self.xoff, self.ys, self.yoff, self.yl, self.zl, nxs, nmax = [int(v) for v in sz]
        lay = (ctypes.c_long * 6)()
        _lib.check(backend.lib.x3d_pfft_part_layout(h, lay))
        self.parts, self.zp = int(lay[0]), int(lay[1])
        self.piece = [2 * int(v) for v in lay[2:6]]  # doubles per group: xy send, xy recv, yz send, yz recv
        # this rank's spectral block, z fastest: waves[x, y, z]
        xsl, ysl = slice(self.xoff, self.xoff + self.xs), slice(self.yoff, self.yoff + self.ys)
        full = self.waves_block(xsl, ysl)                   # [z, y, x] of this rank's modes only
        wl = np.ascontiguousarray(np.transpose(full, (2, 1, 0)))  # [x, y, z]
        self._keep = [np.ascontiguousarray(a, dtype=_lib.NP_REAL) for a in
                      (wl, self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_pfft_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self.sendbuf = torch.zeros(2 * nmax, dtype=_lib.torch_real(), device=backend.device)
        self.recvbuf = torch.zeros(2 * nmax, dtype=_lib.torch_real(), device=backend.device)

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
        if self.parts > 1:
            # a second buffer pair: a group's xy and yz transfers are in flight at the same time
            self.sendbuf2 = torch.zeros(2 * nmax, dtype=_lib.torch_real(), device=backend.device)
            self.recvbuf2 = torch.zeros(2 * nmax, dtype=_lib.torch_real(), device=backend.device)
            zp, cum = self.zp, lambda v: [sum(v[:i]) for i in range(len(v))]
            self.g_xy_send = [c * x * self.yl * zp for x in x_sh]
            self.g_xy_recv = [c * self.xs * self.yl * zp] * self.py
            self.g_yz_send = [c * y * self.xs * zp for y in y_sh]
            self.g_yz_recv = [c * self.ys * self.xs * zp] * self.pz
            self.o_xy_send, self.o_xy_recv = cum(self.g_xy_send), cum(self.g_xy_recv)
            self.o_yz_send, self.o_yz_recv = cum(self.g_yz_send), cum(self.g_yz_recv)

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

    def _ixchg(self, kind, m, sbuf, rbuf, back):
        """group m's exchange, started now; kind 0: xy (py ranks), 1: yz (pz ranks); back: the receive layout is
        sent and the send layout received (the inverse transposes)"""
        if kind == 0:
            so, sc, ro, rc, peers = self.o_xy_send, self.g_xy_send, self.o_xy_recv, self.g_xy_recv, self.peers_y
            ps, pr = self.piece[0], self.piece[1]
        else:
            so, sc, ro, rc, peers = self.o_yz_send, self.g_yz_send, self.o_yz_recv, self.g_yz_recv, self.peers_z
            ps, pr = self.piece[2], self.piece[3]
        if back:
            so, sc, ro, rc, ps, pr = ro, rc, so, sc, pr, ps
        return self.backend.comm.ialltoallv(sbuf, [m * ps + o for o in so], sc, rbuf, [m * pr + o for o in ro], rc,
                                            peers)

    def poisson_000(self, f, temp=None):
        """poisson_000 (src/poisson_fft.f90:216-226) with the groups of planes pipelined through the exchanges"""
        if self.parts == 1:
            return HipPoissonFFT.poisson_000(self, f, temp)
        lib, h, P = self.backend.lib, self.h, range(self.parts)
        s1, r1, s2, r2 = self.sendbuf, self.recvbuf, self.sendbuf2, self.recvbuf2
        a = []
        for m in P:
            _lib.check(lib.x3d_pfft_fwd_a_part(h, f.ptr, s1.data_ptr(), m))
            a.append(self._ixchg(0, m, s1, r1, False))
        b = []
        for m in P:
            a[m].wait()
            _lib.check(lib.x3d_pfft_fwd_b_part(h, r1.data_ptr(), s2.data_ptr(), m))
            b.append(self._ixchg(1, m, s2, r2, False))
        for m in P:
            b[m].wait()
            _lib.check(lib.x3d_pfft_fwd_c_part(h, r2.data_ptr(), m))
        _lib.check(lib.x3d_pfft_fft_z(h, 0))
        _lib.check(lib.x3d_pfft_postprocess_000(h))
        _lib.check(lib.x3d_pfft_fft_z(h, 1))
        b = []
        for m in P:  # (r2 / s2 swap roles: what was received forward is sent now)
            _lib.check(lib.x3d_pfft_bwd_c_part(h, r2.data_ptr(), m))
            b.append(self._ixchg(1, m, r2, s2, True))
        a = []
        for m in P:
            b[m].wait()
            _lib.check(lib.x3d_pfft_bwd_b_part(h, s2.data_ptr(), r1.data_ptr(), m))
            a.append(self._ixchg(0, m, r1, s1, True))
        for m in P:
            a[m].wait()
            _lib.check(lib.x3d_pfft_bwd_a_part(h, s1.data_ptr(), f.ptr, m))

    def get_spectral(self):
        raise X3dError("get_spectral: single-rank test hook")

    def set_spectral(self, c):
        raise X3dError("set_spectral: single-rank test hook")


class HipSlabPoissonFFT(HipPoissonFFT):
    """000 solver over a z-slab decomposition [1, 1, pz] with ny = 512 (csrc/sfft.hip): y is local, the strided
    y pass writes / reads the exchange layout directly, one all-to-all pair per solve among all pz ranks (every
    xGMI link of a GPU is used at once), z stage on the received array.  The hooks keep the reference's order
    (src/poisson_fft.f90:45-62: fft_forward ; fft_postprocess_000 ; fft_backward).  With 512 planes per rank the
    whole z stage -- forward transform over the pz chunks, spectral division, inverse -- is ONE kernel launched by
    fft_postprocess_000 (csrc/fft512.hip, k_fft512_peers); for other plane counts fft_forward leaves the full 3-D
    spectrum in this rank's block [nz][ys][nx/2+1], fft_postprocess_000 divides, fft_backward returns.

    Overlap (poisson_000): this rank's share of the y modes travels in `parts` pieces on the communication
    stream (parallel.Comm.ialltoall); the z stage of a piece (transpose, z transform, spectral division, inverse
    transform, transpose) runs as soon as the piece has arrived, beside the transfer of the next ones, and its
    result leaves at once.  X3D_SLAB_PARTS (default 4; 1 = no overlap)."""

    def _create(self):
        import torch
        backend, mesh = self.backend, self.mesh
        self.pz, self.rz = int(mesh.nproc_dir[2]), int(mesh.nrank_dir[2])
        ys = self.ny_glob // self.pz
        parts = int(os.environ.get("X3D_SLAB_PARTS", "4"))
        while parts > 1 and ys % parts:
            parts -= 1
        self.parts = max(parts, 1)
        h = VP()
        _lib.check(backend.lib.x3d_sfft_create_parts(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob), self.pz, self.rz,
            self.parts))
        self.h = h
        sz = (ctypes.c_long * 4)()
        _lib.check(backend.lib.x3d_sfft_sizes(h, sz))
        self.chunk, self.zl, self.ys, nxs = [int(v) for v in sz]
        ysl = slice(self.rz * self.ys, (self.rz + 1) * self.ys)
        wl = np.ascontiguousarray(np.transpose(self.waves_block(slice(None), ysl), (1, 2, 0)),
                                  dtype=_lib.NP_REAL)  # [ys][nx/2+1][nz], z fastest
        if nxs > wl.shape[1]:  # the library pads the spectral rows (pad columns: zeros, wave numbers one)
            wl = np.ascontiguousarray(np.pad(wl, ((0, 0), (0, nxs - wl.shape[1]), (0, 0)), constant_values=1.0))
        self._keep = [wl] + [np.ascontiguousarray(a, dtype=_lib.NP_REAL) for a in
                             (self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_sfft_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        n = 2 * self.pz * self.chunk
        self.sbuf = torch.zeros(n, dtype=_lib.torch_real(), device=backend.device)
        # X3D_EMULATE_ALIAS=1 (one process standing in for an N > 1 run): the "all-to-all with itself" needs no copy when
        # the receive buffer IS the send buffer -- what is left is exactly the kernels an N > 1 run adds
        alias = self.pz == 1 and os.environ.get("X3D_EMULATE_ALIAS") == "1"
        self.rbuf = self.sbuf if alias else torch.zeros(n, dtype=_lib.torch_real(), device=backend.device)
        npy = int(mesh.nproc_dir[1])
        ry = int(mesh.nrank_dir[1])
        self.peers = [ry + npy * r for r in range(self.pz)]
        self.sub = 2 * self.chunk // self.parts  # doubles per (peer, part) message
        self.poisson = self.poisson_000

    def __del__(self):
        try:
            self.backend.lib.x3d_sfft_destroy(self.h)
        except Exception:
            pass

    # S = [peer][part][...], R = [part][peer][...] (csrc/sfft.hip)
    def _send_part(self, m):
        return self.backend.comm.ialltoall(self.sbuf, self.rbuf, self.sub, self.peers, send_off=m * self.sub,
                                           send_stride=self.parts * self.sub, recv_off=m * self.pz * self.sub,
                                           recv_stride=self.sub)

    def _return_part(self, m):
        return self.backend.comm.ialltoall(self.rbuf, self.sbuf, self.sub, self.peers, send_off=m * self.pz * self.sub,
                                           send_stride=self.sub, recv_off=m * self.sub,
                                           recv_stride=self.parts * self.sub)

    def fft_forward(self, f_in):
        lib = self.backend.lib
        _lib.check(lib.x3d_sfft_forward_local(self.h, f_in.ptr, self.sbuf.data_ptr()))
        for hnd in [self._send_part(m) for m in range(self.parts)]:
            hnd.wait()
        _lib.check(lib.x3d_sfft_fft_z(self.h, self.rbuf.data_ptr(), 0))

    def fft_postprocess_000(self):
        _lib.check(self.backend.lib.x3d_sfft_postprocess_000(self.h, self.rbuf.data_ptr()))

    def fft_backward(self, f_out):
        lib = self.backend.lib
        _lib.check(lib.x3d_sfft_fft_z(self.h, self.rbuf.data_ptr(), 1))
        for hnd in [self._return_part(m) for m in range(self.parts)]:
            hnd.wait()
        _lib.check(lib.x3d_sfft_backward_local(self.h, self.sbuf.data_ptr(), f_out.ptr))

    def poisson_000(self, f, temp):
        """fft_forward ; fft_postprocess_000 ; fft_backward (src/poisson_fft.f90:216-226) with the pieces of the
        spectrum pipelined: transfer of piece m + 1 beside the z stage of piece m"""
        lib, h, rb = self.backend.lib, self.h, self.rbuf.data_ptr()
        _lib.check(lib.x3d_sfft_forward_local(h, f.ptr, self.sbuf.data_ptr()))
        there = [self._send_part(m) for m in range(self.parts)]
        back = []
        for m in range(self.parts):
            there[m].wait()
            _lib.check(lib.x3d_sfft_fft_z_part(h, rb, 0, m))
            _lib.check(lib.x3d_sfft_postprocess_000_part(h, rb, m))
            _lib.check(lib.x3d_sfft_fft_z_part(h, rb, 1, m))
            back.append(self._return_part(m))
        for hnd in back:
            hnd.wait()
        _lib.check(lib.x3d_sfft_backward_local(h, self.sbuf.data_ptr(), f.ptr))

    def get_spectral(self):
        raise X3dError("get_spectral: single-rank test hook")

    def set_spectral(self, c):
        raise X3dError("set_spectral: single-rank test hook")


class HipSlabPoissonFFTZ(HipPoissonFFT):
    """000 solver over y slabs [1, py, 1] of 512^3 cells per rank, z-first (csrc/sfftz.hip): the z transforms ride on
    the z operator pairs next to the solve (HipBackend.tds_pair_zfirst -> x3d_sfftz_tds_pair), this rank's spectrum
    C[kz][yl][x] goes through x forward -> all-to-all of the x modes inside the py ranks -> y forward + division + y
    inverse on the received rows, in one kernel -> all-to-all back -> x inverse, in `parts` groups of kz planes
    (X3D_SLAB_PARTS; default 4 on several ranks) so that a group's y stage runs beside the transfers of the others.
    The hooks (fft_forward ; fft_postprocess_000 ; fft_backward, src/poisson_fft.f90:45-62) keep their meaning one by
    one: forward transforms z from the field in memory, x, exchanges, transforms y; postprocess divides; backward
    returns (the y stage's kernel in its forward-only / division-only / inverse-only forms)."""

    def _create(self):
        import torch
        backend, mesh = self.backend, self.mesh
        self.py, self.ry = int(mesh.nproc_dir[1]), int(mesh.nrank_dir[1])
        parts = int(os.environ.get("X3D_SLAB_PARTS", "0"))
        h = VP()
        _lib.check(backend.lib.x3d_sfftz_create(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob), self.py, self.ry, parts))
        self.h = h
        sz = (ctypes.c_long * 16)()
        _lib.check(backend.lib.x3d_sfftz_sizes(h, sz))
        self.parts, self.xs, self.xoff, nbuf = int(sz[0]), int(sz[1]), int(sz[2]), int(sz[3])
        self.kz0 = [int(sz[4 + m]) for m in range(self.parts + 1)]
        # -1 / waves of this rank's modes [kz][x][y] (y fastest): the half axis is z, x is the full axis (mirrored)
        kx = np.arange(self.xoff, self.xoff + self.xs)
        kxm = np.where(kx <= self.nx_glob // 2, kx, self.nx_glob - kx)
        w = self.waves_block(kxm, slice(None), slice(0, self.nz_glob // 2 + 1))      # [kz, y, x]
        with np.errstate(divide="ignore"):
            rw = np.where(w < 1.e-16, 0.0, -1.0 / w)
        rw = np.ascontiguousarray(np.transpose(rw, (0, 2, 1)), dtype=_lib.NP_REAL)     # [kz, x, y]
        self._keep = [rw] + [np.ascontiguousarray(a, dtype=_lib.NP_REAL) for a in
                             (self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_sfftz_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self._keep = None
        self.sbuf = torch.zeros(2 * nbuf, dtype=_lib.torch_real(), device=backend.device)
        alias = self.py == 1 and os.environ.get("X3D_EMULATE_ALIAS") == "1"
        self.rbuf = self.sbuf if alias else torch.zeros(2 * nbuf, dtype=_lib.torch_real(), device=backend.device)
        self.peers = [r for r in range(self.py)]  # rank = ry (x and z undivided)
        self.poisson = self.poisson_000
        # round 5: groups of local y rows x groups of kz planes (csrc/sfftz.hip, "BLOCKS"): X3D_SLAB_YPARTS groups of
        # rows (default 4 on several ranks, 1 in a single process; 1 = rounds 3-4's schedule, kz groups only)
        yp = int(os.environ.get("X3D_SLAB_YPARTS", "0"))
        self.yparts = yp if yp > 0 else (4 if self.py > 1 else 1)
        if 512 % self.yparts:
            raise X3dError("X3D_SLAB_YPARTS must divide 512")
        self.rows = [(a * (512 // self.yparts), 512 // self.yparts) for a in range(self.yparts)]
        rs = os.environ.get("X3D_SLAB_ROWS")  # uneven groups, e.g. "64,160,160,128": a small first group fills the pipeline
        if rs:                                 # sooner, a small last one drains it sooner
            cnt = [int(v) for v in rs.split(",")]
            if sum(cnt) != 512 or min(cnt) <= 0:
                raise X3dError("X3D_SLAB_ROWS must be positive row counts that add up to 512")
            self.yparts = len(cnt)
            self.rows = [(sum(cnt[:a]), cnt[a]) for a in range(len(cnt))]
        self.n_pipelined = 0  # solves through zfirst_solve_pipelined (tests)

    def __del__(self):
        try:
            self.backend.lib.x3d_sfftz_destroy(self.h)
        except Exception:
            pass

    def _xchg(self, m, sbuf, rbuf):
        kzc = self.kz0[m + 1] - self.kz0[m]
        return self.backend.comm.ialltoall(sbuf, rbuf, 2 * 512 * kzc * self.xs, self.peers,
                                           send_off=2 * self.kz0[m] * 512 * 512, recv_off=2 * self.kz0[m] * 512 * 512)

    def _group(self, blocks, sbuf, rbuf):
        """ONE exchange group of the blocks [(y0, nyr, m)]: rows [y0, y0 + nyr) of kz group m -- the rows' piece of every
        (part, peer) chunk is one contiguous run"""
        chunks = []
        for y0, nyr, m in blocks:
            kzc = self.kz0[m + 1] - self.kz0[m]
            chunks.append((2 * nyr * kzc * self.xs, 2 * (self.kz0[m] * 512 * 512 + y0 * kzc * self.xs),
                           2 * 512 * kzc * self.xs))
        return self.backend.comm.ialltoall_chunks(sbuf, rbuf, chunks, self.peers)

    def _pipelined(self, front, behind):
        """the solve between the two z stages, rows groups a = 0 .. A - 1 x kz groups m = 0 .. K - 1:
        front(a) -- rows group a of the spectrum is written (the divergence's z pair, or the z transform of a field) --
        then its x transforms, and it leaves while front(a + 1) runs.  The y stage of kz group m needs m's planes of ALL
        rows: the rows groups before the last travel whole (one group of K runs per peer), the last one kz group by kz
        group, so that the y stage of m starts when (A - 1, m) is in, beside the transfers behind it.  Its result goes
        back at once -- kz groups before the last whole, the last one rows group by rows group, so that rows group a's x
        transforms and behind(a) (the gradient's z pair, or the inverse z transform) start when (a, K - 1) is back,
        beside the transfers of the rows groups behind it.  (A + K - 1) + (K - T + A) exchange groups per solve
        (X3D_SLAB_SCHEDULE=blocks: every (a, m) block its own group, 2 A K)."""
        lib, h, sb, rb = self.backend.lib, self.h, self.sbuf, self.rbuf
        A, K = self.yparts, self.parts
        every = os.environ.get("X3D_SLAB_SCHEDULE") == "blocks"
        there = []
        for a, (y0, nyr) in enumerate(self.rows):
            front(a, y0, nyr)
            for m in range(K):
                _lib.check(lib.x3d_sfftz_x_forward_rows(h, sb.data_ptr(), m, y0, nyr))
            if a < A - 1 and not every:
                there.append((None, self._group([(y0, nyr, m) for m in range(K)], sb, rb)))
            else:
                there += [(m, self._group([(y0, nyr, m)], sb, rb)) for m in range(K)]
        back = []
        # the way back: the last T kz groups travel rows group by rows group (one exchange group of T runs per peer each), so
        # that rows group a's x transforms and z pair start when ITS blocks are in; kz groups before them whole, as their y
        # stages complete.  Default T = K: the whole way back is rows-major -- the y stages have run beside the forward
        # transfers anyway, and the z pairs (1.2 ms per solve at 8 ranks) then overlap all of the transfers instead of the
        # last kz group's only (8 emulated ranks: T = 1 57.8, 2 56.1, 3 55.0, 4 54.7 ms per step; X3D_SLAB_TAIL)
        T = K if every else max(1, min(K, int(os.environ.get("X3D_SLAB_TAIL", str(K)))))
        for m in range(K):
            for k, hnd in there:
                if k is None or k == m:
                    hnd.wait()
            there = [(k, hnd) for k, hnd in there if not (k is None or k == m)]
            _lib.check(lib.x3d_sfftz_y_stage(h, rb.data_ptr(), m, 0))
            if every:
                back += [(a, self._group([(y0, nyr, m)], rb, sb)) for a, (y0, nyr) in enumerate(self.rows)]
            elif m < K - T:
                back.append((None, self._group([(0, 512, m)], rb, sb)))
        if not every:
            back += [(a, self._group([(y0, nyr, m) for m in range(K - T, K)], rb, sb)) for a, (y0, nyr) in enumerate(self.rows)]
        for a, (y0, nyr) in enumerate(self.rows):
            for k, hnd in back:
                if k is None or k == a:
                    hnd.wait()
            back = [(k, hnd) for k, hnd in back if not (k is None or k == a)]
            for m in range(K):
                _lib.check(lib.x3d_sfftz_x_backward_rows(h, sb.data_ptr(), m, y0, nyr))
            behind(a, y0, nyr)
        self.n_pipelined += 1

    # ---- the z-first interface of the fused driver (HipBackend.tds_pair_zfirst / Solver.pressure_correction_fused)
    def zfirst_ok(self):
        return not getattr(self.backend, "lazy", False)

    def zfirst_solve_pipelined(self, in1, in2, out1, out2, t_a0, t_b0, t_a1, t_b1):
        """div = A0(in1) + B0(in2) along z ; p = poisson_000(div) ; out1 = A1(p), out2 = B1(p) along z with the z pairs, the
        x transforms and the exchanges cut into rows groups (_pipelined); False: these operators are not served by the
        z-transforming pair kernels (nothing was done)"""
        lib, h = self.backend.lib, self.h
        if self.yparts <= 1 or not self.backend.zfirst_pairs_ok(t_a0, t_b0) or not self.backend.zfirst_pairs_ok(t_a1, t_b1):
            return False
        flag = ctypes.c_int(0)

        def front(a, y0, nyr):
            _lib.check(lib.x3d_sfftz_tds_pair_rows(h, 0, None, None, in1.ptr, in2.ptr, t_a0.handle, t_b0.handle, y0, nyr,
                                                   ctypes.byref(flag)))
            if not flag.value:
                raise X3dError("y-slab solve: the z pair declined operators its probe had accepted")

        def behind(a, y0, nyr):
            _lib.check(lib.x3d_sfftz_tds_pair_rows(h, 1, out1.ptr, out2.ptr, None, None, t_a1.handle, t_b1.handle, y0, nyr,
                                                   ctypes.byref(flag)))
            if not flag.value:
                raise X3dError("y-slab solve: the z pair declined operators its probe had accepted")
        self._pipelined(front, behind)
        return True

    def zfirst_pair(self, mode, out1, out2, in1, in2, t_a, t_b):
        flag = ctypes.c_int(0)
        ptr = lambda f: f.ptr if f is not None else None
        _lib.check(self.backend.lib.x3d_sfftz_tds_pair(self.h, int(mode), ptr(out1), ptr(out2), ptr(in1), ptr(in2),
                                                       t_a.handle, t_b.handle, ctypes.byref(flag)))
        return bool(flag.value)

    def zfirst_middle(self):
        """spectrum (z transformed) -> x ; exchange ; y + division + y ; exchange ; x, the groups of planes pipelined"""
        lib, h, sb, rb = self.backend.lib, self.h, self.sbuf, self.rbuf
        there = []
        for m in range(self.parts):
            _lib.check(lib.x3d_sfftz_x_forward(h, sb.data_ptr(), m))
            there.append(self._xchg(m, sb, rb))
        back = []
        for m in range(self.parts):
            there[m].wait()
            _lib.check(lib.x3d_sfftz_y_stage(h, rb.data_ptr(), m, 0))
            back.append(self._xchg(m, rb, sb))
        for m in range(self.parts):
            back[m].wait()
            _lib.check(lib.x3d_sfftz_x_backward(h, sb.data_ptr(), m))

    # ---- the reference's hooks
    def fft_forward(self, f_in):
        lib = self.backend.lib
        _lib.check(lib.x3d_sfftz_z(self.h, f_in.ptr, 0))
        hs = []
        for m in range(self.parts):
            _lib.check(lib.x3d_sfftz_x_forward(self.h, self.sbuf.data_ptr(), m))
            hs.append(self._xchg(m, self.sbuf, self.rbuf))
        for m in range(self.parts):
            hs[m].wait()
            _lib.check(lib.x3d_sfftz_y_stage(self.h, self.rbuf.data_ptr(), m, 1))

    def fft_postprocess_000(self):
        for m in range(self.parts):
            _lib.check(self.backend.lib.x3d_sfftz_y_stage(self.h, self.rbuf.data_ptr(), m, 3))

    def fft_backward(self, f_out):
        lib = self.backend.lib
        hs = []
        for m in range(self.parts):
            _lib.check(lib.x3d_sfftz_y_stage(self.h, self.rbuf.data_ptr(), m, 2))
            hs.append(self._xchg(m, self.rbuf, self.sbuf))
        for hnd in hs:
            hnd.wait()
        for m in range(self.parts):
            _lib.check(lib.x3d_sfftz_x_backward(self.h, self.sbuf.data_ptr(), m))
        _lib.check(lib.x3d_sfftz_z(self.h, f_out.ptr, 1))

    def poisson_000(self, f, temp=None):
        """poisson_000 (src/poisson_fft.f90:216-226) on a field in memory"""
        lib, h = self.backend.lib, self.h
        if self.yparts > 1:  # the z transforms of a field in memory cut into rows groups like the z pairs
            self._pipelined(lambda a, y0, nyr: _lib.check(lib.x3d_sfftz_z_rows(h, f.ptr, 0, y0, nyr)),
                            lambda a, y0, nyr: _lib.check(lib.x3d_sfftz_z_rows(h, f.ptr, 1, y0, nyr)))
            return
        _lib.check(lib.x3d_sfftz_z(h, f.ptr, 0))
        self.zfirst_middle()
        _lib.check(lib.x3d_sfftz_z(h, f.ptr, 1))

    def interleaved_rows(self):
        return 0

    def get_spectral(self):
        raise X3dError("get_spectral: single-rank test hook")

    def set_spectral(self, c):
        raise X3dError("set_spectral: single-rank test hook")


class HipSlabPoissonFFT010(HipPoissonFFT):
    """010 solver (non-periodic y, uniform or stretched: the channel case) over a z-slab decomposition [1, 1, pz]
    (csrc/sfft010.hip) -- BASELINE configs[4] on several GPUs.  The reference stops here ("Multiple ranks are not yet
    supported for non-periodic BCs!", src/poisson_fft.f90:177-180): 2decomp's / cuFFTMp's pencils split y in spectral
    space, and process_spectral_010 pairs the rows j, ny - j + 2 while the stretched operator is pentadiagonal along y
    (src/backend/cuda/poisson_fft.f90:822-924).  Here the one transpose pair of a solve splits the x MODES: every rank
    ends up with xs = ceil((nx/2 + 1) / pz) mode columns x all ny rows x all nz modes, so enforce / undo_periodicity_y,
    the paired split, the pentadiagonal solves (factored once, this rank's columns of stretching_matrix only) are the
    single-rank kernels.  Hooks in the reference's order (src/poisson_fft.f90:228-242)."""

    def _create(self):
        import torch
        backend, mesh = self.backend, self.mesh
        if int(mesh.nproc_dir[0]) != 1 or int(mesh.nproc_dir[1]) != 1:
            raise X3dError("Poisson 010 on several ranks: z slabs only (nproc_dir = 1, 1, N)")
        self.pz, self.rz = int(mesh.nproc_dir[2]), int(mesh.nrank_dir[2])
        if self.nx_glob // 2 < self.pz:
            raise X3dError("Poisson 010 on z slabs: fewer x modes than ranks")
        h = VP()
        # the rank's columns travel in groups (overlap of the transfers with the z stage and the pentadiagonal solves);
        # X3D_SLAB_PARTS: how many (0 = the library's choice; default: that on several ranks, 1 in a single process)
        parts = int(__import__("os").environ.get("X3D_SLAB_PARTS", "0" if self.pz > 1 else "1"))
        _lib.check(backend.lib.x3d_sfft010_create_parts(
            backend.h, ctypes.byref(h), _lib.ints(self.nx_glob, self.ny_glob, self.nz_glob), self.pz, self.rz, parts))
        self.h = h
        sz = (ctypes.c_long * 6)()
        _lib.check(backend.lib.x3d_sfft010_sizes(h, sz))
        self.chunk, self.zl, self.xs, self.i0, nxm, self.parts = [int(v) for v in sz]
        self.sub = 2 * self.chunk // self.parts  # doubles per (peer, group) message
        i1 = min(self.i0 + self.xs, nxm)          # (the last rank's columns beyond nx/2 + 1 are padding)
        xsl = slice(self.i0, max(i1, self.i0))
        npad = self.xs - (xsl.stop - xsl.start)
        wl = self.waves_block(xsl)                # [nz][ny][real columns]
        if npad:
            wl = np.pad(wl, ((0, 0), (0, 0), (0, npad)), constant_values=1.0)
        self._keep = [np.ascontiguousarray(wl, dtype=_lib.NP_REAL)] + \
            [np.ascontiguousarray(a, dtype=_lib.NP_REAL) for a in (self.ax, self.bx, self.ay, self.by, self.az, self.bz)]
        _lib.check(backend.lib.x3d_sfft010_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in self._keep]))
        self._keep = None
        if self.stretched_y:
            stretching_matrix(self, mesh, *self._dirps, *self._es, xsl=xsl)
            mats = (self.a_odd, self.a_even) if self.stretched_y_sym else (self.a_full, self.a_full)
            if npad:
                mats = [np.pad(a, ((0, 0), (0, 0), (0, 0), (0, npad))) for a in mats]
            mats = [np.ascontiguousarray(a, dtype=_lib.NP_REAL) for a in mats]
            _lib.check(backend.lib.x3d_sfft010_set_stretching(
                h, int(self.stretched_y_sym), *[a.ctypes.data_as(_lib.c_double_p) for a in mats]))
            if not getattr(self, "keep_matrices", False):
                self.a_odd = self.a_even = self.a_full = None
        n = 2 * self.pz * self.chunk
        self.sbuf = torch.zeros(n, dtype=_lib.torch_real(), device=backend.device)
        alias = self.pz == 1 and __import__("os").environ.get("X3D_EMULATE_ALIAS") == "1"  # (see HipSlabPoissonFFT)
        self.rbuf = self.sbuf if alias else torch.zeros(n, dtype=_lib.torch_real(), device=backend.device)
        self.peers = [r for r in range(self.pz)]  # (x and y undivided: rank = rz)
        self.poisson = self.poisson_010

    def __del__(self):
        try:
            self.backend.lib.x3d_sfft010_destroy(self.h)
        except Exception:
            pass


    def enforce_periodicity_y(self, f_out, f_in):
        _lib.check(self.backend.lib.x3d_sfft010_periodicity_y(self.h, f_out.ptr, f_in.ptr, 0))

    def undo_periodicity_y(self, f_out, f_in):
        _lib.check(self.backend.lib.x3d_sfft010_periodicity_y(self.h, f_out.ptr, f_in.ptr, 1))

    def fft_forward(self, f_in):
        lib = self.backend.lib
        _lib.check(lib.x3d_sfft010_forward_local(self.h, f_in.ptr, self.sbuf.data_ptr()))
        for hnd in [self._send_part(m) for m in range(self.parts)]:
            hnd.wait()
        _lib.check(lib.x3d_sfft010_fft_z(self.h, self.rbuf.data_ptr(), 0))

    def fft_postprocess_010(self):
        _lib.check(self.backend.lib.x3d_sfft010_postprocess_010(self.h, self.rbuf.data_ptr()))

    def fft_backward(self, f_out):
        lib = self.backend.lib
        _lib.check(lib.x3d_sfft010_fft_z(self.h, self.rbuf.data_ptr(), 1))
        for hnd in [self._return_part(m) for m in range(self.parts)]:
            hnd.wait()
        _lib.check(lib.x3d_sfft010_backward_local(self.h, self.sbuf.data_ptr(), f_out.ptr))

    # S = [peer][part][...], R = [part][peer][...] (csrc/sfft010.hip)
    def _send_part(self, m):
        return self.backend.comm.ialltoall(self.sbuf, self.rbuf, self.sub, self.peers, send_off=m * self.sub,
                                           send_stride=self.parts * self.sub, recv_off=m * self.pz * self.sub,
                                           recv_stride=self.sub)

    def _return_part(self, m):
        return self.backend.comm.ialltoall(self.rbuf, self.sbuf, self.sub, self.peers, send_off=m * self.pz * self.sub,
                                           send_stride=self.sub, recv_off=m * self.sub,
                                           recv_stride=self.parts * self.sub)

    def poisson_010(self, f, temp):
        """poisson_010 (src/poisson_fft.f90:228-242) with the column groups pipelined: the z transforms, the paired
        split and the pentadiagonal solves of group m run beside the transfer of the groups behind it"""
        if temp is None:
            raise X3dError("poisson_010 needs a scratch block")
        self.enforce_periodicity_y(temp, f)
        self._solve_rows_interleaved(temp)
        self.undo_periodicity_y(f, temp)

    def _solve_rows_interleaved(self, temp):
        lib, h, rb = self.backend.lib, self.h, self.rbuf.data_ptr()
        _lib.check(lib.x3d_sfft010_forward_local(h, temp.ptr, self.sbuf.data_ptr()))
        there = [self._send_part(m) for m in range(self.parts)]
        back = []
        for m in range(self.parts):
            there[m].wait()
            _lib.check(lib.x3d_sfft010_fft_z_part(h, rb, 0, m))
            _lib.check(lib.x3d_sfft010_postprocess_010_part(h, rb, m))
            _lib.check(lib.x3d_sfft010_fft_z_part(h, rb, 1, m))
            back.append(self._return_part(m))
        for hnd in back:
            hnd.wait()
        _lib.check(lib.x3d_sfft010_backward_local(h, self.sbuf.data_ptr(), temp.ptr))

    def interleaved_rows(self):
        """as HipPoissonFFT.interleaved_rows; the caller's z operators are then the halo forms
        (HipBackend.tds_tile_ok / tds_halo_main / tds_halo_finish with yperm)"""
        return self.ny_glob if self.ny_glob % 2 == 0 and os.environ.get("X3D_NO_YPERM") != "1" else 0

    def solve_interleaved(self, f):
        self._solve_rows_interleaved(f)

    def _no_000(self, *a):
        raise X3dError("HipSlabPoissonFFT010 serves the 010 case only")

    fft_postprocess_000 = poisson_000 = _no_000

    def get_spectral(self):
        raise X3dError("get_spectral: single-rank test hook")

    def set_spectral(self, c):
        raise X3dError("set_spectral: single-rank test hook")
