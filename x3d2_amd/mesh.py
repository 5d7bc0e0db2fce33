"""Host-side mesh / decomposition mirror of the reference's mesh_t
(/root/reference/src/mesh.f90:16-29, 37-194; geometry
src/mesh_content.f90:142-253; par_t :72-102).  Read-only inputs of the hot
path: local extents, BCs per face (BC_HALO on internal faces), neighbours,
grid spacing and stretching factors."""
import math

import numpy as np

from .common import (BC_DIRICHLET, BC_HALO, BC_NAMES, BC_NEUMANN, BC_PERIODIC, CELL, NULL_LOC, VERT,
                     X_EDGE, X_FACE, Y_EDGE, Y_FACE, Z_EDGE, Z_FACE, X3dError)


class Mesh:
    def __init__(self, dims_global, nproc_dir, L_global, BC_x, BC_y, BC_z,
                 stretching=("uniform", "uniform", "uniform"), beta=(1.0, 1.0, 1.0), nrank=0):
        bcs = []
        for pair in (BC_x, BC_y, BC_z):
            try:
                bcs.append([BC_NAMES[p] for p in pair])
            except KeyError:
                raise X3dError("Unknown BC")
        self.BCs_global = np.array(bcs, dtype=int)
        self.periodic_BC = []
        for d in range(3):
            per = self.BCs_global[d] == BC_PERIODIC
            if per.any() and not per.all():
                raise X3dError("BCs are incompatible: in a direction make sure to have "
                               "either both sides periodic or none.")
            self.periodic_BC.append(bool(per.all()))
        self.global_vert_dims = np.array(dims_global, dtype=int)
        self.global_cell_dims = np.array(
            [n if p else n - 1 for n, p in zip(self.global_vert_dims, self.periodic_BC)], dtype=int)
        self.nproc_dir = np.array(nproc_dir, dtype=int)
        self.nproc = int(self.nproc_dir.prod())
        self.nrank = int(nrank)
        if not 0 <= self.nrank < self.nproc:
            raise X3dError("rank outside the decomposition")
        # decomposition_generic, src/mesh.f90:160-194: ranks laid out x fastest
        npx, npy, npz = (int(v) for v in self.nproc_dir)
        rx = self.nrank % npx
        ry = (self.nrank // npx) % npy
        rz = self.nrank // (npx * npy)
        self.nrank_dir = np.array([rx, ry, rz], dtype=int)

        def rank_of(ix, iy, iz):
            return ix + npx * (iy + npy * iz)

        pos = [rx, ry, rz]
        self.pnext, self.pprev = np.zeros(3, dtype=int), np.zeros(3, dtype=int)
        for d in range(3):  # periodic ring per direction, src/mesh_content.f90:72-102
            up, dn = list(pos), list(pos)
            up[d] = (pos[d] + 1) % int(self.nproc_dir[d])
            dn[d] = (pos[d] - 1) % int(self.nproc_dir[d])
            self.pnext[d], self.pprev[d] = rank_of(*up), rank_of(*dn)
        if np.any(self.global_vert_dims % self.nproc_dir):
            raise X3dError("dims_global must be divisible by nproc_dir")
        self.vert_dims = self.global_vert_dims // self.nproc_dir
        self.cell_dims = self.vert_dims.copy()
        for d in range(3):  # copy_vert2cell_dims, src/mesh_content.f90:104-121
            if not self.periodic_BC[d] and self.nrank_dir[d] == self.nproc_dir[d] - 1:
                self.cell_dims[d] -= 1
        self.n_offset = self.vert_dims * self.nrank_dir
        self.BCs = np.zeros((3, 2), dtype=int)
        for d in range(3):  # src/mesh.f90:116-133
            first = self.nrank_dir[d] == 0
            last = self.nrank_dir[d] + 1 == self.nproc_dir[d]
            self.BCs[d, 0] = self.BCs_global[d, 0] if first else BC_HALO
            self.BCs[d, 1] = self.BCs_global[d, 1] if last else BC_HALO
        self.L = np.array(L_global, dtype=np.float64)
        self.d = self.L / self.global_cell_dims
        self.stretching = [str(s) for s in stretching]
        self.beta = np.array(beta, dtype=np.float64)
        self.stretched = [s != "uniform" for s in self.stretching]
        self.alpha = np.zeros(3)  # geo%alpha, src/mesh_content.f90:166, 185
        self._obtain_coordinates()

    def is_root(self):
        return self.nrank == 0

    def _obtain_coordinates(self):
        """src/mesh_content.f90:142-253"""
        pi = 4.0 * math.atan(1.0)
        keys = ("vert_coords", "vert_ds", "vert_ds2", "vert_d2s", "midp_coords", "midp_ds", "midp_ds2",
                "midp_d2s")
        for k in keys:
            setattr(self, k, [None, None, None])
        for dr in range(3):
            nv, nc, off = int(self.vert_dims[dr]), int(self.cell_dims[dr]), int(self.n_offset[dr])
            gv = np.arange(1, nv + 1, dtype=np.float64) + off
            gc = np.arange(1, nc + 1, dtype=np.float64) + off
            if not self.stretched[dr]:
                self.vert_coords[dr] = (gv - 1.0) * self.d[dr]
                self.midp_coords[dr] = (gc - 0.5) * self.d[dr]
                self.vert_ds[dr], self.vert_ds2[dr], self.vert_d2s[dr] = np.ones(nv), np.ones(nv), np.zeros(nv)
                self.midp_ds[dr], self.midp_ds2[dr], self.midp_d2s[dr] = np.ones(nc), np.ones(nc), np.zeros(nc)
                continue
            kind, L, beta = self.stretching[dr], float(self.L[dr]), float(self.beta[dr])
            if beta <= np.finfo(np.float64).eps:
                raise X3dError("Invalid beta in domain_settings")
            if kind not in ("centred", "top-bottom", "bottom"):
                raise X3dError("Invalid stretching type")
            L_inf = L / 2
            alpha = abs((L_inf - math.sqrt((pi * beta) ** 2 + L_inf ** 2)) / (2 * beta * L_inf))
            self.alpha[dr] = alpha
            r = math.sqrt((alpha * beta + 1) / (alpha * beta))
            const = math.sqrt(beta) / (2 * math.sqrt(alpha) * math.sqrt(alpha * beta + 1))
            s = self.d[dr] / L

            def yeta(g, half):
                base = (g - half) * s
                return {"centred": base, "top-bottom": base - 0.5, "bottom": base / 2 - 0.5}[kind]

            def metric(y):
                sy, cy = np.sin(pi * y), np.cos(pi * y)
                coord = const * np.arctan2(r * sy, cy) * (2 * alpha * beta - np.cos(2 * pi * y) + 1) \
                    / (sy ** 2 + alpha * beta) + pi * const
                ds = L * (alpha / pi + sy ** 2 / (pi * beta))
                return coord, ds, ds ** 2, 2 * cy * sy / beta

            cv = metric(yeta(gv, 1.0))
            cm = metric(yeta(gc, 0.5))
            vc, mc, vd2s, md2s = cv[0], cm[0], cv[3], cm[3]
            if kind == "centred":
                vc, mc = vc - L_inf, mc - L_inf
            elif kind == "bottom":
                vc, mc, vd2s, md2s = 2 * vc, 2 * mc, vd2s / 2, md2s / 2
            self.vert_coords[dr], self.vert_ds[dr], self.vert_ds2[dr], self.vert_d2s[dr] = vc, cv[1], cv[2], vd2s
            self.midp_coords[dr], self.midp_ds[dr], self.midp_ds2[dr], self.midp_d2s[dr] = mc, cm[1], cm[2], md2s

    # ---- getters, src/mesh.f90:196-305
    @staticmethod
    def _dims_dataloc(data_loc, v, c):
        table = {VERT: (v[0], v[1], v[2]), CELL: (c[0], c[1], c[2]),
                 X_FACE: (v[0], c[1], c[2]), Y_FACE: (c[0], v[1], c[2]), Z_FACE: (c[0], c[1], v[2]),
                 X_EDGE: (c[0], v[1], v[2]), Y_EDGE: (v[0], c[1], v[2]), Z_EDGE: (v[0], v[1], c[2])}
        if data_loc not in table:
            raise X3dError("Unknown location in get_dims_dataloc")
        return [int(x) for x in table[data_loc]]

    def get_dims(self, data_loc):
        return self._dims_dataloc(data_loc, self.vert_dims, self.cell_dims)

    def get_global_dims(self, data_loc):
        return self._dims_dataloc(data_loc, self.global_vert_dims, self.global_cell_dims)

    def get_n(self, direction, data_loc):
        if data_loc == NULL_LOC:
            raise X3dError("Unknown direction in get_n_dir")
        return self.get_dims(data_loc)[direction - 1]
