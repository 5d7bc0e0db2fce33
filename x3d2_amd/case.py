"""Flow cases: mirror of base_case_t%run (/root/reference/src/case/base_case.f90:181-353),
the TGV case (src/case/tgv.f90) and the monitoring series
(src/postprocess/monitoring.f90:46-90)."""
import time

import numpy as np

from .common import DIR_X, DIR_Z, VERT


class Monitoring:
    """enstrophy = 1/(2N) sum |curl u|^2 ; max / mean |div u|"""

    def __init__(self, solver):
        self.solver = solver
        self.rows = []

    def write_step(self, t, u, v, w):
        s = self.solver
        b, al = s.backend, s.backend.allocator
        du, dv, dw = (al.get_block(DIR_X, VERT) for _ in range(3))
        s.curl(du, dv, dw, u, v, w)
        enstrophy = 0.5 * (b.scalar_product(du, du) + b.scalar_product(dv, dv)
                           + b.scalar_product(dw, dw)) / s.ngrid
        for f in (du, dv, dw):
            al.release_block(f)
        div_u = al.get_block(DIR_Z)
        s.divergence_v2p(div_u, u, v, w)
        div_u_max, div_u_mean = b.field_max_mean(div_u)
        al.release_block(div_u)
        row = (t, enstrophy, div_u_max, div_u_mean)
        self.rows.append(row)
        return row

    def kinetic_energy(self):
        """1/(2N) sum (u^2+v^2+w^2): not written by the reference, asked for by the north star"""
        s = self.solver
        b = s.backend
        return 0.5 * (b.scalar_product(s.u, s.u) + b.scalar_product(s.v, s.v)
                      + b.scalar_product(s.w, s.w)) / s.ngrid


class BaseCase:
    def __init__(self, solver):
        self.solver = solver
        self.monitoring = Monitoring(solver)
        self.step_times = []
        self.initial_conditions()

    # hooks, src/case/base_case.f90:42-46
    def initial_conditions(self):
        raise NotImplementedError

    def define_BC(self):
        pass

    def forcings(self, du, dv, dw, it):
        pass

    def apply_BC(self, u, v, w):
        pass

    def postprocess(self, it, t):
        return self.monitoring.write_step(t, self.solver.u, self.solver.v, self.solver.w)

    def substep(self, it):
        """body of the sub_iter loop, base_case.f90:261-289"""
        s = self.solver
        al = s.backend.allocator
        curr = [s.u, s.v, s.w]
        self.define_BC()
        deriv = [al.get_block(DIR_X) for _ in range(s.nvars)]
        s.transeq(deriv, curr)
        self.forcings(deriv[0], deriv[1], deriv[2], it)
        s.time_integrator.step(curr, deriv, s.dt)
        for f in deriv:
            al.release_block(f)
        self.apply_BC(s.u, s.v, s.w)
        s.pressure_correction(s.u, s.v, s.w)

    def step(self, it):
        for _ in range(self.solver.time_integrator.nstage):
            self.substep(it)

    def run(self, n_iters=None, verbose=False):
        s = self.solver
        n_iters = s.n_iters if n_iters is None else n_iters
        self.postprocess(s.current_iter, s.current_iter * s.dt)
        start = s.current_iter + 1
        for it in range(start, n_iters + 1):
            t0 = time.perf_counter()
            self.step(it)
            s.current_iter = it
            if s.n_output > 0 and it % s.n_output == 0:
                row = self.postprocess(it, it * s.dt)
                if verbose and s.mesh.is_root():
                    print("time = %g iteration = %d enstrophy: %.13e div u max mean: %.3e %.3e"
                          % (row[0], it, row[1], row[2], row[3]))
            s.backend.sync()
            self.step_times.append(time.perf_counter() - t0)
        return self.monitoring.rows


class TGVCase(BaseCase):
    """src/case/tgv.f90:40-72"""

    def initial_conditions(self):
        s = self.solver
        m = s.mesh
        x = m.vert_coords[0][None, None, :]
        y = m.vert_coords[1][None, :, None]
        z = m.vert_coords[2][:, None, None]
        for f in (s.u, s.v, s.w):
            f.set_data_loc(VERT)
        s.backend.set_field_data(s.u, np.sin(x) * np.cos(y) * np.cos(z))
        s.backend.set_field_data(s.v, -np.cos(x) * np.sin(y) * np.cos(z))
        s.w.fill(0.0)
