"""Flow cases: mirror of base_case_t%run (/root/reference/src/case/base_case.f90:181-353),
the TGV case (src/case/tgv.f90) and the monitoring series
(src/postprocess/monitoring.f90:46-90)."""
import os
import time

import numpy as np

from .common import BC_DIRICHLET, CELL, DIR_X, DIR_Z, VERT, Y_FACE


class Monitoring:
    """enstrophy = 1/(2N) sum |curl u|^2 ; max / mean |div u|"""

    def __init__(self, solver):
        self.solver = solver
        self.rows = []

    def write_step(self, t, u, v, w):
        s = self.solver
        s.flush_grad()  # a velocity correction left pending by step(more=True) is completed first
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
        s.flush_grad()
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

    def deferred_walls(self):
        """three fields whose y faces are ALL that apply_BC would stamp on u, v, w (then the fused driver does it
        inside the kernels that form the new velocity), or None: apply_BC has to run as it is"""
        return None

    def forcings_idle(self, it):
        """True: forcings() does not touch the derivatives in iteration `it` whenever Solver.transeq_fused leaves the last
        accumulation of transeq pending (the fused driver may then fold it and the RK stage together)"""
        return type(self).forcings is BaseCase.forcings

    def correction_deferrable(self):
        """True: between a sub-step's pressure correction and the next sub-step's transeq_x no hook of this case reads
        the velocity in a way that needs the correction applied -- it may then wait for that kernel
        (Solver.transeq_fused).  The base hooks read nothing; a case that overrides define_BC / apply_BC says so itself."""
        return type(self).define_BC is BaseCase.define_BC and type(self).apply_BC is BaseCase.apply_BC

    def mean_target_of_next_BC(self):
        """the target of the field_mean_shift(u, target) this case's define_BC will ask for at the start of the next
        sub-step, or None"""
        return None

    def postprocess(self, it, t):
        return self.monitoring.write_step(t, self.solver.u, self.solver.v, self.solver.w)

    def substep(self, it, last=True):
        """body of the sub_iter loop, base_case.f90:261-289"""
        s = self.solver
        al = s.backend.allocator
        curr = [s.u, s.v, s.w] + list(s.species)  # base_case.f90:236-241
        self.define_BC()
        deriv = [al.get_block(DIR_X) for _ in range(s.nvars)]
        # nothing reads the new velocity between the stage and the pressure correction when the case has no
        # BC hook: its update may then be formed inside the pressure correction's first kernels
        # (a case whose BC hook only stamps wall values hands them over instead: deferred_walls)
        walls = self.deferred_walls() if s.fused and os.environ.get("X3D_NO_DEFER_WALLS") != "1" else None
        defer_upd = (s.fused and (type(self).apply_BC is BaseCase.apply_BC or walls is not None)
                     and os.environ.get("X3D_NO_DEFER") != "1" and s.time_integrator.sname.upper().startswith("RK"))
        if s.fused and self.forcings_idle(it):
            # nothing touches the derivatives between transeq and the RK / AB stage: the last accumulation of
            # transeq may be folded into the stage's linear combination (Solver.transeq_fused)
            pending = s.transeq(deriv, curr, defer=True)
            if type(self).forcings is not BaseCase.forcings:
                self.forcings(deriv[0], deriv[1], deriv[2], it)  # (idle, see forcings_idle -- or nothing is pending)
            s.time_integrator.step(curr, deriv, s.dt, pending=pending, defer_update=defer_upd)
        else:
            s.transeq(deriv, curr)
            self.forcings(deriv[0], deriv[1], deriv[2], it)
            if s.fused:
                s.time_integrator.step(curr, deriv, s.dt, defer_update=defer_upd)
            else:
                s.time_integrator.step(curr, deriv, s.dt)
        if not defer_upd:
            for f in deriv:
                al.release_block(f)
            deriv = []
        if defer_upd and walls is not None:
            s.pending_walls = walls  # apply_BC happens inside the pressure correction's first kernels
        else:
            self.apply_BC(s.u, s.v, s.w)
        # not the last sub-step of the step and no hook looks at the velocity before the next transeq: its
        # pressure-gradient correction can wait for that kernel (Solver.transeq_fused)
        defer_grad = not last and s.fused and self.correction_deferrable()
        if defer_grad:
            # (a case whose next define_BC wants the volume integral of u says so: the kernel that forms the new u takes it
            #  along, Solver.pressure_correction_fused / take_mean_shift)
            s.mean_request = self.mean_target_of_next_BC()
            s.pressure_correction(s.u, s.v, s.w, defer_grad=True)
        else:
            s.pressure_correction(s.u, s.v, s.w)
        for f in deriv:  # (kept until the deferred updates that read them were done)
            al.release_block(f)

    def step(self, it, more=False):
        """one time step.  more=True: another step follows at once and nothing looks at the fields in between, so
        the last sub-step's velocity correction may also wait for the next transeq_x kernel; whoever reads the
        velocity next without going through step() must call solver.flush_grad() first (run() does)."""
        ns = self.solver.time_integrator.nstage
        for i in range(ns):
            self.substep(it, last=(i == ns - 1 and not more))

    def run(self, n_iters=None, verbose=False):
        s = self.solver
        n_iters = s.n_iters if n_iters is None else n_iters
        self.postprocess(s.current_iter, s.current_iter * s.dt)
        start = s.current_iter + 1
        for it in range(start, n_iters + 1):
            t0 = time.perf_counter()
            output_due = s.n_output > 0 and it % s.n_output == 0
            self.step(it, more=(it < n_iters and not output_due))
            s.current_iter = it
            if s.n_output > 0 and it % s.n_output == 0:
                row = self.postprocess(it, it * s.dt)
                if verbose and s.mesh.is_root():
                    print("time = %g iteration = %d enstrophy: %.13e div u max mean: %.3e %.3e"
                          % (row[0], it, row[1], row[2], row[3]))
            s.backend.sync()
            self.step_times.append(time.perf_counter() - t0)
        s.flush_grad()
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


class ChannelConfig:
    """channel_config_t, src/config.f90:46-54, 207-249 (namelist channel_nml)"""

    def __init__(self, init_noise=(0.0, 0.0, 0.0), inlet_noise=(0.0, 0.0, 0.0), rotation=False, omega_rot=0.0,
                 n_rotate=0, seed=None):
        self.init_noise, self.inlet_noise = tuple(init_noise), tuple(inlet_noise)
        self.rotation, self.omega_rot, self.n_rotate = bool(rotation), float(omega_rot), int(n_rotate)
        self.seed = seed  # the reference draws unseeded random_number; a seed makes runs repeatable


class ChannelCase(BaseCase):
    """src/case/channel.f90: bulk velocity held at 2/3 by a shift of u, optional rotation forcing
    for the first n_rotate iterations, no-slip y walls stamped after each sub-step.

    The reference rebuilds the wall values on the host and uploads three full DIR_C blocks per
    sub-step (:97-130); here the three wall fields live on the device, are zero-filled once when
    inlet_noise = 0 (the example's setting) and only rewritten when noise is asked for."""

    def __init__(self, solver, channel_cfg=None):
        self.channel_cfg = channel_cfg or ChannelConfig()
        self.rng = np.random.default_rng(self.channel_cfg.seed)  # initial condition (host, once)
        seed = self.channel_cfg.seed
        self.noise_seed = int(seed) if seed is not None else int(np.random.SeedSequence().entropy) & (2 ** 63 - 1)
        self.noise_draws = 0
        self.bc_start_y = None
        super().__init__(solver)

    def initial_conditions(self):  # :139-189
        s, m = self.solver, self.solver.mesh
        nx, ny, nz = m.get_dims(VERT)
        y = m.vert_coords[1][None, :, None] - m.L[1] / 2.0
        um = np.exp(-0.2 * y * y)
        noise = [self.channel_cfg.inlet_noise[2]] * 3  # :154 takes inlet_noise(3) for all three
        shape = (nz, ny, nx)
        fields = []
        for c, base in enumerate((1.0 - y * y, 0.0, 0.0)):
            r = self.rng.random(shape) if noise[c] != 0.0 else 0.5
            f = base + noise[c] * um * (2 * r - 1.0) * np.ones(shape)
            f[:, 0, :] = 0.0
            f[:, -1, :] = 0.0
            fields.append(f)
        for f, a in zip((s.u, s.v, s.w), fields):
            f.set_data_loc(VERT)
            s.backend.set_field_data(f, a)

    def define_BC(self):  # :53-137
        s, b = self.solver, self.solver.backend
        # ub -> 2/3, no host round trip on one rank; fused driver: the shift itself is left to transeq_x's kernel
        sh = None
        if s.fused and os.environ.get("X3D_NO_ROT_FUSED") != "1":
            # (round 6: the integral may have been taken by the kernel that formed this u, mean_target_of_next_BC)
            sh = s.take_mean_shift(s.u, 2.0 / 3.0) or b.field_mean_shift(s.u, 2.0 / 3.0)
        if sh is not None:
            s.shift_request = sh
        else:
            b.field_shift_to_mean(s.u, 2.0 / 3.0)
        noise = self.channel_cfg.inlet_noise
        first = self.bc_start_y is None
        if first:
            self.bc_start_y = [b.allocator.get_block(DIR_X, VERT) for _ in range(3)]
            for f in self.bc_start_y:
                f.set_data_loc(VERT)
                f.fill(0.0)
        if any(n != 0.0 for n in noise):
            # the reference draws six random_number planes on the host and uploads three full blocks per
            # sub-step (:97-130); here the wall planes are generated in place (x3d_wall_noise).  `draw` numbers
            # the draws so that a run is repeatable for a given seed (the reference's is not: unseeded)
            for c, (f, n) in enumerate(zip(self.bc_start_y, noise)):
                b.wall_noise(f, n, self.noise_seed, 3 * self.noise_draws + c)
            self.noise_draws += 1

    def substep(self, it, last=True):
        c, s = self.channel_cfg, self.solver
        # fused driver: the rotation forcing is offered to transeq_x's kernel (Solver.transeq_fused); forcings()
        # below applies it with the reference's two vecadd's when that kernel did not take it
        s.rot_request = c.omega_rot if (s.fused and c.rotation and it < c.n_rotate) else 0.0
        super().substep(it, last)

    def forcings(self, du, dv, dw, it):  # :191-207
        c, s = self.channel_cfg, self.solver
        if s.rot_applied:
            s.rot_applied = False
            return
        if c.rotation and it < c.n_rotate:
            s.backend.vecadd(-c.omega_rot, s.v, 1.0, du)
            s.backend.vecadd(c.omega_rot, s.u, 1.0, dv)

    def deferred_walls(self):
        return self.bc_start_y

    def mean_target_of_next_BC(self):
        return 2.0 / 3.0 if os.environ.get("X3D_NO_MEAN_IN_LINCOMB") != "1" else None

    def correction_deferrable(self):
        # define_BC reads the volume integral of u only, and the pending correction does not change it: dp/dx is a
        # periodic compact x derivative (stagder_p2v: A d = B p with circulant A, B and zero row sums of B), so every x
        # pencil of it sums to zero up to rounding; apply_BC's walls are stamped BEFORE the correction (as in the
        # reference: apply_BC ; pressure_correction) by the kernels that form the new velocity (deferred_walls).
        # One rank only: field_mean_shift hands the shift over as a device scalar (X3D_NO_ROT_FUSED=1: host path).
        # Offered where the x kernel takes the correction along: 1024-row x pencils (csrc/xwide.hip, k_xwide_transeq3_upd).
        s = self.solver
        return (s.fused and s.backend.comm.size == 1 and int(s.mesh.get_dims(VERT)[0]) == 1024
                and os.environ.get("X3D_NO_ROT_FUSED") != "1" and os.environ.get("X3D_NO_CHANNEL_DEFER_GRAD") != "1"
                and type(self).define_BC is ChannelCase.define_BC and type(self).apply_BC is ChannelCase.apply_BC)

    def forcings_idle(self, it):
        # no rotation in this iteration, or transeq_x's kernel applies it (rot_request, substep above) -- and where that
        # kernel does not, Solver.transeq_fused leaves nothing pending and forcings() below acts as in the reference
        # (a subclass with its own forcings() must see complete derivatives: ADVICE round 5)
        return type(self).forcings is ChannelCase.forcings

    def apply_BC(self, u, v, w):  # :214-231
        b = self.solver.backend
        for f, st in zip((u, v, w), self.bc_start_y):
            b.field_set_face_from_field(f, st, 0.0, Y_FACE)
