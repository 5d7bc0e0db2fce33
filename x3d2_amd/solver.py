"""solver_t mirror (/root/reference/src/solver.f90): the "Incompact3D
algorithm" sequencing -- transeq -> time integrator -> pressure correction --
over the backend's operator interface.  No arithmetic on field data here."""
import os
import numpy as np

from .common import (BC_DIRICHLET, BC_NEUMANN, CELL, DIR_C, DIR_X, DIR_Y, DIR_Z, RDR_C2Z, RDR_X2Y, RDR_X2Z,
                     RDR_Y2Z, RDR_Z2C, RDR_Z2X, VERT, Y_FACE, X3dError)
from .tdsops import Dirps
from .time_integrator import TimeIntegrator
from .vector_calculus import VectorCalculus


class SolverConfig:
    """the solver_params namelist (src/config.f90:170-173)"""

    def __init__(self, Re=1600.0, dt=1e-3, n_iters=10, n_output=0, time_intg="RK3", poisson_solver_type="FFT",
                 der1st_scheme="compact6", der2nd_scheme="compact6", interpl_scheme="classic",
                 stagder_scheme="compact6", lowmem_transeq=False, fused=False, n_species=0, pr_species=None):
        self.Re, self.dt, self.n_iters, self.n_output = Re, dt, n_iters, n_output
        self.time_intg, self.poisson_solver_type = time_intg, poisson_solver_type
        self.der1st_scheme, self.der2nd_scheme = der1st_scheme, der2nd_scheme
        self.interpl_scheme, self.stagder_scheme = interpl_scheme, stagder_scheme
        self.lowmem_transeq = lowmem_transeq
        # transported scalars (src/config.f90:36-37, 161-162): Prandtl / Schmidt number per species, default 1
        self.n_species = int(n_species)
        self.pr_species = [1.0] * self.n_species if pr_species is None else [float(x) for x in pr_species]
        # fused = False: op-granular sequences exactly as the reference issues them;
        # fused = True : same arithmetic with reorders / sums / axpy chains folded
        #                into the kernels
        self.fused = fused


def allocate_tdsops(dirps, backend, mesh, der1st_scheme, der2nd_scheme, interpl_scheme, stagder_scheme):
    """src/solver.f90:214-289"""
    d0 = dirps.dir - 1
    d = float(mesh.d[d0])
    bc_start, bc_end = int(mesh.BCs[d0, 0]), int(mesh.BCs[d0, 1])
    # FFT Poisson needs Neumann pressure BCs: Dirichlet -> Neumann for the midpoint operators (:232-245)
    bc_mp_start = BC_NEUMANN if bc_start == BC_DIRICHLET else bc_start
    bc_mp_end = BC_NEUMANN if bc_end == BC_DIRICHLET else bc_end
    n_vert, n_cell = mesh.get_n(dirps.dir, VERT), mesh.get_n(dirps.dir, CELL)
    vds, vds2, vd2s, mds = mesh.vert_ds[d0], mesh.vert_ds2[d0], mesh.vert_d2s[d0], mesh.midp_ds[d0]
    A = backend.alloc_tdsops
    dirps.der1st = A(n_vert, d, "first-deriv", der1st_scheme, bc_start, bc_end, stretch=vds[:n_vert])
    dirps.der1st_sym = A(n_vert, d, "first-deriv", der1st_scheme, bc_start, bc_end, stretch=vds[:n_vert],
                         sym=True)
    dirps.der2nd = A(n_vert, d, "second-deriv", der2nd_scheme, bc_start, bc_end, stretch=vds2[:n_vert],
                     stretch_correct=vd2s[:n_vert])
    dirps.der2nd_sym = A(n_vert, d, "second-deriv", der2nd_scheme, bc_start, bc_end, stretch=vds2[:n_vert],
                         stretch_correct=vd2s[:n_vert], sym=True)
    dirps.stagder_v2p = A(n_cell, d, "stag-deriv", stagder_scheme, bc_mp_start, bc_mp_end, from_to="v2p",
                          stretch=mds[:n_cell])
    dirps.stagder_p2v = A(n_vert, d, "stag-deriv", stagder_scheme, bc_mp_start, bc_mp_end, from_to="p2v",
                          stretch=vds[:n_vert])
    dirps.interpl_v2p = A(n_cell, d, "interpolate", interpl_scheme, bc_mp_start, bc_mp_end, from_to="v2p",
                          stretch=np.ones(n_cell))
    dirps.interpl_p2v = A(n_vert, d, "interpolate", interpl_scheme, bc_mp_start, bc_mp_end, from_to="p2v",
                          stretch=np.ones(n_vert))


class Solver:
    def __init__(self, backend, mesh, cfg=None):
        """src/solver.f90:110-212"""
        cfg = cfg or SolverConfig()
        self.backend, self.mesh, self.cfg = backend, mesh, cfg
        self.xdirps, self.ydirps, self.zdirps = Dirps(DIR_X), Dirps(DIR_Y), Dirps(DIR_Z)
        self.vector_calculus = VectorCalculus(backend)
        al = backend.allocator
        self.u, self.v, self.w = (al.get_block(DIR_X) for _ in range(3))
        self.nvars = 3
        # transported species (src/solver.f90:139-155): one DIR_X block each, nu_i = 1 / (Re Pr_i)
        self.nspecies = int(getattr(cfg, "n_species", 0))
        self.species = []
        self.nu_species = []
        if self.nspecies > 0:
            if len(cfg.pr_species) != self.nspecies:
                raise X3dError("pr_species needs one entry per species")
            self.nvars += self.nspecies
            self.nu_species = [1.0 / cfg.Re / pr for pr in cfg.pr_species]
            self.species = [al.get_block(DIR_X) for _ in range(self.nspecies)]
        self.fused = bool(cfg.fused)
        if self.fused and getattr(backend, "lazy", False):
            raise X3dError("deferred execution (HipBackend(lazy=True)) is for the op-granular driver: the fused driver "
                           "already issues the fused kernels itself and swaps block buffers behind the library's back")
        self.time_integrator = TimeIntegrator(backend, al, cfg.time_intg, self.nvars, fused=self.fused)
        self.dt, self.nu = cfg.dt, 1.0 / cfg.Re
        self.n_iters, self.n_output = cfg.n_iters, cfg.n_output
        self.ngrid = int(np.prod(mesh.get_global_dims(VERT)))
        self.current_iter = 0
        for dp in (self.xdirps, self.ydirps, self.zdirps):
            allocate_tdsops(dp, backend, mesh, cfg.der1st_scheme, cfg.der2nd_scheme, cfg.interpl_scheme,
                            cfg.stagder_scheme)
        if cfg.poisson_solver_type == "FFT":
            backend.init_poisson_fft(mesh, self.xdirps, self.ydirps, self.zdirps)
            self.poisson = self.poisson_fft
        elif cfg.poisson_solver_type == "CG":
            self.poisson = self.poisson_cg
        else:
            raise X3dError('poisson_solver_type is not valid. Use "FFT" or "CG".')
        self.pending_grad = None
        # RK stages whose z launch of transeq also does the stage (transeq_fused): every stage (AB schemes: their one update)
        # (X3D_EPI3_STAGES="1,3" overrides; X3D_NO_EPI3=1: none).  Field passes go away in the first stage (the derivative is
        # not read back) and the last (it is not even stored); a middle stage saves a launch only -- same-box A/B at
        # 512^3 RK3, ms per step: none 43.1-43.4, {1} 42.0, {3} 41.7-41.9, {1,3} 41.4-41.6, {1,2,3} 41.3-41.5
        ns = self.time_integrator.nstage
        e = os.environ.get("X3D_EPI3_STAGES")
        self._epi3_stages = {int(v) for v in e.split(",")} if e else set(range(1, ns + 1))
        self.n_epi3 = 0
        self.rot_request, self.rot_applied = 0.0, False  # rotation forcing handed to transeq_x (transeq_fused)
        self.n_rot_fused = self.n_interleaved = 0        # how often those two fusions were taken (tests)
        self.n_zfirst = 0                                # pressure corrections through the z-first 000 solve
        pf = backend.poisson_fft if cfg.poisson_solver_type == "FFT" else None
        self._zfirst = bool(pf is not None and getattr(pf, "zfirst_ok", lambda: False)())
        if self._zfirst:
            # ... and BOTH z operator pairs around the solve are ones the z-transforming kernels take: asked once, here,
            # so that _zfirst_solve never finds the divergence's pair served and the gradient's declined
            z = self.zdirps
            self._zfirst = (backend.zfirst_pairs_ok(z.interpl_v2p, z.stagder_v2p)
                            and backend.zfirst_pairs_ok(z.interpl_p2v, z.stagder_p2v))
        self.shift_request = None  # device scalar to add to u before transeq_x uses it (field_mean_shift)
        self.pending_walls = None  # wall fields to stamp on u, v, w inside the divergence's first kernels
        # (round 6) the case's next define_BC wants field_mean_shift(u, mean_request): taken along by the kernel that
        # forms the new u (BaseCase.substep); _mean_ready = (u's buffer, target, shift scalar, reduction epoch)
        self.mean_request, self._mean_ready, self.n_mean_taken = None, None, 0
        # readers of field data outside step() (get_field_data) first complete a pending velocity correction
        backend.before_read.append(self.flush_grad)
        # src/solver.f90:206-210: lowmem_transeq selects the variant that gives the x-oriented velocity blocks back to the
        # pool while y and z are worked on.  The fused driver never makes y / z copies (one device layout for every
        # DIR tag): it already holds fewer blocks than transeq_lowmem, the flag changes nothing there.
        self.lowmem_transeq = bool(getattr(cfg, "lowmem_transeq", False))
        if self.fused:
            self.transeq = self.transeq_fused
        else:
            self.transeq = self.transeq_lowmem if self.lowmem_transeq else self.transeq_default
        if self.fused:
            self.pressure_correction = self.pressure_correction_fused

    # ---- src/solver.f90:291-389
    def transeq_default(self, rhs, variables):
        b, al = self.backend, self.backend.allocator
        du, dv, dw = rhs[:3]
        u, v, w = variables[:3]
        b.transeq_x(du, dv, dw, u, v, w, self.nu, self.xdirps)
        u_y, v_y, w_y, du_y, dv_y, dw_y = (al.get_block(DIR_Y) for _ in range(6))
        b.reorder(u_y, u, RDR_X2Y)
        b.reorder(v_y, v, RDR_X2Y)
        b.reorder(w_y, w, RDR_X2Y)
        b.transeq_y(du_y, dv_y, dw_y, u_y, v_y, w_y, self.nu, self.ydirps)
        for f in (u_y, v_y, w_y):
            al.release_block(f)
        b.sum_yintox(du, du_y)
        b.sum_yintox(dv, dv_y)
        b.sum_yintox(dw, dw_y)
        for f in (du_y, dv_y, dw_y):
            al.release_block(f)
        u_z, v_z, w_z, du_z, dv_z, dw_z = (al.get_block(DIR_Z) for _ in range(6))
        b.reorder(u_z, u, RDR_X2Z)
        b.reorder(v_z, v, RDR_X2Z)
        b.reorder(w_z, w, RDR_X2Z)
        b.transeq_z(du_z, dv_z, dw_z, u_z, v_z, w_z, self.nu, self.zdirps)
        for f in (u_z, v_z, w_z):
            al.release_block(f)
        b.sum_zintox(du, du_z)
        b.sum_zintox(dv, dv_z)
        b.sum_zintox(dw, dw_z)
        for f in (du_z, dv_z, dw_z):
            al.release_block(f)
        if self.nspecies > 0:  # :384-387
            self.transeq_species(rhs[3:], variables)

    # ---- src/solver.f90:391-505
    def transeq_lowmem(self, rhs, variables):
        """low-memory variant: u, v, w go back to the pool once their y copies exist, the y copies once the z copies
        exist, and the x-oriented velocity is rebuilt from the z copies at the end (RDR_Z2X) -- `variables[0:3]` and
        self.u / v / w are REBOUND to the new blocks, as the reference rebinds variables(i)%ptr and self%u (:492-497)"""
        b, al = self.backend, self.backend.allocator
        du, dv, dw = rhs[:3]
        u, v, w = variables[:3]
        b.transeq_x(du, dv, dw, u, v, w, self.nu, self.xdirps)
        u_y, v_y, w_y = (al.get_block(DIR_Y) for _ in range(3))
        b.reorder(u_y, u, RDR_X2Y)
        b.reorder(v_y, v, RDR_X2Y)
        b.reorder(w_y, w, RDR_X2Y)
        for f in (u, v, w):  # "now release the x-directional fields for saving memory"
            al.release_block(f)
        du_y, dv_y, dw_y = (al.get_block(DIR_Y) for _ in range(3))
        b.transeq_y(du_y, dv_y, dw_y, u_y, v_y, w_y, self.nu, self.ydirps)
        b.sum_yintox(du, du_y)
        b.sum_yintox(dv, dv_y)
        b.sum_yintox(dw, dw_y)
        for f in (du_y, dv_y, dw_y):
            al.release_block(f)
        u_z, v_z, w_z = (al.get_block(DIR_Z) for _ in range(3))
        b.reorder(u_z, u_y, RDR_Y2Z)
        b.reorder(v_z, v_y, RDR_Y2Z)
        b.reorder(w_z, w_y, RDR_Y2Z)
        for f in (u_y, v_y, w_y):
            al.release_block(f)
        du_z, dv_z, dw_z = (al.get_block(DIR_Z) for _ in range(3))
        b.transeq_z(du_z, dv_z, dw_z, u_z, v_z, w_z, self.nu, self.zdirps)
        b.sum_zintox(du, du_z)
        b.sum_zintox(dv, dv_z)
        b.sum_zintox(dw, dw_z)
        for f in (du_z, dv_z, dw_z):
            al.release_block(f)
        u, v, w = (al.get_block(DIR_X) for _ in range(3))
        b.reorder(u, u_z, RDR_Z2X)
        b.reorder(v, v_z, RDR_Z2X)
        b.reorder(w, w_z, RDR_Z2X)
        for f in (u_z, v_z, w_z):
            al.release_block(f)
        variables[0], variables[1], variables[2] = u, v, w
        self.u, self.v, self.w = u, v, w
        if self.nspecies > 0:  # :499-502
            self.transeq_species(rhs[3:], variables)

    # ---- src/solver.f90:507-601
    def transeq_species(self, rhs, variables):
        """skew-symmetric convection-diffusion of the transported species, velocity grid in and out"""
        b, al = self.backend, self.backend.allocator
        u, v, w = variables[:3]
        for i, r in enumerate(rhs):
            b.transeq_species(r, u, variables[3 + i], self.nu_species[i], self.xdirps, i <= 0)
        for vel, dirps, rdr, summ in ((v, self.ydirps, RDR_X2Y, b.sum_yintox), (w, self.zdirps, RDR_X2Z, b.sum_zintox)):
            d = dirps.dir
            vel_d, spec_d, dspec_d = (al.get_block(d) for _ in range(3))
            b.reorder(vel_d, vel, rdr)
            for i, r in enumerate(rhs):
                b.reorder(spec_d, variables[3 + i], rdr)
                b.transeq_species(dspec_d, vel_d, spec_d, self.nu_species[i], dirps, i <= 0)
                summ(r, dspec_d)
            for f in (vel_d, spec_d, dspec_d):
                al.release_block(f)

    def transeq_species_fused(self, rhs, variables):
        """the same without reorders / sums: y and z contributions accumulate in place"""
        b = self.backend
        vel = variables[:3]
        for i, r in enumerate(rhs):
            spec = variables[3 + i]
            for k, dirps in enumerate((self.xdirps, self.ydirps, self.zdirps)):
                b.transeq_species(r, vel[k], spec, self.nu_species[i], dirps, i <= 0, accumulate=k > 0,
                                  direction=dirps.dir)
            r.set_data_loc(spec.data_loc)

    def transeq_fused(self, rhs, variables, defer=False):
        """transeq_default without the 6 reorders and 6 sum_*intox: every block
        shares one device layout, so the y and z passes read u, v, w in place and
        accumulate straight into du, dv, dw.

        defer: the caller (BaseCase.substep) hands the result straight to the fused RK / AB stage; the z
        contribution to du, dv, dw is then left pending and returned as {buffer address of d*: description}
        for TimeIntegrator to fold into its linear combinations: ("tile", ...) = the component is computed
        inside the stage's kernel (csrc/xscan.hip, k_ytile_transeq<EPI>), ("copies", block, dir) = its result
        waits in pencil layout for the transposing stage kernel (csrc/viax.hip)."""
        b = self.backend
        du, dv, dw = rhs[:3]
        u, v, w = variables[:3]
        b.mesh.get_n(DIR_X, u.data_loc)
        # a rotation forcing the case asked for (ChannelCase.substep) can ride on the x kernel
        rot, self.rot_request, self.rot_applied = self.rot_request, 0.0, False
        # ... and so can the second half of its bulk-velocity shift (ChannelCase.define_BC)
        shift, self.shift_request = self.shift_request, None
        x = self.xdirps
        served = False
        if self.pending_grad is not None:
            # the previous sub-step's velocity correction is still pending (pressure_correction_fused(defer_grad)):
            # the x kernel applies it to each pencil before using it -- together with the channel case's shift and
            # rotation forcing where those were asked for (round 6: k_xwide_transeq3_upd; the shift's mean was then
            # taken of the uncorrected u, BaseCase.correction_deferrable)
            g, self.pending_grad = self.pending_grad, None
            if rot != 0.0 or shift is not None:
                if os.environ.get("X3D_NO_ROT_FUSED") != "1" and \
                        b.transeq_x_update_rot(du, dv, dw, u, v, w, self.nu, x, g, x.stagder_p2v, x.interpl_p2v, -1.0, rot, shift):
                    served = True
                    if rot != 0.0:
                        self.rot_applied = True
                        self.n_rot_fused += 1
            elif b.transeq_x_update(du, dv, dw, u, v, w, self.nu, x, g, x.stagder_p2v, x.interpl_p2v, -1.0):
                served = True
            if not served:
                self._apply_grad(g, u, v, w)
            for f in g:
                b.allocator.release_block(f)
        if served:
            pass
        elif (rot != 0.0 or shift is not None) and os.environ.get("X3D_NO_ROT_FUSED") != "1" and \
                b.transeq_x_rot(du, dv, dw, u, v, w, self.nu, self.xdirps, rot, shift):
            if rot != 0.0:
                self.rot_applied = True  # (ChannelCase.forcings: nothing left to do)
                self.n_rot_fused += 1
        else:
            if shift is not None:
                b.field_shift_by(u, shift)
            b.transeq_dir(DIR_X, du, dv, dw, u, v, w, self.nu, self.xdirps, accumulate=False)
        if b._decomposed(DIR_Y) or b._decomposed(DIR_Z):
            self._transeq_yz_decomposed(du, dv, dw, u, v, w)
            for f in rhs[:3]:
                f.set_data_loc(u.data_loc)
            if self.nspecies > 0:
                self.transeq_species_fused(rhs[3:], variables)
            return None
        b.transeq_dir(DIR_Y, du, dv, dw, u, v, w, self.nu, self.ydirps, accumulate=True)
        # z last: its accumulation may be deferred (when z is decomposed it is not: the y components then
        # take the tile kernel K3y, which is cheaper than transposed copies + the fused RK stage)
        pending = None
        ti = self.time_integrator
        # (a rotation forcing the x kernel did not take is applied by the case's forcings() to the complete derivatives)
        defer = defer and not (rot != 0.0 and not self.rot_applied)
        if (defer and os.environ.get("X3D_NO_DEFER") != "1" and os.environ.get("X3D_NO_EPI3") != "1"
                and not b._decomposed(DIR_Z) and self.nspecies == 0
                and ti.istage in self._epi3_stages and b.transeq_stage_ok(DIR_Z, self.zdirps)):
            # round 5: the z launch of the three components also does the RK stage of u, v, w (k_ytile_transeq3<EPI>,
            # HipBackend.transeq_lincomb3); TimeIntegrator.runge_kutta_fused issues it
            pending = {"tile3": (DIR_Z, self.nu, self.zdirps)}
            for f in rhs[:3]:
                f.set_data_loc(u.data_loc)
            return pending
        if defer and os.environ.get("X3D_NO_DEFER") != "1" and not b._decomposed(DIR_Z):
            if b.transeq_stage_ok(DIR_Z, self.zdirps):
                if os.environ.get("X3D_STAGE_IN_TILE") != "1":
                    # default: tile kernel (accumulating) + plain lincomb; the stage inside the tile kernel saves
                    # one read of d* but runs its extra streams at the tile pattern's rate: 59.15 vs 59.4 ms per
                    # step at 512^3 (same-box A/B) -- opt-in
                    pending = None
                else:
                    # the z components are computed inside the stage's linear combinations; entries in the
                    # integrator's variable order, w (the advecting component of z) last: its own update comes last
                    pending = {du.data.data_ptr(): ("tile", DIR_Z, 1, u.ptr, w.ptr, self.nu, self.zdirps),
                               dv.data.data_ptr(): ("tile", DIR_Z, 1, v.ptr, w.ptr, self.nu, self.zdirps),
                               dw.data.data_ptr(): ("tile", DIR_Z, 0, w.ptr, w.ptr, self.nu, self.zdirps)}
            else:
                al = b.allocator
                pend = [al.get_block(DIR_X) for _ in range(3)]
                if b.transeq_dir_defer(DIR_Z, pend, u, v, w, self.nu, self.zdirps):
                    pending = {r.data.data_ptr(): ("copies", pf, DIR_Z) for r, pf in zip((du, dv, dw), pend)}
                else:
                    for pf in pend:
                        al.release_block(pf)
        if pending is None:
            b.transeq_dir(DIR_Z, du, dv, dw, u, v, w, self.nu, self.zdirps, accumulate=True)
        for f in rhs[:3]:
            f.set_data_loc(u.data_loc)
        if self.nspecies > 0:
            self.transeq_species_fused(rhs[3:], variables)
        return pending

    def _transeq_yz_decomposed(self, du, dv, dw, u, v, w):
        """the y and z contributions when at least one of the two directions is decomposed: the boundary rows
        of u, v, w travel while the first half of a local direction is computed, the decomposed directions run
        their single-pass kernels, their boundary values travel while the second half of the local direction is
        computed, and the boundary strips are corrected last (HipBackend.transeq_halo_*).  Directions the
        single-pass kernels do not take use the two-sweep DistD2 form with its own exchanges."""
        b, nu = self.backend, self.nu
        dirs = ((DIR_Y, self.ydirps), (DIR_Z, self.zdirps))
        halo = [(d, dp) for d, dp in dirs if b.halo_tile_ok(d, dp)]
        slow = [(d, dp) for d, dp in dirs if b._decomposed(d) and not b.halo_tile_ok(d, dp)]
        local = [(d, dp) for d, dp in dirs if not b._decomposed(d)]
        hs = {d: b.transeq_halo_begin(d, u, v, w) for d, _ in halo}
        second = []
        for d, dp in local:
            nall = int(b.mesh.vert_dims[2] if d == DIR_Y else b.mesh.vert_dims[1])
            h = nall // 2 if halo else nall
            if b.transeq_planes(d, du, dv, dw, u, v, w, nu, dp, True, 0, h):
                if h < nall:
                    second.append((d, dp, h, nall - h))
            else:
                b.transeq_dir(d, du, dv, dw, u, v, w, nu, dp, accumulate=True)
        hb = {d: b.transeq_halo_main(d, du, dv, dw, u, v, w, nu, dp, True, hs[d]) for d, dp in halo}
        for d, dp, o0, no in second:
            b.transeq_planes(d, du, dv, dw, u, v, w, nu, dp, True, o0, no)
        for d, dp in slow:
            b.transeq_dir(d, du, dv, dw, u, v, w, nu, dp, accumulate=True)
        for d, dp in halo:
            b.transeq_halo_finish(d, du, dv, dw, u, v, w, nu, dp, hb[d])

    def pressure_correction_fused(self, u, v, w, defer_grad=False):
        """pressure_correction (:693-739) = divergence_v2c + Poisson + gradient_c2v +
        velocity update with the 10 reorders removed and the 5 vecadd's folded
        into the accumulating form of the last tds_solve of each chain."""
        b, al = self.backend, self.backend.allocator
        x, y, z = self.xdirps, self.ydirps, self.zdirps
        t1, t2, t3, a1, a2 = (al.get_block(DIR_X) for _ in range(5))
        # divergence_v2c, src/vector_calculus.f90:142-246
        # (a velocity update the time integrator left pending is formed by the operator's own kernel)
        # (pending_walls: wall values the case left to be stamped on the new velocity, ChannelCase.deferred_walls)
        upd = self.time_integrator.pending_update
        walls, self.pending_walls = self.pending_walls or (None, None, None), None
        mean, self.mean_request, self._mean_ready = self.mean_request, None, None
        for out, fld, op, wall in ((t1, u, x.stagder_v2p, walls[0]), (t2, v, x.interpl_v2p, walls[1]),
                                   (t3, w, x.interpl_v2p, walls[2])):
            spec = upd.pop(fld.data.data_ptr(), None)
            take_mean = fld is u and mean is not None and defer_grad and b.comm.size == 1 and not b._decomposed(DIR_X)
            if spec is None:
                if wall is not None:
                    b.field_set_face_from_field(fld, wall, 0.0, Y_FACE)
                if take_mean:  # (the stage was done by transeq's z launch: the operator's own kernel sums the new u)
                    sh = b.tds_apply_mean(out, fld, op, DIR_X, mean)
                    self._mean_ready = (u.data.data_ptr(), float(mean), sh, b.red_epoch)
                else:
                    b.tds_apply(out, fld, op, DIR_X)
            elif take_mean:
                sh = b.tds_lincomb(out, op, DIR_X, *spec, wall=wall, mean_target=mean)
                self._mean_ready = (u.data.data_ptr(), float(mean), sh, b.red_epoch)
            else:
                b.tds_lincomb(out, op, DIR_X, *spec, wall=wall)
        self.time_integrator.flush_updates()
        div = t1
        jy = [(0, a1, None, t1, t2, y.interpl_v2p, y.stagder_v2p),   # a1 = interpl(t1) + stagder(t2)
              (2, a2, None, t3, None, y.interpl_v2p, None)]
        jz = [(0, div, None, a1, a2, z.interpl_v2p, z.stagder_v2p)]
        # 010 Poisson solve (non-periodic y): its interleave of the y rows before / after the transforms is done by
        # the z pairs on either side (4 field passes less per solve); nil = rows to interleave, 0 = not on offer
        nil = 0
        if self.cfg.poisson_solver_type == "FFT" and not b._decomposed(DIR_Y):
            nil = getattr(b.poisson_fft, "interleaved_rows", lambda: 0)()
        if self._zfirst and b._decomposed(DIR_Y) and not b._decomposed(DIR_Z):
            # y slabs: z is whole on this rank, the z-first solve applies as on one rank (csrc/sfftz.hip)
            b.tds_jobs(DIR_Y, jy)
            if self._zfirst_solve(a1, a2, t2, t3):
                b.tds_jobs(DIR_Y, [(1, a1, a2, t2, None, y.interpl_p2v, y.stagder_p2v),   # p_sx, dpdy_sx
                                   (2, t1, None, t3, None, y.interpl_p2v, None)])         # dpdz_sx
                return self._finish_pressure_correction(u, v, w, (t1, t2, t3, a1, a2), defer_grad)
            b.tds_jobs(DIR_Z, jz)
        elif b._decomposed(DIR_Y) or b._decomposed(DIR_Z):
            # (z slabs: the halo form of the interleaving pair, t2 as above)
            nil = nil if nil and self._zpairs_interleave(nil, jz[0]) else 0
            if nil:
                jz = [(0, t2, None, a1, a2, z.interpl_v2p, z.stagder_v2p)]
                div = t2
                self.n_interleaved += 1
            self._div_grad_decomposed(jy, jz, forward=True, nil=nil)
        else:
            b.tds_pair(*jy[0], DIR_Y)
            b.tds_apply(a2, t3, y.interpl_v2p, DIR_Y)
            # 000 solve at 512^3: z-first -- the z pairs on either side transform along z on their tiles, the divergence
            # and the pressure never exist as fields (csrc/zfirst.hip)
            if self._zfirst and self._zfirst_solve(a1, a2, t2, t3):
                b.tds_pair(1, a1, a2, t2, None, y.interpl_p2v, y.stagder_p2v, DIR_Y)   # p_sx, dpdy_sx
                b.tds_apply(t1, t3, y.interpl_p2v, DIR_Y)                              # dpdz_sx
                return self._finish_pressure_correction(u, v, w, (t1, t2, t3, a1, a2), defer_grad)
            if nil and b.tds_pair_yperm(0, t2, None, a1, a2, z.interpl_v2p, z.stagder_v2p, nil):
                div = t2  # (t2, t3 are free after the y stage) rows interleaved
                self.n_interleaved += 1
            else:
                nil = 0
                b.tds_pair(*jz[0], DIR_Z)
        # poisson: the cell-centred divergence is already Cartesian (no Z2C / C2Z)
        p = div
        if nil:
            b.poisson_fft.solve_interleaved(p)
        elif self.cfg.poisson_solver_type == "FFT":
            b.poisson_fft.solve_poisson(p, t2)  # t2 is free here: scratch of poisson_010
        else:
            p.fill(0.0)
        # gradient_c2v, :248-332, + velocity correction solver.f90:731-733
        if nil and b._decomposed(DIR_Z):
            jz = [(1, t1, t3, p, None, z.interpl_p2v, z.stagder_p2v)]    # as below, through the halo forms
            jy = [(1, a1, a2, t1, None, y.interpl_p2v, y.stagder_p2v),
                  (2, t2, None, t3, None, y.interpl_p2v, None)]
            self._div_grad_decomposed(jy, jz, forward=False, nil=nil)
            t1, t2 = t2, t1
        elif nil:
            # p = t2 in the solver's row order: read through the interleave; p_sxy -> t1, dpdz_sxy -> t3
            if not b.tds_pair_yperm(1, t1, t3, p, None, z.interpl_p2v, z.stagder_p2v, nil):
                raise X3dError("pressure_correction: the interleaving z pair served the divergence but not the gradient")
            b.tds_pair(1, a1, a2, t1, None, y.interpl_p2v, y.stagder_p2v, DIR_Y)   # p_sx, dpdy_sx
            b.tds_apply(t2, t3, y.interpl_p2v, DIR_Y)                              # dpdz_sx
            t1, t2 = t2, t1  # (below: t1 = dpdz_sx, t2 and t3 free)
        else:
            jz = [(1, t2, t3, p, None, z.interpl_p2v, z.stagder_p2v)]    # p_sxy, dpdz_sxy
            jy = [(1, a1, a2, t2, None, y.interpl_p2v, y.stagder_p2v),   # p_sx, dpdy_sx
                  (2, t1, None, t3, None, y.interpl_p2v, None)]          # dpdz_sx
            if b._decomposed(DIR_Y) or b._decomposed(DIR_Z):
                self._div_grad_decomposed(jy, jz, forward=False)
            else:
                b.tds_pair(*jz[0], DIR_Z)
                b.tds_pair(*jy[0], DIR_Y)
                b.tds_apply(t1, t3, y.interpl_p2v, DIR_Y)
        self._finish_pressure_correction(u, v, w, (t1, t2, t3, a1, a2), defer_grad)

    def _zfirst_solve(self, a1, a2, t2, t3):
        """div = interpl_z(a1) + stagder_z(a2) ; p = poisson(div) ; t2 = interpl_z(p), t3 = stagder_z(p) with the z
        transforms of the 000 solve on the tiles of the two z pairs; False: not served for these operators, nothing done"""
        b, z = self.backend, self.zdirps
        pipe = getattr(b.poisson_fft, "zfirst_solve_pipelined", None)
        if pipe is not None and not b._decomposed(DIR_Z) and \
                pipe(a1, a2, t2, t3, z.interpl_v2p, z.stagder_v2p, z.interpl_p2v, z.stagder_p2v):
            # y slabs: the z pairs, the x transforms and the all-to-alls in blocks of rows x kz planes (csrc/sfftz.hip)
            self.n_zfirst += 1
            return True
        if not b.tds_pair_zfirst(0, None, None, a1, a2, z.interpl_v2p, z.stagder_v2p):
            return False
        b.poisson_fft.zfirst_middle()
        if not b.tds_pair_zfirst(1, t2, t3, None, None, z.interpl_p2v, z.stagder_p2v):
            raise X3dError("pressure_correction: the z-first pair served the divergence but not the gradient")
        self.n_zfirst += 1
        return True

    def _finish_pressure_correction(self, u, v, w, blocks, defer_grad):
        """velocity correction from (a1, a2, t1) = (dpdx_sx, dpdy_sx, dpdz_sx) -- the last x operators of gradient_c2v"""
        al = self.backend.allocator
        t1, t2, t3, a1, a2 = blocks
        if (defer_grad and os.environ.get("X3D_NO_DEFER") != "1" and os.environ.get("X3D_NO_DEFER_GRAD") != "1"
                and self.nspecies == 0):
            # the next sub-step's transeq_x applies the correction inside its own kernel (transeq_fused)
            self.pending_grad = (a1, a2, t1)
            for f in (t2, t3):
                al.release_block(f)
            return
        self._apply_grad((a1, a2, t1), u, v, w)
        for f in (t1, t2, t3, a1, a2):
            al.release_block(f)

    def _zpairs_interleave(self, nil, jz0):
        """z slabs, y local: both z pairs around the 010 solve are served by the interleaving halo form"""
        b, z = self.backend, self.zdirps
        if b._decomposed(DIR_Y) or not b._decomposed(DIR_Z) or int(b.mesh.vert_dims[2]) <= 2 * 4:
            return False
        back = (1, jz0[1], jz0[4], jz0[3], None, z.interpl_p2v, z.stagder_p2v)  # (a probe: nothing is written)
        return b.tds_tile_ok(DIR_Z, jz0, True, yperm=nil) and b.tds_tile_ok(DIR_Z, back, True, yperm=nil)

    def _div_grad_decomposed(self, jy, jz, forward, nil=0):
        """the y and z operators of divergence_v2c (forward: y then z) / gradient_c2v (z then y) with at least
        one decomposed direction.  Exchanges are hidden behind the planes that do not need them:
        forward, z decomposed, y local: the y jobs do the 4 + 4 boundary z planes first, their rows leave for the
          neighbours, the remaining planes follow;
        backward, z decomposed, y local: while the z job's boundary values travel the y jobs run on the z planes
          its strip correction does not touch, the strip planes follow the correction."""
        b = self.backend
        ly, lz = not b._decomposed(DIR_Y), not b._decomposed(DIR_Z)
        nz = int(b.mesh.vert_dims[2])
        if forward:
            if ly and not lz and nz > 2 * 4 and b.tds_tile_ok(DIR_Z, jz[0], True):
                state = {}
                b.tds_jobs(DIR_Y, jy, lead=[(0, 4), (nz - 4, 4)],
                           after_lead=lambda: state.update(h=[b.tds_halo_begin(DIR_Z, j, k) for k, j in enumerate(jz)]))
                hb = [b.tds_halo_main(DIR_Z, j, k, state["h"][k], yperm=nil) for k, j in enumerate(jz)]
                for k, j in enumerate(jz):
                    b.tds_halo_finish(DIR_Z, j, k, hb[k], yperm=nil)
            else:
                if nil:
                    raise X3dError("divergence: the interleaving z pair was promised (_zpairs_interleave) but not taken")
                b.tds_jobs(DIR_Y, jy)
                b.tds_jobs(DIR_Z, jz)
            return
        if ly and not lz and b.tds_tile_ok(DIR_Z, jz[0], True) and all(b.tds_tile_ok(DIR_Y, j, False) for j in jy):
            ws, we = b.halo_strip_rows(jz[0][5], jz[0][6])
            if ws + we < nz:
                def interior():
                    for j in jy:
                        b.tds_tile_planes(DIR_Y, j, ws, nz - ws - we)
                b.tds_jobs(DIR_Z, jz, between=interior, yperm=nil)
                for j in jy:
                    b.tds_tile_planes(DIR_Y, j, 0, ws)
                    b.tds_tile_planes(DIR_Y, j, nz - we, we)
                return
        b.tds_jobs(DIR_Z, jz, yperm=nil)
        b.tds_jobs(DIR_Y, jy)

    def _apply_grad(self, g, u, v, w):
        """velocity correction, src/solver.f90:731-733 folded into the last x operators of gradient_c2v"""
        b, x = self.backend, self.xdirps
        b.tds_apply(u, g[0], x.stagder_p2v, DIR_X, accumulate=True, scale=-1.0)
        b.tds_apply(v, g[1], x.interpl_p2v, DIR_X, accumulate=True, scale=-1.0)
        b.tds_apply(w, g[2], x.interpl_p2v, DIR_X, accumulate=True, scale=-1.0)

    def take_mean_shift(self, f, target):
        """the device scalar of field_mean_shift(f, target) if the kernel that formed f took the integral along
        (pressure_correction_fused) and f has not changed since -- its velocity correction is still pending -- and no other
        reduction has used the backend's buffer; else None"""
        r, self._mean_ready = self._mean_ready, None
        if (r is None or self.pending_grad is None or r[0] != f.data.data_ptr() or r[1] != float(target)
                or r[3] != self.backend.red_epoch):
            return None
        self.n_mean_taken += 1
        return r[2]

    def flush_grad(self):
        """apply a pending velocity correction now (anything that reads u, v, w before the next transeq)"""
        if self.pending_grad is not None:
            g, self.pending_grad = self.pending_grad, None
            self._apply_grad(g, self.u, self.v, self.w)
            for f in g:
                self.backend.allocator.release_block(f)

    # ---- :603-651
    def divergence_v2p(self, div_u, u, v, w):
        x, y, z = self.xdirps, self.ydirps, self.zdirps
        self.vector_calculus.divergence_v2c(div_u, u, v, w, x.stagder_v2p, x.interpl_v2p, y.stagder_v2p,
                                            y.interpl_v2p, z.stagder_v2p, z.interpl_v2p)

    def gradient_p2v(self, dpdx, dpdy, dpdz, pressure):
        x, y, z = self.xdirps, self.ydirps, self.zdirps
        self.vector_calculus.gradient_c2v(dpdx, dpdy, dpdz, pressure, x.stagder_p2v, x.interpl_p2v,
                                          y.stagder_p2v, y.interpl_p2v, z.stagder_p2v, z.interpl_p2v)

    def curl(self, o_i_hat, o_j_hat, o_k_hat, u, v, w):
        self.vector_calculus.curl(o_i_hat, o_j_hat, o_k_hat, u, v, w, self.xdirps.der1st,
                                  self.ydirps.der1st, self.zdirps.der1st)

    # ---- :653-691
    def poisson_fft(self, pressure, div_u):
        b, al = self.backend, self.backend.allocator
        p_temp = al.get_block(DIR_C)
        b.reorder(p_temp, div_u, RDR_Z2C)
        temp = al.get_block(DIR_C)
        b.poisson_fft.solve_poisson(p_temp, temp)
        al.release_block(temp)
        b.reorder(pressure, p_temp, RDR_C2Z)
        al.release_block(p_temp)

    def poisson_cg(self, pressure, div_u):
        pressure.fill(0.0)  # placeholder in the reference too, :680-691

    # ---- :693-739
    def pressure_correction(self, u, v, w):
        b, al = self.backend, self.backend.allocator
        div_u = al.get_block(DIR_Z)
        self.divergence_v2p(div_u, u, v, w)
        p = al.get_block(DIR_Z)
        self.poisson(p, div_u)
        al.release_block(div_u)
        dpdx, dpdy, dpdz = (al.get_block(DIR_X) for _ in range(3))
        self.gradient_p2v(dpdx, dpdy, dpdz, p)
        al.release_block(p)
        b.vecadd(-1.0, dpdx, 1.0, u)
        b.vecadd(-1.0, dpdy, 1.0, v)
        b.vecadd(-1.0, dpdz, 1.0, w)
        for f in (dpdx, dpdy, dpdz):
            al.release_block(f)
