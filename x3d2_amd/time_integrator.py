"""time_intg_t mirror (/root/reference/src/time_integrator.f90): explicit
Adams-Bashforth 1-4 and Runge-Kutta 1-4 expressed through the backend's
veccopy/vecadd, with the same `olds` block ownership and stage counters."""
from .common import DIR_X, X3dError


class TimeIntegrator:
    # rk_a(j, istage, order) and rk_b(j, order), :83-106
    RK_A = {1: ((0.0, 0.0, 0.0),) * 3,
            2: ((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
            3: ((0.5, 0.0, 0.0), (0.0, 3.0 / 4.0, 0.0), (0.0, 0.0, 0.0)),
            4: ((0.5, 0.0, 0.0), (0.0, 0.5, 0.0), (0.0, 0.0, 1.0))}
    RK_B = {1: (1.0, 0.0, 0.0, 0.0), 2: (0.0, 1.0, 0.0, 0.0),
            3: (2.0 / 9.0, 1.0 / 3.0, 4.0 / 9.0, 0.0), 4: (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0)}
    # Adams-Bashforth, :110-118
    AB = {1: (1.0, 0.0, 0.0, 0.0), 2: (1.5, -0.5, 0.0, 0.0),
          3: (23.0 / 12.0, -4.0 / 3.0, 5.0 / 12.0, 0.0),
          4: (55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -3.0 / 8.0)}

    def __init__(self, backend, allocator, method, nvars=3, fused=False):
        self.backend, self.allocator, self.sname = backend, allocator, method
        self.defer_update, self.pending_update = False, {}
        self.fused = fused
        self.gdt = 0.0
        try:
            self.order = int(method[2])
        except (ValueError, IndexError):
            raise X3dError("Error reading integration order")
        if self.order >= 5 or self.order < 1:
            raise X3dError("Integration order >4 is not supported")
        if method[:2] == "AB":
            self.nstep, self.nstage, self.nolds = self.order, 1, self.order - 1
            self.step = self.adams_bashforth_fused if fused else self.adams_bashforth
        elif method[:2] == "RK":
            self.nstep, self.nstage, self.nolds = 1, self.order, self.order
            self.step = self.runge_kutta_fused if fused else self.runge_kutta
        else:
            raise X3dError("Integration method " + method + " is not defined")
        self.nvars, self.istep, self.istage = nvars, 1, 1
        self.olds = [[allocator.get_block(DIR_X) for _ in range(self.nolds)] for _ in range(nvars)]

    def finalize(self):
        for row in self.olds:
            for f in row:
                self.allocator.release_block(f)
        self.olds = []

    def runge_kutta(self, curr, deriv, dt):
        """:166-231"""
        b, ns = self.backend, self.nstage
        a, bb = self.RK_A[ns], self.RK_B[ns]
        self.gdt = bb[self.istage - 1] * dt
        if self.istage == ns:
            for i in range(self.nvars):
                if ns > 1:
                    b.veccopy(curr[i], self.olds[i][0])
                for j in range(1, ns):
                    b.vecadd(bb[j - 1] * dt, self.olds[i][j], 1.0, curr[i])
                b.vecadd(bb[ns - 1] * dt, deriv[i], 1.0, curr[i])
            self.istage = 1
        else:
            for i in range(self.nvars):
                if self.istage == 1:
                    b.veccopy(self.olds[i][0], curr[i])
                b.veccopy(self.olds[i][self.istage], deriv[i])
                if self.istage > 1:
                    b.veccopy(curr[i], self.olds[i][0])
                for j in range(1, self.istage + 1):
                    b.vecadd(a[self.istage - 1][j - 1] * dt, self.olds[i][j], 1.0, curr[i])
            self.istage += 1

    def adams_bashforth(self, curr, deriv, dt):
        """:233-282"""
        b = self.backend
        self.gdt = dt
        nstep = min(self.istep, self.nstep)
        c = self.AB[nstep]
        for i in range(self.nvars):
            b.vecadd(c[0] * dt, deriv[i], 1.0, curr[i])
            for j in range(2, nstep + 1):
                b.vecadd(c[j - 1] * dt, self.olds[i][j - 2], 1.0, curr[i])
            if nstep < self.nstep:
                if self.istep > 1:
                    self._rotate(self.olds[i], nstep)
            elif self.nstep > 2:
                self._rotate(self.olds[i], nstep - 1)
            if self.nstep > 1:
                b.veccopy(self.olds[i][0], deriv[i])
        self.istep += 1

    # ---- fused forms: same update formulas, one pass per variable.  The
    # veccopy's become block swaps (the Field objects keep their identity, the
    # device buffers change hands) and each vecadd chain becomes one lincomb.
    @staticmethod
    def _swap(a, b):
        a.data, b.data = b.data, a.data

    def _lincomb(self, y, base, coefs, fields, pending, store, var=None):
        """lincomb; a term whose transeq component is still pending (Solver.transeq_fused(defer=True)) is
        completed inside the same kernel.  With defer_update (set by step(..., defer_update=True)) the update
        of the first three variables is not executed but left in self.pending_update for the pressure
        correction, whose first x operators form the new velocity in their own kernel
        (Solver.pressure_correction_fused, csrc/xscan.hip k_xscan_tds_lin)."""
        b = self.backend
        if pending:
            for k, f in enumerate(fields):
                ptr = f.data.data_ptr()
                ent = pending.pop(ptr, None)
                if ent is None:
                    continue
                if ent[0] == "tile":
                    _, direction, kind, u_ptr, conv_ptr, nu, dirps = ent
                    b.transeq_lincomb(direction, kind, u_ptr, conv_ptr, nu, dirps, y, base, coefs, fields, k, store)
                else:
                    _, pf, direction = ent
                    b.lincomb_pending(y, base, coefs, fields, k, pf, direction, store)
                    self.allocator.release_block(pf)
                return
        if self.defer_update and var is not None and var < 3:
            self.pending_update[y.data.data_ptr()] = (y, base, list(coefs), list(fields))
            return
        b.lincomb(y, base, coefs, fields)

    def flush_updates(self):
        """execute the deferred updates nobody consumed"""
        for y, base, coefs, fields in list(self.pending_update.values()):
            self.backend.lincomb(y, base, coefs, fields)
        self.pending_update.clear()

    def _flush(self, pending):
        """complete the pending components no linear combination consumed"""
        for ptr, ent in list((pending or {}).items()):
            if ent[0] == "tile":
                _, direction, kind, u_ptr, conv_ptr, nu, dirps = ent
                self.backend.transeq_component_acc(direction, kind, ptr, u_ptr, conv_ptr, nu, dirps)
            else:
                _, pf, direction = ent
                self.backend.pending_flush(direction, ptr, pf)
                self.allocator.release_block(pf)
        if pending:
            pending.clear()

    def runge_kutta_fused(self, curr, deriv, dt, pending=None, defer_update=False):
        b, ns = self.backend, self.nstage
        self.defer_update = bool(defer_update)
        a, bb = self.RK_A[ns], self.RK_B[ns]
        self.gdt = bb[self.istage - 1] * dt
        # round 5: transeq's z launch is still pending for u, v, w and will do their stage itself (Solver.transeq_fused,
        # HipBackend.transeq_lincomb3): (direction, nu, dirps); the fields are the ones this stage's swaps leave in place
        tile3 = pending.pop("tile3", None) if pending else None
        in_transeq = False
        if self.istage == ns:
            specs = []
            for i in range(self.nvars):
                terms = [(bb[j - 1] * dt, self.olds[i][j]) for j in range(1, ns) if bb[j - 1] != 0.0]
                terms.append((bb[ns - 1] * dt, deriv[i]))
                base = self.olds[i][0] if ns > 1 else curr[i]
                specs.append((curr[i], base, [c for c, _ in terms], [f for _, f in terms], False))
            if tile3 is not None:
                # (the velocity the launch reads is curr itself, overwritten tile by tile behind its last use; the
                #  derivative of the last stage is not needed afterwards: no store)
                in_transeq = self._stage_in_transeq(tile3, deriv[:3], curr[:3], specs[:3])
            for i, sp in enumerate(specs):
                if not (in_transeq and i < 3):
                    self._lincomb(sp[0], sp[1], sp[2], sp[3], pending, False, var=i)
            self._flush(pending)
            self.istage = 1
        else:
            st = self.istage
            specs = []
            for i in range(self.nvars):
                if st == 1:
                    self._swap(self.olds[i][0], curr[i])     # olds1 <- curr (curr is rewritten below)
                self._swap(self.olds[i][st], deriv[i])       # olds_{st+1} <- deriv
                terms = [(a[st - 1][j - 1] * dt, self.olds[i][j]) for j in range(1, st + 1)
                         if a[st - 1][j - 1] != 0.0]
                specs.append((curr[i], self.olds[i][0], [c for c, _ in terms], [f for _, f in terms], True))
            if tile3 is not None:
                # after the swaps the launch's derivative blocks sit in olds[.][st] and, in the first stage, the velocity
                # it reads in olds[.][0]; the derivative must be a term of the combination (a[st][st] != 0: RK2 - RK4)
                rhs = [self.olds[i][st] for i in range(3)]
                vel = [self.olds[i][0] if st == 1 else curr[i] for i in range(3)]
                ok = all(any(f is r for f in sp[3]) for sp, r in zip(specs[:3], rhs))
                in_transeq = self._stage_in_transeq(tile3, rhs, vel, specs[:3]) if ok else self._complete_transeq(tile3, rhs, vel)
            for i, sp in enumerate(specs):
                if in_transeq and i < 3:
                    continue
                if sp[2]:
                    self._lincomb(sp[0], sp[1], sp[2], sp[3], pending, True, var=i)
                else:
                    b.veccopy(sp[0], sp[1])
            self._flush(pending)
            self.istage += 1

    def _complete_transeq(self, tile3, rhs, vel):
        """the pending z launch as a plain accumulating transeq (the stage then runs as usual); returns False"""
        direction, nu, dirps = tile3
        self.backend.transeq_dir(direction, rhs[0], rhs[1], rhs[2], vel[0], vel[1], vel[2], nu, dirps, accumulate=True)
        return False

    def _stage_in_transeq(self, tile3, rhs, vel, specs):
        """transeq's pending z launch with the stage of u, v, w in its store phases (csrc/xscan.hip k_ytile_transeq3<EPI>);
        False: not served for these pencils -- transeq was completed the plain way, the caller runs the stage as usual"""
        direction, nu, dirps = tile3
        if self.backend.transeq_lincomb3(direction, rhs, vel, nu, dirps, specs):
            self.n_stage_in_transeq = getattr(self, "n_stage_in_transeq", 0) + 1
            return True
        return self._complete_transeq(tile3, rhs, vel)

    def adams_bashforth_fused(self, curr, deriv, dt, pending=None, defer_update=False):
        b = self.backend
        self.defer_update = False  # (the AB update rotates its history right after the combination)
        # (round 5: transeq's z launch may be pending for u, v, w -- it then does their update: u += dt sum b_k f_k, in place,
        #  and stores the complete derivative where the history keeps it)
        tile3 = pending.pop("tile3", None) if pending else None
        self._flush(pending)
        self.gdt = dt
        nstep = min(self.istep, self.nstep)
        c = self.AB[nstep]
        specs = []
        for i in range(self.nvars):
            terms = [(c[0] * dt, deriv[i])] + [(c[j - 1] * dt, self.olds[i][j - 2]) for j in range(2, nstep + 1)]
            specs.append((curr[i], curr[i], [x for x, _ in terms], [f for _, f in terms], self.nstep > 1))
        in_transeq = tile3 is not None and self._stage_in_transeq(tile3, deriv[:3], curr[:3], specs[:3])
        for i in range(self.nvars):
            if not (in_transeq and i < 3):
                b.lincomb(curr[i], curr[i], specs[i][2], specs[i][3])
            if nstep < self.nstep:
                if self.istep > 1:
                    self._rotate(self.olds[i], nstep)
            elif self.nstep > 2:
                self._rotate(self.olds[i], nstep - 1)
            if self.nstep > 1:
                self._swap(self.olds[i][0], deriv[i])
        self.istep += 1

    @staticmethod
    def _rotate(sol, n):
        """:284-300"""
        last = sol[n - 1]
        for i in range(n - 1, 0, -1):
            sol[i] = sol[i - 1]
        sol[0] = last
