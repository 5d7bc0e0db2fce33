"""x3d2_amd: MI355X-native backend for x3d2's per-timestep hot path.

The compute path is libx3d2_hip.so (hand-written HIP for gfx950 behind the C
ABI of include/x3d2_hip.h); the modules here mirror the reference's host-side
operator interface (base_backend_t, tdsops_t, poisson_fft_t, solver_t, ...)."""
from .common import *  # noqa: F401,F403
from .mesh import Mesh  # noqa: F401
from .tdsops import Dirps, Tdsops  # noqa: F401


def make_tgv(n, nproc_dir=(1, 1, 1), rank=0, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT", comm=None,
             device=None, fused=False):
    """TGV set-up of examples/TGV/input.x3d on an n^3 (or (nx,ny,nz)) grid."""
    from .backend import HipBackend
    from .case import TGVCase
    from .solver import Solver, SolverConfig
    dims = (n, n, n) if isinstance(n, int) else tuple(n)
    twopi = 6.283185307179586
    mesh = Mesh(dims, nproc_dir, (twopi,) * 3, ("periodic",) * 2, ("periodic",) * 2, ("periodic",) * 2,
                nrank=rank)
    backend = HipBackend(mesh, device=device, comm=comm)
    solver = Solver(backend, mesh, SolverConfig(Re=Re, dt=dt, time_intg=time_intg, poisson_solver_type=poisson,
                                                 fused=fused))
    return TGVCase(solver)
