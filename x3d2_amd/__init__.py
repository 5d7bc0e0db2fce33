"""x3d2_amd: MI355X-native backend for x3d2's per-timestep hot path.

The compute path is libx3d2_hip.so (hand-written HIP for gfx950 behind the C
ABI of include/x3d2_hip.h); the modules here mirror the reference's host-side
operator interface (base_backend_t, tdsops_t, poisson_fft_t, solver_t, ...)."""
from .common import *  # noqa: F401,F403
from .mesh import Mesh  # noqa: F401
from .tdsops import Dirps, Tdsops  # noqa: F401


def make_tgv(n, nproc_dir=(1, 1, 1), rank=0, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT", comm=None,
             device=None, fused=False, n_species=0, pr_species=None, lazy=None, lowmem_transeq=False, L=None):
    """TGV set-up of examples/TGV/input.x3d on an n^3 (or (nx,ny,nz)) grid; L: domain lengths (default 2 pi each)."""
    from .backend import HipBackend
    from .case import TGVCase
    from .solver import Solver, SolverConfig
    dims = (n, n, n) if isinstance(n, int) else tuple(n)
    twopi = 6.283185307179586
    mesh = Mesh(dims, nproc_dir, (twopi,) * 3 if L is None else tuple(L), ("periodic",) * 2, ("periodic",) * 2,
                ("periodic",) * 2, nrank=rank)
    backend = HipBackend(mesh, device=device, comm=comm, lazy=lazy)
    solver = Solver(backend, mesh, SolverConfig(Re=Re, dt=dt, time_intg=time_intg, poisson_solver_type=poisson,
                                                 fused=fused, n_species=n_species, pr_species=pr_species,
                                                 lowmem_transeq=lowmem_transeq))
    return TGVCase(solver)


def make_channel(dims=(128, 65, 64), L=(4.0, 2.0, 2.0), stretching="top-bottom", beta=0.259065151, Re=4200.0,
                 dt=5e-3, time_intg="RK3", poisson="FFT", fused=False, device=None, comm=None, nproc_dir=(1, 1, 1),
                 rank=0, lazy=None, lowmem_transeq=False, **channel_kw):
    """channel set-up of examples/channel/input.x3d: periodic x/z, no-slip y walls (Dirichlet),
    y stretched towards the walls; channel_kw -> ChannelConfig (rotation, omega_rot, n_rotate, noise).
    dims, L: the GLOBAL grid; nproc_dir = (1, 1, N): z slabs (the wall-normal direction stays whole on every rank:
    poisson_fft.HipSlabPoissonFFT010)"""
    from .backend import HipBackend
    from .case import ChannelCase, ChannelConfig
    from .solver import Solver, SolverConfig
    st = ("uniform", stretching, "uniform")
    mesh = Mesh(tuple(dims), tuple(nproc_dir), tuple(L), ("periodic",) * 2, ("dirichlet",) * 2, ("periodic",) * 2,
                st, (1.0, beta, 1.0), nrank=rank)
    backend = HipBackend(mesh, device=device, comm=comm, lazy=lazy)
    solver = Solver(backend, mesh, SolverConfig(Re=Re, dt=dt, time_intg=time_intg, poisson_solver_type=poisson,
                                                 fused=fused, lowmem_transeq=lowmem_transeq))
    return ChannelCase(solver, ChannelConfig(**channel_kw))
