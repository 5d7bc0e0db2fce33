"""BASELINE configs[4] on several ranks: the channel case (non-periodic, stretched y; 010 Poisson solve) on z slabs.
The reference itself stops on this combination (src/poisson_fft.f90:177-180), so the checks are
 (a) the slab solver in ONE process (pz = 1) == the single-rank 010 solver (same kernels, other data path),
 (b) N ranks sharing cuda:0 == the single-rank HIP run (1e-11) == the single-rank oracle (1e-10): y is never split,
     so the single-rank oracle is the reference result for any number of z slabs.
FP64; tolerances stated per assertion."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_hip_poisson_010 import _channel_steps, hip_poisson_solve, oracle_solver, product_solver
from util import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims,stretching,beta,split", [
    ((32, 17, 24), "uniform", 1.0, False), ((34, 17, 24), "top-bottom", 0.259065151, False),
    ((48, 33, 16), "centred", 0.4, False), ((32, 16, 20), "bottom", 0.3, False),
    ((1024, 33, 16), "top-bottom", 0.259065151, False),
    ((34, 17, 24), "top-bottom", 0.259065151, True),   # 1-D x and y plans (where rocFFT refuses the 2-D real plan)
    ((48, 33, 16), "top-bottom", 0.259065151, 3),      # the columns in 3 groups (25 modes -> 27 columns: padding)
    ((1024, 33, 16), "top-bottom", 0.259065151, 4)])
def test_slab_010_solver_in_one_process_equals_the_single_rank_solver(dims, stretching, beta, split, monkeypatch):
    """csrc/sfft010.hip with pz = 1 (pack / all-to-all-with-itself / strided z transform / the spectral kernels on the
    packed layout) against HipPoissonFFT (3-D rocFFT plan) and the oracle, on a seeded right-hand side"""
    if split is True:
        monkeypatch.setenv("X3D_SFFT010_SPLIT_XY", "1")
    elif split:
        monkeypatch.setenv("X3D_SLAB_PARTS", str(split))
    from x3d2_amd.poisson_fft import HipPoissonFFT, HipSlabPoissonFFT010
    rng = np.random.default_rng(11)
    s1 = product_solver(dims, stretching, beta)
    assert type(s1.backend.poisson_fft) is HipPoissonFFT
    nx, ny, nz = s1.mesh.get_dims(1110)
    f = rng.standard_normal((nz, ny, nx))
    f -= f.mean()
    ref = hip_poisson_solve(s1, f)
    monkeypatch.setenv("X3D_FORCE_PENCIL_FFT", "slab")
    s2 = product_solver(dims, stretching, beta)
    assert type(s2.backend.poisson_fft) is HipSlabPoissonFFT010
    got = hip_poisson_solve(s2, f)
    assert relerr(got, ref) < 1e-12
    if split and split is not True:
        pf = s2.backend.poisson_fft
        assert pf.parts == split
        # the hooks one after the other (src/poisson_fft.f90:228-242) == the pipelined solve
        from x3d2_amd.common import CELL, DIR_C
        b, al = s2.backend, s2.backend.allocator
        p_, t_ = al.get_block(DIR_C, CELL), al.get_block(DIR_C)
        p_.fill(0.0)
        b.set_field_data(p_, f, CELL)
        pf.enforce_periodicity_y(t_, p_)
        pf.fft_forward(t_)
        pf.fft_postprocess_010()
        pf.fft_backward(t_)
        pf.undo_periodicity_y(p_, t_)
        assert np.array_equal(b.get_field_data(p_, CELL), got)
        al.release_block(p_); al.release_block(t_)
    o = oracle_solver(dims, stretching, beta)
    assert relerr(got, o.poisson_fft.solve(f)) < 1e-10


def _run_channel_ranks(nproc_dir, dims, nsteps, stretching, beta, fused, tmp_path, timeout=900):
    nproc = int(np.prod(nproc_dir))
    out = str(tmp_path / "mpc")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(os.path.dirname(__file__), "mp_channel_worker.py"), ",".join(map(str, nproc_dir)),
           ",".join(map(str, dims)), str(nsteps), stretching, repr(beta), "fused" if fused else "op", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    parts = [dict(np.load(out + f".{k}.npz")) for k in range(nproc)]
    g = {}
    for name in ("u", "v", "w"):
        full = np.zeros((dims[2], dims[1], dims[0]))
        for p in parts:
            ox, oy, oz = (int(v) for v in p["offset"])
            a = p[name]
            full[oz:oz + a.shape[0], oy:oy + a.shape[1], ox:ox + a.shape[2]] = a
        g[name] = full
    g["row"] = parts[0]["row"]
    g["halo_launches"] = int(parts[0]["halo_launches"][0])
    g["n_interleaved"] = int(parts[0]["n_interleaved"][0])
    return g


@pytest.mark.parametrize("nproc_dir,dims,stretching,beta,fused,halo", [
    # (>= 48 planes per rank: the DistD2 truncation dist_sa(n_local) is then below round-off, src/tdsops.f90:196-201)
    ((1, 1, 2), (24, 33, 96), "top-bottom", 0.259065151, False, False),   # two-sweep DistD2 along z, op-granular
    ((1, 1, 2), (24, 33, 96), "uniform", 1.0, True, False),
    ((1, 1, 3), (18, 17, 144), "top-bottom", 0.259065151, True, False),   # 10 modes over 3 ranks: a padded column
    ((1, 1, 2), (32, 17, 512), "top-bottom", 0.259065151, True, True),    # 256 planes per rank: single-pass HALO kernels
    ((1, 1, 4), (16, 17, 1024), "top-bottom", 0.259065151, True, True),
])
def test_channel_on_z_slabs_equals_single_rank_and_oracle(nproc_dir, dims, stretching, beta, fused, halo, tmp_path):
    """two channel steps (define_BC with the bulk-velocity all-reduce, transeq + rotation forcing, RK3, wall values,
    pressure correction with the slab 010 solve) on N ranks that share the GPU"""
    single = _channel_steps(dims, stretching, beta, fused, 2)  # (asserts HIP single rank == oracle, 1e-10)
    g = _run_channel_ranks(nproc_dir, dims, 2, stretching, beta, fused, tmp_path)
    s = single.solver
    for f, nm in ((s.u, "u"), (s.v, "v"), (s.w, "w")):
        ref = s.backend.get_field_data(f)
        assert np.max(np.abs(g[nm] - ref)) < 1e-11 * max(np.max(np.abs(ref)), 1.0), nm
    _, ens, dmax, _ = single.postprocess(2, 0.01)
    assert abs(g["row"][1] - ens) < 1e-11 * abs(ens)
    assert abs(g["row"][2] - dmax) < 1e-6 * dmax + 1e-12
    if halo:
        assert g["halo_launches"] > 0  # the single-pass kernels of the decomposed z direction engaged
        assert g["n_interleaved"] == 6  # and their z pairs carried the 010 solver's row interleave (2 steps x 3)
    else:
        assert g["n_interleaved"] == 0


def test_channel_two_slabs_at_the_bench_pencil_lengths(tmp_path):
    """1024-point x pencils, 257 stretched wall-normal vertices, 2 x 256 planes: BASELINE configs[4]'s kernels (K3w
    along x, K3g along y, the HALO tile kernels along the decomposed z, the slab 010 solve with 513 x modes over two
    ranks) in one step on two ranks against the single-rank HIP run and the oracle"""
    dims = (1024, 257, 512)
    single = _channel_steps(dims, "top-bottom", 0.259065151, True, 1)
    # (single rank at this size: the z pairs carry the solver's row interleave -- and, round 6, its z transforms:
    #  the z-first form of the 010 solve, csrc/zfirst.hip)
    assert single.solver.n_zfirst == 3 or single.solver.n_interleaved > 0
    g = _run_channel_ranks((1, 1, 2), dims, 1, "top-bottom", 0.259065151, True, tmp_path, timeout=1500)
    s = single.solver
    for f, nm in ((s.u, "u"), (s.v, "v"), (s.w, "w")):
        ref = s.backend.get_field_data(f)
        assert np.max(np.abs(g[nm] - ref)) < 1e-11 * max(np.max(np.abs(ref)), 1.0), nm
    assert g["halo_launches"] > 0 and g["n_interleaved"] == 3
