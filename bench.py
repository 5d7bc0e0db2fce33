#!/usr/bin/env python3
"""Headline benchmark: Taylor-Green vortex, full fractional step (transeq ->
RK/AB substep -> pressure correction with the FFT Poisson solve), FP64, on N
MI355X GPUs of one node.  Metric (BASELINE.json): DoF*steps/s, whole job.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full time step (3 substeps for RK3) of the 512^3-per-GPU TGV
(BASELINE.json configs[2]; weak scaling: the global grid grows with N).
Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (dominant kernel: the fused transport-equation component, timed with
HIP events inside the timed region) and `cpu_baseline` (the oracle restatement
of the reference's OpenMP path on the host cores, bounded sample, N=1 only).
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def decomposition(n_gpus, kind="slabs"):
    """x stays whole (src/poisson_fft.f90:131).
    "yslabs" (the TGV default since round 3) = [1, N, 1]: like z slabs ONE transpose pair per solve among all ranks, and z
    stays whole on every rank so that the z-first Poisson solve of the single-rank path applies (csrc/sfftz.hip;
    one process standing in for a rank: 49.8 ms per step against 52.6 on z slabs, scratch/yslab_emul.sh).
    "slabs" = [1, 1, N], the layout the reference's GPU backend needs too
    (src/backend/cuda/poisson_fft.f90:219): y stays local -- no y exchanges at all, and the Poisson solve needs
    ONE transpose pair, an all-to-all among ALL ranks in which every GPU drives all of its N - 1 point-to-point
    xGMI links at once.  "pencils" = [1, 2, N / 2] (BASELINE configs[3]'s 2-D split, [1, 2, 4] on 8 GPUs): half
    the halo surface in z, but y halos on top and TWO transpose pairs per solve, the x-y one between pairs of ranks
    only (one link each): offered for comparison, not what the headline number uses."""
    if n_gpus not in (1, 2, 4, 8):
        raise SystemExit(f"--gpus {n_gpus}: supported 1, 2, 4, 8")
    if kind == "pencils" and n_gpus > 1:
        return (1, 2, n_gpus // 2)
    if kind == "yslabs":
        # y slabs [1, N, 1]: z stays whole on every rank, so the z-first Poisson solve of the single-rank path applies
        # (csrc/sfftz.hip: 6.5 instead of 10.5 passes over the spectrum, the same ONE all-to-all pair)
        return (1, n_gpus, 1)
    return (1, 1, n_gpus)


def auto_decomposition(case, n, lazy=False, op_granular=False):
    """--decomp auto: y slabs where the z-first Poisson solve applies to them (the fused TGV driver at 512^3 per GPU),
    z slabs otherwise (the channel's wall-normal y must stay whole; other sizes use the z-slab / pencil solvers)"""
    return "yslabs" if (case == "tgv" and n == 512 and not lazy and not op_granular) else "slabs"


def effective_cpus():
    """(physical cores, CPUs this process may actually use): the pool's boxes show 128 cores / 256 threads but run in a
    container with a CFS quota of 16 CPUs (/sys/fs/cgroup/cpu.max = "1600000 100000") -- more busy threads or MPI ranks than
    that are throttled (16 busy-polling ranks at the quota ran 45 x slower per DoF than 16 ranks of a short run)"""
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except ImportError:
        phys = os.cpu_count() or 1
    eff = phys
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            eff = max(1, min(phys, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return phys, eff


def cpu_baseline(n, steps, threads):
    """the port baseline in a FRESH child process whose environment carries OMP_NUM_THREADS / OMP_PLACES /
    OMP_PROC_BIND: libgomp reads them once, when it is loaded -- and in this process torch has loaded it long before
    (setting os.environ afterwards does not reach the oracle's OpenMP loops)."""
    import subprocess

    def child(nn, st, th):
        th, fw = th if isinstance(th, tuple) else (th, th)  # (OpenMP threads of the C loops, pocketfft workers)
        env = dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND="close", OMP_PLACES="cores",
                   X3D_ORACLE_FFT_WORKERS=str(fw))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-n", str(nn),
                            "--cpu-steps", str(st)], env=env, capture_output=True, text=True, timeout=1500)
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    try:
        probed = None
        if isinstance(threads, (list, tuple)):
            # which thread count suits this host: one 128^3 step each (about a second), the fastest runs the sample
            probed = {th: child(128, 1, th)["value"] for th in threads}
            threads = max(probed, key=probed.get)
        out = child(n, steps, threads)
        # `cores` = the CPUs the run could actually keep busy (the container's quota), `threads` = what it started
        out["threads"] = out.get("cores")
        if isinstance(out.get("cores"), int):
            out["cores"] = min(out["cores"], effective_cpus()[1])
        if isinstance(threads, tuple):
            out["fft_workers"] = threads[1]
        if probed:
            out["thread_counts_probed_at_128"] = {("%d omp, %d fft" % k if isinstance(k, tuple) else str(k)): v
                                                  for k, v in probed.items()}
        return out
    except Exception as e:  # noqa: BLE001 -- the GPU line must still be printed
        return {"value": None, "unit": "DoF*steps/s", "cores": threads if isinstance(threads, int) else None,
                "kind": "port", "sample": "failed: %s" % (str(e)[:200],)}


def cpu_baseline_child(n, steps):
    """oracle (CPU restatement of the reference OpenMP backend, SZ=16 layout: C + OpenMP kernels for every
    operator, reorder, sum and BLAS-1 pass, pocketfft with one worker per core for the DFT) timed on this host's
    cores: TGV n^3 RK3 full step incl. FFT Poisson.  n = 0: 512^3 (the GPU leg's size) when the host has the
    memory for it (~60 GiB), else 256^3.  `cores` = omp_get_max_threads() of the library that ran."""
    from oracle import x3d_oracle as orc
    lib = orc.lib()
    lib.orc_max_threads.restype = __import__("ctypes").c_int
    threads = int(lib.orc_max_threads())
    fw = os.environ.get("X3D_ORACLE_FFT_WORKERS")
    if n <= 0:
        try:
            import psutil
            n = 512 if psutil.virtual_memory().available > 256 * 2 ** 30 else 256
        except ImportError:
            n = 256
    twopi = 6.283185307179586
    mesh = orc.Mesh([n] * 3, [1, 1, 1], [twopi] * 3, ["periodic"] * 2, ["periodic"] * 2, ["periodic"] * 2)
    s = orc.Solver(mesh, Re=1600.0, dt=1e-3, time_intg="RK3", poisson="FFT")
    s.init_tgv()
    s.step()  # warm-up (first touch, library load)
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    dt = time.perf_counter() - t0
    return {"value": n ** 3 * steps / dt, "unit": "DoF*steps/s", "cores": threads, "kind": "port",
            "sample": f"TGV {n}^3 RK3 full fractional step (FFT Poisson), {steps} steps after 1 warm-up, "
                      f"oracle/x3d_oracle (C + OpenMP kernels on {threads} threads, pocketfft on {fw or threads} workers)",
            "seconds": dt, "n": n}


REF_NML = """&domain_settings
flow_case_name = 'tgv'
L_global = 6.283185307179586d0, 6.283185307179586d0, 6.283185307179586d0
dims_global = {n}, {n}, {n}
nproc_dir = {nproc_dir}
BC_x = 'periodic', 'periodic'
BC_y = 'periodic', 'periodic'
BC_z = 'periodic', 'periodic'
/End
&solver_params
Re = 1600d0
time_intg = 'RK3'
dt = 0.001d0
n_iters = {iters}
n_output = 1000
poisson_solver_type = 'CG'
der1st_scheme = 'compact6'
der2nd_scheme = 'compact6'
interpl_scheme = 'classic'
stagder_scheme = 'compact6'
/End
"""


def cpu_reference(n, iters, threads, nproc_dir=(1, 1, 1)):
    """the REAL reference -- its own xcompact (OpenMP backend), compiled from /root/reference's sources where they lie
    by oracle/ref/Makefile (flang -O3, AVX2; shipped prebuilt in oracle/_ref/fast, git-ignored) -- timed on this
    host: TGV n^3, RK3, poisson_solver_type = 'CG' (the reference's placeholder = NO pressure solve: its FFT Poisson
    needs 2decomp&FFT, which cannot be built here), its own "Averaged time per step" (first step excluded,
    src/case/base_case.f90:256-260, 339-342).  nproc_dir = [1, py, pz]: under mpirun on py * pz MPI ranks of `threads`
    OpenMP threads each -- the reference is an MPI code first (one rank per core group), a single OpenMP rank is not its
    CPU path at its best.  None when the binary did not travel (or there is no mpirun for a multi-rank shape)."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "fast", "xcompact")
    if not os.path.exists(exe):
        return None
    ranks = nproc_dir[0] * nproc_dir[1] * nproc_dir[2]
    cmd = [exe, "input.x3d"]
    if ranks > 1:
        mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
        if not os.path.exists(mpirun):
            return None
        cmd = [mpirun, "-n", str(ranks)] + cmd
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "input.x3d"), "w") as f:
            f.write(REF_NML.format(n=n, iters=iters, nproc_dir=", ".join(str(p) for p in nproc_dir)))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
        env.pop("OMP_PLACES", None)
        try:
            r = subprocess.run(cmd, cwd=wd, env=env, capture_output=True, text=True, timeout=120)
        except (OSError, subprocess.TimeoutExpired):
            return None
    m = re.search(r"Averaged time per step \(s\):\s*([0-9.eE+-]+)", r.stdout)
    if r.returncode != 0 or not m:
        return None
    t = float(m.group(1))
    return {"value": n ** 3 / t, "unit": "DoF*steps/s", "cores": min(threads * ranks, effective_cpus()[1]),
            "threads": threads * ranks, "kind": "reference", "mpi_ranks": ranks, "omp_threads_per_rank": threads, "nproc_dir": list(nproc_dir),
            "sample": f"the reference's xcompact (OpenMP backend, flang -O3 build of /root/reference's sources), TGV {n}^3 "
                      f"RK3, derivatives + RK only (poisson_solver_type='CG': no pressure solve), {iters} steps, its own "
                      f"average without the first step, {ranks} MPI rank(s) x {threads} OpenMP threads",
            "seconds_per_step": t, "n": n}


def live_traffic(kernel_substr, extra_args, timeout=120):
    """HBM bytes per launch of the kernels whose name contains `kernel_substr`, measured NOW: two child runs of this
    script (--pmc-child: warm-up + ONE step, nothing timed) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` --
    separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes, with its gfx950 correction (FETCH_SIZE counts
    64 B per 128-B request: x 2; both counters in KB).  The children are started from here as ordinary child processes
    (the program itself follows `--`).  None (+ the reason) when rocprofv3 is not there or a pass fails: the caller then
    falls back to the figure recorded in profiles/traffic.json."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as wd:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(wd, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"] + extra_args
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                                   timeout=timeout)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, "%s pass: %r" % (counter, e)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "%s pass failed (rc %s): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-200:])
            vals, vals_e = [], []
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") == counter and kernel_substr in name:
                        # (k_ytile_transeq3's seventh template flag: the launches that also do the RK stage, counted apart)
                        m = re.search(kernel_substr + r"<([^>]*)>", name)
                        targs = [a.strip() for a in m.group(1).split(",")] if m else []
                        (vals_e if len(targs) >= 7 and targs[6] == "true" else vals).append(float(row["Counter_Value"]))
            if not vals:
                return None, "%s pass: no launch of %s in the counter file" % (counter, kernel_substr)
            out[counter] = (sum(vals) / len(vals), len(vals), sum(vals_e) / len(vals_e) if vals_e else None, len(vals_e))
    fetch = out["FETCH_SIZE"][0] * 1024.0 * 2.0
    write = out["WRITE_SIZE"][0] * 1024.0
    with_stage = None
    if out["FETCH_SIZE"][2] is not None and out["WRITE_SIZE"][2] is not None:
        with_stage = {"bytes_per_launch": out["FETCH_SIZE"][2] * 2048.0 + out["WRITE_SIZE"][2] * 1024.0,
                      "fetch_bytes": out["FETCH_SIZE"][2] * 2048.0, "write_bytes": out["WRITE_SIZE"][2] * 1024.0,
                      "launches_counted": out["FETCH_SIZE"][3]}
    return {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
            "launches_counted": out["FETCH_SIZE"][1], "launches_with_rk_stage": with_stage,
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate child runs of this command (1 step), KB -> B, "
                   "FETCH x 2 (gfx950: 64 B counted per 128-B request, MI355X_MICROARCH.md)"}, None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here, as FRESH children of a
    parent that has not touched the GPU (never re-exec a process that has), one per device, rendezvous on
    127.0.0.1; rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    import tempfile
    procs = []
    out_f = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out_f if r == 0 else subprocess.DEVNULL, text=True))
    # watchdog: a rank that dies (RCCL init, out of memory) leaves the others waiting in a collective for ever --
    # as soon as one child has exited non-zero, or the limit is reached, the remaining ones are ended (exact PIDs)
    limit = float(os.environ.get("X3D_BENCH_TIMEOUT", "1500"))
    t0 = time.monotonic()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.monotonic() - t0 > limit:
            failed = ("rank %d exited with %s" % (bad[0], rcs[bad[0]])) if bad else "time limit of %.0f s" % limit
            time.sleep(5.0 if bad else 0.0)  # (let the others notice a closed connection by themselves first)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.2)
    out_f.seek(0)
    sys.stdout.write(out_f.read())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: %s; the other ranks were ended\n" % failed)
        return 1
    return max(abs(p.returncode) for p in procs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=512, help="grid points per GPU per direction")
    ap.add_argument("--time-intg", default="RK3")
    ap.add_argument("--no-poisson", action="store_true", help="BASELINE configs[1]: derivatives + RK only")
    ap.add_argument("--cpu-n", type=int, default=256,
                    help="CPU baseline grid (default 256^3: a bounded sample, ~5 s per step on the box's host; the rate per "
                         "DoF is the same at 512^3 -- 3.30e6 against 3.34e6, profiles/r04_bench_512_fused.json; 0: 512 if the "
                         "host has the memory, else 256)")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed steps of the port baseline after its warm-up step")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the port baseline (0: the usable CPUs and twice that are probed at 128^3, the fastest is used)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic with two rocprofv3 --pmc child runs (use profiles/traffic.json)")
    ap.add_argument("--case", default="tgv", choices=["tgv", "channel"],
                    help="channel: BASELINE configs[4]-style wall-bounded case (1 GPU), dims from --dims")
    ap.add_argument("--dims", default="1024,257,512", help="channel vertex dims nx,ny,nz")
    ap.add_argument("--decomp", default="auto", choices=["auto", "slabs", "yslabs", "pencils"],
                    help="N > 1: y slabs [1,N,1] (TGV default: z-first Poisson solve), z slabs [1,1,N] (the channel case: "
                         "y must stay whole) or the 2-D pencil split [1,2,N/2] of BASELINE configs[3]")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="ONE process stands in for rank 0 of a V-rank job (V = 2, 4, 8; --gpus 1): the V-rank kernels and "
                         "exchange pattern, every peer this rank itself through RCCL (X3D_COMM_FAKE_PEERS / _SELF_VIA_NCCL), "
                         "every exchange holding its stream for the time xGMI links of X3D_COMM_EMULATE_LINKS GB/s (default "
                         "61.4 = 80 %% of 76.8 per link and direction) would take -- a schedule timeline on a one-GPU box, "
                         "NOT a measurement of links; the JSON line says so")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1: do not time configs[1] (256^3, no Poisson) and the channel case after the headline")
    ap.add_argument("--no-validate", action="store_true",
                    help="N > 1: skip the one-step validation of the decomposition (and its fall-back chain)")
    ap.add_argument("--op-granular", action="store_true",
                    help="issue the reference's op sequence verbatim (reorders as copies, separate axpys)")
    ap.add_argument("--lazy", action="store_true",
                    help="the reference's op sequence through the library's deferred execution (csrc/lazy.hip): what the "
                         "unchanged solver.f90 gets through the Fortran shim; implies --op-granular")
    args = ap.parse_args()
    if args.lazy:
        args.op_granular = True
    if args.cpu_baseline_child:  # (never touches the GPU)
        print(json.dumps(cpu_baseline_child(args.cpu_n, args.cpu_steps)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.virtual_ranks:
        if args.gpus != 1 or args.virtual_ranks not in (2, 4, 8) or args.case != "tgv":
            raise SystemExit("--virtual-ranks V: V = 2, 4 or 8, with --gpus 1 and the TGV case")
        os.environ["X3D_COMM_FAKE_PEERS"] = "1"
        os.environ["X3D_COMM_SELF_VIA_NCCL"] = "1"
        os.environ.setdefault("X3D_COMM_EMULATE_LINKS", "61.4")
        args.no_cpu_baseline = args.no_other_configs = True

    # the CPU baseline uses the cores the host really gives this process (set before libgomp starts)
    phys, eff_cpus = effective_cpus()
    os.environ.setdefault("OMP_NUM_THREADS", str(eff_cpus))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # X3D_BENCH_SHARE_GPU=1 (dry runs of the multi-rank path on a one-GPU box): all ranks on cuda:0, exchanges
    # staged through gloo -- never what the driver measures
    share = os.environ.get("X3D_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective whose peer never arrives (a rank that raised while building or validating a layout) ends after
        # this limit instead of never: gloo raises in the waiting ranks (validated() below turns that into a failed
        # layout), RCCL's watchdog ends the process and spawn_ranks / the launcher end the others
        import datetime
        pg_to = datetime.timedelta(seconds=float(os.environ.get("X3D_BENCH_PG_TIMEOUT", "300")))
        if share:
            dist.init_process_group("gloo", timeout=pg_to)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_to)
    elif os.environ.get("X3D_COMM_SELF_VIA_NCCL") == "1":
        # one GPU, world size 1: with X3D_EMULATE_DECOMP the N > 1 code path then exchanges with itself THROUGH RCCL
        # (x3d2_amd/parallel.py) -- the RCCL calls, the communication stream and the wait semantics run for real
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))

    from x3d2_amd import make_tgv
    from x3d2_amd.parallel import Comm

    requested = args.decomp
    if args.decomp == "auto":
        args.decomp = auto_decomposition(args.case, args.n, args.lazy, args.op_granular)
    comm = Comm()
    tried = []

    def build(decomp):
        nproc_dir = decomposition(args.virtual_ranks or args.gpus, decomp)
        dims = tuple(args.n * p for p in nproc_dir)
        # (virtual ranks: the slabs are periodic replicas -- L grows with the rank count, the spacing stays -- so that
        #  rank 0 exchanging with itself IS the V-rank job's rank 0)
        twopi = 6.283185307179586
        case = make_tgv(dims, nproc_dir=nproc_dir, rank=rank, time_intg=args.time_intg,
                        poisson="CG" if args.no_poisson else "FFT", comm=comm, fused=not args.op_granular,
                        lazy=args.lazy, L=tuple(twopi * p for p in nproc_dir) if args.virtual_ranks else None)
        return case, nproc_dir, dims

    def validated(decomp):
        """N > 1, TGV: build the case on this decomposition, run ONE step and look at what a wrong exchange or transpose
        cannot get right -- the enstrophy of the Taylor-Green field after a step (3/8 at t = 0 whatever the grid) and
        max |div u| behind the pressure correction; every rank must agree.  None: this decomposition is not usable here
        (the reason is kept in config.decompositions_tried), the caller moves on to the next one."""
        rec = {"decomp": decomp}
        ok, built = 1.0, None
        try:
            if os.environ.get("X3D_BENCH_FAIL_DECOMP") == decomp:  # (test hook: exercise the fall-back chain)
                raise RuntimeError("X3D_BENCH_FAIL_DECOMP")
            built = build(decomp)
            case = built[0]
            case.step(1)
            s_ = case.solver
            row = case.monitoring.write_step(s_.dt, s_.u, s_.v, s_.w)
            rec["enstrophy"], rec["max_div_u"] = float(row[1]), float(row[2])
            good = abs(row[1] - 0.375) < 2e-3 and (args.no_poisson or row[2] < 1e-8)
            if not (good and row[1] == row[1]):
                ok = 0.0
                rec["error"] = "one step from the Taylor-Green field: enstrophy / max |div u| out of range"
        except Exception as e:  # noqa: BLE001 -- whatever it is, the next decomposition gets its chance
            ok = 0.0
            rec["error"] = repr(e)[:300]
        if comm.size > 1:
            ok = -comm.allreduce(-ok, "max")  # (min over the ranks)
        rec["ok"] = bool(ok)
        tried.append(rec)
        if not ok:
            built = None
            torch.cuda.empty_cache()
        return built

    if args.case == "tgv" and args.gpus > 1 and not args.no_validate and not args.virtual_ranks:
        # the N > 1 layouts have only ever run as several ranks on ONE GPU, as one process standing in for a rank, and
        # through RCCL to self: the first real multi-GPU run checks what it is about to time, and falls back
        chain = [args.decomp] + [d for d in ("yslabs", "slabs", "pencils") if d != args.decomp]
        if requested != "auto":
            chain = [args.decomp]
        built = None
        for d in chain:
            built = validated(d)
            if built is not None:
                args.decomp = d
                break
        if built is None:
            if rank == 0:
                print(json.dumps({"metric": "DoF*steps/s (whole node), TGV 512^3 per GPU, full fractional step",
                                  "value": None, "unit": "DoF*steps/s", "n_gpus": args.gpus, "error":
                                  "no decomposition passed its one-step validation", "decompositions_tried": tried}))
            raise SystemExit(3)
        case, nproc_dir, dims = built
    elif args.case == "tgv":
        case, nproc_dir, dims = build(args.decomp)
    if args.case == "channel":
        # BASELINE configs[4]; N > 1: z slabs [1, 1, N] of --dims vertices each (weak scaling: the span grows with N,
        # the wall-normal direction stays whole on every rank -- the reference itself has no multi-rank solver for
        # non-periodic y, src/poisson_fft.f90:177-180; here poisson_fft.HipSlabPoissonFFT010)
        from x3d2_amd import make_channel
        per = tuple(int(v) for v in args.dims.split(","))
        nproc_dir = (1, 1, args.gpus)
        dims = (per[0], per[1], per[2] * args.gpus)
        case = make_channel(dims, L=(4.0, 2.0, 2.0 * args.gpus), time_intg=args.time_intg,
                            poisson="CG" if args.no_poisson else "FFT", fused=not args.op_granular, rotation=True,
                            omega_rot=0.12, n_rotate=5000, comm=comm, nproc_dir=nproc_dir, rank=rank,
                            lazy=args.lazy)
    if args.pmc_child:  # (under rocprofv3 --pmc: a warm-up step and ONE counted step, nothing timed, nothing printed)
        case.step(1)
        case.step(2)
        case.solver.backend.sync()
        torch.cuda.synchronize()
        return

    def measure(case, nproc_dir, dims, decomp_name):
        """W warm-up steps, K timed steps between barriers + device syncs (max over the ranks), one more step with every
        kernel class and the exchanges timed; returns the JSON object of this case on this decomposition"""
        solver, backend = case.solver, case.solver.backend
        nstage = solver.time_integrator.nstage

        def sync_all():
            backend.sync()  # (x3d_device_sync: also runs what the deferred-execution layer still holds)
            torch.cuda.synchronize()
            comm.barrier()
            torch.cuda.synchronize()

        it = 0
        for _ in range(args.warmup):
            it += 1
            case.step(it)
        # HIP-event timers around the launches of the dominant kernel class (the roofline's kernel) during the timed
        # region; the other classes are timed in one extra step afterwards (all classes on cost 0.4 ms per step of
        # event records: same-box A/B 48.4 -> 48.0)
        backend.prof_enable(True)
        backend.prof_select(None if os.environ.get("X3D_BENCH_PROF_ALL") == "1" else ("transeq_fwd", "transeq_bwd"))
        backend.prof_reset()
        backend.rk_fused_passes = 0
        backend.rk_fused_launches = 0
        backend.rk_in_tile3_passes = 0
        backend.rk_in_tile3_launches = 0
        tq3_before = int(backend.lib.x3d_backend_counter(backend.h, 0))
        upd_before = int(backend.lib.x3d_backend_counter(backend.h, 1))
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            it += 1
            case.step(it, more=(k < args.steps - 1))  # (the last step completes its velocity correction itself)
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        dof_global = dims[0] * dims[1] * dims[2]
        dof_local = args.n ** 3 if args.case == "tgv" else dof_global // args.gpus
        value = dof_global * args.steps / elapsed
        # (bytes per DoF below are SURVEY 8(d)'s FP64 figures; the FP32 flavour of the library moves half of them)
        dofb = dof_local * (0.5 if os.environ.get("X3D_SINGLE_PREC") == "1" else 1.0)

        # ---- roofline of the dominant kernel class: one transport-equation
        # component = k_transeq_fwd + k_transeq_bwd (64 B/DoF for the three
        # components of one direction: u-component reads 1 + writes 1, the other two
        # read 2 + write 1 fields)
        n_f, ms_f = backend.prof_get("transeq_fwd")
        n_b, ms_b = backend.prof_get("transeq_bwd")
        per_dir_raw = {d: (backend.prof_get("transeq_fwd", d), backend.prof_get("transeq_bwd", d)) for d in (1, 2, 3)}
        n_tq3 = int(backend.lib.x3d_backend_counter(backend.h, 0)) - tq3_before
        n_upd = int(backend.lib.x3d_backend_counter(backend.h, 1)) - upd_before
        rk_fused_passes, rk_fused_launches = backend.rk_fused_passes, backend.rk_fused_launches
        # every kernel class, from ONE more step outside the timed region
        backend.prof_select(None)
        backend.prof_reset()
        comm.timed = comm.size > 1 or getattr(comm, "self_via_nccl", False)
        it += 1
        case.step(it)
        sync_all()
        exchanges = comm.timing_report() if comm.timed else None
        comm.timed = False
        # N > 1 (round 6): the same step once more with the exchanges ORDERED on the compute stream (no overlap) and once
        # as timed -- what the overlap hides, and how much exchange time stays exposed, from this very run
        overlap_report = None
        if comm.size > 1 or getattr(comm, "self_via_nccl", False):
            def one_step(ordered):
                nonlocal it
                keep = comm.overlap
                if ordered:
                    comm.overlap = False
                sync_all()
                t1 = time.perf_counter()
                it += 1
                case.step(it)
                sync_all()
                dt = time.perf_counter() - t1
                comm.overlap = keep
                if world > 1:
                    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                return dt * 1e3
            ms_over, ms_ord = one_step(False), one_step(True)
            xms = sum(v["ms"] for v in (exchanges or {}).values())
            overlap_report = {"one_step_overlapped_ms": ms_over, "one_step_ordered_ms": ms_ord,
                              "hidden_by_overlap_ms": ms_ord - ms_over,
                              "exchange_ms_one_step": xms,
                              "exposed_exchange_ms": max(0.0, xms - max(0.0, ms_ord - ms_over)),
                              "overlap_active": bool(comm.overlap) and not getattr(comm, "host_staged", False),
                              "note": "max over ranks; exposed = the exchanges' own HIP-event time minus what ordering them costs"}
        prof = {"note": "one extra step after the timed region, all classes timed"}
        for kind in backend.KINDS:
            n_l, ms = backend.prof_get(kind)
            prof[kind] = {"launches": n_l, "ms": ms}
        per_dir = {}
        for d, name in ((1, "x"), (2, "y"), (3, "z")):
            (nf, mf), (nb, mb) = per_dir_raw[d]
            if nf:
                per_dir[name] = {"ms_per_component": (mf + mb) / nf,
                                 "GB/s_at_64B_per_3_components": (64.0 / 3.0) * dofb / ((mf + mb) / nf * 1e-3) / 1e9,
                                 "GB/s_at_48B_per_3_components": 16.0 * dofb / ((mf + mb) / nf * 1e-3) / 1e9}
        # algorithmic bytes per launch.  SURVEY.md 8d's per-unit figures price every operation on its own: a transeq
        # component 24 B/DoF (16 when conv == u) = 64 B/DoF per direction, an accumulating tds_solve 24 B/DoF.  The fused
        # launches of this backend have a smaller compulsory traffic -- a three-in-one launch reads the advecting
        # velocity once (the table's "fully fused floor" of 48 B/DoF), and a transeq_x launch that also applies the
        # pending velocity correction reads 3 gradients and writes u, v, w on top (+48 B/DoF, the velocity itself
        # being an input it reads anyway).  Headline `achieved` / `frac`: that compulsory traffic of what a launch
        # does; `achieved_survey_per_unit`: the per-operation figures (larger: the fusion removed re-reads).
        comps3 = min(3 * n_tq3, n_f)
        rk_bytes = 8.0 * dofb * rk_fused_passes  # RK stage done by a transeq launch
        n_fused = rk_fused_launches
        total_floor = ((n_f - comps3) * (64.0 / 3.0) + comps3 * 16.0 + 48.0 * n_upd) * dofb + rk_bytes
        total_survey = (n_f * (64.0 / 3.0) + 72.0 * n_upd) * dofb + rk_bytes
        avg_ms = (ms_f + ms_b) / max(n_f, 1)
        transeq_bytes = (64.0 / 3.0) * dofb
        bytes_per_launch = total_floor / max(n_f, 1)  # per component
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if n_f else 0.0
        achieved_survey = total_survey / max(n_f, 1) / (avg_ms * 1e-3) / 1e9 if n_f else 0.0
        # HBM bytes per launch from the PMC counters cannot be collected inside a timed run (rocprofv3 --pmc passes,
        # scratch/round_artifacts.sh): the figure printed here is the one measured at the commit named next to it
        traffic = traffic_commit = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.case == "tgv" and args.gpus == 1 and not args.op_granular:
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("n") == args.n and tj.get("commit"):
                    # HBM bytes of ONE launch of the dominant kernel (k_ytile_transeq3), PMC FETCH_SIZE + WRITE_SIZE
                    traffic = tj.get("dominant_kernel_bytes_per_launch") or tj.get("transeq_component_bytes_per_launch")
                    traffic_commit = tj.get("commit")
            except Exception:
                traffic = traffic_commit = None
        # the dominant KERNEL by itself: the three-in-one tile kernel of the y and z directions (k_ytile_transeq3 at
        # periodic 256 / 512-row pencils, k_ygen_transeq3 for the channel's wall-normal pencils), event-timed inside the
        # timed region; bytes = SURVEY 8(d)'s unit figure for transeq_{y,z}: 64 B/DoF per launch of three components
        dominant = None
        # (channel: the wall-normal direction's k_ygen_transeq3 by itself -- its z launches are another kernel)
        dom_dirs = (2,) if args.case == "channel" else (2, 3)
        # (several ranks: the decomposed direction(s) only -- one HALO-form launch + its strip correction per sub-step; the
        #  local direction runs in two half launches beside the exchanges, which would count as two launches of full bytes)
        split_dirs = tuple(d + 1 for d in (1, 2) if nproc_dir[d] > 1)
        if args.case == "tgv" and split_dirs:
            dom_dirs = split_dirs
        yz = [(per_dir_raw[d][0][0], per_dir_raw[d][0][1] + per_dir_raw[d][1][1]) for d in dom_dirs]
        n_yz, ms_yz = sum(c for c, _ in yz), sum(m for _, m in yz)
        if n_yz and n_tq3:
            launches = n_yz / 3.0
            d_ms = ms_yz / launches
            d_bytes = 64.0 * dofb
            d_ach = d_bytes / (d_ms * 1e-3) / 1e9
            if args.case == "tgv" and not split_dirs:
                name = ("k_ytile_transeq3<%d,true,true,false,UNI,P12> (transeq_y and transeq_z, three components per launch; "
                        "round 6: the y launch in the circulant form, no lane tables)" % (args.n // 64))
            elif args.case == "tgv":
                name = ("k_ytile_transeq3<..,HALO> + k_transeq_halo_fix (transeq_%s: the decomposed direction in one pass + its "
                        "strip correction)" % "/".join("xyz"[d - 1] for d in split_dirs))
            else:
                name = "k_ygen_transeq3<5,..,DIRECT> (transeq_y on the 257-row wall-normal pencils, three components per launch)"
            dominant = {"name": name, "launches": launches, "avg_launch_ms": d_ms, "timed": "HIP events on the backend's "
                        "stream around every launch, inside the timed region",
                        "algorithmic_bytes_per_launch": d_bytes, "bytes_convention": "SURVEY 8(d): transeq_{y,z} 64 B/DoF x DoF",
                        "achieved": d_ach, "frac": d_ach / HBM_PEAK_GBS,
                        "frac_at_48B_fused_floor": 48.0 * dofb / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        # what the launch computes is rhs += transeq_<d>(u, v, w): the transeq unit AND the three
                        # sum_<d>intox units of SURVEY 8(d) in one pass -- compulsory traffic R u, v, w + R rhs x 3 + W rhs x 3
                        # = 72 B/DoF (the counters say 72.0: roofline.traffic); stated next to the 64 B figure, not instead
                        "frac_at_72B_accumulating_launch": 72.0 * dofb / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the same kernel's launches that also do the RK stage of u, v, w (template flag EPI, x3d_transeq_lincomb3):
            # timed apart (direction slot 0 of the timers); bytes = the transeq unit + the stage as the operation
            # vecadd / lincomb it replaces (8 B/DoF per field it reads or writes)
            n_e = n_f - sum(per_dir_raw[d][0][0] for d in (1, 2, 3))
            ms_e = ms_f + ms_b - sum(per_dir_raw[d][0][1] + per_dir_raw[d][1][1] for d in (1, 2, 3))
            if n_e and getattr(backend, "rk_in_tile3_launches", 0):
                e_l = n_e / 3.0
                e_ms = ms_e / e_l
                e_bytes = 64.0 * dofb + 8.0 * dofb * backend.rk_in_tile3_passes / backend.rk_in_tile3_launches
                dominant["with_rk_stage"] = {
                    "name": dominant["name"].split(" ")[0].replace("P12>", "P12,EPI>") + " (transeq_z + the RK stage of u, v, w)",
                    "launches": e_l, "avg_launch_ms": e_ms, "algorithmic_bytes_per_launch": e_bytes,
                    "bytes_convention": "64 B/DoF + 8 B/DoF per field the stage reads or writes",
                    "achieved": e_bytes / (e_ms * 1e-3) / 1e9, "frac": e_bytes / (e_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        class_average = {"what": "average over the x, y and z launches of the transport-equation class; x launches that also "
                                 "apply the pending velocity correction are credited its 48 B/DoF",
                         "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms}
        head_ach = dominant["achieved"] if dominant else achieved
        roofline = {"bound": "hbm",
                    "kernel": dominant["name"] if dominant else
                              "transeq component (one third of a k_xscan_transeq2x3 (x) / k_ytile_transeq3 (y, z) launch at "
                              "512^3; x launches that also apply the pending velocity correction include its bytes)",
                    "dominant_kernel": dominant, "class_average": class_average,
                    "three_in_one_launches": n_tq3, "launches_with_velocity_correction": n_upd,
                    "achieved_survey_per_unit": achieved_survey, "frac_survey_per_unit": achieved_survey / HBM_PEAK_GBS,
                    "achieved": head_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head_ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_measured_at_commit": traffic_commit,
                    "algorithmic_bytes_per_launch": dominant["algorithmic_bytes_per_launch"] if dominant else bytes_per_launch,
                    "survey_transeq_bytes_per_component": transeq_bytes, "rk_stage_fused_launches": n_fused,
                    "rk_stage_bytes_per_launch_avg": rk_bytes / max(n_f, 1),
                    "avg_launch_ms": dominant["avg_launch_ms"] if dominant else avg_ms, "launches": n_f, "per_direction": per_dir,
                    "share_of_step": (ms_f + ms_b) / (elapsed * 1e3),
                    # NOT an achieved bandwidth: the bytes the reference's OP-GRANULAR derivative pass of one sub-step would
                    # move (3 transeq + 6 reorders + 6 sum_intox = 54 field passes = 432 B/DoF) over the time
                    # this backend's transeq phase takes -- most of those passes are never made here (folded into the kernels)
                    "tdsops_pass_GBs_nominal_op_granular":
                        432.0 * dofb / ((ms_f + ms_b) / max(args.steps * nstage, 1) * 1e-3) / 1e9 if n_f else 0.0}
        # (b) the instantiation that takes the most time, and the whole transport phase on the bytes it must move
        if dominant and dominant.get("with_rk_stage"):
            e = dominant["with_rk_stage"]
            roofline["by_time_dominant"] = dict(e, share_of_step=e["launches"] * e["avg_launch_ms"] / (elapsed * 1e3),
                                                what="the instantiation of the dominant kernel with the largest share of the step "
                                                     "(z direction + RK stage); `frac` above stays the plain instantiation's")
        if n_f and n_tq3:
            lx, ly, lz = (per_dir_raw[d][0][0] / 3.0 for d in (1, 2, 3))
            le = (n_f - sum(per_dir_raw[d][0][0] for d in (1, 2, 3))) / 3.0
            # x: R u, v, w + W rhs x 3 (+ R g x 3 + W u, v, w where the velocity correction rides along); y, z: R u, v, w +
            # R rhs x 3 + W rhs x 3; + 8 B/DoF per field the RK stage inside a z launch reads or writes
            comp = (lx * 48.0 + n_upd * 48.0 + (ly + lz + le) * 72.0) * dofb + rk_bytes
            ph_ms = ms_f + ms_b
            roofline["transeq_phase"] = {
                "what": "all transport-equation launches (x + y + z, three components each) of the timed region on their "
                        "COMPULSORY bytes: x 48 B/DoF (+ 48 with the velocity correction), y and z 72 B/DoF (rhs is read, "
                        "added to and written), + the RK stage's fields where a z launch carries it",
                "launches": lx + ly + lz + le, "ms": ph_ms, "bytes": comp,
                "achieved": comp / (ph_ms * 1e-3) / 1e9, "frac": comp / (ph_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

        emu = None
        if args.virtual_ranks:
            emu = {"what": "EMULATION, not a multi-GPU measurement: ONE process on ONE GPU runs rank 0's kernels and exchange "
                           "pattern of a %d-rank job; every peer is the rank itself through RCCL; every exchange holds its "
                           "stream for the time the message would spend on xGMI links" % args.virtual_ranks,
                   "virtual_ranks": args.virtual_ranks,
                   "link_GBs_per_direction": float(os.environ["X3D_COMM_EMULATE_LINKS"]),
                   "link_model": "all-to-all among n peers: bytes / n / rate (n - 1 links at once); neighbour exchange: "
                                 "largest message / rate; 76.8 GB/s per link and direction by the public MI355X "
                                 "specification (7 links x 153.6 GB/s bidirectional), 61.4 = 80 % of it",
                   "value_if_every_rank_ran_like_this_one": value,
                   "slab_yparts": getattr(backend.poisson_fft, "yparts", None),
                   "slab_parts": getattr(backend.poisson_fft, "parts", None)}
        out = {
            "metric": ("DoF*steps/s (whole node), TGV 512^3 per GPU, full fractional step" if args.case == "tgv"
                       else "DoF*steps/s, channel (stretched y, 010 Poisson), full fractional step")
                      + (" -- EMULATED %d ranks on one GPU" % args.virtual_ranks if args.virtual_ranks else ""),
            "emulation": emu,
            "value": value, "unit": "DoF*steps/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if os.environ.get("X3D_SINGLE_PREC") == "1" else "f64",
            "data": "synthetic",
            # results differ from the reference's OpenMP backend by FMA contraction and re-associated scans only: every
            # -m gpu parity test holds 1e-12 relative per operator (1e-11 on traces / full steps, 1e-10 on the stretched
            # 010 Poisson solve); the north star asks for 1e-6 on the enstrophy trace
            "parity_tol": {"operators_rel": 1e-12, "full_step_rel": 1e-11, "poisson_010_rel": 1e-10,
                           "north_star_enstrophy_rel": 1e-6},
            "config": {"workload": (f"TGV {dims[0]}x{dims[1]}x{dims[2]} all-periodic, Re=1600, dt=1e-3, "
                                    if args.case == "tgv" else
                                    f"channel {dims[0]}x{dims[1]}x{dims[2]} verts, y Dirichlet + top-bottom "
                                    f"stretching, Re=4200, dt=5e-3, rotation forcing, ")
                                   + f"{args.time_intg} ({nstage} substeps/step), compact6/classic schemes, "
                                   + ("no pressure solve (configs[1])" if args.no_poisson
                                      else "rocFFT Poisson (configs[2])" if args.case == "tgv"
                                      else "rocFFT Poisson 010, pentadiagonal spectral solve"),
                       "per_gpu": f"{args.n}^3" if args.case == "tgv" else args.dims, "nproc_dir": list(nproc_dir),
                       "driver": ("op-granular calls recorded and fused inside the library (deferred execution)" if args.lazy
                                  else "op-granular" if args.op_granular else "fused"),
                       "parallelism": f"domain decomposition {nproc_dir[0]}x{nproc_dir[1]}x{nproc_dir[2]}",
                       # N > 1: did the overlapped exchange path pass its first-use check against the ordered path
                       # (x3d2_amd/parallel.py, Comm.self_check; None: not exercised, e.g. one rank or host-staged)
                       "overlap_self_check": getattr(comm, "self_check_result", None),
                       "overlap_self_check_error": getattr(comm, "self_check_error", None),
                   # (ms of two 2 ms spins: one on the candidate communication stream + one on the compute stream, both on
                   # the compute stream) per candidate: equal = the candidate shares the compute stream's hardware queue
                   # and was passed over (parallel.Comm._pick_stream)
                   "comm_stream_probe_ms": getattr(comm, "stream_probe", None),
                       # N > 1: which decomposition ran, and what its one-step validation (and any it replaced) showed
                       "decomposition": decomp_name if args.case == "tgv" else "slabs",
                       "decomposition_requested": requested,
                       "decompositions_tried": tried or None,
                       "transport": ("gloo, host staged (ranks share a GPU: a dry run)" if share else
                                     "RCCL %s" % ".".join(str(v) for v in torch.cuda.nccl.version())) if world > 1 or
                                    dist.is_initialized() else None,
                       # exchanges of ONE step after the timed region, HIP events on the stream they were posted from:
                       # sendrecv = halo rows + boundary values of the decomposed direction, alltoall = Poisson transposes
                       "exchanges_one_step": exchanges,
                       "overlap_report": overlap_report,
                       # 000 solve at 512^3 on one rank: transforms ordered z, x, y with the z transforms inside the
                       # neighbouring z operator pairs (csrc/zfirst.hip); counted pressure corrections of the fused driver
                       "poisson_z_first": (int(case.solver.n_zfirst) if not args.lazy
                                           else int(backend.lazy_stats().get("zfirst", 0)))},
            "dof_substeps_per_s": value * nstage,
            "roofline": roofline,
            "kernel_ms": prof,
        }
        if args.lazy:
            out["lazy_stats"] = backend.lazy_stats()
        return out

    out = measure(case, nproc_dir, dims, args.decomp)

    def brief(o):
        r = o["roofline"]
        return {"workload": o["config"]["workload"], "nproc_dir": o["config"]["nproc_dir"], "value": o["value"],
                "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
                "roofline": {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"],
                             "unit": r["unit"], "frac": r["frac"], "traffic": r.get("traffic"),
                             "avg_launch_ms": r["avg_launch_ms"],
                             "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]},
                "exchanges_one_step": o["config"].get("exchanges_one_step"),
                "overlap_report": o["config"].get("overlap_report"),
                "overlap_self_check": o["config"].get("overlap_self_check"),
                "comm_stream_probe_ms": o["config"].get("comm_stream_probe_ms")}

    if args.case == "tgv" and args.gpus > 1 and requested == "auto" and os.environ.get("X3D_BENCH_ONE_LAYOUT") != "1":
        # N > 1: BOTH layouts in one run -- the validated default (y slabs: one all-to-all pair per solve over all links)
        # and the north star's / BASELINE configs[3]'s 2-D pencil split [1, 2, N/2] (src/decomp/decomp_2decompfft.f90:42-48);
        # `value` is the better of the two, config.decomposition names it, both stay under `decompositions`
        import gc
        other = "pencils" if args.decomp != "pencils" else "yslabs"
        layouts = {args.decomp: brief(out)}
        if decomposition(args.gpus, other) == tuple(nproc_dir):
            layouts[other] = {"nproc_dir": list(nproc_dir), "same_layout_as": args.decomp,
                              "note": "at N = %d the two coincide" % args.gpus}
        else:
            case = built = None
            gc.collect()
            torch.cuda.empty_cache()
            built = validated(other)
            if built is None:
                layouts[other] = {"nproc_dir": list(decomposition(args.gpus, other)), "value": None,
                                  "error": tried[-1].get("error", "validation failed")}
            else:
                # (an exception while timing the second layout must not cost the first one's line; every rank takes the
                #  same branch: the verdict is all-reduced like the validation's)
                ok2, o2 = 1.0, None
                try:
                    o2 = measure(built[0], built[1], built[2], other)
                except Exception as e:  # noqa: BLE001
                    ok2 = 0.0
                    layouts[other] = {"nproc_dir": list(built[1]), "value": None, "error": repr(e)[:300]}
                if comm.size > 1:
                    ok2 = -comm.allreduce(-ok2, "max")
                if ok2 and o2 is not None:
                    layouts[other] = brief(o2)
                    if o2["value"] > out["value"]:
                        out = o2
                elif other not in layouts:
                    layouts[other] = {"nproc_dir": list(built[1]), "value": None, "error": "failed on another rank"}
            out["config"]["decompositions_tried"] = tried or None
        out["decompositions"] = layouts
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and args.case == "tgv":
        # the port is measured fastest on ONE socket's worth of threads or fewer (numpy-allocated blocks are
        # first-touched by one thread, so more threads only add remote-memory traffic: profiles/README.md)
        # (threads of the C / OpenMP loops, pocketfft workers): the CPUs the container's quota allows, and twice that
        # (the port was measured fastest on 32 threads of a 16-CPU quota: its numpy passes overlap with the C loops)
        eff = eff_cpus
        th = min(phys, args.cpu_threads) if args.cpu_threads > 0 else \
            sorted({(eff, eff), (min(2 * eff, phys), eff), (min(2 * eff, phys), min(2 * eff, phys))})
        out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.cpu_steps, th)
        out["cpu_baseline"]["host_cpus"] = {"physical_cores": phys, "usable_by_cgroup_quota": eff}
        # the real reference as the MPI code it is: R ranks of ONE OpenMP thread under mpirun (its OpenMP loops do not
        # scale -- 1 rank x 4 threads is slower than 1 x 1 on the build host, 4 ranks x 1 thread 3.3 x faster --, so
        # ranks carry the cores), R = half and a quarter of the usable CPUs (AT the quota busy-polling ranks are throttled
        # into each other's way), next to the single OpenMP rank rounds 1-3 timed; best shape reported, the others
        # kept; 3 steps each (its average excludes the first)
        def shape(r):
            py = 1
            while py * py < r:
                py *= 2
            return (1, py, r // py)
        shapes = [((1, 1, 1), min(phys, 2 * eff))]
        for r in sorted({max(2, eff // 2), max(2, eff // 4)}, reverse=True):
            if r & (r - 1) == 0:
                shapes.append((shape(r), 1))
        # (the multi-rank shapes at 128^3: on this pool they are 3 - 5 x slower per DoF than the single rank, two more 256^3
        #  runs of them cost the bench 1.5 minutes for a number that is not the best one)
        refs = [r for r in (cpu_reference(256 if nd == (1, 1, 1) else 128, 2 if nd == (1, 1, 1) else 3, t, nd) for nd, t in shapes)
                if r is not None]
        if refs:
            best = max(refs, key=lambda r: r["value"])
            best["other_shapes"] = [{"mpi_ranks": r["mpi_ranks"], "omp_threads_per_rank": r["omp_threads_per_rank"], "n": r["n"],
                                     "value": r["value"]} for r in refs if r is not best]
            out["cpu_baseline"]["reference_nopoisson"] = best
    if (rank == 0 and args.gpus == 1 and not args.virtual_ranks and not args.no_live_traffic and not args.op_granular
            and os.environ.get("X3D_BENCH_NO_LIVE_TRAFFIC") != "1"):
        # roofline.traffic measured in THIS run: the dominant kernel's HBM bytes per launch from the PMC counters of two
        # child runs of the same workload (one step each); falls back to the recorded figure if a pass fails
        import gc
        case = None
        gc.collect()
        torch.cuda.empty_cache()
        dom = "k_ygen_transeq3" if args.case == "channel" else "k_ytile_transeq3"
        extra = ["--case", args.case, "--n", str(args.n), "--dims", args.dims, "--time-intg", args.time_intg]
        if args.no_poisson:
            extra.append("--no-poisson")
        t0 = time.perf_counter()
        live, why = live_traffic(dom, extra)
        r = out["roofline"]
        if live is not None:
            r["traffic_recorded"] = {"bytes_per_launch": r.get("traffic"), "commit": r.get("traffic_measured_at_commit")}
            r["traffic"] = live["bytes_per_launch"]
            r["traffic_measured_at_commit"] = "this run"
            r["traffic_live"] = dict(live, kernel=dom, seconds=time.perf_counter() - t0)
            # the kernel's HBM rate on what the counters saw it move (not `frac`: that prices SURVEY's 64 B/DoF)
            r["frac_on_measured_traffic"] = live["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            r["traffic_live"] = {"failed": why, "seconds": time.perf_counter() - t0}
    if (rank == 0 and args.gpus == 1 and args.case == "tgv" and args.n == 512 and not args.no_poisson
            and not args.no_other_configs and not args.op_granular):
        # the other BASELINE configs one GPU holds, each in a fresh child process after this one has let go of its
        # blocks (5 timed + 2 warm-up steps): configs[1] = TGV 256^3 derivatives + RK only, and the one-GPU form of
        # configs[4] = channel 1024 x 257 x 512; each with the roofline of ITS dominant kernel
        import gc
        import subprocess
        case = None
        gc.collect()
        torch.cuda.empty_cache()
        others = {}
        # (round 6: configs[1] and the channel with the PMC traffic of THEIR dominant kernel measured in their child runs too;
        #  the headline workload under AB3, the scheme of the reference's examples/TGV/input.x3d:22 -- one sub-step per step,
        #  3 warm-up steps fill its history)
        for key, extra in (("configs[1] TGV 256^3, derivatives + RK only (no pressure solve)", ["--n", "256", "--no-poisson"]),
                           ("configs[4] on one GPU: channel 1024x257x512", ["--case", "channel", "--dims", "1024,257,512"]),
                           ("configs[2] with time_intg = 'AB3' (examples/TGV/input.x3d:22)",
                            ["--time-intg", "AB3", "--warmup", "3", "--no-live-traffic"])):
            t0 = time.perf_counter()
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "5", "--warmup", "2",
                                    "--no-cpu-baseline", "--no-other-configs"] + extra, capture_output=True,
                                   text=True, timeout=300)
                o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                others[key] = brief(o)
                for extra_key in ("by_time_dominant", "transeq_phase", "frac_on_measured_traffic"):
                    if extra_key in o["roofline"]:
                        others[key]["roofline"][extra_key] = o["roofline"][extra_key]
                others[key]["kernel_ms"] = {k: v for k, v in o["kernel_ms"].items() if isinstance(v, dict) and v["launches"]}
            except Exception as e:  # noqa: BLE001 -- the headline line must still be printed
                others[key] = {"value": None, "error": repr(e)[:300]}
            others[key]["seconds_in_all"] = time.perf_counter() - t0
        # ... and the reference's UNCHANGED solver.f90 through the Fortran shim (fortran/_build/xcompact_hip: built where the
        # reference tree is mounted, shipped prebuilt like the library), TGV 512^3, 20 steps: its own "Averaged time per step"
        shim = os.path.join(ROOT, "fortran", "_build", "xcompact_hip")
        if os.path.exists(shim):
            import re
            import tempfile
            t0 = time.perf_counter()
            key = "configs[2] through the reference's own solver.f90 (Fortran shim over the C ABI, deferred execution)"
            try:
                with tempfile.TemporaryDirectory() as wd:
                    r = subprocess.run([shim, os.path.join(ROOT, "fortran", "tgv512.x3d")], cwd=wd, capture_output=True,
                                       text=True, timeout=300, env=dict(os.environ, X3D_LAZY_REPORT="1"))
                m = re.search(r"Averaged time per step \(s\):\s*([0-9.eE+-]+)", r.stdout)
                st = dict(re.findall(r"(\w+)=(-?\d+)", r.stderr.split("x3d_lazy_report pid")[-1])) if "x3d_lazy_report" in r.stderr else {}
                others[key] = {"workload": "TGV 512x512x512, RK3, FFT Poisson, 20 steps, monitoring every 10 (fortran/tgv512.x3d); the "
                                           "program's own average, first step and outputs included",
                               "ms_per_step": float(m.group(1)) * 1e3 if m else None, "value": 512 ** 3 / float(m.group(1)) if m else None,
                               "unit": "DoF*steps/s", "returncode": r.returncode,
                               "lazy_declined": int(st.get("declined", -1)), "lazy_transeq_acc": int(st.get("transeq_acc", -1))}
            except Exception as e:  # noqa: BLE001
                others[key] = {"value": None, "error": repr(e)[:300]}
            others[key]["seconds_in_all"] = time.perf_counter() - t0
        out["other_configs"] = others
    if rank == 0:
        # the JSON line is the LAST thing on stdout: whatever libraries left in C stdio's buffer (RCCL's version banner
        # is printf'ed at init and would otherwise surface after Python's own output, at exit) goes out first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
