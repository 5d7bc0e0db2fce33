"""gpurun_out/ (scratch) -> profiles/r04_* (tracked): what scratch/round4_artifacts.sh produced, besides what
tools_summarize.py copies (kernel stats / PMC traffic of the TGV bench)"""
import glob
import json
import os
import re
import shutil

R = "r04"
pairs = {f"bench_{R}.json": f"{R}_bench_512_fused.json", f"bench_{R}_lazy.json": f"{R}_bench_512_op_sequence_deferred.json",
         f"bench_{R}_opg.json": f"{R}_bench_512_opgranular.json", f"bench_{R}_256.json": f"{R}_bench_256_nopoisson.json",
         f"bench_{R}_channel.json": f"{R}_bench_channel_1024x257x512.json",
         f"bench_{R}_emulz.json": f"{R}_bench_512_emulated_z_slabs.json",
         f"bench_{R}_channel_emulz.json": f"{R}_bench_channel_emulated_z_slabs.json",
         f"bench_{R}_emuly.json": f"{R}_bench_512_emulated_y_slabs.json", f"bench_{R}_emuly_rccl.json": f"{R}_bench_512_emulated_y_slabs_rccl_to_self.json",
f"bench_{R}_ops.jsonl": f"{R}_bench_ops.jsonl",
         f"{R}_pmc_utilisation.txt": f"{R}_pmc_utilisation.txt"}
for src, dst in pairs.items():
    p = os.path.join("gpurun_out", src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join("profiles", dst))
        print("copied", dst)
    else:
        print("MISSING", src)
for tag, dst in (("chan", "channel"), ("emuly", "emulated_y_slabs")):
    fs = glob.glob(f"gpurun_out/prof_{tag}/*/*kernel_stats.csv")
    if fs:
        shutil.copy(max(fs, key=os.path.getmtime), f"profiles/{R}_kernel_stats_{dst}.csv")
        print("copied kernel stats", dst)
# the Fortran shim: its own "Averaged time per step", deferred execution vs call by call
shim = {"what": "fortran/_build/xcompact_hip = the reference's UNCHANGED solver.f90 / time_integrator.f90 / vector_calculus.f90 / "
                "case/tgv.f90 linked with fortran/m_hip_backend.f90; TGV 512^3, RK3, FFT Poisson, 20 steps with monitoring "
                "every 10 (fortran/tgv512.x3d); the program's own 'Averaged time per step' (includes the first step and "
                "the two monitoring outputs)", "runs": {}}
for mode in ("lazy", "eager"):
    p = f"gpurun_out/shim_tgv512_{mode}.log"
    if os.path.exists(p):
        m = re.search(r"Averaged time per step \(s\):\s*([0-9.eE+-]+)", open(p).read())
        if m:
            shim["runs"]["deferred execution (default)" if mode == "lazy" else "call by call (X3D_NO_LAZY=1)"] = {
                "ms_per_step": float(m.group(1)) * 1e3}
    c = f"gpurun_out/shim_tgv512_{mode}_monitoring.csv"
    if os.path.exists(c):
        shutil.copy(c, f"profiles/{R}_shim_512_monitoring_{'deferred' if mode == 'lazy' else 'call_by_call'}.csv")
json.dump(shim, open(f"profiles/{R}_shim_512.json", "w"), indent=1)
print(json.dumps(shim["runs"]))
