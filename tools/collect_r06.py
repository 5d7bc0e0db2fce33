"""gpurun_out/ (scratch) -> profiles/r06_* (tracked): what tools/round6_artifacts.sh produced, besides what
tools/summarize.py copies (kernel stats / PMC traffic of the TGV bench)"""
import csv
import glob
import os
import shutil

R = "r06"
pairs = {f"bench_{R}.json": f"{R}_bench_512_fused.json", f"bench_{R}_lazy.json": f"{R}_bench_512_op_sequence_deferred.json",
         f"bench_{R}_channel.json": f"{R}_bench_channel_1024x257x512.json",
         f"bench_{R}_share2_dryrun.json": f"{R}_bench_2_ranks_shared_gpu_dryrun.json",
         f"{R}_channel_ab.txt": f"{R}_channel_same_box_ab.txt", f"{R}_tgv_ab.txt": f"{R}_tgv_256_fp32_same_box_ab.txt"}
for src, dst in pairs.items():
    p = os.path.join("gpurun_out", src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join("profiles", dst))
        print("copied", dst)
    else:
        print("MISSING", src)
fs = glob.glob("gpurun_out/prof_chan/*/*kernel_stats.csv")
if fs:
    shutil.copy(max(fs, key=os.path.getmtime), f"profiles/{R}_kernel_stats_channel.csv")
    print("copied kernel stats channel")
# roctx: per range name, calls and total time (the marker trace itself is MBs of rows)
for tag, dst in (("mark_chan", f"{R}_marker_trace_channel_summary.csv"), ("mark_lazy", f"{R}_marker_trace_deferred_tgv_summary.csv")):
    fs = glob.glob(f"gpurun_out/{tag}/*/*marker_api_trace.csv") + glob.glob(f"gpurun_out/{tag}/*/*marker*trace.csv")
    if not fs:
        print("MISSING marker trace", tag)
        continue
    acc = {}
    with open(max(fs, key=os.path.getmtime)) as f:
        for row in csv.DictReader(f):
            name = row.get("Function") or row.get("Name") or row.get("Message") or "?"
            try:
                dt = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            except (KeyError, ValueError):
                continue
            a = acc.setdefault(name, [0, 0])
            a[0] += 1
            a[1] += dt
    with open(os.path.join("profiles", dst), "w") as f:
        f.write("range,calls,total_host_ns,avg_host_ns\n")
        for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.0f\n' % (k, n, t, t / max(n, 1)))
    print("wrote", dst, len(acc), "ranges")
    st = glob.glob(f"gpurun_out/{tag}/*/*marker_api_stats.csv") + glob.glob(f"gpurun_out/{tag}/*/*marker*stats.csv")
    if st:
        shutil.copy(max(st, key=os.path.getmtime), os.path.join("profiles", dst.replace("_summary", "_stats")))
