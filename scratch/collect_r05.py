"""gpurun_out/ (scratch) -> profiles/r05_* (tracked): what scratch/round5_artifacts.sh produced, besides what
tools_summarize.py copies (kernel stats / PMC traffic of the TGV bench)"""
import glob
import os
import shutil

R = "r05"
pairs = {f"bench_{R}.json": f"{R}_bench_512_fused.json", f"bench_{R}_lazy.json": f"{R}_bench_512_op_sequence_deferred.json",
         f"bench_{R}_fp32.json": f"{R}_bench_512_fp32.json", f"bench_{R}_channel.json": f"{R}_bench_channel_1024x257x512.json",
         f"bench_{R}_share4_dryrun.json": f"{R}_bench_4_ranks_shared_gpu_dryrun.json", f"bench_{R}_ops.jsonl": f"{R}_bench_ops.jsonl",
         f"bench_{R}_virtual2.json": f"{R}_bench_emulated_2_ranks.json", f"bench_{R}_virtual4.json": f"{R}_bench_emulated_4_ranks.json",
         f"bench_{R}_virtual8.json": f"{R}_bench_emulated_8_ranks.json"}
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
fs = glob.glob("gpurun_out/prof_v8/*/*kernel_stats.csv")
if fs:
    shutil.copy(max(fs, key=os.path.getmtime), f"profiles/{R}_kernel_stats_emulated_8_ranks.csv")
    print("copied kernel stats emulated 8 ranks")
