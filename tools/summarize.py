#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked):
kernel-trace stats of the bench command, PMC FETCH/WRITE per kernel with the
gfx950 correction (FETCH_SIZE counts 64 B per 128-B request -> x2; calibrated
here on k_tds_fwd, which reads exactly one 1 GiB field), and traffic.json that
bench.py reads for roofline.traffic.

    python tools/summarize.py <prof_tag> <pmc_tag> <round>
"""
import csv
import glob
import json
import shutil
import subprocess
import sys

prof_tag, pmc_tag, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
import os
# (gpurun merges every run of a tag into the same directory: take the newest)
stats = max(glob.glob(f"gpurun_out/prof_{prof_tag}/*/*kernel_stats.csv"), key=os.path.getmtime)
shutil.copy(stats, f"profiles/{rnd}_kernel_stats.csv")
pmc = json.load(open(f"gpurun_out/pmc_{pmc_tag}_summary.json"))
GiB = 1024.0 ** 3
rows = []
for k, v in pmc.items():
    fetch = v.get("FETCH_SIZE", 0.0) * 1024 * 2     # KB -> B, x2 (MI355X_MICROARCH.md, HBM section)
    write = v.get("WRITE_SIZE", 0.0) * 1024
    rows.append((k, v.get("n", 0), fetch, write))
rows.sort(key=lambda r: -(r[2] + r[3]))
with open(f"profiles/{rnd}_pmc_traffic.csv", "w") as f:
    f.write("kernel,launches,fetch_bytes_corrected,write_bytes,total_GiB\n")
    for k, n, fe, wr in rows:
        f.write(f"\"{k}\",{n},{fe:.0f},{wr:.0f},{(fe + wr) / GiB:.3f}\n")


# one transport-equation component = k_transeq_fwd + k_transeq_bwd (y, z two-sweep), k_xtranseq_fwd + _bwd
# (x, LDS-tiled) or ONE k_xscan_transeq / k_transeq_onchip launch (single pass)
heads = ("k_transeq_fwd", "k_xtranseq_fwd", "k_xscan_transeq", "k_ytile_transeq", "k_transeq_onchip")
# + the transposes of the y / z components that run through the scan kernel (viax.hip)
tot_bytes = sum(n * (fe + wr) for k, n, fe, wr in rows if "transeq" in k or "k_transpose" in k)
# (k_xscan_transeq2x3 / k_ytile_transeq3: three components per launch)
n_comp = sum(n * (3 if ("transeq2x3" in k or "transeq3" in k) else 1) for k, n, fe, wr in rows if any(h in k for h in heads))
comp = tot_bytes / n_comp if n_comp else 0.0
calib = [(fe, n) for k, n, fe, wr in rows if "k_lincomb" in k]
# X3D_ARTIFACT_COMMIT: the tree the gpurun session measured, when HEAD has moved on since
commit = os.environ.get("X3D_ARTIFACT_COMMIT") or \
    subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = not os.environ.get("X3D_ARTIFACT_COMMIT") and bool(
    subprocess.run(["git", "status", "--porcelain", "x3d2_amd", "bench.py"], capture_output=True, text=True).stdout.strip())
def _epi(k):
    # k_ytile_transeq3's seventh template flag (printed only when true): the launches that also do the RK stage
    a = k[k.index("<") + 1:k.rindex(">")].split(",") if "<" in k else []
    return len(a) >= 7 and a[6].strip() == "true"
dom = [(fe + wr) for k, n, fe, wr in rows if "k_ytile_transeq3" in k and not _epi(k)]
dom_e = [(fe + wr) for k, n, fe, wr in rows if "k_ytile_transeq3" in k and _epi(k)]
json.dump({"n": 512, "round": rnd, "commit": commit + ("+uncommitted" if dirty else ""),
           "transeq_component_bytes_per_launch": comp,
           "dominant_kernel": "k_ytile_transeq3<8,true,true,false>",
           "dominant_kernel_bytes_per_launch": dom[0] if dom else None,
           "dominant_kernel_with_rk_stage_bytes_per_launch": dom_e[0] if dom_e else None,
           "components_profiled": n_comp,
           "note": "HBM bytes per transport-equation component (all k_*transeq* + k_transpose64 + k_transpose_lincomb kernels / number of components; the latter also does the RK stage's linear combination) "
                   "from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per "
                   "MI355X_MICROARCH.md (calibrated in round 1 on k_tds_fwd: one 1 GiB field read = 0.508 GiB raw)"},
          open("profiles/traffic.json", "w"), indent=1)
print("component traffic GiB:", comp / GiB, "components:", n_comp)


# per-variant rows of k_xscan_tds_lin (one kernel name, three operand counts: the RK3 stages read 2, 2 and 4 fields
# besides what they write): the dispatches of the kernel trace, in launch order, fall into groups of three per stage
try:
    trace = max(glob.glob(f"gpurun_out/prof_{prof_tag}/*/*kernel_trace.csv"), key=os.path.getmtime)
    durs = []
    for row in csv.DictReader(open(trace)):
        if "k_xscan_tds_lin" in row["Kernel_Name"]:
            durs.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    durs = [d for _, d in sorted(durs)]
    with open(f"profiles/{rnd}_tds_lin_variants.csv", "w") as f:
        f.write("position_in_step,what,launches,avg_us,min_us,max_us\n")
        names = {0: "stage 1: u = u + c d (reads base, 1 term; writes y, du)", 1: "stage 2 (reads base, 1 term; writes y, du)",
                 2: "stage 3: u = olds1 + 3 terms (reads 4; writes y, du)"}
        for g in range(3):
            v = [d for i, d in enumerate(durs) if (i // 3) % 3 == g]
            if v:
                f.write(f"{g},\"{names[g]}\",{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n")
    print("k_xscan_tds_lin dispatches:", len(durs))
except Exception as e:  # noqa: BLE001
    print("no per-dispatch trace for the k_xscan_tds_lin variants:", e)
