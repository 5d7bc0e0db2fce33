#!/bin/bash
# Everything under profiles/r04_* that comes from the CURRENT tree, in one gpurun session:
#   gpurun -- 'bash scratch/round4_artifacts.sh'   then here: python tools_summarize.py r04 r04 r04 && python scratch/collect_r04.py
rnd=r04
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# kernel-trace stats + PMC traffic of the bench command (TGV 512^3, fused driver)
bash tools_prof.sh ${rnd} | grep -E "calls|total" | head -24
bash tools_pmc.sh ${rnd} > gpurun_out/pmc_${rnd}.txt
# bench lines
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${rnd}.json 2> gpurun_out/bench_${rnd}.err; tail -c 300 gpurun_out/bench_${rnd}.json
python bench.py --steps 10 --warmup 2 --lazy --no-cpu-baseline > gpurun_out/bench_${rnd}_lazy.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --op-granular --no-cpu-baseline > gpurun_out/bench_${rnd}_opg.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --n 256 --no-poisson --no-cpu-baseline > gpurun_out/bench_${rnd}_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel.json 2>/dev/null
X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emuly.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emulz.json 2>/dev/null
X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab X3D_COMM_SELF_VIA_NCCL=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${rnd}_emuly_rccl.json 2>/dev/null
X3D_EMULATE_DECOMP=z X3D_FORCE_PENCIL_FFT=slab python bench.py --steps 5 --warmup 2 --case channel --no-cpu-baseline > gpurun_out/bench_${rnd}_channel_emulz.json 2>/dev/null
python bench_ops.py > gpurun_out/bench_${rnd}_ops.jsonl 2>/dev/null
for f in lazy opg 256 channel emuly emulz emuly_rccl channel_emulz; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${rnd}_$f.json').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'])"; done
# the unchanged reference solver through the Fortran shim on one rank: deferred execution vs call by call
bash scratch/shim_run.sh fortran/tgv512.x3d tgv512
# kernel-trace stats of the channel bench and of the emulated y-slab path
for tag in chan emuly; do rm -rf gpurun_out/prof_$tag; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chan -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --case channel > gpurun_out/prof_chan.log 2>&1
export X3D_EMULATE_DECOMP=y X3D_FORCE_PENCIL_FFT=yslab
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emuly -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_emuly.log 2>&1
unset X3D_EMULATE_DECOMP X3D_FORCE_PENCIL_FFT
# utilisation counters of the dominant kernel (separate passes; counters only with --kernel-trace)
for grp in "LdsUtil VALUBusy" "LdsBankConflict MemUnitBusy" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf gpurun_out/pmc_util_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_util_$tag -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_util_$tag.log 2>&1
done
python3 - <<'PY' > gpurun_out/${rnd}_pmc_utilisation.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_util_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        if "ytile" in k or "xscan" in k or "onchip" in k or "fft512" in k or "c2c512" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 3) for c, v in sorted(d.items())})
PY
cat gpurun_out/${rnd}_pmc_utilisation.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
