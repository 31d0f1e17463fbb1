#!/bin/bash
# LDS counters of the training forward with and without the in-kernel record count (COUNT template flag): tools/pmc_fwd_lds.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in plan noplan; do
  [ $mode = noplan ] && export SCANERF_NO_FORWARD_PLAN=1 || unset SCANERF_NO_FORWARD_PLAN
  timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $out/$mode -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-side-legs --arith-side-off --steps 4 --warmup 1 > $out/$mode.log 2>&1 || echo "pass failed: $mode"
done
python3 - <<EOF
import csv,glob,collections
for mode in ("plan","noplan"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$out/"+mode+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render_fwd" in r["Kernel_Name"] or "k_bin_count" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].replace("void (anonymous namespace)::","")[:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(mode, k, "%.4g"%(sum(v)/len(v)), len(v))
EOF
