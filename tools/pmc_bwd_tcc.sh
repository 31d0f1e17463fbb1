#!/bin/bash
# L2 / fabric-side counters of the backward kernel (separate --pmc passes): tools/pmc_bwd_tcc.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/tcc -- python3 $GRAFT_REPO_ROOT/tools/bwd_emit_only.py > $out/tcc.log 2>&1 || echo "pass failed: $set"
done
python3 - <<EOF
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/tcc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_bwd" in r["Kernel_Name"] or "k_bin" in r["Kernel_Name"] or "k_render_fwd" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].replace("void (anonymous namespace)::","")[:44],r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, "%.4g"%(sum(v)/len(v)), len(v))
EOF
