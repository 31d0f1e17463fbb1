#!/bin/bash
# SQ counters of the backward kernel in the bench step (k_render_bwd_t16).  Usage (GPU box): tools/pmc_bwd.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1 || echo "pass $i failed"
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/p$i 2>/dev/null | grep "k_render_bwd_t16<0, true>\|k_bin_accumulate<512\|k_render_fwd_h3<0, true>" >> $out/summary.txt || true
  find $out/p$i -name "*.csv" -size +1M -delete
done
cat $out/summary.txt
