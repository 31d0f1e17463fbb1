#!/bin/bash
# PMC passes over the render-time decode kernels (bench.py --workload configs4-render).  Usage: tools/pmc_render.sh <outdir under gpurun_out>
# One small counter set per pass (rocprofv3 --pmc, no trace domains besides the kernel trace).
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload configs4-render --steps 2 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1 || echo "pass $i failed"
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/p$i 2>/dev/null | grep "k_pts_inference" >> $out/summary.txt || true
  find $out/p$i -name "*.csv" -size +2M -delete
done
cat $out/summary.txt
