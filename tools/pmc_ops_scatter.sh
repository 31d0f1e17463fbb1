#!/bin/bash
# Counters of the stand-alone binned scatter as the op-by-op route runs it (tools/ops_path_profile.py): what bounds k_bin_scatter?
# Usage (GPU box): tools/pmc_ops_scatter.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/summary.txt
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/ops_path_profile.py 65536 128 hip > $out/p$i.log 2>&1 || echo "pass $i failed"
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/p$i 2>/dev/null | grep -E "k_bin_scatter|k_bin_accumulate|k_bin_count" >> $out/summary.txt || true
  find $out/p$i -name "*.csv" -size +2M -delete
done
cat $out/summary.txt
