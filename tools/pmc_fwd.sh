#!/bin/bash
# PMC passes over the forward kernel alone (tools/bench_fwd.py, h3 arithmetic).  Usage: tools/pmc_fwd.sh <outdir under gpurun_out>
# One small counter set per pass (rocprofv3 --pmc, no trace domains besides the kernel trace).
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_INFLIGHT_MAX" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  ARITH=h3 B=${B:-65536} timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_fwd.py > $out/p$i.log 2>&1 || echo "pass $i failed"
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/p$i 2>/dev/null | grep "k_render_fwd_h3<0" >> $out/summary.txt || true
  find $out/p$i -name "*.csv" -size +2M -delete
done
cat $out/summary.txt
