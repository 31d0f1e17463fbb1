#!/bin/bash
# rocprofv3 passes of the default bench command (separate passes: kernel stats, FETCH_SIZE, WRITE_SIZE, SQ counters) and the
# JSON bench.py's `roofline` reads (profiles/rNN_pmc.json).  Usage (GPU box): tools/profile_bench.sh <outdir under gpurun_out> [bench args]
# (one rank only: bench.py refuses to launch ranks from a profiled process)
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-side-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B --steps 10 --warmup 2 "$@" > $out/stats.log 2>&1 || echo "stats pass failed"
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py stats $out/stats > $out/kernel_stats.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- $B --steps 4 --warmup 1 --arith-side-off "$@" > $out/p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/make_pmc_json.py $out > $out/pmc.json 2> $out/pmc_table.txt
find $out -name "*.csv" -size +1M -delete
cat $out/kernel_stats.txt | head -16; cat $out/pmc_table.txt
