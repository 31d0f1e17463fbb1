#!/bin/bash
# Upper bound of what run-merged records on the coarse levels could buy the backward (review item 3), WITHOUT building the merge:
# the `merge_emu` variant (tools/build_variant.py merge_emu render_bwd_t16="-DT16_MERGE_EMU") emits, on levels 0-3, only the record
# COUNT a merged emission would leave (every 4th / 3rd / 2nd / 2nd sample's records: -15 % of all records) -- no segmented scans, no
# single-entry records, wrong results: timing and fabric write requests only.  `base` = the same unit built the same way, unchanged.
# Usage (GPU box): tools/ab_merge_emu.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
D=$(ls -d scanerf-*/lib/debug)
: > $out/summary.txt
for rep in 1 2 3; do for tag in base merge_emu; do
  echo -n "$tag rep $rep: " >> $out/summary.txt
  SCANERF_LIB=$D/libscanerf_hip_$tag.so timeout -k 10 100 python tools/bwd_emit_only.py 2>/dev/null | tail -1 >> $out/summary.txt
done; done
cd /tmp && export TMPDIR=/tmp
for tag in base merge_emu; do
  SCANERF_LIB=$GRAFT_REPO_ROOT/$D/libscanerf_hip_$tag.so timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $out/$tag -- python3 $GRAFT_REPO_ROOT/tools/bwd_emit_only.py > $out/$tag.log 2>&1
  echo "== $tag" >> $out/summary.txt
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc $out/$tag 2>/dev/null | grep "k_render_bwd_t16" | awk '{print "  ", $1, $NF}' >> $out/summary.txt
  find $out/$tag -name "*.csv" -size +1M -delete
done
cat $out/summary.txt
