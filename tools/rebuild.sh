#!/bin/bash
# Build libscanerf_hip.so and record the listings as the ones about to be validated on the GPU (isa_audit --update + --check).
# EXP=1 tools/rebuild.sh builds the experiments library (environment switches of csrc/common.h); switching between the two
# modes cleans first, so that objects of the two builds never mix.
set -e
cd "$(dirname "$0")/.."
mode=${EXP:-0}
stamp=$(echo scanerf-*/lib)/.build_mode
if [ "$(cat $stamp 2>/dev/null)" != "$mode" ]; then make -C scanerf-*/csrc clean > /dev/null; fi
mkdir -p $(dirname $stamp); echo $mode > $stamp
SCANERF_SKIP_ISA_AUDIT=1 make -C scanerf-*/csrc -j8 EXP=$mode 2>&1 | grep -E "error|Error|warning" | grep -v "not a recognized feature" || true
python tools/isa_audit.py --update | tail -1
python tools/isa_audit.py --check | tail -1
