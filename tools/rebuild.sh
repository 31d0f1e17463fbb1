#!/bin/bash
# Build libscanerf_hip.so and record the listings as the ones about to be validated on the GPU (isa_audit --update + --check).
set -e
cd "$(dirname "$0")/.."
SCANERF_SKIP_ISA_AUDIT=1 make -C scanerf-*/csrc -j8 EXP=${EXP:-0} 2>&1 | grep -E "error|Error|warning" | grep -v "not a recognized feature" || true
python tools/isa_audit.py --update | tail -1
python tools/isa_audit.py --check | tail -1
