#!/bin/bash
# randomized parity hunts of the round's new paths (GPU box): bash scripts/r3_fuzz.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; ( "$@" ) > gpurun_out/fuzz_tmp.log 2>&1; rc=$?; grep -c "^ok" gpurun_out/fuzz_tmp.log; grep "FAIL\|failures\|Error" gpurun_out/fuzz_tmp.log | head -5; cat gpurun_out/fuzz_tmp.log >> gpurun_out/r3_fuzz_all.log; return $rc; }
: > gpurun_out/r3_fuzz_all.log
run env CRGPU_HOT_MIN_READS=1 timeout -k 10 280 python3 scripts/fuzz_barcode.py 120 101
run env CRGPU_HOT_MIN_READS=1 CRGPU_K2_SORTED=1 timeout -k 10 280 python3 scripts/fuzz_barcode.py 120 102
run env CRGPU_HOT_MIN_READS=1 CRGPU_K1_SPLIT=1 CRGPU_K2_SORTED=1 timeout -k 10 280 python3 scripts/fuzz_barcode.py 80 103
run env CRGPU_HOT_MIN_READS=1 CRGPU_K1_MODE=count timeout -k 10 280 python3 scripts/fuzz_barcode.py 80 104
run timeout -k 10 280 python3 scripts/fuzz_barcode.py 80 105
run timeout -k 10 280 python3 scripts/fuzz_parity.py 40 106 120000
run timeout -k 10 200 python3 scripts/fuzz_features.py 40 107
