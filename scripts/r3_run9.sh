#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_barcode.py -m gpu -x -q > gpurun_out/r3_t12.log 2>&1; echo rc=$? >> gpurun_out/r3_t12.log; tail -5 gpurun_out/r3_t12.log
grep -q "rc=0" gpurun_out/r3_t12.log || exit 1
echo "== 737K"; bash scripts/envab.sh "CRGPU_K1_MODE=full CRGPU_K1_MODE=count" --steps 3 --warmup 1 --no-end-to-end
echo "== 3M"; bash scripts/envab.sh "CRGPU_K1_MODE=split CRGPU_K1_MODE=count" --whitelist 6794880 --steps 3 --warmup 1 --no-end-to-end
echo "== cfg2"; bash scripts/envab.sh "CRGPU_K1_MODE=full CRGPU_K1_MODE=count" --workload cfg2 --steps 20 --warmup 3
