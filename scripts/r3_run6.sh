#!/bin/bash
# pass B in barcode order: parity + A/B (GPU box)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_barcode.py -m gpu -x -q > gpurun_out/r3_t9.log 2>&1; echo rc=$? >> gpurun_out/r3_t9.log; tail -5 gpurun_out/r3_t9.log
grep -q "rc=0" gpurun_out/r3_t9.log || exit 1
echo "== 3M list"; bash scripts/envab.sh "CRGPU_K2_SORTED=0 CRGPU_K2_SORTED=1" --whitelist 6794880 --steps 3 --warmup 1 --no-end-to-end
echo "== 737K list"; bash scripts/envab.sh "CRGPU_K2_SORTED=0 CRGPU_K2_SORTED=1" --steps 3 --warmup 1 --no-end-to-end
timeout -k 10 400 bash scripts/prof.sh r3k2s --whitelist 6794880 --steps 3 --warmup 1 --no-end-to-end > gpurun_out/r3_k2s_prof.log 2>&1; grep -i "correct\|compact\|region\|radix_scatter<unsigned int\|radix_hist\|scan_digits\|flag_corr" gpurun_out/r3_k2s_prof.log | grep calls
