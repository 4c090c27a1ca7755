#!/bin/bash
# usage (GPU box): bash scripts/pmc_fx.sh <tag> [n_reads] [stride]
# SQ counters of the feature-extraction kernels under scripts/fx_bench.py (one --pmc pass, kernel trace only)
tag=$1; n=${2:-8000000}; stride=${3:-96}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fx_$tag
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_fx_$tag -- python3 scripts/fx_bench.py $n $stride > gpurun_out/pmc_fx_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_fx_%s/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in res.items():
    if "extract" not in k:
        continue
    print(k)
    for c, xs in sorted(v.items()):
        print("   %-24s n=%d avg=%.4g" % (c, len(xs), sum(xs) / len(xs)))
PY
