#!/bin/bash
# usage: bash scripts/prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ + gpurun_out/prof_<tag>.json
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py "$@" --no-cpu-baseline --no-verify --no-default-options > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
cat gpurun_out/prof_$tag/*/*_kernel_stats.csv | cut -c1-60,200- | head -30
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-48s calls %4s avg %10.1f us total %9.2f ms" % (r["Name"][:48], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
