#!/bin/bash
# usage (GPU box): bash scripts/prof_variants.sh "<variant names>" "<kernel name regex>" <bench args...>: rocprof kernel stats per variant
names=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in $names; do
  rm -rf gpurun_out/pv_$v
  CRGPU_LIB_PATH=$GRAFT_REPO_ROOT/cellranger_amd/variants/libcrgpu_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv_$v -- python3 bench.py "$@" --no-cpu-baseline --no-verify --no-end-to-end > /dev/null 2> gpurun_out/pv_$v.err
  python3 - "$v" "$pat" <<'PY'
import csv, glob, re, sys
v, pat = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/pv_%s/*/*_kernel_stats.csv" % v)[0]
for r in csv.DictReader(open(f)):
    if re.search(pat, r["Name"]):
        print("%-10s %-44s calls %4s avg %9.1f us total %8.2f ms" % (v, r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
