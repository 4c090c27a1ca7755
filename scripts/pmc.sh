#!/bin/bash
# usage: bash scripts/pmc.sh <tag> <bench args...> : FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py "$@" --no-cpu-baseline --no-verify > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.err
done
python3 - <<PY
import csv,glob,collections
tag="$tag"
res=collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv"%(tag,c))
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name")==c:
                res[r["Kernel_Name"][:60]][c].append(float(r["Counter_Value"]))
print("%-62s %6s %14s %14s" % ("kernel","calls","FETCH_SIZE avg","WRITE_SIZE avg"))
for k,v in sorted(res.items(), key=lambda kv:-sum(kv[1].get("FETCH_SIZE",[0]))):
    f=v.get("FETCH_SIZE",[0]); w=v.get("WRITE_SIZE",[0])
    print("%-62s %6d %14.1f %14.1f" % (k,len(f),sum(f)/max(len(f),1),sum(w)/max(len(w),1)))
PY
