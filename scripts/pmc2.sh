#!/bin/bash
# usage: bash scripts/pmc2.sh <tag> "<C1 C2;C3;...>" <bench args...>
# one rocprofv3 --pmc pass per ';'-separated group (counters of a group are collected together)
tag=$1; shift
groups=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
IFS=';' read -ra G <<< "$groups"
p=0
for g in "${G[@]}"; do
  rm -rf gpurun_out/pmc_${tag}_p$p
  timeout -k 5 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_p$p -- python3 bench.py "$@" --no-cpu-baseline --no-verify > gpurun_out/pmc_${tag}_p$p.json 2> gpurun_out/pmc_${tag}_p$p.err
  p=$((p+1))
done
python3 - <<PY
import csv,glob,collections
tag="$tag"
res=collections.defaultdict(lambda: collections.defaultdict(list))
names=[]
for f in sorted(glob.glob("gpurun_out/pmc_%s_p*/*/*counter_collection.csv"%tag)):
    for r in csv.DictReader(open(f)):
        c=r["Counter_Name"]
        if c not in names: names.append(c)
        res[r["Kernel_Name"][:44]][c].append(float(r["Counter_Value"]))
for k,v in sorted(res.items(), key=lambda kv:-sum(kv[1].get(names[0],[0]))):
    print(k)
    for c in names:
        x=v.get(c,[])
        if x: print("    %-44s calls %4d  avg %18.1f" % (c,len(x),sum(x)/len(x)))
PY
