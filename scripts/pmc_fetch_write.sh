#!/bin/bash
# usage (GPU box): bash scripts/pmc_fetch_write.sh <tag> <bench args...>
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section) ->
# gpurun_out/pmc_<tag>_fetch_write.json with per-kernel averages and the derived HBM bytes per key of the
# dominant kernel (k_radix_scatter<unsigned long>): read bytes = 2 * FETCH_SIZE * 1024 on gfx950, write
# bytes = WRITE_SIZE * 1024.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py "$@" --no-cpu-baseline --no-verify > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.err
done
python3 - "$tag" "$*" <<'PY'
import csv, glob, collections, json, sys
tag, args = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, c)):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                res[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
bench = json.loads(open("gpurun_out/pmc_%s_FETCH_SIZE.json" % tag).read().strip().splitlines()[-1])
kern = []
for k, v in sorted(res.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0])) - sum(kv[1].get("WRITE_SIZE", [0]))):
    f, w = v.get("FETCH_SIZE", [0]), v.get("WRITE_SIZE", [0])
    kern.append({"kernel": k, "launches": len(f), "fetch_size_kb_avg": sum(f) / max(len(f), 1),
                 "write_size_kb_avg": sum(w) / max(len(w), 1)})
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py %s "
                 "(scripts/pmc_fetch_write.sh)" % args,
       "note": "units KB as reported; on gfx950 FETCH_SIZE counts half of a coalesced streaming read (MI355X_MICROARCH.md, "
               "HBM): hbm_read_bytes = 2*FETCH_SIZE*1024, hbm_write_bytes = WRITE_SIZE*1024",
       "kernels": kern, "derived": {}}
roof = bench.get("roofline") or {}
for k in kern:
    if k["kernel"].startswith("void k_radix_scatter<unsigned long"):
        per_launch = (2 * k["fetch_size_kb_avg"] + k["write_size_kb_avg"]) * 1024
        out["derived"] = {"kernel": k["kernel"], "hbm_bytes_per_launch": per_launch,
                          "elements_per_launch": roof.get("elements_per_launch"),
                          "k_radix_scatter_hbm_bytes_per_element_weighted":
                              per_launch / roof["elements_per_launch"] if roof.get("elements_per_launch") else None,
                          "algorithmic_bytes_per_element": 16}
        break
json.dump(out, open("gpurun_out/pmc_%s_fetch_write.json" % tag, "w"), indent=1)
print(json.dumps(out["derived"]))
PY
