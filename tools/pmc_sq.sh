#!/bin/bash
# SQ counters per kernel over one 16-walker sweep: tools/pmc_sq.sh "<counter list>" <tag>
ctrs=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$tag
rm -rf $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o run -- python3 tools/one_stream.py ${PMC_WALKERS:-16} ${PMC_WORKLOAD:-} > gpurun_out/pmcsq_$tag.log 2>&1 || exit 1
python3 - "$out" > gpurun_out/pmc_sq_$tag.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if max(len(v) for v in d.values()) < 500: continue
    print(k[:80])
    for c, v in sorted(d.items()):
        v.sort(); print(f"   {c:28s} median {v[len(v)//2]:14.0f}")
PY
rm -rf $out
cat gpurun_out/pmc_sq_$tag.txt
