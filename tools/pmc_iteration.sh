#!/bin/bash
# HBM-side traffic of every kernel of the CG iteration at 16 systems per launch: FETCH_SIZE and WRITE_SIZE in two separate
# rocprofv3 --pmc passes over one sweep of a 16-walker batch.  -> gpurun_out/pmc_iteration.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmci_$ctr
  rm -rf $out
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 tools/one_stream.py ${PMC_WALKERS:-16} ${PMC_WORKLOAD:-} > gpurun_out/pmci_$ctr.log 2>&1 || exit 1
done
python3 - "${PMC_WORKLOAD:-holstein_honeycomb_L16_Ltau128}" "${PMC_WALKERS:-16}" "${PMC_TAG:-}" <<'PY'
import csv, glob, json, collections, sys
wl, nwk, tag = sys.argv[1], sys.argv[2], sys.argv[3]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmci_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if len(v) >= 500:  # the four kernels of the iteration
            v.sort()
            res[k][ctr] = {"launches": len(v), "median_KB": v[len(v) // 2], "mean_KB": sum(v) / len(v)}
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[k] = dict(d, traffic_MB=(2 * d["FETCH_SIZE"]["median_KB"] + d["WRITE_SIZE"]["median_KB"]) / 1024)
out["_workload"] = wl
out["_note"] = f"median per launch over one sweep of {nwk} walkers ({wl}); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); early-exit launches after convergence pull the mean down, hence the median"
json.dump(out, open(f"gpurun_out/pmc_iteration{('_' + tag) if tag else ''}.json", "w"), indent=1)
for k, d in out.items():
    if not k.startswith("_"):
        print(f"{k[:70]:70s} fetch {d['FETCH_SIZE']['median_KB']/1024:7.1f} MB (x2)  write {d['WRITE_SIZE']['median_KB']/1024:7.1f} MB  traffic {d['traffic_MB']:7.1f} MB")
PY
rm -rf gpurun_out/pmci_FETCH_SIZE gpurun_out/pmci_WRITE_SIZE
