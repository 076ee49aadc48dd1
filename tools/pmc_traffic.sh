#!/bin/bash
# HBM-side traffic of the fused MᵀM kernel: FETCH_SIZE and WRITE_SIZE in two separate rocprofv3 --pmc passes
# (no trace domains other than --kernel-trace).  usage: tools/pmc_traffic.sh [batch] [workload] [tag]   -> gpurun_out/pmc_traffic_fdm_mtm_b<batch>[_<tag>].json
nb=${1:-16}; wl=${2:-holstein_honeycomb_L16_Ltau128}; tag=${3:-}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$ctr
  rm -rf $out
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 tools/matvec_only.py $nb 20 $wl > gpurun_out/pmc_$ctr.log 2>&1 || exit 1
done
python3 - "$nb" "$wl" "$tag" <<'PY'
import csv, glob, json, sys
nb = int(sys.argv[1]); wl = sys.argv[2]; tag = sys.argv[3]
import re
alg = float(re.search(r'algorithmic_bytes (\d+)', open('gpurun_out/pmc_WRITE_SIZE.log').read()).group(1))
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "fdm_" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    vals = [float(r["Counter_Value"]) for r in rows]
    res[ctr] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals), "kernel": rows[0]["Kernel_Name"],
                "grid": rows[0].get("Grid_Size"), "wg": rows[0].get("Workgroup_Size"), "vgpr": rows[0].get("VGPR_Count"), "lds": rows[0].get("LDS_Block_Size")}
res.update({
    "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/matvec_only.py {nb} 20 {wl}  (two separate passes, tools/pmc_traffic.sh)",
    "systems_per_launch": nb, "workload": wl,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md §HBM) -> doubled; WRITE_SIZE exact",
    "traffic_bytes_per_launch": (2 * res["FETCH_SIZE"]["mean_KB"] + res["WRITE_SIZE"]["mean_KB"]) * 1024,
    "algorithmic_bytes_per_launch": alg,
})
json.dump(res, open(f"gpurun_out/pmc_traffic_fdm_mtm_b{nb}{('_' + tag) if tag else ''}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
