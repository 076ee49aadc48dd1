"""(SMOQY_TEAM_HMC=host for the step-by-step HMC of the members) team_threads_scan of bench.py alone (Python member threads, native member threads, one caller): python tools/team_scan.py [K,K,...] [sweeps] [TxK,...]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

counts = [int(k) for k in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32").split(",")]
args = argparse.Namespace(workload="holstein_honeycomb_L16_Ltau128", hmc=os.environ.get("SMOQY_TEAM_HMC", "device"), scan_sweeps=int(sys.argv[2]) if len(sys.argv) > 2 else 4, team_multi=sys.argv[3] if len(sys.argv) > 3 else "", no_prefetch=os.environ.get("SMOQY_PREFETCH", "1") == "0")
for p in bench.team_scan(args, counts, 0, 0):
    print(json.dumps(p), flush=True)
