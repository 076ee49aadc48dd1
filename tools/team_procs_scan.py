"""(SMOQY_TEAM_HMC=host for the step-by-step HMC of the members) team_procs_scan of bench.py alone (member PROCESSES joined to served teams): python tools/team_procs_scan.py [K|TxK,...] [sweeps]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

points = (sys.argv[1] if len(sys.argv) > 1 else "16,32,4x32").split(",")
args = argparse.Namespace(workload="holstein_honeycomb_L16_Ltau128", hmc=os.environ.get("SMOQY_TEAM_HMC", "device"), scan_sweeps=int(sys.argv[2]) if len(sys.argv) > 2 else 4, no_prefetch=os.environ.get("SMOQY_PREFETCH", "1") == "0")
for p in bench.team_procs_scan(args, points, 0, 0):
    print(json.dumps(p), flush=True)
