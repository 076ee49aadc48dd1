"""K "MPI ranks" sharing one GPU through a served walker team ("Several walkers on one GPU", (c)).

This process plays the serving rank: it owns the batched handle (K walkers) and publishes the team in shared memory.  The K member ranks are
separate PROCESSES running examples/team_member_demo.c — plain C, no GPU access, no Python — each driving its own walker with the per-walker
update sequence of the reference tutorial.

    python examples/walker_team_ranks.py [workload] [K] [sweeps]
"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smoqyelphqmc_amd import _lib as L  # noqa: E402
from smoqyelphqmc_amd.walkers import WalkerTeam  # noqa: E402


def build_member(out_dir):
    exe = os.path.join(out_dir, "team_member_demo")
    lib_dir = os.path.dirname(L.LIB_PATH)
    subprocess.run([shutil.which("gcc") or "gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "team_member_demo.c"),
                    "-L" + lib_dir, "-lsmoqy_member", "-lm", "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    return exe


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "holstein_honeycomb_L16_Ltau128"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        exe = build_member(tmp)
        team = WalkerTeam(workload, K, device_efa=True)          # the serving rank: handle with nwalkers = K, couplings, bare model, EFA tables
        info = team.serve(f"/smoqy-demo-{os.getpid()}")
        t0 = time.perf_counter()
        ranks = [subprocess.Popen([exe, info["name"], str(w), str(sweeps), str(info["Nt"]), repr(info["tol"])], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")) for w in range(K)]
        failed = 0
        for p in ranks:
            out, err = p.communicate(timeout=900)
            failed += p.returncode != 0
            print(out.strip() if p.returncode == 0 else f"rank failed ({p.returncode}): {err.strip()[-300:]}")
        span = time.perf_counter() - t0
        team.close()
    print(f"{K} ranks x {sweeps} sweeps in {span:.2f} s (start-up included): {K * sweeps / span:.1f} sweeps/s on one GPU")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
