#!/usr/bin/env python3
"""bench.py — QMC sweeps/sec + FermionDetMatrix matvec GB/s vs the HBM roofline, fp64.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; every rank owns ``--walkers-per-gpu`` independent Monte Carlo walkers of
the synthetic Holstein-honeycomb workload (default L = 16, Ltau = 128 — the configuration
BASELINE.json's target is quoted on).  Walkers never exchange state (the reference's
one-MPI-rank-per-walker model), so there is no data-path collective: RCCL is only used for the
barrier and the max-over-ranks of the wall time.  A "step" is one synthetic sweep of every
walker: reflection + swap + HMC(Nt = 24) = 27 CG solves of MᵀM x = b with the KPM
preconditioner (smoqyelphqmc.jl_amd/walkers.py).  Inputs (phonon fields, lattice tables) are
generated before the timed region; random vectors are drawn on the host inside it, as the
reference does.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel = the fused MᵀM apply.  `achieved` / `frac`: ALGORITHMIC bytes per launch (2·(2S+F) per
                system, F once per walker) over the average duration of the launches sampled INSIDE the timed region by the device clock
                (first workgroup start -> last workgroup end, the interval rocprofv3 reports; the HIP event pairs around the same launches
                are reported next to it).  `traffic` / `frac_traffic`: bytes really moved beyond L2 (committed PMC passes) over the same
                duration.  `isolated`: the kernel alone on the GPU.  `copy_ceiling`: a device stream copy measured in this run.
                `batch_scan`: 1..128 systems per launch; `hbm_resident_point`: the 128-system launch (working set > Infinity Cache).
  one_stream    sweeps/s of 1, 8, 16 walkers on one stream (single_walker = BASELINE.json's literal configuration)
  procs_per_gpu_scan    the reference's own execution model (one walker per MPI rank, tutorials/holstein_honeycomb_mpi.jl:60-72) on ONE
                GPU: K fresh child processes, each with a single-walker handle, started together; aggregate sweeps/s per K.  Runs before
                this process touches the GPU.  K stops at 6: the GPU pool admits at most six processes on a card at once.
  threads_per_gpu_scan  the same with K host threads in one process, each owning a single-walker handle (its own HIP stream).
  team_threads_scan     K host threads, each driving ONE walker through a walker team (smoqy_team_*): per-walker control flow, batched launches.
  cpu_baseline  the CPU oracle (a single-threaded restatement of the reference algorithm, NOT the Julia reference, which cannot run
                here) timed on a bounded sample of the same workload, one walker per host core, all at once.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="holstein_honeycomb_L16_Ltau128")
    ap.add_argument("--walkers-per-gpu", type=int, default=0, help="0 = the workload's measured best shape (DEFAULT_SHAPES; 128 for the headline lattice)")
    ap.add_argument("--streams", type=int, default=0, help="concurrent lock-step batches per GPU (one HIP stream + host thread each); 0 = the workload's measured best shape")
    ap.add_argument("--solve-concurrency", type=int, default=4, help="at most this many batches inside the CG at once (0 = no limit)")
    ap.add_argument("--gate", choices=["library", "python"], default="library", help="where --solve-concurrency is enforced: inside the library around each CG loop, or in Python around whole calls")
    ap.add_argument("--cg-split", type=int, default=1, choices=[0, 1, 2, 3, 4],
                    help="two-part CG pipeline inside each handle of the TIMED batches (smoqy_cg_split): 1 = off (default here: six streams already overlap, and the roofline samples "
                         "full-batch MtM launches), 0 = the library's automatic choice, 2 = on.  The one-stream legs always use the library default (automatic).")
    ap.add_argument("--tfft-form", choices=["auto", "two-image", "in-place"], default="auto",
                    help="τ-FFT of the timed batches (smoqy_tfft_form): in-place has more workgroups per CU (+2.7 %% sweeps/s with six streams), two-image is faster alone; "
                         "auto = in-place when more than one stream shares the GPU, Ltau = 2^a 3^b 5^c and the lattice has at least 14400 space-time sites (round 4: "
                         "bond-SSH chain 122 -> 135, optical-SSH square 178 -> 197 sweeps/s; honeycomb L = 8 and 4 indifferent, profiles/r04_tfft_form_scan.txt).  The one-stream legs keep the library default (two-image).")
    ap.add_argument("--no-mtm-sampling", action="store_true", help="do not sample MtM launches inside the timed region (roofline falls back to the isolated leg)")
    ap.add_argument("--measure-nrv", type=int, default=0, help="add update_greens_estimator! + measure_GΔ0! with this many random vectors to every sweep (27 + Nrv solves)")
    ap.add_argument("--hmc", choices=["device", "host"], default="device",
                    help="device: the EFA leapfrog of the HMC trajectory runs on the GPU (x, p and the force never leave it; smoqy_hmc_trajectory_v); "
                         "host: round-1 form, a synthetic drift on the host with x uploaded and the force downloaded every step")
    ap.add_argument("--no-prefetch", action="store_true", help="draw every sweep's random numbers when they are needed instead of one sweep ahead on the host-thread pool (WalkerBatch(prefetch_randoms=...)); same numbers either way")
    ap.add_argument("--tau-chunk", type=int, default=0)
    ap.add_argument("--check-every", type=int, default=0)
    ap.add_argument("--matvec-reps", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)  # child process of cpu_baseline: time the oracle sample for this walker
    ap.add_argument("--cpu-tol", type=float, default=1e-10, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-nt", type=int, default=24, help=argparse.SUPPRESS)
    ap.add_argument("--rank-worker", type=int, default=-1, help=argparse.SUPPRESS)  # child process of procs_per_gpu_scan: one single-walker handle, walker id = this
    ap.add_argument("--scan-sweeps", type=int, default=6, help="sweeps each rank / thread of the procs_per_gpu / threads_per_gpu scans times")
    ap.add_argument("--proc-scan", default="1,2,4,6", help="process counts of procs_per_gpu_scan (the pool admits at most 6 GPU processes per card)")
    ap.add_argument("--thread-scan", default="1,2,4,8,16", help="thread counts of threads_per_gpu_scan")
    ap.add_argument("--team-scan", default="8,16,32", help="member counts of team_threads_scan")
    ap.add_argument("--member-worker", default="", help=argparse.SUPPRESS)  # child process of team_procs_scan: "<info json>|<w>|<sweeps>", never touches the GPU
    ap.add_argument("--team-procs", default="16,32,4x32", help="points of team_procs_scan: K member PROCESSES on one served team, or TxK (T teams)")
    ap.add_argument("--team-multi", default="4x32", help="TxK points of team_threads_scan: T teams of K native member threads side by side")
    ap.add_argument("--no-proc-scan", action="store_true", help="skip procs_per_gpu_scan, threads_per_gpu_scan and team_threads_scan")
    ap.add_argument("--roofline-only", action="store_true", help="run only the isolated roofline leg (for a rocprofv3 pass whose kernel average must match roofline.avg_launch_us)")
    ap.add_argument("--batch-scan", action="store_true", help="try tau chunks 1..4 at every point of the batch scan (the default scan uses the heuristic chunk)")
    ap.add_argument("--timed-only", action="store_true",
                    help="only warm-up + timed region: no isolated leg, batch scan, copy ceiling, one-stream or CPU legs (for a rocprofv3 pass whose per-kernel averages must agree with roofline.avg_launch_us)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N>1 control flow on a one-GPU box: every rank uses cuda:0 and the barrier / MAX reduction run over gloo")
    return ap.parse_args()


# walkers per GPU x streams per workload, from the scans committed as profiles/r0*_config_scan.txt (headline lattice, round 3) and
# profiles/r04_shape_scan.txt (the other lattices, round 4: their launches fill half a chip or less at 16 systems, so they want 64 per stream)
DEFAULT_SHAPES = {
    "holstein_honeycomb_L16_Ltau128": (128, 8),
    "holstein_honeycomb_L8_Ltau80": (256, 4),     # 3355 sweeps/s against 2034 at 128 x 8
    "holstein_honeycomb_L4_Ltau40": (256, 4),
    "ossh_square_L12_Ltau100": (256, 4),          # 178.0 against 119.9
    "ossh_square_L12_Ltau100_alpha0p2": (256, 4),
    "bssh_chain_L256_Ltau200": (128, 8),          # 121.3; 256 x 8: 121.0, 256 x 4: 115.2 — flat
    "bssh_chain_L256_Ltau200_alpha0p2": (128, 8),
}


def available_cores():
    """Host cores this process may really use: the scheduler affinity capped by the cgroup CPU quota (a one-GPU box exposes every
    core of the node but grants 16 cores' worth of time)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


CPU_SWEEPS = 3  # whole sweeps' worth of solves each CPU worker times (about 5 s per core at the headline lattice)


def cpu_baseline(workload, tol, Nt):
    """The CPU oracle on the host cores the box gives this process: ONE walker per core, as the reference's MPI mode runs it
    (tutorials/holstein_honeycomb_mpi.jl) — every core times the same bounded sample (`cpu_sample`) in its own child process (no
    GPU, no threads inside), all at once, so that memory-bandwidth contention between the walkers is part of the figure.
    `value` is the aggregate over the cores; the single-core figure is reported next to it."""
    import subprocess

    cores = available_cores()
    single = cpu_sample(workload, tol, Nt, walker=0)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(w), "--workload", workload, "--cpu-tol", repr(tol), "--cpu-nt", str(Nt)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True) for w in range(cores)]
    rates, failures, its_rates, its_avgs = [], [], [], []
    for w, pr in enumerate(procs):
        try:
            out, err = pr.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, err = pr.communicate()
            failures.append({"worker": w, "returncode": "timeout", "stderr_tail": (err or "")[-300:]})
            continue
        try:
            if pr.returncode != 0:
                raise ValueError(f"exit code {pr.returncode}")
            rec = json.loads(out.strip().splitlines()[-1])
            rates.append(float(rec["value"]))
            its_rates.append(float(rec["cg_iterations_per_s"]))
            its_avgs.append(float(rec["avg_cg_iters"]))
        except (ValueError, IndexError, KeyError) as e:
            failures.append({"worker": w, "returncode": pr.returncode, "error": str(e), "stderr_tail": (err or "")[-300:]})
    agg = dict(single)
    agg["single_core_value"] = single["value"]
    agg["workers_failed"] = len(failures)
    if failures:
        # never substituted silently: the aggregate covers the workers that DID finish, the record says how many did not and why
        agg["degraded"] = True
        agg["worker_failures"] = failures[:4]
    if rates:
        agg.update({
            "value": sum(rates),
            "cores": len(rates),
            "sample": f"{len(rates)} walkers, one per core and all at once, each: " + single["sample"].split(": ", 1)[1],
            "per_core_min": min(rates),
            "per_core_max": max(rates),
            "cg_iterations_per_s": sum(its_rates),
            "avg_cg_iters": sum(its_avgs) / len(its_avgs),
        })
    return agg


def cpu_sample(workload, tol, Nt, walker=0):
    """Time the CPU oracle (single thread) on CPU_SWEEPS whole sweeps of one walker — THE SAME SWEEP the GPU leg times
    (oracle/sweep.py restates WalkerBatch._sweep with the device trajectory): two global moves with an action solve each, then
    hmc_update! with the EFA leapfrog — Nt force solves at sqrt(tol) on the EVOLVING fields, each followed by the force terms, the kick,
    evolve_eom! and update!, then the final action solve at tol — with update_preconditioner! before every solve, the walker's own
    PCG64 stream in the order the GPU walker draws it, and the move rejected at the end.  Field refreshes, force terms and the leapfrog
    are inside the timed span, as they are on the GPU.  Nothing is extrapolated."""
    import numpy as np

    from oracle.sweep import timed_sweeps

    t_total, its, w = timed_sweeps(workload, walker, CPU_SWEEPS, tol=tol, Nt=Nt)
    per = len(its) // CPU_SWEEPS
    it_a = sum(its[k] for k in range(len(its)) if k % per in (0, 1, per - 1))
    it_f = sum(its) - it_a
    # matvec alone, for the GB/s comparison
    g = np.random.default_rng(1 + walker)
    b = np.asfortranarray(g.standard_normal((w.Lt, w.N)) + 1j * g.standard_normal((w.Lt, w.N)))
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        w.fdm.mul_MtM(b)
    t_mv = (time.perf_counter() - t0) / reps
    V = w.Lt * w.N
    alg = 2 * (2 * 16 * V + 8 * V + 16 * w.Lt * w.nt.shape[1])
    return {
        "value": CPU_SWEEPS / t_total,
        "unit": "sweeps/s",
        "cores": 1,
        "kind": "port",
        "avg_cg_iters": sum(its) / len(its),
        "cg_iterations_per_s": sum(its) / t_total,
        "solves_per_sweep": per,
        "sample": f"walker {walker} of {workload}: {CPU_SWEEPS} whole sweeps, {t_total:.1f} s — the sweep the GPU leg times (oracle/sweep.py): per sweep 2 global moves + hmc_update! with the EFA "
        f"leapfrog, i.e. 3 action solves (tol {tol:g}, {it_a / (3 * CPU_SWEEPS):.1f} iters) + {Nt} force solves on the evolving fields (tol {float(np.sqrt(tol)):g}, {it_f / (Nt * CPU_SWEEPS):.1f} iters), "
        f"force terms, leapfrog and field refreshes included, KPM preconditioner updated before every solve, single thread, nothing extrapolated",
        "matvec_MtM_ms": t_mv * 1e3,
        "matvec_MtM_GBs": alg / t_mv / 1e9,
        "host_cores_available": available_cores(),
        "host_cores_visible": os.cpu_count(),
    }


def _only_factors(n, primes):
    for f in primes:
        while n % f == 0:
            n //= f
    return n == 1


def _in_place_tfft_exists(n):
    """The in-place τ-FFT exists for Lτ = 2^a 3^b 5^c (smoqy_tfft_form keeps the two-image form otherwise)."""
    return _only_factors(n, (2, 3, 5))


def rank_worker(args):
    """Child process of procs_per_gpu_scan: ONE walker behind ONE single-walker handle — exactly what an MPI rank of the reference owns
    (tutorials/holstein_honeycomb_mpi.jl:60-72) — sharing the GPU with its sibling ranks.  Protocol on stdin/stdout: build + one warm-up
    sweep, print "ready", wait for a line, time `--scan-sweeps` sweeps, print one JSON line with the wall-clock start / end."""
    from smoqyelphqmc_amd.walkers import WalkerBatch

    b = WalkerBatch(args.workload, nwalkers=1, walker0=args.rank_worker, device=0, device_efa=args.hmc == "device")
    b.sweep()
    b.h.call("smoqy_sync")
    b.stats.solves = b.stats.iters_sum = 0
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.time()
    for _ in range(args.scan_sweeps):
        b.sweep()
    b.h.call("smoqy_sync")
    t1 = time.time()
    print(json.dumps({"start": t0, "end": t1, "sweeps": args.scan_sweeps, "avg_cg_iters": b.stats.iters_sum / max(b.stats.solves, 1)}), flush=True)
    b.h.close()


def procs_per_gpu_scan(args, counts):
    """The reference's execution model on one GPU: K processes ("MPI ranks"), one single-walker handle each, all started together.
    Called BEFORE this process initialises the GPU, so the children are the only processes on the card.  Aggregate sweeps/s =
    K * sweeps / (last end - first start).  A failed point is reported as such, never substituted."""
    import subprocess

    out = []
    for K in counts:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--hmc", args.hmc, "--scan-sweeps", str(args.scan_sweeps)]
        procs = [subprocess.Popen(cmd + ["--rank-worker", str(1000 + w)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w in range(K)]
        rec = {"procs": K, "walkers_per_proc": 1}
        try:
            for pr in procs:
                line = pr.stdout.readline()
                if line.strip() != "ready":
                    raise RuntimeError("worker did not come up: " + (pr.stderr.read() or "")[-300:])
            for pr in procs:
                pr.stdin.write("go\n")
                pr.stdin.flush()
            res = []
            for pr in procs:
                o, e = pr.communicate(timeout=300)
                if pr.returncode != 0:
                    raise RuntimeError(f"worker exit code {pr.returncode}: " + (e or "")[-300:])
                res.append(json.loads(o.strip().splitlines()[-1]))
            span = max(r["end"] for r in res) - min(r["start"] for r in res)
            rec.update({"sweeps_per_s": K * args.scan_sweeps / span, "per_proc_sweeps_per_s": [r["sweeps"] / (r["end"] - r["start"]) for r in res],
                        "avg_cg_iters": sum(r["avg_cg_iters"] for r in res) / K, "sweeps_each": args.scan_sweeps})
        except Exception as ex:  # noqa: BLE001 — the scan must never take the bench line down with it
            rec["error"] = str(ex)[:400]
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
            for pr in procs:
                try:
                    pr.communicate(timeout=30)
                except Exception:  # noqa: BLE001
                    pass
        out.append(rec)
    return out


def threads_per_gpu_scan(args, counts, dev, walker0):
    """K host threads in ONE process, each owning a single-walker handle (its own HIP stream) — what a threaded driver that keeps the
    reference's one-walker objects gets.  Same timing rule as procs_per_gpu_scan."""
    from concurrent.futures import ThreadPoolExecutor

    from smoqyelphqmc_amd.walkers import WalkerBatch

    out = []
    for K in counts:
        bs = [WalkerBatch(args.workload, nwalkers=1, walker0=walker0 + 2000 + w, device=dev, device_efa=args.hmc == "device", host_threads=1) for w in range(K)]
        with ThreadPoolExecutor(K) as pool:
            list(pool.map(lambda b: (b.sweep(), b.h.call("smoqy_sync")), bs))
            for b in bs:
                b.stats.solves = b.stats.iters_sum = 0
            t0 = time.perf_counter()
            list(pool.map(lambda b: ([b.sweep() for _ in range(args.scan_sweeps)], b.h.call("smoqy_sync")), bs))
            span = time.perf_counter() - t0
        out.append({"threads": K, "walkers_per_thread": 1, "sweeps_per_s": K * args.scan_sweeps / span, "sweeps_each": args.scan_sweeps,
                    "avg_cg_iters": sum(b.stats.iters_sum for b in bs) / max(sum(b.stats.solves for b in bs), 1)})
        for b in bs:
            b.h.close()
    return out


def team_scan(args, counts, dev, walker0):
    """K host threads, each running the per-walker sweep of the reference against ITS OWN walker only, through a walker team (C ABI
    "walker teams"): the calls of the K members rendezvous inside the library and run as one batched call.  Next to it the same sweep
    (host-driven HMC: x uploaded, force downloaded every step) driven by one caller through the batched entry points."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from smoqyelphqmc_amd import _lib as L
    from smoqyelphqmc_amd.walkers import WalkerBatch, WalkerTeam

    out = []
    dev_hmc = getattr(args, "hmc", "device") == "device"  # the members' hmc_update! with the trajectory on the device (smoqy_team_hmc_update) or host-driven, step by step
    member_sweep = (lambda m: m.sweep_device_hmc()) if dev_hmc else (lambda m: m.sweep())
    for K in counts:
        team = WalkerTeam(args.workload, K, walker0=walker0 + 3000, device=dev, device_efa=dev_hmc)
        draw_pool = None
        if dev_hmc and not getattr(args, "no_prefetch", False):
            draw_pool = ThreadPoolExecutor(min(K, 16))  # the members' random numbers, one sweep ahead (as the lock-step batches draw theirs)
            for m in team.members:
                m.prefetch_randoms(draw_pool)
        with ThreadPoolExecutor(K) as pool:
            list(pool.map(member_sweep, team.members))
            t0 = time.perf_counter()
            list(pool.map(lambda m: [member_sweep(m) for _ in range(args.scan_sweeps)], team.members))
            span = time.perf_counter() - t0
        iters = sum(m.iters_sum for m in team.members) / max(sum(m.solves for m in team.members), 1)
        # the same member sweep from K native threads (no interpreter lock): smoqy_team_bench_sweeps
        b = team.batch
        x0 = np.ascontiguousarray(np.stack([np.asarray(b.xs_force[w]) for w in range(K)]))
        secs, so, itn = L.C.c_double(0.0), L.C.c_long(0), L.C.c_long(0)
        team.call("smoqy_team_bench_sweeps", L.ptr(x0), int(b.Nph), L.C.c_double(b.drift), int(b.Nt), L.C.c_double(b.tol), L.C.c_double(b.tol_force), int(b.maxiter), int(dev_hmc), 1,
                  int(args.scan_sweeps), 4242 + walker0, L.C.byref(secs), L.C.byref(so), L.C.byref(itn))
        native = K * args.scan_sweeps / secs.value
        native_iters = itn.value / max(so.value, 1)
        if draw_pool is not None:
            draw_pool.shutdown(wait=True)
        team.close()
        ob = WalkerBatch(args.workload, nwalkers=K, walker0=walker0 + 3000, device=dev, device_efa=dev_hmc, prefetch_randoms=not getattr(args, "no_prefetch", False))
        ob.sweep()
        ob.h.call("smoqy_sync")
        t1 = time.perf_counter()
        for _ in range(args.scan_sweeps):
            ob.sweep()
        ob.h.call("smoqy_sync")
        span_b = time.perf_counter() - t1
        ob.h.close()
        out.append({"threads": K, "walkers_per_thread": 1, "handles": 1, "hmc": "device" if dev_hmc else "host", "sweeps_per_s": K * args.scan_sweeps / span, "avg_cg_iters": iters,
                    "native_threads_sweeps_per_s": native, "native_avg_cg_iters": native_iters,
                    "one_caller_batched_sweeps_per_s": K * args.scan_sweeps / span_b, "sweeps_each": args.scan_sweeps})
    # several teams side by side (one handle and stream each), native member threads only: T x K per-walker control flows on one GPU
    for spec in [q for q in getattr(args, "team_multi", "").split(",") if q]:
        T, K = (int(v) for v in spec.split("x"))
        teams = [WalkerTeam(args.workload, K, walker0=walker0 + 5000 + 64 * q, device=dev, device_efa=dev_hmc) for q in range(T)]

        def native_run(team, warm, sweeps):
            b = team.batch
            x0 = np.ascontiguousarray(np.stack([np.asarray(b.xs_force[w]) for w in range(K)]))
            secs, so, itn = L.C.c_double(0.0), L.C.c_long(0), L.C.c_long(0)
            team.call("smoqy_team_bench_sweeps", L.ptr(x0), int(b.Nph), L.C.c_double(b.drift), int(b.Nt), L.C.c_double(b.tol), L.C.c_double(b.tol_force), int(b.maxiter), int(dev_hmc), warm,
                      sweeps, 777 + walker0, L.C.byref(secs), L.C.byref(so), L.C.byref(itn))
            return itn.value, so.value

        with ThreadPoolExecutor(T) as pool:
            list(pool.map(lambda tm: native_run(tm, 0, 1), teams))            # warm-up, untimed
            t0 = time.perf_counter()
            res = list(pool.map(lambda tm: native_run(tm, 0, args.scan_sweeps), teams))
            span = time.perf_counter() - t0
        for tm in teams:
            tm.close()
        out.append({"teams": T, "threads": T * K, "walkers_per_thread": 1, "handles": T, "hmc": "device" if dev_hmc else "host", "native_threads_sweeps_per_s": T * K * args.scan_sweeps / span,
                    "native_avg_cg_iters": sum(r[0] for r in res) / max(sum(r[1] for r in res), 1), "sweeps_each": args.scan_sweeps})
    return out


def member_worker(spec):
    """Child process of team_procs_scan: ONE rank of the reference's one-walker-per-rank model (tutorials/holstein_honeycomb_mpi.jl:60-72),
    joined to a team that the parent process serves (smoqy_team_serve).  No GPU, no handle: the per-walker sweep against its own walker,
    its own rng, every library call through shared memory.  One warm-up sweep, then the timed ones."""
    from smoqyelphqmc_amd.walkers import RemoteMember

    info, w, sweeps = spec.rsplit("|", 2)
    info = json.loads(info)
    m = RemoteMember(info, int(w), seed=4711, wait_seconds=120.0)
    sweep = m.sweep_device_hmc if info.get("device_efa") else m.sweep
    if info.get("device_efa") and info.get("prefetch", True):
        from concurrent.futures import ThreadPoolExecutor

        m.prefetch_randoms(ThreadPoolExecutor(1))  # the rank draws its next sweep's random numbers on a thread of its own while it waits in the rendezvous
    sweep()
    m.solves = m.iters_sum = 0
    t0 = time.time()
    for _ in range(int(sweeps)):
        sweep()
    t1 = time.time()
    print(json.dumps({"start": t0, "end": t1, "solves": m.solves, "iters": m.iters_sum}), flush=True)
    m.close()


def team_procs_scan(args, points, dev, walker0):
    """The reference's execution model made to scale on one GPU: K member PROCESSES (ranks), each driving its own walker only, joined
    through shared memory to ONE batched handle that this process owns and serves (TxK: T served teams side by side, one handle and
    stream each).  Aggregate sweeps/s = members * sweeps / (last end - first start)."""
    import subprocess

    from smoqyelphqmc_amd.walkers import WalkerTeam

    out = []
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for spec in points:
        T, K = (int(v) for v in spec.split("x")) if "x" in spec else (1, int(spec))
        dev_hmc = getattr(args, "hmc", "device") == "device"
        rec = {"teams": T, "procs": T * K, "walkers_per_proc": 1, "handles": T, "hmc": "device" if dev_hmc else "host"}
        teams, procs = [], []
        try:
            for q in range(T):
                tm = WalkerTeam(args.workload, K, walker0=walker0 + 7000 + 64 * q, device=dev, device_efa=dev_hmc)
                teams.append(tm)
                info = tm.serve(f"/smoqy-bench-{os.getpid()}-{q}")
                info["prefetch"] = not getattr(args, "no_prefetch", False)
                procs += [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--member-worker", json.dumps(info) + f"|{w}|{args.scan_sweeps}"],
                                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True) for w in range(K)]
            res = []
            for pr in procs:
                o, e = pr.communicate(timeout=600)
                if pr.returncode != 0:
                    raise RuntimeError((e or "")[-300:])
                res.append(json.loads(o.strip().splitlines()[-1]))
            span = max(r["end"] for r in res) - min(r["start"] for r in res)
            rec.update({"sweeps_per_s": T * K * args.scan_sweeps / span, "avg_cg_iters": sum(r["iters"] for r in res) / max(sum(r["solves"] for r in res), 1), "sweeps_each": args.scan_sweeps})
        except Exception as ex:  # noqa: BLE001 — a failed point is reported, never substituted
            rec["error"] = str(ex)[:300]
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        finally:
            for tm in teams:
                tm.close()
        out.append(rec)
    return out


def iteration_kernels(args, dev, walker0, L, np):
    """All four kernels of the fused CG iteration against the roofline, measured live inside real solves: 16 walkers on one stream, one part
    (full-batch launches), HIP events in front of each launch of up to 512 iterations of one sweep (smoqy_cg_iteration_timing).  Bytes: the
    least each fused kernel can move (every array it reads or writes counted once; S = one state vector, F = the field arrays) and the
    committed PMC traffic of the same kernels (profiles/r*_pmc_iteration.json)."""
    from smoqyelphqmc_amd.walkers import WalkerBatch

    nw = 16
    ob = WalkerBatch(args.workload, nwalkers=nw, walker0=walker0, device=dev, device_efa=args.hmc == "device", cg_split=1)
    ob.sweep()
    ob.h.call("smoqy_cg_iteration_timing", 512)
    ob.sweep()
    us = (L.C.c_double * 4)()
    n = L.C.c_int(0)
    ob.h.call("smoqy_cg_iteration_timing_read", us, L.C.byref(n))
    Sb = 16.0 * ob.Lt * ob.N * nw
    Fb = (8.0 * ob.Lt * ob.N + 16.0 * ob.Lt * ob.Nh) * nw
    names = ob.h.describe()
    ob.h.close()
    _, per_kernel, src = committed_iteration_traffic(args.workload)
    per_kernel = per_kernel or {}

    def hit(k, prefix):  # "tfft<2" stands for both τ-FFT kernels' CG modes: tfft_kernel<2, …> and tfft_rb_kernel<2, …>
        return (k.startswith("tfft_kernel<" + prefix[5:]) or k.startswith("tfft_rb_kernel<" + prefix[5:])) if prefix.startswith("tfft<") else k.startswith(prefix)

    def traffic(prefix):
        hits = [v for k, v in per_kernel.items() if hit(k, prefix)]
        return hits[0] if hits else None

    tf = names.get("tfft", "tfft_kernel")
    rows = [(f"fused MtM ({names['mtm']})", us[0], 2 * Sb + Fb, traffic("fdm_")),
            (f"forward tau-FFT + r update ({tf}, mode 2)", us[1], 3 * Sb, traffic("tfft<2")),
            (f"Chebyshev apply ({names['cheb']})", us[2], 2 * Sb, traffic("cheb_")),
            (f"inverse tau-FFT + x, p updates ({tf}, mode 3)", us[3], 5 * Sb, traffic("tfft<3"))]
    prof, prof_src = committed_solo_durations(args.workload)
    out = []
    for (name, t_us, least, tr), key in zip(rows, ("fdm_", "tfft<2", "cheb_", "tfft<3")):
        t_s = t_us * 1e-6
        p_us = next((v for k, v in prof.items() if hit(k, key)), None)
        out.append({"kernel": name, "us": t_us, "least_bytes": least, "frac_least": (least / t_s / 1e9 / HBM_PEAK_GBS) if t_s else None,
                    "traffic_bytes": tr, "frac_traffic": (tr / t_s / 1e9 / HBM_PEAK_GBS) if (tr and t_s) else None,
                    "rocprof_us": p_us, "frac_least_rocprof": (least / (p_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if p_us else None})
    tot = sum(r[1] for r in rows)
    return {"systems_per_launch": nw, "iterations_sampled": n.value, "us_per_iteration": tot,
            "timing": "us: HIP events in front of each of the four dependent launches inside real solves, one stream, one part — event to event, i.e. the kernel plus the hand-over to "
                      "the next launch, which the event commands themselves lengthen by about 3 us per launch; rocprof_us: rocprofv3's per-dispatch average of the same kernels in the "
                      "same one-stream run (committed file, not measured by this run)",
            "rocprof_source": prof_src,
            "traffic_source": src, "traffic_measured_live": False, "peak_GBs": HBM_PEAK_GBS,
            "note": "at 16 systems the working set (5 vectors, 84 MB) sits in the Infinity Cache: these are not HBM figures; frac_least = least bytes the fused kernel can move / time / 8 TB/s",
            "kernels": out}


SOLO_TAGS = {"holstein_honeycomb_L16_Ltau128": "hc16", "holstein_honeycomb_L8_Ltau80": "hc8", "ossh_square_L12_Ltau100": "ossh", "bssh_chain_L256_Ltau200": "bssh",
             "holstein_honeycomb_L4_Ltau40": "hc4"}  # tags of profiles/r*_solo_kernel_stats_<tag>.txt and r*_pmc_iteration_<tag>.json


def _async_stats(batches, L):
    """smoqy_hmc_async counters of rank 0's batches: trajectories launched without host waits, and how many of those had to be repeated
    with polls (a repeat costs the trajectory twice; the library stops trying where solves are long or misses recur)."""
    runs = misses = 0
    for b in batches:
        r, m = L.C.c_long(0), L.C.c_long(0)
        b.h.call("smoqy_hmc_async", -1, L.C.byref(r), L.C.byref(m))
        runs, misses = runs + r.value, misses + m.value
    return {"trajectories_without_host_waits": runs, "of_which_repeated_with_polls": misses, "scope": "rank 0, warm-up included"}


def committed_solo_durations(workload):
    """rocprofv3 per-dispatch averages (us) of the iteration's kernels in the newest committed one-stream, 16-walker profile
    (profiles/r*_solo_kernel_stats_<tag>.txt) of this workload."""
    import glob
    tag = SOLO_TAGS.get(workload)
    if tag is None:
        return {}, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_solo_kernel_stats_{tag}.txt")), reverse=True):
        out = {}
        try:
            for line in open(path):
                if not line.startswith("void smoqy::"):
                    continue
                name = line.replace("void ", "").replace("smoqy::", "").replace("(anonymous namespace)::", "")
                head = name.split("(")[0]
                nums = [c for c in line.split() if c.replace(".", "", 1).isdigit()]
                if len(nums) >= 6:
                    out[head] = float(nums[-4])  # calls total_ms avg_us min_us max_us pct
        except OSError:
            continue
        if out:
            return out, "profiles/" + os.path.basename(path)
    return {}, None


def committed_iteration_traffic(workload):
    """HBM-side bytes of ONE CG iteration at 16 systems per launch (all four kernels), from the newest committed rocprofv3 PMC pass
    (profiles/r*_pmc_iteration[_<tag>].json; FETCH_SIZE x 2 + WRITE_SIZE per kernel, medians over a sweep) of this workload."""
    import glob
    tag = SOLO_TAGS.get(workload)
    names = [f"r*_pmc_iteration_{tag}.json"] + (["r*_pmc_iteration.json"] if tag == "hc16" else [])  # (rounds 1-3 profiled the headline lattice only, untagged)
    paths = sorted((q for n in names for q in glob.glob(os.path.join(ROOT, "profiles", n))), key=os.path.basename, reverse=True) if tag else []
    for path in paths:
        try:
            pmc = json.load(open(path))
            if pmc.get("_workload", "holstein_honeycomb_L16_Ltau128") != workload:
                continue
            per_kernel = {k.replace("void ", "").replace("smoqy::", "").replace("(anonymous namespace)::", "").split("(")[0]: v["traffic_MB"] * 2**20
                          for k, v in pmc.items() if isinstance(v, dict) and "traffic_MB" in v}
        except (OSError, ValueError, KeyError):
            continue
        if per_kernel:
            return sum(per_kernel.values()), per_kernel, "profiles/" + os.path.basename(path)
    return None, None, None


def measure_copy_ceiling(h, L, gib=1.0, reps=10):
    """Device stream-copy ceiling measured on THIS box: a plain 16-byte-per-lane copy kernel over 2 x `gib` GiB
    (far beyond the 256 MiB Infinity Cache), HIP events on the handle's stream.  Moved bytes = read + write."""
    nbytes = int(gib * (1 << 30))
    ms = L.C.c_double(0.0)
    h.call("smoqy_bench_copy", L.C.c_size_t(nbytes), reps, L.C.byref(ms))
    gbs = 2.0 * nbytes * reps / (ms.value * 1e-3) / 1e9
    return {"GBs": gbs, "bytes_per_copy": 2 * nbytes, "reps": reps, "frac_of_spec": gbs / HBM_PEAK_GBS,
            "note": "hand-written 16-byte-per-lane nontemporal copy kernel (shape from tools/copy_probe.hip), src and dst 1 GiB each; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy"}


def committed_traffic(workload, systems):
    """HBM-side bytes per fused-MᵀM launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, collected in
    separate --pmc runs per MI355X_MICROARCH.md; counters cannot be read from inside bench.py).  Newest round first; only quoted
    when the profile was taken at this workload and batch size."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_fdm_mtm*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        recs = pmc if isinstance(pmc, list) else [pmc]
        for r in recs:
            if r.get("workload") == workload and r.get("systems_per_launch") == systems and "traffic_bytes_per_launch" in r:
                return r["traffic_bytes_per_launch"], "profiles/" + os.path.basename(path)
    return None, None


def roofline_record(args, batch, per, S, dev, insitu, extra, L, np):
    """The `roofline` object of the JSON line.  Dominant kernel: the fused MᵀM apply.

    achieved / frac          ALGORITHMIC bytes per launch (2(2S+F) per system, F once per walker) over the average
                             duration of the launches sampled INSIDE the timed region by the device clock (agrees with rocprofv3's
                             per-dispatch duration of the same command, profiles/).
    traffic / frac_traffic   bytes the kernel really moves beyond L2 (committed PMC passes; the fused kernel moves each array once,
                             ~0.41 of the two-pass algorithmic count) over the same duration.
    isolated                 the same kernel alone on the GPU (back-to-back launches, HIP events; `--roofline-only` repeats this leg).
    copy_ceiling             device stream-copy rate measured in this run; every fraction is also quoted against it.
    frac_single_pass         (2S+F) per system over the same duration: what the fused kernel must at least move (it reads v, the fields and
                             writes MᵀM v once); cannot exceed 1.  `frac` counts two passes because SURVEY.md §8(d) says to, so at large
                             batch it can — every value above 1 is flagged `"physical": false`.
    batch_scan               1..128 systems per launch; the 128-system point is the HBM-resident one (`hbm_resident_point`, working set
                             512 MiB > the 256 MiB Infinity Cache).
    """
    h = batch.h
    alg = h.algorithmic_bytes(L.OP_MTM)
    if args.timed_only:  # nothing but the timed region's own launches (rocprofv3 cross-check)
        if not insitu["device_us"]:
            return {"bound": "hbm", "note": "no MtM launches were sampled in this run"}
        t_s = insitu["device_us"] * 1e-6
        traffic, traffic_src = committed_traffic(args.workload, per)
        return {"bound": "hbm", "kernel": h.describe()["mtm"] + " (fused MtM; the kernel family the handle's last full-batch launch ran, smoqy_describe)", "achieved": alg / t_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t_s / 1e9 / HBM_PEAK_GBS,
                "frac_single_pass": 0.5 * alg / t_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_live": False, "frac_traffic": (traffic / t_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "avg_launch_us": insitu["device_us"], "duration_source": "timed region, device clock", "launches_sampled": insitu["device_n"], "event_pair_avg_us": insitu["event_us"],
                "algorithmic_bytes_per_launch": alg, "systems_per_launch": per, "concurrent_streams": S}
    a, b = h.vec_alloc(), h.vec_alloc()
    g = np.random.default_rng(3)
    h.vec_upload(a, np.asfortranarray(g.standard_normal((batch.Lt, batch.N, per)) + 1j * g.standard_normal((batch.Lt, batch.N, per))))
    h.bench_matvec(L.OP_MTM, b, a, 50)
    ms = h.bench_matvec(L.OP_MTM, b, a, args.matvec_reps)
    iso_s = ms * 1e-3 / args.matvec_reps
    tc = L.C.c_int(0)
    h.call("smoqy_get_tau_chunk", L.C.byref(tc))
    traffic, traffic_src = committed_traffic(args.workload, per)
    copy = measure_copy_ceiling(h, L)

    def fr(bytes_, sec):
        gbs = bytes_ / sec / 1e9
        return {"GBs": gbs, "frac": gbs / HBM_PEAK_GBS, "frac_of_copy_ceiling": gbs / copy["GBs"]}

    isolated = {"avg_launch_us": iso_s * 1e6, "launches": args.matvec_reps, "algorithmic": fr(alg, iso_s), "single_pass": fr(0.5 * alg, iso_s), "traffic": fr(traffic, iso_s) if traffic else None,
                "note": "back-to-back launches of the kernel alone on the handle's stream, HIP events around all of them; 16 systems (64 MiB in + out + fields) sit in the Infinity Cache"}
    timed_s = insitu["device_us"] * 1e-6 if insitu["device_us"] else None
    prim_s, prim_src = (timed_s, "timed region, device clock") if timed_s else (iso_s, "isolated leg (no timed region in this run)")
    roofline = {
        "bound": "hbm",
        "kernel": h.describe()["mtm"] + " (fused MᵀM apply: the kernel family the timed batches' last full-batch launch ran, from smoqy_describe; batch_scan points carry their own)",
        "iteration_kernels_of_the_timed_batches": h.describe(),
        "achieved": alg / prim_s / 1e9,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": alg / prim_s / 1e9 / HBM_PEAK_GBS,
        "achieved_is": "algorithmic bytes (SURVEY.md §8(d): two passes counted for MᵀM although the kernel is fused) / launch duration; the physical readings are frac_single_pass and frac_traffic",
        "frac_single_pass": 0.5 * alg / prim_s / 1e9 / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "traffic_measured_live": False,
        "frac_traffic": (traffic / prim_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "frac_of_copy_ceiling": alg / prim_s / 1e9 / copy["GBs"],
        "frac_traffic_of_copy_ceiling": (traffic / prim_s / 1e9 / copy["GBs"]) if traffic else None,
        "avg_launch_us": prim_s * 1e6,
        "duration_source": prim_src,
        "launches_sampled": insitu["device_n"] if timed_s else args.matvec_reps,
        "algorithmic_bytes_per_launch": alg,
        "systems_per_launch": per,
        "tau_chunk": tc.value,
        "concurrent_streams": S,
        "solves_in_flight": int(args.solve_concurrency) if args.solve_concurrency else S,
        "shared_device": "the sampled launches run while the kernels of up to solves_in_flight - 1 other CG solves share the GPU, so their duration is a shared-device figure: it grows "
                         "with the gate (22 us at 3 solves in flight, 29 us at 4) while sweeps/s rises; the kernel alone is `isolated`, the whole step in real bytes is cg_iteration_traffic",
        "in_timed_region": None if not timed_s else {
            "device_clock_avg_us": insitu["device_us"], "event_pair_avg_us": insitu["event_us"], "launches_sampled": insitu["device_n"],
            "note": "every 16th full-batch MtM launch of the CG loops while all streams run; the device clock spans first workgroup start -> last workgroup end "
                    "(rocprofv3's dispatch duration), the event pair also holds the dependency gap to the previous launch on the stream",
        },
        "isolated": isolated,
        "copy_ceiling": copy,
        "note": "frac = algorithmic bytes (two passes counted for the fused kernel) over the in-run duration; frac_single_pass = one pass (2S+F per system), the least the fused kernel "
                "can move; frac_traffic = bytes really moved beyond L2 over the same duration; at 16 systems per launch the working set is Infinity-Cache resident, "
                "the HBM-resident figure is batch_scan's 128-system point (hbm_resident_point)",
    }
    # GB/s versus batch size (SURVEY.md §8(d) latency caveat), always reported: heuristic tau chunk, 200 launches per point
    scan = []
    for nb in (() if args.roofline_only else (1, 2, 4, 8, 16, 32, 64, 128)):  # --roofline-only: the isolated leg alone, for its rocprofv3 cross-check
        hb = L.Handle(batch.Lt, batch.N, batch.nt, batch.colors, True, nb, 1, dev)
        for w in range(nb):
            m = batch.models[w % per]
            hb.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(batch.perm), L.C.c_double(m.fpi.dtau))
        va, vb = hb.vec_alloc(), hb.vec_alloc()
        hb.vec_upload(va, np.asfortranarray(g.standard_normal((batch.Lt, batch.N, nb)) + 1j * g.standard_normal((batch.Lt, batch.N, nb))))
        best = None
        for tcand in ((1, 2, 3, 4) if args.batch_scan else (0,)):
            hb.call("smoqy_set_tau_chunk", tcand)
            hb.bench_matvec(L.OP_MTM, vb, va, 20)
            t_s = hb.bench_matvec(L.OP_MTM, vb, va, 200) / 200 * 1e-3
            algb = hb.algorithmic_bytes(L.OP_MTM)
            tcv = L.C.c_int(0)
            hb.call("smoqy_get_tau_chunk", L.C.byref(tcv))
            tr, _ = committed_traffic(args.workload, nb)
            rec = {"batch": nb, "kernel": hb.describe()["mtm"], "tau_chunk": tcv.value, "us": t_s * 1e6, "GBs": algb / t_s / 1e9, "frac": algb / t_s / 1e9 / HBM_PEAK_GBS,
                   "frac_single_pass": 0.5 * algb / t_s / 1e9 / HBM_PEAK_GBS, "physical": algb / t_s / 1e9 <= HBM_PEAK_GBS,
                   "working_set_MiB": (2 * 16.0 * batch.Lt * batch.N * nb + nb * (8.0 * batch.Lt * batch.N + 16.0 * batch.Lt * batch.Nh)) / 2**20,
                   "traffic_GBs": (tr / t_s / 1e9) if tr else None, "frac_traffic": (tr / t_s / 1e9 / HBM_PEAK_GBS) if tr else None}
            rec["hbm_resident"] = rec["working_set_MiB"] > 256.0  # beyond the 256 MiB Infinity Cache
            if best is None or rec["GBs"] > best["GBs"]:
                best = rec
        scan.append(best)
        hb.close()
    roofline["batch_scan"] = scan
    roofline["batch_at_40pct"] = next((r["batch"] for r in scan if r["frac"] >= 0.4), None)
    roofline["hbm_resident_point"] = next((r for r in scan if r["hbm_resident"]), None)
    h.call("smoqy_vec_free", a)
    h.call("smoqy_vec_free", b)
    return roofline


def main():
    args = parse()
    if args.cpu_worker >= 0:  # CPU-only child of cpu_baseline(): never touches the GPU
        print(json.dumps(cpu_sample(args.workload, args.cpu_tol, args.cpu_nt, walker=args.cpu_worker)))
        return
    if args.member_worker:  # CPU-only child of team_procs_scan
        member_worker(args.member_worker)
        return
    if args.rank_worker >= 0:  # GPU child of procs_per_gpu_scan
        rank_worker(args)
        return
    # the reference's rank-per-walker model on one GPU, measured first: this process has not touched the GPU yet, so the K children are
    # the only processes on the card (N = 1 only; outside the timed region)
    proc_scan = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not (args.no_proc_scan or args.roofline_only or args.timed_only)):
        proc_scan = procs_per_gpu_scan(args, [int(k) for k in args.proc_scan.split(",") if k])
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = 0 if args.rehearse_one_gpu else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    import numpy as np

    import smoqyelphqmc_amd as sq
    from smoqyelphqmc_amd import _lib as L
    from smoqyelphqmc_amd.walkers import WalkerBatch

    from concurrent.futures import ThreadPoolExecutor

    from smoqyelphqmc_amd.sharding import aggregate_throughput, reduce_max_time, walker_range

    shape = DEFAULT_SHAPES.get(args.workload, (128, 8))
    if not args.walkers_per_gpu:
        args.walkers_per_gpu = shape[0] if not args.streams else 16 * args.streams
    if not args.streams:
        args.streams = shape[1] if args.walkers_per_gpu % shape[1] == 0 else 1
    wpg, S = args.walkers_per_gpu, max(1, args.streams)
    if wpg % S:
        raise SystemExit("--walkers-per-gpu must be a multiple of --streams")
    mine = walker_range(rank, world, wpg)  # walkers [rank*wpg, (rank+1)*wpg): no overlap between ranks, no exchange
    per = wpg // S
    # auto: the in-place τ-FFT pays where the launches fill the chip and the transform has no radix-5 pass (headline lattice +2.7 %;
    # measured slower on the Lτ = 80 / 100 / 200 lattices of BASELINE.json, whose launches are half a chip or less)
    from smoqyelphqmc_amd import lattice as _lat
    _m0 = _lat.CONFIGS[args.workload](walker=0)
    lat_Lt, lat_N = _m0.fpi.Ltau, _m0.fpi.N
    # host threads of this rank: S stream threads + a small RNG pool per batch, bounded by the cores the rank can count on (a full node
    # gives every rank available_cores() // world; one GPU box gives 16)
    cores_rank = max(1, available_cores() // world)
    rng_threads = max(1, min(max(2, 16 // S), cores_rank // S))
    batches = [WalkerBatch(args.workload, nwalkers=per, walker0=mine.start + s * per, device=dev, check_every=args.check_every or None, tau_chunk=args.tau_chunk or None,
                           host_threads=rng_threads, measure_nrv=args.measure_nrv, device_efa=args.hmc == "device", cg_split=args.cg_split, prefetch_randoms=not args.no_prefetch,
                           tfft_in_place=(S > 1 and _in_place_tfft_exists(lat_Lt) and lat_Lt * lat_N >= 14400) if args.tfft_form == "auto" else args.tfft_form == "in-place") for s in range(S)]  # the box gives one GPU 16 cores: S stream threads + small RNG pools
    batch = batches[0]
    if os.environ.get("SMOQY_BENCH_ASYNC"):  # A/B aid: the asynchronous trajectory (smoqy_hmc_async) on / off in the timed batches
        for b in batches:
            b.h.call("smoqy_hmc_async", int(os.environ["SMOQY_BENCH_ASYNC"]), None, None)
    if os.environ.get("SMOQY_BENCH_GRAPH"):  # A/B aid: hipGraph replay of the CG iterations in the timed batches (needs --no-mtm-sampling)
        for b in batches:
            b.h.call("smoqy_cg_use_graph", int(os.environ["SMOQY_BENCH_GRAPH"]))
    if args.solve_concurrency > 0:
        if args.gate == "library":
            # the library's own process-wide gate: held around each CG loop only, so the preconditioner updates, force kernels and leapfrog
            # steps of a trajectory overlap with the other batches' solves
            L.load().smoqy_cg_gate(args.solve_concurrency)
        else:
            import threading
            WalkerBatch.solve_gate = threading.Semaphore(args.solve_concurrency)  # held around whole library calls (a trajectory = 24 solves)
    pool = ThreadPoolExecutor(S) if S > 1 else None

    def run(nsweeps):
        if pool is None:
            for _ in range(nsweeps):
                batch.sweep()
        else:
            list(pool.map(lambda b: [b.sweep() for _ in range(nsweeps)], batches))

    def fence():
        for b in batches:
            b.h.call("smoqy_sync")
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if args.roofline_only:
        args.warmup, args.steps = 0, 0
    if args.timed_only or os.environ.get("SMOQY_BENCH_MAPS"):
        # profiled runs keep the process's module map (every library loaded, every handle and stream created) so that an abort under
        # rocprofv3 can be attributed: the two aborts on record (profiles/r03_profiler_sigsegv_stack.txt) had only "(unknown)" frames
        try:
            path = os.environ.get("SMOQY_BENCH_MAPS") or os.path.join(ROOT, "gpurun_out", f"bench_maps_rank{rank}.txt")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(f"# /proc/self/maps of bench.py pid {os.getpid()} after the handles were created, before the first launch of the sweeps\n")
                f.write(open("/proc/self/maps").read())
        except OSError:
            pass
    # the FIRST sweep of every batch runs here, on the thread that created the handles, one batch after the other: the first launch of
    # every kernel family (lazy code-object loading, per-stream runtime state, the profiler's queue interception when one is attached)
    # never happens from two threads at once.  It is the first of the `--warmup` sweeps, untimed like the rest of them.
    warm_seq = 1 if (args.warmup > 0 and pool is not None) else 0
    for b in (batches if warm_seq else ()):
        b.sweep()
        b.h.call("smoqy_sync")
    run(args.warmup - warm_seq)
    for b in batches:
        b.stats.solves = b.stats.iters_sum = 0
        # roofline: the dominant kernel's launches inside the timed region are sampled with HIP events on the stream
        # they run on (every 16th full-batch fused MᵀM launch of the CG loop)
        if not args.no_mtm_sampling:
            b.h.call("smoqy_matvec_timing", 16, 1024)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = reduce_max_time(time.perf_counter() - t0, device="cpu" if args.rehearse_one_gpu else "cuda")
    value = aggregate_throughput(wpg * args.steps, world, elapsed) if args.steps else 0.0
    # sampled fused-MᵀM launches of the timed region: by the device's own clock (first workgroup's start -> last workgroup's end,
    # what rocprofv3 reports per dispatch) and by the HIP event pairs around them (which also hold the gap to the previous launch)
    ev_us = ev_n = dv_us = dv_n = 0.0
    for b in batches:
        us, n = L.C.c_double(0.0), L.C.c_int(0)
        b.h.call("smoqy_matvec_timing_read_device", L.C.byref(us), L.C.byref(n))
        dv_us += us.value * n.value
        dv_n += n.value
        b.h.call("smoqy_matvec_timing_read", L.C.byref(us), L.C.byref(n))
        ev_us += us.value * n.value
        ev_n += n.value
    insitu = {"device_us": dv_us / dv_n if dv_n else None, "device_n": int(dv_n), "event_us": ev_us / ev_n if ev_n else None, "event_n": int(ev_n)}

    if rank == 0:
        extra = {}
        roofline = roofline_record(args, batch, per, S, dev, insitu, extra, L, np)
        # the literal BASELINE.json configuration — ONE walker per GPU — and the 8- / 16-walker one-stream figures next to the
        # batched headline (outside the timed region): what a user who keeps the reference's rank-per-walker model gets
        if world == 1 and not args.roofline_only and not args.timed_only:
            one_stream = []
            for nw1 in (1, 8, 16):
                ob = WalkerBatch(args.workload, nwalkers=nw1, walker0=mine.start, device=dev, device_efa=args.hmc == "device", prefetch_randoms=not args.no_prefetch)
                ob.sweep()
                ob.h.call("smoqy_sync")
                t1 = time.perf_counter()
                for _ in range(2):
                    ob.sweep()
                ob.h.call("smoqy_sync")
                ms_one = (time.perf_counter() - t1) / 2 * 1e3
                ob.h.close()
                one_stream.append({"walkers_per_gpu": nw1, "streams": 1, "sweeps_per_s": nw1 * 1e3 / ms_one, "ms_per_sweep": ms_one})
            extra["single_walker"] = dict(one_stream[0], note="one walker on one stream: the launch-latency regime (4 dependent launches per CG iteration)")
            extra["one_stream"] = one_stream
            try:
                extra["iteration_kernels"] = iteration_kernels(args, dev, mine.start, L, np)
            except Exception as ex:  # noqa: BLE001 — reported, never fatal to the bench line
                extra["iteration_kernels"] = {"error": str(ex)[:400]}
            if proc_scan is not None:
                extra["procs_per_gpu_scan"] = {"model": "K processes ('MPI ranks', tutorials/holstein_honeycomb_mpi.jl:60-72), one single-walker handle each, sharing cuda:0; "
                                                        "started together after a warm-up sweep; K <= 6 (the GPU pool's process guard)", "points": proc_scan}
                try:
                    extra["threads_per_gpu_scan"] = {"model": "K host threads in one process, one single-walker handle (own HIP stream) each",
                                                     "points": threads_per_gpu_scan(args, [int(k) for k in args.thread_scan.split(",") if k], dev, mine.start)}
                except Exception as ex:  # noqa: BLE001 — reported, never fatal to the bench line
                    extra["threads_per_gpu_scan"] = {"error": str(ex)[:400]}
                try:
                    extra["team_threads_scan"] = {"model": "K host threads, each driving ONE walker with the reference's per-walker update sequence, through a walker team "
                                                           "(smoqy_team_*: the members' calls rendezvous in the library and run as one batched call on one handle); "
                                                           "each point says whether the members' hmc_update! ran its trajectory on the device (smoqy_team_hmc_update, the bench's own mode) or host-driven, step by step (--hmc host); "
                                                           "native_threads: the same member sweep from K std::threads inside the library (smoqy_team_bench_sweeps) — "
                                                           "what a caller without an interpreter lock gets",
                                                  "points": team_scan(args, [int(k) for k in args.team_scan.split(",") if k], dev, mine.start)}
                except Exception as ex:  # noqa: BLE001
                    extra["team_threads_scan"] = {"error": str(ex)[:400]}
                try:
                    extra["team_procs_scan"] = {"model": "K member PROCESSES ('MPI ranks', tutorials/holstein_honeycomb_mpi.jl:60-72), each driving ONE walker with the per-walker update "
                                                         "sequence and no GPU access of its own, joined through shared memory (smoqy_team_serve / smoqy_member_*) to one batched handle "
                                                         "per team in this process; HMC mode as in team_threads_scan; compare with procs_per_gpu_scan",
                                                "points": team_procs_scan(args, [q for q in args.team_procs.split(",") if q], dev, mine.start)}
                except Exception as ex:  # noqa: BLE001
                    extra["team_procs_scan"] = {"error": str(ex)[:400]}
        # the CPU baseline is a rank-0, N = 1 measurement (it would only hold the other ranks at the final barrier)
        cpu = None if (args.no_cpu_baseline or args.roofline_only or args.timed_only or world > 1) else cpu_baseline(args.workload, batch.tol, batch.Nt)
        # the whole sweep against the roofline: algorithmic bytes of one preconditioned CG iteration per walker as SURVEY.md §8(d)
        # counts them for the reference's pass structure — MᵀM 2(2S+F), forward and inverse FourierTransformer 2S each, the
        # per-frequency Chebyshev apply 2S, the BLAS-1 lines 10S — times the iterations all walkers ran per second
        V_ = batch.Lt * batch.N
        S_, F_ = 16.0 * V_, 8.0 * V_ + 16.0 * batch.Lt * batch.Nh
        it_bytes = 2 * (2 * S_ + F_) + 2 * S_ + 2 * S_ + 2 * S_ + 10 * S_
        avg_iters = sum(b.stats.iters_sum for b in batches) / max(sum(b.stats.solves for b in batches), 1)
        sweep_gbs = value * batch.solves_per_sweep * avg_iters * it_bytes / 1e9
        extra["sweep_roofline"] = {"algorithmic_bytes_per_cg_iteration_per_walker": it_bytes, "achieved": sweep_gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS * world,
                                   "frac": sweep_gbs / (HBM_PEAK_GBS * world), "physical": sweep_gbs <= HBM_PEAK_GBS * world,
                                   "achieved_is": "algorithmic bytes of the REFERENCE's pass structure (SURVEY.md §8(d): 20S + 2F per CG iteration and walker) x iterations per second; "
                                                  "not a bandwidth the device sustains — the physical reading of the whole step is cg_iteration_traffic",
                                   "note": "CG iterations only (the solves are >85 % of the sweep); the fused kernels move 12 vectors per iteration where this count assumes 20S + 2F"}
        # real bytes: one CG iteration of a 16-system batch moves `it_traffic` bytes beyond L2 (committed PMC pass); iterations per second
        # come from this run
        it_traffic, it_per_kernel, it_src = committed_iteration_traffic(args.workload)
        if it_traffic and args.steps:
            its_per_s = value * batch.solves_per_sweep * avg_iters / 16.0  # 16-system iterations per second over all streams and ranks (the PMC pass was
            gbs = its_per_s * it_traffic / 1e9                            # taken at 16 systems per launch; bytes scale with the systems of a launch)
            copy_gbs = (roofline.get("copy_ceiling") or {}).get("GBs")
            extra["cg_iteration_traffic"] = {"systems_per_launch_in_this_run": per, "bytes_per_16_system_iteration": it_traffic, "per_kernel_bytes": it_per_kernel, "source": it_src, "measured_live": False,
                                             "iterations_per_s_16_systems": its_per_s, "achieved": gbs, "unit": "GB/s", "frac": gbs / (HBM_PEAK_GBS * world),
                                             "frac_of_copy_ceiling": (gbs / (copy_gbs * world)) if copy_gbs else None,
                                             "note": "whole step, everything outside the CG loops counted as zero bytes: measured traffic of the four iteration kernels x the iterations all walkers ran per second"}
        out = {
            "metric": "QMC sweeps/sec (27 preconditioned CG solves per sweep) + FermionDetMatrix matvec GB/s vs HBM roofline, fp64",
            "value": value,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3 if args.steps else None,
            # work-normalised rate: CG iterations (one walker's system advanced by one iteration) per second over all walkers and ranks;
            # cpu_baseline carries the same two fields for the same sweep on the host cores
            "avg_cg_iters": avg_iters,
            "cg_iterations_per_s": value * batch.solves_per_sweep * avg_iters,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "walkers_per_gpu": wpg,
                "streams_per_gpu": S,
                "walkers_total": world * wpg,
                "solves_per_sweep": batch.solves_per_sweep,
                "cg_tol": batch.tol,
                "avg_cg_iters": sum(b.stats.iters_sum for b in batches) / max(sum(b.stats.solves for b in batches), 1),
                "preconditioner": "KPM (Sym)",
                "tfft_kernel": batch.h.describe().get("tfft"),  # what the timed batches' τ-FFT launches ran (the register-blocked forms stand in for either request)
                "tfft_form": ("in-place" if _in_place_tfft_exists(batch.Lt) else "two-image (in-place requested; Ltau has a factor 7)") if batch.tfft_in_place else "two-image",
                "hmc": ("EFA leapfrog on the device, Nt = %d steps of dt = pi/(2 Nt), trajectory always rejected (x restored) so the field distribution stays the one SURVEY.md 8(d) defines" % batch.Nt)
                if args.hmc == "device" else "synthetic host-side drift (round-1 form)",
                "hmc_async": _async_stats(batches, L) if args.hmc == "device" else None,
                "random_numbers": "one PCG64 generator per walker on the host" + ("" if args.no_prefetch else ", drawn one sweep ahead by the batch's host-thread pool while the device runs the current sweep (inside the timed region; same numbers as drawn on demand)"),
                "parallelism": f"walker-parallel, {world} rank(s) x {wpg} walkers ({S} lock-step batches of {per}), no collective",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
