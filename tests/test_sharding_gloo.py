"""world_size-2 gloo test of the N>1 path of bench.py: walkers are sharded across ranks with no
overlap and no collective on the data path; the only reduction is MAX over the wall times."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import smoqyelphqmc_amd as sq


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wpg = 3
    mine = list(sq.sharding.walker_range(rank, world, wpg))
    # every rank seeds its walkers from its own ids: gather them only to CHECK disjointness
    ids = [None] * world
    dist.all_gather_object(ids, mine)
    elapsed = 1.0 + rank  # rank 1 is the slow one
    tmax = sq.sharding.reduce_max_time(elapsed)
    if rank == 0:
        out.put((ids, tmax, sq.sharding.aggregate_throughput(wpg * 5, world, tmax)))
    dist.barrier()
    dist.destroy_process_group()


def test_walker_sharding_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ids, tmax, thr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ids == [[0, 1, 2], [3, 4, 5]]
    assert tmax == 2.0
    assert thr == 3 * 5 * 2 / 2.0


def test_single_process_is_identity():
    assert sq.sharding.reduce_max_time(0.25) == 0.25
    assert list(sq.sharding.walker_range(0, 1, 4)) == [0, 1, 2, 3]
    # synthetic walkers with different ids get different phonon fields (independent seeds)
    a = sq.lattice.holstein_honeycomb(2, 4, walker=0).elph.x
    b = sq.lattice.holstein_honeycomb(2, 4, walker=1).elph.x
    assert abs(a - b).max() > 0.1
