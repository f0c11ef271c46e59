"""N > 1 path on CPU: two gloo ranks drive bench.py's measurement contract (barrier, exactly K
steps, max-over-ranks time, whole-job aggregate).  The data path has no collective: every rank
optimises its own replica ('Multi-GPU')."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def run_steps(k):
        calls.append(k)
        time.sleep(0.002 * k * (1 + rank))      # rank 1 is the slow replica

    def all_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    resets = []
    dt = bench.timed_steps(run_steps, lambda: None, dist.barrier, all_max, steps=10, warmup=3,
                           reset=lambda: resets.append(1))
    q.put((rank, dt, calls, len(resets), bench.dist_env()))
    dist.barrier()
    dist.destroy_process_group()


def test_timed_steps_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, dt0, calls0, resets0, env0), (r1, dt1, calls1, resets1, env1) = res
    assert calls0 == [3, 10] and calls1 == [3, 10]          # W untimed, then exactly K timed
    assert resets0 == resets1 == 1
    assert abs(dt0 - dt1) < 1e-9                            # both ranks report the MAX
    assert dt0 >= 0.002 * 10 * 2 * 0.9                      # ... which is the slow rank's time
    assert env0 == (0, 0, 2) and env1 == (1, 1, 2)
    # whole-job value = units of ALL ranks / max time
    assert (2 * 10) / dt0 < 2 * 10 / (0.002 * 10 * 2 * 0.9)
