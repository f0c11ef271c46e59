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


# ---- the sharded Gauss-Newton driver over 2 gloo ranks (fake shards: the numerics are the GPU tests'
# business; this covers the stage / sync / all-reduce protocol of rustrobotics_amd.sharded_gauss_newton)

class _FakeShard:
    """A 'graph' whose chi2 is sum_i (target_i - x_i)^2 over 8 entries, 4 owned by each rank.  Stage 0
    publishes the owned residuals (buffer 0), stage 1 turns the all-reduced residuals into the owned
    part of the step (buffer 1), stage 2 applies the all-reduced step."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.x = torch.zeros(8, dtype=torch.float64)
        self.target = torch.arange(8, dtype=torch.float64)
        self.buf = [torch.zeros(8, dtype=torch.float64), torch.zeros(8, dtype=torch.float64)]
        self.log, self.own = [], slice(4 * rank, 4 * rank + 4)
        self._chi = self._nrm = 0.0

    def stage(self, k):
        self.log.append(("stage", k))
        if k == 0:
            self.buf[0].zero_()
            self.buf[0][self.own] = (self.target - self.x)[self.own]
        elif k == 1:
            self._chi = float((self.buf[0] ** 2).sum())          # needs the REDUCED buffer 0
            self.buf[1].zero_()
            self.buf[1][self.own] = self.buf[0][self.own]
        else:
            self._nrm = float(self.buf[1].norm())                # needs the REDUCED buffer 1
            self.x += self.buf[1]

    def sync(self):
        self.log.append(("sync",))

    def stage_scalars(self):
        return self._chi, self._nrm

    def global_error(self):
        return float(((self.target - self.x) ** 2).sum())


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from rustrobotics_amd.mapping import sharded_gauss_newton
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _FakeShard(rank, world)
    calls = []

    def allreduce(which):
        calls.append(which)
        dist.all_reduce(g.buf[which], op=dist.ReduceOp.SUM)

    errors = sharded_gauss_newton([g], 5, allreduce)
    q.put((rank, errors, calls, g.log[:8], g.x.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_driver_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, e0, c0, log0, x0), (_, e1, c1, log1, x1) = res
    expected0 = float(sum(i * i for i in range(8)))
    assert e0 == e1 == [expected0, 0.0, 0.0]        # one full step solves it; 2nd iteration sees |dx| = 0 and stops
    assert c0 == c1 == [0, 1, 0, 1]                  # two all-reduces per iteration, in order
    assert log0[:6] == [("stage", 0), ("sync",), ("stage", 1), ("sync",), ("stage", 2), ("stage", 0)]
    assert x0 == x1 == [float(i) for i in range(8)]  # both ranks hold the full, identical state
