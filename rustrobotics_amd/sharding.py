"""ONE pose graph sharded over ranks: the host side of include/rr_pgo.h "sharding".

One Gauss-Newton iteration = stage 0, ALL-GATHER of exchange buffer 0 (boundary update matrices, rank r owns
chunk r), stage 1, sum ALL-REDUCE of exchange buffer 1 (two doubles: chi2 and |dx|^2 partial sums).  The
reference has no counterpart (its only parallel construct is rayon in update_nodes,
pose_graph_optimization.rs:230); the loop control mirrors `optimize` (:247-303).

  gauss_newton(shards, iters, coll)      the loop, over whatever collectives `coll` provides
  EmulatedCollectives                    P ranks emulated in one process on one GPU (tests, projections)
  HostStagedCollectives                  one rank of a torch.distributed group, collectives staged through host memory (gloo):
                                         several PROCESSES that share one GPU
  TorchShardDriver                       one rank of a torch.distributed (RCCL) group: collectives are issued on
                                         the handle's own HIP stream, an iteration has no host synchronisation
"""
import numpy as np

from .mapping import PoseGraph


def gauss_newton(shards, num_iterations, coll, tolerance=1e-4):
    """`PoseGraph::optimize` (Gauss-Newton branch, :247-303) on a sharded graph.

    shards : the handles this process drives -- [own handle] under torch.distributed, or all P handles when
             the ranks are emulated in one process.
    coll   : object with all_gather_boundary() and all_reduce_scalars(), acting on the buffers bound to the shards.
    Returns (errors, norms) with the reference's semantics: 1 + iterations chi2 values, |dx| per iteration."""
    errors, norms = [], []
    for _ in range(num_iterations):
        for g in shards:
            g.stage(0)
        coll.all_gather_boundary()
        for g in shards:
            g.stage(1)
        coll.all_reduce_scalars()
        chi, nrm = shards[0].stage_scalars()
        for g in shards[1:]:
            g.sync()
        errors.append(chi)
        norms.append(nrm)
        if nrm < tolerance:   # :298-300
            break
    errors.append(global_error(shards, coll))
    return errors, norms


def global_error(shards, coll):
    """global_error (:537-574) of a sharded graph: every rank sums the edges it owns, one all-reduce."""
    for g in shards:
        g.stage(2)
    coll.all_reduce_scalars()
    chi, _ = shards[0].stage_scalars()
    for g in shards[1:]:
        g.sync()
    return chi


def gather_state(shards):
    """The full state from P emulated ranks: every node from the rank that owns it (shared nodes from rank 0)."""
    owner = shards[0].node_owner()
    states = [np.asarray(g.state()) for g in shards]
    # state packing follows the node kinds; all 2D graphs here are SE2 (3) / XY (2), SE3 uses 7
    arrays = shards[0].graph_arrays()
    lens = np.array([{0: 3, 1: 2, 2: 7}[int(k)] for k in arrays[0]])
    offs = np.concatenate([[0], np.cumsum(lens)])
    out = states[0].copy()
    for i, o in enumerate(owner):
        if o > 0:
            out[offs[i]:offs[i + 1]] = states[o][offs[i]:offs[i + 1]]
    return out


class EmulatedCollectives:
    """P ranks in one process on one GPU: the collectives are device copies between the ranks' buffers (torch
    tensors bound with rr_pgo_set_exchange_buffer exactly as TorchShardDriver binds them)."""

    def __init__(self, torch, shards):
        self.torch, self.shards = torch, shards
        self.P = len(shards)
        self.xch, self.scal = [], []
        for g in shards:
            _, n, es = g.exchange_info(0)
            t = torch.zeros(n, dtype=torch.float64 if es == 8 else torch.float32, device="cuda")
            g.bind_exchange(0, t.data_ptr(), t.numel())
            self.xch.append(t)
            s = torch.zeros(2, dtype=torch.float64, device="cuda")
            g.bind_exchange(1, s.data_ptr(), 2)
            self.scal.append(s)
        self.chunk = self.xch[0].numel() // self.P
        self.bytes_moved = 0

    def _sync(self):
        # device-wide, like the stream order a real collective relies on; NOT PoseGraph.sync(), which would also
        # report (and clear) a rank's device error flag before the stage that lets the group agree on it
        self.torch.cuda.synchronize()

    def all_gather_boundary(self):
        self._sync()
        c = self.chunk
        for r in range(self.P):
            for q in range(self.P):
                if q != r:
                    self.xch[q][r * c:(r + 1) * c].copy_(self.xch[r][r * c:(r + 1) * c])
        self.torch.cuda.synchronize()
        self.bytes_moved += (self.P - 1) * c * self.xch[0].element_size()

    def all_reduce_scalars(self):
        self._sync()
        tot = self.torch.stack(self.scal).sum(0)
        for s in self.scal:
            s.copy_(tot)
        self.torch.cuda.synchronize()


class HostStagedCollectives:
    """One rank of a torch.distributed group whose collectives go through HOST memory: the rank's chunk of exchange buffer
    0 and the two scalars of buffer 1 are copied to CPU tensors, all-gathered / all-reduced there (gloo) and copied
    back.  For ranks that cannot use RCCL between them -- several processes that share ONE GPU (the closest thing to
    N > 1 a one-GPU box allows: tests/test_gpu_parity.py) -- the protocol, the buffers and every kernel are the ones
    TorchShardDriver drives over RCCL."""

    def __init__(self, torch, dist, shard):
        self.torch, self.dist, self.g = torch, dist, shard
        self.rank, self.P = dist.get_rank(), dist.get_world_size()
        _, n, es = shard.exchange_info(0)
        self.xch = torch.zeros(n, dtype=torch.float64 if es == 8 else torch.float32, device="cuda")
        shard.bind_exchange(0, self.xch.data_ptr(), self.xch.numel())
        self.scal = torch.zeros(2, dtype=torch.float64, device="cuda")
        shard.bind_exchange(1, self.scal.data_ptr(), 2)
        self.chunk = n // self.P
        self.bytes_moved = 0

    def _sync(self):
        self.torch.cuda.synchronize()   # (not PoseGraph.sync(): that would report -- and clear -- the rank's error flag early)

    def all_gather_boundary(self):
        self._sync()
        c, r = self.chunk, self.rank
        mine = self.xch[r * c:(r + 1) * c].cpu()
        parts = [self.torch.empty_like(mine) for _ in range(self.P)]
        self.dist.all_gather(parts, mine)
        for q, t in enumerate(parts):
            if q != r:
                self.xch[q * c:(q + 1) * c].copy_(t)
        self.torch.cuda.synchronize()
        self.bytes_moved += (self.P - 1) * c * self.xch.element_size()

    def all_reduce_scalars(self):
        self._sync()
        t = self.scal.cpu()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.scal.copy_(t)
        self.torch.cuda.synchronize()


class HostStagedShardDriver:
    """One rank of a torch.distributed group WITHOUT RCCL between the ranks (backend gloo): TorchShardDriver's interface over
    HostStagedCollectives.  `bench.py --collectives host-staged` runs its N > 1 plan through this when several ranks share one
    GPU -- the rehearsal of the multi-GPU run a one-GPU box allows: same handles (opt.world_size = N), same stages, same
    buffers, same record; only the transport of the two collectives differs (and synchronises with the host)."""

    def __init__(self, arrays, precision, device, rank, world, dist):
        import torch
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.graph = PoseGraph.from_arrays(*arrays, precision=precision, device=device, rank=rank, world_size=world, sharded=True)
        self.coll = HostStagedCollectives(torch, dist, self.graph)
        self.use_dist = True

    def run_steps(self, k):
        g, c = self.graph, self.coll
        for _ in range(k):
            g.stage(0)
            c.all_gather_boundary()
            g.stage(1)
            c.all_reduce_scalars()

    def optimize(self, num_iterations, tolerance=1e-4):
        return gauss_newton([self.graph], num_iterations, self.coll, tolerance)

    def global_error(self):
        return global_error([self.graph], self.coll)

    def exchange_bytes_per_step(self):
        x = self.coll.xch
        return {"all_gather_bytes_total": int(x.numel() * x.element_size()),
                "all_gather_bytes_contributed_per_rank": int(self.coll.chunk * x.element_size()), "all_reduce_bytes": 16}

    def collectives_description(self):
        return ("per iteration: all-gather of the boundary update matrices + all-reduce(sum) of two doubles, STAGED THROUGH HOST MEMORY "
                "over gloo (ranks that share one GPU: a rehearsal of the protocol, not a transport to time)")


def emulate(arrays, P, precision="f64", device=-1):
    """P sharded handles of the same graph in this process + their emulated collectives."""
    import torch
    shards = [PoseGraph.from_arrays(*arrays, precision=precision, device=device, rank=r, world_size=P, sharded=True)
              for r in range(P)]
    return shards, EmulatedCollectives(torch, shards)


class TorchShardDriver:
    """One rank of a torch.distributed group (backend "nccl" = RCCL over xGMI).

    The exchange buffers are torch tensors bound to the handle; torch's current stream is switched to the
    handle's own HIP stream (torch.cuda.ExternalStream), so the RCCL collectives are ordered after the stage that
    fills their buffer and before the stage that reads it by stream order alone."""

    def __init__(self, arrays, precision, device, rank, world, dist, force_collectives=False):
        import torch
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.graph = PoseGraph.from_arrays(*arrays, precision=precision, device=device, rank=rank, world_size=world,
                                           sharded=True)
        g = self.graph
        _, n, es = g.exchange_info(0)
        self.xch = torch.zeros(n, dtype=torch.float64 if es == 8 else torch.float32, device="cuda")
        g.bind_exchange(0, self.xch.data_ptr(), n)
        self.scal = torch.zeros(2, dtype=torch.float64, device="cuda")
        g.bind_exchange(1, self.scal.data_ptr(), 2)
        self.chunk = n // world
        self.mine = self.xch[rank * self.chunk:(rank + 1) * self.chunk]
        self.stream = torch.cuda.ExternalStream(g.stream_ptr())
        self.use_dist = dist is not None and (world > 1 or force_collectives)
        torch.cuda.synchronize()

    # the two collectives, on the handle's stream
    def all_gather_boundary(self):
        if self.use_dist:
            self.dist.all_gather_into_tensor(self.xch, self.mine)

    def all_reduce_scalars(self):
        if self.use_dist:
            self.dist.all_reduce(self.scal, op=self.dist.ReduceOp.SUM)

    def run_steps(self, k):
        """k Gauss-Newton iterations back to back: no convergence break, no host synchronisation."""
        g = self.graph
        with self.torch.cuda.stream(self.stream):
            for _ in range(k):
                g.stage(0)
                self.all_gather_boundary()
                g.stage(1)
                self.all_reduce_scalars()

    def optimize(self, num_iterations, tolerance=1e-4):
        with self.torch.cuda.stream(self.stream):
            return gauss_newton([self.graph], num_iterations, self, tolerance)

    def global_error(self):
        with self.torch.cuda.stream(self.stream):
            return global_error([self.graph], self)

    def exchange_bytes_per_step(self):
        return {"all_gather_bytes_total": int(self.xch.numel() * self.xch.element_size()),
                "all_gather_bytes_contributed_per_rank": int(self.chunk * self.xch.element_size()),
                "all_reduce_bytes": 16}

    def collectives_description(self):
        return ("per iteration: all_gather_into_tensor (boundary update matrices, in place) + all_reduce(sum) of two doubles, "
                + ("RCCL through torch.distributed, issued on the library's own stream" if self.use_dist else "skipped (single rank, no group)"))
