"""Multi-GPU = independent replicas: utterances are embarrassingly parallel, so requests are partitioned
across one-process-per-GPU replicas with NO data-path collective (xGMI idle); torch.distributed is used only for the barrier /
max-over-ranks timing and for gathering results on rank 0.  Determinism contract: the output of an utterance does not depend
on which rank ran it."""
import time


def shard_requests(costs, world):
    """Longest-processing-time-first partition of request indices by expected cost (e.g. text length): returns `world` lists.
    Deterministic (ties broken by index) so every rank computes the same assignment without communicating."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads, shards = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def timed_region(fn, steps, dist=None, sync=None):
    """barrier + sync, K steps, barrier + sync; returns the MAX elapsed seconds over ranks (bench.py contract)."""
    def fence():
        if dist is not None:
            dist.barrier()
        if sync is not None:
            sync()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def run_sharded(requests, costs, worker, rank, world, dist=None):
    """Each rank runs `worker(request)` on its shard; results are gathered (host objects) on rank 0 in request order."""
    mine = shard_requests(costs, world)[rank]
    local = {i: worker(requests[i]) for i in mine}
    if dist is None or world == 1:
        return [local[i] for i in range(len(requests))]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for d in gathered:
        merged.update(d)
    return [merged[i] for i in range(len(requests))]
