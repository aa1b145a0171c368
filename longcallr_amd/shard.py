"""Multi-GPU sharding of the hot path.

Regions are independent units (reference: one rayon task per region, src/thread.rs:77), so the
path shards with NO data-path collective: one process per GPU, regions assigned by
longest-processing-time on the cost estimate len x max_coverage (Region.max_coverage, util.rs:28).
The only exchange is the final variable-length gather of result records to rank 0
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np


def assign_regions(costs, world_size):
    """LPT: returns owner[rank] = sorted list of region indices. Deterministic (ties -> lower index)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    owner = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[r].append(i)
        load[r] += costs[i]
    return [sorted(o) for o in owner]


def gather_records(records, dist, device=None, dst=0):
    """Gather a structured numpy array (fixed-size records, e.g. _abi.CAND_DTYPE) from every rank to
    `dst`: all_gather of the counts, then a padded gather of the raw bytes.  Returns the
    concatenation in rank order on dst, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([records.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    item = records.dtype.itemsize
    cap = max(max(counts), 1) * item
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    raw = np.frombuffer(records.tobytes(), dtype=np.uint8)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    if rank == dst:
        parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(buf, gather_list=parts, dst=dst)
        out = [np.frombuffer(p.cpu().numpy().tobytes()[:c * item], dtype=records.dtype) for p, c in zip(parts, counts)]
        return np.concatenate(out) if out else records[:0]
    dist.gather(buf, gather_list=None, dst=dst)
    return None
