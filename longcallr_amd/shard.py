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


class _DevBytes:
    """A raw device allocation as a uint8 vector for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class RecordGather:
    """Overlapped variant of gather_records for a steady stream of batches: fixed-capacity buffers (capacity
    agreed once, from the first batch: 2 x the largest count, every rank computes the same value), the record
    count travels in an 8-byte header, and the gather runs asynchronously (async_op) so that the collective of
    batch i overlaps the kernels of batch i+1.  start() enqueues, finish() waits and (on dst) returns the
    concatenated records.  A later batch that does not fit the agreed capacity raises: call reset() on every
    rank to renegotiate."""

    def __init__(self, dist, device, dtype, dst=0):
        import torch
        self.torch, self.dist, self.dev, self.dtype, self.dst = torch, dist, device, np.dtype(dtype), dst
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.cap = None
        self.slot = 0

    def reset(self):
        self.cap = None

    def _negotiate(self, n):
        t = self.torch.tensor([n], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.cap = max(2 * int(t.item()), 64)
        nbytes = 8 + self.cap * self.dtype.itemsize
        pin = self.dev.type == "cuda"
        self.host = [self.torch.zeros(nbytes, dtype=self.torch.uint8, pin_memory=pin) for _ in range(2)]
        self.send = [self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.dev) for _ in range(2)]
        self.parts = [[self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.dev) for _ in range(self.world)]
                      for _ in range(2)] if self.rank == self.dst else [None, None]

    def start(self, records):
        """records: a structured numpy array (host), or (device pointer, count) of records already in this
        rank's HBM (Engine.candidates_device()): then nothing passes through the host."""
        on_dev = isinstance(records, tuple)
        n = int(records[1]) if on_dev else int(records.size)
        if self.cap is None:
            self._negotiate(n)
        if n > self.cap:
            raise ValueError("RecordGather: %d records exceed the negotiated capacity %d" % (n, self.cap))
        k = self.slot
        self.slot ^= 1
        nb = n * self.dtype.itemsize
        if on_dev:
            self.send[k][:8].copy_(self.torch.from_numpy(np.frombuffer(np.int64(n).tobytes(), dtype=np.uint8).copy()), non_blocking=True)
            if n:
                src = self.torch.as_tensor(_DevBytes(records[0], nb), device=self.dev)
                self.send[k][8:8 + nb].copy_(src, non_blocking=True)
        else:
            h = self.host[k].numpy()
            h[:8] = np.frombuffer(np.int64(n).tobytes(), dtype=np.uint8)
            if n:
                h[8:8 + nb] = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
            self.send[k].copy_(self.host[k], non_blocking=True)
        work = self.dist.gather(self.send[k], gather_list=self.parts[k], dst=self.dst, async_op=True)
        return (work, k)

    def finish(self, handle, parse=True):
        """Wait for the gather; with parse (on dst) copy the parts to the host and return the concatenated
        records, without it they stay in dst's HBM (self.parts) for a consumer that reads them later."""
        work, k = handle
        work.wait()
        if self.rank != self.dst or not parse:
            return None
        out = []
        for p in self.parts[k]:
            raw = p.cpu().numpy()
            n = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
            out.append(np.frombuffer(raw[8:8 + n * self.dtype.itemsize].tobytes(), dtype=self.dtype))
        return np.concatenate(out) if out else np.zeros(0, self.dtype)
