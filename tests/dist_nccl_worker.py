"""Worker of tests/test_gpu_parity.py::test_nccl_group_gathers_device_records (launched by torch.distributed.run, backend nccl =
RCCL): the branch an N > 1 bench run takes -- shard.RecordGather over device-pointer records (Engine.candidates_device(),
Engine.read_records_device(): nothing passes through the host), asynchronous gathers of three batches overlapped with the
next batch's kernels -- compared byte for byte with lcr_get_candidates / lcr_get_phase_result.  World size 1 on the one GPU
of the test box covers every line of it (thread.rs:204-221 is what the gather replaces)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from longcallr_amd import _abi, api, shard, synth
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    assert dist.get_backend() == "nccl"
    params = _abi.make_params("ont-cdna", seed=31)
    E = api.Engine(local, params)
    E.set_stream(torch.cuda.current_stream().cuda_stream)
    G = (shard.RecordGather(dist, dev, _abi.CAND_DTYPE), shard.RecordGather(dist, dev, _abi.READ_REC_DTYPE))
    # three overlapped batches inside the capacity negotiated on the first (2 x its count), then a batch beyond it: RecordGather.start
    # refuses it (every rank would have to agree on new buffers), reset() on every rank renegotiates
    batches = [synth.make_batch("ont-cdna", n_genes=n, gene_len=9000, depth=30, seed=40 + k + 10 * rank) for k, n in enumerate((4, 5, 2, 12))]
    want, pending, got = [], None, []
    for k, b in enumerate(batches):
        E.load_batch(b).run_all()
        c = E.candidates()[0].copy()
        pr = E.phase_result()
        r = np.zeros(pr["haplotag"].size, dtype=_abi.READ_REC_DTYPE)
        r["row"], r["haplotag"], r["assignment"], r["phase_set"] = np.arange(r.size), pr["haplotag"], pr["assignment"], pr["phase_set"]
        want.append((c, r))
        if k == 3:
            got.append((G[0].finish(pending[0]), G[1].finish(pending[1])))
            pending = None
            try:
                G[1].start(E.read_records_device())
                raise AssertionError("a batch beyond the negotiated capacity must be refused")
            except ValueError:
                pass
            G[0].reset(); G[1].reset()
        h = (G[0].start(E.candidates_device()), G[1].start(E.read_records_device()))   # gathers of batch i run under the kernels of batch i + 1
        if pending is not None:
            got.append((G[0].finish(pending[0]), G[1].finish(pending[1])))
        pending = h
    got.append((G[0].finish(pending[0]), G[1].finish(pending[1])))
    if rank == 0:
        assert world == 1
        for k, ((wc, wr), (gc, gr)) in enumerate(zip(want, got)):
            assert gc.size == wc.size and gc.tobytes() == wc.tobytes(), "batch %d: gathered candidate records differ" % k
            assert gr.size == wr.size and gr.tobytes() == wr.tobytes(), "batch %d: gathered read records differ" % k
        print("NCCL-GATHER-OK %s candidates, %s reads over %d rank(s)" % ([c.size for c, _ in want], [r.size for _, r in want], world))
    E.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
