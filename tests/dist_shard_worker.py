"""Worker of tests/test_gpu_parity.py::test_two_ranks_share_one_gpu (launched by torch.distributed.run, backend gloo):
every rank takes its LPT shard of ONE list of distinct MAS-Seq genes (bench.py's construction, build_shard), runs the hot path on cuda:0 and the candidate / read records are gathered to rank 0, which compares them with
a single-process run over all regions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bench
    from longcallr_amd import _abi, api, shard, synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # bench.py's construction at N > 1 (build_shard): ONE list of distinct MAS-Seq genes, LPT on len x max_coverage
    # (LCR_TEST_SHARD_PROFILE=ont-drna: chain regions, so that LCR_GRID_MIN_ENTRIES=0 forces persistent all-CU launches)
    profile = os.environ.get("LCR_TEST_SHARD_PROFILE", "masseq")
    per_rank, kw = 4, dict(gene_len=12000, depth=35.0, seed=77)
    n_global = per_rank * world
    costs = synth.gene_costs(profile, range(n_global), **kw)
    owner = shard.assign_regions(costs, world)
    assert sorted(sum(owner, [])) == list(range(n_global))
    assert bench.build_shard("c4", world, rank, genes=per_rank, workers=1, profile=profile, **kw)[1] == owner[rank]
    params = _abi.make_params(synth.preset_for(profile), seed=31)

    def run(ids):
        b = synth.make_genes(profile, workers=2, gene_ids=ids, **kw)
        E = api.Engine(0, params)
        E.load_batch(b).run_all()
        c, off = E.candidates()
        c = c.copy()
        c["region"] = np.asarray(ids, dtype=np.int32)[c["region"]]        # batch-local -> global region index
        pr, fm = E.phase_result(), E.fragmat()
        reads = np.zeros(fm["row_read"].size, dtype=[("region", "<i4"), ("row", "<i4"), ("hp", "i1"), ("asg", "u1"), ("ps", "<u4")])
        rr = np.repeat(np.arange(len(ids)), np.diff(fm["row_region_off"]))
        reads["region"] = np.asarray(ids, dtype=np.int32)[rr]
        reads["row"] = np.arange(reads.size) - fm["row_region_off"][rr]
        reads["hp"], reads["asg"], reads["ps"] = pr["haplotag"], pr["assignment"], pr["phase_set"]
        E.close()
        return c, reads
    c, reads = run(owner[rank])
    # the gather bench.py uses (asynchronous, fixed-capacity buffers; host records over gloo here, device pointers over RCCL:
    # tests/dist_nccl_worker.py), and once more through the synchronous variable-length form
    G = (shard.RecordGather(dist, torch.device("cpu"), c.dtype), shard.RecordGather(dist, torch.device("cpu"), reads.dtype))
    h = (G[0].start(c), G[1].start(reads))
    gc, gr = G[0].finish(h[0]), G[1].finish(h[1])
    gc2 = shard.gather_records(c, dist)
    if rank == 0:
        assert gc2.tobytes() == gc.tobytes()
    if rank == 0:
        wc, wr = run(list(range(n_global)))
        gc = gc[np.argsort(gc["region"], kind="stable")]
        gr = gr[np.lexsort((gr["row"], gr["region"]))]
        assert gc.tobytes() == wc.tobytes(), "gathered candidate records differ from the single-process run"
        assert gr.tobytes() == wr.tobytes(), "gathered read records differ from the single-process run"
        print("SHARD-OK %d candidates, %d reads, %d regions over %d ranks" % (gc.size, gr.size, n_global, world))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
