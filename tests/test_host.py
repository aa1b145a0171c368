"""Host logic: BAM decode, region discovery, synthetic generator invariants, LPT sharding, and the
N > 1 gather path on gloo (world_size 2, CPU)."""
import os

import numpy as np
import pytest

import helpers
from longcallr_amd import _abi, bamio, shard, synth


def test_demo_bam_decode():
    refs, recs = bamio.read_bam(os.path.join(helpers.GOLDEN, "demo.bam"))
    assert len(recs) == 1713 and refs[recs[0]["ref_id"]] == ("chr20", 64444167)
    keep = [r for r in recs if bamio.passes_filter(r)]
    assert len(keep) == 1697
    assert all(r["ts"] == 1 and r["de"] is not None and not (r["flag"] & 16) for r in keep)
    regs = bamio.discover_regions(keep, keep[0]["ref_id"], 64444167)
    assert regs == [(16729960, 13256, 1649)]
    b = helpers.demo_batch()
    ops, lens = b.cigar & 15, b.cigar >> 4
    assert int(lens[np.isin(ops, [0, 7, 8])].sum()) == 2162368  # aligned bases
    assert int(b.seq_len.sum()) == b.bases.size == b.quals.size


@pytest.mark.parametrize("profile", ["ont-cdna", "masseq", "ont-drna"])
def test_synthetic_batches_are_well_formed(profile):
    b = synth.make_batch(profile, n_genes=3, gene_len=7000, depth=20, seed=9)
    assert b.quals.min() >= 1 and set(np.unique(b.bases)) <= set(b"ACGT")
    cig_read = np.repeat(np.arange(b.n_reads), b.n_cig)
    ops, lens = b.cigar & 15, (b.cigar >> 4).astype(np.int64)
    qlen = np.bincount(cig_read, weights=np.where(np.isin(ops, [0, 1, 4, 7, 8]), lens, 0), minlength=b.n_reads)
    assert np.array_equal(qlen.astype(np.int64), b.seq_len.astype(np.int64))
    rlen = np.bincount(cig_read, weights=np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens, 0), minlength=b.n_reads)
    for g in range(b.n_regions):
        s = slice(int(b.read_begin[g]), int(b.read_begin[g + 1]))
        assert np.all(np.diff(b.pos[s]) >= 0)
        assert b.pos[s].min() >= b.start0[g] and (b.pos[s] + rlen[s]).max() <= b.start0[g] + b.len[g]
    depth = lens[np.isin(ops, [0, 7, 8])].sum() / b.col_off[-1]
    assert 0.6 * 20 < depth < 1.6 * 20


def test_lpt_assignment_is_balanced_and_deterministic():
    costs = [9, 7, 6, 5, 5, 4, 3, 1]
    own = shard.assign_regions(costs, 3)
    assert sorted(sum(own, [])) == list(range(8)) and own == shard.assign_regions(costs, 3)
    loads = [sum(costs[i] for i in o) for o in own]
    assert max(loads) - min(loads) <= max(costs)
    assert shard.assign_regions([], 2) == [[], []]


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = np.zeros(3 + 2 * rank, dtype=_abi.CAND_DTYPE)
    rec["pos"] = np.arange(rec.size) + 100 * rank
    rec["region"] = rank
    rec["qual"] = 1.5 + rank
    out = shard.gather_records(rec, dist)
    if rank == 0:
        q.put((out["pos"].tolist(), out["region"].tolist(), out["qual"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_gloo_world2():
    """The only collective of the path: variable-length gather of result records to rank 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pos, region, qual = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert pos == [0, 1, 2, 100, 101, 102, 103, 104] and region == [0] * 3 + [1] * 5
    assert qual == [1.5] * 3 + [2.5] * 5


def _stream_gather_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = shard.RecordGather(dist, torch.device("cpu"), _abi.CAND_DTYPE)
    got, prev = [], None
    for batch in range(3):   # batch i's gather is finished only after batch i+1's has been started
        rec = np.zeros(2 + rank + batch, dtype=_abi.CAND_DTYPE)
        rec["pos"] = 1000 * batch + 100 * rank + np.arange(rec.size)
        h = G.start(rec)
        if prev is not None:
            got.append(G.finish(prev))
        prev = h
    got.append(G.finish(prev))
    if rank == 0:
        q.put([g["pos"].tolist() for g in got])
    dist.barrier()
    dist.destroy_process_group()


def test_record_gather_stream_gloo_world2():
    """RecordGather (bench.py's overlapped gather): fixed capacity agreed once, counts in the header."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_stream_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for batch, pos in enumerate(got):
        want = [1000 * batch + k for k in range(2 + batch)] + [1000 * batch + 100 + k for k in range(3 + batch)]
        assert pos == want


def test_vcf_formatter_matches_oracle_text(orc):
    """longcallr_amd.vcf (product formatter) on oracle candidates == the oracle's own text."""
    from longcallr_amd import vcf
    p = _abi.make_params("hifi-masseq")
    R = orc.Region(helpers.demo_batch(), 0, p).set_tie_mask(orc.TIE_MASK_LIBLCR).run_all(orc.MODE_TIE)
    assert vcf.format_records(R.cands(), "chr20", p.min_phase_score) == R.vcf_text("chr20")
    b = synth.make_batch("ont-drna", n_genes=1, gene_len=12000, depth=30, seed=2)
    p = _abi.make_params("ont-drna")
    R = orc.Region(b, 0, p).set_tie_mask(orc.TIE_MASK_LIBLCR).run_all(orc.MODE_TIE)
    assert vcf.format_records(R.cands(), "c", p.min_phase_score) == R.vcf_text("c")


def test_region_discovery_quirks():
    """util.rs:287-330 restated (oracle_np) vs the host version: single-column islands stay pending."""
    from oracle import oracle_np
    spans = [(5, 6), (10, 20), (12, 18), (30, 31), (40, 41), (50, 60), (70, 71)]
    want = oracle_np.discover_regions(spans, 80)
    # island [5,5] is pending -> region 5..19 (gap 6..9 included), max 2; [30,30] pending, [40,40] -> region 30..40;
    # [50,59] alone; trailing single column [70,70] never emitted
    assert want == [(5, 15, 2), (30, 11, 1), (50, 10, 1)]
    recs = [dict(ref_id=0, pos=s, ref_len=e - s) for s, e in spans]
    assert bamio.discover_regions(recs, 0, 80) == want


def test_bench_shards_partition_one_region_list():
    """bench.py at N > 1: ONE list of distinct genes (synth.make_genes), LPT-partitioned by shard.assign_regions on
    len x max_coverage computed from the genes' read spans alone (synth.gene_costs); a rank builds exactly its regions
    (build_shard), the shards together are the job, and the batch does not depend on the number of generator workers."""
    import bench
    kw = dict(genes=4, gene_len=5000, depth=10.0, seed=5)
    whole, ids, n = bench.build_shard("c4", 1, 0, workers=1, **kw)
    assert ids == [0, 1, 2, 3] and n == 4
    again = bench.build_shard("c4", 1, 0, workers=3, **kw)[0]
    for f in _abi.ReadBatch.FIELDS + ["start0", "len", "read_begin", "ref", "col_off"]:
        assert np.array_equal(getattr(whole, f), getattr(again, f)), f
    # the cost model sees the read spans before poly-A tails: within a tail's length of the region table's numbers
    cost = synth.gene_costs("masseq", range(4), gene_len=5000, depth=10.0, seed=5)
    true = whole.len.astype(np.float64) * bench.region_max_coverage(whole)
    assert np.all(np.abs(cost - true) <= 0.05 * true)
    for world in (2, 3):
        big = bench.build_shard("c4", 1, 0, workers=2, genes=4 * world, gene_len=5000, depth=10.0, seed=5)[0]
        costs = synth.gene_costs("masseq", range(4 * world), gene_len=5000, depth=10.0, seed=5)
        owner = shard.assign_regions(costs, world)
        assert sorted(sum(owner, [])) == list(range(4 * world))
        loads = [float(costs[o].sum()) for o in owner]
        assert max(loads) - min(loads) <= costs.max()          # LPT: no rank is more than one region ahead
        for r in range(world):
            sb, mine, n_global = bench.build_shard("c4", world, r, workers=1, **kw)
            assert mine == owner[r] and n_global == 4 * world and sb.n_regions == len(mine)
            for j, k in enumerate(mine):
                r0, r1, q0, q1 = int(sb.read_begin[j]), int(sb.read_begin[j + 1]), int(big.read_begin[k]), int(big.read_begin[k + 1])
                assert (int(sb.start0[j]), int(sb.len[j])) == (int(big.start0[k]), int(big.len[k]))
                assert np.array_equal(sb.pos[r0:r1], big.pos[q0:q1])
                assert sb.bases[int(sb.seq_off[r0]):int(sb.seq_off[r1 - 1] + sb.seq_len[r1 - 1])].tobytes() == \
                    big.bases[int(big.seq_off[q0]):int(big.seq_off[q1 - 1] + big.seq_len[q1 - 1])].tobytes()

