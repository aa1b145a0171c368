#!/usr/bin/env python
"""bench.py — one "step" = one pass of the hot path (bind batch -> pileup -> candidates/GT ->
fragment matrix -> phasing) over one synthetic batch already resident in HBM.

Workloads (config.workload):
  N = 1   BASELINE.json configs[2], "synthetic 10 Mb ONT-cDNA, 40x" (C3: 400 regions x 25 kb) -- the largest configuration
          whose metric is quoted on one GPU; demo.bam (configs[0]/[1]) is ~5 MB of traffic and a parity fixture, not a
          bench line.  The C5 stress (configs[4], ONE 1 Mb island at 500x) is run once beside it and reported in
          `stages.c5`, and the default workload once more with three batches in flight as `stages.batches_in_flight`
          (--no-c5 skips both).
  N > 1   BASELINE.json configs[3], "synthetic 200 Mb PacBio MAS-Seq, 60x, region-sharded across 8 GPUs" scaled to
          N GPUs: ONE list of N x 1 000 regions (25 Mb x 60x per GPU), partitioned over the ranks by
          shard.assign_regions (longest-processing-time on len x max_coverage, the reference's unit of sharding is the
          region, thread.rs:77); a rank materialises and processes only its own regions.  Weak scaling.
`python bench.py --gpus N` starts the N ranks itself (re-exec under torch.distributed.run) when it is not already
running under a launcher.  Regions never span ranks, the only collective is the final gather of candidate records to
rank 0 (RCCL), overlapped with the next batch's kernels.

Prints ONE JSON line on rank 0 (see the driver contract): value = candidate sites (pileup columns evaluated) per
second, whole job, over the full step time; roofline = the pileup stage's algorithmic bytes / its HIP-event time
(k1_pileup alone beside it); cpu_baseline = the CPU oracle (a C++ restatement of the reference, NOT the Rust binary)
timed on a bounded sample of the same workload, rank 0 at N = 1 only.
--inflight N: N contexts on N host threads keep N batches in flight per GPU (default 1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def tile_batch(base, copies, gap=1000):
    """Replicate a ReadBatch `copies` times at shifted coordinates (synthetic-data generation speed)."""
    from longcallr_amd import _abi
    if copies == 1:
        return base
    span = int(base.start0[-1] + base.len[-1] - base.start0[0]) + gap
    rep = lambda a: np.concatenate([a] * copies)
    shift_r = np.repeat(np.arange(copies, dtype=np.int64) * span, base.n_reads)
    shift_g = np.repeat(np.arange(copies, dtype=np.int64) * span, base.n_regions)
    nb, nc = int(base.bases.size), int(base.cigar.size)
    rb = np.concatenate([base.read_begin[:-1] + k * base.n_reads for k in range(copies)] + [[copies * base.n_reads]])
    return _abi.ReadBatch(
        pos=(rep(base.pos).astype(np.int64) + shift_r), seq_len=rep(base.seq_len), lead_clip=rep(base.lead_clip),
        trail_clip=rep(base.trail_clip), flags=rep(base.flags),
        seq_off=rep(base.seq_off) + np.repeat(np.arange(copies, dtype=np.uint64) * np.uint64(nb), base.n_reads),
        cig_off=rep(base.cig_off) + np.repeat(np.arange(copies, dtype=np.uint64) * np.uint64(nc), base.n_reads),
        n_cig=rep(base.n_cig), bases=rep(base.bases), quals=rep(base.quals), cigar=rep(base.cigar),
        start0=rep(base.start0) + shift_g, len=rep(base.len), read_begin=rb, ref=rep(base.ref))


def region_max_coverage(b):
    """Region.max_coverage (util.rs:28, 281-285: every reference position of a read span counts) per region."""
    ops, lens = b.cigar & 15, (b.cigar >> 4).astype(np.int64)
    cig_read = np.repeat(np.arange(b.n_reads), b.n_cig)
    ref_len = np.bincount(cig_read, weights=np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens, 0), minlength=b.n_reads).astype(np.int64)
    out = np.zeros(b.n_regions, dtype=np.int64)
    for g in range(b.n_regions):
        r0, r1 = int(b.read_begin[g]), int(b.read_begin[g + 1])
        d = np.zeros(int(b.len[g]) + 2, dtype=np.int64)
        s = np.clip(b.pos[r0:r1].astype(np.int64) - int(b.start0[g]), 0, int(b.len[g]))
        e = np.clip(s + ref_len[r0:r1], 0, int(b.len[g]) + 1)
        np.add.at(d, s, 1); np.add.at(d, e, -1)
        out[g] = int(np.cumsum(d).max()) if r1 > r0 else 0
    return out


def subset_batch(base, ids, gap=1000):
    """The regions `ids` (ascending) of the global list whose region k is unique region k % U of `base` at copy k // U
    (copies lie `span` apart): a rank's shard, built without materialising the other ranks' regions."""
    from longcallr_amd import _abi
    U = base.n_regions
    span = int(base.start0[-1] + base.len[-1] - base.start0[0]) + gap
    parts = {k: [] for k in ("pos", "seq_len", "lead_clip", "trail_clip", "flags", "n_cig", "bases", "quals", "cigar", "ref")}
    start0, length, read_begin = [], [], [0]
    base_off, cig_off = base.seq_off.astype(np.int64), base.cig_off.astype(np.int64)
    for k in ids:
        u, c = int(k) % U, int(k) // U
        r0, r1 = int(base.read_begin[u]), int(base.read_begin[u + 1])
        b0 = int(base_off[r0]) if r1 > r0 else 0
        b1 = int(base_off[r1 - 1] + base.seq_len[r1 - 1]) if r1 > r0 else 0
        c0 = int(cig_off[r0]) if r1 > r0 else 0
        c1 = int(cig_off[r1 - 1] + base.n_cig[r1 - 1]) if r1 > r0 else 0
        parts["pos"].append(base.pos[r0:r1].astype(np.int64) + c * span)
        for f in ("seq_len", "lead_clip", "trail_clip", "flags", "n_cig"):
            parts[f].append(getattr(base, f)[r0:r1])
        parts["bases"].append(base.bases[b0:b1]); parts["quals"].append(base.quals[b0:b1]); parts["cigar"].append(base.cigar[c0:c1])
        o = int(base.col_off[u])
        parts["ref"].append(base.ref[o:o + int(base.len[u])])
        start0.append(int(base.start0[u]) + c * span); length.append(int(base.len[u]))
        read_begin.append(read_begin[-1] + (r1 - r0))
    cat = lambda f, dt: np.concatenate(parts[f]).astype(dt) if parts[f] else np.zeros(0, dt)
    seq_len, n_cig = cat("seq_len", np.int64), cat("n_cig", np.int64)
    return _abi.ReadBatch(pos=cat("pos", np.int64), seq_len=seq_len, lead_clip=cat("lead_clip", np.int32),
                          trail_clip=cat("trail_clip", np.int32), flags=cat("flags", np.uint8),
                          seq_off=(np.cumsum(seq_len) - seq_len).astype(np.uint64), cig_off=(np.cumsum(n_cig) - n_cig).astype(np.uint64),
                          n_cig=n_cig, bases=cat("bases", np.uint8), quals=cat("quals", np.uint8), cigar=cat("cigar", np.uint32),
                          start0=start0, len=length, read_begin=read_begin, ref=cat("ref", np.uint8))


WORKLOADS = {   # name -> (profile, regions per GPU, gene_len, depth, BASELINE config it stands for)
    "c3": ("ont-cdna", 400, 25000, 40.0, "C3 = BASELINE configs[2]: synthetic 10 Mb ONT-cDNA, 40x"),
    "c4": ("masseq", 1000, 25000, 60.0, "C4 = BASELINE configs[3]: synthetic 200 Mb PacBio MAS-Seq, 60x, region-sharded (25 Mb per GPU)"),
}


def build_shard(name, world=1, rank=0, seed=1, genes=None, gene_len=None, depth=None, profile=None, workers=0):
    """Rank `rank`'s regions of workload `name` on `world` GPUs.  The job is ONE list of world x (regions per GPU)
    DISTINCT genes (synth.make_genes: gene k has its own generator); it is partitioned by
    shard.assign_regions -- LPT on len x max_coverage, computed for the whole list from the genes' read spans without
    building them (synth.gene_costs) -- and a rank builds only its own genes.  Returns (batch, ids, n_global)."""
    from longcallr_amd import shard, synth
    w_profile, w_genes, w_len, w_depth, _ = WORKLOADS[name]
    profile, genes, gene_len, depth = profile or w_profile, genes or w_genes, gene_len or w_len, depth or w_depth
    n_global = world * genes
    if world == 1:
        mine = list(range(n_global))
    else:
        costs = synth.gene_costs(profile, range(n_global), gene_len=gene_len, depth=depth, seed=seed)
        mine = shard.assign_regions(costs, world)[rank]
    if workers <= 0:   # the ranks of one node generate side by side
        workers = max(1, min(64, (os.cpu_count() or 1) // max(1, world)))
    batch = synth.make_genes(profile, gene_len=gene_len, depth=depth, seed=seed, workers=workers, gene_ids=mine)
    return batch, mine, n_global


def build_workload(name, seed=1, workers=0):
    """The single-GPU form of a workload: "c3" = the 400 genes of BASELINE configs[2], "c4" = one GPU's 1 000 genes of
    configs[3] (bench.py --workload c4; tests/test_gpu_parity.py compares both with the oracle at this size)."""
    return build_shard(name, 1, 0, seed=seed, workers=workers)[0]


def batches_in_flight_stage(api, torch, device, params, dev_batch, cols, contexts=3, steps=60):
    """The same step with several batches in flight on the GPU (one context and one host thread each): the queue gaps of
    one batch's host round trips are filled by the others' kernels.  Reported beside `value`, never as it: the co-scheduled
    kernels stretch each other, so the roofline is quoted on the undisturbed single-batch run."""
    import threading
    engines = [api.Engine(device, params) for _ in range(contexts)]
    def run(E, n):
        torch.cuda.set_device(device)
        for _ in range(n):
            E.load_batch(dev_batch)
            E.fill_data_into_freq_vec().get_candidate_snps().get_fragments().phase()
        E.sync()
    for E in engines:
        run(E, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(E, steps // contexts)) for E in engines]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for E in engines:
        E.close()
    n = contexts * (steps // contexts)
    return {"contexts": contexts, "steps": n, "ms_per_step": dt / n * 1e3, "sites_per_sec": cols * n / dt,
            "note": "%d contexts on %d host threads share the GPU (bench.py --inflight %d times the whole run this way)" % (contexts, contexts, contexts)}


def c5_stage(api, _abi, synth, device):
    """BASELINE configs[4] once: ONE 1 Mb island at 500x ONT-dRNA, ~4 700 candidate sites; per-call wall times (ms)."""
    t0 = time.perf_counter()
    b = synth.make_island("ont-drna-c5", n_loci=40, locus_len=25000, depth=500, seed=5)
    gen = time.perf_counter() - t0
    E = api.Engine(device, _abi.make_params("ont-drna", seed=5))
    ms = {}
    for rep in range(2):   # the second pass has its buffers
        for name, fn in (("lcr_load_batch", lambda: E.load_batch(b)), ("lcr_pileup", E.fill_data_into_freq_vec),
                         ("lcr_candidates", E.get_candidate_snps), ("lcr_fragments", E.get_fragments), ("lcr_phase", E.phase)):
            ts = time.perf_counter(); fn(); E.sync(); ms[name] = (time.perf_counter() - ts) * 1e3
    c = E.candidates()[0]
    fm = E.fragmat()
    n_phased = int(fm["row_for_phasing"].sum())
    t_phase = (ms["lcr_fragments"] + ms["lcr_phase"]) * 1e-3
    out = dict(workload="C5 = BASELINE configs[4]: one island of %d columns, %d reads, %d aligned bases" % (int(b.len[0]), b.n_reads, int(b.bases.size)),
               generate_s=gen, api_ms=ms, candidates=int(c.size), fragment_nnz=int(fm["col"].size), phasing_reads=n_phased,
               cross_optimize_calls=1 + 2 * (int(c.size) // 4 + 1), phased_reads_per_sec=n_phased / t_phase,
               sites_per_sec_full_step=int(b.len[0]) / (sum(ms.values()) * 1e-3),
               note="phase stage: k4_stage_grid + k4_chain_grid + k4_gpost (all CUs on the one region, no host epilogue)")
    E.close()
    return out


def to_device(batch, torch, dev):
    from longcallr_amd import _abi
    t = {}
    for f in batch.FIELDS + ["start0", "len", "col_off", "read_begin", "ref"]:
        a = getattr(batch, f)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        t[f] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    reads, regions = batch.c_reads(), batch.c_regions()
    reads.mem = regions.mem = _abi.LCR_MEM_DEVICE
    for f in batch.FIELDS:
        setattr(reads, f, C.c_void_p(t[f].data_ptr()))
    for f in ["start0", "len", "col_off", "read_begin", "ref"]:
        setattr(regions, f, C.c_void_p(t[f].data_ptr()))
    return reads, regions, t


def cpu_baseline(batch, params, budget_s=15.0):
    """Oracle (kind 'port': C++ restatement of the reference's scalar loops, reference-order f64 arithmetic with libm
    per observation) region-parallel over the host's cores, the analogue of the reference's rayon par_iter over regions
    (thread.rs:77): worker threads pull region indices of the same batch for ~budget_s of wall time (ctypes releases
    the GIL inside the oracle).  A 1-thread figure over the first regions is reported next to it."""
    from concurrent.futures import ThreadPoolExecutor
    import itertools
    import threading
    from oracle import orc
    orc.build()

    def run_region(g):
        R = orc.Region(batch, g, params)
        R.run_all(orc.MODE_F64)
        return int(batch.len[g]), int(batch.read_begin[g + 1] - batch.read_begin[g])

    t0 = time.perf_counter()
    cols1 = g1 = 0
    while g1 < batch.n_regions and (time.perf_counter() - t0 < budget_s / 4 or g1 < 2):
        cols1 += run_region(g1)[0]
        g1 += 1
    single = cols1 / (time.perf_counter() - t0)

    threads = max(1, min(os.cpu_count() or 1, 64))
    counter = itertools.count()
    lock = threading.Lock()
    done = []
    t0 = time.perf_counter()

    def worker():
        while time.perf_counter() - t0 < budget_s:
            with lock:
                k = next(counter)
            done.append(run_region(k % batch.n_regions))   # the batch is re-walked if the budget outlasts it

    with ThreadPoolExecutor(max_workers=threads) as ex:
        for f in [ex.submit(worker) for _ in range(threads)]:
            f.result()
    dt = time.perf_counter() - t0
    cols, reads = sum(d[0] for d in done), sum(d[1] for d in done)
    return dict(value=cols / dt, unit="candidate_sites/s", cores=threads, kind="port", single_thread_value=single,
                sample="%d region passes over the %d regions of the rank-0 batch (%d columns, %d reads) by %d threads, "
                       "full hot path, %.1f s; 1 thread: first %d regions"
                       % (len(done), batch.n_regions, cols, reads, threads, dt, g1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="auto", choices=["auto", "c3", "c4"],
                    help="auto: C3 at N = 1, the region-sharded C4 at N > 1")
    ap.add_argument("--profile", default=None, help="read profile of longcallr_amd.synth (default: the workload's)")
    ap.add_argument("--genes", type=int, default=None, help="regions per GPU (default: the workload's)")
    ap.add_argument("--seed", type=int, default=1, help="generator seed of the synthetic genes (SURVEY §8(d): seeds 1..5)")
    ap.add_argument("--gene-len", type=int, default=None)
    ap.add_argument("--depth", type=float, default=None)
    ap.add_argument("--no-c5", action="store_true", help="skip the single pass over the C5 island (N = 1)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: smoke-test the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs, records travel "
                         "through host memory); the numbers of such a run mean nothing")
    ap.add_argument("--prewarm", type=int, default=40,
                    help="untimed passes during setup, before the W warm-up steps: allocations, thread pool, GPU clocks")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU: N contexts driven by N host threads (a context per worker thread, as "
                         "the reference's rayon workers would hold); 1 = one batch at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    a = ap.parse_args()

    import torch
    from longcallr_amd import _abi, api, synth, shard

    if a.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start the N ranks ourselves, one per GPU over RCCL
        have = torch.cuda.device_count()
        if have < a.gpus and a.dist_backend == "nccl":
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (a.gpus, have))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP GPU (no CPU fallback)")
    if a.dist_backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if a.dist_backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    if a.gpus != world:
        raise SystemExit("bench.py --gpus %d under a launcher with WORLD_SIZE=%d" % (a.gpus, world))
    wl = a.workload if a.workload != "auto" else ("c3" if world == 1 else "c4")
    w_profile, w_genes, w_len, w_depth, w_name = WORKLOADS[wl]
    a.profile = a.profile or w_profile
    a.genes = a.genes or w_genes
    a.gene_len = a.gene_len or w_len
    a.depth = a.depth or w_depth
    # ONE region list for the whole job: world x genes distinct genes, LPT on len x max_coverage assigns them to the
    # ranks; a rank builds only its own regions (build_shard)
    batch, mine, n_global = build_shard(wl, world, rank, seed=a.seed, genes=a.genes, gene_len=a.gene_len, depth=a.depth, profile=a.profile)
    assert batch.n_regions == len(mine)
    params = _abi.make_params(synth.preset_for(a.profile), seed=2025)
    reads, regions, keep = to_device(batch, torch, dev)
    torch.cuda.synchronize()

    F = max(1, a.inflight)
    engines = [api.Engine(local, params, timing=True) for _ in range(F)]
    E = engines[0]
    if F == 1:
        E.set_stream(torch.cuda.current_stream().cuda_stream)  # torch.cuda.synchronize() then covers liblcr

    G = shard.RecordGather(dist, cdev, _abi.CAND_DTYPE) if dist is not None else None
    pending = [None]

    def step(Ej):
        Ej.load_batch((reads, regions, keep))
        Ej.fill_data_into_freq_vec()
        Ej.get_candidate_snps().get_fragments().phase()
        # HIP events on the ctx stream, read after the step (lcr_pileup returns while K1 is still running)
        return (Ej.kernel_ms(_abi.K_PILEUP), Ej.kernel_ms(_abi.K_SPANS))

    def publish(Ej):   # the gather of this batch's records (HBM to rank 0's HBM) overlaps the next batch's kernels
        if G is None:
            return
        h = G.start(Ej.candidates_device() if a.dist_backend == "nccl" else Ej.candidates()[0])
        if pending[0] is not None:
            G.finish(pending[0], parse=False)
        pending[0] = h

    def drain():   # the last batch is also brought to rank 0's host and decoded
        if G is not None and pending[0] is not None:
            G.finish(pending[0], parse=True)
            pending[0] = None

    def run_steps(n):
        """n passes; with F > 1 batches in flight, thread j drives context j over passes j, j + F, ... and the
        gathers are issued in pass order on every rank (collectives must line up across ranks)."""
        piles = [None] * n
        if F == 1:
            for k in range(n):
                piles[k] = step(E)
                publish(E)
            return piles
        import threading
        turn = threading.Condition()
        nxt = [0]
        errs = []

        def worker(j):
            try:
                torch.cuda.set_device(local)
                for k in range(j, n, F):
                    piles[k] = step(engines[j])
                    if G is not None:
                        with turn:
                            while nxt[0] < k:
                                turn.wait()
                            if nxt[0] != k:
                                raise RuntimeError("another worker failed")
                        publish(engines[j])
                        torch.cuda.current_stream().synchronize()   # the records left the context's buffer
                        with turn:
                            nxt[0] = k + 1
                            turn.notify_all()
            except BaseException as e:   # noqa: BLE001 -- re-raised by the caller
                errs.append(e)
                with turn:
                    nxt[0] = 1 << 60
                    turn.notify_all()

        ths = [threading.Thread(target=worker, args=(j,)) for j in range(F)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        return piles

    def sync_all():
        for Ej in engines:
            Ej.sync()
        torch.cuda.synchronize()

    run_steps(a.prewarm + a.warmup)
    drain()
    if dist is not None:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    pile_ms = run_steps(a.steps)
    drain()   # the last batch's records are on rank 0 before the clock stops
    sync_all()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # stage breakdown (untimed extra pass: wall clock per ABI call with a sync after each, + HIP events)
    api_ms = {}
    def timed(name, fn):
        ts = time.perf_counter(); fn(); E.sync(); api_ms[name] = (time.perf_counter() - ts) * 1e3
    timed("lcr_load_batch", lambda: E.load_batch((reads, regions, keep)))
    timed("lcr_pileup", E.fill_data_into_freq_vec)
    timed("lcr_candidates", E.get_candidate_snps)
    timed("lcr_fragments", E.get_fragments)
    timed("lcr_phase", E.phase)
    t_call = (api_ms["lcr_load_batch"] + api_ms["lcr_pileup"] + api_ms["lcr_candidates"]) * 1e-3
    t_phase = (api_ms["lcr_fragments"] + api_ms["lcr_phase"]) * 1e-3
    fm = E.fragmat()
    n_phased = int(fm["row_for_phasing"].sum())
    cands = E.candidates()[0]
    kms = {n: E.kernel_ms(k) for n, k in (("k0_bin", _abi.K_SPANS), ("k1_pileup", _abi.K_PILEUP),
                                          ("k2_filter", _abi.K_CAND_FILTER), ("k2_hist", _abi.K_CAND_HIST),
                                          ("k2_gt", _abi.K_CAND_GT), ("k3_count", _abi.K_FRAG_COUNT),
                                          ("k3_fill", _abi.K_FRAG_FILL))}

    # whole-job totals (the shards of an LPT partition differ slightly in size)
    tot = np.array([int(batch.col_off[-1]), int(batch.bases.size), batch.n_reads, int(cands.size), int(fm["col"].size), n_phased], dtype=np.int64)
    if dist is not None:
        tt = torch.from_numpy(tot.copy()).to(cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        tot = tt.cpu().numpy()
    if rank == 0:
        cols = int(batch.col_off[-1])
        # covered sites (depth >= min_depth on A+C+G+T, candidate.rs:90-94) of this rank's batch
        covered = int((E.columns()[:4].sum(axis=0) >= params.min_depth).sum())
        pbytes = E.pileup_bytes()
        avg_ms = float(np.mean([t[0] for t in pile_ms]))
        avg_k0_ms = float(np.mean([t[1] for t in pile_ms]))
        achieved = pbytes / (avg_ms * 1e-3) / 1e9
        stage_bytes = E.pileup_stage_bytes()
        stage_ms = avg_ms + avg_k0_ms
        traffic = None  # HBM bytes per launch from the committed PMC passes (same workload only; not a same-run measurement)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "k1_traffic.json")))
            w = tj["workload"]
            if (w["profile"], w["genes"], w["gene_len"], w["depth"], w.get("seed")) == (a.profile, a.genes, a.gene_len, a.depth, a.seed):
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        nnz = int(fm["col"].size)
        out = {
            "metric": "candidate_sites_per_sec", "value": int(tot[0]) * a.steps / dt, "unit": "sites/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 in, u32 counts, f64 likelihoods, i64 fixed-point phase scores", "data": "synthetic",
            "config": {"workload": "%s; synthetic %s reads, %d distinct genes (regions) x %d bp per GPU at %.0fx mean aligned depth, "
                                   "generator seed %d; %d regions in the job, LPT-partitioned over %d rank(s), preset %s; "
                                   "step = bind+pileup+candidates+fragments+phase"
                                   % (w_name, a.profile, a.genes, a.gene_len, a.depth, a.seed, n_global, world, synth.preset_for(a.profile)),
                       "columns": int(tot[0]), "aligned_bases": int(tot[1]), "reads": int(tot[2]), "candidates": int(tot[3]),
                       "fragment_nnz": int(tot[4]), "phasing_reads": int(tot[5]),
                       "columns_rank0": cols, "aligned_bases_rank0": int(batch.bases.size),
                       "parallelism": "regions sharded over %d GPU(s) by shard.assign_regions (LPT on len x max_coverage), gather of records to rank 0" % world,
                       "batches_in_flight_per_gpu": F},
            "roofline": {"bound": "hbm", "kernel": "pileup stage = k0_bin + intron scan + k1_tile_order + k1_pileup (+ k1_zonefix on HiFi presets): what replaces fill_data_into_freq_vec",
                         "achieved": stage_bytes / (stage_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": stage_bytes / (stage_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic,
                         "algorithmic_bytes": stage_bytes, "avg_ms": stage_ms,
                         "note": "algorithmic bytes B + 4C + 37R + 53L (bases once, CIGAR, read headers, ref byte + 13 u32 planes per column); "
                                 "HIP events on the ctx stream, rank 0; traffic = committed PMC passes of the same workload (profiles/), not this run",
                         "k1_pileup": {"algorithmic_bytes": pbytes, "avg_ms": avg_ms, "achieved": achieved, "frac": achieved / 8000.0,
                                       "note": "the tally kernel with its tile-ordering pass (k1_tile_order + k1_pileup): bases once + 8-byte records + 57 B/column"}},
            "stages": {"pileup_plus_candidates_s": t_call, "fragments_plus_phase_s": t_phase,
                       "sites_per_sec_pileup_gt": cols / t_call, "covered_sites_per_sec_pileup_gt": covered / t_call,
                       "candidates_per_sec_pileup_gt": int(cands.size) / t_call, "phased_reads_per_sec": n_phased / t_phase,
                       "covered_sites_rank0": covered,
                       "k3_fragment_bytes": {"algorithmic_bytes": 4 * int(batch.cigar.size) + 64 * batch.n_reads + 15 * nnz,
                                             "ms": kms["k3_count"] + kms["k3_fill"],
                                             "achieved_GBps": (4 * int(batch.cigar.size) + 64 * batch.n_reads + 15 * nnz) / ((kms["k3_count"] + kms["k3_fill"]) * 1e-3 + 1e-12) / 1e9},
                       "k4_phase": {"ms": api_ms["lcr_phase"], "nnz": nnz, "phasing_reads": n_phased,
                                    "note": "per-kernel times: profiles/ (rocprofv3 --kernel-trace --stats of this command)"},
                       "api_ms": api_ms, "kernel_ms": kms},
        }
        if world == 1 and F == 1 and not a.no_c5:
            out["stages"]["batches_in_flight"] = batches_in_flight_stage(api, torch, local, params, (reads, regions, keep), cols)
        if world == 1 and not a.no_c5:
            out["stages"]["c5"] = c5_stage(api, _abi, synth, local)
        if not a.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the other ranks of a node would sit idle behind it
            out["cpu_baseline"] = cpu_baseline(batch, params, a.cpu_budget)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
