#!/usr/bin/env python
"""bench.py — one "step" = one pass of the hot path (bind batch -> pileup -> candidates/GT ->
fragment matrix -> phasing) over one synthetic batch already resident in HBM.

Workload (config.workload): BASELINE.json configs[2], "synthetic 10 Mb ONT-cDNA, 40x" (C3) per GPU —
the largest single-GPU configuration; demo.bam (configs[0]/[1]) is ~5 MB of traffic and is a parity
fixture, not a bench line.  Weak scaling: every rank gets its own C3-sized region set; regions never
span ranks, the only collective is the final gather of candidate records to rank 0 (RCCL).

Prints ONE JSON line on rank 0 (see the driver contract): value = candidate sites (pileup columns
evaluated) per second, whole job, over the full step time; roofline = the pileup kernel's
algorithmic bytes / its HIP-event time; cpu_baseline = the CPU oracle (a C++ restatement of the
reference, NOT the Rust binary) timed on a bounded sample of the same workload, rank 0 at N = 1 only.
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
    ap.add_argument("--profile", default="ont-cdna")
    ap.add_argument("--genes", type=int, default=400)
    ap.add_argument("--unique-genes", type=int, default=50)
    ap.add_argument("--gene-len", type=int, default=25000)
    ap.add_argument("--depth", type=float, default=40.0)
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

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    copies = max(1, a.genes // a.unique_genes)
    base = synth.make_batch(a.profile, n_genes=a.unique_genes, gene_len=a.gene_len, depth=a.depth, seed=1000)  # every rank owns an identically distributed (same seed) C3-sized region set: weak scaling with equal work
    batch = tile_batch(base, copies)
    params = _abi.make_params(synth.preset_for(a.profile), seed=2025)
    reads, regions, keep = to_device(batch, torch, dev)
    torch.cuda.synchronize()

    F = max(1, a.inflight)
    engines = [api.Engine(local, params, timing=True) for _ in range(F)]
    E = engines[0]
    if F == 1:
        E.set_stream(torch.cuda.current_stream().cuda_stream)  # torch.cuda.synchronize() then covers liblcr

    G = shard.RecordGather(dist, dev, _abi.CAND_DTYPE) if dist is not None else None
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
        h = G.start(Ej.candidates_device())
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
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
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

    if rank == 0:
        cols = int(batch.col_off[-1])
        pbytes = E.pileup_bytes()
        avg_ms = float(np.mean([t[0] for t in pile_ms]))
        avg_k0_ms = float(np.mean([t[1] for t in pile_ms]))
        achieved = pbytes / (avg_ms * 1e-3) / 1e9
        stage_bytes = E.pileup_stage_bytes()
        traffic = None  # HBM bytes per K1 launch from the committed PMC passes (same workload only)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "k1_traffic.json")))
            w = tj["workload"]
            if (w["profile"], w["genes"], w["unique_genes"], w["gene_len"], w["depth"]) == (
                    a.profile, a.genes, a.unique_genes, a.gene_len, a.depth):
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "candidate_sites_per_sec", "value": cols * world * a.steps / dt, "unit": "sites/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 in, u32 counts, f64 likelihoods, i64 fixed-point phase scores", "data": "synthetic",
            "config": {"workload": "%s: synthetic %s reads, %d regions x %d bp, %.0fx mean aligned depth per GPU "
                                   "(%d unique genes tiled x%d), preset %s; step = bind+pileup+candidates+fragments+phase"
                                   % ("C3" if a.profile == "ont-cdna" else "C3-shaped", a.profile, batch.n_regions, a.gene_len, a.depth,
                                      a.unique_genes, copies, synth.preset_for(a.profile)),
                       "columns_per_gpu": cols, "aligned_bases_per_gpu": int(batch.bases.size), "reads_per_gpu": batch.n_reads,
                       "candidates_per_gpu": int(cands.size), "fragment_nnz_per_gpu": int(fm["col"].size),
                       "parallelism": "regions sharded over %d GPU(s), gather to rank 0" % world,
                       "batches_in_flight_per_gpu": F},
            "roofline": {"bound": "hbm", "kernel": "k1_pileup", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "algorithmic_bytes": pbytes, "avg_ms": avg_ms,
                         "note": "k1_pileup (+ k1_zonefix on HiFi presets): read bases once + 8-byte records + 57 B/column",
                         "pileup_stage": {"kernels": "k0_bin + intron scan + k1_pileup", "ms": avg_ms + avg_k0_ms,
                                          "algorithmic_bytes": stage_bytes,
                                          "achieved": stage_bytes / ((avg_ms + avg_k0_ms) * 1e-3) / 1e9,
                                          "frac": stage_bytes / ((avg_ms + avg_k0_ms) * 1e-3) / 1e9 / 8000.0}},
            "stages": {"pileup_plus_candidates_s": t_call, "fragments_plus_phase_s": t_phase,
                       "sites_per_sec_pileup_gt": cols / t_call, "phased_reads_per_sec": n_phased / t_phase,
                       "api_ms": api_ms, "kernel_ms": kms},
        }
        if not a.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the other ranks of a node would sit idle behind it
            out["cpu_baseline"] = cpu_baseline(batch, params, a.cpu_budget)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
