#!/usr/bin/env python
"""bench.py — one "step" = one pass of the hot path (bind batch -> pileup -> candidates/GT ->
fragment matrix -> phasing) over one synthetic batch already resident in HBM.

Workloads (config.workload):
  N = 1   BASELINE.json configs[2], "synthetic 10 Mb ONT-cDNA, 40x" (C3: 400 distinct genes x 25 kb) -- the
          largest configuration whose metric is quoted on one GPU; demo.bam (configs[0]/[1]) is ~5 MB of traffic and a
          parity fixture.  Beside it, in `stages` (--quick / --no-extras skip them): the same step with three batches in
          flight, generator seeds 2..5, one GPU's share of C4 (the workload N > 1 runs), the C5 stress (configs[4], ONE
          1 Mb island at 500x), demo.bam from the file, and C3 end to end from a BAM file -- each with the CPU oracle timed
          beside it.
  N > 1   BASELINE.json configs[3], "synthetic 200 Mb PacBio MAS-Seq, 60x, region-sharded across 8 GPUs" scaled to
          N GPUs: ONE list of N x 1 000 distinct genes (25 Mb x 60x per GPU), partitioned over the ranks by
          shard.assign_regions (longest-processing-time on len x max_coverage, the reference's unit of sharding is the
          region, thread.rs:77); a rank builds and processes only its own regions.  Weak scaling.
`python bench.py --gpus N` starts the N ranks itself (re-exec under torch.distributed.run) when it is not already
running under a launcher.  Regions never span ranks, the only collective is the final gather of the two record types
(candidate records, read -> HP / PS records; thread.rs:204-221) to rank 0 over RCCL, overlapped with the next batch's
kernels.

Prints ONE JSON line on rank 0 (see the driver contract): value = candidate sites (pileup columns evaluated) per
second, whole job, over the full step time; roofline = the pileup stage's algorithmic bytes / its HIP-event time
(k1 alone beside it), roofline.traffic = HBM bytes of the stage's kernels from two rocprofv3 --pmc passes run by this
script over the same batch; cpu_baseline = the CPU oracle (a C++ restatement of the reference, NOT the Rust binary) on a
native thread pool over the same batch, rank 0 at N = 1 only.
--inflight N: N contexts on N host threads keep N batches in flight per GPU (default 1).
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

# The asynchronous phase stage (lcr_ctx_set_async_phase, include/lcr.h) uses four queues; ROCm's default of 4 hardware queues per process
# maps two of them onto one (measured: no gain then).  Read by the HIP runtime when it starts, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {   # name -> (profile, regions per GPU, gene_len, depth, BASELINE config it stands for)
    "c3": ("ont-cdna", 400, 25000, 40.0, "C3 = BASELINE configs[2]: synthetic 10 Mb ONT-cDNA, 40x"),
    "c4": ("masseq", 1000, 25000, 60.0, "C4 = BASELINE configs[3]: synthetic 200 Mb PacBio MAS-Seq, 60x, region-sharded (25 Mb per GPU)"),
}
PILEUP_KERNELS = ("k0_ops", "k1_tiles_a", "k1_tiles_b", "k0_desc_bin", "k1_pileup", "k1_empty_tiles", "k1_zonefix")


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


bench_costs = None


def build_shard(name, world=1, rank=0, seed=1, genes=None, gene_len=None, depth=None, profile=None, workers=0):
    """Rank `rank`'s regions of workload `name` on `world` GPUs.  The job is ONE list of world x (regions per GPU)
    DISTINCT genes (synth.make_genes: gene k has its own generator); it is partitioned by
    shard.assign_regions -- LPT on len x max_coverage, computed for the whole list from the genes' read spans without
    building them (synth.gene_costs) -- and a rank builds only its own genes.  Returns (batch, ids, n_global)."""
    from longcallr_amd import shard, synth
    w_profile, w_genes, w_len, w_depth, _ = WORKLOADS[name]
    profile, genes, gene_len, depth = profile or w_profile, genes or w_genes, gene_len or w_len, depth or w_depth
    n_global = world * genes
    global bench_costs
    if world == 1:
        mine = list(range(n_global))
        bench_costs = None
    else:
        costs = synth.gene_costs(profile, range(n_global), gene_len=gene_len, depth=depth, seed=seed)
        mine = shard.assign_regions(costs, world)[rank]
        bench_costs = np.asarray(costs, dtype=np.float64)   # (the LPT costs of the whole job: the N > 1 line reports every rank's share)
    if workers <= 0:   # the ranks of one node generate side by side
        workers = max(1, min(64, (os.cpu_count() or 1) // max(1, world)))
    batch = synth.make_genes(profile, gene_len=gene_len, depth=depth, seed=seed, workers=workers, gene_ids=mine)
    return batch, mine, n_global


def build_workload(name, seed=1, workers=0):
    """The single-GPU form of a workload: "c3" = the 400 genes of BASELINE configs[2], "c4" = one GPU's 1 000 genes of
    configs[3] (bench.py --workload c4; tests/test_gpu_parity.py compares both with the oracle at this size)."""
    return build_shard(name, 1, 0, seed=seed, workers=workers)[0]


def head_batch(b, n_regions):
    """The first n_regions regions of a batch (bounded CPU samples)."""
    from longcallr_amd import _abi
    g = min(n_regions, b.n_regions)
    r = int(b.read_begin[g])
    nb = int(b.seq_off[r - 1] + b.seq_len[r - 1]) if r else 0
    nc = int(b.cig_off[r - 1] + b.n_cig[r - 1]) if r else 0
    return _abi.ReadBatch(pos=b.pos[:r], seq_len=b.seq_len[:r], lead_clip=b.lead_clip[:r], trail_clip=b.trail_clip[:r], flags=b.flags[:r],
                          seq_off=b.seq_off[:r], cig_off=b.cig_off[:r], n_cig=b.n_cig[:r], bases=b.bases[:nb], quals=b.quals[:nb],
                          cigar=b.cigar[:nc], start0=b.start0[:g], len=b.len[:g], read_begin=b.read_begin[:g + 1], ref=b.ref[:int(b.col_off[g])])


def to_device(batch, torch, dev):
    from longcallr_amd import _abi
    t = {}
    for f in batch.FIELDS + ["start0", "len", "col_off", "read_begin", "ref"]:
        a = getattr(batch, f)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        # (through a page-locked copy: the runtime's path for pageable sources -- it pins the caller's pages chunk by chunk -- raised a
        # device memory fault in one test-suite run out of five on this stack; liblcr stages its own host uploads for the same reason)
        t[f] = torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev)
    reads, regions = batch.c_reads(), batch.c_regions()
    reads.mem = regions.mem = _abi.LCR_MEM_DEVICE
    for f in batch.FIELDS:
        setattr(reads, f, C.c_void_p(t[f].data_ptr()))
    for f in ["start0", "len", "col_off", "read_begin", "ref"]:
        setattr(regions, f, C.c_void_p(t[f].data_ptr()))
    return reads, regions, t


PILE_TIMERS = ("K_BIND_TABLE", "K_BIND", "K_SPANS", "K_PILEUP")   # the kernel groups of the pileup stage (bind kernels included: VERDICT r05)


def consume(res, sink):
    """What a caller does with a batch's results (thread.rs:204-221 pushes them onto the output queues): here the records are read --
    candidates, phased reads -- and counted, so that every timed step delivers its batch's results to the host."""
    sink[0] += int(res["cand"].size)
    sink[1] += int(np.count_nonzero(res["assignment"]))
    sink[2] += 1


def pipelined_step(E, dev_batch, sink, have):
    """One step of a long-lived worker: bind batch k, queue its pileup, THEN collect batch k - 1's results (lcr_collect_phase: with the
    asynchronous phase stage this is where the stage in flight is waited for -- batch k's pileup is already queued beside its tails),
    then candidates / fragments / phase of batch k.  With the synchronous stage the same calls in the same order cost nothing extra."""
    E.load_batch(dev_batch)
    E.fill_data_into_freq_vec()
    if have[0]:
        consume(E.collect_phase(), sink)
    E.get_candidate_snps().get_fragments().phase()
    have[0] = True


def run_steps_simple(E, dev_batch, n, sink=None, have=None, stamps=None):
    sink = sink if sink is not None else [0, 0, 0]
    have = have if have is not None else [False]
    for _ in range(n):
        pipelined_step(E, dev_batch, sink, have)
        if stamps is not None:
            stamps.append(time.perf_counter())
    if have[0]:
        consume(E.collect_phase(), sink); have[0] = False
    E.sync()
    return sink


def step_stats(stamps, t0):
    """p50 / p99 / max of the host-side duration of the timed steps (stamps: perf_counter after every step's last call)"""
    d = np.diff(np.array([t0] + list(stamps))) * 1e3
    return {"p50": float(np.percentile(d, 50)), "p99": float(np.percentile(d, 99)), "max": float(d.max()), "max_over_median": float(d.max() / np.median(d)), "steps": int(d.size)}


def pile_ms_of(E, _abi):
    return sum(E.kernel_ms(getattr(_abi, k)) for k in PILE_TIMERS)


ASYNC_PHASE = [True]   # (--sync-phase clears it: the workloads beside the headline one run the way the headline steps do)


def side_engine(api, _abi, device, params):
    """The context of a side stage: a long-lived worker's (thread.rs:77-143 keeps one per rayon thread)"""
    E = api.Engine(device, params, timing=tuple(getattr(_abi, k) for k in PILE_TIMERS))   # (events around the pileup stage only: every timer is two records on the stream)
    return E


def time_workload(api, _abi, torch, device, params, batch, steps=20, warm=5, E=None):
    """ms per step (every step's results collected and read: pipelined_step), per-step p50 / p99 / max, pileup-stage time and roofline
    fraction, per-call wall times of one more pass: a workload beside the headline one.  E: a context to reuse (one long-lived worker
    across workloads, as a caller would hold it; profiles/r06_stall.txt: a context created right behind a fresh upload can lose 65-85 ms
    once in its first steps)."""
    dv = to_device(batch, torch, torch.device("cuda", device))
    own = E is None
    if own:
        E = side_engine(api, _abi, device, params)
    E.set_async_phase(ASYNC_PHASE[0])
    run_steps_simple(E, dv, warm)
    torch.cuda.synchronize()
    sink, have, stamps, pile = [0, 0, 0], [False], [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        pipelined_step(E, dv, sink, have)
        pile.append(pile_ms_of(E, _abi))
        stamps.append(time.perf_counter())
    consume(E.collect_phase(), sink)
    E.sync()
    dt = (time.perf_counter() - t0) / steps
    assert sink[2] == steps
    E.set_async_phase(False)
    ms = {}
    for name, fn in (("lcr_load_batch", lambda: E.load_batch(dv)), ("lcr_pileup", E.fill_data_into_freq_vec),
                     ("lcr_candidates", E.get_candidate_snps), ("lcr_fragments", E.get_fragments), ("lcr_phase", E.phase)):
        ts = time.perf_counter(); fn(); E.sync(); ms[name] = (time.perf_counter() - ts) * 1e3
    fm = E.fragmat()
    out = dict(columns=int(batch.col_off[-1]), reads=batch.n_reads, aligned_bases=int(batch.bases.size), ms_per_step=dt * 1e3,
               sites_per_sec=int(batch.col_off[-1]) / dt, pileup_stage_ms=float(np.mean(pile)),
               pileup_stage_frac_of_hbm_peak=E.pileup_stage_bytes() / (float(np.mean(pile)) * 1e-3) / 8e12, api_ms=ms,
               candidates=int(E.candidates()[0].size), phasing_reads=int(fm["row_for_phasing"].sum()), fragment_nnz=int(fm["col"].size))
    out["phased_reads_per_sec"] = out["phasing_reads"] / ((ms["lcr_fragments"] + ms["lcr_phase"]) * 1e-3)
    out["step_ms"] = step_stats(stamps, t0)
    out["results_collected_per_step"] = {"candidates": sink[0] // steps, "assigned_reads": sink[1] // steps}
    if own:
        E.close()
    del dv
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CPU side: the oracle on a native thread pool (orc_run_batch), decision arithmetic of the reference only (ORC_MODE_F64_ONLY)

def cpu_pool(batch, params, threads=0, upto="post"):
    from oracle import orc
    B = orc.Batch(batch, params, mode=orc.MODE_F64_ONLY, threads=threads, upto=upto, keep_planes=False)
    dt, th = B.seconds, B.threads
    B.close()
    return dt, th


def cpu_baseline(batch, params, budget_s=20.0):
    """Oracle (kind 'port': C++ restatement of the reference's scalar loops, reference-order f64 arithmetic with libm per
    observation, no second arithmetic beside it) region-parallel on a NATIVE thread pool -- the analogue of the reference's
    rayon par_iter over regions (thread.rs:77) -- over the regions of the same batch; the thread-scaling points beside it
    run on the first regions of the batch so that everything stays inside ~budget_s of CPU wall time."""
    from oracle import orc
    orc.build()
    ncpu = os.cpu_count() or 1
    cols = int(batch.col_off[-1])
    # every region of the batch once per pool size: all hardware threads, the physical cores, 64 -- the stated baseline is the
    # BEST of them (round 4 quoted the all-threads run although 64 threads were 27 % faster: SMT + two sockets do not pay here)
    full = {}
    spent = 0.0
    for th in sorted({ncpu, max(1, ncpu // 2), min(64, ncpu)}, reverse=True):
        if full and spent > 0.7 * budget_s:
            continue
        dt, th_used = cpu_pool(batch, params, th if th < ncpu else 0)
        full[th_used] = dt
        spent += dt
    th_best = min(full, key=lambda k: full[k])
    t_all, th_all = full[th_best], th_best
    out = dict(value=cols / t_all, unit="candidate_sites/s", cores=th_all, kind="port",
               sample="all %d regions of the rank-0 batch (%d columns, %d reads) once through orc_run_batch (native std::thread pool, "
                      "heaviest regions first), ORC_MODE_F64_ONLY = the reference's arithmetic, full hot path P1-P17, at %s threads: %s s; "
                      "value = the best point (%d threads)"
                      % (batch.n_regions, cols, batch.n_reads, sorted(full), ["%.2f" % full[k] for k in sorted(full)], th_all))
    # thread scaling below that on the first regions of the batch (a fixed number per point: ~2-6 s of wall time each)
    scaling = {}
    for th, n_reg in ((1, 12), (16, 96)):
        if th >= ncpu or spent > budget_s:
            continue
        hb = head_batch(batch, n_reg)
        dt, _ = cpu_pool(hb, params, th)
        spent += dt
        scaling[str(th)] = dict(sites_per_sec=int(hb.col_off[-1]) / dt, regions=hb.n_regions, seconds=dt)
    for k in sorted(full):
        scaling[str(k)] = dict(sites_per_sec=cols / full[k], regions=batch.n_regions, seconds=full[k])
    out["thread_scaling"] = scaling
    if "1" in scaling:
        out["single_thread_value"] = scaling["1"]["sites_per_sec"]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# same-run HBM traffic of the pileup stage: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over a
# child process that loads the SAME batch and runs the pileup stage a few times

def pmc_child(path):
    import torch
    from longcallr_amd import _abi, api
    z = np.load(path)
    b = _abi.ReadBatch(**{k: z[k] for k in z.files if k not in ("preset",)})
    p = _abi.make_params(str(z["preset"]), seed=2025)
    dv = to_device(b, torch, torch.device("cuda", 0))
    E = api.Engine(0, p)
    for _ in range(4):
        E.load_batch(dv)
        E.fill_data_into_freq_vec()
        E.sync()
    E.close()


def measure_traffic(batch, preset):
    """HBM bytes per launch of the pileup stage's kernels, or (None, reason).  FETCH_SIZE is doubled (gfx950 tallies a 128-byte
    request of a wide coalesced read as 64, MI355X_MICROARCH.md "HBM"); WRITE_SIZE as reported; both are in KiB."""
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    import csv, glob
    tmp = tempfile.mkdtemp(prefix="lcr_pmc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        path = os.path.join(tmp, "batch.npz")
        np.savez(path, preset=np.array(preset), **{f: getattr(batch, f) for f in batch.FIELDS + ["start0", "len", "read_begin", "ref"]})
        per = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR=tmp)
            r = subprocess.run([rp, "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "p", "--output-format", "csv", "--",
                                sys.executable, os.path.abspath(__file__), "--pmc-child", path], cwd=tmp, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (ctr, r.returncode, r.stdout.decode(errors="replace")[-300:])
            agg = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != ctr:
                    continue
                k = next((n for n in PILEUP_KERNELS if n in row["Kernel_Name"]), None)
                if k:
                    a = agg.setdefault(k, [0.0, set()])
                    a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
            per[ctr] = {k: v[0] / max(len(v[1]), 1) for k, v in agg.items()}
        kernels = {}
        total = 0.0
        for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
            f, w = per["FETCH_SIZE"].get(k, 0.0), per["WRITE_SIZE"].get(k, 0.0)
            kernels[k] = dict(fetch_kib=f, write_kib=w, bytes=int((2.0 * f + w) * 1024))
            total += (2.0 * f + w) * 1024
        return dict(bytes_per_launch=int(total), kernels=kernels,
                    note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes of 4 pileup calls over this run's batch; FETCH x 2 per "
                         "MI355X_MICROARCH.md (calibrated for 16 B/lane streaming reads only), WRITE as reported"), None
    except Exception as e:   # noqa: BLE001 -- the bench line must not depend on the profiler
        return None, "traffic pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------
# stages beside the headline workload (rank 0, N = 1)

def batches_in_flight_stage(api, torch, device, params, dev_batch, cols, contexts=3, steps=60):
    """The same step with several batches in flight on the GPU (one context and one host thread each): the queue gaps of
    one batch's host round trips are filled by the others' kernels.  Reported beside `value`, never as it: the co-scheduled
    kernels stretch each other, so the roofline is quoted on the undisturbed single-batch run."""
    import threading
    engines = [api.Engine(device, params) for _ in range(contexts)]
    def run(E, n):
        torch.cuda.set_device(device)
        run_steps_simple(E, dev_batch, n)
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


def seeds_stage(api, _abi, torch, device, wl, params, seeds, steps=15, E=None):
    """generator seeds 1..5 per config.  Seed 1 is the headline run; the others: ms per step and pileup fraction.
    E: the headline run's context (ONE long-lived worker for all seeds, thread.rs:77-143)."""
    out = {}
    own = E is None
    if own:
        E = side_engine(api, _abi, device, params)
    for s in seeds:
        b = build_workload(wl, seed=s)
        r = time_workload(api, _abi, torch, device, params, b, steps=steps, warm=4, E=E)   # (one pass, reported as it comes)
        out[str(s)] = {k: r[k] for k in ("columns", "aligned_bases", "ms_per_step", "sites_per_sec", "pileup_stage_ms", "pileup_stage_frac_of_hbm_peak", "candidates", "step_ms")}
    if own:
        E.close()
    return out


def c4_share_stage(api, _abi, synth, torch, device, cpu=True, E=None):
    """One GPU's share of BASELINE configs[3] (the workload N > 1 runs: 1 000 distinct MAS-Seq genes x 25 kb, 60x) at N = 1, and the
    CPU oracle pool on its first regions."""
    t0 = time.perf_counter()
    b = build_workload("c4")
    gen = time.perf_counter() - t0
    p = _abi.make_params("hifi-masseq", seed=2025)
    if E is not None:   # (the long-lived context of the run, with this workload's preset)
        keep_params, E.params = E.params, p
    out = time_workload(api, _abi, torch, device, p, b, steps=20, warm=5, E=E)
    if E is not None:
        E.params = keep_params
    out["workload"], out["generate_s"] = WORKLOADS["c4"][4] + ": 1 000 distinct genes, hifi-masseq preset", gen
    if cpu:
        hb = head_batch(b, 256)
        dt, th = cpu_pool(hb, p, 0)
        out["cpu_oracle"] = dict(sites_per_sec=int(hb.col_off[-1]) / dt, threads=th, seconds=dt, kind="port",
                                 sample="first %d regions (%d columns), full hot path, ORC_MODE_F64_ONLY" % (hb.n_regions, int(hb.col_off[-1])))
        out["gpu_over_cpu_oracle"] = out["sites_per_sec"] / out["cpu_oracle"]["sites_per_sec"]
    return out


def c5_stage(api, _abi, synth, torch, device, cpu=True):
    """BASELINE configs[4] once: ONE 1 Mb island at 500x ONT-dRNA, ~4 700 candidate sites; per-call wall times (ms).  CPU: the oracle's
    P1-P6 (pileup, candidates, fragment matrix) of the one region on ONE thread -- the reference's unit of parallelism is the region, and
    its optimiser on this matrix is out of reach (phase.rs:890-898 is quadratic in a column's depth)."""
    t0 = time.perf_counter()
    b = synth.make_island("ont-drna-c5", n_loci=40, locus_len=25000, depth=500, seed=5)
    gen = time.perf_counter() - t0
    p = _abi.make_params("ont-drna", seed=5)
    E = api.Engine(device, p)
    ms = {}
    dv = to_device(b, torch, torch.device("cuda", device))   # inputs resident in HBM, as in the headline step
    for rep in range(2):   # the second pass has its buffers
        for name, fn in (("lcr_load_batch", lambda: E.load_batch(dv)), ("lcr_pileup", E.fill_data_into_freq_vec),
                         ("lcr_candidates", E.get_candidate_snps), ("lcr_fragments", E.get_fragments), ("lcr_phase", E.phase)):
            ts = time.perf_counter(); fn(); E.sync(); ms[name] = (time.perf_counter() - ts) * 1e3
    c = E.candidates()[0]
    fm = E.fragmat()
    n_phased = int(fm["row_for_phasing"].sum())
    t_phase = (ms["lcr_fragments"] + ms["lcr_phase"]) * 1e-3
    t_p16 = (ms["lcr_load_batch"] + ms["lcr_pileup"] + ms["lcr_candidates"] + ms["lcr_fragments"]) * 1e-3
    out = dict(workload="C5 = BASELINE configs[4]: one island of %d columns, %d reads, %d aligned bases" % (int(b.len[0]), b.n_reads, int(b.bases.size)),
               generate_s=gen, api_ms=ms, candidates=int(c.size), fragment_nnz=int(fm["col"].size), phasing_reads=n_phased,
               cross_optimize_calls=1 + 2 * (int(c.size) // 4 + 1), phased_reads_per_sec=n_phased / t_phase,
               sites_per_sec_full_step=int(b.len[0]) / (sum(ms.values()) * 1e-3), sites_per_sec_p1_p6=int(b.len[0]) / t_p16,
               note="phase stage: k4_stage_grid + k4_chain_grid + k4_gpost (all CUs on the one region, no host epilogue); inputs resident in HBM "
                    "(from host buffers lcr_load_batch adds 22 ms of PCIe)")
    E.close()
    del dv
    if cpu:
        dt, th = cpu_pool(b, p, 1, upto="frag")
        out["cpu_oracle_p1_p6"] = dict(sites_per_sec=int(b.len[0]) / dt, threads=th, seconds=dt, kind="port",
                                       sample="the whole island through P1-P6 on one thread (one region = one rayon task)")
        out["gpu_over_cpu_oracle_p1_p6"] = out["sites_per_sec_p1_p6"] / out["cpu_oracle_p1_p6"]["sites_per_sec"]
    return out


def host_fed_stage(api, device, batch, params, cols, steps=8):
    """The headline step for a caller whose reads are decoded on the HOST (the reference's loop): (a) lcr_load_batch of
    pageable host arrays, stages behind it; (b) the asynchronous input path -- page-locked arrays, lcr_load_batch_async of batch i + 1
    into the other staging slot while batch i's stages run, lcr_bind_batch.  1.06 GB cross PCIe per step either way (never part of `value`)."""
    E = api.Engine(device, params)
    out = {}
    try:
        E.load_batch(batch).run_all(); E.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            E.load_batch(batch).run_all()
        E.sync()
        out["sync_pageable_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
        pinned = api.host_register(batch.bases, batch.quals, batch.cigar, batch.ref, batch.pos, batch.seq_len, batch.seq_off, batch.cig_off, batch.n_cig)
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                E.load_batch(batch).run_all()
            E.sync()
            out["sync_pinned_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
            E.load_batch_async(batch, 0)
            E.bind_batch(0); E.load_batch_async(batch, 1); E.run_all()   # (warm: the slots' buffers are allocated)
            E.bind_batch(1); E.load_batch_async(batch, 0); E.run_all()
            t0 = time.perf_counter()
            for k in range(steps):
                E.bind_batch(k & 1)
                E.load_batch_async(batch, (k + 1) & 1)
                E.run_all()
            E.sync()
            out["async_pinned_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
        finally:
            E.close()
            api.host_unregister(pinned)
    finally:
        E.close()
    nbytes = int(batch.bases.nbytes + batch.quals.nbytes + batch.cigar.nbytes + batch.ref.nbytes) + 36 * batch.n_reads
    out["bytes_over_pcie_per_step"] = nbytes
    out["async_pcie_GBps"] = nbytes / (out["async_pinned_ms_per_step"] * 1e-3) / 1e9
    out["sites_per_sec_async"] = cols / (out["async_pinned_ms_per_step"] * 1e-3)
    out["note"] = ("host-fed steps are PCIe-bound: %.2f GB per step at >= 17 ms on a Gen5 x16 link against ~2 ms of kernels; the asynchronous path hides "
                   "the kernels under the upload of the next batch, it cannot go below the link time" % (nbytes / 1e9))
    return out


def demo_stage(api, _abi, device, cpu=True):
    """BASELINE configs[0]/[1]: demo.bam from the file -- native decode, region discovery on the GPU, batch, the four stage calls -- wall time
    per pass, against the oracle's compute on the same decoded reads (the oracle has no decoder of its own: its side is compute only)."""
    from longcallr_amd import bamio
    path = os.path.join(ROOT, "tests", "golden", "demo.bam")
    lines = open(os.path.join(ROOT, "tests", "golden", "demo_pseudo_ref.fa")).read().split("\n")
    ref = np.frombuffer("".join(lines[1:]).encode(), dtype=np.uint8).copy()
    p = _abi.make_params("hifi-masseq")
    E = api.Engine(device, p)

    n_thr = max(1, min(8, os.cpu_count() or 1))   # (a 1.2 MB file: more inflate threads than BGZF blocks buy nothing)
    parts = {}

    def one_pass(acc=None):
        tk = [time.perf_counter()]
        def lap(name):
            tk.append(time.perf_counter())
            if acc is not None:
                acc[name] = acc.get(name, 0.0) + tk[-1] - tk[-2]
        nb = bamio.NativeBam(path, n_thr); lap("open_inflate_index")
        rid = [n for n, _ in nb.refs].index("chr20")
        rs, re_ = nb.spans(rid, **_abi.READ_FILTER); lap("spans")
        regions = E.discover_regions(rs, re_, nb.refs[rid][1]); lap("discover_regions")
        b = nb.batch(rid, [(s, l) for s, l, _ in regions], [ref], name_format="blob", **_abi.READ_FILTER); lap("batch")
        E.load_batch(b).run_all()
        c = E.candidates()[0]; lap("load_and_four_stage_calls")
        nb.close(); lap("close")
        return b, c
    for _ in range(3):
        b, c = one_pass()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        b, c = one_pass(parts)
    gpu = (time.perf_counter() - t0) / n
    # the four stage calls alone, inputs already decoded (host buffers, upload included)
    for _ in range(20):
        E.load_batch(b).run_all()
    E.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        E.load_batch(b).run_all()
    E.sync()
    stages_only = (time.perf_counter() - t0) / 50
    E.close()
    L = int(b.len[0])
    out = dict(workload="demo.bam (chr20:16 729 961-16 743 217, %d reads pass the filter, %d columns, %d candidates), hifi-masseq preset, pseudo-reference"
                        % (b.n_reads, L, int(c.size)),
               file_to_candidates_ms=gpu * 1e3, file_to_candidates_parts_ms={k: v / n * 1e3 for k, v in parts.items()}, decode_threads=n_thr,
               stage_calls_ms=stages_only * 1e3, sites_per_sec_from_file=L / gpu, sites_per_sec_stage_calls=L / stages_only,
               note="from the file = lcr_bam_open (inflate on `decode_threads` threads) + spans + lcr_discover_regions (dense passes over the covered window of the contig only) + lcr_bam_batch + load_batch + four stage "
                    "calls + lcr_get_candidates, per pass; ~5 MB of traffic: launch- and latency-bound, not a roofline workload")
    if cpu:
        m = 3
        cpu_t = sum(cpu_pool(b, p, 1)[0] for _ in range(m)) / m
        out["cpu_oracle"] = dict(sites_per_sec=L / cpu_t, ms=cpu_t * 1e3, threads=1, kind="port",
                                 sample="the one region on one thread (the reference runs one rayon task per region), compute only, ORC_MODE_F64_ONLY")
        out["gpu_over_cpu_oracle_stage_calls"] = cpu_t / stages_only
        out["gpu_from_file_over_cpu_oracle_compute"] = cpu_t / gpu
    return out


def end_to_end_stage(api, _abi, torch, device, batch, params, cpu_compute_s=None, cpu_threads=None):
    """The headline batch once more, END TO END FROM A BAM FILE: the batch is written as a BAM (lcr_bam_write_reads, not timed), then
    per pass: lcr_bam_open (inflate + index on all host threads) -> spans -> lcr_discover_regions -> lcr_bam_batch -> lcr_load_batch (host
    buffers: 1 GB over PCIe) -> four stage calls -> candidates on the host.  CPU side: the same native decode + the oracle pool's compute."""
    from longcallr_amd import bamio
    tmp = tempfile.mkdtemp(prefix="lcr_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        path = os.path.join(tmp, "c3.bam")
        t0 = time.perf_counter()
        clen = bamio.write_reads_bam(path, batch, "chrS", level=1, threads=0)
        t_write = time.perf_counter() - t0
        size = os.path.getsize(path)
        flt = dict(min_mapq=0, min_read_length=0, divergence=2.0)
        E = api.Engine(device, params)
        best = None
        want = list(zip(batch.start0.tolist(), batch.len.tolist()))
        wins = [batch.ref[int(batch.col_off[g]):int(batch.col_off[g + 1])] for g in range(batch.n_regions)]
        for rep in range(3):
            t = {}
            t0 = time.perf_counter(); nb = bamio.NativeBam(path, 0); t["open_inflate_index"] = time.perf_counter() - t0
            t0 = time.perf_counter(); rs, re_ = nb.spans(0, **flt); regions = E.discover_regions(rs, re_, clen); t["spans_discover"] = time.perf_counter() - t0
            assert [(s, l) for s, l, _ in regions] == want, "region discovery must find the generator's genes"
            t0 = time.perf_counter(); b2 = nb.batch(0, want, wins, name_format="blob", copy=False, **flt); t["batch"] = time.perf_counter() - t0
            t0 = time.perf_counter(); E.load_batch(b2); E.sync(); t["load_batch_h2d"] = time.perf_counter() - t0
            t0 = time.perf_counter(); E.run_all(); c = E.candidates()[0]; t["stages"] = time.perf_counter() - t0
            nb.close()
            t["total"] = sum(t.values())
            if best is None or t["total"] < best["total"]:
                best = t
        E.close()
        cols = int(batch.col_off[-1])
        out = dict(workload="the headline batch as a BAM file: %d reads, %.0f MB compressed (deflate level 1), %d columns" % (batch.n_reads, size / 1e6, cols),
                   bam_write_s=t_write, gpu_path_s=best, sites_per_sec=cols / best["total"], candidates=int(c.size),
                   note="best of 3 passes; decode = liblcr's native BGZF / BAM decoder on all host threads (the reference inflates every region "
                        "twice through htslib); the kernels' share of the path is `stages`")
        if cpu_compute_s is not None:
            dec = best["open_inflate_index"] + best["spans_discover"] + best["batch"]
            out["cpu_path_s"] = dict(decode=dec, oracle_compute=cpu_compute_s, total=dec + cpu_compute_s, threads=cpu_threads,
                                     note="same native decode (the oracle has none) + the oracle pool over all regions (cpu_baseline)")
            out["gpu_over_cpu_path"] = (dec + cpu_compute_s) / best["total"]
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------

def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--pmc-child":
        return pmc_child(sys.argv[2])
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
    ap.add_argument("--no-c5", action="store_true", help="alias of --no-extras")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (+ cpu_baseline, + traffic)")
    ap.add_argument("--quick", action="store_true", help="--no-extras --no-cpu-baseline --no-traffic")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: smoke-test the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs, records travel "
                         "through host memory); the numbers of such a run mean nothing")
    ap.add_argument("--prewarm", type=int, default=40,
                    help="untimed passes during setup, before the W warm-up steps: allocations, thread pool, GPU clocks")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU: N contexts driven by N host threads (a context per worker thread, as "
                         "the reference's rayon workers would hold); 1 = one batch at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-phase", action="store_true",
                    help="lcr_phase waits for its kernels before it returns (rounds 1-4).  Default at N = 1: lcr_ctx_set_async_phase -- the next "
                         "step's bind + pileup are queued under the stage's resolve / post-phase tails (DESIGN.md §5)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()
    if a.quick:
        a.no_extras = a.no_cpu_baseline = a.no_traffic = True
    a.no_extras = a.no_extras or a.no_c5

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
    t_gen = time.perf_counter()
    batch, mine, n_global = build_shard(wl, world, rank, seed=a.seed, genes=a.genes, gene_len=a.gene_len, depth=a.depth, profile=a.profile)
    t_gen = time.perf_counter() - t_gen
    assert batch.n_regions == len(mine)
    preset = synth.preset_for(a.profile)
    params = _abi.make_params(preset, seed=2025)
    reads, regions, keep = to_device(batch, torch, dev)
    torch.cuda.synchronize()

    F = max(1, a.inflight)
    # HIP events around the pileup stage only during the timed steps (the roofline's measurement); the other kernel groups
    # are timed in one more pass behind them (every timer costs two event records on the stream: all eight, ~0.06 ms per step)
    engines = [api.Engine(local, params, timing=tuple(getattr(_abi, k) for k in PILE_TIMERS)) for _ in range(F)]
    E = engines[0]
    # asynchronous phase stage: one context, one rank (at N > 1 every step hands its records to the gather, which collects them first)
    ASYNC_PHASE[0] = not a.sync_phase
    async_phase = not a.sync_phase and dist is None and F == 1
    if async_phase:
        E.set_async_phase(True)
    if F == 1:
        E.set_stream(torch.cuda.current_stream().cuda_stream)  # torch.cuda.synchronize() then covers liblcr

    # the two record types of the final gather (thread.rs:204-221): candidate records and read -> HP / PS records
    G = (shard.RecordGather(dist, cdev, _abi.CAND_DTYPE), shard.RecordGather(dist, cdev, _abi.READ_REC_DTYPE)) if dist is not None else None
    pending = [None]
    gathered = [0, 0]
    gstat = {"wait_s": 0.0, "bytes": 0, "batches": 0}   # this rank's share of the final gathers: time spent waiting for them, bytes sent

    sink, have, stamps = [0, 0, 0], {}, []   # results read per step (consume), "this context has a batch's results to collect"

    def step(Ej):
        # bind batch k, queue its pileup, collect and read batch k - 1's results (lcr_collect_phase: the pipelined getter, valid across the
        # binding), then the other three stages of batch k.  At N > 1 the gathers take the records instead (publish, right behind phase).
        Ej.load_batch((reads, regions, keep))
        Ej.fill_data_into_freq_vec()
        if G is None and have.get(id(Ej)):
            consume(Ej.collect_phase(), sink)
        Ej.get_candidate_snps().get_fragments().phase()
        have[id(Ej)] = True
        # HIP events on the ctx stream, read after the step (lcr_pileup returns while K1 is still running); the bind kernels count
        # (they read the 37R of the numerator)
        return (Ej.kernel_ms(_abi.K_PILEUP), Ej.kernel_ms(_abi.K_SPANS) + Ej.kernel_ms(_abi.K_BIND) + Ej.kernel_ms(_abi.K_BIND_TABLE))

    def collect_last():   # the last batch's results: collected and read before the clock stops
        for Ej in engines:
            if G is None and have.get(id(Ej)):
                consume(Ej.collect_phase(), sink)
                have[id(Ej)] = False

    def read_records_host(Ej):
        pr = Ej.phase_result()
        r = np.zeros(pr["haplotag"].size, dtype=_abi.READ_REC_DTYPE)
        r["row"], r["haplotag"], r["assignment"], r["phase_set"] = np.arange(r.size), pr["haplotag"], pr["assignment"], pr["phase_set"]
        return r

    def publish(Ej):   # the gathers of this batch's records (HBM to rank 0's HBM) overlap the next batch's kernels
        if G is None:
            return
        on_dev = a.dist_backend == "nccl"
        cr = Ej.candidates_device() if on_dev else Ej.candidates()[0]
        rr = Ej.read_records_device() if on_dev else read_records_host(Ej)
        nb = lambda x, dt: (int(x[1]) if isinstance(x, tuple) else int(x.size)) * np.dtype(dt).itemsize
        h = (G[0].start(cr) + (nb(cr, _abi.CAND_DTYPE),), G[1].start(rr) + (nb(rr, _abi.READ_REC_DTYPE),))
        if pending[0] is not None:
            tw = time.perf_counter()
            G[0].finish(pending[0][0][:2], parse=False); G[1].finish(pending[0][1][:2], parse=False)
            gstat["wait_s"] += time.perf_counter() - tw
        gstat["bytes"] += h[0][2] + h[1][2]
        gstat["batches"] += 1
        pending[0] = h

    def drain():   # the last batch is also brought to rank 0's host and decoded
        if G is not None and pending[0] is not None:
            tw = time.perf_counter()
            c = G[0].finish(pending[0][0][:2], parse=True); r = G[1].finish(pending[0][1][:2], parse=True)
            gstat["wait_s"] += time.perf_counter() - tw
            if c is not None:
                gathered[0], gathered[1] = int(c.size), int(r.size)
            pending[0] = None

    def run_steps(n):
        """n passes; with F > 1 batches in flight, thread j drives context j over passes j, j + F, ... and the
        gathers are issued in pass order on every rank (collectives must line up across ranks)."""
        piles = [None] * n
        if F == 1:
            for k in range(n):
                piles[k] = step(E)
                publish(E)
                stamps.append(time.perf_counter())
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
    collect_last()
    if dist is not None:
        dist.barrier()
    sync_all()
    gstat.update(wait_s=0.0, bytes=0, batches=0)
    sink[:] = [0, 0, 0]; del stamps[:]
    t0 = time.perf_counter()
    pile_ms = run_steps(a.steps)
    drain()   # the last batch's records are on rank 0 before the clock stops
    collect_last()   # (N = 1: every one of the K batches' results has been collected and read inside the timed region)
    sync_all()
    dt_own = time.perf_counter() - t0      # this rank's K steps + its share of the gathers, before the closing barrier
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        assert dist.get_world_size() == a.gpus and dist.get_backend() == a.dist_backend, (dist.get_world_size(), dist.get_backend())
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # what makes the first N > 1 line explain itself: every rank's own time, its share of the work, the LPT cost it was
        # dealt (len x max_coverage of its regions) and what the gathers cost it
        lpt = float(np.sum(bench_costs[mine])) if bench_costs is not None else 0.0
        mine_v = torch.tensor([dt_own / a.steps * 1e3, float(batch.col_off[-1]), float(batch.bases.size), float(batch.n_reads), float(batch.n_regions), lpt,
                               gstat["wait_s"] / max(a.steps, 1) * 1e3, float(gstat["bytes"]) / max(gstat["batches"], 1)], dtype=torch.float64, device=cdev)
        allv = [torch.zeros_like(mine_v) for _ in range(world)]
        dist.all_gather(allv, mine_v)
        tab = np.stack([v.cpu().numpy() for v in allv])
        ms = tab[:, 0]
        per_rank = {"ms_per_step": ms.tolist(), "columns": tab[:, 1].astype(np.int64).tolist(), "aligned_bases": tab[:, 2].astype(np.int64).tolist(),
                    "reads": tab[:, 3].astype(np.int64).tolist(), "regions": tab[:, 4].astype(np.int64).tolist(), "lpt_cost": tab[:, 5].tolist(),
                    "gather_wait_ms_per_step": tab[:, 6].tolist(), "gather_bytes_per_step": tab[:, 7].astype(np.int64).tolist(),
                    "time_imbalance_max_over_mean": float(ms.max() / ms.mean()), "lpt_cost_imbalance_max_over_mean": float(tab[:, 5].max() / max(tab[:, 5].mean(), 1e-30)),
                    "ms_per_lpt_cost_unit_spread": float((ms / np.maximum(tab[:, 5], 1e-30)).max() / max((ms / np.maximum(tab[:, 5], 1e-30)).min(), 1e-30)),
                    "backend": dist.get_backend(), "world_size": int(dist.get_world_size()),
                    "note": "ms_per_step = a rank's own K steps incl. its share of the gathers, before the closing barrier; `value` uses the slowest rank"}

    # stage breakdown (untimed extra pass: wall clock per ABI call with a sync after each, + HIP events)
    step_ms = step_stats(stamps, t0) if (F == 1 and stamps) else None
    collected = (sink[0] // max(sink[2], 1), sink[1] // max(sink[2], 1), sink[2])
    if G is None and F == 1:
        assert sink[2] == a.steps, (sink, a.steps)
    iso_ms = None
    if async_phase:   # the pileup stage's kernels once more WITHOUT the previous step's tails beside them: five synchronous steps
        E.set_async_phase(False)
        iso = [step(E) for _ in range(5)]
        E.sync()
        iso_ms = float(np.mean([t[0] + t[1] for t in iso[1:]]))
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
    E.debug_set("timing_mask", 0)   # one more step with every kernel group timed
    step(E)
    E.sync()
    kms = {n: E.kernel_ms(k) for n, k in (("k0_ops", _abi.K_SPANS), ("k1_tiles_bin_pileup", _abi.K_PILEUP),
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
        nnz = int(fm["col"].size)
        n_items = int((pbytes - int(batch.bases.size) - 53 * cols) // 8)
        for Ej in engines[1:]:
            Ej.close()
        traffic, traffic_note = (None, "skipped (--no-traffic, or N > 1)")
        if world == 1 and not a.no_traffic:
            traffic, traffic_note = measure_traffic(batch, preset)
        out = {
            "metric": "candidate_sites_per_sec", "value": int(tot[0]) * a.steps / dt, "unit": "sites/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "step_ms": step_ms,   # host-side duration of every timed step: p50 / p99 / max (profiles/r06_stall.txt)
            "results_collected_per_step": ({"candidates": collected[0], "assigned_reads": collected[1], "batches": collected[2],
                                            "how": "lcr_collect_phase of batch k - 1 behind lcr_pileup of batch k: host records read and counted inside the timed region"}
                                           if G is None and F == 1 else "the RCCL gathers take every batch's records (config.gathered_records_last_batch)"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 in, u32 counts, f64 likelihoods, i64 fixed-point phase scores", "data": "synthetic",
            "config": {"workload": "%s; synthetic %s reads, %d distinct genes (regions) x %d bp per GPU at %.0fx mean aligned depth, "
                                   "generator seed %d; %d regions in the job, LPT-partitioned over %d rank(s), preset %s; "
                                   "step = bind+pileup+candidates+fragments+phase"
                                   % (w_name, a.profile, a.genes, a.gene_len, a.depth, a.seed, n_global, world, preset),
                       "columns": int(tot[0]), "aligned_bases": int(tot[1]), "reads": int(tot[2]), "candidates": int(tot[3]),
                       "fragment_nnz": int(tot[4]), "phasing_reads": int(tot[5]),
                       "columns_rank0": cols, "aligned_bases_rank0": int(batch.bases.size), "generate_s_rank0": t_gen,
                       "parallelism": "regions sharded over %d GPU(s) by shard.assign_regions (LPT on len x max_coverage); final gather of candidate "
                                      "records and read -> HP / PS records to rank 0 (RCCL), overlapped with the next batch" % world,
                       "gathered_records_last_batch": {"candidates": gathered[0], "reads": gathered[1]} if dist is not None else None,
                       "per_rank": per_rank,
                       "batches_in_flight_per_gpu": F,
                       "phase_stage": ("asynchronous (lcr_ctx_set_async_phase: lcr_phase returns with its kernels in flight, the next step's pileup is queued "
                                       "behind its restarts and runs beside its resolve / post-phase tails; every batch's results collected by lcr_collect_phase before the next lcr_candidates); "
                                       "GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES")) if async_phase else "synchronous",
                       "scaling_reference": ("the N = 1 point of THIS workload (one GPU's 1 000-gene share of C4) is `stages.c4_share.sites_per_sec` of the "
                                             "N = 1 line; the N = 1 headline `value` is C3, a different workload") if world > 1 else None},
            "roofline": {"bound": "hbm", "kernel": "pileup stage = k0_bind_a/b (read headers, tables) + k0_ops + k1_tiles_a/b + k0_desc_bin + k1_pileup + k1_empty_tiles (+ k1_zonefix on HiFi presets): what replaces fill_data_into_freq_vec",
                         "achieved": stage_bytes / (stage_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": stage_bytes / (stage_ms * 1e-3) / 1e9 / 8000.0,
                         "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_detail": traffic if traffic else traffic_note,
                         "algorithmic_bytes": stage_bytes, "avg_ms": stage_ms,
                         "isolated": ({"avg_ms": iso_ms, "frac": stage_bytes / (iso_ms * 1e-3) / 8e12,
                                       "note": "the same kernels in synchronous steps (nothing of the previous step beside them): in the timed steps the stage "
                                               "runs beside the previous step's resolve / post-phase kernels (asynchronous phase stage), which costs it ~10 % and buys the step ~12 %"}
                                      if iso_ms else None),
                         "note": "algorithmic bytes B + 4C + 37R + 53L (bases once, CIGAR, read headers, ref byte + 13 u32 planes per column); "
                                 "HIP events on the ctx stream, rank 0, mean over the timed steps",
                         "k0_ops": {"avg_ms": avg_k0_ms, "algorithmic_bytes": 4 * int(batch.cigar.size) + 64 * batch.n_reads + 8 * n_items,
                                    "frac": (4 * int(batch.cigar.size) + 64 * batch.n_reads + 8 * n_items) / (avg_k0_ms * 1e-3) / 8e12,
                                    "note": "4C + 64R read, 8 B per record item written; avg_ms includes the two bind kernels of lcr_load_batch"},
                         "k1": {"algorithmic_bytes": pbytes, "avg_ms": avg_ms, "achieved": achieved, "frac": achieved / 8000.0,
                                "note": "the tally with its tile passes and the chunk binning (k1_tiles_a/b + k0_desc_bin + k1_pileup + k1_empty_tiles): bases once + 8 B per record item + 53 B/column"}},
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
        extras = world == 1 and not a.no_extras
        cpu = None
        if not a.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the other ranks of a node would sit idle behind it
            cpu = cpu_baseline(batch, params, a.cpu_budget)
            out["cpu_baseline"] = cpu
        st = out["stages"]
        if extras and F == 1:
            st["batches_in_flight"] = batches_in_flight_stage(api, torch, local, params, (reads, regions, keep), cols)
            st["end_to_end_from_bam"] = end_to_end_stage(api, _abi, torch, local, batch, params,
                                                         (cols / cpu["value"]) if cpu else None, cpu["cores"] if cpu else None)
            st["host_fed"] = host_fed_stage(api, local, batch, params, cols)
        del keep, reads, regions
        torch.cuda.empty_cache()
        if extras:
            # the other seeds and the C4 share run through the headline run's context: ONE long-lived worker, as the caller holds it
            # (thread.rs:77-143); every pass is reported as it comes, with its per-step p50 / p99 / max
            E.set_async_phase(ASYNC_PHASE[0] and F == 1)
            if wl == "c3" and a.seed == 1:
                st["seeds"] = seeds_stage(api, _abi, torch, local, "c3", params, [2, 3, 4, 5], E=E if F == 1 else None)
                st["seeds"]["1"] = {"columns": cols, "aligned_bases": int(batch.bases.size), "ms_per_step": out["ms_per_step"], "sites_per_sec": out["value"],
                                    "pileup_stage_ms": stage_ms, "pileup_stage_frac_of_hbm_peak": out["roofline"]["frac"], "candidates": int(cands.size), "step_ms": step_ms}
            st["c4_share"] = c4_share_stage(api, _abi, synth, torch, local, cpu=not a.no_cpu_baseline, E=E if F == 1 else None)
        E.close()
        if extras:
            st["demo"] = demo_stage(api, _abi, local, cpu=not a.no_cpu_baseline)
            st["c5"] = c5_stage(api, _abi, synth, torch, local, cpu=not a.no_cpu_baseline)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
