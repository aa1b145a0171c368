"""The driver loop around the hot path: BAM + reference FASTA -> phased VCF (+ phased BAM).

Host orchestration only; the work is done behind the C ABI (include/lcr.h).  It replaces what
`multithread_phase_haplotag_germline` (reference src/thread.rs:26-361) and its callers in src/main.rs do around
the five hot-path calls, with one context per GPU instead of one rayon task per region:

  util.rs:214-234 load_reference / parse_fai                 -> load_reference / parse_fai below
  util.rs:558-600 extract_isolated_regions_parallel          -> lcr_bam_spans + lcr_discover_regions per contig
  thread.rs:77-221 the per-region closure                    -> one lcr_load_batch + pileup / candidates /
                                                                fragments / phase per contig (all its regions)
  thread.rs:223-305 VCF header + records                     -> write_vcf (records from vcf.format_records)
  thread.rs:307-361 phased BAM                               -> lcr_bam_write_phased

Not here (out of scope): gene annotation / exon filter, external VCF candidates, down-sampling,
region truncation, the somatic model.  Records are written in contig order of the .fai and position order inside
a contig (the reference writes them in region-completion order, thread.rs:216-221: compare as a set)."""
import os

import numpy as np

from . import _abi, api, bamio, vcf

VCF_HEADER_TAIL = (   # thread.rs:232-262, verbatim
    '##FILTER=<ID=PASS,Description="All filters passed">\n'
    '##FILTER=<ID=LowQual,Description="Low phasing quality">\n'
    '##FILTER=<ID=HomRef,Description="Homo reference">\n'
    '##FILTER=<ID=RnaEdit,Description="RNA editing">\n'
    '##FILTER=<ID=Multiallelic,Description="Multiallelic SNP">\n'
    '##FILTER=<ID=dn,Description="Dense cluster of variants">\n'
    '##INFO=<ID=RDS,Number=1,Type=String,Description="RNA editing or Dense SNP or Single SNP.">\n'
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    '##FORMAT=<ID=PS,Number=1,Type=Integer,Description="Phase Set">\n'
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n'
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">\n'
    '##FORMAT=<ID=AF,Number=A,Type=Float,Description="Allele Frequency">\n'
    '##FORMAT=<ID=PQ,Number=1,Type=Float,Description="Phasing Quality">\n'
    '##FORMAT=<ID=AE,Number=A,Type=Integer,Description="Haplotype expression of two alleles">\n'
    '##FORMAT=<ID=SQ,Number=1,Type=Float,Description="Somatic Score">\n'
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample\n")


def load_reference(path):
    """util.rs:214-222: record id (first word of the '>' line) -> sequence bytes, case preserved."""
    seqs, name, parts = {}, None, []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    seqs[name] = np.frombuffer(b"".join(parts), dtype=np.uint8)
                name, parts = line[1:].split()[0].decode(), []
            else:
                parts.append(line.strip())
    if name is not None:
        seqs[name] = np.frombuffer(b"".join(parts), dtype=np.uint8)
    return seqs


def parse_fai(path):
    """util.rs:224-234: [(contig, length)] in file order."""
    out = []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if len(p) >= 2:
                out.append((p[0], int(p[1])))
    return out


def write_vcf(path, contig_lengths, body_lines):
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.3\n")
        for name, length in contig_lengths:   # thread.rs:225-230
            f.write("##contig=<ID=%s,length=%d>\n" % (name, length))
        f.write(VCF_HEADER_TAIL)
        f.writelines(body_lines)


def chunk_regions(regions, max_cost=2.0e9, max_cols=2.0e8, max_regions=8192):
    """Cut a contig's regions [(start0, len, max_cov)] into consecutive chunks whose estimated work (sum of
    len x max_coverage, an upper bound of the aligned bases) and column count stay below the budgets: liblcr's
    per-batch limits (32-bit record pool, 2^31 reads, 2^28 matrix entries) then never surface as a failed run.
    Results are independent of batch composition, so the cut changes nothing in the output."""
    chunks, cur, cost, cols = [], [], 0.0, 0
    for r in regions:
        c = float(r[1]) * max(int(r[2]), 1)
        if cur and (cost + c > max_cost or cols + r[1] > max_cols or len(cur) >= max_regions):
            chunks.append(cur)
            cur, cost, cols = [], 0.0, 0
        cur.append(r)
        cost += c
        cols += r[1]
    if cur:
        chunks.append(cur)
    return chunks


def _gather_names(name_off, blob, rows):
    """(name_off, blob) of the reads `rows` out of a batch's name blob, without a Python object per read"""
    off = name_off.astype(np.int64)
    lens = off[rows + 1] - off[rows]
    new_off = np.zeros(rows.size + 1, dtype=np.int64)
    np.cumsum(lens, out=new_off[1:])
    idx = np.repeat(off[rows] - new_off[:-1], lens) + np.arange(int(new_off[-1]))
    return new_off.astype(np.uint64), blob[idx]


def run(bam_path, ref_path, out_vcf, out_bam=None, preset="hifi-masseq", contigs=None, device=0, threads=0, seed=2025,
        read_filter=None, devices=None, chunk_cost=2.0e9, async_phase=True, **param_overrides):
    """BAM + FASTA (+ .fai) -> phased VCF and, with out_bam, the phased BAM.  Returns a dict of counts.
    devices: GPUs to use (default [device]); a contig's regions are cut into chunks (chunk_regions) that the engines --
    one context and one host thread per device -- take in turn (regions are independent units, thread.rs:77; the BAM
    decoder cuts the batches on the calling thread).  A GPU may be named more than once (devices=[0, 0, 0]): that many
    chunks are then in flight on it, each filling the queue gaps of the others' host round trips (bench.py
    stages.batches_in_flight: +25 % at three).  async_phase: the engines run the asynchronous phase stage (a chunk's upload + pileup
    beside the previous chunk's resolve / post-phase tails, results through lcr_collect_phase).  The output does not depend on devices,
    chunk_cost or async_phase."""
    from concurrent.futures import ThreadPoolExecutor
    import threading
    fai = ref_path + ".fai"
    if not os.path.exists(fai):
        raise FileNotFoundError("Reference index file .fai does not exist.")   # util.rs:575-577
    contig_lengths = parse_fai(fai)
    refs = load_reference(ref_path)
    flt = dict(_abi.READ_FILTER)
    flt.update(read_filter or {})
    params = _abi.make_params(preset, seed=seed, **param_overrides)
    nb = bamio.NativeBam(bam_path, threads)
    bam_ids = {n: i for i, (n, _) in enumerate(nb.refs)}
    devices = list(devices) if devices else [device]
    engines = [api.Engine(d, params) for d in devices]
    # a ctx is single-threaded (include/lcr.h): the producer thread below never touches a working engine -- region
    # discovery has a context of its own (its kernels share scratch buffers and a stream with nothing else)
    scout = api.Engine(devices[0], params)
    free = list(range(len(engines)))
    free_lock = threading.Condition()
    stats = dict(contigs=0, regions=0, reads=0, candidates=0, vcf_records=0, chunks=0)

    # Every engine is a long-lived worker with the ASYNCHRONOUS phase stage (lcr_ctx_set_async_phase, include/lcr.h): a chunk is uploaded
    # into one of the context's two staging slots (lcr_load_batch_async + lcr_bind_batch: the device-resident form, which does not wait
    # for the stage in flight), its pileup is queued, THEN the previous chunk's results are collected (lcr_collect_phase: the getter that
    # outlives the binding) and turned into VCF text / read tags while this chunk's kernels run, then candidates / fragments / phase.
    results = {}            # chunk index -> dict(text, n_cand[, hp, ps, names])
    in_flight = [None] * len(engines)   # per engine: (chunk index, name, batch, want_reads, fragmat info) of the chunk whose phase stage runs
    n_slot = [0] * len(engines)
    for E in engines:
        E.set_async_phase(async_phase)

    def finish(k):          # collect engine k's chunk in flight
        E = engines[k]
        idx, name, batch, want_reads, fm = in_flight[k]
        in_flight[k] = None
        res = E.collect_phase()
        cands = res["cand"]
        lines = vcf.format_records(cands, name, params.min_phase_score)
        out = dict(text=lines if isinstance(lines, str) else "".join(lines), n_cand=int(cands.size))
        if want_reads:
            asg = res["assignment"].astype(np.int32)
            # thread.rs:204-214: every for_phasing fragment has an assignment entry (0 / 1 / 2), a phase set only if set
            out["hp"] = np.where((fm["row_for_phasing"] != 0) | (asg != 0), asg, -1)
            out["ps"] = res["phase_set"].copy()
            out["names"] = _gather_names(batch.name_off, batch.name_blob, fm["row_read"].astype(np.int64))
        results[idx] = out

    def work(idx, batch, name, want_reads):   # one chunk on whichever engine is free
        with free_lock:
            while not free:
                free_lock.wait()
            k = free.pop()
        try:
            E = engines[k]
            slot = n_slot[k] & 1
            n_slot[k] += 1
            E.load_batch_async(batch, slot).bind_batch(slot)
            E.fill_data_into_freq_vec()
            if in_flight[k] is not None:
                finish(k)
            E.get_candidate_snps().get_fragments()
            fm = None
            if want_reads:      # rows of the fragment matrix (read of every row, num_hete_links >= min_linkers): known before the phase stage
                f = E.fragmat()
                fm = dict(row_for_phasing=f["row_for_phasing"], row_read=f["row_read"])
            E.phase()
            in_flight[k] = (idx, name, batch, want_reads, fm)
        finally:
            with free_lock:
                free.append(k)
                free_lock.notify()

    regions_out = []
    n_chunks = 0
    with ThreadPoolExecutor(max_workers=len(engines)) as pool:
        pending = []
        for name, length in contig_lengths:
            if contigs is not None and name not in contigs:
                continue
            if name not in bam_ids or name not in refs:
                continue
            rid = bam_ids[name]
            rs, re_ = nb.spans(rid, **flt)
            if rs.size == 0:
                continue
            regions = scout.discover_regions(rs, re_, length)     # util.rs:236-332
            if not regions:
                continue
            ref = refs[name]
            stats["contigs"] += 1; stats["regions"] += len(regions)
            for chunk in chunk_regions(regions, max_cost=chunk_cost):
                wins = [ref[s:s + l] if s + l <= ref.size else np.concatenate([ref[s:], np.full(s + l - ref.size, ord("N"), np.uint8)])
                        for s, l, _ in chunk]
                batch = nb.batch(rid, [(s, l) for s, l, _ in chunk], wins, name_format="blob", **flt)
                stats["reads"] += batch.n_reads; stats["chunks"] += 1
                while len(pending) > len(engines):     # bounded: at most one batch waiting per engine
                    pending.pop(0).result()
                pending.append(pool.submit(work, n_chunks, batch, name, out_bam is not None))
                n_chunks += 1
                regions_out.extend((rid, s, l) for s, l, _ in chunk)
        for f in pending:
            f.result()
    for k in range(len(engines)):       # the last chunk of every engine
        if in_flight[k] is not None:
            finish(k)
    results = [results[i] for i in range(n_chunks)]
    for E in engines + [scout]:
        E.close()
    text = "".join(r["text"] for r in results)
    stats["candidates"] = sum(r["n_cand"] for r in results)
    stats["vcf_records"] = text.count("\n")
    write_vcf(out_vcf, contig_lengths, [text])
    if out_bam is not None:
        hp = np.concatenate([r["hp"] for r in results]) if results else np.zeros(0, np.int32)
        ps = np.concatenate([r["ps"] for r in results]) if results else np.zeros(0, np.uint32)
        offs, blobs, base = [np.zeros(1, np.uint64)], [], 0
        for r in results:
            o, bl = r["names"]
            offs.append(o[1:] + np.uint64(base)); blobs.append(bl)
            base += int(o[-1])
        name_off = np.concatenate(offs)
        blob = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
        nb.write_phased(out_bam, regions_out, (name_off, blob), hp, ps, threads=threads)
    nb.close()
    return stats
