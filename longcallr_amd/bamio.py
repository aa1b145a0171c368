"""Host-side BGZF/BAM decoding -> ReadBatch (SoA) for liblcr.

`NativeBam` is the product path: liblcr's multithreaded decoder (csrc/lcr_bam.cpp,
`lcr_bam_*` in include/lcr.h).  The pure-Python functions below restate the same rules record by record;
tests use them as the checker of the native decoder and to build the demo fixture.

Host plumbing only: the reference uses rust-htslib for this
(src/util.rs:636-691, src/fragment.rs:19-59).  Implements exactly what the hot path needs:
record decode, the read filter of util.rs:652-668, `leading/trailing_softclips`, the `de:f` and
`ts:A` aux tags, htslib's region-overlap rule for `fetch`, and the coverage-island region
discovery of util.rs:236-332.
"""
import os
import struct
import zlib

import numpy as np

from ._abi import ReadBatch

_NT16 = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
_CONSUMES_REF = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1], dtype=bool)  # MIDNSHP=X


def bgzf_decompress(path):
    raw = open(path, "rb").read()
    out, off = [], 0
    while off < len(raw):
        if raw[off:off + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % off)
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        p, bsize = off + 12, None
        while p < off + 12 + xlen:
            si1, si2, slen = raw[p], raw[p + 1], struct.unpack_from("<H", raw, p + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", raw, p + 4)[0]
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        cdata = raw[off + 12 + xlen: off + bsize + 1 - 8]
        out.append(zlib.decompress(cdata, -15))
        off += bsize + 1
    return b"".join(out)


def _aux_scan(buf, p, end, cg_out=None):
    """Return (de or None, ts code 0/1/2) from the aux block buf[p:end]; a CG:B,I tag's values go to cg_out."""
    de, ts = None, 0
    while p + 3 <= end:
        tag, typ = buf[p:p + 2], chr(buf[p + 2])
        p += 3
        if typ in "AcC":
            if tag == b"ts" and typ == "A":
                ts = 1 if buf[p:p + 1] == b"+" else (2 if buf[p:p + 1] == b"-" else 0)
            p += 1
        elif typ in "sS":
            p += 2
        elif typ in "iI":
            p += 4
        elif typ == "f":
            if tag == b"de":
                de = struct.unpack_from("<f", buf, p)[0]
            p += 4
        elif typ in "ZH":
            while buf[p] != 0:
                p += 1
            p += 1
        elif typ == "B":
            sub, cnt = chr(buf[p]), struct.unpack_from("<I", buf, p + 1)[0]
            if tag == b"CG" and sub in "Ii" and cg_out is not None:
                cg_out.append(np.frombuffer(buf, dtype="<u4", count=cnt, offset=p + 5).copy())
            p += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
        else:
            raise ValueError("bad aux type %r" % typ)
    return de, ts


def read_bam(path, keep_raw=False):
    """Decode a BAM file. Returns (refs [(name, length)], records list of dicts) in file order; with keep_raw every
    record also carries its bytes (`raw`, without block_size), the offset of its aux block (`aux_off`) and the
    first record carries the inflated header bytes (`header`)."""
    buf = bgzf_decompress(path)
    if buf[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text = struct.unpack_from("<i", buf, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", buf, p)[0]
        name = buf[p + 4:p + 4 + l_name - 1].decode()
        l_ref = struct.unpack_from("<i", buf, p + 4 + l_name)[0]
        refs.append((name, l_ref))
        p += 8 + l_name
    recs = []
    header_end = p
    while p < len(buf):
        (bs, ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl) = struct.unpack_from(
            "<iiiBBHHHiiii", buf, p)
        q = p + 36
        name = buf[q:q + l_rn - 1].decode()
        q += l_rn
        cigar = np.frombuffer(buf, dtype="<u4", count=n_cig, offset=q).copy()
        q += 4 * n_cig
        packed = np.frombuffer(buf, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q)
        seq = np.empty(2 * packed.size, dtype=np.uint8)
        seq[0::2] = _NT16[packed >> 4]
        seq[1::2] = _NT16[packed & 15]
        seq = seq[:l_seq].copy()
        q += (l_seq + 1) // 2
        qual = np.frombuffer(buf, dtype=np.uint8, count=l_seq, offset=q).copy()
        q += l_seq
        cg = []
        de, ts = _aux_scan(buf, q, p + 4 + bs, cg)
        # long CIGAR (SAM spec 4.2.2, htslib applies it when reading): placeholder <l_seq>S<n>N + the real CIGAR in CG:B,I
        # (htslib's bam_tag2cigar: mapped record, first op <l_seq>S, CG:B,I or B,i present; without the tag the record stays as it is)
        if n_cig >= 1 and ref_id >= 0 and pos >= 0 and (cigar[0] & 15) == 4 and int(cigar[0] >> 4) == l_seq and cg:
            cigar = cg[0]
            n_cig = int(cigar.size)
        ops, lens = cigar & 15, cigar >> 4
        ref_len = int(lens[_CONSUMES_REF[np.minimum(ops, 8)] & (ops <= 8)].sum())
        lead = trail = 0
        if n_cig:
            if ops[0] == 4:
                lead = int(lens[0])
            elif ops[0] == 5 and n_cig > 1 and ops[1] == 4:
                lead = int(lens[1])
            if ops[-1] == 4:
                trail = int(lens[-1])
            elif ops[-1] == 5 and n_cig > 1 and ops[-2] == 4:
                trail = int(lens[-2])
        recs.append(dict(name=name, ref_id=ref_id, pos=pos, mapq=mapq, flag=flag, l_seq=l_seq,
                         cigar=cigar, seq=seq, qual=qual, de=de, ts=ts, ref_len=ref_len,
                         lead=lead, trail=trail))
        if keep_raw:
            recs[-1]["raw"] = bytes(buf[p + 4:p + 4 + bs])
            recs[-1]["aux_off"] = q - (p + 4)
            if len(recs) == 1:
                recs[-1]["header"] = bytes(buf[:header_end])
        p += 4 + bs
    return refs, recs


def passes_filter(r, min_mapq=20, min_read_length=500, divergence=0.5):
    """util.rs:652-668 / fragment.rs:32-49."""
    if r["mapq"] < min_mapq or r["l_seq"] < min_read_length:
        return False
    if r["flag"] & 0x4 or r["flag"] & 0x100 or r["flag"] & 0x800:
        return False
    if r["de"] is not None and r["de"] >= divergence:
        return False
    return True


def discover_regions(recs, ref_id, ref_len):
    """util.rs:236-332 (no truncation) on the host: coverage islands as (start0, len, max_cov).

    The reference emits 1-based [start, end) = [first0+1, last0+2); we return the 0-based column
    window (start0=first0, len=last0-first0+1).  As in the reference, a single-column island is not
    emitted on its own: it stays pending and starts the region that ends with the next island
    (cursors and max_coverage are only reset when a region is emitted).  `lcr_discover_regions` is the
    GPU version of the same function.
    """
    diff = np.zeros(ref_len + 1, dtype=np.int64)
    for r in recs:
        if r["ref_id"] != ref_id:
            continue
        s, e = r["pos"], r["pos"] + (r["ref_len"] if r["ref_len"] > 0 else 1)
        diff[s] += 1
        diff[min(e, ref_len)] -= 1
    depth = np.cumsum(diff[:-1])
    cov = depth > 0
    edges = np.flatnonzero(np.diff(np.concatenate(([0], cov.view(np.int8), [0]))))
    out, pend, running = [], -1, 0
    for s, e in zip(edges[0::2], edges[1::2]):  # island [s, e)
        running = max(running, int(depth[s:e].max()))
        if pend < 0:
            pend = int(s)
        if e - 1 > pend:
            out.append((pend, int(e - pend), running))
            pend, running = -1, 0
    return out


def build_batch(recs, regions, ref_windows):
    """Group filtered records by region with htslib's fetch rule.

    regions: list of (start0, len); the reference calls fetch((chr, start, end)) with the 1-based
    numbers used as a 0-based half-open interval (util.rs:637), i.e. [start0+1, start0+len+1).
    ref_windows: list of uint8 arrays, one per region (len bytes each).
    """
    cols = {f: [] for f in ["pos", "seq_len", "lead_clip", "trail_clip", "flags", "n_cig"]}
    bases, quals, cigars, names = [], [], [], []
    seq_off, cig_off, read_begin = [], [], [0]
    so = co = 0
    for (start0, length) in regions:
        beg, end = start0 + 1, start0 + length + 1
        for r in recs:
            rend = r["pos"] + (r["ref_len"] if r["ref_len"] > 0 else 1)
            if not (r["pos"] < end and rend > beg):
                continue
            cols["pos"].append(r["pos"])
            cols["seq_len"].append(r["l_seq"])
            cols["lead_clip"].append(r["lead"])
            cols["trail_clip"].append(r["trail"])
            cols["flags"].append((1 if r["flag"] & 0x10 else 0) | (r["ts"] << 1))
            cols["n_cig"].append(len(r["cigar"]))
            seq_off.append(so)
            cig_off.append(co)
            bases.append(r["seq"])
            quals.append(r["qual"])
            cigars.append(r["cigar"])
            names.append(r["name"])
            so += r["l_seq"]
            co += len(r["cigar"])
        read_begin.append(len(seq_off))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return ReadBatch(
        seq_off=np.array(seq_off, dtype=np.uint64), cig_off=np.array(cig_off, dtype=np.uint64),
        bases=cat(bases, np.uint8), quals=cat(quals, np.uint8), cigar=cat(cigars, np.uint32),
        start0=[s for s, _ in regions], len=[l for _, l in regions], read_begin=read_begin,
        ref=cat(list(ref_windows), np.uint8), names=names, **cols)


def phased_stream(recs, regions, names, hp, ps):
    """Record-by-record restatement of thread.rs:307-361 on read_bam(keep_raw=True) records: the inflated bytes of
    the phased BAM (header + kept records with HP:i / PS:I appended).  regions = [(ref_id, start0, len)]."""
    m_hp, m_ps = {}, {}
    for n, h, p in zip(names, hp, ps):
        if h >= 0 and n not in m_hp:
            m_hp[n] = int(h)
        if p != 0 and n not in m_ps:
            m_ps[n] = int(p)

    def has_tag(raw, q, tag):
        while q + 3 <= len(raw):
            if raw[q:q + 2] == tag:
                return True
            typ = chr(raw[q + 2])
            q += 3
            if typ in "AcC":
                q += 1
            elif typ in "sS":
                q += 2
            elif typ in "iIf":
                q += 4
            elif typ in "ZH":
                q = raw.index(b"\0", q) + 1
            elif typ == "B":
                sub, cnt = chr(raw[q]), struct.unpack_from("<I", raw, q + 1)[0]
                q += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            else:
                raise ValueError("bad aux type")
        return False
    out = [recs[0]["header"]] if recs else []
    for ref_id, start0, length in regions:
        beg, end = start0 + 1, start0 + length + 1
        for r in recs:
            rend = r["pos"] + (r["ref_len"] if r["ref_len"] > 0 else 1)
            if r["ref_id"] != ref_id or not (r["pos"] < end and rend > beg):
                continue
            if r["flag"] & (0x4 | 0x100 | 0x800):
                continue
            if r["pos"] + 1 < beg or rend + 1 > end:
                continue
            raw = r["raw"]
            h = m_hp.get(r["name"])
            if h is not None and h != 0 and not has_tag(raw, r["aux_off"], b"HP"):
                raw = raw + b"HPi" + struct.pack("<i", h)
            p = m_ps.get(r["name"])
            if p is not None and not has_tag(r["raw"], r["aux_off"], b"PS"):
                raw = raw + b"PSI" + struct.pack("<I", p)
            out.append(struct.pack("<i", len(raw)) + raw)
    return b"".join(out)


def write_reads_bam(path, batch, contig="chrS", contig_len=None, level=1, threads=0):
    """A ReadBatch as a coordinate-sorted one-contig BAM (lcr_bam_write_reads): synthetic data sets from file, round trips.
    The batch's reads must be sorted by position over ALL its regions (synth batches are: regions ascend)."""
    import ctypes as C
    from . import _lib
    l = _lib.load()
    rd = batch.c_reads()
    if contig_len is None:
        contig_len = int(batch.start0[-1] + batch.len[-1]) + 1000
    rc = l.lcr_bam_write_reads(os.fsencode(path), contig.encode(), int(contig_len), C.byref(rd), level, threads)
    if rc:
        raise _lib.LcrError("lcr_bam_write_reads(%s) failed (%d)" % (path, rc))
    return contig_len


class NativeBam:
    """liblcr's BAM decoder (lcr_bam_* in include/lcr.h): the file is mapped, one contig at a time is inflated (in
    parallel) and indexed, batches for `Engine.load_batch` are cut out of that index.  Mirrors read_bam /
    passes_filter / build_batch above."""

    def __init__(self, path, threads=0, keep_bytes=None):
        """keep_bytes: files whose inflated size is within it stay inflated from the open pass on (None: lcr_bam_open's 4 GiB;
        0: one contig at a time, inflated when it is first used)"""
        import ctypes as C
        from . import _lib
        self._C, self._l = C, _lib.load()
        self._h = C.c_void_p()
        rc = (self._l.lcr_bam_open(os.fsencode(path), threads, C.byref(self._h)) if keep_bytes is None else
              self._l.lcr_bam_open_keep(os.fsencode(path), threads, int(keep_bytes), C.byref(self._h)))
        if rc:
            msg = self._l.lcr_bam_last_error(self._h).decode() if self._h else "out of memory"
            self.close()
            raise _lib.LcrError("lcr_bam_open(%s): %s" % (path, msg))
        n, names, lens = C.c_int32(), C.POINTER(C.c_char_p)(), C.POINTER(C.c_int64)()
        self._chk(self._l.lcr_bam_refs(self._h, C.byref(n), C.byref(names), C.byref(lens)))
        self.refs = [(names[i].decode(), int(lens[i])) for i in range(n.value)]
        nrec = C.c_int64()
        self._chk(self._l.lcr_bam_n_records(self._h, C.byref(nrec)))
        self.n_records = nrec.value

    def resident(self):
        """(bytes of inflated stream + record index held now, their peak since open)"""
        now, peak = self._C.c_int64(), self._C.c_int64()
        self._chk(self._l.lcr_bam_resident(self._h, self._C.byref(now), self._C.byref(peak)))
        return int(now.value), int(peak.value)

    def _chk(self, rc):
        if rc:
            from . import _lib
            raise _lib.LcrError(self._l.lcr_bam_last_error(self._h).decode())

    def _filter(self, min_mapq=20, min_read_length=500, divergence=0.5):
        from . import _abi
        return _abi.LcrReadFilter(min_mapq, min_read_length, divergence)

    def spans(self, ref_id, **flt):
        """(reference_start, reference_end) of the passing reads of a contig: input of lcr_discover_regions."""
        C = self._C
        f, n = self._filter(**flt), C.c_int32()
        s, e = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        self._chk(self._l.lcr_bam_spans(self._h, ref_id, C.byref(f), C.byref(n), C.byref(s), C.byref(e)))
        return (np.ctypeslib.as_array(s, (n.value,)).copy() if n.value else np.zeros(0, np.int32),
                np.ctypeslib.as_array(e, (n.value,)).copy() if n.value else np.zeros(0, np.int32))

    def batch(self, ref_id, regions, ref_windows, name_format="list", copy=True, **flt):
        """ReadBatch of the passing reads grouped by region (fetch rule of util.rs:637).
        copy=True: the arrays are copies; copy=False: the large arrays (bases, quals, cigar) are VIEWS of the decoder's buffers,
        as the C ABI hands them out -- valid until the next batch() / close() of this handle (a caller that loads the batch
        into an engine straight away saves a GB of memcpy).
        name_format="list": `batch.names` is a list of str; "blob": `batch.name_off` (uint64, n + 1) and
        `batch.name_blob` (uint8, NUL-terminated names) -- no Python object per read."""
        C = self._C
        from . import _abi
        f = self._filter(**flt)
        start0 = np.ascontiguousarray([s for s, _ in regions], dtype=np.int64)
        length = np.ascontiguousarray([l for _, l in regions], dtype=np.int32)
        rd, rb = _abi.LcrReads(), C.POINTER(C.c_int32)()
        noff, names = C.POINTER(C.c_uint64)(), C.c_char_p()
        self._chk(self._l.lcr_bam_batch(self._h, ref_id, C.byref(f), len(regions), start0.ctypes.data, length.ctypes.data,
                                        C.byref(rd), C.byref(rb), C.byref(noff), C.byref(names)))
        nr = rd.n_reads

        def arr(ptr, n, dt, view=False):
            if n == 0 or not ptr:
                return np.zeros(0, dt)
            a = np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt)
            return a if view else a.copy()
        kw = {fld: arr(getattr(rd, fld), nr, _abi.ReadBatch.DTYPES[fld])
              for fld in ("pos", "seq_len", "lead_clip", "trail_clip", "flags", "seq_off", "cig_off", "n_cig")}
        kw["bases"] = arr(rd.bases, rd.n_bases, np.uint8, not copy)
        kw["quals"] = arr(rd.quals, rd.n_bases, np.uint8, not copy)
        kw["cigar"] = arr(rd.cigar, rd.n_cigar, np.uint32, not copy)
        offs = np.ctypeslib.as_array(noff, (nr + 1,)).copy() if nr else np.zeros(1, np.uint64)
        blob = C.string_at(C.cast(names, C.c_void_p), int(offs[-1])) if nr else b""
        nm = [blob[int(offs[i]):int(offs[i + 1]) - 1].decode() for i in range(nr)] if name_format == "list" else None
        read_begin = np.ctypeslib.as_array(rb, (len(regions) + 1,)).copy()
        cat = np.concatenate([np.asarray(w, np.uint8) for w in ref_windows]) if len(ref_windows) else np.zeros(0, np.uint8)
        out = ReadBatch(start0=start0, len=length, read_begin=read_begin, ref=cat, names=nm, **kw)
        if name_format == "blob":
            out.name_off, out.name_blob = offs, np.frombuffer(blob, dtype=np.uint8)
        return out

    def write_phased(self, out_path, regions, names, hp, ps, level=-1, threads=0):
        """thread.rs:307-361: regions = [(ref_id, start0, len)] in output order; names / hp / ps = the read
        assignment and phase-set entries in queue order (hp < 0: no assignment entry, ps == 0: no phase set)."""
        C = self._C
        ref = np.ascontiguousarray([r for r, _, _ in regions], dtype=np.int32)
        start0 = np.ascontiguousarray([s for _, s, _ in regions], dtype=np.int64)
        length = np.ascontiguousarray([l for _, _, l in regions], dtype=np.int32)
        if isinstance(names, tuple):     # (name_off uint64[n + 1], blob of NUL-terminated names): no Python strings
            off = np.ascontiguousarray(names[0], dtype=np.uint64)
            blob = np.ascontiguousarray(names[1], dtype=np.uint8).tobytes()
            enc = range(off.size - 1)
        else:
            enc = [n.encode() + b"\0" for n in names]
            off = np.zeros(len(enc) + 1, dtype=np.uint64)
            if enc:
                off[1:] = np.cumsum([len(e) for e in enc])
            blob = b"".join(enc)
        hp = np.ascontiguousarray(hp, dtype=np.int32)
        ps = np.ascontiguousarray(ps, dtype=np.uint32)
        assert hp.size == ps.size == len(enc)
        self._chk(self._l.lcr_bam_write_phased(self._h, os.fsencode(out_path), len(regions), ref.ctypes.data, start0.ctypes.data,
                                               length.ctypes.data, len(enc), off.ctypes.data, blob, hp.ctypes.data, ps.ctypes.data,
                                               level, threads))

    def close(self):
        if getattr(self, "_h", None):
            self._l.lcr_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass
