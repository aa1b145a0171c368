"""liblcr's BGZF / BAM decoder (csrc/lcr_bam.cpp, lcr_bam_* in include/lcr.h) against the
record-by-record Python restatement of the same rules (longcallr_amd/bamio.py) — on demo.bam, the reference's
own fixture, and on hand-written BAM files that exercise the corners of the format the hot path depends on
(util.rs:636-691, fragment.rs:19-59).  Host only: no GPU needed."""
import os
import struct
import zlib

import numpy as np
import pytest

import helpers
from longcallr_amd import _abi, _lib, bamio

DEMO = os.path.join(helpers.GOLDEN, "demo.bam")
CIG = {c: i for i, c in enumerate("MIDNSHP=X")}
NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def bgzf(payload, block=300):
    """BGZF stream of `payload` in blocks of `block` inflated bytes (+ the empty EOF block)."""
    out = []
    for off in list(range(0, len(payload), block)) + [None]:
        chunk = b"" if off is None else payload[off:off + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(cdata) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                   + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    return b"".join(out)


def bam_bytes(refs, reads):
    """reads: dict(ref, pos, name, mapq, flag, cigar 'str', seq 'str', qual [..], aux bytes)."""
    import re
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    buf = [b"BAM\1", struct.pack("<i", len(text)), text, struct.pack("<i", len(refs))]
    for name, ln in refs:
        buf.append(struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln))
    for r in reads:
        ops = [(int(n), CIG[c]) for n, c in re.findall(r"(\d+)([MIDNSHP=X])", r.get("cigar", ""))]
        seq = r["seq"]
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i // 2] |= NT16[c] << (4 if i % 2 == 0 else 0)
        qual = bytes(r.get("qual", [30] * len(seq)))
        name = r["name"].encode() + b"\0"
        body = (struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), r.get("mapq", 60), 4680, len(ops), r.get("flag", 0),
                            len(seq), -1, -1, 0)
                + name + b"".join(struct.pack("<I", (n << 4) | o) for n, o in ops) + bytes(packed) + qual + r.get("aux", b""))
        buf.append(struct.pack("<i", len(body)) + body)
    return b"".join(buf)


def same_batch(a, b):
    for f in _abi.ReadBatch.FIELDS + ["start0", "len", "read_begin", "ref", "col_off"]:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert getattr(a, f).dtype == getattr(b, f).dtype, f
    assert a.names == b.names


@pytest.mark.parametrize("threads", [1, 8])
def test_demo_bam_native_vs_python(threads):
    refs, recs = bamio.read_bam(DEMO)
    nb = bamio.NativeBam(DEMO, threads)
    assert nb.refs == refs and nb.n_records == len(recs)
    keep = [r for r in recs if bamio.passes_filter(r, **_abi.READ_FILTER)]
    rid = keep[0]["ref_id"]
    s, e = nb.spans(rid, **_abi.READ_FILTER)
    assert np.array_equal(s, [r["pos"] for r in keep]) and np.array_equal(e, [r["pos"] + max(r["ref_len"], 1) for r in keep])
    assert nb.spans(rid + 1, **_abi.READ_FILTER)[0].size == 0
    (start0, length, _), = bamio.discover_regions(keep, rid, refs[rid][1])
    ref = helpers.load_pseudo_ref()
    same_batch(bamio.build_batch(keep, [(start0, length)], [ref]), nb.batch(rid, [(start0, length)], [ref], **_abi.READ_FILTER))
    # several regions: a read that overlaps two windows is listed in both (fetch rule of util.rs:637), an
    # uncovered window is empty
    cuts = [(start0, 4000), (start0 + 4000, 3000), (start0 + 7000, length - 7000), (start0 + length + 50000, 100)]
    wins = [ref[0:4000], ref[4000:7000], ref[7000:], np.full(100, ord("N"), np.uint8)]
    a, b = bamio.build_batch(keep, cuts, wins), nb.batch(rid, cuts, wins, **_abi.READ_FILTER)
    same_batch(a, nb.batch(rid, cuts, wins, copy=False, **_abi.READ_FILTER))   # views of the decoder's buffers, as the C ABI hands them out
    same_batch(a, b)
    assert b.n_reads > len(keep) and b.read_begin[-1] == b.read_begin[-2]
    # a different filter
    flt = dict(min_mapq=0, min_read_length=2000, divergence=0.02)
    keep2 = [r for r in recs if bamio.passes_filter(r, **flt) and r["ref_id"] == rid]
    same_batch(bamio.build_batch(keep2, cuts[:3], wins[:3]), nb.batch(rid, cuts[:3], wins[:3], **flt))
    nb.close()


def test_handwritten_bam_format_corners(tmp_path):
    f32 = lambda v: struct.pack("<f", v)
    reads = [
        # hard clip before the soft clip; odd l_seq; IUPAC codes; de:f below the cut; ts '+'
        dict(ref=0, pos=100, name="r1", cigar="3H5S20M2I10M1D7M4S2H", seq="ACGTN" + "ACGTRYKMSWBDHV" * 3 + "ACG", aux=b"def" + f32(0.01) + b"tsA+"),
        # de of another type is ignored (util.rs:660-666 looks at Aux::Float only); ts '-' ; B array, Z string, ints
        dict(ref=0, pos=100, name="r2", cigar="30M", seq="A" * 30, flag=16,
             aux=b"deC\x63" + b"NMi" + struct.pack("<i", 3) + b"MDZ10A19\0" + b"xxBs" + struct.pack("<I", 3) + struct.pack("<hhh", 1, -2, 3) + b"tsA-"),
        dict(ref=0, pos=105, name="dropped_de", cigar="30M", seq="C" * 30, aux=b"def" + f32(0.5)),            # de >= divergence
        dict(ref=0, pos=106, name="dropped_mapq", cigar="30M", seq="C" * 30, mapq=5),
        dict(ref=0, pos=107, name="dropped_unmapped", cigar="30M", seq="C" * 30, flag=4),
        dict(ref=0, pos=108, name="dropped_secondary", cigar="30M", seq="C" * 30, flag=256),
        dict(ref=0, pos=109, name="dropped_supp", cigar="30M", seq="C" * 30, flag=2048),
        dict(ref=0, pos=110, name="dropped_short", cigar="9M", seq="C" * 9),
        # spliced read; = and X ops; ts of a non-A type is not a ts tag
        dict(ref=0, pos=120, name="r3", cigar="10=1X500N12M", seq="G" * 23, aux=b"tsi" + struct.pack("<i", 1)),
        # no CIGAR at all: reference length 0 -> bam_endpos = pos + 1
        dict(ref=0, pos=700, name="r4_nocigar", cigar="", seq="T" * 12),
        dict(ref=0, pos=701, name="r5", cigar="12S12M", seq="T" * 24, qual=list(range(24))),
        dict(ref=1, pos=5, name="other_contig", cigar="40M", seq="A" * 40),
    ]
    refs = [("chrA", 5000), ("chrB", 900)]
    raw = bam_bytes(refs, reads)
    for block in (64, 300, 60000):   # records span many blocks / everything in one block
        path = str(tmp_path / ("t%d.bam" % block))
        open(path, "wb").write(bgzf(raw, block))
        prefs, recs = bamio.read_bam(path)
        nb = bamio.NativeBam(path, 3)
        assert nb.refs == prefs == refs and nb.n_records == len(reads)
        flt = dict(min_mapq=20, min_read_length=10, divergence=0.5)
        for rid in (0, 1):
            keep = [r for r in recs if r["ref_id"] == rid and bamio.passes_filter(r, **flt)]
            s, e = nb.spans(rid, **flt)
            assert np.array_equal(s, [r["pos"] for r in keep]) and np.array_equal(e, [r["pos"] + max(r["ref_len"], 1) for r in keep])
            regions = [(90, 100), (190, 400), (590, 200)] if rid == 0 else [(0, 900)]
            wins = [np.full(l, ord("A"), np.uint8) for _, l in regions]
            a, b = bamio.build_batch(keep, regions, wins), nb.batch(rid, regions, wins, **flt)
            same_batch(a, b)
            if rid == 0:
                assert a.names == ["r1", "r2", "r3", "r3", "r3", "r4_nocigar", "r5"]   # r3's intron spans the middle window
                assert list(b.lead_clip[:2]) == [5, 0] and list(b.trail_clip[:2]) == [4, 0]
                assert list(b.flags) == [2, 1 | 4, 0, 0, 0, 0, 0]
        nb.close()


def test_long_cigar_in_the_cg_tag(tmp_path):
    """> 65535 ops do not fit the core field: BAM then stores the placeholder <l_seq>S<ref_len>N and the real CIGAR in a
    CG:B,I tag (SAM spec 4.2.2); htslib -- the reference's reader -- restores it (bam_tag2cigar), so does liblcr.  A placeholder
    without the tag stays what it says, as in htslib."""
    real = [(10, 0), (1, 1), (12, 0), (300, 3), (8, 0), (2, 2), (9, 0)]          # 10M1I12M300N8M2D9M: l_seq 40, ref 341
    tag = b"CGBI" + struct.pack("<I", len(real)) + b"".join(struct.pack("<I", (n << 4) | o) for n, o in real)
    reads = [dict(ref=0, pos=50, name="long", cigar="40S341N", seq="ACGT" * 10, aux=b"tsA+" + tag + b"NMi" + struct.pack("<i", 2)),
             dict(ref=0, pos=60, name="plain", cigar="40M", seq="TTGCA" * 8)]
    refs = [("chrA", 2000)]
    path = str(tmp_path / "cg.bam")
    open(path, "wb").write(bgzf(bam_bytes(refs, reads), 300))
    prefs, recs = bamio.read_bam(path)
    assert [(int(w) >> 4, int(w) & 15) for w in recs[0]["cigar"]] == real and recs[0]["ref_len"] == 341 and recs[0]["lead"] == 0
    nb = bamio.NativeBam(path, 2)
    flt = dict(min_mapq=20, min_read_length=10, divergence=0.5)
    s, e = nb.spans(0, **flt)
    assert list(s) == [50, 60] and list(e) == [50 + 341, 100]
    regions = [(40, 400)]
    wins = [np.full(400, ord("A"), np.uint8)]
    a, b = bamio.build_batch([r for r in recs if bamio.passes_filter(r, **flt)], regions, wins), nb.batch(0, regions, wins, **flt)
    same_batch(a, b)
    assert list(b.n_cig) == [7, 1] and [(int(w) >> 4, int(w) & 15) for w in b.cigar[:7]] == real
    assert bytes(b.bases[:40]) == b"ACGT" * 10
    nb.close()
    bad = str(tmp_path / "cg_missing.bam")
    reads[0]["aux"] = b"tsA+"
    open(bad, "wb").write(bgzf(bam_bytes(refs, reads), 300))
    # htslib (bam_tag2cigar) leaves such a record as it is: one soft clip + one intron, which covers no column
    nb = bamio.NativeBam(bad, 1)
    b = nb.batch(0, [(0, 1000)], [np.full(1000, ord("A"), np.uint8)], **flt)
    _, recs2 = bamio.read_bam(bad)
    assert list(b.n_cig) == [2, 1] and [(int(w) >> 4, int(w) & 15) for w in b.cigar[:2]] == [(40, 4), (341, 3)]
    assert recs2[0]["cigar"].tolist() == b.cigar[:2].tolist()
    nb.close()


def test_decoder_holds_one_contig_at_a_time(tmp_path):
    """The file is mapped and only ONE contig's inflated bytes + record index are resident (the reference streams region
    by region through the .bai): three contigs made of 20 / 10 / 5 copies of demo.bam's records (128 MB inflated) never
    need more than the largest contig (+ its index) or the 64 MiB scan window; batches and the phased-BAM writer give
    what the all-in-memory restatement gives."""
    src = os.path.join(helpers.GOLDEN, "demo.bam")
    raw = bamio.bgzf_decompress(src)
    p = 8 + struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", raw, p)[0]
    body = bytearray(raw[p:])
    hdr = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 3)
    for nm in (b"cA", b"cB", b"cC"):
        hdr += struct.pack("<i", len(nm) + 1) + nm + b"\0" + struct.pack("<i", 64444167)
    offs, q = [], 0
    while q < len(body):
        offs.append(q)
        q += 4 + struct.unpack_from("<i", body, q)[0]

    def contig(rid, copies):   # every record `copies` times in a row: still sorted by position
        out = bytearray()
        for k, o in enumerate(offs):
            e = offs[k + 1] if k + 1 < len(offs) else len(body)
            rec = bytearray(body[o:e])
            struct.pack_into("<i", rec, 4, rid)
            out += rec * copies
        return bytes(out)
    parts = [contig(0, 20), contig(1, 10), contig(2, 5)]
    path = str(tmp_path / "three.bam")
    # (fast deflate: the test is about memory, not compression)
    payload = hdr + b"".join(parts)
    blocks = []
    for off in list(range(0, len(payload), 65280)) + [None]:
        chunk = b"" if off is None else payload[off:off + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        blocks.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(cdata) + 8 - 1)
                      + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    open(path, "wb").write(b"".join(blocks))
    total = len(payload)
    assert total > 120e6
    nb = bamio.NativeBam(path, 8, keep_bytes=0)     # (the bounded-memory form; lcr_bam_open keeps a file of this size inflated: below)
    assert nb.n_records == len(offs) * 35
    now, peak = nb.resident()
    assert now <= 2 ** 20, now                                     # open validates in the scan window: no contig is resident afterwards
    limit = max(len(parts[0]) * 1.15, 64 * 2 ** 20 + 2 ** 20)     # largest contig + ~80 B per record, or the scan window
    assert peak <= limit < 0.7 * total, (peak, limit, total)
    flt = dict(min_mapq=20, min_read_length=500, divergence=0.5)
    s, e = nb.spans(1, **flt)
    assert nb.resident()[0] <= len(parts[1]) * 1.2 + 2 ** 20 and len(s) > 10000
    regions = [(16729960, 13256)]
    win = [np.full(13256, ord("N"), np.uint8)]
    b2 = nb.batch(2, regions, win, **flt)
    assert nb.resident()[0] <= len(parts[2]) * 1.2 + 2 ** 20
    refs, recs = bamio.read_bam(src)
    keep = [r for r in recs if bamio.passes_filter(r, **flt)]
    one = bamio.build_batch(keep, regions, win)
    assert b2.n_reads == 5 * one.n_reads and np.array_equal(b2.pos[::5], one.pos) and np.array_equal(b2.bases[:one.seq_len[0]], one.bases[:one.seq_len[0]])
    # the writer streams: contig A's records (20 copies) with an HP tag on one read name
    out = str(tmp_path / "phased.bam")
    name = recs[10]["name"]
    nb.write_phased(out, [(0, 16729960, 13256)], [name], [1], [777], level=1)
    assert nb.resident()[1] <= limit + 17 * 2 ** 20              # + the 16 MiB output chunk
    orefs, orecs = bamio.read_bam(out, keep_raw=True)
    want = [r for r in recs if not (r["flag"] & (0x4 | 0x100 | 0x800)) and r["pos"] + 1 >= 16729961 and r["pos"] + max(r["ref_len"], 1) + 1 <= 16729961 + 13256]
    assert len(orecs) == 20 * len(want) and [r["name"] for r in orecs[::20]] == [r["name"] for r in want]
    tagged = [r for r in orecs if r["name"] == name]
    assert tagged and all(b"HPi" + struct.pack("<i", 1) in r["raw"][r["aux_off"]:] and b"PSI" + struct.pack("<I", 777) in r["raw"][r["aux_off"]:] for r in tagged)
    # a file within keep_bytes (lcr_bam_open: 4 GiB) is inflated ONCE, at open, and stays inflated: same spans, batches, output
    nk = bamio.NativeBam(path, 8)
    assert total <= nk.resident()[0] <= total + 2 ** 21
    s2, e2 = nk.spans(1, **flt)
    assert np.array_equal(s2, s) and np.array_equal(e2, e)
    bk = nk.batch(2, regions, win, **flt)
    for f in _abi.ReadBatch.FIELDS + ["read_begin"]:
        assert np.array_equal(getattr(bk, f), getattr(b2, f)), f
    outk = str(tmp_path / "phased_keep.bam")
    nk.write_phased(outk, [(0, 16729960, 13256)], [name], [1], [777], level=1)
    assert bamio.bgzf_decompress(outk) == bamio.bgzf_decompress(out)
    assert nk.resident()[1] <= total + len(parts[0]) * 0.2 + 18 * 2 ** 20
    nk.close()
    nb.close()


def test_interleaved_contigs_fall_back_to_the_record_chain(tmp_path):
    """A file whose contigs interleave (records of c1, c2, c1): the record-size table of the open pass only indexes contigs
    whose records lie back to back; the others are found by walking the block_size chain of their range, as before."""
    recs = [dict(ref=0, pos=10, name="a", cigar="20M", seq="ACGT" * 5), dict(ref=1, pos=5, name="b", cigar="12M", seq="ACG" * 4),
            dict(ref=0, pos=40, name="c", cigar="8M2D8M", seq="ACGT" * 4), dict(ref=1, pos=30, name="d", cigar="16M", seq="ACGT" * 4),
            dict(ref=1, pos=31, name="e", cigar="16M", seq="TTGA" * 4)]
    p = str(tmp_path / "mixed.bam")
    open(p, "wb").write(bgzf(bam_bytes([("c1", 1000), ("c2", 1000)], recs), 90))
    flt = dict(min_mapq=0, min_read_length=1, divergence=2.0)
    nb = bamio.NativeBam(p, 4)
    assert nb.n_records == 5
    s0, e0 = nb.spans(0, **flt)
    s1, e1 = nb.spans(1, **flt)
    assert s0.tolist() == [10, 40] and e0.tolist() == [30, 58]
    assert s1.tolist() == [5, 30, 31] and e1.tolist() == [17, 46, 47]
    s0b, _ = nb.spans(0, **flt)                       # and back: contigs are loaded on demand, one at a time
    assert s0b.tolist() == [10, 40]
    b = nb.batch(1, [(0, 100)], [np.full(100, ord("A"), np.uint8)], **flt)
    assert b.n_reads == 3 and b.names == ["b", "d", "e"]
    nb.close()


def test_bad_inputs_are_errors_not_crashes(tmp_path):
    with pytest.raises(_lib.LcrError, match="cannot open"):
        bamio.NativeBam(str(tmp_path / "missing.bam"))
    p = str(tmp_path / "garbage.bam")
    open(p, "wb").write(b"this is not a BGZF file at all, not even close......")
    with pytest.raises(_lib.LcrError, match="BGZF"):
        bamio.NativeBam(p)
    raw = bam_bytes([("c", 100)], [dict(ref=0, pos=1, name="x", cigar="10M", seq="A" * 10)])
    good = bgzf(raw, 64)
    p = str(tmp_path / "trunc.bam")
    open(p, "wb").write(good[:len(good) // 2])
    with pytest.raises(_lib.LcrError):
        bamio.NativeBam(p)
    corrupt = bytearray(good)
    corrupt[40] ^= 0xFF   # inside the first block's deflate data
    p = str(tmp_path / "crc.bam")
    open(p, "wb").write(bytes(corrupt))
    with pytest.raises(_lib.LcrError):
        bamio.NativeBam(p)
    p = str(tmp_path / "notbam.bam")
    open(p, "wb").write(bgzf(b"SAM\1" + b"\0" * 40, 64))
    with pytest.raises(_lib.LcrError, match="not a BAM"):
        bamio.NativeBam(p)
    # malformed fixed fields (l_read_name = 0) are refused at open, in the same pass that inflates the file once
    l_text = struct.unpack_from("<i", raw, 4)[0]
    rec0 = 4 + 4 + l_text + 4 + (4 + 2 + 4)
    assert struct.unpack_from("<i", raw, rec0 + 4)[0] == 0 and raw[rec0 + 12] == 2    # refID 0, l_read_name = len("x\0")
    bad = bytearray(raw); bad[rec0 + 12] = 0
    p = str(tmp_path / "badrec.bam")
    open(p, "wb").write(bgzf(bytes(bad), 64))
    with pytest.raises(_lib.LcrError, match="malformed record 0"):
        bamio.NativeBam(p)
    # unsorted file: refused when a batch is cut (the reference needs a sorted, indexed file as well)
    raw = bam_bytes([("c", 1000)], [dict(ref=0, pos=50, name="a", cigar="10M", seq="A" * 10), dict(ref=0, pos=10, name="b", cigar="10M", seq="A" * 10)])
    p = str(tmp_path / "unsorted.bam")
    open(p, "wb").write(bgzf(raw, 64))
    nb = bamio.NativeBam(p)
    with pytest.raises(_lib.LcrError, match="sorted"):
        nb.batch(0, [(0, 100)], [np.zeros(100, np.uint8)], min_mapq=0, min_read_length=1, divergence=1.0)


def test_phased_bam_writer(tmp_path):
    """SURVEY §8(f) N4 (thread.rs:307-361): lcr_bam_write_phased against the record-by-record restatement, on the
    inflated stream; the output is a valid BAM again (both readers take it)."""
    f32 = lambda v: struct.pack("<f", v)
    reads = [
        dict(ref=0, pos=100, name="inside_hp1", cigar="30M", seq="A" * 30),
        dict(ref=0, pos=101, name="inside_hp0", cigar="30M", seq="A" * 30, aux=b"NMi" + struct.pack("<i", 1)),
        dict(ref=0, pos=102, name="has_hp_already", cigar="30M", seq="A" * 30, aux=b"HPi" + struct.pack("<i", 2) + b"xyZabc\0"),
        dict(ref=0, pos=103, name="has_ps_already", cigar="30M", seq="A" * 30, aux=b"abBc" + struct.pack("<I", 2) + b"\1\2" + b"PSI" + struct.pack("<I", 7)),
        dict(ref=0, pos=104, name="secondary", cigar="30M", seq="A" * 30, flag=256),
        dict(ref=0, pos=105, name="low_mapq_is_written_too", cigar="30M", seq="A" * 30, mapq=0),
        dict(ref=0, pos=106, name="not_in_maps", cigar="30M", seq="A" * 30),
        dict(ref=0, pos=180, name="sticks_out_right", cigar="30M", seq="A" * 30),       # ends beyond the region
        dict(ref=0, pos=199, name="ends_exactly", cigar="1M", seq="A"),                  # reference_end + 1 == region.end
        dict(ref=0, pos=300, name="second_region", cigar="10M5D10M", seq="C" * 20),
        dict(ref=0, pos=300, name="inside_hp1", cigar="20M", seq="C" * 20),              # same name again: first entry counts
        dict(ref=1, pos=10, name="other_contig", cigar="25M", seq="G" * 25, aux=b"def" + f32(0.2)),
    ]
    refs = [("chrA", 5000), ("chrB", 900)]
    src = str(tmp_path / "in.bam")
    open(src, "wb").write(bgzf(bam_bytes(refs, reads), 120))
    regions = [(0, 99, 101), (0, 290, 100), (1, 0, 900), (0, 99, 101)]   # the last repeats the first: written again, as the loop would
    names = ["inside_hp1", "inside_hp0", "has_hp_already", "has_ps_already", "secondary", "low_mapq_is_written_too",
             "sticks_out_right", "ends_exactly", "second_region", "inside_hp1", "other_contig", "no_such_read"]
    hp = [1, 0, 1, 2, 1, 2, 1, -1, 2, 2, 1, 1]
    ps = [101, 0, 101, 101, 101, 0, 101, 200, 301, 999, 11, 5]
    _, recs = bamio.read_bam(src, keep_raw=True)
    want = bamio.phased_stream(recs, regions, names, hp, ps)
    nb = bamio.NativeBam(src, 2)
    for level, threads in ((-1, 1), (1, 4)):
        out = str(tmp_path / ("out%d.bam" % threads))
        nb.write_phased(out, regions, names, hp, ps, level=level, threads=threads)
        assert bamio.bgzf_decompress(out) == want
        assert open(out, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # BGZF EOF marker
        prefs, precs = bamio.read_bam(out)
        nb2 = bamio.NativeBam(out, 2)
        assert prefs == nb2.refs == refs and nb2.n_records == len(precs)
        got = [r["name"] for r in precs]
        assert got == ["inside_hp1", "inside_hp0", "has_hp_already", "has_ps_already", "low_mapq_is_written_too", "not_in_maps", "ends_exactly",
                       "second_region", "inside_hp1", "other_contig",
                       "inside_hp1", "inside_hp0", "has_hp_already", "has_ps_already", "low_mapq_is_written_too", "not_in_maps", "ends_exactly"]
        nb2.close()
    # demo.bam with made-up tags for every third read
    refs, recs = bamio.read_bam(DEMO, keep_raw=True)
    keep = [r for r in recs if bamio.passes_filter(r, **_abi.READ_FILTER)]
    rid = keep[0]["ref_id"]
    (start0, length, _), = bamio.discover_regions(keep, rid, refs[rid][1])
    names = [r["name"] for r in keep[::3]]
    hp = [(i % 3) for i in range(len(names))]
    ps = [(start0 + 1 if i % 2 else 0) for i in range(len(names))]
    nb = bamio.NativeBam(DEMO, 4)
    out = str(tmp_path / "demo_phased.bam")
    nb.write_phased(out, [(rid, start0, length)], names, hp, ps)
    assert bamio.bgzf_decompress(out) == bamio.phased_stream(recs, [(rid, start0, length)], names, hp, ps)


def _aux_tags(raw, q):
    """{tag: value} of a BAM record's aux block (the integer / float / string types the tests write)"""
    out = {}
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while q + 3 <= len(raw):
        tag, typ = raw[q:q + 2].decode(), chr(raw[q + 2])
        q += 3
        if typ == "A":
            out[tag] = chr(raw[q]); q += 1
        elif typ in fmt:
            out[tag] = struct.unpack_from(fmt[typ], raw, q)[0]; q += struct.calcsize(fmt[typ])
        elif typ in "ZH":
            e = raw.index(b"\0", q); out[tag] = raw[q:e].decode(); q = e + 1
        elif typ == "B":
            sub, cnt = chr(raw[q]), struct.unpack_from("<I", raw, q + 1)[0]
            q += 5 + cnt * struct.calcsize(fmt[sub]); out[tag] = "array"
        else:
            raise ValueError(typ)
    return out


def test_phased_bam_against_the_oracle_of_the_loop(tmp_path):
    """lcr_bam_write_phased against oracle_np.phased_bam_records -- thread.rs:307-361 restated on plain tuples in oracle/ (which
    records, in which order, which tags): boundary rule `reference_start + 1 >= start && reference_end + 1 <= end`
    (thread.rs:340-345), first-entry-wins maps, tags already present, a record inside two regions, deletions / introns in the
    reference length, a zero-length alignment (htslib: reference_end = start + 1), 400 random cases and demo.bam."""
    from oracle import oracle_np
    rng = np.random.default_rng(17)

    def check(src, regions, names, hp, ps):
        _, recs = bamio.read_bam(src, keep_raw=True)
        tup = []
        for r in recs:
            t = _aux_tags(r["raw"], r["aux_off"])
            tup.append((r["ref_id"], r["pos"], r["pos"] + (r["ref_len"] if r["ref_len"] > 0 else 1), r["flag"], r["name"], "HP" in t, "PS" in t))
        want = oracle_np.phased_bam_records(tup, [(rid, s0 + 1, s0 + ln + 1) for rid, s0, ln in regions],
                                            [(n, h) for n, h in zip(names, hp) if h >= 0], [(n, p) for n, p in zip(names, ps) if p != 0])
        out = str(tmp_path / "o.bam")
        nb = bamio.NativeBam(src, 2)
        nb.write_phased(out, regions, names, hp, ps, level=1, threads=2)
        nb.close()
        _, got = bamio.read_bam(out, keep_raw=True)
        assert len(got) == len(want)
        for g, (idx, h, p) in zip(got, want):
            src_r = recs[idx]
            assert (g["name"], g["pos"], g["flag"]) == (src_r["name"], src_r["pos"], src_r["flag"])
            t0, t1 = _aux_tags(src_r["raw"], src_r["aux_off"]), _aux_tags(g["raw"], g["aux_off"])
            exp = dict(t0)
            if h is not None:
                exp["HP"] = h
            if p is not None:
                exp["PS"] = p
            assert t1 == exp, (g["name"], t1, exp)
        return len(want)
    refs = [("chrA", 4000), ("chrB", 1500)]
    total = 0
    for case in range(40):
        reads = []
        for i in range(30):
            ref = int(rng.integers(0, 2))
            pos = int(rng.integers(0, 900))
            kind = int(rng.integers(0, 5))
            cigar = ["40M", "10M30D10M", "15M200N15M", "5S20M5S", "*"][kind]
            ln = {0: 40, 1: 20, 2: 30, 3: 30, 4: 12}[kind]
            aux = [b"", b"HPi" + struct.pack("<i", 2), b"PSI" + struct.pack("<I", 9), b"NMi" + struct.pack("<i", 3)][int(rng.integers(0, 4))]
            flag = [0, 16, 256, 2048, 1024, 512][int(rng.integers(0, 6))]
            reads.append(dict(ref=ref, pos=pos, name="q%d" % int(rng.integers(0, 40)), cigar=cigar, seq="A" * ln, aux=aux, flag=flag))
        reads.sort(key=lambda r: (r["ref"], r["pos"]))
        src = str(tmp_path / ("r%d.bam" % case))
        open(src, "wb").write(bgzf(bam_bytes(refs, reads), 300))
        regions = []
        for _ in range(int(rng.integers(1, 4))):
            s0 = int(rng.integers(0, 600))
            regions.append((int(rng.integers(0, 2)), s0, int(rng.integers(50, 900))))
        names = ["q%d" % int(rng.integers(0, 44)) for _ in range(30)]
        hp = [int(rng.integers(-1, 3)) for _ in names]
        ps = [int(rng.integers(0, 3)) * 501 for _ in names]
        total += check(src, regions, names, hp, ps)
    assert total > 100
    refs, recs = bamio.read_bam(DEMO)
    keep = [r for r in recs if bamio.passes_filter(r, **_abi.READ_FILTER)]
    rid = keep[0]["ref_id"]
    (start0, length, _), = bamio.discover_regions(keep, rid, refs[rid][1])
    names = [r["name"] for r in keep[::2]]
    n = check(DEMO, [(rid, start0, length), (rid, start0 + 2000, 3000)], names, [i % 3 for i in range(len(names))], [(start0 + 1) * (i % 2) for i in range(len(names))])
    assert n > 1000


def test_reads_to_bam_and_back(tmp_path):
    """lcr_bam_write_reads is the inverse of lcr_bam_batch: a synthetic batch written as BAM and decoded again by the
    native decoder (and by the record-by-record Python reader) gives the arrays it was made of -- positions, CIGARs,
    bases, qualities, strands, `ts` tags, soft clips."""
    from longcallr_amd import synth
    for profile in ("ont-cdna", "masseq"):
        b = synth.make_batch(profile, n_genes=3, gene_len=6000, depth=12, seed=21)
        path = str(tmp_path / (profile + ".bam"))
        clen = bamio.write_reads_bam(path, b, "chrT", threads=3)
        nb = bamio.NativeBam(path, 2)
        assert nb.refs == [("chrT", clen)] and nb.n_records == b.n_reads
        back = nb.batch(0, list(zip(b.start0.tolist(), b.len.tolist())), [b.ref[int(b.col_off[g]):int(b.col_off[g + 1])] for g in range(b.n_regions)],
                        min_mapq=0, min_read_length=0, divergence=2.0)
        for f in _abi.ReadBatch.FIELDS + ["read_begin"]:
            assert np.array_equal(getattr(back, f), getattr(b, f)), (profile, f)
        assert back.names[:3] == ["r0", "r1", "r2"]
        refs, recs = bamio.read_bam(path)
        assert len(recs) == b.n_reads and [r["pos"] for r in recs[:50]] == b.pos[:50].tolist()
        nb.close()
