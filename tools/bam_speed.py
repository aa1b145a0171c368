"""Throughput of liblcr's BAM decoder (lcr_bam_open + lcr_bam_batch) on demo.bam replicated N times."""
import os, sys, time, struct, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from longcallr_amd import bamio, _abi
import helpers

def bgzf(payload, block=65280, level=6):
    out = []
    for off in list(range(0, len(payload), block)) + [None]:
        chunk = b"" if off is None else payload[off:off + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(cdata) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                   + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    return b"".join(out)

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
src = os.path.join(helpers.GOLDEN, "demo.bam")
raw = bamio.bgzf_decompress(src)
l_text = struct.unpack_from("<i", raw, 4)[0]
p = 8 + l_text
n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
for _ in range(n_ref):
    l_name = struct.unpack_from("<i", raw, p)[0]; p += 8 + l_name
body = raw[p:]
path = "/tmp/lcr_bam_speed_%d.bam" % rep
if not os.path.exists(path):
    open(path, "wb").write(bgzf(raw[:p] + body * rep))   # (records repeat in blocks: still sorted per copy only -> batch() is not used on it)
size, inflated = os.path.getsize(path), p + len(body) * rep
for th in (1, 8, 32, 64):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); nb = bamio.NativeBam(path, th); dt = time.perf_counter() - t; nb.close(); best = min(best, dt)
    print("threads %2d: open (inflate + CRC + record index) %.3f s = %.0f MB/s compressed, %.0f MB/s inflated, %d records"
          % (th, best, size / best / 1e6, inflated / best / 1e6, 1713 * rep))
nb = bamio.NativeBam(src, 8)
refs, recs = bamio.read_bam(src)
keep = [r for r in recs if bamio.passes_filter(r, **_abi.READ_FILTER)]
rid = keep[0]["ref_id"]
(start0, length, _), = bamio.discover_regions(keep, rid, refs[rid][1])
t = time.perf_counter()
for _ in range(20): b = nb.batch(rid, [(start0, length)], [helpers.load_pseudo_ref()], **_abi.READ_FILTER)
print("demo batch (1697 reads, 2.2 MB of bases): %.2f ms per lcr_bam_batch + numpy copies" % ((time.perf_counter() - t) / 20 * 1e3))
