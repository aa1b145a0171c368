"""decode timing vs host threads on the GPU box: C3 as a BAM in /dev/shm"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from longcallr_amd import bamio
b = bench.build_workload("c3")
d = tempfile.mkdtemp(dir="/dev/shm"); p = os.path.join(d, "x.bam")
clen = bamio.write_reads_bam(p, b, "chrS", level=1, threads=0)
want = list(zip(b.start0.tolist(), b.len.tolist())); wins = [b.ref[int(b.col_off[g]):int(b.col_off[g+1])] for g in range(b.n_regions)]
flt = dict(min_mapq=0, min_read_length=0, divergence=2.0)
print("cpus", os.cpu_count(), "thp", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
for nt in ((256, 64) if os.environ.get("LCR_LIB") else (256, 128, 64, 32, 16, 256)):
    best = None
    for rep in range(2):
        t0=time.perf_counter(); nb = bamio.NativeBam(p, nt); t1=time.perf_counter(); rs, re_ = nb.spans(0, **flt); t2=time.perf_counter()
        b2 = nb.batch(0, want, wins, name_format="blob", copy=False, **flt); t3=time.perf_counter(); nb.close()
        t = (t1-t0, t2-t1, t3-t2)
        if best is None or sum(t) < sum(best): best = t
    print("threads %3d  open %.3f spans %.3f batch %.3f" % ((nt,) + best), flush=True)
shutil.rmtree(d)
