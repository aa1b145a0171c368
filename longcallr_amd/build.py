"""Builds liblcr.so (HIP, gfx950 only) in-tree with hipcc. No fallbacks, no other targets."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "liblcr.so")
SOURCES = ["k0_ops.hip", "k1_pileup.hip", "k2_candidates.hip", "k3_fragments.hip", "k4_phase.hip", "k4_enum.hip", "k4_stage.hip", "k4_post.hip", "k4_grid.hip", "k5_regions.hip", "lcr_api.hip",
           "lcr_bam.cpp"]   # lcr_bam.cpp: host-only BGZF / BAM decode (zlib)
HEADERS = ["lcr_dev.h", "lcr_phase_host.h", "k4_dev.h", "k4_types.h", "k4_grid.h", "k4_grid_batch.h", "k4_kernels.h", "k4_post.h", os.path.join("..", "..", "include", "lcr.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    if force or procs or _stale(SO, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", SO] + objs + ["-lz"]
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
