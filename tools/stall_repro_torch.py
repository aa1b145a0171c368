"""tools/stall_repro_torch.py -- is the 70 ms stall liblcr's?  The same rhythm WITHOUT liblcr: per pass a 1 GB page-locked host -> device copy
(what bench.to_device does), then 60 "steps" of three ~0.3 ms PyTorch kernels + torch.cuda.synchronize(), every step timed on the host.
variant idle: the copy is replaced by 150 ms of sleep.  (profiles/r06_stall.txt)"""
import json, sys, time
import torch

dev = torch.device("cuda", 0)
a = torch.empty(64 << 20, dtype=torch.float32, device=dev)   # 256 MB
host = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def step():
    a.mul_(1.0001); a.add_(0.5); a.mul_(0.9999)
    torch.cuda.synchronize()


for variant in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["copy", "idle", "copy"]):
    for _ in range(20):
        step()
    stalls, med = [], []
    for p in range(40):
        if variant == "copy":
            dst.copy_(host, non_blocking=True); torch.cuda.synchronize()
        else:
            time.sleep(0.15)
        ts = []
        for i in range(60):
            t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e3)
        m = sorted(ts)[len(ts) // 2]
        med.append(m)
        if max(ts) > 20:
            stalls.append((round(max(ts), 1), ts.index(max(ts))))
    print(json.dumps(dict(variant=variant, passes=40, median_step_ms=round(sorted(med)[20], 3), passes_with_a_stall=len(stalls), stalls_ms_at_step=stalls)), flush=True)
