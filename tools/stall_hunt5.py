"""tools/stall_hunt5.py -- WHEN, relative to the host-side events of a pass, does a stall END?  (profiles/r06_stall.txt)
tools/stall_hunt4.py: a stall needs a fresh upload AND a fresh context, sits in timed steps 0-2, lasts 63 / 73 / 83 ms (10 ms apart: it ends
on a timer tick), and sleeping 2 x 50 ms before the timed steps removes it.  So something armed ~100 ms earlier fires; this tool timestamps
every host-side event of bench.py's side-stage pass and prints, for the stalled passes, how long before the stall's END each one lies."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import bench
from longcallr_amd import _abi, api, synth


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mode = sys.argv[2] if len(sys.argv) > 2 else "no_empty"
    params = _abi.make_params(synth.preset_for("ont-cdna"))
    dev = torch.device("cuda", 0)
    batches = [bench.build_workload("c3", seed=s) for s in (2, 3, 4, 5)]
    ev = {}
    now = time.perf_counter
    n_stall = 0
    for rep in range(reps):
        for b in batches:
            for k in range(2):
                ev["upload_begin"] = now()
                dv = bench.to_device(b, torch, dev)
                ev["upload_returned"] = now()
                torch.cuda.synchronize()
                ev["upload_synced"] = now()
                E = api.Engine(0, params, timing=(_abi.K_SPANS, _abi.K_PILEUP)); E.set_async_phase(True)
                ev["ctx_created"] = now()
                for w in range(4):
                    E.load_batch(dv); E.fill_data_into_freq_vec().get_candidate_snps().get_fragments().phase()
                    ev["warm_step_%d_queued" % w] = now()
                E.sync(); torch.cuda.synchronize()
                ev["warm_synced"] = now()
                stall = None
                for i in range(15):
                    t0 = now()
                    E.load_batch(dv); E.fill_data_into_freq_vec().get_candidate_snps().get_fragments().phase()
                    E.kernel_ms(_abi.K_SPANS) + E.kernel_ms(_abi.K_PILEUP)
                    t1 = now()
                    if (t1 - t0) > 0.02 and stall is None:
                        stall = (i, t0, t1)
                E.sync()
                if stall:
                    n_stall += 1
                    i, t0, t1 = stall
                    print(json.dumps(dict(step=i, stall_ms=round((t1 - t0) * 1e3, 1),
                                          ms_before_stall_end={k2: round((t1 - v) * 1e3, 1) for k2, v in sorted(ev.items(), key=lambda kv: kv[1])})), flush=True)
                ev = {}
                ev["prev_close_begin"] = now()
                E.close()
                ev["prev_close_end"] = now()
                del dv
                if mode != "no_empty":
                    torch.cuda.empty_cache()
                ev["prev_freed"] = now()
    print(json.dumps(dict(passes=reps * 8, stalled=n_stall)))


if __name__ == "__main__":
    main()
