"""prints the A/B of two bench lines: tools/ab_cmp.py gpurun_out/b_new.json gpurun_out/b_old.json"""
import json, sys
for n in sys.argv[1:]:
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "ERR", e); continue
    r = d["roofline"]
    print("%-28s step %.3f ms  value %.3e  pileup frac %.3f (%.3f ms)" % (n, d["ms_per_step"], d["value"], r["frac"], r.get("avg_ms", 0)))
    print("    kernels", {k: round(v, 3) for k, v in d["stages"]["kernel_ms"].items()})
    print("    api    ", {k: round(v, 3) for k, v in d["stages"]["api_ms"].items()})
