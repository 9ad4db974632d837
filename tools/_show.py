import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["roofline"]["all_kernels"].items() if v["ms"] > 0.5})
