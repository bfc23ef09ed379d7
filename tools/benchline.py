"""One-line digest of a bench.py JSON line read from stdin (value, ms/step, k_assoc launch, exactness, the secondary legs)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
g = lambda k: (d.get(k) or {}).get("frames_per_s") or (d.get(k) or {}).get("calls_per_s")
print(d["value"], "ms/step", d["ms_per_step"], "assoc us", r.get("mean_launch_us"), "frac", r.get("frac"), "exact", d.get("frames_bit_exact"),
      "f16", g("throughput_mode"), "all32", g("all_fp32"), "api", g("api_path"), "api16", g("api_path_f16"))
