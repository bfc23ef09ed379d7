"""How busy the GPU is inside the timed region of a bench.py run: union of the kernel intervals of a rocprofv3 kernel trace between
the first and the last of the LAST n k_assoc launches (one per 32-frame step), time with >= 2 kernels in flight, idle gaps, and the
kernel time by name per step.  usage: python tools/trace_busy.py <kernel_trace.csv> [last_n_steps=16]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("void ", "")
    r["k"] = k[:k.index("(")] if "(" in k else k
rows.sort(key=lambda r: r["s"])
assoc = [r for r in rows if r["k"].startswith("k_assoc")]
t0, t1 = assoc[-n - 1]["s"], assoc[-1]["s"]                 # n steps: from one association launch to the n-th after it
win = [r for r in rows if r["e"] > t0 and r["s"] < t1]
ev = []
for r in win:
    ev.append((max(r["s"], t0), 1)); ev.append((min(r["e"], t1), -1))
ev.sort()
busy = multi = 0
depth, last = 0, t0
gaps = []
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    if depth == 0 and t > last: gaps.append(t - last)
    depth += d; last = t
span = t1 - t0
by = collections.Counter()
cnt = collections.Counter()
for r in win:
    by[r["k"]] += min(r["e"], t1) - max(r["s"], t0); cnt[r["k"]] += 1
print(f"window: {n} steps, {span / n / 1e3:.1f} us per step; some kernel running {busy / span:.3f} of the time, >= 2 kernels in flight {multi / span:.3f}, idle {1 - busy / span:.3f}")
g = sorted(gaps, reverse=True)
print(f"idle gaps: {len(g)} per window = {len(g) / n:.1f} per step, total {sum(g) / n / 1e3:.1f} us per step, longest {g[0] / 1e3 if g else 0:.1f} us, gaps > 5 us: {sum(1 for x in g if x > 5000) / n:.1f} per step ({sum(x for x in g if x > 5000) / n / 1e3:.1f} us)")
print(f"sum of kernel durations per step: {sum(by.values()) / n / 1e3:.1f} us")
print("kernel                                              launches/step   us/step")
for k, v in by.most_common(28):
    print(f"{k[:50]:50s} {cnt[k] / n:8.1f} {v / n / 1e3:10.1f}")
