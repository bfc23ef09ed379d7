"""One step of a bench.py run as a per-stream Gantt: consecutive launches of a HIP stream merged into runs (gaps < 3 us), the idle gaps of
the whole GPU marked.  usage: python tools/step_gantt.py <kernel_trace.csv> [step from the end = 4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("void ", "")
    r["k"] = k[:k.index("(")] if "(" in k else k
    r["q"] = r.get("Stream_Id") or r.get("Queue_Id")
rows.sort(key=lambda r: r["s"])
assoc = [r for r in rows if r["k"].startswith("k_assoc")]
t0, t1 = assoc[-back - 1]["s"], assoc[-back]["s"]
win = [r for r in rows if r["e"] > t0 and r["s"] < t1]
print(f"step of {(t1 - t0) / 1e3:.1f} us, {len(win)} launches; times in us from the step's association launch")
qs = sorted({r["q"] for r in win})
for q in qs:
    rs = [r for r in win if r["q"] == q]
    runs = []
    for r in rs:
        if runs and r["s"] - runs[-1][1] < 3000:
            runs[-1][1] = max(runs[-1][1], r["e"]); runs[-1][2].append(r)
        else:
            runs.append([r["s"], r["e"], [r]])
    print(f"stream/queue {q}: {len(rs)} launches, busy {sum(r['e'] - r['s'] for r in rs) / 1e3:.1f} us")
    for s, e, ks in runs:
        big = sorted(ks, key=lambda r: r["s"] - r["e"])[:3]
        print(f"   {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} ({(e - s) / 1e3:7.1f} us, {len(ks):3d} launches)  {ks[0]['k'][:28]} .. {ks[-1]['k'][:28]}   longest: " +
              ", ".join(f"{r['k'][:24]} {(r['e'] - r['s']) / 1e3:.0f}" for r in big))
ev = sorted([(max(r["s"], t0), 1) for r in win] + [(min(r["e"], t1), -1) for r in win])
depth, last = 0, t0
for t, d in ev:
    if depth == 0 and t - last > 4000:
        prev = max((r for r in win if r["e"] <= last + 1), key=lambda r: r["e"], default=None)
        nxt = min((r for r in win if r["s"] >= t - 1), key=lambda r: r["s"], default=None)
        print(f"   idle {(last - t0) / 1e3:8.1f} -> {(t - t0) / 1e3:8.1f} ({(t - last) / 1e3:5.1f} us)  after {prev['k'][:30] if prev else None} [{prev['q'] if prev else None}]  before {nxt['k'][:30] if nxt else None} [{nxt['q'] if nxt else None}]")
    depth += d; last = max(last, t) if depth else t
