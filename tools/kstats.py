"""Print a rocprofv3 kernel_stats.csv compactly.  usage: python tools/kstats.py <stats.csv> [replays] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rep = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r['Name'].replace('void ', '')
    n = n[:n.index('(')] if '(' in n else n
    print(f"{n[:60]:60s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:8.1f} us  per-replay {float(r['TotalDurationNs'])/1e3/rep:8.1f} us")
