"""Kernels of the last detector pass in a rocprofv3 kernel trace of tools/nets_eager.py (from the last k_conv0 to the OSNet stem
after it), grouped by name: launches, total us.  usage: python tools/detector_kernels.py <kernel_trace.csv> [rows=40]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
stems = [i for i, r in enumerate(rows) if 'k_osnet_stem' in r['Kernel_Name']]
c0 = [i for i, r in enumerate(rows) if 'k_conv0' in r['Kernel_Name']]
if c0:
    s = max(c0)
    e = next((i for i in stems if i > s), len(rows))
else:                                    # a detector whose first convolution is not k_conv0 (yolov7): everything between the last two OSNet passes
    e = stems[-1]                        # except OSNet's own kernels (a few library kernels of its head stay in the listing)
    s = next(i for i in range(stems[-2], e) if 'k_osnet_tail<32, 128, 128' in rows[i]['Kernel_Name']) + 5
rows = [r for i, r in enumerate(rows) if not (s <= i < e) or ('osnet' not in r['Kernel_Name'] and 'k_gate_vec' not in r['Kernel_Name'])]
agg, tot = collections.OrderedDict(), 0.0
for r in rows[s:e]:
    n = r['Kernel_Name'].replace('void ', '')
    n = n[:n.index('(')] if '(' in n else n
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    a = agg.setdefault(n[:70], [0, 0.0]); a[0] += 1; a[1] += d
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n:70s} {c:4d} launches {d:8.1f} us")
print(f"launches {e - s}, sum {tot:.1f} us, span {(int(rows[e - 1]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e3:.1f} us")
