"""k_assoc launches of a rocprofv3 kernel trace, one line per launch (for a run whose launches all have one shape, e.g.
tools/batched_assoc.py 1 32): start-to-end microseconds, grid, and the mean of the last n.  usage: python tools/assoc_trace_filter.py <kernel_trace.csv> [last_n=1]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].replace("void ", "").startswith("k_assoc")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("launch,grid_x,workgroup,start_ns,duration_us")
for i, r in enumerate(rows):
    print(f"{i},{r['Grid_Size_X']},{r['Workgroup_Size_X']},{r['Start_Timestamp']},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.2f}")
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-n:]]
print(f"# mean of the last {n} launch(es) (galleries full): {sum(d) / len(d):.2f} us")
