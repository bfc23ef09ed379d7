"""Kernel sequence of the last OSNet pass in a rocprofv3 kernel trace.  usage: python tools/osnet_sequence.py <kernel_trace.csv>"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_osnet_stem' in r['Kernel_Name']]
s=idx[-1]
heads=[i for i in range(s,len(rows)) if 'k_osnet_head' in rows[i]['Kernel_Name']]
if heads:                                # the head is one launch (r04): the pass ends with it; in a pipeline trace other streams' kernels interleave
    e=heads[0]; seg=[r for r in rows[s:e+1] if 'osnet' in r['Kernel_Name'] or 'k_gate_vec' in r['Kernel_Name'] or 'k_pw<' in r['Kernel_Name']]
else:
    e=min(i for i in range(s,len(rows)) if 'Cijk' in rows[i]['Kernel_Name'] or i==len(rows)-1); seg=rows[s:e+3]
tot=0
for r in seg:
    n=r['Kernel_Name'].replace('void ','')
    n=n[:n.index('(')] if '(' in n else n
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    print(f"{n[:34]:34s} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):2d} {d:6.1f}")
print('sum',tot)
