"""Kernel sequence of the last detector pass in a rocprofv3 kernel trace (name, workgroups, us).  usage: python tools/detector_sequence.py <kernel_trace.csv>"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_v8_decode' in r['Kernel_Name']]
e=idx[-1]
s=max(i for i in range(e) if 'k_conv0' in rows[i]['Kernel_Name'] or 'igemm' in rows[i]['Kernel_Name'])
tot=0
for r in rows[s:e+1]:
    n=r['Kernel_Name'].replace('void ','')
    n=n[:n.index('(')] if '(' in n else n
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    print(f"{n[:34]:34s} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):2d} {d:6.1f}")
print('sum',tot)
