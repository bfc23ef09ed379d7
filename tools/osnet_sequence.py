"""Kernel sequence of the last OSNet pass in a rocprofv3 kernel trace.  usage: python tools/osnet_sequence.py <kernel_trace.csv>"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_osnet_stem' in r['Kernel_Name']]
s=idx[-1]
e=min(i for i in range(s,len(rows)) if 'Cijk' in rows[i]['Kernel_Name'] or i==len(rows)-1)
tot=0
for r in rows[s:e+3]:
    n=r['Kernel_Name'].replace('void ','')
    n=n[:n.index('(')] if '(' in n else n
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    print(f"{n[:34]:34s} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):2d} {d:6.1f}")
print('sum',tot)
