#!/usr/bin/env python3
"""Per-op breakdown of a rocprofv3 kernel trace of bench.py (SqueezeSegV2): tools_trace.py <trace.csv> <launches per micro-batch>"""
import csv, collections, statistics, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'pclseg' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def cam(p): return [p+'/pool-rows',p+'/pool-cols',p+'/squeeze',p+'/excite*']
def fire(p,up=False): return [p+'/squeeze']+([p+'/upconv'] if up else [])+[p+'/expand']
ops=['normalize','conv1']+cam('cam1')+['conv1_skip','pool1']+fire('fire2')+cam('cam2')+fire('fire3')+cam('cam3')+['pool3']+fire('fire4')+fire('fire5')+['pool5']
for f in ('fire6','fire7','fire8','fire9'): ops+=fire(f)
for f in ('fire10','fire11','fire12','fire13'): ops+=fire(f,True)
ops+=['conv14+head']
per=int(sys.argv[2]) if len(sys.argv)>2 else len(ops)
agg=collections.defaultdict(list)
for i,r in enumerate(rows): agg[i%per].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
gaps=[]
for a,b in zip(rows,rows[1:]): gaps.append(int(b['Start_Timestamp'])-int(a['End_Timestamp']))
tot=0
for i in range(per):
    m=statistics.median(agg[i]); tot+=m
    nm=ops[i] if i<len(ops) else '?'
    print(f"{i:2d} {nm:18s} {m/1e3:8.1f} us grid={int(rows[i]['Grid_Size_X'])//256}x{rows[i]['Grid_Size_Y']} lds={rows[i]['LDS_Block_Size']} vgpr={rows[i]['VGPR_Count']} {rows[i]['Kernel_Name'][8:40]}")
print('sum of medians per micro-batch us', tot/1e3, ' median gap ns', statistics.median(gaps), 'launches', len(rows))
