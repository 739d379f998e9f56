import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(list)
for r in rows:
    for n in ('setup_kernel','raster_kernel','resolve_surface_fast','vertex_kernel'):
        if n in r['Kernel_Name']: d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,v in d.items():
    v=sorted(v); print(n, len(v), 'median %.1f min %.1f'%(v[len(v)//2], v[0]))
