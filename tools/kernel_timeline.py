"""Start, end and duration (ms) of every kernel of a rocprofv3 --kernel-trace run (gpurun_out/tl/), in order of start, with
its stream: what ran beside what.  usage: kernel_timeline.py [from_ms to_ms]"""
import csv,glob,re,sys
f=glob.glob('gpurun_out/tl/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the overlapped phase: last 40 kernels where draw overlaps others
t0=int(rows[0]['Start_Timestamp'])
out=[]
for r in rows:
    m=re.search(r'(\w+_kernel|__amd\w+|\w+)(<[^(]*>)?\(', r['Kernel_Name'].replace('(anonymous namespace)::',''))
    n=(m.group(1) if m else r['Kernel_Name'])[:34]
    s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
    out.append((s,e,n,r.get('Stream_Id','?')))
# print the window: kernels within the longest run of steps: choose by argument
lo=float(sys.argv[1]) if len(sys.argv)>1 else 0
hi=float(sys.argv[2]) if len(sys.argv)>2 else 1e9
for s,e,n,st in out:
    if lo<=s<=hi: print('%10.3f %10.3f %8.3f  %s %s'%(s,e,e-s,n,st))
