#!/bin/bash
# SQ counters of the draw kernel (wide and, with CUDABROT_AMD_NO_WIDE=1, draw_wave_kernel) from a short bench run.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
B="--steps 4 --warmup 1 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
for v in ${VARIANTS:-wide nowide}; do
  rm -rf gpurun_out/pmc_$v
  if [ $v = nowide ]; then export CUDABROT_AMD_DEBUG=1 CUDABROT_AMD_NO_WIDE=1; else unset CUDABROT_AMD_NO_WIDE; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE} --output-format csv -d gpurun_out/pmc_$v -- python3 bench.py $B > gpurun_out/pmc_$v.log 2>&1
  echo "== $v exit $?"
  python3 - gpurun_out/pmc_$v <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(d+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'draw_w' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
dur=[]
for f in glob.glob(d+'/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'draw_w' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
med=lambda v: sorted(v)[len(v)//2] if v else 0
ms=med(dur)
print('dispatch ms median %.3f  (%s)'%(ms,' '.join('%.2f'%x for x in dur)))
for k,v in sorted(agg.items()):
    print('  %-22s %.4g'%(k,med(v)))
if agg.get('SQ_INSTS_VALU') and ms:
    iv=med(agg['SQ_INSTS_VALU']); clk=med(agg['GRBM_GUI_ACTIVE'])/8/(ms*1e-3)/1e9
    print('  clock %.3f GHz; VALU busy (x4 cycles) %.3f at that clock'%(clk, iv*4/(1024*ms*1e-3*clk*1e9)))
PY
  find gpurun_out/pmc_$v -type f ! -name '*.csv' -delete; find gpurun_out/pmc_$v -name '*.csv' -size +2M -delete
done
echo PMC DONE
