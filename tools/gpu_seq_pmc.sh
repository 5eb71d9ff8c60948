#!/bin/bash
# PMC counters per kernel of sequential launches (tools/seq_profile.py): tools/gpu_seq_pmc.sh C3 [C4 ...]
# Separate rocprofv3 passes (counters only ever beside --kernel-trace, as the pool requires); prints the median
# per dispatch of every counter for every cb:: kernel.  FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE counts a wide
# coalesced read at half its bytes on gfx950, MI355X_MICROARCH.md).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "$@"; do
  n=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; do
    n=$((n+1))
    out=gpurun_out/seqpmc_${cfg}_$n
    rm -rf "$out"
    timeout -k 10 ${LIMIT:-240} rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out" -- python3 tools/seq_profile.py "$cfg" ${LAUNCHES:-3} > "$out.log" 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; tail -5 "$out.log"; exit 1; fi
  done
  echo "== $cfg"
  python3 - "$cfg" <<'PY'
import collections, csv, glob, re, sys
cfg = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/seqpmc_%s_*/*/*_counter_collection.csv" % cfg):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m and "cb::" in r["Kernel_Name"]:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: "%.4g" % sorted(v)[len(v) // 2] for c, v in sorted(d.items())})
PY
  find gpurun_out -path "gpurun_out/seqpmc_${cfg}_*" -type f -size +1M -delete
done
