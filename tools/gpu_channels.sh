#!/bin/bash
# Fused three-channel render against three separate runs (same passes), C3-sized and C5-sized canvases.
set -u
mkdir -p gpurun_out
one() { local log=$1; shift; timeout -k 10 300 ./cudabrot "$@" > "$log" 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi; grep "passes took" "$log"; }
P=${PASSES:-1024}
echo "== 4096x4096, windows (20000,2000) (2000,200) (200,20), $P passes"
one gpurun_out/ch_fused.log --passes $P -w 4096 -h 4096 --channel 20000:2000:/dev/null --channel 2000:200:/dev/null --channel 200:20:/dev/null
one gpurun_out/ch_a.log --passes $P -w 4096 -h 4096 -m 20000 -c 2000 -o /dev/null
one gpurun_out/ch_b.log --passes $P -w 4096 -h 4096 -m 2000 -c 200 -o /dev/null
one gpurun_out/ch_c.log --passes $P -w 4096 -h 4096 -m 200 -c 20 -o /dev/null
echo "== 20000x15000 on [-2,2]x[-1.5,1.5], windows (60000,45000) (8000,1000) (500,20), $P passes (the colour recipe)"
C5="-w 20000 -h 15000 --min-imag -1.5 --max-imag 1.5"
one gpurun_out/c5_fused.log --passes $P $C5 --channel 60000:45000:/dev/null --channel 8000:1000:/dev/null --channel 500:20:/dev/null
one gpurun_out/c5_a.log --passes $P $C5 -m 60000 -c 45000 -o /dev/null
one gpurun_out/c5_b.log --passes $P $C5 -m 8000 -c 1000 -o /dev/null
one gpurun_out/c5_c.log --passes $P $C5 -m 500 -c 20 -o /dev/null
