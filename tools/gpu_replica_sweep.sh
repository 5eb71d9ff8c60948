#!/bin/bash
# Same-box sweep of the level-A replica count of the two-level sort (scatter.hip kReplicas) at C4.
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
mkdir -p gpurun_out
for R in "$@"; do
  sed -i "s/^constexpr uint32_t kReplicas = [0-9]*;/constexpr uint32_t kReplicas = $R;/" cudabrot_amd/csrc/scatter.hip
  (cd cudabrot_amd/csrc && make -j8 -s > ../../gpurun_out/replica_build_$R.log 2>&1) || { echo "build failed for $R"; exit 1; }
  for rep in 1 2 3; do
    timeout -k 10 200 ./cudabrot --passes 1280 -w 20000 -h 20000 -m 20000 -o /dev/null > gpurun_out/replica_$R.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "run failed ($rc) at $R"; tail -3 gpurun_out/replica_$R.log; exit 1; fi
    echo "replicas $R: $(grep 'passes took' gpurun_out/replica_$R.log)"
  done
done
