#!/bin/bash
# Timings of the CLI at the BASELINE shapes; prints passes, samples/s and (timed variant) stage stamps.
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
mkdir -p gpurun_out
one() {  # one <label> <env...> -- <args...>
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local log=gpurun_out/ab_$label.log
  timeout -k 10 120 env "${envs[@]}" ./cudabrot "$@" --stats -o /dev/null > "$log" 2>&1
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $label: stopping"; exit 1; fi
  echo "== $label: $(grep 'passes took' "$log")"
  grep '^{' "$log" | SECS=$(grep 'passes took' "$log" | sed 's/.*took \([0-9.]*\) seconds.*/\1/') python3 -c "
import sys, json, os
for line in sys.stdin:
    c = json.loads(line)
    tot = c['cycles_total']
    if tot:
        print('   stage shares: head %.1f%% long %.1f%% replay %.1f%%' % (100*c['cycles_head']/tot, 100*c['cycles_long']/tot, 100*c['cycles_replay']/tot))
    if c.get('rt_span'):
        n_waves = 4096
        print('   last launch: span %.2f ms, mean wave life %.2f ms (residency %.1f%%), shader clock %.0f MHz' % (c['rt_span']/1e5, c['rt_wave_life_sum']/n_waves/1e5, 100*c['rt_wave_life_sum']/n_waves/c['rt_span'], tot/c['rt_wave_life_sum']*100))
    algo = c['iterate_steps'] + c['replay_steps']
    secs = float(os.environ['SECS'])
    print('   %.2f Gsamples/s; iterations %.3f T algorithmic, %.3f T executed (%.1f%% skipped as exactly periodic); %.1f G increments/s' % (
        c['samples']/secs/1e9, algo/1e12, (algo - c.get('skipped_steps', 0))/1e12, 100.0*c.get('skipped_steps', 0)/algo, c['increments']/secs/1e9))
"
}
C3="-w 4096 -h 4096 -m 20000"
C2="-w 4096 -h 4096 -m 2000"
one c3_p64_timed CUDABROT_AMD_WAVE_DUMP=gpurun_out/wave_dump_c3.bin -- $C3 --passes 64 --kernel timed
one c2_p64_timed CUDABROT_AMD_WAVE_DUMP=gpurun_out/wave_dump_c2.bin -- $C2 --passes 64 --kernel timed
one c3_t5 X=1 -- $C3 -t 5
one c2_t5 X=1 -- $C2 -t 5
one def_t5 X=1 -- -t 5
one c3_full_t5 X=1 -- $C3 -t 5 --kernel full
echo AB DONE
