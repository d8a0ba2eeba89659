#!/bin/bash
# round 4, visit P: VALU wave-instructions per launch of the tolerance-mode build (libzetaray_amd_fast.so) next to the contract build's (profiles/r04_pmc_*.json),
# Cornell + atrium 1080p -- VERDICT r3 item 3 asked for the instruction counts, not only the times
mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY"
for wl in cornell atrium; do
  ARGS=""; [ $wl = atrium ] && ARGS="--config 4"
  CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --no-cpu-baseline --no-extra-workloads --arith fast $ARGS"
  rm -rf /tmp/p_d; timeout 600 rocprofv3 --pmc $SQ_A -d /tmp/p_d -- $CMD > $R/gpurun_out/r04p_fast_$wl.log 2>&1
  DB=$(find /tmp/p_d -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py pmc $DB $R/gpurun_out/r04p_fast_${wl}_sqA.csv > /dev/null
  grep -E "pathtrace|temporal|stc|gbuffer|replay" $R/gpurun_out/r04p_fast_${wl}_sqA.csv | grep "SQ_INSTS_VALU\|SQ_THREAD_CYCLES_VALU" | cut -c1-120
done
