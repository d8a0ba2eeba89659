#!/bin/bash
# round 4, visit K: does instruction fetch limit the megakernels?  (K11's code is ~300 KB; the instruction cache is 64 KB per two CUs.)  SQC / SQ instruction-fetch counters of the Cornell ReSTIR PT frame.
mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_FETCH|SQC_" | cut -c1-160 | sort -u | head -60 > $R/gpurun_out/r04k_avail.txt
cat $R/gpurun_out/r04k_avail.txt | head -40
CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --no-cpu-baseline --no-extra-workloads"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_ICACHE_MISSES_DUPLICATE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/k_d; timeout 600 rocprofv3 --pmc $set -d /tmp/k_d -- $CMD > $R/gpurun_out/r04k_$tag.log 2>&1
  DB=$(find /tmp/k_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB $R/gpurun_out/r04k_$tag.csv > /dev/null; grep -E "pathtrace|temporal|stc|gbuffer" $R/gpurun_out/r04k_$tag.csv | head -30; else tail -5 $R/gpurun_out/r04k_$tag.log; fi
done
