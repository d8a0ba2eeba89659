# Round-3 GPU visit A: calibrate the VALU / SALU / LDS issue rates with tools/valu_calib.hip (plain run + two SQ passes),
# then five SQ counter passes over the heavy ReSTIR kernels so that the shares of a SIMD's cycles can be added up.
#   WORKLOADS="rpt_cornell rpt_atrium gi_cornell" bash scripts/gpu_r03_a.sh
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03a}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY"
SQ_B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ_C="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS"
SQ_D="SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ_E="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
SQ_F="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL"

pmc() { # $1 = output stem, $2 = counters, rest = command
  local O=$1 CTR=$2; shift 2
  rm -rf ${O}_d
  timeout 600 rocprofv3 --pmc $CTR -d ${O}_d -- "$@" > ${O}.log 2>&1
  local DB=$(find ${O}_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${O}.csv > /dev/null; else echo "no db for $O"; tail -5 ${O}.log; fi
  rm -rf ${O}_d
}

if [ -z "$SKIP_CALIB" ]; then
  [ -x $R/build/valu_calib ] || hipcc --offload-arch=gfx950 -O2 -o $R/build/valu_calib $R/tools/valu_calib.hip
  timeout 300 $R/build/valu_calib > $R/gpurun_out/${TAG}_valu_calib.jsonl 2>&1
  tail -5 $R/gpurun_out/${TAG}_valu_calib.jsonl
  pmc $R/gpurun_out/${TAG}_calib_sqA "$SQ_A" $R/build/valu_calib
  pmc $R/gpurun_out/${TAG}_calib_sqB "$SQ_B" $R/build/valu_calib
  pmc $R/gpurun_out/${TAG}_calib_sqD "$SQ_D" $R/build/valu_calib
  pmc $R/gpurun_out/${TAG}_calib_sqG "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES" $R/build/valu_calib
fi
[ "$WORKLOADS" = none ] && exit 0

for wl in ${WORKLOADS:-rpt_cornell rpt_atrium gi_cornell}; do
  case $wl in
    rpt_cornell) ARGS="";;
    rpt_atrium) ARGS="--scene synthetic";;
    gi_cornell) ARGS="--integrator restir_gi";;
    gi_atrium) ARGS="--integrator restir_gi --scene synthetic";;
    pt_cornell) ARGS="--integrator pt";;
    pt_atrium) ARGS="--integrator pt --scene synthetic";;
  esac
  CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline $ARGS"
  O=$R/gpurun_out/${TAG}_$wl
  for p in A B C D E F; do
    eval CTR=\$SQ_$p
    pmc ${O}_sq$p "$CTR" $CMD
  done
  if [ -n "$WITH_TRAFFIC" ]; then
    pmc ${O}_fetch FETCH_SIZE $CMD
    pmc ${O}_write WRITE_SIZE $CMD
    rm -rf ${O}_stats
    timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- $CMD > ${O}_stats.log 2>&1
    python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $R/gpurun_out/${TAG}_kernel_stats_$wl.csv > /dev/null
    rm -rf ${O}_stats
  fi
  echo "== $wl done"; ls $R/gpurun_out | grep ${TAG}_$wl | tr '\n' ' '
done
