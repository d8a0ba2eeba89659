# Round-4 profile set (TAG=r04 by default): for each workload a --kernel-trace --stats run and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ sets A / B / C / E) of
# the same bench.py command; summaries -> gpurun_out/${TAG:-r04}_*, copied into profiles/ by hand.  tools/pmc_profile.py stamps the kernel sources' hash
# into the JSON, and bench.py only reports traffic / valu from a profile whose hash matches.
#   WORKLOADS="rpt_cornell rpt_atrium gi_cornell" bash scripts/gpu_${TAG:-r04}_profiles.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY"
SQ_B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ_C="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS"
SQ_E="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
pmc() { local O=$1 CTR=$2; shift 2; rm -rf ${O}_d
  timeout 600 rocprofv3 --pmc $CTR -d ${O}_d -- "$@" > ${O}.log 2>&1
  local DB=$(find ${O}_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${O}.csv > /dev/null; else echo "no db for $O"; tail -5 ${O}.log; fi
  rm -rf ${O}_d; }
for wl in ${WORKLOADS:-rpt_cornell rpt_atrium gi_cornell}; do
  case $wl in
    rpt_cornell) ARGS="";;
    rpt_atrium) ARGS="--config 4";;
    rpt_3840x2160_atrium) ARGS="--config 5";;      # (config 5 = 4k + the denoise pass: one profile serves both presets)
    gi_cornell) ARGS="--config 3";;
    gi_atrium) ARGS="--integrator restir_gi --scene synthetic";;
    pt_cornell) ARGS="--config pt";;
    pt_atrium) ARGS="--integrator pt --scene synthetic";;
  esac
  # (--no-extra-workloads: the default line would otherwise render configs 4 and 5 in the same process, and their launches share kernel names with Cornell's)
  CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --no-cpu-baseline --no-extra-workloads $ARGS"
  O=$R/gpurun_out/${TAG:-r04}_$wl
  pmc ${O}_fetch FETCH_SIZE $CMD; pmc ${O}_write WRITE_SIZE $CMD
  for p in A B C E; do eval CTR=\$SQ_$p; pmc ${O}_sq$p "$CTR" $CMD; done
  rm -rf ${O}_stats
  timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- $CMD > ${O}_stats.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $R/gpurun_out/${TAG:-r04}_kernel_stats_$wl.csv > /dev/null
  rm -rf ${O}_stats
  echo "== $wl"
  python $R/tools/pmc_profile.py $R/gpurun_out ${TAG:-r04}_$wl $R/gpurun_out/${TAG:-r04}_pmc_$wl.json
done
