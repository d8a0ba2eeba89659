#!/bin/bash
# One runner for every GPU visit (replaces the per-visit gpu_r0N_*.sh scripts of rounds 1 - 4).  Runs ON the GPU box, from the repository root:
#   gpurun --timeout 900 -- 'bash scripts/gpu.sh ab base "" stc4 ; bash scripts/gpu.sh suite'
# Tasks (TAG=r06 by default; everything lands in gpurun_out/${TAG}_*, copy what should be judged into profiles/):
#   ab LIB...          A/B of library builds: for each name ("" or "default" = libzetaray_amd.so, X = libzetaray_amd_X.so) the bench lines of the
#                      workloads in WORKLOADS (default "cornell atrium"), then one table of frame + kernel times
#   suite [PYTEST_ARGS] the -m gpu parity suite (+ smoke)
#   bench [ARGS]       the default driver line (python bench.py ARGS)
#   profiles           per workload in WORKLOADS (rpt_cornell rpt_atrium gi_cornell ...): --kernel-trace --stats + separate --pmc passes
#                      (FETCH_SIZE, WRITE_SIZE, SQ sets A / B / C / E) of one bench command -> ${TAG}_pmc_<wl>.json, ${TAG}_kernel_stats_<wl>.csv
#   stats [ARGS]       one rocprofv3 --kernel-trace --stats run of bench.py ARGS -> ${TAG}_kernel_stats.csv
#   tiles              tools/tile_balance.py on Cornell + atrium (per-tile times of the 8-way split on one device)
#   prof LIB           section profiler (-DZR_PROF build LIB) on Cornell + atrium -> ${TAG}_section_profile_*.json
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-r06}; OUT=$R/gpurun_out; mkdir -p $OUT
task=$1; shift
libpath() { case "$1" in ""|default) echo "";; *) echo "$R/zetaray_amd/libzetaray_amd_$1.so";; esac; }
wl_args() { case $1 in
    cornell|rpt_cornell) echo "";; atrium|rpt_atrium) echo "--config 4";; rpt_3840x2160_atrium|4k) echo "--config 5";;
    gi|gi_cornell) echo "--config 3";; gi_atrium) echo "--integrator restir_gi --scene synthetic";; pt|pt_cornell) echo "--config pt";;
    pt_atrium) echo "--integrator pt --scene synthetic";; di|2a) echo "--config 2a";; sky|2b) echo "--config 2b";; *) echo "$1";; esac; }
case $task in
ab)
  cd $R
  for lib in "$@"; do
    tag=${lib:-default}; p=$(libpath "$lib")
    if [ -n "$p" ]; then export ZETARAY_AMD_LIB=$p; else unset ZETARAY_AMD_LIB; fi
    for wl in ${WORKLOADS:-cornell atrium}; do
      extra="--no-extra-workloads --no-cpu-baseline"; [ $wl = cornell ] && extra="$extra --steps ${STEPS:-256} --warmup ${WARMUP:-64}"
      timeout 600 python bench.py $(wl_args $wl) $extra > $OUT/${TAG}_ab_${wl}_$tag.json 2> $OUT/${TAG}_ab_err.log || { echo "FAILED $tag $wl"; tail -5 $OUT/${TAG}_ab_err.log; }
    done
  done
  unset ZETARAY_AMD_LIB
  python - "$OUT" "$TAG" "${WORKLOADS:-cornell atrium}" "$@" <<'P'
import json, sys
out, tag, wls, libs = sys.argv[1], sys.argv[2], sys.argv[3].split(), [l or "default" for l in sys.argv[4:]]
for w in wls:
    for t in libs:
        try:
            d = json.loads(open(f"{out}/{tag}_ab_{w}_{t}.json").read().strip().splitlines()[-1])
        except Exception as e:
            print(w, t, "no line", e); continue
        k = d["roofline"]["kernel_ms_per_frame"]
        print(f"{w:8s} {t:10s} {d['ms_per_step']:8.4f} ms", {n: round(v, 4) for n, v in k.items() if v >= 0.02})
P
  ;;
suite)
  cd $R
  timeout 1500 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -15 | tee $OUT/${TAG}_gpu_suite.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/${TAG}_smoke.log
  ;;
bench)
  cd $R; timeout 1500 python bench.py "$@" 2> $OUT/${TAG}_bench_err.log | tail -1 | tee $OUT/${TAG}_bench.json
  ;;
stats)
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st_d
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st_d -- python $R/bench.py --gpus 1 --steps 16 --warmup 4 --settle 8 --ramp-s 0 --no-device-state --no-general-kernels --no-cpu-baseline --no-extra-workloads "$@" > $OUT/${TAG}_stats.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find /tmp/st_d -name "*results.db" | head -1) $OUT/${TAG}_kernel_stats.csv | head -25
  ;;
profiles)
  cd /tmp && export TMPDIR=/tmp
  SQ_A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY"
  SQ_B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  SQ_C="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS"
  SQ_E="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
  # counters in their own runs (no --kernel-trace / --stats beside --pmc: gpurun refuses the combination with trace domains)
  pmc() { local O=$1 CTR=$2; shift 2; rm -rf ${O}_d
    timeout 600 rocprofv3 --pmc $CTR -d ${O}_d -- "$@" > ${O}.log 2>&1
    local DB=$(find ${O}_d -name "*results.db" | head -1)
    if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${O}.csv > /dev/null; else echo "no db for $O"; tail -5 ${O}.log; fi
    rm -rf ${O}_d; }
  for wl in ${WORKLOADS:-rpt_cornell rpt_atrium gi_cornell}; do
    # (--no-extra-workloads: the default line would otherwise render configs 4 and 5 in the same process, and their launches share kernel names with Cornell's)
    # (--frame-overlap 0: a kernel's counters and duration alone on the device; --ramp-s 0: the counter passes serialise the kernels, nothing to ramp)
    CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --ramp-s 0 --frame-overlap 0 --no-device-state --no-general-kernels --no-cpu-baseline --no-extra-workloads $(wl_args $wl)"
    O=$OUT/${TAG}_$wl
    pmc ${O}_fetch FETCH_SIZE $CMD; pmc ${O}_write WRITE_SIZE $CMD
    for p in ${SQ_SETS:-A B C E}; do eval CTR=\$SQ_$p; pmc ${O}_sq$p "$CTR" $CMD; done
    rm -rf ${O}_stats
    timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- $CMD > ${O}_stats.log 2>&1
    python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $OUT/${TAG}_kernel_stats_$wl.csv > /dev/null
    rm -rf ${O}_stats
    echo "== $wl"
    python $R/tools/pmc_profile.py $OUT ${TAG}_$wl $OUT/${TAG}_pmc_$wl.json
  done
  ;;
tiles)
  cd $R
  for wl in ${WORKLOADS:-cornell atrium}; do
    a=""; [ $wl = atrium ] && a="--scene synthetic"
    timeout 900 python tools/tile_balance.py $a "$@" > $OUT/${TAG}_tiles_$wl.jsonl 2> $OUT/${TAG}_tiles_err.log || tail -5 $OUT/${TAG}_tiles_err.log
    tail -3 $OUT/${TAG}_tiles_$wl.jsonl | cut -c1-400
  done
  ;;
prof)
  cd $R; export ZETARAY_AMD_LIB=$(libpath "$1")
  P='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({"ms": d["ms_per_step"], "kernels": d["roofline"]["kernel_ms_per_frame"], "prof": d.get("prof")}, indent=1))'
  for wl in ${WORKLOADS:-cornell atrium}; do
    timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --frame-overlap 0 --no-device-state --no-general-kernels --no-cpu-baseline --no-extra-workloads $(wl_args $wl) 2> $OUT/${TAG}_prof_err.log | tail -1 | python -c "$P" > $OUT/${TAG}_section_profile_$wl.json || tail -5 $OUT/${TAG}_prof_err.log
  done
  ;;
*) echo "unknown task $task"; exit 2;;
esac
