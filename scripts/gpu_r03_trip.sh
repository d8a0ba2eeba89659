# K11 path-state round trip (ZR_K11=trip: what a per-bounce relaunch with compaction would move) against the megakernel, with the HBM traffic
# of both (FETCH_SIZE / WRITE_SIZE passes); then the k_rgi primary-hit rematerialisation A/B (libzr_rgi_noremat.so = without)
R=$GRAFT_REPO_ROOT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.3}, "trip": d["config"].get("k11_bounce_boundaries")}))'
for mode in inline trip; do
  for a in "" "--config 4"; do
    echo "== K11 $mode $a"; ZR_K11=$mode timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
cd /tmp && export TMPDIR=/tmp
pmc() { local O=$1 CTR=$2; shift 2; rm -rf ${O}_d
  timeout 600 rocprofv3 --pmc $CTR -d ${O}_d -- "$@" > ${O}.log 2>&1
  local DB=$(find ${O}_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${O}.csv > /dev/null; else echo "no db for $O"; tail -5 ${O}.log; fi
  rm -rf ${O}_d; }
for a in cornell atrium; do
  ARGS=""; [ $a = atrium ] && ARGS="--config 4"
  export ZR_K11=trip
  CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --no-cpu-baseline $ARGS"
  O=$R/gpurun_out/r03_trip_$a
  pmc ${O}_fetch FETCH_SIZE $CMD; pmc ${O}_write WRITE_SIZE $CMD
  grep pathtrace ${O}_fetch.csv ${O}_write.csv
  unset ZR_K11
done
cd $R
for lib in libzetaray_amd.so libzr_rgi_noremat.so; do
  export ZETARAY_AMD_LIB=$R/zetaray_amd/$lib
  for a in "--config 3" "--integrator restir_gi --scene synthetic"; do
    echo "== $lib $a"; timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
unset ZETARAY_AMD_LIB
cd /tmp
CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --no-cpu-baseline --config 3"
O=$R/gpurun_out/r03_gi_remat
pmc ${O}_fetch FETCH_SIZE $CMD; pmc ${O}_write WRITE_SIZE $CMD
grep "k_rgi" ${O}_fetch.csv ${O}_write.csv
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "gi or GI or lvg" 2>&1 | tail -3
ZR_K11=trip timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "restir_pt or rpt" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_multi_rank" 2>&1 | tail -15
