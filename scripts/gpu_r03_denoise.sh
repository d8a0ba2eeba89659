# the denoise pass: parity tests, timings at 1080p (Cornell) and 3840x2160 (atrium = config 5), HBM traffic of its kernels
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_denoise.py tests/test_gpu_parity.py -q -m gpu -x -k "denoise or bench_multi_rank" 2>&1 | tail -8
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.02}, "roofline": {x: d["roofline"][x] for x in ("kernel", "achieved", "frac")}}))'
for a in "--denoise" "--config 5"; do
  echo "== $a"; timeout 900 python bench.py --gpus 1 --steps 16 --warmup 4 --settle 8 --no-cpu-baseline $a 2>&1 | tail -1 | tee gpurun_out/r03_bench_denoise_$(echo $a | tr -d ' -').json | python -c "$P"
done
cd /tmp && export TMPDIR=/tmp
pmc() { local O=$1 CTR=$2; shift 2; rm -rf ${O}_d
  timeout 600 rocprofv3 --pmc $CTR -d ${O}_d -- "$@" > ${O}.log 2>&1
  local DB=$(find ${O}_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${O}.csv > /dev/null; else echo "no db for $O"; tail -5 ${O}.log; fi
  rm -rf ${O}_d; }
CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 4 --no-cpu-baseline --denoise --width 3840 --height 2160"
O=$R/gpurun_out/r03_denoise4k
pmc ${O}_fetch FETCH_SIZE $CMD; pmc ${O}_write WRITE_SIZE $CMD
grep svgf ${O}_fetch.csv ${O}_write.csv
rm -rf ${O}_stats
timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- $CMD > ${O}_stats.log 2>&1
python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $R/gpurun_out/r03_kernel_stats_denoise4k.csv > /dev/null
rm -rf ${O}_stats
grep svgf $R/gpurun_out/r03_kernel_stats_denoise4k.csv
