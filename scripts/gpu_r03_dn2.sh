# LDS-tiled a-trous (steps 1, 2) against the plain kernel; tile balance of the 8-way split at 3840x2160
R=$GRAFT_REPO_ROOT
cd $R
ZR_DENOISE=lds timeout 900 python -m pytest tests/test_denoise.py -q -m gpu -x 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if a.startswith("denoise")}}))'
cd /tmp && export TMPDIR=/tmp
for mode in plain lds; do
  export ZR_DENOISE=$mode
  O=$R/gpurun_out/r03_denoise_$mode
  rm -rf ${O}_stats
  timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 4 --no-cpu-baseline --config 5 > ${O}_stats.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $R/gpurun_out/r03_kernel_stats_denoise_$mode.csv > /dev/null
  rm -rf ${O}_stats
  echo "== $mode"; grep svgf $R/gpurun_out/r03_kernel_stats_denoise_$mode.csv
done
unset ZR_DENOISE
cd $R
timeout 900 python tools/tile_balance.py --scene synthetic --width 3840 --height 2160 --frames 6 --layout equal 2>&1 | tail -1 | tee gpurun_out/r03_tile_balance_atrium_4k.jsonl | cut -c1-700
