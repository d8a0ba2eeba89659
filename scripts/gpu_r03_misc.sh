# (a) the megakernel with a 336-byte path-state round trip per bounce boundary (ZR_K11=trip), (b) denoise: LDS tiles for step 4 too, (c) 4K cost-balanced tiles
R=$GRAFT_REPO_ROOT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if a in ("rpt_pathtrace",) or a.startswith("denoise")}, "trip": d["config"].get("k11_bounce_boundaries")}))'
for mode in inline trip; do
  for a in "" "--config 4"; do
    echo "== K11 $mode $a"; ZR_K11=$mode timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
ZR_K11=trip timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "restir_pt_bit_exact or restir_pt_materials" 2>&1 | tail -2
for mode in plain lds lds4; do
  echo "== denoise $mode"; ZR_DENOISE=$mode timeout 900 python bench.py --gpus 1 --steps 8 --warmup 2 --settle 4 --no-cpu-baseline --config 5 2>&1 | tail -1 | python -c "$P"
done
ZR_DENOISE=lds4 timeout 900 python -m pytest tests/test_denoise.py -q -m gpu -x 2>&1 | tail -2
timeout 900 python tools/tile_balance.py --scene synthetic --width 3840 --height 2160 --frames 6 --layout cost 2>&1 | tail -1 | tee gpurun_out/r03_tile_balance_atrium_4k_cost.jsonl | cut -c1-800
