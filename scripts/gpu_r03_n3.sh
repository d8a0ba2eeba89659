# N3 by measurement: K15 with plain gathers against the LDS-tiled variant (ZR_SEARCH=tile), parity first
cd $GRAFT_REPO_ROOT
ZR_SEARCH=tile timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "restir_pt_bit_exact or rpt or tile_split" 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "spatial_search": k.get("rpt_spatial_search")}))'
for m in plain tile; do for c in default 4; do
  A=""; [ $c = default ] || A="--config $c"
  echo "== $m $c"; ZR_SEARCH=$m timeout 600 python bench.py --gpus 1 --no-cpu-baseline --steps 32 --warmup 8 $A 2>&1 | tail -1 | python -c "$P"
done; done
