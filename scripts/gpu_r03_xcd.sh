# XCD-aware tile order against the natural block order, A/B in one visit.  The remap is scripts/xcd_remap.patch (git apply, build -> libzetaray_amd.so;
# the same sources with -DZR_XCD_REMAP=0 -> libzr_noxcd.so); measured slower and not in the tree, DESIGN.md section 7.
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "restir_pt_bit_exact or restir_gi_bit_exact or restir_di_bit_exact or sky_di_bit_exact or materials or tile_split or sort" 2>&1 | grep -E "passed|failed" | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.3}}))'
for a in "" "--config 4" "--config 3" "--config 2a" "--config pt" "--config 4k"; do
  for lib in libzr_noxcd.so libzetaray_amd.so; do
    export ZETARAY_AMD_LIB=$R/zetaray_amd/$lib
    echo "== $lib $a"; timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
