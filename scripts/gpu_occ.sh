cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_per_frame"])'
for v in base occ; do
  echo "== $v"
  if [ $v = occ ]; then cp zetaray_amd/libzr_occ.so zetaray_amd/libzetaray_amd.so; fi
  timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --scene tests/golden/cornell.npz --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
