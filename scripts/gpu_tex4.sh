cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_per_frame"])'
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --integrator pt --textured --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --textured --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
python scripts/gpu_taa.py 2>&1 | tail -2
