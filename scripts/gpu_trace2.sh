cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for m in 0 1; do
echo "== ZR_TRACE_MODE=$m"
ZR_TRACE_MODE=$m timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_frame'])"
ZR_TRACE_MODE=$m timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_frame'])"
done
timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_frame'])"
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_frame'])"
