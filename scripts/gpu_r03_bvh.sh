# device BVH build: parity test, build latency and the frame time on the device-built tree (atrium), next to the host SAH tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "device_built or device_refit or cost_balanced" 2>&1 | tail -4
for mode in refit rebuild; do
  ZR_SCENE_UPDATE=$mode timeout 600 python tools/refit_bench.py 2>&1 | tail -1 | tee -a gpurun_out/r03_refit_latency.jsonl
done
echo "== atrium on the host SAH tree / on the device LBVH tree"
timeout 600 python bench.py --gpus 1 --no-cpu-baseline --config 4 --steps 16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host SAH', d['ms_per_step'], d['roofline']['kernel_ms_per_frame'])"
ZR_BVH_BUILD=device timeout 600 python bench.py --gpus 1 --no-cpu-baseline --config 4 --steps 16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device LBVH', d['ms_per_step'], d['roofline']['kernel_ms_per_frame'])"
