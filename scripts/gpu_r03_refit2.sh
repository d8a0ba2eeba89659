cd $GRAFT_REPO_ROOT
timeout 600 python tools/refit_bench.py 2>&1 | tail -1 | tee gpurun_out/r03_refit_latency2.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py tests/test_cpu_parity.py -q -m gpu -x -k "variants or cpp_denoise or defaults" 2>&1 | tail -3
