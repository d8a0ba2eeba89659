#!/bin/bash
# update latency + frame time after motion: device refit vs host rebuild (tools/refit_bench.py) -> gpurun_out/refit.json
mkdir -p gpurun_out
python tools/refit_bench.py 2> gpurun_out/refit_err.log | tail -1 > gpurun_out/refit.json
ZR_SCENE_UPDATE=rebuild python tools/refit_bench.py 2>> gpurun_out/refit_err.log | tail -1 >> gpurun_out/refit.json
cat gpurun_out/refit.json; tail -3 gpurun_out/refit_err.log
