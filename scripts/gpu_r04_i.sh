#!/bin/bash
# round 4, visit I: the reconnect kernels (K14 k_rpt_temporal, K16 k_rpt_stc) as one-wave blocks (-DZR_RECON_BLOCK=64, make variant NAME=rb64)
# against the 256-thread blocks of the default build: Cornell + atrium 1080p frame and kernel times, and the parity tests on the variant
mkdir -p gpurun_out
for lib in "" rb64; do
  tag=${lib:-default}
  if [ -n "$lib" ]; then export ZETARAY_AMD_LIB=$PWD/zetaray_amd/libzetaray_amd_$lib.so; else unset ZETARAY_AMD_LIB; fi
  timeout 300 python bench.py --no-extra-workloads --no-cpu-baseline --steps 128 --warmup 32 > gpurun_out/r04i_cornell_$tag.json 2> gpurun_out/r04i_err.log || tail -5 gpurun_out/r04i_err.log
  timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r04i_atrium_$tag.json 2>> gpurun_out/r04i_err.log || tail -5 gpurun_out/r04i_err.log
done
python - <<'P'
import json
for t in ("default", "rb64"):
    for w in ("cornell", "atrium"):
        d = json.loads(open(f"gpurun_out/r04i_{w}_{t}.json").read().strip().splitlines()[-1])
        k = d["roofline"]["kernel_ms_per_frame"]
        print(t, w, d["ms_per_step"], {n: k[n] for n in k if "reconnect" in n or "pathtrace" in n})
P
export ZETARAY_AMD_LIB=$PWD/zetaray_amd/libzetaray_amd_rb64.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "restir_pt and (cornell or bit_exact or tiles or sort)" 2>&1 | tail -3
