mkdir -p gpurun_out
for lib in "" vdiv; do
  tag=${lib:-default}
  if [ -n "$lib" ]; then export ZETARAY_AMD_LIB=$PWD/zetaray_amd/libzetaray_amd_$lib.so; else unset ZETARAY_AMD_LIB; fi
  timeout 300 python bench.py --no-extra-workloads --no-cpu-baseline --steps 128 --warmup 32 > gpurun_out/ab_cornell_$tag.json 2> gpurun_out/ab_err.log || tail -5 gpurun_out/ab_err.log
  timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/ab_atrium_$tag.json 2>> gpurun_out/ab_err.log || tail -5 gpurun_out/ab_err.log
done
python - <<'P'
import json
for t in ("default", "vdiv"):
    for w in ("cornell", "atrium"):
        d = json.loads(open(f"gpurun_out/ab_{w}_{t}.json").read().strip().splitlines()[-1])
        k = d["roofline"]["kernel_ms_per_frame"]
        print(t, w, d["ms_per_step"], {n: k[n] for n in k if "reconnect" in n or "pathtrace" in n or "gbuffer" in n})
P
