# BVH builder experiments on the atrium (scripts: ZR_BVH_MAX_LEAF / ZR_BVH_SAH_LEAF, zr_bvh.h): K9 trace ms and ReSTIR PT kernel ms per variant
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:k[a] for a in k if a in ("gbuffer","trace","pt_shade","rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial")})'
IFS=","; for v in ${VARIANTS:-4 0,2 0,1 0,3 0,2 1.0,3 1.0}; do IFS=" "
  set -- $v
  export ZR_BVH_MAX_LEAF=$1; if [ "$2" = "0" ]; then unset ZR_BVH_SAH_LEAF; else export ZR_BVH_SAH_LEAF=$2; fi
  echo "== max leaf $1, SAH node cost $2"
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
