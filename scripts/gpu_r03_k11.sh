# Round-3 K11 A/B: the inline megakernel (ZR_K11=inline) against the pooled-trace kernel (ZR_K11=pool): parity tests of the ReSTIR PT path with
# the pooled kernel, then bench lines for Cornell and the atrium with both.   LIBS="libzetaray_amd.so libzr_x.so" selects library builds.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  ZR_K11=pool timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -x -k "${TESTS:-rpt or restir_pt or stream_ordered}" 2>&1 | tail -8
fi
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "mrays": d["value"], "k11": k.get("rpt_pathtrace"), "temporal": k.get("rpt_reconnect_temporal"), "spatial": k.get("rpt_reconnect_spatial")}))'
for lib in ${LIBS:-libzetaray_amd.so}; do
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/$lib
  for mode in ${MODES:-inline pool}; do
    echo "== $lib $mode cornell"; ZR_K11=$mode timeout 600 python bench.py --gpus 1 --steps ${STEPS:-64} --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
    echo "== $lib $mode atrium"; ZR_K11=$mode timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  done
done
