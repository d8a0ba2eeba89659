# Round-4 GPU visit C: the whole GPU suite on the build with denoise definition 2 + tiles, the TAA fast conversion and RcPark; post-chain timings
# (both forms of the a-trous stencil) with SQ counters; K11 with the selected reconnection parked in LDS (time, parity, FETCH / WRITE); the tolerance-mode
# build with its final flags; config 5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r04c_gpu_suite.log 2>&1; tail -25 $O/r04c_gpu_suite.log
ZR_K11_PARK=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -m gpu -q -k "restir_pt or rpt" > $O/r04c_park_parity.log 2>&1; tail -3 $O/r04c_park_parity.log
timeout 300 python tools/post_chain_bench.py > $O/r04c_post_chain.jsonl 2>/dev/null; cat $O/r04c_post_chain.jsonl
ZR_DENOISE_TAPS=row timeout 300 python tools/post_chain_bench.py > $O/r04c_post_chain_row.jsonl 2>/dev/null; cat $O/r04c_post_chain_row.jsonl
timeout 300 python bench.py --no-extra-workloads --no-cpu-baseline > $O/r04c_bench_cornell.json 2>/dev/null
ZR_K11_PARK=1 timeout 300 python bench.py --no-extra-workloads --no-cpu-baseline > $O/r04c_bench_cornell_park.json 2>/dev/null
timeout 300 python bench.py --arith fast --no-extra-workloads --no-cpu-baseline > $O/r04c_bench_fast_cornell.json 2>/dev/null
ZR_K11_PARK=1 timeout 300 python bench.py --arith fast --no-extra-workloads --no-cpu-baseline > $O/r04c_bench_fast_cornell_park.json 2>/dev/null
timeout 300 python bench.py --arith fast --config 4 --no-cpu-baseline > $O/r04c_bench_fast_atrium.json 2>/dev/null
timeout 400 python bench.py --config 5 --no-cpu-baseline > $O/r04c_bench_5.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r04c_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unparsable", e); continue
    print(os.path.basename(f), d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_frame"])
PY
cd /tmp
pmc() { local OO=$1 CTR=$2; shift 2; rm -rf ${OO}_d
  timeout 600 rocprofv3 --pmc $CTR -d ${OO}_d -- "$@" > ${OO}.log 2>&1
  local DB=$(find ${OO}_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${OO}.csv > /dev/null; else echo "no db for $OO"; tail -5 ${OO}.log; fi
  rm -rf ${OO}_d; }
pmc $O/r04c_post_sqA "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_INSTS_SALU" python $R/tools/post_chain_bench.py
grep -h "svgf\|taa" $O/r04c_post_sqA.csv | cut -c1-110
CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --settle 8 --no-cpu-baseline --no-extra-workloads"
pmc $O/r04c_k11_fetch FETCH_SIZE $CMD; pmc $O/r04c_k11_write WRITE_SIZE $CMD
export ZR_K11_PARK=1
pmc $O/r04c_k11park_fetch FETCH_SIZE $CMD; pmc $O/r04c_k11park_write WRITE_SIZE $CMD
unset ZR_K11_PARK
grep -h "pathtrace" $O/r04c_k11_fetch.csv $O/r04c_k11_write.csv $O/r04c_k11park_fetch.csv $O/r04c_k11park_write.csv | cut -c1-140
cd $R && timeout 900 python -m pytest tests/test_fast_arith.py -m gpu -q > $O/r04c_tests_fast.log 2>&1; tail -4 $O/r04c_tests_fast.log; cp $O/fast_arith_parity.json $O/r04c_fast_arith_parity.json
