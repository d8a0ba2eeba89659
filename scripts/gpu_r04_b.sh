# Round-4 GPU visit B: the whole GPU suite on the round's build (incl. the at-size atrium window tests and num_spatial_passes cases), the second
# tolerance-mode variant (afn + denormal flush), and SQ / TCC counter passes over the post chain (a-trous, TAA) at 3840 x 2160.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r04b_gpu_suite.log 2>&1; tail -25 $O/r04b_gpu_suite.log
timeout 300 python bench.py --arith fast --no-extra-workloads --no-cpu-baseline > $O/r04b_bench_fast_cornell.json 2>/dev/null
ZETARAY_AMD_LIB=$R/zetaray_amd/libzetaray_amd_fast2.so timeout 300 python bench.py --no-extra-workloads --no-cpu-baseline > $O/r04b_bench_fast2_cornell.json 2>/dev/null
ZETARAY_AMD_LIB=$R/zetaray_amd/libzetaray_amd_fast2.so timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/r04b_bench_fast2_atrium.json 2>/dev/null
ZETARAY_AMD_LIB=$R/zetaray_amd/libzetaray_amd_fast2.so timeout 600 python tools/fast_arith_check.py > $O/r04b_fast2_parity.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r04b_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unparsable", e); continue
    print(os.path.basename(f), d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_frame"])
PY
cat $O/r04b_fast2_parity.json
cd /tmp
pmc() { # $1 = output stem, $2 = counters, rest = command
  local OO=$1 CTR=$2; shift 2
  rm -rf ${OO}_d
  timeout 600 rocprofv3 --pmc $CTR -d ${OO}_d -- "$@" > ${OO}.log 2>&1
  local DB=$(find ${OO}_d -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py pmc $DB ${OO}.csv > /dev/null; else echo "no db for $OO"; tail -5 ${OO}.log; fi
  rm -rf ${OO}_d
}
timeout 300 python $R/tools/post_chain_bench.py > $O/r04b_post_chain.jsonl 2>/dev/null; cat $O/r04b_post_chain.jsonl
pmc $O/r04b_post_sqA "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_INSTS_VMEM_RD" python $R/tools/post_chain_bench.py
pmc $O/r04b_post_sqB "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY" python $R/tools/post_chain_bench.py
pmc $O/r04b_post_fetch "FETCH_SIZE" python $R/tools/post_chain_bench.py
pmc $O/r04b_post_write "WRITE_SIZE" python $R/tools/post_chain_bench.py
pmc $O/r04b_post_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" python $R/tools/post_chain_bench.py
grep -h "svgf\|taa" $O/r04b_post_sqA.csv $O/r04b_post_sqB.csv $O/r04b_post_fetch.csv $O/r04b_post_write.csv $O/r04b_post_tcp.csv | head -80
