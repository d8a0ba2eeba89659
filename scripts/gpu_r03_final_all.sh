# closing visit: gpu_r03_final.sh (1080p profile set -> profiles/, every bench preset, whole GPU suite, smoke) + the 4K profile set and its two bench lines
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/gpu_r03_final.sh 2>&1 | tail -26
bash scripts/gpu_r03_4kprof.sh 2>&1 | tail -4
python tools/post_chain_bench.py 2>&1 | grep "^{" > gpurun_out/r03_post_chain.jsonl; cut -c1-300 gpurun_out/r03_post_chain.jsonl
