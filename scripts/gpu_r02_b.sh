# round-2 GPU visit B: full GPU suite on the current build, then the r02 profile set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r02b_pytest.log
WORKLOADS="${WORKLOADS:-rpt_cornell rpt_atrium gi_cornell pt_cornell}" bash scripts/gpu_r02_profiles.sh 2>&1 | tail -80
