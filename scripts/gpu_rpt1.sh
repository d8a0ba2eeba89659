set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "restir" 2>&1 | tail -25
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
