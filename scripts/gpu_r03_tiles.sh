# what the N-way screen split can reach, measured on one device: equal-area grid against the cost-balanced kd-split (tools/tile_balance.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sc in ${SCENES:-cornell synthetic}; do for lay in equal cost; do for wh in ${SIZES:-1920x1080}; do
  W=${wh%x*}; H=${wh#*x}
  timeout 900 python tools/tile_balance.py --scene $sc --world ${WORLD:-8} --layout $lay --width $W --height $H 2>&1 | tail -1 | tee -a gpurun_out/r03_tile_balance.jsonl
done; done; done
