# K11's 3- / 4-wave build chosen by the number of launch rounds (zr_api.hip FewerRoundsAtFourWaves): per-tile kernel times of the 2-, 4- and 8-way split
cd $GRAFT_REPO_ROOT
for w in 2 4 8; do for m in 0 1; do
  echo "== world $w ZR_K11_ROUNDS=$m"; ZR_K11_ROUNDS=$m python scripts/tile_kernels.py --world $w 2>&1 | grep '^{' | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
print('max tile', max(r['sum_ms'] for r in rows), 'K11 per tile', [r['k']['rpt_pathtrace'] for r in rows])"
done; done
