cd $GRAFT_REPO_ROOT
for lib in libzetaray_amd.so libzr_novec.so libzr_occ.so libzr_o2.so; do
  if [ -f zetaray_amd/$lib ]; then echo "== $lib"; ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/$lib timeout 300 python scripts/gpu_dbg_tex2.py 2>&1 | grep -E "^bad|^all|Error|error" | cut -c1-200; fi
done
