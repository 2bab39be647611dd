cd $GRAFT_REPO_ROOT
echo "== new (3 wavefronts per SIMD for multi-column both-kinds)"; python3 scratch/twc_multi.py 2>&1 | grep -v "^[WE]2026"
echo "== old"; BOWGPU_LIB=bow_amd/libbowgpu_twcold.so python3 scratch/twc_multi.py 2>&1 | grep -v "^[WE]2026"
