cd $GRAFT_REPO_ROOT
echo "== new (4 wavefronts per SIMD for the multi-column form)"; python3 scratch/fused_multi.py 2>&1 | grep -v "^[WE]2026"
echo "== old (5: spills)"; BOWGPU_LIB=bow_amd/libbowgpu_fold.so python3 scratch/fused_multi.py 2>&1 | grep -v "^[WE]2026"
