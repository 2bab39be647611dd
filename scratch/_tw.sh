cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== new"; python3 scratch/general_bench.py 2>&1 | grep -E "rolling_tw_kernel|simple kernel"
echo "== old"; BOWGPU_LIB=bow_amd/libbowgpu_twold.so python3 scratch/general_bench.py 2>&1 | grep -E "rolling_tw_kernel|simple kernel"
done
