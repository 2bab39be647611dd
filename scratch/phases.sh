cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for st in 1 2 3 4 5 0; do
  BOWGPU_DBG_STOP=$st BOWGPU_FAST_PERSIST=0 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d gpurun_out/ph/s$st -- python3 bench.py --rows 200000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/ph_$st.log 2>&1
  echo "stop=$st"; python3 scratch/pmc_summary.py gpurun_out/ph/s$st | grep -v "^void\|^bowgpu" | awk '{printf "%s %s  ", $1, $3} END {print ""}'
done
