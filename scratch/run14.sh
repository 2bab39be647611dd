#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/pmc_any.sh r02_longw200 long_stream_kernel scratch/one_shape.py longw200 > /dev/null 2>&1
bash scratch/pmc_any.sh r02_longw1000 long_stream_kernel scratch/one_shape.py longw1000 > /dev/null 2>&1
bash scratch/pmc_any.sh r02_tw_was rolling_tw_kernel scratch/one_shape.py tw_was > /dev/null 2>&1
bash scratch/pmc_any.sh r02_tw_3int rolling_tw_kernel scratch/one_shape.py tw_3int > /dev/null 2>&1
bash scratch/pmc_any.sh r02_interp interp_wave2 scratch/one_shape.py interp > /dev/null 2>&1
for t in longw200 longw1000 tw_was tw_3int interp; do echo "== $t"; cat gpurun_out/pmc_r02_$t/summary.txt; done
