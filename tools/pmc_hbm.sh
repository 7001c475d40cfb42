#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only) over tools/hbm_micro.py; summary by tools/pmc_hbm_parse.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_hbm; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/tools/hbm_micro.py > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/tools/hbm_micro.py > $O/write.log 2>&1 || exit 2
find $O -name "*kernel_trace*" -delete
python3 $R/tools/pmc_hbm_parse.py $O
