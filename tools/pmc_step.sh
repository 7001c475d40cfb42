#!/bin/bash
# rocprofv3 passes over the benchmark step: (1) kernel trace, (2) FETCH_SIZE, (3) WRITE_SIZE  (separate passes:
# the TCC block cannot hold both; PMC passes carry --kernel-trace only).  Outputs under gpurun_out/pmc_step/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_step
mkdir -p $O
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-other-math --no-sub-workloads ${GCA_BENCH_MATH:+--math $GCA_BENCH_MATH} $GCA_BENCH_ARGS"
export GCA_TUNE_CACHE=$O/tune_cache.json      # pass 1 measures the launch configurations, passes 2-3 reuse them
rm -f $GCA_TUNE_CACHE
# GCA_SEED_CACHE=1: start from the committed launch shapes (profiles/tune_cache.json), so the profiled kernels are the ones of
# the bench line that wrote that file
[ -n "$GCA_SEED_CACHE" ] && cp $R/profiles/tune_cache.json $GCA_TUNE_CACHE
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py $ARGS > $O/trace.json 2> $O/trace.err || exit 1
echo trace done
[ -n "$GCA_TRACE_ONLY" ] && exit 0
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py $ARGS > $O/fetch.json 2> $O/fetch.err || exit 2
echo fetch done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py $ARGS > $O/write.json 2> $O/write.err || exit 3
echo write done
ls -la $O/*/ | head -30
# keep the merge-back small: drop the big per-dispatch traces of the PMC passes (counter csv has what we need)
cp $GCA_TUNE_CACHE $R/gpurun_out/tune_cache.json
find $O/fetch $O/write -name "*kernel_trace*" -delete
du -sh $O
