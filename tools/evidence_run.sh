#!/bin/bash
# One gpurun call that refreshes the committed evidence: driver-shaped bench line (+ fresh launch-shape cache), rocprofv3
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of the configs[1] step and of the fp16 step, SQ counters of the weight-gradient
# kernels.  Outputs under gpurun_out/ev/.
R=$GRAFT_REPO_ROOT; cd $R; E=$R/gpurun_out/ev; mkdir -p $E
GCA_TUNE_CACHE=$E/tune_cache.json timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $E/bench.json 2> $E/bench.err || { echo bench failed; tail -5 $E/bench.err; exit 1; }
echo bench done; cp $E/tune_cache.json profiles/tune_cache.json
ARENA=33.51e6
export GCA_SEED_CACHE=1
bash tools/pmc_step.sh > $E/pmc_step.log 2>&1 || { echo pmc_step failed; tail -5 $E/pmc_step.log; exit 2; }
python tools/pmc_parse.py gpurun_out/pmc_step $ARENA > $E/pmc_traffic.json
python tools/step_breakdown.py $(ls gpurun_out/pmc_step/trace/*kernel_trace.csv | head -1) 4 > $E/step_breakdown.txt
cp $(ls gpurun_out/pmc_step/trace/*kernel_stats.csv | head -1) $E/bench_kernel_stats.csv
rm -rf gpurun_out/pmc_step; echo configs1 pmc done
GCA_BENCH_MATH=fp16 bash tools/pmc_step.sh > $E/pmc_step_f16.log 2>&1 || { echo pmc_step f16 failed; tail -5 $E/pmc_step_f16.log; exit 3; }
python tools/pmc_parse.py gpurun_out/pmc_step 46.2e6 > $E/pmc_traffic_f16.json
python tools/step_breakdown.py $(ls gpurun_out/pmc_step/trace/*kernel_trace.csv | head -1) 4 > $E/f16_step_breakdown.txt
cp $(ls gpurun_out/pmc_step/trace/*kernel_stats.csv | head -1) $E/f16_bench_kernel_stats.csv
rm -rf gpurun_out/pmc_step; echo f16 pmc done
bash tools/pmc_conv.sh L00,L01,L02 wgrad > $E/sq_counters_wgrad.txt 2>&1 || echo pmc_conv failed
rm -rf gpurun_out/pmc_conv
ls -la $E
