cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc1 -- python tools/conv_micro.py --layers L01,L02,L12,L30 --what fwd --reps 2 > gpurun_out/pmc1.log 2>&1
echo EXIT1=$?
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc2 -- python tools/conv_micro.py --layers L01,L02,L12,L30 --what fwd --reps 2 > gpurun_out/pmc2.log 2>&1
echo EXIT2=$?
tail -3 gpurun_out/pmc1.log | cut -c1-300
ls gpurun_out/pmc1/*/ gpurun_out/pmc2/*/
