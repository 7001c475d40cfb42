cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GCA_AUTOTUNE=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc3 -- python tools/conv_micro.py --layers L01,L02,L12 --what fwd --reps 3 > gpurun_out/pmc3.log 2>&1
echo EXIT=$?
