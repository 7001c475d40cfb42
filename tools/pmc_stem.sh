cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_stem; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -o a -- python3 $R/tools/stem_check.py --only "R(2+1)D" > $O/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p2 -o b -- python3 $R/tools/stem_check.py --only "R(2+1)D" > $O/p2.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o c -- python3 $R/tools/stem_check.py --only "R(2+1)D" > $O/p3.log 2>&1 || echo pass3 failed
find $O -name "*kernel_trace*" -delete
python3 $R/tools/pmc_conv_parse.py $O
