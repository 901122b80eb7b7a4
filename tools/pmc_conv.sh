# PMC passes over the conv microbench (fwd only). usage: bash tools/pmc_conv.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export REPS=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $R/tools/bench_conv.py ${2:-fwd} > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o p -- python $R/tools/bench_conv.py ${2:-fwd} > $OUT/p2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p3 -o p -- python $R/tools/bench_conv.py ${2:-fwd} > $OUT/p3.log 2>&1
ls $OUT/p1 $OUT/p2 $OUT/p3; tail -3 $OUT/p3.log
