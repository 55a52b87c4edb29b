#!/bin/bash
# usage: prof_cmd.sh <outdir> <n> <env> <steps> <auto>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py $2 $3 $4 $5 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py $2 $3 $4 $5 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py $2 $3 $4 $5 > $OUT/pmc2.log 2>&1
find $OUT -name "*stats*.csv" | head; 
