#!/bin/bash
# counter passes only (no tests / bench lines): C3 step, then the other workloads
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r03_final}; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/sq --output-format csv -- $P > $O/sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -d $O/sq2 --output-format csv -- $P > $O/sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $P > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- $P > $O/write.log 2>&1 || exit 1
python3 $R/tools/counters.py $O/fetch $O/write $O/sq,$O/sq2 $O/${T}_counters_C3_step.json C3 step 4096 64 || exit 1
rm -rf $O/sq $O/sq2 $O/fetch $O/write
bash $R/tools/measure_counters.sh $T
