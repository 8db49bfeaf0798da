#!/bin/bash
# usage: scratch/prof_sq.sh OUTDIR cmd...   — two rocprofv3 --pmc passes (8 SQ slots each) + per-kernel summary in OUTDIR/sq_summary.txt
OUT=$1; shift
ROOT=$(pwd)
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    -d $ROOT/$OUT/sq1 --output-format csv -- "$@" > $ROOT/$OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 \
    -d $ROOT/$OUT/sq2 --output-format csv -- "$@" > $ROOT/$OUT/sq2.log 2>&1
cd $ROOT
python scratch/pmc_summary.py $(find $OUT/sq1 $OUT/sq2 -name '*counter_collection.csv') > $OUT/sq_summary.txt
find $OUT -name '*counter_collection.csv' -size +2M -delete
find $OUT -name '*.db' -delete
