#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/px8; mkdir -p $OUT
SH="${SHAPES:-8x256 16x256 32x256 64x256 16x512 64x512}"
VARIANTS="$VARIANTS" SHAPES="$SH" timeout 900 python scripts/k6_variants.py 2> $OUT/err.log | tee $OUT/variants.jsonl
K6V_FLAGS=128 SHAPES="$SH" timeout 900 python scripts/k6_variants.py 2>> $OUT/err.log | sed 's/product/legacy/g' | tee $OUT/legacy.jsonl
tail -2 $OUT/err.log
