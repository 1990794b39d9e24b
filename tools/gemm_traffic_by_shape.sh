#!/bin/bash
# HBM-side traffic of the GEMM per shape: one shape per rocprofv3 --pmc pass (FETCH_SIZE and WRITE_SIZE in
# separate passes, program directly after `--`), summarised by tools/gemm_traffic_collect.py into
# gpurun_out/gemm_traffic_by_shape.json (copy it to profiles/rNN_gemm_traffic_by_shape.json).
# usage (on the GPU box): [LN=0] bash tools/gemm_traffic_by_shape.sh [tile, default 0 = what the library picks]   (LN=1, the default: the
# folded-LayerNorm forms of the launches, as the encoders issue them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TILE=${1:-0}
OUT=gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT
# SHAPES="M N K epi;..." overrides the list (round 4: + the ViT-B-width shapes of config 3 / 5 and DINOv2)
SHAPES=${SHAPES:-"65536 3840 1280 0;65536 1280 1280 2;65536 5120 1280 1;65536 1280 5120 2;4096 3840 1280 0;4096 1280 1280 2;4096 5120 1280 1;4096 1280 5120 2;65536 2304 768 0;65536 768 768 2;65536 3072 768 1;65536 768 3072 2"}
IFS=';' read -ra SHL <<< "$SHAPES"
for sh in "${SHL[@]}"; do
  set -- $sh
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --pmc $c -d $OUT/${1}x${2}x${3}x${4}_$c -o p -- python3 tools/gemm_one.py $1 $2 $3 $TILE 3 $4 ${LN:-1} > /dev/null 2>&1
  done
done
python3 tools/gemm_traffic_collect.py $OUT gpurun_out/gemm_traffic_by_shape.json $TILE
rm -rf $OUT     # the raw counter databases are tens of MiB; gpurun_out/ only travels back when it is under 64 MiB
