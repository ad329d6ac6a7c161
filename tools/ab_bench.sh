#!/bin/bash
# A/B on ONE box: tools/ab_bench.sh libA.so libB.so ... ; each lib is benched ROUNDS times, interleaved
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    echo "== $lib (round $r)"
    SSW_AMD_LIB=$PWD/$lib tools/quick_bench.sh $BENCH_ARGS
  done
done
