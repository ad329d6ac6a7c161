#!/bin/bash
# A/B on one box: compact output by wave-local transposition (the tree's library) against the
# direct scattered stores (gpurun_tl/libssw_amd_masked.so), config 5 and config 3 sizes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ab
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_compact.py tests/test_gpu_config5.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -n "^E " $O/pytest.log | head -5
for r in 1 2 3; do
  for lib in tree masked; do
    if [ $lib = masked ]; then export SSW_AMD_LIB=$R/gpurun_tl/libssw_amd_masked.so; else unset SSW_AMD_LIB; fi
    timeout 300 python tools/bench_align.py --utts 2048 2>/dev/null | python3 -c "
import sys,json
c=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib round $r', {k: round(c[k],3) for k in ('wall_ms','score_ms','align_ms')}, c['alignment_crc32'])"
  done
done
unset SSW_AMD_LIB
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $R/tools/bench_align.py --utts 2048 --reps 2 > /dev/null 2> $O/c5_rocprof.err
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/config5_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
head -4 $O/config5_kernel_stats.csv
