#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_active.py tests/test_gpu_dropin.py tests/test_gpu_selftest.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -n "^E " $O/pytest.log | head
timeout 300 python tools/bench_align_active.py > $O/align_active.json 2> $O/align_active.err; cat $O/align_active.json
timeout 300 python tools/bench_align_active.py --ms > $O/align_active_ms.json 2> $O/align_active_ms.err; cat $O/align_active_ms.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -o a -- python3 $R/tools/bench_align_active.py --reps 2 > $O/active_under_rocprof.json 2> $O/a_rocprof.err
find $O/prof_a -name "*kernel_stats.csv" -exec cp {} $O/align_active_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
head -6 $O/align_active_kernel_stats.csv
