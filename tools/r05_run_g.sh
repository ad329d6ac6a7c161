#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_ptm.py tests/test_gpu_selftest.py tests/test_gpu_active.py tests/test_gpu_ms.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -o a -- python3 $R/tools/bench_align_active.py --reps 2 > $O/active_under_rocprof.json 2> $O/a_rocprof.err
find $O/prof_a -name "*kernel_stats.csv" -exec cp {} $O/align_active_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
cd $R
head -8 $O/align_active_kernel_stats.csv
cat $O/active_under_rocprof.json | cut -c1-900
timeout 300 python tools/bench_host_boundary.py > $O/host_boundary.json 2> $O/host_boundary.err; cat $O/host_boundary.json
