#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=40 > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3
grep -A45 "slowest" $O/pytest.log | head -60
timeout 300 python tools/bench_host_boundary.py > $O/host_boundary.json 2> $O/host_boundary.err; cat $O/host_boundary.json
timeout 300 python tools/bench_align_active.py > $O/align_active.json 2> $O/align_active.err; cat $O/align_active.json; tail -3 $O/align_active.err
timeout 300 python tools/bench_align_active.py --ms > $O/align_active_ms.json 2> $O/align_active_ms.err; cat $O/align_active_ms.json
