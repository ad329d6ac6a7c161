#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_compact.py tests/test_gpu_active.py tests/test_gpu_config5.py tests/test_gpu_ms.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -n "^E " $O/pytest.log | head
for i in 1 2; do timeout 300 python tools/bench_align.py --utts 2048 > $O/config5_$i.json 2>/dev/null; done
SSW_JOB_ROWS=full timeout 300 python tools/bench_align.py --utts 2048 > $O/config5_full.json 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $R/tools/bench_align.py --utts 2048 --reps 2 > /dev/null 2> $O/c5_rocprof.err
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/config5_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import json,os,glob
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05i/"
for f in sorted(glob.glob(O+"config5_*.json")):
    c=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k:round(c.get(k),3) if isinstance(c.get(k),float) else c.get(k) for k in ("wall_ms","score_ms","align_ms","alignment_crc32")})
PY
head -5 $O/config5_kernel_stats.csv
