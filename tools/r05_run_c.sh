#!/bin/bash
# Round 5, third call: compact rows + non-finite tests, config 3 / 5 timings with compact rows,
# kernel table and counters of the alignment kernel on compact rows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_compact.py tests/test_gpu_nonfinite.py tests/test_gpu_config5.py tests/test_gpu_align.py tests/test_golden_fixtures.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
tail -40 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err
tail -3 $O/bench.err
timeout 300 python tools/bench_align.py --utts 2048 > $O/config5_compact.json 2>/dev/null
SSW_JOB_ROWS=full timeout 300 python tools/bench_align.py --utts 2048 > $O/config5_full.json 2>/dev/null
timeout 300 python tools/bench_align.py --utts 256 > $O/config3_compact.json 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $R/tools/bench_align.py --utts 2048 --reps 2 > /dev/null 2> $O/c5_rocprof.err
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/config5_compact_kernel_stats.csv \;
find $O/prof_c5 -name "*kernel_trace.csv" -delete
cd $R
timeout 900 python3 tools/pmc_cmd.py gpurun_out/r05c/align_pmc_2048_compact.json 'viterbi_align\w+|ptm_senone_kernel|ptm_topn_mfma_kernel|compact_\w+' -- python3 tools/bench_align.py --utts 2048 --reps 1 > $O/pmc2048.log 2>&1
rm -rf $R/gpurun_out/pmc_align_pmc_2048_compact
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05c/"
d=json.loads(open(O+"bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["frac"])
c=d.get("config5") or {}
print("config5", {k:c.get(k) for k in ("wall_ms","score_ms","align_ms","gather_ms","alignment_crc32")})
print("align", d.get("align"))
for f in ("config5_compact.json","config5_full.json","config3_compact.json"):
    try:
        c=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, {k:c.get(k) for k in ("wall_ms","score_ms","align_ms","alignment_crc32")})
    except Exception as e: print(f, e)
PY
cat $O/config5_compact_kernel_stats.csv | head -8
tail -4 $O/pmc2048.log
