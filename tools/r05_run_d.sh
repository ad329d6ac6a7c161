#!/bin/bash
# Round 5, fourth call: byte-token alignment kernel -- tests, config 3 / 5 timings, kernel table.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_compact.py tests/test_gpu_dropin.py tests/test_gpu_config5.py tests/test_golden_fixtures.py tests/test_gpu_first_pass.py tests/test_gpu_active.py tests/test_gpu_two_pass_history.py tests/test_gpu_reference_pins.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1
tail -30 $O/pytest.log
timeout 300 python tests/soak_parity.py --mode align --seconds 60 > $O/soak_align.json 2>$O/soak_align.err; tail -2 $O/soak_align.json
for rows in compact full; do for bt in 1 0; do
SSW_JOB_ROWS=$rows SSW_ALIGN_BT=$bt timeout 300 python tools/bench_align.py --utts 2048 > $O/config5_${rows}_bt$bt.json 2>/dev/null
SSW_JOB_ROWS=$rows SSW_ALIGN_BT=$bt timeout 300 python tools/bench_align.py --utts 256 > $O/config3_${rows}_bt$bt.json 2>/dev/null
done; done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $R/tools/bench_align.py --utts 2048 --reps 2 > /dev/null 2> $O/c5_rocprof.err
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/config5_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 $R/tools/bench_align.py --utts 256 --reps 3 > /dev/null 2> $O/c3_rocprof.err
find $O/prof_c3 -name "*kernel_stats.csv" -exec cp {} $O/config3_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import json,os,glob
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05d/"
for f in sorted(glob.glob(O+"config*_bt*.json")):
    try:
        c=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k:round(c.get(k),3) if isinstance(c.get(k),float) else c.get(k) for k in ("wall_ms","score_ms","align_ms","alignment_crc32")})
    except Exception as e: print(f, e)
PY
head -6 $O/config5_kernel_stats.csv; head -6 $O/config3_kernel_stats.csv
