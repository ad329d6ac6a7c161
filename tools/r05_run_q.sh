#!/bin/bash
O=gpurun_out/r05q; mkdir -p $O
for w in 120 200 300 400; do
  fr=$((w * 40))
  for t in 256 512 1024; do
    SSW_FP_WIN_TPB=$t timeout 300 python tools/bench_first_pass.py --utts 32 --words $w --frames $fr --reps 3 > $O/fp_${w}_t$t.json 2>/dev/null
  done
done
for t in 256 512 1024; do
  SSW_FP_WIN_TPB=$t timeout 300 python tools/bench_first_pass.py --utts 256 --words 200 --frames 8000 --reps 2 > $O/fp256_200_t$t.json 2>/dev/null
  SSW_FP_WIN_TPB=$t timeout 300 python tools/bench_page.py > $O/page_t$t.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05q/fp*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['hmms_per_text']), 'first_pass_ms', round(j['first_pass_ms'],2), j['first_pass_completed'])
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r05q/page*.json')):
    print(f.split('/')[-1], open(f).read()[:700])
PY
