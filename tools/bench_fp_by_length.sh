#!/bin/bash
# First pass by text length (ON the GPU box): the default kernel choice against the sliding-window
# kernel forced (SSW_FP_KERNEL=big), 32 and 256 texts of 60..400 words.  Round 5 ran it with the
# register instances of 2 and 4 HMMs per thread still in place: they
# lost from ~1,200 HMMs up and were removed; now both columns read the same beyond 1,024 HMMs.
O=gpurun_out/r05o; mkdir -p $O
for w in 60 100 120 160 200 250 300 400; do
  fr=$((w * 40))
  timeout 300 python tools/bench_first_pass.py --utts 32 --words $w --frames $fr --reps 3 > $O/fp_${w}_default.json 2>/dev/null
  SSW_FP_KERNEL=big timeout 300 python tools/bench_first_pass.py --utts 32 --words $w --frames $fr --reps 3 > $O/fp_${w}_win.json 2>/dev/null
done
for w in 60 100 200; do
  fr=$((w * 40))
  timeout 300 python tools/bench_first_pass.py --utts 256 --words $w --frames $fr --reps 3 > $O/fp256_${w}_default.json 2>/dev/null
  SSW_FP_KERNEL=big timeout 300 python tools/bench_first_pass.py --utts 256 --words $w --frames $fr --reps 3 > $O/fp256_${w}_win.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05o/*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['hmms_per_text']), 'first_pass_ms', round(j['first_pass_ms'],2), 'decoder_alignment_ms', round(j['decoder_alignment_ms'],2), j['first_pass_completed'], j['aligned'])
    except Exception as e: print(f, 'ERR', e)
PY
