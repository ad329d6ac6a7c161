#!/bin/bash
# short bench line for kernel tuning: value, ms_per_step, per-kernel ms, 65536-frame line
python bench.py --no-cpu-baseline --no-align "$@" 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = b.get('kernels', {})
print('value %.2f M  ms/step %.4f  scan %.4f  senone %.4f  frac %.3f  exact %.5f' % (b['value'] / 1e6, b['ms_per_step'], k.get('topn_kernel', {}).get('ms', 0), k.get('senone_kernel', {}).get('ms', 0), b['roofline']['frac'], b.get('exact_pass_share', 0)))
x = b.get('batch_65536')
if x: print('65536: %.2f M  ms %.4f  kernel %.4f' % (x['frames_per_s'] / 1e6, x['ms_per_step'], x['kernel_ms']))
"
