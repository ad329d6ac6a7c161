#!/bin/bash
# closing call after the last source change (self-test clock): smoke, the collection, the bench
# lines with this build's counters in place
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -1 $O/smoke.log
bash tools/collect_profiles.sh r05
cp gpurun_out/r05_pmc.json profiles/r05_pmc.json          # (the box's copy of the tree: scratch)
python3 bench.py > $O/bench_line_with_pmc.json 2> $O/bench_with_pmc.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_style.json 2>> $O/bench_with_pmc.err
SSW_BENCH_BACKEND=gloo SSW_BENCH_DEVICE=0 python3 bench.py --gpus 2 > $O/bench_line_2ranks_self_launched.json 2>> $O/bench_with_pmc.err
python3 -c "
import json
for n in ('bench_line_with_pmc','bench_line_driver_style','bench_line_2ranks_self_launched'):
    b=json.loads(open('$O/'+n+'.json').read().strip().splitlines()[-1]); print(n, b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['traffic'], b['scan'], b['config5']['wall_ms'])
"
