#!/bin/bash
# The round's closing call ON the GPU box, after the last source change:
#   gpurun --timeout 3000 -- 'bash tools/closing_call.sh r06 [soak seconds] [soak seed]'
# smoke, the whole collection, then the bench lines again with THIS build's counters in place
# (plain, driver-style and 2 ranks self-launched), then -- with a second argument -- soaks.
tag=${1:-r06}
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/$tag; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1
tail -1 $O/smoke.log
bash tools/collect_profiles.sh $tag
cp gpurun_out/${tag}_pmc.json profiles/${tag}_pmc.json      # (the box's copy of the tree: scratch)
python3 bench.py > $O/bench_line_with_pmc.json 2> $O/bench_with_pmc.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_style.json 2>> $O/bench_with_pmc.err
SSW_BENCH_BACKEND=gloo SSW_BENCH_DEVICE=0 python3 bench.py --gpus 2 > $O/bench_line_2ranks_self_launched.json 2>> $O/bench_with_pmc.err
python3 -c "
import json
for n in ('bench_line_with_pmc','bench_line_driver_style','bench_line_2ranks_self_launched'):
    b=json.loads(open('$O/'+n+'.json').read().strip().splitlines()[-1]); print(n, b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['traffic'], b['config5']['wall_ms'])
"
if [ -n "$2" ]; then bash tools/run_soaks.sh ${tag}${3:+_seed$3} $2 ${3:-0}; fi
