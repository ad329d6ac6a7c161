#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter-set> [<counter-set> ...]
# One rocprofv3 --pmc pass per argument; an argument may hold several counters of one block
# separated by commas ("SQ_WAVE_CYCLES,SQ_WAIT_ANY").  --pmc is never combined with a trace
# domain other than --kernel-trace.  Prints per-kernel averages; raw CSVs stay under
# gpurun_out/pmc_<tag>/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
for c in "$@"; do
  out=gpurun_out/pmc_$tag/$(echo "$c" | tr ',' '+')
  mkdir -p "$out"
  rocprofv3 --pmc $(echo "$c" | tr ',' ' ') --kernel-trace --output-format csv -d "$out" -o p -- python3 bench.py --steps 5 --warmup 2 --no-align --no-extra --no-cpu-baseline > "$out/bench.log" 2>&1
  python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
if not f:
    print(out, 'no output'); sys.exit(0)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    k = (r['Counter_Name'], r['Kernel_Name'].split('(')[0].split('<')[0][-34:])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (c, k), (v, n) in sorted(acc.items()):
    print(f'{c:28s} {k:36s} avg {v / n:16.1f} over {n}')
PY
done
