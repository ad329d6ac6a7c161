#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   (one rocprofv3 --pmc pass per counter)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
for c in "$@"; do
  out=gpurun_out/pmc_$tag/$c
  mkdir -p "$out"
  rocprofv3 --pmc "$c" --kernel-trace --output-format csv -d "$out" -o p -- python3 bench.py --steps 5 --warmup 2 > "$out/bench.log" 2>&1
  python3 - "$out" "$c" <<'PY'
import csv, sys, glob, collections
out, c = sys.argv[1], sys.argv[2]
f = glob.glob(out + '/*counter_collection.csv')
if not f:
    print(c, 'no output'); sys.exit(0)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for k, (v, n) in acc.items():
    print(f'{c:28s} {k:42s} avg {v / n:14.1f} over {n}')
PY
done
