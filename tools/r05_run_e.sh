#!/bin/bash
# Round 5, fifth call: full GPU suite after the host-edge changes, host boundary, bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 300 python tools/bench_host_boundary.py > $O/host_boundary.json 2> $O/host_boundary.err; cat $O/host_boundary.json; tail -2 $O/host_boundary.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-align --no-extra --steps 20 --warmup 5 > $O/bench_line_20.json 2>> $O/bench.err
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05e/"
for f in ("bench_line.json","bench_line_20.json"):
    d=json.loads(open(O+f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d["roofline"]["frac_kernels"])
    c=d.get("config5") or {}
    print("  config5", {k:c.get(k) for k in ("wall_ms","score_ms","align_ms","gather_ms","alignment_crc32")})
    print("  align", d.get("align"))
PY
