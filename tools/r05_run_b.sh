#!/bin/bash
# Round 5, second call: the GPU suite, the bench line, the self-launched 2-rank line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
tail -40 $O/pytest.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -3 $O/bench.err
SSW_BENCH_BACKEND=gloo SSW_BENCH_DEVICE=0 timeout 600 python3 bench.py --gpus 2 --steps 20 --no-extra --no-cpu-baseline > $O/bench_2ranks.json 2> $O/bench_2ranks.err
echo "2-rank rc=$?"; tail -3 $O/bench_2ranks.err
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05b/"
for f in ("bench_line.json","bench_2ranks.json"):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1])
        print(f, d["value"], d["n_gpus"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_kernels"], d.get("scan"))
        c=d.get("config5") or {}
        print("  config5", {k:c.get(k) for k in ("wall_ms","score_ms","align_ms","gather_ms","alignment_crc32","rccl_ranks","n_ranks","per_rank_min")})
        print("  align", d.get("align")); print("  default", d.get("align_default_config"))
    except Exception as e:
        print(f, "unreadable", e)
PY
