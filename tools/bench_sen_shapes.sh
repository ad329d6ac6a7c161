#!/bin/bash
# The senone kernel's selectable shapes (SSW_SEN_R quads per thread x SSW_SEN_FPB frames per
# group), timed on the headline workload on ONE box.  Run ON the GPU box:
#   bash tools/bench_sen_shapes.sh <tag>     ->  gpurun_out/<tag>/sen_shapes.json
# tools/kernel_shapes.py --times that file puts them beside the compiler's register table.
TAG=${1:-r05}
O=gpurun_out/$TAG
mkdir -p $O
python3 - "$O" <<'PY'
import json, os, subprocess, sys
out = {}
for r, f in (("", ""), ("2", "1"), ("2", "2"), ("3", "1"), ("3", "2"), ("4", "1"), ("4", "2")):
    env = dict(os.environ)
    if r:
        env["SSW_SEN_R"], env["SSW_SEN_FPB"] = r, f
    p = subprocess.run([sys.executable, "bench.py", "--steps", "200", "--warmup", "20",
                        "--no-cpu-baseline", "--no-align", "--no-extra"],
                       env=env, capture_output=True, text=True)
    key = f"R={r} FPB={f}" if r else "default (R=3 FPB=2 on en-us)"
    try:
        b = json.loads(p.stdout.strip().splitlines()[-1])
        out[key] = {"frames_per_s": round(b["value"]), "ms_per_step": round(b["ms_per_step"], 5),
                    "senone_kernel_us": round(1e3 * b["kernels"]["senone_kernel"]["ms"], 2)}
    except Exception as e:  # a shape the model does not admit
        out[key] = {"error": (p.stderr.strip().splitlines() or [repr(e)])[-1][:200]}
json.dump(out, open(os.path.join(sys.argv[1], "sen_shapes.json"), "w"), indent=1)
print(json.dumps(out))
PY
