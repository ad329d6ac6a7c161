#!/usr/bin/env python3
"""PMC passes over ANY of the repo's bench tools (run ON the GPU box):

    python3 tools/pmc_cmd.py <out.json> <kernel-regex> -- python3 tools/bench_align.py --utts 2048 --reps 1

One `rocprofv3 --pmc <counters of one block> --kernel-trace` pass per counter set (never combined
with another trace domain; the program itself follows `--`, no shell in between), per-kernel
averages over the dispatches whose name matches <kernel-regex>.  tools/pmc_pass.py is the same
thing fixed to the bench step (bench.py quotes it); this one is for the alignment and first-pass
kernels.  HBM bytes per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on
gfx950 FETCH_SIZE reads half the bytes of a wide (16 B per lane) coalesced stream, other widths
are uncalibrated -- raw and doubled figures are both kept.  SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SETS = [["FETCH_SIZE"], ["WRITE_SIZE"],
        ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES",
         "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"],
        ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_BUSY_CYCLES",
         "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS", "SQ_INSTS_BRANCH"],
        ["TCC_HIT_sum", "TCC_MISS_sum"],
        ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
        ["GRBM_GUI_ACTIVE"]]


def main():
    if "--" not in sys.argv or len(sys.argv) < 5:
        raise SystemExit(__doc__)
    k = sys.argv.index("--")
    out_json, kre = sys.argv[1], re.compile(sys.argv[2])
    cmd = sys.argv[k + 1:]
    os.chdir(ROOT)
    env = dict(os.environ, TMPDIR="/tmp")
    tag = os.path.splitext(os.path.basename(out_json))[0]
    per = collections.defaultdict(dict)
    for cs in SETS:
        out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "+".join(cs))
        os.makedirs(out, exist_ok=True)
        full = ["rocprofv3", "--pmc", *cs, "--kernel-trace", "--output-format", "csv", "-d", out,
                "-o", "p", "--"] + [c if not c.endswith(".py") else os.path.join(ROOT, c) for c in cmd]
        r = subprocess.run(full, env=env, capture_output=True, text=True, cwd="/tmp")
        files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
        if not files:
            print("no counter output for", cs, r.stderr[-300:], file=sys.stderr)
            continue
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(files[0])):
            m = kre.search(row["Kernel_Name"])
            if not m:
                continue
            key = (m.group(0), row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
        for (kn, c), (v, n) in acc.items():
            per[kn][c] = v / n
            per[kn]["dispatches"] = n
        for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
            os.remove(f)
    for kn, v in per.items():
        if "FETCH_SIZE" in v:
            v["fetch_bytes_raw"] = v["FETCH_SIZE"] * 1024
            v["fetch_bytes_doubled"] = 2 * v["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in v:
            v["write_bytes"] = v["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in v and v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0) > 0:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    res = {"source": "tools/pmc_cmd.py: rocprofv3 --pmc <one counter set per pass> --kernel-trace "
                     "-- " + " ".join(cmd) + "; averages over the dispatches of each matching "
                     "kernel; FETCH_SIZE / WRITE_SIZE in KiB",
           "kernel_src_sha": bench.kernel_src_sha(), "per_kernel": per}
    with open(os.path.join(ROOT, out_json), "w") as fh:
        json.dump(res, fh, indent=1)
    for kn, v in per.items():
        print(kn, {c: (round(x, 3) if x < 10 else round(x)) for c, x in v.items()})


if __name__ == "__main__":
    main()
