#!/usr/bin/env python3
"""PMC passes over the bench step (run ON the GPU box: `python3 tools/pmc_pass.py <tag>`).

One `rocprofv3 --pmc <counters of one block> --kernel-trace` pass per counter set (never
combined with another trace domain), over `python3 bench.py --steps 5 --warmup 2 --no-align
--no-extra --no-cpu-baseline` with the spin-up loop off; per-kernel averages go to
gpurun_out/<tag>_pmc.json together with the hash of the kernel sources they were taken on
(bench.py quotes `roofline.traffic` from profiles/<tag>_pmc.json only when that hash is the
tree's).  HBM bytes per MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are KiB;
on gfx950 FETCH_SIZE reads half the bytes of a wide (16 B per lane) coalesced stream, other
widths are uncalibrated -- both the raw figure and the doubled one are kept."""
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
        ["SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT",
         "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES"],
        ["GRBM_GUI_ACTIVE"]]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    os.chdir(ROOT)
    env = dict(os.environ, TMPDIR="/tmp", SSW_BENCH_NO_SPIN="1")
    per = collections.defaultdict(dict)
    sets = list(SETS)
    if os.environ.get("SSW_PMC_EXTRA"):     # "A,B;C,D": further counter sets, one pass each
        only = os.environ.get("SSW_PMC_ONLY_EXTRA")
        sets = ([] if only else sets) + [x.split(",") for x in os.environ["SSW_PMC_EXTRA"].split(";")]
    for cs in sets:
        out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "+".join(cs))
        os.makedirs(out, exist_ok=True)
        cmd = ["rocprofv3", "--pmc", *cs, "--kernel-trace", "--output-format", "csv", "-d", out,
               "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-align",
               "--no-extra", "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd="/tmp")
        files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
        if not files:
            print("no counter output for", cs, r.stderr[-300:], file=sys.stderr)
            continue
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(files[0])):
            m = re.search(r"(ptm_\w+|ms_\w+|viterbi\w+|first_pass\w+)", row["Kernel_Name"])
            if not m:
                continue
            key = (m.group(1), row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
        for (k, c), (v, n) in acc.items():
            per[k][c] = v / n
            per[k]["dispatches"] = n
    step = {k: v for k, v in per.items() if k in ("ptm_topn_mfma_kernel", "ptm_topn_frames_kernel",
                                                   "ptm_senone_kernel")}
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in step.values()) * 1024
    write = sum(v.get("WRITE_SIZE", 0.0) for v in step.values()) * 1024
    res = {
        "source": "tools/pmc_pass.py: rocprofv3 --pmc <one counter set per pass> --kernel-trace -- "
                  "python3 bench.py --steps 5 --warmup 2 --no-align --no-extra --no-cpu-baseline; "
                  "averages over the dispatches of each kernel; FETCH_SIZE / WRITE_SIZE in KiB",
        "workload": "bench.py default: en-us PTM, 4096 frames per step",
        "kernel_src_sha": bench.kernel_src_sha(),
        "per_kernel": per,
        "fetch_bytes_raw": fetch,
        "fetch_bytes_doubled": 2 * fetch,
        "write_bytes": write,
        "hbm_bytes": 2 * fetch + write,
        "hbm_bytes_note": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced "
                          "streams; an upper estimate for this step's mixed access widths) + "
                          "WRITE_SIZE",
        "valu_wave_instr_per_step": sum(v.get("SQ_INSTS_VALU", 0.0) for v in step.values()),
    }
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.json")
    with open(path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: res[k] for k in ("kernel_src_sha", "fetch_bytes_raw", "write_bytes",
                                          "hbm_bytes", "valu_wave_instr_per_step")}))
    for k, v in step.items():
        print(k, {c: round(x) for c, x in v.items()})


if __name__ == "__main__":
    main()
