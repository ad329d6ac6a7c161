"""Every kernel of libssw_amd.so with its registers, spills, scratch and LDS, from the compiler's
own metadata ("prune or measure the non-default kernel shapes").

    make -C soundswallower_amd/csrc asm          # writes ssw_kernels.s (git-ignored)
    python tools/kernel_shapes.py > profiles/r05_kernel_shapes.txt

CPU only.  The times of the shapes the launch logic can be told to use are in the same file's
second half when --times <json> (written on the GPU box by tools/bench_sen_shapes.sh) is given.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "soundswallower_amd", "csrc", "ssw_kernels.s")
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True, check=True).stdout.split("\n")
    res = []
    for s in out[:len(names)]:
        s = s.replace("(anonymous namespace)::", "").replace("void ", "")
        s = re.sub(r"\(.*$", "", s)            # drop the parameter list
        res.append(s)
    return res


def main():
    text = open(ASM).read()
    meta = text[text.index(".amdgpu_metadata"):]
    kernels = []
    cur = None
    for line in meta.split("\n"):
        m = re.match(r"\s+(?:- )?(\.[a-z_]+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == ".agpr_count" and (cur is None or ".agpr_count" in cur):
            cur = {}
            kernels.append(cur)
        if cur is not None and (key in FIELDS or key == ".name"):
            cur[key] = val
    kernels = [k for k in kernels if ".name" in k]
    names = demangle([k[".name"] for k in kernels])
    rows = []
    for k, n in zip(kernels, names):
        rows.append((n, *[int(k.get(f, 0)) for f in FIELDS]))
    rows.sort()
    print(f"{len(rows)} kernels in libssw_amd.so (gfx950); registers per lane, bytes per lane of "
          "scratch, bytes of static LDS per workgroup")
    print(f"{'kernel':72s} vgpr agpr sgpr vspill sspill scratch   lds")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:6d} {r[5]:6d} {r[6]:7d} {r[7]:5d}")
    spilt = [r for r in rows if r[4] or r[6]]
    print(f"\nkernels with spilt vector registers or scratch: {len(spilt)} of {len(rows)}")
    if len(sys.argv) > 2 and sys.argv[1] == "--times":
        print("\nTimes of the shapes the launch logic can be told to use "
              "(tools/bench_sen_shapes.sh, one box):")
        for k, v in json.load(open(sys.argv[2])).items():
            print(f"  {k:40s} {v}")


if __name__ == "__main__":
    main()
