"""Fold rocprofv3 counter passes (each `--pmc` set collected in its own run: TCC has 4 slots and FETCH_SIZE / WRITE_SIZE need
3 + 2, SQ has 8) into per-kernel per-launch averages: profiles/rNN_pmc_summary.json, the file bench.py quotes
`roofline.traffic` and `roofline.bound2` from.  Dev aid, not a test.

    python tests/pmc_summary.py OUT.json P W H "CMD" DIR [DIR ...]

Values stay in the counters' own units (FETCH_SIZE / WRITE_SIZE: KB, uncorrected; the gfx950 correction — FETCH_SIZE x2,
/opt/skills/guides/MI355X_MICROARCH.md "HBM" — is applied where the numbers are used, bench.py: pmc_summary)."""
import csv
import glob
import json
import os
import sys
import time
from collections import defaultdict


def fold(directories):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for directory in directories:
        for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if "gsr::" in name:
                        name = name[name.index("gsr::"):].split("(")[0].split("<")[0]
                    else:
                        name = name[:60]
                    a = acc[name][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    return acc


def main():
    out, P, W, H, cmd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    acc = fold(sys.argv[6:])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dig = os.path.join(root, "gaussian-splatting-reflection_amd", "csrc", "_obj", "digest.txt")
    # which build of the kernels these counters belong to: bench.py quotes them only while the library in the tree has this digest.  The
    # commit is whatever the caller exports (the GPU box has no .git): GSR_COMMIT=$(git rev-parse --short HEAD) in the gpurun command line.
    res = {"_config": {"P": P, "W": W, "H": H, "cmd": cmd, "when": time.strftime("%Y-%m-%d"),
                       "digest": open(dig).read().strip() if os.path.exists(dig) else None, "commit": os.environ.get("GSR_COMMIT", "unrecorded")},
           "_units": "per launch, averaged over launches; FETCH_SIZE / WRITE_SIZE in KB, uncorrected (double FETCH_SIZE on gfx950); SQ_* "
                     "cycle counters in quad-cycles summed over the chip"}
    for k in sorted(acc):
        if not k.startswith("gsr::") and "rocprim" not in k:
            continue
        res[k] = {c: v[0] / v[1] for c, v in sorted(acc[k].items())}
        res[k]["launches"] = max(v[1] for v in acc[k].values())
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        if k.startswith("_") or "render" not in k and "preprocess" not in k:
            continue
        print(k)
        for c, x in v.items():
            print("   %-26s %16.0f" % (c, x))


if __name__ == "__main__":
    main()
