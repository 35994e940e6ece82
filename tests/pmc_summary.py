"""Fold two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected separately: TCC has 4 PMC slots and the
two counters need 3 + 2) into per-kernel averages: profiles/r01_pmc_traffic.json, the file bench.py's roofline.traffic
reads.  Dev aid, not a test.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python bench.py ...
    python tests/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_pmc_traffic.json P W H

Values stay in the counters' own unit (KB, uncorrected); the gfx950 correction (FETCH_SIZE x2,
/opt/skills/guides/MI355X_MICROARCH.md "HBM") is applied where the numbers are used (bench.py: pmc_traffic)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def fold(directory, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                if "gsr::" in name:
                    name = name[name.index("gsr::"):].split("(")[0].split("<")[0]
                else:
                    name = name[:60]
                a = acc[name]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    cfg = [int(x) for x in sys.argv[4:7]] if len(sys.argv) >= 7 else [None] * 3
    fetch, write = fold(fdir, "FETCH_SIZE"), fold(wdir, "WRITE_SIZE")
    res = {"_config": {"P": cfg[0], "W": cfg[1], "H": cfg[2]},
           "_units": "KB per launch, averaged over launches; uncorrected (double FETCH_SIZE on gfx950)"}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("gsr::") and "rocprim" not in k:
            continue
        res[k] = {"FETCH_SIZE": fetch.get(k, (0.0, 0))[0], "WRITE_SIZE": write.get(k, (0.0, 0))[0],
                  "launches": max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        if k.startswith("_"):
            continue
        print("%-50s fetch %10.1f KB (x2 = %8.1f MB)  write %10.1f KB" % (k, v["FETCH_SIZE"], v["FETCH_SIZE"] * 2 / 1024, v["WRITE_SIZE"]))


if __name__ == "__main__":
    main()
