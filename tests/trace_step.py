"""Development aid: timeline of one headline step from a rocprofv3 kernel trace (tests/trace_step.sh): start offset, duration,
stream/queue and name of every kernel between two consecutive surfel_preprocess_kernel launches that enclose a backward."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[3] if len(sys.argv) > 3 else "surfel_preprocess_kernel"      # (C5: gauss_preprocess_kernel)
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
steps = [(a, b) for a, b in zip(idx, idx[1:]) if any("render_bwd" in r["Kernel_Name"] for r in rows[a:b])]
a, b = steps[min(len(steps) - 1, int(sys.argv[2]) if len(sys.argv) > 2 else 4)]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f .. %8.1f  %7.1f us  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                                                       r["Kernel_Name"].split("(")[0][-60:]))
print("step: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
