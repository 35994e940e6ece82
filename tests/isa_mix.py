"""Development aid (runs here, no GPU): static instruction mix of the tile kernels' hot loops from the compiler's assembly, priced with the
issue costs measured by tests/microbench/inst_cost.hip (profiles/r02_inst_cost_microbench.txt: 1.1 ns for a VGPR-operand VOP1/2/3, 1.85 ns
for anything that reads an SGPR / v_min / v_max / v_cmp / v_cndmask / DPP / packed fp32, 3.6 ns for v_exp / v_rcp, per wave64 instruction and
SIMD at >= 4 waves).  Writes profiles/r04_isa_mix.json stamped with the build digest; bench.py quotes the issue bound of the backward's mix
from it while the digest matches (roofline.bound2.issue_bound_of_this_mix).
The hot loop is taken to be the largest basic block of the kernel (the straight-line pair code; its rare side branches are other blocks)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-splatting-reflection_amd", "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402

NS = {"v_fast": 1.1, "v_slow": 1.85, "v_trans": 3.6}
KERNELS = {"surfel_render_bwd_rows_kernel": "gsr_surfel.hip"}   # (the forward's pair code is spread over many small blocks: not modelled this way)


def classify(ins):
    op = ins.split()[0]
    if not op.startswith("v_"):
        return op.split("_")[0]            # s / ds / global / buffer ...
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log")):
        return "v_trans"
    if "dpp" in op or "row_" in ins or "quad_perm" in ins or op.startswith("v_pk_") or op.startswith(("v_min", "v_max", "v_cmp", "v_cndmask")):
        return "v_slow"
    srcs = ins.split(None, 1)[1].split(",")[1:] if " " in ins else []
    if any(re.search(r"\bs\d+\b|\bs\[|vcc|exec", s) for s in srcs):
        return "v_slow"
    return "v_fast"


def main():
    B.build()
    digest = open(os.path.join(CSRC, "_obj", "digest.txt")).read().strip()
    out = {"_config": {"digest": digest, "costs_ns": NS, "what": __doc__.split("\n\n")[0]}}
    asm = {}
    for kernel, src in KERNELS.items():
        if src not in asm:
            path = "/tmp/isa_mix_%s.s" % src.replace(".hip", "")
            cmd = [B.HIPCC] + [f for f in B.FLAGS if f != "-c"] + B.EXTRA_FLAGS.get(src, []) + ["-S", "--cuda-device-only", "-o", path, os.path.join(CSRC, src)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(r.stderr[-2000:])
            asm[src] = open(path).read().splitlines()
        lines = asm[src]
        start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3gsr\d+%s\w*:" % kernel, l))
        end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
        blocks, cur = [], ["entry", []]
        blocks.append(cur)
        for l in lines[start + 1:end]:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = [m.group(1), []]
                blocks.append(cur)
                continue
            t = l.strip()
            if t and not t.startswith((";", ".")):
                cur[1].append(t)
        name, ins = max(blocks, key=lambda b: len(b[1]))
        mix = collections.Counter(classify(i) for i in ins)
        valu = sum(v for k, v in mix.items() if k.startswith("v_"))
        ns = sum(NS[k] * v for k, v in mix.items() if k in NS)
        out[kernel] = {"block": name, "instructions": len(ins), "mix": dict(mix), "valu": valu, "ns_per_trip_per_simd": round(ns, 1),
                       "avg_ns_per_valu": round(ns / valu, 4)}
        print(kernel, out[kernel])
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_isa_mix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
