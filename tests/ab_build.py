"""Development aid: build variants of libgsr_hip.so that differ in -D switches of one translation unit, for A/B timing on
the GPU box (tests/ab_run.sh).  Usage: python tests/ab_build.py name:file.hip:-DX=1,-DY=2 [name2:...]
Variants land in gaussian-splatting-reflection_amd/csrc/_ab/lib_<name>.so (git-ignored, travels with gpurun)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-splatting-reflection_amd", "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402


def main():
    B.build()
    out_dir = os.path.join(CSRC, "_ab")
    os.makedirs(out_dir, exist_ok=True)
    for spec in sys.argv[1:]:
        name, src, defs = (spec.split(":") + ["", ""])[:3]
        defs = [d for d in defs.split(",") if d]
        obj = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
        cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"{name}: hipcc failed\n{r.stderr}")
        objs = [obj if s == src else os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES]
        lib = os.path.join(out_dir, f"lib_{name}.so")
        r = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"{name}: link failed\n{r.stderr}")
        print("built", lib)


if __name__ == "__main__":
    main()
