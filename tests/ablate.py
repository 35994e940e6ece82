"""Development ablation: time the S tile kernels under gsr_set_option("dev", bits).  Not a test."""
import json, subprocess, sys, os
for dev in [int(a, 0) for a in sys.argv[1:]] or [0]:
    env = dict(os.environ, GSR_DEV=str(dev))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("dev", hex(dev), "ms/step", d["ms_per_step"], {k: d["stage_ms_per_step"][k] for k in ("render_fwd", "render_bwd", "refl_bwd", "sort")}, flush=True)
    except Exception as e:
        print("dev", dev, "failed", r.stderr[-800:])
