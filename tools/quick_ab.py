"""kernel A/B on the GPU box: `python tools/quick_ab.py "trk_split=0" "trk_split=8" ...` runs `bench.py --quick` once per option set (SVS_CTX_OPTIONS) and prints one line each"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for opts in sys.argv[1:] or [""]:
    env = dict(os.environ, SVS_CTX_OPTIONS=opts)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick"], env=env, capture_output=True, text=True)
    try:
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        print(f"{opts or 'default':28s} value {d['value']:9.1f}  ms/step {d['ms_per_step']:.4f}  f64-accept {d['ms_per_step_f64_accept']:.4f}  one-stream {d['ms_one_stream']:.4f}  "
              f"stages {' '.join('%s %.3f' % (k[:5], v) for k, v in d['stage_ms'].items())}  B1 tracker {d['latency_B1_ms']['parity_tracker_ms']:.4f}  err {d['track_err']:.2e} passes {d['passes']}", flush=True)
    except Exception as e:
        print(opts, "FAILED", repr(e), r.stdout[-300:], r.stderr[-600:], flush=True)
