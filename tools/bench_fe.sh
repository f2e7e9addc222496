#!/bin/bash
# front-end headline only: python bench.py with the BA / CPU legs as they are (35 s); prints value, ms_per_step, stage times
python bench.py --no-cpu "$@" 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read()); f=d["frontend"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "one-stream", f["ms_per_step_one_stream"], "stages", {k: round(v,3) for k,v in f["stage_ms_per_batch"].items()}, "lat", f["latency_mode_B1"]["ms_per_frame"], "err", f["refined_pose_err_vs_true_motion"], "ok", f["tracking_ok_fraction"], "nc", f["new_college_512x384"]["frames_per_s"])'
