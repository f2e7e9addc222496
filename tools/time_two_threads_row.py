import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import two_threads as TT
sf, sb, cf, cb, errs, wall = TT.run_serial_and_concurrent(n_frames=600, n_rounds=96, device=0)
print(json.dumps(dict(TT.summarize(sf, sb, cf, cb, wall), errors=errs)))
