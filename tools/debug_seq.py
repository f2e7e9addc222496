import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import seq_common as S, oracle as O
from scavislam_amd.ctypes_types import level_cams
camname = 'default'
cam = S.cam_of(camname); cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for hip in (True, False):
    seq = O.RefSequence(cams, hip_branch=hip)
    rec = S.run(seq, camname, n)
    for i, r in enumerate(rec):
        print(hip, i, r['dropped'], r['id_counter'], r['n_new_points'], [len(x) for x in r['lines']], r['fast_thr'], r['T'][:, 3])
    for l in range(3):
        c = seq.recompute_fast_corners(0, l)
        print(' recompute level', l, None if c is None else (len(c), c[:3].tolist()))
    ids, val = seq.new_points(0)
    print(' new points', len(ids), ids[:5].tolist(), np.bincount(ids[:, 1], minlength=3))
    seq.close()
