import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import seq_common as S, oracle as O
from scavislam_amd.ctypes_types import level_cams
camname = sys.argv[2] if len(sys.argv) > 2 else 'default'
cam = S.cam_of(camname); cams = level_cams(cam["f"], cam["cx"], cam["cy"], cam["b"], cam["w"], cam["h"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
recs = []
for hip in (True, False):
    seq = O.RefSequence(cams, hip_branch=hip)
    if hip and len(sys.argv) > 3: seq.set_var('svs.trk_seq_chi2', int(sys.argv[3]))
    recs.append(S.run(seq, camname, n))
    seq.close()
a, b = recs
for i, (ra, rb) in enumerate(zip(a, b)):
    ia, ib = S.points_int(ra), S.points_int(rb)
    same = ia.shape == ib.shape and np.array_equal(ia, ib) and np.array_equal(S.decisions(ra), S.decisions(rb)) and np.array_equal(ra['fast_thr'], rb['fast_thr'])
    dT = np.abs(ra["T"] - rb["T"]).max()
    msg = f"{i}: int {same} dT {dT:.2e} lines {[len(x) for x in ra['lines']]} {[len(x) for x in rb['lines']]} drop {ra['dropped']}/{rb['dropped']} sw {ra['switched']}/{rb['switched']} key {ra['actkey_id']}/{rb['actkey_id']}"
    if ra['dropped'] and rb['dropped'] and ra['new_val'].shape == rb['new_val'].shape:
        d = np.abs(ra['new_val'] - rb['new_val'])
        msg += f" newval maxdiff {d.max(0)}"
    if not same:
        msg += f" thr_eq {np.array_equal(ra['fast_thr'], rb['fast_thr'])}"
        for l in range(3):
            sa = set(map(tuple, ra['lines'][l][:, :3])); sb = set(map(tuple, rb['lines'][l][:, :3]))
            msg += f" L{l}: only_hip {len(sa - sb)} only_ref {len(sb - sa)}"
    if len(ra['lines'][0]) == len(rb['lines'][0]) and len(ra['lines'][0]):
        msg += f" p2diff {max(np.abs(ra['lines'][l][:, 3:5] - rb['lines'][l][:, 3:5]).max() for l in range(3) if len(ra['lines'][l]) == len(rb['lines'][l]) and len(ra['lines'][l])):.2e}"
    if (not same) or i % 10 == 0 or ra['dropped'] or dT > 1e-9: print(msg)

st = S.compare(a, b, "hip vs ref")
print(st)
