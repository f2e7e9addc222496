#!/usr/bin/env python3
"""profiles/pmc_latest.json: HBM bytes per launch of the roofline kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
runs, as MI355X_MICROARCH.md prescribes), keyed by kernel, with the hash of the kernel's source file at profiling time: bench.py reports
`roofline.traffic` from here and prints null when the source has changed since.
usage: make_pmc_latest.py <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> <bench.log>   (the .txt files are tools/pmc_any.py outputs)"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([0-9.]+)(?:\s+max=\s*([0-9.]+))?", line)
        if m:
            out[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)) * 1024.0, float(m.group(5) or m.group(4)) * 1024.0)      # the counters are in KB
    return out


def sha(rel):
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]


def main(fetch_txt, write_txt, bench_log, df_fetch_txt=None, df_write_txt=None, df_log=None):
    F, W = parse(fetch_txt), parse(write_txt)
    bench = json.loads([l for l in open(bench_log).read().splitlines() if l.startswith("{")][-1])
    res = {"note": "bytes per launch; FETCH_SIZE counts a 16-byte-per-lane streaming read at half its size on gfx950 (MI355X_MICROARCH.md, HBM section): "
                   "`traffic` applies that correction to the part of the fetch that is such a stream, `fetch_raw` / `write_raw` are the counters as read",
           "kernels": {}}

    def pick(sub):
        ks = [k for k in F if sub in k]
        return max(ks, key=lambda k: F[k][1]) if ks else None
    k = pick("ba_landmark_kernel<0")
    if k:
        f, w = F[k][1], W.get(k, (0, 0.0, 0.0))[1]
        res["kernels"]["ba_landmark_kernel<0>"] = {
            "kernel": k, "launches": F[k][0], "fetch_raw": round(f), "write_raw": round(w), "traffic": round(2 * f + w),
            "correction": "edge records, poses and psi arrive as 16 B/lane loads: 2 x FETCH_SIZE + WRITE_SIZE",
            "workload": {"keyframes": bench["schur"]["keyframes"], "landmarks": bench["schur"]["landmarks"], "edges": bench["schur"]["edges"]},
            "source": "scavislam_amd/csrc/ba_schur.inc", "source_sha16": sha("scavislam_amd/csrc/ba_schur.inc")}
    # the dominant kernel of every front-end stage at the bench's batch size (the launch with the largest fetch is the batched one): raw counters per launch
    B = bench["config"]["batch_streams_per_gpu"]
    for key, sub, src in (("fast_score_kernel", "fast_score_kernel", "fast.hip"), ("match_kernel3", "match_kernel3", "match.hip"),
                          ("dense_track_cpu_sem_kernel", "dense_track_cpu_sem_kernel", "dense.hip"), ("motion_only_fused_kernel", "motion_only_fused_kernel", "dense.hip"),
                          ("stereo_bm_kernel", "stereo_bm_kernel", "stereo.hip"), ("stereo_speckle_strip_kernel", "stereo_speckle_strip_kernel", "stereo.hip"),
                          ("pyr_down_u8_kernel", "pyr_down_u8_kernel", "image.hip")):
        k = pick(sub)
        if sub == "dense_track_cpu_sem_kernel":      # the instantiation the bench batch runs by default: grid order by last frame's work (5th argument true), one workgroup per stream
            bal = [kk for kk in F if sub in kk and kk.rstrip().endswith("false, true>")]
            k = max(bal, key=lambda kk: F[kk][2]) if bal else k
            # round 6: the bench batch runs the flat kernel (first launch: <U8SRC, BAL, 0>; the continuation <., ., 1> is reported beside it)
            flat = [kk for kk in F if "dense_track_batch_kernel" in kk and ", 0>" in kk]
            if flat:
                k = max(flat, key=lambda kk: F[kk][2])
                cont = [kk for kk in F if "dense_track_batch_kernel" in kk and ", 1>" in kk]
                if cont:
                    kc = max(cont, key=lambda kk: F[kk][2])
                    res["kernels"]["dense_track_batch_kernel_continuation"] = {"kernel": kc, "launches_all_batch_sizes": F[kc][0], "fetch_raw": round(F[kc][2]), "write_raw": round(W.get(kc, (0, 0.0, 0.0))[2]),
                                                                               "workload": {"batch_streams_per_gpu": B}, "source": "scavislam_amd/csrc/dense.hip", "source_sha16": sha("scavislam_amd/csrc/dense.hip")}
        if k:
            f, w = F[k][2], W.get(k, (0, 0.0, 0.0))[2]
            res["kernels"][key] = {"kernel": k, "launches_all_batch_sizes": F[k][0], "fetch_raw": round(f), "write_raw": round(w),
                                   "note": "the LARGEST launch of the run = the launch over all streams of the bench's batch (bench.py also launches smaller batches)",
                                   "workload": {"batch_streams_per_gpu": B}, "source": "scavislam_amd/csrc/" + src, "source_sha16": sha("scavislam_amd/csrc/" + src)}
    if df_fetch_txt and "dense_full" in bench:
        Fd, Wd = parse(df_fetch_txt), parse(df_write_txt)
        ks = [k for k in Fd if "<false" in k]
        line = [l for l in open(df_log).read().splitlines() if l.startswith("B=") and "fuse=0" in l]
        if ks and line:
            k = ks[0]
            alg = float(re.search(r"alg\s+([0-9.]+) MB", line[-1]).group(1)) * 1e6
            f, w = Fd[k][1], Wd.get(k, (0, 0.0, 0.0))[1]
            traffic = f + alg / 4 + w
            res["kernels"]["dense_track_full_kernel"] = {
                "kernel": k, "launches": Fd[k][0], "fetch_raw": round(f), "write_raw": round(w), "alg_bytes_of_the_profiled_launch": round(alg),
                "traffic": round(traffic), "traffic_over_algorithmic": round(traffic / alg, 4),
                "correction": "only the float4 cloud stream (16 of the 32 algorithmic B/px) is a 16 B/lane load: FETCH_SIZE + (algorithmic bytes / 2) / 2 + WRITE_SIZE",
                "workload": {"streams_per_launch": 64, "command": "tools/time_dense_full.py 64 (the batched launches only)"},
                "source": "scavislam_amd/csrc/dense_full.hip", "source_sha16": sha("scavislam_amd/csrc/dense_full.hip")}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:7])
