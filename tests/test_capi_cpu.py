"""CPU tests of the boundary and the host-side logic: the C-ABI library loads and exports every
symbol include/scavislam_hip.h declares (no compute calls without a GPU), POD layouts agree between
C and Python, host bookkeeping mirrors the reference's, and the product never imports the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "scavislam_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(svs_[a-z0-9_]+)\s*\(", hdr)) - {"svs_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    from scavislam_amd import capi
    lib = capi.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/scavislam_hip.h but not exported"
    assert sorted(capi.EXPORTS) == names, "scavislam_amd/capi.py signature table out of sync with the header"


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device svs_ctx_create must fail with SVS_ERR_NO_DEVICE; nothing falls back to the CPU."""
    import torch
    from scavislam_amd import capi
    if torch.cuda.is_available():
        return
    try:
        capi.Context(0)
    except capi.SvsError as e:
        assert "status 3" in str(e) or "svs_ctx_create failed" in str(e)
    else:
        raise AssertionError("Context() succeeded without a GPU")


def test_pod_layouts_match_the_c_header(tmp_path):
    """sizeof/offsetof of every POD struct, compiled from the header with gcc, against ctypes/numpy."""
    from scavislam_amd import capi
    from scavislam_amd.ctypes_types import (BA_CONSTRAINT_DTYPE, BA_EDGE_DTYPE, CANDIDATE_DTYPE, DENSE_SUMS_DTYPE,
                                            GATED_POINT_DTYPE, KEYFRAME_DTYPE, MATCH_RESULT_DTYPE, POINT_STATS_DTYPE, BaParams, BaStats, Cam,
                                            FastGrid, PoseOptParams, PoseOptStats, StereoParams)
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "scavislam_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(svs_cam),sizeof(svs_fastgrid),sizeof(svs_candidate_point),sizeof(svs_keyframe),sizeof(svs_match_result),"
                   "sizeof(svs_dense_sums),sizeof(svs_ba_edge),sizeof(svs_ba_constraint),sizeof(svs_ba_params),sizeof(svs_ba_stats),"
                   "sizeof(svs_match_args),sizeof(svs_dense_track_args),sizeof(svs_stereo_params),sizeof(svs_pose_opt_params),sizeof(svs_pose_opt_stats),"
                   "sizeof(svs_gated_point),sizeof(svs_point_stats));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(Cam), C.sizeof(FastGrid), CANDIDATE_DTYPE.itemsize, KEYFRAME_DTYPE.itemsize, MATCH_RESULT_DTYPE.itemsize,
            DENSE_SUMS_DTYPE.itemsize, BA_EDGE_DTYPE.itemsize, BA_CONSTRAINT_DTYPE.itemsize, C.sizeof(BaParams), C.sizeof(BaStats),
            C.sizeof(capi.MatchArgs), C.sizeof(capi.DenseTrackArgs), C.sizeof(StereoParams), C.sizeof(PoseOptParams), C.sizeof(PoseOptStats),
            GATED_POINT_DTYPE.itemsize, POINT_STATS_DTYPE.itemsize]
    assert got == want
    # the oracle header shares the POD definitions
    src2 = tmp_path / "sz2.c"
    src2.write_text('#include <stdio.h>\n#include "svs_oracle.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",sizeof(svs_cam),'
                    "sizeof(svs_fastgrid),sizeof(svs_candidate_point),sizeof(svs_keyframe),sizeof(svs_match_result),sizeof(svs_dense_sums),"
                    "sizeof(svs_ba_edge),sizeof(svs_ba_constraint));return 0;}\n")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "oracle"), str(src2), "-o", str(exe)])
    assert [int(x) for x in subprocess.check_output([str(exe)]).split()] == want[:8]


def test_host_fastgrid_parameters_match_reference_formulas():
    """scavislam_amd.frontend.fastgrid_for_level (host bookkeeping) == the oracle's restatement of
    stereo_frontend.cpp:73-88 + fast_grid.cpp:23-58, for both shipped camera sizes."""
    import oracle as O
    from scavislam_amd.frontend import fastgrid_for_level
    for (w, h) in ((640, 480), (512, 384)):
        for l in range(3):
            a, b = fastgrid_for_level(w >> l, h >> l, l), O.fastgrid_for_level(w >> l, h >> l, l)
            for f in ("gx", "gy", "cell_w", "cell_h", "min_inner", "min_outer", "max_inner", "max_outer", "fast_min", "fast_max"):
                assert getattr(a, f) == getattr(b, f), (w, h, l, f)
            assert list(a.thr) == list(b.thr)
    g = fastgrid_for_level(640, 480, 0)
    assert (g.gx, g.cell_w, g.cell_h, g.min_outer, g.min_inner, g.max_inner, g.max_outer) == (3, 213, 160, 148, 197, 246, 296)


def test_level_cams_follow_frame_grabber():
    from scavislam_amd.ctypes_types import level_cams
    cams = level_cams(570.342, 320.0, 240.0, 0.075, 640, 480)
    assert (cams[1].w, cams[1].h, cams[2].w, cams[2].h) == (320, 240, 160, 120)
    assert cams[2].f == 570.342 / 4 and cams[2].cx == 80.0 and cams[2].b == 0.3      # baseline * 2^level


def test_shard_problem_partitions_landmarks():
    from scavislam_amd import synth
    from scavislam_amd.backend import shard_problem
    prob = synth.ba_window(10, 700, seed=1)
    seen = np.zeros(len(prob["edges"]), int)
    for world in (2, 4, 8):
        total = 0
        for r in range(world):
            sh = shard_problem(prob, r, world)
            total += len(sh["edges"])
            assert sh["add_pose_terms"] == (r == 0)
            pts = np.unique(sh["edges"]["point"])
            assert np.all((pts // 64) % world == r)             # contiguous chunks of 64, round-robin
        assert total == len(prob["edges"])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "scavislam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "svs_oracle" not in txt and "svs_ref_" not in txt, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "svs_ref_" not in open(os.path.join(ROOT, "include", f)).read()


def test_cpp_adaptor_header_compiles_and_links(tmp_path):
    """include/scavislam_hip.hpp (reference-named C++ adaptor classes) builds with plain g++ -std=c++11
    against the C-ABI library; on a box without a GPU the context reports !ok() instead of crashing."""
    src = tmp_path / "a.cpp"
    src.write_text('#include "scavislam_hip.hpp"\n'
                   'int main(){ scavislam_hip::Context c(0); if(!c.ok()){ std::puts("nodev"); return 0; }\n'
                   '  scavislam_hip::FrameDev fr(c, 64, 48); scavislam_hip::SlamGraphBA ba(c); scavislam_hip::GuidedMatcher m(c);\n'
                   '  std::puts("dev"); return 0; }\n')
    exe = tmp_path / "a"
    libdir = os.path.join(ROOT, "scavislam_amd")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lscavislam_hip", f"-Wl,-rpath,{libdir}"])
    out = subprocess.check_output([str(exe)]).decode().strip()
    assert out in ("nodev", "dev")
