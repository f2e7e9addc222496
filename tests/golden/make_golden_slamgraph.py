#!/usr/bin/env python3
"""Generates tests/golden/ref_slamgraph_recorded.npz FROM THE REFERENCE ITSELF: the graph that the reference's own copyDataToG2o (slam_graph.cpp:983-1032, compiled where
it lies into oracle/_ref/libsvs_ref_slamgraph.so against a RECORDING g2o stand-in) hands to g2o for a 9 + 3 keyframe double window -- vertices in addVertex order,
edges in addEdge order (every marginalised pose-pose edge twice, as the reference adds them), estimates, measurements, information.  tests/test_gpu_hipbranch.py feeds
that graph to the HIP back end and to the oracle when oracle/_ref is absent.  Run from the repo root (needs /root/reference):  python tests/golden/make_golden_slamgraph.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as O  # noqa: E402
from scavislam_amd import synth  # noqa: E402
from test_gpu_hipbranch import _graph_tables  # noqa: E402

P, L, n_outer = 12, 300, 3
prob = synth.ba_window(P=P, L=L, seed=5, n_outer=n_outer)
tables = _graph_tables(prob, n_outer, np.random.default_rng(1))
rec = O.ref_slamgraph_optimize(*tables, 2, True, 3.0, 0.0)                    # Backend's own call: OptParams(2, true, 3) (backend.cpp:187)
path = os.path.join(HERE, "ref_slamgraph_recorded.npz")
np.savez_compressed(path, vertices=rec["vertices"], estimates=rec["estimates"], edges=rec["edges"], edge_data=rec["edge_data"],
                    n_cons=np.int32(len(prob["cons"])), n_edges=np.int32(len(prob["edges"])), cam=np.array([prob["cam"][k] for k in ("f", "cx", "cy", "b", "w", "h")]))
print(path, os.path.getsize(path), "bytes;", len(rec["edges"]), "edges recorded")
