import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """Context + torch stream on cuda:0; fails loudly (no fallback) if the HIP library is missing."""
    import torch
    from scavislam_amd import capi
    assert torch.cuda.is_available(), "GPU test collected without a GPU"
    ctx, stream = capi.torch_context(0)
    yield ctx, stream
    ctx.close()


@pytest.fixture(scope="session")
def scene_frames():
    """Three consecutive synthetic 640x480 frames + exact disparity."""
    from scavislam_amd import synth
    sc = synth.Scene(2011)
    traj = synth.trajectory(8)
    frames = [sc.render(synth.CAM_DEFAULT, traj[i], seed=i) for i in (0, 5, 6)]
    return dict(cam=synth.CAM_DEFAULT, poses=[traj[i] for i in (0, 5, 6)], frames=frames)
