"""The accept test of the quarter-grid tracker on the GPU (csrc/seqsum.h + dense.hip exact_seq_sum_f32): the value of the reference's sequential
`float chi2 += res * res` (dense_tracking.cpp:229-262) formed in parallel must carry the bits of the sequential sum -- on the tracker's kind of terms and on hostile
ones (ties, carries at powers of two, huge dynamic range, zeros), for both ways the tracker reads the terms, at every sample count of the configurations."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(rng, n, kinds):
    rows = []
    for kind in kinds:
        if kind == "tracker":
            res = np.clip(rng.normal(0, rng.choice([0.003, 0.01, 0.03, 0.08]), n), -0.1, 0.1).astype(np.float32)
            res[rng.random(n) < rng.choice([0.0, 0.2, 0.7])] = 0
            t = res * res
        elif kind == "ties":
            t = (rng.integers(0, 64, n) * 2.0 ** rng.integers(-30, -8)).astype(np.float32)
        elif kind == "const":
            t = np.full(n, rng.choice([0.01, 0.25, 1.0, 3.0]), np.float32)
        elif kind == "half_ulp":
            t = np.full(n, 2.0 ** -int(rng.integers(20, 30)), np.float32)
            t[0] = 2.0 ** int(rng.integers(-3, 6))
        elif kind == "range":
            t = (10.0 ** rng.uniform(-35, 3, n)).astype(np.float32)
            t[rng.random(n) < 0.3] = 0
        elif kind == "sparse":
            t = np.zeros(n, np.float32)
            idx = rng.integers(0, n, max(1, n // 7))
            t[idx] = rng.integers(1, 1 << 20, len(idx)).astype(np.uint32).view(np.float32)
        else:
            t = np.zeros(n, np.float32)
        rows.append(t.astype(np.float32))
    return np.stack(rows)


@pytest.mark.parametrize("n", [19200, 4800, 1200, 12288, 1, 63, 64, 65, 511, 513, 12345, 230400])
def test_seq_sum_bits(gpu_ctx, n):
    import torch
    ctx, stream = gpu_ctx
    rng = np.random.default_rng(n)
    kinds = ["tracker"] * 10 + ["ties", "const", "half_ulp", "range", "sparse", "zeros"] * 2
    t = _rows(rng, n, kinds)
    want = np.array([np.cumsum(r, dtype=np.float32)[-1] for r in t], np.float32)
    stride = (n + 3) // 4 * 4 + 64                 # rows 16-byte aligned and padded like the tracker's term buffers (a lane reads its run as float4)
    tp = np.zeros((len(t), stride), np.float32)
    tp[:, :n] = t
    tp[:, n:] = 1e30                                 # whatever lies behind a row must not matter
    with torch.cuda.stream(stream):
        d_t = torch.as_tensor(tp).cuda()
        d_out = torch.zeros((3, len(t)), dtype=torch.float32, device="cuda")
        d_fb = torch.zeros((3, len(t)), dtype=torch.int32, device="cuda")
    for how in range(3):
        ctx.call("svs_dense_seq_sum_f32", d_t.data_ptr(), n, stride, len(t), how, d_out[how].data_ptr(), d_fb[how].data_ptr())
    ctx.sync()
    out, fb = d_out.cpu().numpy(), d_fb.cpu().numpy()
    for how in range(3):
        assert np.array_equal(out[how].view(np.uint32), want.view(np.uint32)), (how, n, np.nonzero(out[how].view(np.uint32) != want.view(np.uint32))[0], out[how], want)
    # on the tracker's kind of terms no self-check of the parallel form fails (the chain is never needed); sums of more than 20 480 terms (images beyond
    # 640 x 512) take the chain by design
    assert fb[:2].all() if n > 20480 else not fb[:2, :10].any(), fb[:2]
    print(f"n = {n}: {len(t)} rows bit-equal to the sequential float sum (parallel form, past-the-caches form, chain); fallbacks on hostile rows: {int(fb[:2, 10:].sum())}")
