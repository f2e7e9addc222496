"""How long does one sum of the tracker's accept test take?  svs_dense_seq_sum_f32 (the diagnostic entry around exact_seq_sum_f32) on rows of tracker-like terms:
how = 0 the parallel form, 1 the same reading past the caches, 2 the sequential chain; batch = 1 (alone on the device) and 512 (two workgroups per CU)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scavislam_amd import capi

ctx, stream = capi.torch_context(0)
rng = np.random.default_rng(1)
out = {}
for n in (19200, 4800, 1200):
    stride = n + 64
    for batch in (1, 512):
        res = np.clip(rng.normal(0, 0.03, (batch, stride)), -0.1, 0.1).astype(np.float32)
        res[rng.random(res.shape) < 0.2] = 0
        with torch.cuda.stream(stream):
            d_t = torch.as_tensor(res * res).cuda()
            d_o = torch.zeros(batch, dtype=torch.float32, device="cuda")
        for how in (0, 1, 2):
            reps = 20 if how < 2 else 3
            for _ in range(2):
                ctx.call("svs_dense_seq_sum_f32", d_t.data_ptr(), n, stride, batch, how, d_o.data_ptr(), None)
            ctx.sync()
            ctx.call("svs_timer_start")
            for _ in range(reps):
                ctx.call("svs_dense_seq_sum_f32", d_t.data_ptr(), n, stride, batch, how, d_o.data_ptr(), None)
            import ctypes as C
            ms = C.c_float(0)
            ctx.call("svs_timer_stop_ms", C.byref(ms))
            out[f"n{n}_b{batch}_how{how}"] = round(ms.value / reps * 1e3, 2)
print(out)
