"""trace_nms_noalloc (the entry point that allocates nothing: its scratch is a static __device__ array inside
libtrace_amd.so) against trace_nms with a caller workspace, through the C-ABI: bit-identical keep lists on two calls
in a row, the second with a smaller input (the array keeps the first call's contents), then the size limit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _boxes(rng, n):
    x1 = rng.uniform(0, 160, n)
    y1 = rng.uniform(0, 160, n)
    w = rng.uniform(16, 96, n)
    h = rng.uniform(16, 96, n)
    return np.stack([x1, y1, np.minimum(x1 + w, 223), np.minimum(y1 + h, 223)], 1).astype(np.float32)


def _call(lib, L, name, boxes, scores, thresh, mode, ws=None):
    n = boxes.shape[0]
    keep = torch.full((max(n, 1),), -1, dtype=torch.int64, device=DEV)
    num = torch.full((1,), -7, dtype=torch.int32, device=DEV)
    args = [L.ptr(boxes), L.ptr(scores), n, float(thresh), int(mode), L.ptr(keep), L.ptr(num)]
    if ws is not None:
        args += [L.ptr(ws), ws.numel()]
    rc = getattr(lib, name)(*args, L.stream())
    return rc, keep, num


def test_nms_noalloc_matches_nms_twice_then_rejects_oversize():
    from trace_amd import _lib as L
    from oracle import c_oracle
    lib = L.lib()
    rng = np.random.default_rng(123)
    with torch.cuda.device(0):
        for n, thresh, mode in ((3000, 0.5, 0), (700, 0.7, 1)):      # second call: smaller input, other rule
            b_np = _boxes(rng, n)
            s_np = (rng.permutation(n) / n * 0.9 + 0.05).astype(np.float32)
            b, s = torch.from_numpy(b_np).to(DEV), torch.from_numpy(s_np).to(DEV)
            ws = torch.empty((lib.trace_nms_workspace_bytes(n),), dtype=torch.uint8, device=DEV)
            rc0, keep0, num0 = _call(lib, L, "trace_nms", b, s, thresh, mode, ws)
            rc1, keep1, num1 = _call(lib, L, "trace_nms_noalloc", b, s, thresh, mode)
            assert rc0 == 0 and rc1 == 0, lib.trace_last_error()
            torch.cuda.synchronize()
            k = int(num0.item())
            assert int(num1.item()) == k and 0 < k < n
            assert torch.equal(keep0[:k], keep1[:k])
            assert np.array_equal(keep1[:k].cpu().numpy(), c_oracle.nms(b_np, s_np, thresh, mode))
        big = torch.zeros((4097, 4), dtype=torch.float32, device=DEV)
        rc, _, _ = _call(lib, L, "trace_nms_noalloc", big, torch.zeros((4097,), device=DEV), 0.5, 0)
        assert rc != 0 and b"4096" in lib.trace_last_error()
        rc, _, num = _call(lib, L, "trace_nms_noalloc", big[:0], torch.zeros((0,), device=DEV), 0.5, 0)   # empty in, empty out
        torch.cuda.synchronize()
        assert rc == 0 and int(num.item()) == 0
