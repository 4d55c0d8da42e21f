"""rt_postprocess_noalloc (the entry point for callers that cannot allocate) against rt_postprocess.

Both run the same three kernels; the only difference is where the per-block reduction slots live (a
fixed device array inside the library instead of a context buffer), so the outputs must be
bit-identical.  Two calls in a row, the second on a smaller image: slots the first call wrote and the
second does not use must not leak into the second result.
"""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(be, torch, src, no_alloc):
    w, h = src.shape[0], src.shape[1]
    dst = torch.full((w, h, 3), -1.0, dtype=torch.float32, device=src.device)   # allocated by the TEST, before the call
    stream = torch.cuda.current_stream().cuda_stream
    be.postprocess(w, h, dst.data_ptr(), src_ptr=src.data_ptr(), scale=0.5, flip_y=True, exposure=2.0, brightness=0.02,
                   contrast=1.1, saturation=1.2, gamma=0.8, stream=stream, no_alloc=no_alloc)
    torch.cuda.synchronize()
    return dst.cpu().numpy()


def test_noalloc_postprocess_equals_the_allocating_entry_twice_in_a_row():
    import torch

    from raytracing_amd.backend.lib import BackendError, HipBackend
    from raytracing_amd.post_processing import PostProcessing

    be = HipBackend(0)
    g = torch.Generator(device="cpu").manual_seed(11)
    big = (torch.rand((1920, 1080, 3), generator=g) * 3.0).to("cuda:0")        # 8100 blocks wanted -> every slot in use
    small = (torch.rand((37, 23, 3), generator=g) * 0.7).to("cuda:0")          # 4 blocks: almost all slots stale
    # "allocates nothing": with every buffer the calls touch allocated beforehand, the device's free memory must not
    # move across three calls of the entry point (a hipMalloc inside it would show here; torch's own caching allocator
    # is kept out of the span: the destination tensors exist before it and nothing is copied to the host inside it)
    stream = torch.cuda.current_stream().cuda_stream
    dst_big = torch.full((1920, 1080, 3), -1.0, dtype=torch.float32, device="cuda:0")
    dst_small = torch.full((37, 23, 3), -1.0, dtype=torch.float32, device="cuda:0")
    kw = dict(scale=0.5, flip_y=True, exposure=2.0, brightness=0.02, contrast=1.1, saturation=1.2, gamma=0.8, stream=stream, no_alloc=True)
    be.postprocess(37, 23, dst_small.data_ptr(), src_ptr=small.data_ptr(), **kw)   # (the runtime loads a kernel's code on its first launch)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for src, dst in ((big, dst_big), (small, dst_small), (big, dst_big)):
        be.postprocess(src.shape[0], src.shape[1], dst.data_ptr(), src_ptr=src.data_ptr(), **kw)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free0, "rt_postprocess_noalloc changed the device's free memory"
    got_big = _run(be, torch, big, True)
    assert np.array_equal(dst_big.cpu().numpy(), got_big)
    got_small = _run(be, torch, small, True)
    want_big = _run(be, torch, big, False)
    want_small = _run(be, torch, small, False)
    assert np.array_equal(got_big, want_big)
    assert np.array_equal(got_small, want_small)
    assert np.isfinite(got_big).all() and got_big.min() >= 0.0 and got_big.max() <= 1.0
    # and both equal the stage-by-stage numpy chain (upstream's arithmetic) on the same input
    pp = PostProcessing(2.0, 0.02, 1.1, 1.2, 0.8)
    ref = pp.process((small.cpu().numpy().astype(np.float64) * 0.5)[:, ::-1])
    assert np.abs(got_small - ref).max() <= 2e-5
    # a third call after the allocating entry has run in between, same answer again
    assert np.array_equal(_run(be, torch, small, True), want_small)
    # beyond the stated maximum: refused, nothing launched
    with pytest.raises(BackendError) as ei:
        be.postprocess(16384, 8192, 1, src_ptr=1, no_alloc=True)
    assert "RT_ERR_INVALID_ARG" in str(ei.value) and "rt_postprocess_noalloc" in str(ei.value)
    be.close()
