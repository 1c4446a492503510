"""The entry point that allocates nothing (ipx_equalize_hist_noalloc_u8, include/ipx.h) against the existing
ipx_equalize_hist_u8, through the C ABI: exact equality of pixels, histogram and table on two calls in a row, the
second with a smaller input (the built-in scratch keeps the first call's contents).  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest
import torch

import ipx._lib as L
import ipx.ops as ops

pytestmark = pytest.mark.gpu


def _both(t):
    """(pixels, hist, table) of the existing entry point and of the no-allocation one for the same device tensor."""
    want, wh, wl = ops.equalize_hist(t, return_aux=True)
    n = t.shape[0] if t.ndim == 4 else 1
    out = torch.zeros_like(t)
    hst = torch.zeros((n, 256), dtype=torch.int32, device=t.device)
    lut = torch.zeros((n, 256), dtype=torch.uint8, device=t.device)
    ops.equalize_hist_noalloc(t, out, hst, lut)
    torch.cuda.synchronize()
    return (want, wh, wl), (out, hst, lut)


def test_noalloc_equalize_matches_existing_entry_point_twice_in_a_row(dev):
    rng = np.random.default_rng(123)
    first = np.clip(rng.normal(110, 35, (6, 1080, 1920, 1)), 0, 255).astype(np.uint8)     # 6 images, many partial blocks
    second = rng.integers(40, 200, (2, 37, 53, 1), dtype=np.uint8)                         # smaller: one block per image
    third = rng.integers(0, 256, (301, 700), dtype=np.uint8)                               # single 2-D image
    for a in (first, second, third):
        t = torch.from_numpy(a).to(dev)
        (want, wh, wl), (out, hst, lut) = _both(t)
        assert torch.equal(out, want)
        assert torch.equal(hst, wh) and torch.equal(lut, wl)
    # without the optional outputs (the table then lives in the built-in scratch) and on a row-strided view
    big = torch.from_numpy(rng.integers(0, 256, (3, 90, 400), dtype=np.uint8)).to(dev)
    view = big[:, 5:80, 7:390].unsqueeze(-1)
    out = torch.zeros((3, 75, 383, 1), dtype=torch.uint8, device=dev)
    ops.equalize_hist_noalloc(view, out)
    assert torch.equal(out, ops.equalize_hist(view))


def test_noalloc_equalize_rejects_batches_beyond_the_builtin_scratch(dev):
    lib = L.load()
    n, h, w = 2000, 4, 4
    assert lib.ipx_hist_workspace_bytes(n, h, w) > L.NOALLOC_SCRATCH_BYTES
    src = torch.zeros((n, h, w, 1), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    with pytest.raises(L.IpxError) as e:
        ops.equalize_hist_noalloc(src, dst)
    assert e.value.code == L.IPX_ERR_UNSUPPORTED and "scratch" in str(e.value)
    # the largest batch the header promises still runs: 256 images of any size
    src = torch.randint(0, 256, (256, 64, 64, 1), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    ops.equalize_hist_noalloc(src, dst)
    assert torch.equal(dst, ops.equalize_hist(src))


def test_noalloc_equalize_can_be_captured_into_a_graph_and_replayed(dev):
    """what the entry point is for: after one warm-up call (which resolves the scratch address) the call is captured into a
    HIP graph -- capture fails on any allocation or synchronisation -- and replayed on new pixels in the same buffers."""
    rng = np.random.default_rng(321)
    t = torch.from_numpy(rng.integers(0, 256, (3, 480, 640, 1), dtype=np.uint8)).to(dev)
    out = torch.zeros_like(t)
    hst = torch.zeros((3, 256), dtype=torch.int32, device=dev)
    lut = torch.zeros((3, 256), dtype=torch.uint8, device=dev)
    ops.equalize_hist_noalloc(t, out, hst, lut)                    # warm-up, outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.equalize_hist_noalloc(t, out, hst, lut)
    for seed in (1, 2):
        t.copy_(torch.from_numpy(np.clip(np.random.default_rng(seed).normal(90 + 40 * seed, 30, (3, 480, 640, 1)), 0, 255).astype(np.uint8)))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        want, wh, wl = ops.equalize_hist(t, return_aux=True)
        assert torch.equal(out, want) and torch.equal(hst, wh) and torch.equal(lut, wl)
