"""GPU parity of the 16-bit gray conversion (ipx_gray_from_u16_u8) through the C ABI against the oracle's restatement of
cvtColor(BGR2GRAY) on uint16 + `>> 8` (server.py:516-523; parity unpinned vs OpenCV like every cv2 call), and 16-bit PNG
uploads through the Flask contract on the HIP backend.  Byte-exact."""
import io

import numpy as np
import pytest
import torch

import ipx.app as app_mod
import ipx.codec as codec
import ipx.ops as ops
import oracle as O
from png16 import expected_gray, png16

pytestmark = pytest.mark.gpu


def _samples_as_bytes(a, big_endian):
    return np.frombuffer(a.astype(">u2" if big_endian else "<u2").tobytes(), np.uint8).reshape(a.shape[0], -1)


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_every_layout_and_ragged_width(dev, ch):
    rng = np.random.default_rng(ch)
    for h, w in ((1, 1), (3, 7), (5, 8), (9, 2047), (4, 2048), (3, 2049), (2, 4100), (70, 333)):
        a = rng.integers(0, 65536, (h, w, ch), dtype=np.uint16)
        if (h, w) == (5, 8):
            a[:] = 65535                                               # the largest accumulator value
        for be in (True, False):
            for rgb in (True, False):
                raw = _samples_as_bytes(a, be)
                got = ops.gray_from_u16(torch.from_numpy(raw.copy()).to(dev), w, ch, big_endian=be, rgb_order=rgb)
                torch.cuda.synchronize()
                exp = O.gray_u8_from_u16_bytes(raw, w, ch, be, rgb)
                np.testing.assert_array_equal(got.cpu().numpy(), exp, err_msg=f"{h}x{w} ch {ch} be {be} rgb {rgb}")
                if ch >= 3:                                            # and against the array form of the oracle
                    bgr = a[..., [2, 1, 0]] if rgb else a[..., :3]
                    np.testing.assert_array_equal(exp, O.gray_u8_from_u16(np.ascontiguousarray(bgr)))


def test_batch_and_row_strided_source(dev):
    rng = np.random.default_rng(11)
    n, h, w, ch = 3, 40, 301, 3
    a = rng.integers(0, 65536, (n, h, w, ch), dtype=np.uint16)
    raw = np.frombuffer(a.astype(">u2").tobytes(), np.uint8).reshape(n, h, w * ch * 2)
    t = torch.from_numpy(raw.copy()).to(dev)
    exp = O.gray_u8_from_u16_bytes(raw, w, ch)
    np.testing.assert_array_equal(ops.gray_from_u16(t, w, ch).cpu().numpy(), exp)
    wide = torch.zeros((n, h, w * ch * 2 + 37), dtype=torch.uint8, device=dev)
    wide[:, :, :w * ch * 2] = t
    np.testing.assert_array_equal(ops.gray_from_u16(wide[:, :, :w * ch * 2], w, ch).cpu().numpy(), exp)
    np.testing.assert_array_equal(ops.gray_from_u16(wide[:, ::2, :w * ch * 2], w, ch).cpu().numpy(), exp[:, ::2])


def test_4k_16_bit_rgb_frame(dev):
    rng = np.random.default_rng(12)
    h, w = 2160, 3840
    a = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    raw = _samples_as_bytes(a, True)
    got = ops.gray_from_u16(torch.from_numpy(raw.copy()).to(dev), w, 3)
    np.testing.assert_array_equal(got.cpu().numpy(), O.gray_u8_from_u16(np.ascontiguousarray(a[..., ::-1])))


def test_bad_arguments_are_rejected(dev):
    t = torch.zeros((4, 48), dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        ops.gray_from_u16(t, 8, 5)
    with pytest.raises(ValueError):
        ops.gray_from_u16(t, 9, 3)
    with pytest.raises(RuntimeError):
        ops.gray_from_u16(t.cpu(), 8, 3)
    import ipx._lib as L
    with pytest.raises(L.IpxError):
        L.call("ipx_gray_from_u16_u8", 0, t.data_ptr(), t.data_ptr(), 1, 4, 8, 3, 4, 48, 192, 8, 32)      # unknown flag


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_16_bit_png_uploads_through_the_contract(dev, ch):
    rng = np.random.default_rng(30 + ch)
    a = rng.integers(0, 65536, (131, 200) if ch == 1 else (131, 200, ch), dtype=np.uint16)
    blob = png16(a, rng.integers(0, 5, 131))
    exp = expected_gray(a)
    client = app_mod.create_app(backend=app_mod.GpuBackend("cuda:0")).test_client()

    def post(mode, **extra):
        r = client.post("/api/process", data=dict({"mode": mode, "file": (io.BytesIO(blob), "x.png")}, **extra),
                        content_type="multipart/form-data")
        assert r.status_code == 200, r.get_json()
        return codec.from_data_url_png(r.get_json()["dataURL"])
    np.testing.assert_array_equal(post("gray8"), exp)
    np.testing.assert_array_equal(post("gamma", gamma="0.6"), O.gamma_u8(exp, 0.6))
    np.testing.assert_array_equal(post("histogram_eq"), O.equalize_hist_u8(exp))
    np.testing.assert_array_equal(post("median", k="5"), O.median_blur_u8(exp, 5))
    data = {"mode": "threshold", "file": [(io.BytesIO(blob), "a.png"), (io.BytesIO(blob), "b.png")]}
    js = client.post("/api/process_batch", data=data, content_type="multipart/form-data").get_json()
    for r in js["results"]:
        np.testing.assert_array_equal(codec.from_data_url_png(r["dataURL"]), O.threshold_u8(exp, 128))
