"""CPU suite for the host side: server.py's HTTP contract (statuses, keys, messages, params, aliases, clamps)
replayed against tests/golden/server_contract.json, the batch sharder's ordering, and codec round trips.
The arithmetic behind the dispatch is the test-only oracle backend here; tests/test_gpu_host.py repeats the
contract run with the real GPU backend."""
import io
import json

import numpy as np
import pytest
from PIL import Image

import ipx.app as app_mod
import ipx.codec as codec
from ipx.sharder import BatchSharder, shard_range
from oracle_ops import OracleBackend, OracleOps
import oracle as O


def png(a):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    return buf.getvalue()


def post(client, img, mode, **form):
    data = {k: str(v) for k, v in form.items()}
    if mode is not None:
        data["mode"] = mode
    if img is not None:                                   # a 1-D array is raw file bytes (the "garbage" golden input)
        data["file"] = (io.BytesIO(img.tobytes() if img.ndim == 1 else png(img)), "x.png")
    r = client.post("/api/process", data=data, content_type="multipart/form-data")
    return r.status_code, r.get_json()


@pytest.fixture(scope="module")
def client():
    return app_mod.create_app(backend=OracleBackend()).test_client()


def replay_contract(client, golden):
    g, meta = golden
    n = 0
    for c in meta["cases"]:
        img = g["in_" + c["input"]] if c["input"] else None
        st, js = post(client, img, c["mode"], **c["form"])
        assert st == c["status"], c["name"]
        if st == 200:
            assert set(js) == {"dataURL", "message", "params"}
            assert js["message"] == c["message"], c["name"]
            assert js["params"] == c["params"], c["name"]      # incl. denoise_manual's rounded float32 statistics
            np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), g["out_" + c["name"]], err_msg=c["name"])
        else:
            assert js == {"error": c["error"]}, c["name"]
        n += 1
    return n


def test_contract_replay_matches_real_server(client, golden):
    assert replay_contract(client, golden) >= 70


def test_opencv_backed_modes_contract(client):
    rng = np.random.default_rng(0)
    gray = rng.integers(0, 256, (40, 52), dtype=np.uint8)
    rgb = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    bgr = rgb[:, :, ::-1]
    st, js = post(client, gray, "histogram_eq")
    assert st == 200 and js["message"] == "Histogram Equalized" and js["params"] == {"method": "Histogram Equalization"}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.equalize_hist_u8(gray))
    st, js = post(client, rgb, "histogram_eq")                       # colour input -> gray first (server.py:245)
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.equalize_hist_u8(O.bgr2gray_u8(bgr)))
    st, js = post(client, rgb, "sharpening")
    assert st == 200 and js["message"] == "Sharpened: Laplacian" and js["params"] == {"method": "Laplacian", "color": True}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.laplacian_sharpen_u8(bgr))
    st, js = post(client, gray, "sharpening")                        # IMREAD_COLOR makes gray 3-channel (server.py:252)
    assert js["params"]["color"] is True
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.laplacian_sharpen_u8(np.repeat(gray[:, :, None], 3, 2)))
    st, js = post(client, rgb, "sharpening", sharp_type="other")
    assert js["message"] == "Sharpened: No sharpening" and js["params"]["method"] == "No sharpening"
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), bgr)
    for T, key in ((128, "threshold"), (0, "T"), (-3, "threshold"), (255, "T"), (300, "threshold")):
        st, js = post(client, gray, "threshold", **{key: T})
        assert st == 200 and js["params"] == {"threshold": T} and js["message"] == "ok"
        np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.threshold_u8(gray, T))
    st, js = post(client, gray, "threshold")                         # default 128 (server.py:379)
    assert js["params"] == {"threshold": 128}
    st, js = post(client, gray, "threshold", threshold="abc")        # int('abc') -> 500 with the exception text
    assert st == 500 and "invalid literal" in js["error"]
    st, js = post(client, rgb, "gray8")
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.bgr2gray_u8(bgr))
    st, js = post(client, rgb, "gamma", gamma=0.6)                   # fused gray + table
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.gamma_u8(O.bgr2gray_u8(bgr), 0.6))
    st, js = post(client, rgb, "log", c=2.0)
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.log_u8(O.bgr2gray_u8(bgr), 2.0))


def test_out_of_scope_modes_say_so(client):
    gray = np.zeros((8, 8), np.uint8)
    for mode in app_mod.OUT_OF_SCOPE:
        st, js = post(client, gray, mode)
        assert st == 501 and "outside the GPU hot path" in js["error"]
    for method in app_mod.DENOISE_OUT_OF_SCOPE + ("whatever",):
        st, js = post(client, gray, "denoise_manual", method=method)
        assert st == 501 and "outside the GPU hot path" in js["error"]
    st, js = post(client, gray, "median", nlm=1)
    assert st == 501
    r = client.post("/api/process", data={"mode": "gray8", "file": (io.BytesIO(b"not an image"), "x.png")},
                    content_type="multipart/form-data")
    assert r.status_code == 500 and r.get_json()["error"] == "Cannot decode image to grayscale 8-bit. unknown error"


def test_median_modes_contract(client):
    """mode=median (server.py:390-441) and denoise_manual's default method (server.py:184, 214-216): parameter
    parsing / clamps / params / messages as the reference; pixels against the oracle restatement (OpenCV unpinned)."""
    rng = np.random.default_rng(8)
    gray = rng.integers(0, 256, (40, 52), dtype=np.uint8)
    gray[rng.random(gray.shape) < 0.05] = 255
    st, js = post(client, gray, "median")                            # defaults: k=3, no gauss, no morph, 1 pass
    assert st == 200 and js["message"] == "ok" and js["params"] == {"median_k": 3, "pre_gauss": 0, "morph_k": 0, "passes": 1}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.median_blur_u8(gray, 3))
    st, js = post(client, gray, "median", k=4, gauss=2, morph=2, passes=9)      # even -> +1, passes clamp 4
    assert js["params"] == {"median_k": 5, "pre_gauss": 3, "morph_k": 3, "passes": 4}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.median_pipeline_u8(gray, 5, 3, 3, 4))
    st, js = post(client, gray, "median", k="x", gauss=-3, morph=5, passes=0)   # bad int -> default; negatives -> 0
    assert js["params"] == {"median_k": 3, "pre_gauss": 0, "morph_k": 5, "passes": 1}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.median_pipeline_u8(gray, 3, 0, 5, 1))
    st, js = post(client, gray, "median", k=99)
    assert js["params"]["median_k"] == 31
    st, js = post(client, gray, "denoise_manual")                    # default method is median, kernel 5
    assert st == 200 and js["message"] == "Manual denoise: Median (k=5)" and js["params"]["method"] == "Median (k=5)"
    assert js["params"]["kernel"] == 5 and set(js["params"]["histogram"]) == {"extreme_ratio", "variance", "cv", "suggested_method"}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.median_blur_u8(gray, 5))
    st, js = post(client, gray, "denoise_manual", method="median", kernel=2)    # even -> 3
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.median_blur_u8(gray, 3))


def test_batch_endpoint(client):
    """/api/process_batch: same dispatch per file, results in input order, per-file status."""
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, (20 + i, 30, 3), dtype=np.uint8) for i in range(5)]
    data = {"mode": "sharpening", "file": [(io.BytesIO(png(a)), f"{i}.png") for i, a in enumerate(imgs)]}
    r = client.post("/api/process_batch", data=data, content_type="multipart/form-data")
    js = r.get_json()
    assert r.status_code == 200 and js["count"] == 5
    for a, res in zip(imgs, js["results"]):
        assert res["status"] == 200 and res["message"] == "Sharpened: Laplacian"
        np.testing.assert_array_equal(codec.from_data_url_png(res["dataURL"]), O.laplacian_sharpen_u8(a[:, :, ::-1]))
    data = {"mode": "gamma", "file": [(io.BytesIO(png(imgs[0])), "a.png"), (io.BytesIO(b"junk"), "b.png")]}
    js = client.post("/api/process_batch", data=data, content_type="multipart/form-data").get_json()
    assert [x["status"] for x in js["results"]] == [400, 500]            # gamma missing -> 400 (server.py:365); the junk file
                                                                         # fails its decode first (server.py:360) -> 500
    data = {"mode": "negative", "file": [(io.BytesIO(png(imgs[0])), "a.png"), (io.BytesIO(b"junk"), "b.png")]}
    js = client.post("/api/process_batch", data=data, content_type="multipart/form-data").get_json()
    assert [x["status"] for x in js["results"]] == [200, 500] and "Cannot decode" in js["results"][1]["error"]
    assert client.post("/api/process_batch", data={"mode": "negative"}, content_type="multipart/form-data").status_code == 400


def test_extension_modes(client):
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (30, 44, 3), dtype=np.uint8)
    bgr = rgb[:, :, ::-1]
    st, js = post(client, rgb, "gaussian", k=4)                       # even -> next odd, as server.py:406-407 does
    assert st == 200 and js["params"] == {"ksize": 5}
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.gaussian_blur_u8(bgr, 5))
    st, js = post(client, rgb, "box")
    np.testing.assert_array_equal(codec.from_data_url_png(js["dataURL"]), O.box3_u8(bgr))


def test_codec_alpha_16bit_palette():
    rng = np.random.default_rng(2)
    rgba = rng.integers(0, 256, (9, 11, 4), dtype=np.uint8)
    a = codec.decode_unchanged(png(rgba))
    assert a.shape == (9, 11, 4)
    np.testing.assert_array_equal(a[..., :3], rgba[..., 2::-1])
    g16 = rng.integers(0, 65536, (9, 11), dtype=np.uint16)
    buf = io.BytesIO(); Image.fromarray(g16).save(buf, format="PNG")
    d = codec.decode_unchanged(buf.getvalue())
    assert d.dtype == np.uint16
    np.testing.assert_array_equal(d, g16)
    np.testing.assert_array_equal(codec.decode_color(buf.getvalue())[..., 0], (g16 >> 8).astype(np.uint8))
    np.testing.assert_array_equal(app_mod.ensure_gray_uint8(OracleBackend(), buf.getvalue()), (g16 >> 8).astype(np.uint8))
    np.testing.assert_array_equal(app_mod.ensure_gray_uint8(OracleBackend(), png(rgba)), O.bgr2gray_u8(rgba[..., [2, 1, 0, 3]]))
    bgr = rng.integers(0, 256, (5, 6, 3), dtype=np.uint8)
    np.testing.assert_array_equal(codec.from_data_url_png(codec.to_data_url_png(bgr)), bgr)


def test_decode_color_applies_exif_orientation_decode_unchanged_does_not():
    """cv2.IMREAD_COLOR (mode=sharpening, server.py:252) honours a JPEG's EXIF orientation; IMREAD_UNCHANGED (every
    gray-path mode, server.py:499) ignores it.  OpenCV documents both; parity with the real decoder is unpinned."""
    yy, xx = np.mgrid[0:24, 0:40]
    rgb = np.stack([(xx * 6) % 256, (yy * 10) % 256, (xx + yy) * 3 % 256], -1).astype(np.uint8)
    exif = Image.Exif()
    exif[0x0112] = 6                                                  # rotate 90 degrees clockwise to display
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", quality=95, exif=exif)
    plain = io.BytesIO()
    Image.fromarray(rgb).save(plain, format="JPEG", quality=95)
    base = codec.decode_unchanged(plain.getvalue(), "rgb")
    assert codec.decode_unchanged(buf.getvalue(), "rgb").shape == (24, 40, 3)           # tag ignored
    np.testing.assert_array_equal(codec.decode_unchanged(buf.getvalue(), "rgb"), base)
    col = codec.decode_color(buf.getvalue(), "rgb")
    assert col.shape == (40, 24, 3)                                                   # tag applied
    np.testing.assert_array_equal(col, np.rot90(base, k=-1))
    np.testing.assert_array_equal(codec.decode_color(plain.getvalue(), "rgb"), base)


def test_batch_sharder_order_and_round_robin():
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (10 + i, 12, 3), dtype=np.uint8) for i in range(7)]
    sh = BatchSharder(devices=["cpu", "cpu", "cpu"], ops=OracleOps)
    outs = sh.map(imgs, lambda ops, t: ops.laplacian_sharpen(t))
    for i, o in enumerate(outs):
        np.testing.assert_array_equal(o, O.laplacian_sharpen_u8(imgs[i]))
    outs = sh.map(imgs, lambda ops, t: ops.bgr2gray_equalize(t))
    for i, (g, e) in enumerate(outs):
        np.testing.assert_array_equal(e, O.equalize_hist_u8(O.bgr2gray_u8(imgs[i])))
    assert list(shard_range(7, 1, 3)) == [1, 4] and list(shard_range(2, 3, 4)) == []
    with pytest.raises(RuntimeError, match="failed"):
        sh.map(imgs, lambda ops, t: ops.nonexistent(t))


def test_png_writer_round_trips_and_is_one_valid_stream():
    """ipx.codec.encode_png (band-parallel deflate): any decoder must read back the same pixels -- Pillow here, and
    zlib on the concatenated IDAT stream (checks the joined bands and the Adler-32 trailer)."""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:700, 0:900]
    smooth = (128 + 90 * np.sin(xx / 37.0) * np.cos(yy / 53.0)).astype(np.uint8)
    cases = [rng.integers(0, 256, (1, 1), dtype=np.uint8), rng.integers(0, 256, (3, 5, 3), dtype=np.uint8),
             rng.integers(0, 256, (1200, 1100), dtype=np.uint8),                  # > 1 MiB: two bands, incompressible
             smooth, np.stack([smooth, smooth.T[:700, :700].repeat(2, 1)[:, :900], 255 - smooth], -1),
             np.zeros((2000, 1500), np.uint8)]                                      # many bands of all-zero rows
    for img in cases:
        data = codec.encode_png(img)
        back = np.array(Image.open(io.BytesIO(data)))
        np.testing.assert_array_equal(back, img)
        pos, idat = 8, b""
        while pos < len(data):
            n, tag = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
            if tag == b"IDAT":
                idat += body
            pos += 12 + n
        raw = zlib.decompress(idat)                                                 # raises on a bad stream / checksum
        assert len(raw) == img.shape[0] * (img[0].size + 1)
    bgr = rng.integers(0, 256, (40, 30, 3), dtype=np.uint8)
    np.testing.assert_array_equal(codec.from_data_url_png(codec.to_data_url_png(bgr)), bgr)
    g1 = rng.integers(0, 256, (40, 30, 1), dtype=np.uint8)
    np.testing.assert_array_equal(codec.from_data_url_png(codec.to_data_url_png(g1)), g1[:, :, 0])
