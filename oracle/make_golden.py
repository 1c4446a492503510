#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (`/root/reference/server.py`).

TEST INFRASTRUCTURE.  Runs only in the build container (the reference never
travels to the GPU box); the outputs it writes are data -- input images and the
reference's decoded output pixels / JSON -- and are committed.

How: `server.py:3` imports cv2, which is absent from
this image.  A tiny PIL-backed module providing ONLY `imdecode`, `imencode` and
the `IMREAD_*` constants (image *codec I/O*, no arithmetic) is registered as
`cv2`, `server` is imported from /root/reference without writing bytecode, and
requests are posted through Flask's test client.  Every branch exercised here
does its arithmetic in numpy inside the reference's own code; branches whose
arithmetic lives inside OpenCV (threshold, equalizeHist, Laplacian,
GaussianBlur, cvtColor) cannot run and are NOT captured -- for them parity is
unpinned (oracle/oracle.py header).

Usage:  python oracle/make_golden.py          (writes tests/golden/*)
"""
from __future__ import annotations

import base64
import io
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_codec_standin():
    from PIL import Image

    m = types.ModuleType("cv2")
    m.IMREAD_UNCHANGED, m.IMREAD_GRAYSCALE, m.IMREAD_COLOR, m.IMREAD_ANYDEPTH = -1, 0, 1, 2

    def imdecode(arr, flag):
        try:
            im = Image.open(io.BytesIO(np.asarray(arr, np.uint8).tobytes()))
            im.load()
        except Exception:
            return None
        if im.mode != "L":
            raise RuntimeError("golden generator only feeds 8-bit gray PNGs")
        a = np.array(im, dtype=np.uint8)
        if flag == m.IMREAD_COLOR:
            return None            # keep every request on the gray path
        return a

    def imencode(ext, img):
        assert ext == ".png"
        buf = io.BytesIO()
        Image.fromarray(np.asarray(img)).save(buf, format="PNG")
        return True, np.frombuffer(buf.getvalue(), np.uint8)

    m.imdecode, m.imencode = imdecode, imencode
    sys.modules["cv2"] = m


def _png(a: np.ndarray) -> bytes:
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    return buf.getvalue()


def _decode_data_url(url: str) -> np.ndarray:
    from PIL import Image
    assert url.startswith("data:image/png;base64,")
    raw = base64.b64decode(url.split(",", 1)[1])
    return np.array(Image.open(io.BytesIO(raw)))


def make_inputs():
    rng = np.random.default_rng(0)
    imgs = {
        "rand": rng.integers(0, 256, (48, 80), dtype=np.uint8),
        "ramp": np.arange(256, dtype=np.uint8).reshape(16, 16),
        "lowmax": rng.integers(3, 201, (40, 56), dtype=np.uint8),     # max < 255 (log)
        "const": np.full((9, 13), 77, np.uint8),
        "narrow": np.clip(rng.normal(128, 10, (32, 48)), 0, 255).astype(np.uint8),
        "one": np.array([[200]], np.uint8),
    }
    return imgs


def main():
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    _install_codec_standin()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import server  # the real reference module
    finally:
        os.chdir(cwd)
    client = server.app.test_client()
    imgs = make_inputs()

    def post(img, mode, **form):
        data = {"mode": mode, **{k: str(v) for k, v in form.items()}}
        if img is not None:                               # a 1-D array is sent as raw file bytes (not an image)
            data["file"] = (io.BytesIO(img.tobytes() if img.ndim == 1 else _png(img)), "x.png")
        if mode is None:
            data.pop("mode")
        r = client.post("/api/process", data=data, content_type="multipart/form-data")
        return r.status_code, r.get_json()

    arrays = {f"in_{k}": v for k, v in imgs.items()}
    contract = []

    def case(name, img_key, mode, **form):
        st, js = post(imgs[img_key] if img_key else None, mode, **form)
        rec = {"name": name, "input": img_key, "mode": mode, "form": {k: str(v) for k, v in form.items()},
               "status": st}
        if st == 200:
            out = _decode_data_url(js["dataURL"])
            arrays[f"out_{name}"] = out
            rec["message"] = js["message"]
            rec["params"] = js["params"]
        else:
            rec["error"] = js.get("error")
        contract.append(rec)

    for k in imgs:
        case(f"gray8_{k}", k, "gray8")
        case(f"negative_{k}", k, "negative")
    case("default_mode_rand", "rand", None)
    case("unknown_mode_rand", "rand", "detect")          # app.js:215 sends this; falls to gray
    for g in (0.1, 0.4, 0.6, 1.0, 2.2, 5.0, 10.0):
        for k in ("rand", "ramp"):
            case(f"gamma_{g}_{k}", k, "gamma", gamma=g)
    case("gamma_alias_powerlaw", "rand", "powerlaw", gamma_value=0.6)
    case("gamma_alias_g", "rand", "power-law", g=1.7)
    case("gamma_missing", "rand", "gamma")
    for c in (2.0, 0.5, 10.0, 100.0):
        for k in ("rand", "ramp", "lowmax", "const", "narrow", "one"):
            case(f"log_{c}_{k}", k, "log", c=c)
    case("log_default_c", "lowmax", "logarithmic")
    case("log_alias", "lowmax", "logrithmic", c=3)
    for (r1, s1, r2, s2) in ((70, 10, 180, 245), (1, 0, 255, 255), (0, 300, 0, -5), (100, 50, 254, 200),
                             (254, 255, 255, 0), (128, 128, 129, 128), (30, 200, 31, 20)):
        for k in ("rand", "ramp"):
            case(f"piecewise_{r1}_{s1}_{r2}_{s2}_{k}", k, "piecewise", r1=r1, s1=s1, r2=r2, s2=s2)
    case("piecewise_alias_stretch", "rand", "stretch", r1=70, s1=10, r2=180, s2=245)
    case("piecewise_missing", "rand", "piecewise", r1=70, s1=10, r2=180)
    case("pca_removed", "rand", "pca")
    case("missing_file", None, "gray8")

    # lee_filter called directly (server.py:711-732) -- per-pixel python loop, small inputs
    rng = np.random.default_rng(7)
    base = np.clip(rng.normal(120, 30, (28, 36)), 0, 255)
    speck = np.clip(base * rng.gamma(16.0, 1 / 16.0, base.shape), 0, 255).astype(np.uint8)
    speck[5:9, 5:12] = 50                       # a flat patch -> var == 0 branch
    arrays["in_lee"] = speck
    for wnd in (3, 5, 7):
        arrays[f"out_lee_{wnd}"] = server.lee_filter(speck, window_size=wnd)

    for wnd in (9, 15, 31):                     # large windows: numpy's pairwise sum recurses above 128 values
        arrays[f"out_lee_rand_{wnd}"] = server.lee_filter(imgs["rand"], window_size=wnd)

    # 3 x 3 windows whose result changes by one if the variance sum or the final mean + k * (x - mean) is evaluated
    # with a fused multiply-add (found by random search): the reference rounds after every operation
    patches = [[173, 164, 234, 228, 51, 90, 114, 55, 200], [173, 117, 174, 25, 49, 172, 219, 250, 227],
               [107, 151, 63, 80, 21, 81, 193, 46, 245], [150, 192, 9, 185, 222, 107, 11, 201, 167],
               [163, 81, 254, 11, 236, 136, 164, 72, 18], [245, 7, 10, 77, 199, 14, 197, 186, 224],
               [74, 139, 114, 128, 4, 62, 186, 43, 168], [60, 195, 22, 25, 4, 161, 123, 140, 170],
               [231, 116, 145, 140, 43, 218, 48, 87, 49], [253, 55, 90, 42, 208, 160, 182, 98, 66],
               [182, 97, 153, 177, 28, 143, 58, 224, 162], [107, 204, 129, 244, 17, 226, 67, 90, 23]]
    arrays["in_lee_rounding"] = np.concatenate([np.array(p_, np.uint8).reshape(3, 3) for p_ in patches], axis=1)
    arrays["out_lee_rounding_3"] = server.lee_filter(arrays["in_lee_rounding"], window_size=3)

    # mode=denoise_manual with method=lee through the real dispatch (server.py:181-241): histogram statistics
    # (extreme_ratio, variance, cv, suggested_method) and the Lee output, all numpy inside the reference
    sp = imgs["rand"].copy()
    m = np.random.default_rng(11).random(sp.shape)
    sp = np.where(m < 0.03, 0, np.where(m > 0.97, 255, (sp // 4 + 90))).astype(np.uint8)   # salt & pepper
    imgs["saltpepper"] = sp
    imgs["lee"] = speck
    arrays["in_saltpepper"] = sp
    for key, kern in (("rand", 5), ("lee", 4), ("narrow", 7), ("const", 3), ("saltpepper", 5), ("lowmax", 40), ("ramp", 9)):
        case(f"denoise_lee_{key}_{kern}", key, "denoise_manual", method="lee", kernel=kern)

    # an upload that is not an image: cv2.imdecode returns None for every flag, ensure_gray_uint8_from_bytes raises
    # "unknown error" (server.py:531) and the blanket handler answers 500 -- BEFORE any parameter check of the gray
    # branch (server.py:360 decodes first), also for histogram_eq / sharpening / denoise_manual / median
    imgs["garbage"] = np.frombuffer(b"this is not an image file, just 50 bytes of plain text..", np.uint8).copy()
    arrays["in_garbage"] = imgs["garbage"]
    case("garbage_gray8", "garbage", "gray8")
    case("garbage_gamma_missing_param", "garbage", "gamma")
    case("garbage_gamma", "garbage", "gamma", gamma=0.5)
    case("garbage_piecewise_missing_params", "garbage", "piecewise", r1=1)
    case("garbage_histogram_eq", "garbage", "histogram_eq")
    case("garbage_sharpening", "garbage", "sharpening")
    case("garbage_denoise_manual", "garbage", "denoise_manual", method="lee")
    case("garbage_median_nlm", "garbage", "median", nlm=1)
    case("garbage_pca", "garbage", "pca")                     # decided before any decode: still 400

    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "server_numpy_ops.npz"), **arrays)
    meta = {
        "generator": "oracle/make_golden.py",
        "reference": "duyavaliable/image-processing server.py (real module, codec I/O via PIL stand-in)",
        "numpy": np.__version__,
        "cpu_features": sorted(k for k, v in np._core._multiarray_umath.__cpu_features__.items() if v),
        "cases": contract,
    }
    with open(os.path.join(OUT, "server_contract.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(f"wrote {len(arrays)} arrays, {len(contract)} contract cases to {os.path.normpath(OUT)}")


if __name__ == "__main__":
    main()
