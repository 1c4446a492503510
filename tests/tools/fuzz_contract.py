"""Dev tool (THIS container only: imports the real reference through oracle/make_golden.py's codec stand-in).
Random requests -- modes, parameter aliases, malformed / missing / out-of-range values -- go to the REAL server.py and to
ipx.app (oracle backend); status, JSON body and decoded pixels must be identical.  Only modes whose arithmetic is numpy
inside the reference can be compared (gray8, negative, gamma, log, piecewise/stretch, denoise_manual method=lee, unknown
modes, the 400 paths); OpenCV-backed modes are skipped.
usage: python tests/tools/fuzz_contract.py [--n 400] [--seed 0]"""
import argparse
import io
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")  # repo root (tests/tools/ -> ../..)
for p in ("image-processing_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np

import make_golden as MG

sys.dont_write_bytecode = True
MG._install_codec_standin()
sys.path.insert(0, MG.REF)
cwd = os.getcwd()
os.chdir(MG.REF)
try:
    import server
finally:
    os.chdir(cwd)
import ipx.app as app_mod
import ipx.codec as codec
from oracle_ops import OracleBackend

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
ref = server.app.test_client()
mine = app_mod.create_app(backend=OracleBackend()).test_client()

NUMS = ["0", "1", "2", "0.5", "0.6", "2.2", "-1", "255", "256", "128", "70", "10", "180", "245", "1e3", "3.7", "", " 5", "abc",
        "nan", "inf", "-0.0", "1e-3", "300", "-20", "17", "31", "40", "4", "9"]


def val():
    return str(rng.choice(NUMS))


def request():
    mode = str(rng.choice(["gray8", "negative", "gamma", "log", "piecewise", "stretch", "denoise_manual", "bogus", "pca", None]))
    if mode == "None":
        mode = None
    form = {}
    if mode == "gamma":
        for k in ("gamma", "gamma_value", "g"):
            if rng.random() < 0.4:
                form[k] = val()
    elif mode == "log":
        if rng.random() < 0.7:
            form["c"] = val()
    elif mode in ("piecewise", "stretch"):
        for k in ("r1", "s1", "r2", "s2"):
            if rng.random() < 0.9:
                form[k] = val()
    elif mode == "denoise_manual":
        form["method"] = "lee"
        if rng.random() < 0.8:
            form["kernel"] = val()
    kind = rng.integers(0, 4)
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 50))
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1:
        img = rng.integers(0, int(rng.integers(1, 256)), (h, w), dtype=np.uint8)
    elif kind == 2:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    else:
        img = np.clip(rng.normal(128, 20, (h, w)), 0, 255).astype(np.uint8)
    if rng.random() < 0.03:
        img = None
    return mode, form, img


def post(client, mode, form, img):
    data = dict(form)
    if mode is not None:
        data["mode"] = mode
    if img is not None:
        data["file"] = (io.BytesIO(MG._png(img)), "x.png")
    r = client.post("/api/process", data=data, content_type="multipart/form-data")
    return r.status_code, r.get_json()


bad = 0
for i in range(a.n):
    mode, form, img = request()
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rs, rj = post(ref, mode, form, img)
            ms, mj = post(mine, mode, form, img)
    same = rs == ms
    if same and rs == 200:
        same = rj["message"] == mj["message"] and rj["params"] == mj["params"] and \
            np.array_equal(MG._decode_data_url(rj["dataURL"]), codec.from_data_url_png(mj["dataURL"]))
    elif same:
        same = rj == mj
    if not same:
        bad += 1
        def brief(j):
            return {k: (v if k != "dataURL" else "...") for k, v in (j or {}).items()}
        print(f"DIFF mode={mode} form={form} img={None if img is None else img.shape}: reference {rs} {brief(rj)} | ipx {ms} {brief(mj)}", flush=True)
print(f"{a.n} requests, {bad} differences")
