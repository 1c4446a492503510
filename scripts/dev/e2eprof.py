"""Dev: cProfile of one 4K colour request through the Flask host (GPU box)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np
from PIL import Image
import ipx.app as app_mod

mode = sys.argv[1] if len(sys.argv) > 1 else "sharpening"
c = 3 if mode in ("sharpening", "histogram_eq") else 1
client = app_mod.create_app().test_client()
yy, xx = np.mgrid[0:2160, 0:3840]
img = np.clip(np.stack([128 + 60 * np.sin((xx + 50 * k) / 97.0) * np.cos(yy / 131.0) for k in range(c)], -1)
              + np.random.default_rng(0).normal(0, 4, (2160, 3840, c)), 0, 255).astype(np.uint8)
buf = io.BytesIO()
Image.fromarray(img[:, :, 0] if c == 1 else img).save(buf, format="PNG", compress_level=1)
data = buf.getvalue()

def once():
    r = client.post("/api/process", data={"mode": mode, "file": (io.BytesIO(data), "x.png"), "gamma": "0.6"}, content_type="multipart/form-data")
    assert r.status_code == 200
    return r

once()
pr = cProfile.Profile()
pr.enable()
r = once()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
