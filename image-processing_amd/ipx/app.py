"""Flask host that keeps server.py's request/response contract and runs the pixel arithmetic on the GPU.

Contract (server.py:19-29, 243-283, 358-388, 443-476; client side src/app.js:855-897, 1089-1140):
  POST /api/process   multipart: file (required), mode (default 'gray8'), extras as strings
  200 {'dataURL': 'data:image/png;base64,...', 'message': str, 'params': dict}
  400 {'error': ...} for parameter errors, 500 {'error': str(e)} for any exception.
Mode names, aliases, defaults, clamps, messages and params are the reference's.  Modes whose arithmetic is
outside this build's scope (C10-C13) answer 501 with an explanatory error instead of
silently running on the CPU; three extension modes ('gaussian', 'box', 'sobel') expose the remaining kernels.

The arithmetic goes through a backend object with the ipx.ops surface.  The default (and only shipped) backend
is the HIP library; tests inject a checker backend to exercise this dispatch logic on machines without a GPU.
"""
from __future__ import annotations

import json
import threading
import uuid

import numpy as np
from flask import Flask, jsonify, request

from . import codec, luts

OUT_OF_SCOPE = ("pca_nn", "count_objects", "frequency_filter", "wiener")
DENOISE_OUT_OF_SCOPE = ("bilateral", "nlm")              # cv2 edge-preserving / patch filters


class GpuBackend:
    """numpy in / numpy out over ipx.ops on one GPU.  Each request runs on a per-thread HIP stream (Flask's
    dev server handles requests on threads); libipx itself keeps no shared state."""

    def __init__(self, device: str = "cuda:0"):
        import torch
        from . import _lib, ops
        _lib.load()
        self.torch, self.ops = torch, ops
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            _lib.call("ipx_check_device")
        self._tls = threading.local()

    def _stream(self):
        s = getattr(self._tls, "stream", None)
        if s is None:
            s = self._tls.stream = self.torch.cuda.Stream(device=self.device)
        return s

    def run(self, fn, *arrays):
        """Stage arrays to the device, run fn(ops, *tensors) on this thread's stream, bring results back."""
        torch = self.torch
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream()):
            self._tls.png_status = []
            ts = [self._upload(a, i) for i, a in enumerate(arrays)]
            out = fn(self.ops, *ts)
            outs = out if isinstance(out, tuple) else (out,)
            host = [self._download(o) if isinstance(o, torch.Tensor) else o for o in outs]
            if getattr(self._tls, "upload_pending", False):            # the pinned upload buffers are reused by the next call
                torch.cuda.current_stream().synchronize()
                self._tls.upload_pending = False
            bad = any(int(st.item()) != 0 for st in self._tls.png_status)   # (the stream is idle by now)
            self._tls.png_status = []
        if bad:                                                        # never seen: the GPU reconstruction gave up; Pillow's pixels
            return self.run(fn, *[a.array() if isinstance(a, (codec.LazyPng, codec.LazyPng16)) else a for a in arrays])
        return host if isinstance(out, tuple) else host[0]

    def _upload(self, a, slot: int):
        """numpy -> device tensor through this thread's persistent pinned buffer (one per argument slot): pinning a
        fresh buffer per request (`.pin_memory()`) costs a page-locking system call per upload.  The copy into the
        buffer is a host memcpy; the H2D copy is asynchronous on this thread's stream and ordered before the next use of
        the buffer by the synchronisation every request ends with (`_download`)."""
        torch = self.torch
        lazy = a if isinstance(a, (codec.LazyPng, codec.LazyPng16)) else None
        if lazy is not None:                                           # inflated PNG scan lines: the GPU un-filters them
            a = np.frombuffer(lazy.raw, np.uint8)
        else:
            a = np.ascontiguousarray(a)
        if a.size == 0 or a.dtype != np.uint8:
            return torch.from_numpy(a).to(self.device)
        bufs = getattr(self._tls, "pinned_up", None)
        if bufs is None:
            bufs = self._tls.pinned_up = {}
        buf = bufs.get(slot)
        if buf is None or buf.numel() < a.size:
            buf = bufs[slot] = torch.empty(max(a.size, 1 << 20), dtype=torch.uint8).pin_memory()
        view = buf[:a.size].view(a.shape)
        view.numpy()[...] = a
        t = view.to(self.device, non_blocking=True)
        self._tls.upload_pending = True
        if lazy is not None:
            t, status = self.ops.png_unfilter(t, lazy.h, lazy.w, lazy.bytes_per_pixel)
            self._tls.png_status.append(status)
            t = t.view(lazy.shape)
        return t

    def b64encode(self, data: bytes) -> str:
        """base64.b64encode(data).decode("ascii") (server.py:536-537) on the GPU: binascii holds the GIL for 40 ms per 4K colour
        PNG; this is two pinned copies, one kernel and one pass that builds the str.  Small inputs stay on the host."""
        n = len(data)
        if n < (1 << 18):
            import base64
            return base64.b64encode(data).decode("ascii")
        torch = self.torch
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream()):
            bufs = getattr(self._tls, "pinned_b64", None)
            m = 4 * ((n + 2) // 3)
            if bufs is None or bufs[0].numel() < n:
                bufs = self._tls.pinned_b64 = (torch.empty(max(n, 1 << 22), dtype=torch.uint8).pin_memory(),
                                               torch.empty(4 * ((max(n, 1 << 22) + 2) // 3), dtype=torch.uint8).pin_memory())
            src, dst = bufs
            src[:n].numpy()[...] = np.frombuffer(data, np.uint8)
            out = self.ops.base64_encode(src[:n].to(self.device, non_blocking=True))
            dst[:m].copy_(out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return str(memoryview(dst[:m].numpy()), "ascii")

    def _download(self, t):
        """Device tensor -> numpy through this thread's persistent pinned buffer: a pageable D2H copy of a 4K colour
        frame takes 5.9 ms, the pinned copy plus the copy out of the buffer 1.2 ms (scripts/dev/stagingprobe.py)."""
        torch = self.torch
        t = t.contiguous()
        n = t.numel() * t.element_size()
        if n == 0 or t.dtype != torch.uint8:
            return t.to("cpu").numpy()
        buf = getattr(self._tls, "pinned", None)
        if buf is None or buf.numel() < n:
            buf = self._tls.pinned = torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory()
        view = buf[:n].view(t.shape)
        view.copy_(t, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return np.array(view.numpy())                                  # the buffer is reused by the next call


def data_url(backend, img, order: str = "bgr") -> str:
    """codec.to_data_url_png with the backend's base64 (GPU) when it has one; test backends fall back to the host's."""
    return codec.to_data_url_png(img, order, b64=getattr(backend, "b64encode", None))


def to_bgr(t):
    """RGB(A) tensor as decoded -> BGR(A), on the device the tensor lives on (the host copy this replaces costs more
    than the whole GPU side of a request)."""
    import torch
    if t.shape[-1] == 4:
        return torch.cat([t[..., :3].flip(-1), t[..., 3:]], dim=-1)
    return t.flip(-1)


def ensure_gray_uint8(backend, img_bytes: bytes) -> np.ndarray:
    """server.py:492-531: decode, drop alpha, BGR->gray (GPU), 16-bit >> 8."""
    return gray_from_decoded(backend, codec.decode_lazy(img_bytes, allow16=True))


def gray_from_decoded(backend, img):
    """The part of ensure_gray_uint8 after the decode (`img` in RGB(A) order as decode_unchanged(..., "rgb") or
    decode_lazy returns it; a gray LazyPng passes through and is reconstructed when it is uploaded; a 16-bit PNG is
    gray-converted on its 16-bit samples and shifted afterwards, in the reference's order, server.py:516-523)."""
    if isinstance(img, codec.LazyPng16):
        w, ch = img.w, img.channels
        return backend.run(lambda ops, t: ops.gray_from_u16(t, w, ch, big_endian=True, rgb_order=True), img)
    if img.ndim == 3:
        img = backend.run(lambda ops, t: ops.bgr2gray(to_bgr(t)), img)  # 3 or 4 channels; alpha dropped
    if img.dtype == np.uint16:
        img = (img >> 8).astype(np.uint8)                                # server.py:521-523
    if img.ndim != 2 or img.dtype != np.uint8:
        raise RuntimeError("Cannot decode image to grayscale 8-bit. unknown error")
    return img


def create_app(backend=None) -> Flask:
    app = Flask(__name__)
    state = {"backend": backend}

    def be():
        if state["backend"] is None:
            state["backend"] = GpuBackend()
        return state["backend"]

    def batch_backends():
        """One backend per visible GPU for /api/process_batch (an injected test backend is used as is)."""
        if state.get("pool") is None:
            if backend is not None:
                state["pool"] = [backend]
            else:
                import torch
                n = max(1, torch.cuda.device_count())
                state["pool"] = [be()] + [GpuBackend(f"cuda:{i}") for i in range(1, n)]
        return state["pool"]

    def process_one(b, data: bytes, form):
        """One image through the reference's dispatch (server.py:22-472): returns (payload dict, HTTP status)."""
        mode = form.get("mode", "gray8")
        params = {}

        if mode in OUT_OF_SCOPE:
            return {"error": f"mode '{mode}' is outside the GPU hot path of this backend "
                                     "(convolution filters, histogram equalization, pixel-wise transforms)"}, 501

        if mode == "denoise_manual":                                # server.py:181-241; method=lee runs on the GPU
            img = ensure_gray_uint8(b, data)
            method_name = form.get("method", "median")
            kernel = int(form.get("kernel", "5"))
            if kernel % 2 == 0:
                kernel += 1
            kernel = max(3, min(31, kernel))
            if method_name not in ("lee", "median"):                 # bilateral / NLM / the default branch: cv2 filters (C10)
                return {"error": f"denoise_manual method '{method_name}' is outside the GPU hot path of this "
                                         "backend (methods lee and median are built: SURVEY.md section 8(f))"}, 501
            # histogram analysis (server.py:190-211) and the filter, both on the GPU
            if method_name == "lee":
                st, img_denoised = b.run(lambda ops, t: (ops.denoise_stats(t, method_name), ops.lee_filter(t, kernel)), img)
                method = f"Lee Filter (w={kernel})"
            else:
                st, img_denoised = b.run(lambda ops, t: (ops.denoise_stats(t, method_name), ops.median_blur(t, kernel)), img)
                method = f"Median (k={kernel})"
            params = {"method": method, "kernel": kernel,
                      "histogram": {"extreme_ratio": round(st["extreme_ratio"], 4), "variance": round(st["variance"], 2),
                                    "cv": round(st["cv"], 4), "suggested_method": st["suggested_method"]}}
            return {"dataURL": data_url(b, img_denoised), "message": f"Manual denoise: {method}",
                            "params": params}, 200

        if mode == "histogram_eq":                                  # server.py:243-248
            img = codec.decode_lazy(data, allow16=True)
            if isinstance(img, codec.LazyPng16):                     # 16-bit PNG: gray on 16 bits, >> 8, then equalize
                w16, ch16 = img.w, img.channels
                img_eq = b.run(lambda ops, t: ops.equalize_hist(ops.gray_from_u16(t, w16, ch16)), img)
            elif img.ndim == 3:                                      # colour: gray + histogram in one pass
                img_eq = b.run(lambda ops, t: ops.bgr2gray_equalize(to_bgr(t))[1], img)
            else:
                if img.dtype == np.uint16:
                    img = (img >> 8).astype(np.uint8)
                img_eq = b.run(lambda ops, t: ops.equalize_hist(t), img)
            params = {"method": "Histogram Equalization"}
            return {"dataURL": data_url(b, img_eq), "message": "Histogram Equalized", "params": params}, 200

        if mode == "sharpening":                                    # server.py:250-283
            img = codec.decode_color_lazy(data)                       # a per-channel filter: the channel order is immaterial
            if img is None:
                img = ensure_gray_uint8(b, data)
                is_color = False
            else:
                is_color = True
            sharp_type = form.get("sharp_type", "laplacian")
            if sharp_type == "laplacian":
                img_sharp = b.run(lambda ops, t: ops.laplacian_sharpen(t), img)
                method = "Laplacian"
            else:
                img_sharp = img
                method = "No sharpening"
            params = {"method": method, "color": is_color}
            return {"dataURL": data_url(b, img_sharp, "rgb"), "message": f"Sharpened: {method}", "params": params}, 200

        if mode == "median":                                        # server.py:390-441 (NLM step: out of scope)
            def _int(name, dflt):
                try:
                    return int(form.get(name, dflt))
                except Exception:
                    return int(dflt)
            k = _int("k", "3")
            if k % 2 == 0:
                k += 1
            k = max(3, min(31, k))
            g = _int("gauss", "0")
            if g % 2 == 0 and g > 0:
                g += 1
            if g < 1:
                g = 0
            morph = _int("morph", "0")
            if morph % 2 == 0 and morph > 0:
                morph += 1
            if morph < 1:
                morph = 0
            passes = max(1, min(4, _int("passes", "1")))
            decoded = codec.decode_lazy(data, allow16=True)   # the reference decodes first (server.py:360): a bad
                                                                     # upload is a 500 whatever the parameters say
            if form.get("nlm", "0") == "1":
                return {"error": "mode 'median' with nlm=1 (cv2.fastNlMeansDenoising) is outside the GPU hot path of this backend"}, 501
            if morph > 31 or g > 255:
                return {"error": "mode 'median': morph > 31 or gauss > 255 is not supported by this backend"}, 501
            img = gray_from_decoded(b, decoded)

            def pipeline(ops, t):
                if g > 0:
                    t = ops.gaussian_blur(t, g)
                for _ in range(passes):
                    t = ops.median_blur(t, k)
                if morph > 0:
                    se = ops.ellipse_se(morph)
                    t = ops.morph_close(ops.morph_open(t, se), se)
                return t
            out = b.run(pipeline, img)
            params = {"median_k": int(k), "pre_gauss": int(g), "morph_k": int(morph), "passes": int(passes)}
            return {"dataURL": data_url(b, out), "message": "ok", "params": params}, 200

        if mode == "pca":                                           # server.py:353-356
            return {"error": "mode 'pca' has been removed on server. Use 'pca_nn' for PCA+NN inference "
                                     "or 'detect' for object detection."}, 400

        if mode in ("gaussian", "box", "sobel"):                    # extension modes (not in the reference)
            img = codec.decode_lazy(data)                 # per-channel filters: no channel reversal needed
            if img.dtype == np.uint16:
                img = (img >> 8).astype(np.uint8)
            if img.ndim == 3 and img.shape[2] == 4:
                img = np.ascontiguousarray(img[:, :, :3])
            if mode == "gaussian":
                k = int(form.get("k", "5"))
                if k % 2 == 0:
                    k += 1
                k = max(1, min(255, k))
                out = b.run(lambda ops, t: ops.gaussian_blur(t, k), img)
                params = {"ksize": k}
            elif mode == "box":
                out = b.run(lambda ops, t: ops.box3(t), img)
            else:
                out = b.run(lambda ops, t: ops.sobel(t), img)
            return {"dataURL": data_url(b, out, "rgb"), "message": "ok", "params": params}, 200

        # every other mode works on grayscale (server.py:358-360); unknown modes return plain gray.  The decode comes
        # first, as in the reference: an undecodable upload answers 500 even when a parameter is missing too
        img = codec.decode_lazy(data, allow16=True)
        if isinstance(img, codec.LazyPng16):                        # (no fused gray + table form for 16-bit samples)
            img = gray_from_decoded(b, img)
        lut = None
        if mode in ("gamma", "powerlaw", "power-law"):              # server.py:362-373
            gamma_param = form.get("gamma") or form.get("gamma_value") or form.get("g")
            if gamma_param is None:
                return {"error": "Missing required parameter: gamma"}, 400
            gamma_val = float(gamma_param)
            lut = luts.gamma(gamma_val)
            params = {"gamma": float(gamma_val)}
        elif mode == "negative":                                    # server.py:375-376
            lut = luts.negative()
        elif mode == "threshold":                                   # server.py:378-381
            T = int(form.get("threshold") or form.get("T") or 128)
            lut = luts.threshold(T)
            params["threshold"] = T
        elif mode in ("piecewise", "piecewise_linear", "stretch"):  # server.py:443-469
            if not all(k in form for k in ["r1", "s1", "r2", "s2"]):
                return {"error": "Missing required parameters: r1, s1, r2, s2"}, 400
            r1, s1, r2, s2 = luts.piecewise_clamp(int(form.get("r1")), int(form.get("s1")),
                                                  int(form.get("r2")), int(form.get("s2")))
            lut = luts.piecewise(r1, s1, r2, s2)
            params = {"r1": r1, "s1": s1, "r2": r2, "s2": s2}

        is_log = mode in ("log", "logarithmic", "logrithmic")       # server.py:383-388
        if is_log:
            c_val = float(form.get("c", "2.0"))
            gray = gray_from_decoded(b, img)                          # (the reference decodes a second time here)
            hist = b.run(lambda ops, t: ops.hist(t), gray)           # s.max() needs the levels present
            out = b.run(lambda ops, t: ops.apply_lut(t, luts.log(c_val, hist)), gray)
        elif lut is not None and img.ndim == 3:
            out = b.run(lambda ops, t: ops.bgr2gray_lut(to_bgr(t), lut), img)   # gray + table in one pass
        else:
            gray = gray_from_decoded(b, img)
            out = gray if lut is None else b.run(lambda ops, t: ops.apply_lut(t, lut), gray)
        return {"dataURL": data_url(b, out), "message": "ok", "params": params}, 200

    def respond(payload, status):
        """jsonify, except that the data URLs (tens of MB of base64 that needs no escaping) are spliced in instead of
        being scanned by the JSON encoder (46 ms of a 4K colour request)."""
        urls = []
        token = "@" + uuid.uuid4().hex + "@"                           # cannot occur in anything a client sent

        def strip(o):
            if isinstance(o, dict):
                return {k: strip(v) for k, v in o.items()}
            if isinstance(o, list):
                return [strip(v) for v in o]
            if isinstance(o, str) and o.startswith("data:image/png;base64,") and len(o) > 4096:
                urls.append(o)
                return f"{token}{len(urls) - 1}{token}"
            return o
        body = json.dumps(strip(payload), sort_keys=True, separators=(",", ":"))
        if urls:
            parts = body.split(token)
            for i in range(1, len(parts), 2):
                parts[i] = urls[int(parts[i])]
            body = "".join(parts)
        return app.response_class(body + "\n", status=status, mimetype="application/json")

    def guarded(b, data, form):
        try:
            return process_one(b, data, form)
        except Exception as e:                                          # server.py:473-476
            import traceback
            traceback.print_exc()
            return {"error": str(e)}, 500

    @app.route("/api/process", methods=["POST"])
    def api_process():
        f = request.files.get("file")
        if not f:
            return jsonify({"error": "missing file"}), 400
        payload, status = guarded(be(), f.read(), request.form)
        return respond(payload, status)

    @app.route("/api/process_batch", methods=["POST"])
    def api_process_batch():
        """the same dispatch for MANY files in one request (the browser path is capped
        at one 8 MB upload, src/app.js:548-552).  multipart: file=<f0>, file=<f1>, ... plus the form fields of
        /api/process (applied to every file).  Image i runs on backend i mod G (one backend per visible GPU, one host
        thread per backend); results come back in input order:
        200 {'results': [{'status': int, 'dataURL'|'error': ..., 'message': ..., 'params': ...}, ...], 'count': n}."""
        files = request.files.getlist("file")
        if not files:
            return jsonify({"error": "missing file"}), 400
        blobs = [f.read() for f in files]
        form = request.form
        bes = batch_backends()
        results = [None] * len(blobs)
        # Pillow's decode (35-125 ms per 4K file, GIL released inside the inflate / unfilter loops) runs for ALL files on
        # the codec pool, ahead of the per-GPU workers, instead of one file after another inside each worker
        decoded = codec.DecodeAhead(blobs)

        def work(k):
            for i in range(k, len(blobs), len(bes)):
                with decoded.use(i):
                    payload, status = guarded(bes[k], blobs[i], form)
                results[i] = dict(payload, status=status)

        ths = [threading.Thread(target=work, args=(k,)) for k in range(min(len(bes), len(blobs)))]
        [t.start() for t in ths]
        [t.join() for t in ths]
        return respond({"results": results, "count": len(results)}, 200)

    return app


def main():
    create_app().run(host="0.0.0.0", port=5500, debug=False, threaded=True)


if __name__ == "__main__":
    main()
