"""BASELINE.json configs[0] ... configs[4] at their REAL shapes on one MI355X, through the C ABI, against the
C / numpy oracle (exact equality; all of it is integer / table arithmetic).

configs[0]: one 512 x 512 RGB image: grayscale + 3x3 box blur + histogram equalization (the reference's CPU-runnable case)
configs[1]: one 3840 x 2160 RGB image: 5x5 Gaussian + Sobel + hist-eq (the bench workload, here as ONE image per launch)

configs[2]: batch of 1080 x 1920 RGB frames, full filter pipeline (server.py:360-388, 443-467, 245-246, 263-277, 428)
configs[3]: one 4320 x 7680 RGB frame, separable Gaussian k = 15 and 31 (server.py:428 arithmetic per channel)
configs[4]: one 32768 x 32768 gray image, 8 row partitions of 4096 rows: hist-eq (server.py:246) + 3x3 convs
Parity vs real OpenCV is unpinned (cv2 absent); the numpy branches are pinned by tests/golden.
"""
import numpy as np
import pytest
import torch

import coracle as CO
import oracle as O
import ipx.luts as luts
import ipx.ops as ops
import ipx.pipelines as P
from ipx.sharder import BatchSharder

pytestmark = pytest.mark.gpu


def _frames(n, h, w, seed):
    """n synthetic BGR frames with different brightness ranges (so that every image has its own log / equalize table)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        hi = 256 if i % 3 == 0 else 120 + 15 * (i % 8)            # the log table depends on the image maximum
        out[i] = rng.integers(i % 5, hi, (h, w, 3), dtype=np.uint8)
    return out


def _oracle_full_pipeline(bgr):
    """The same nine outputs for ONE frame from the oracle (C restatement where it exists, numpy otherwise)."""
    gray = CO.bgr2gray_u8(bgr)
    return {"gray": gray, "histogram_eq": CO.equalize_hist_u8(gray),
            "negative": O.negative_u8(gray), "threshold": O.threshold_u8(gray, 128), "gamma": O.gamma_u8(gray, 0.6),
            "log": O.log_u8(gray, 2.0), "piecewise": O.piecewise_u8(gray, 70, 10, 180, 245),
            "sharpening": CO.laplacian_sharpen_u8(bgr), "gaussian5": CO.gaussian_blur_u8(bgr, 5)}


def test_configs0_512_rgb_gray_box_equalize(dev):
    bgr = _frames(1, 512, 512, 10)[0]
    t = torch.from_numpy(bgr).to(dev)
    gray = ops.bgr2gray(t)
    want_gray = CO.bgr2gray_u8(bgr)
    np.testing.assert_array_equal(gray.cpu().numpy(), want_gray)
    np.testing.assert_array_equal(ops.box3(gray).cpu().numpy(), CO.box3_u8(want_gray))
    np.testing.assert_array_equal(ops.box3(t).cpu().numpy(), CO.box3_u8(bgr))
    np.testing.assert_array_equal(ops.equalize_hist(gray).cpu().numpy(), CO.equalize_hist_u8(want_gray))
    g2, eq = ops.bgr2gray_equalize(t)
    np.testing.assert_array_equal(g2.cpu().numpy(), want_gray)
    np.testing.assert_array_equal(eq.cpu().numpy(), CO.equalize_hist_u8(want_gray))


def test_configs1_single_4k_rgb_gaussian_sobel_equalize(dev):
    """The bench step on the literal configs[1] input: ONE 3840 x 2160 RGB image, every pixel against the C oracle."""
    bgr = _frames(1, 2160, 3840, 11)[0]
    t = torch.from_numpy(bgr).to(dev)
    np.testing.assert_array_equal(ops.gaussian_blur(t, 5).cpu().numpy(), CO.gaussian_blur_u8(bgr, 5))
    want_gray = CO.bgr2gray_u8(bgr)
    g2, eq = ops.bgr2gray_equalize(t)
    np.testing.assert_array_equal(g2.cpu().numpy(), want_gray)
    np.testing.assert_array_equal(eq.cpu().numpy(), CO.equalize_hist_u8(want_gray))
    gray = ops.bgr2gray(t)
    np.testing.assert_array_equal(ops.sobel(gray).cpu().numpy(), CO.sobel_u8(want_gray))
    np.testing.assert_array_equal(ops.sobel(t).cpu().numpy(), CO.sobel_u8(bgr))
    # the forms that share one read of the colour frame, on the whole frame (270 bands x 3 tiles, the last tile 208 lanes wide)
    want_g5, want_sharp = CO.gaussian_blur_u8(bgr, 5), CO.laplacian_sharpen_u8(bgr)
    g5, gray1 = ops.gaussian_blur_gray(t, 5)
    np.testing.assert_array_equal(g5.cpu().numpy(), want_g5)
    np.testing.assert_array_equal(gray1.cpu().numpy(), want_gray)
    g5, sharp, gray2 = ops.gaussian_sharpen_gray(t, 5)
    np.testing.assert_array_equal(g5.cpu().numpy(), want_g5)
    np.testing.assert_array_equal(sharp.cpu().numpy(), want_sharp)
    np.testing.assert_array_equal(gray2.cpu().numpy(), want_gray)
    sb, eq2 = ops.sobel_lut(gray2, ops.hist_tables(gray2)[1])
    np.testing.assert_array_equal(sb.cpu().numpy(), CO.sobel_u8(want_gray))
    np.testing.assert_array_equal(eq2.cpu().numpy(), CO.equalize_hist_u8(want_gray))


def test_configs1_one_image_step_replayed_from_a_graph(dev):
    """ipx.pipelines.GraphedStep: the one-image configs[1] step captured once into a HIP graph, linear and with the Gaussian
    as a forked branch; replays on NEW frames give the pixels of the eager calls and of the C oracle, the histogram
    workspace's tickets survive replay after replay, and a frame of another shape is refused."""
    frames = _frames(4, 2160, 3840, 12)
    frames[2][:] = 77                                                   # a constant frame: the single-valued table
    side = torch.cuda.Stream(device=dev)
    for fork in (None, side):
        step = P.GraphedStep(lambda x: P.configs1_one_image(ops, x, fork), torch.from_numpy(frames[0]).to(dev))
        for k in (1, 2, 3, 0, 1):
            t = torch.from_numpy(frames[k]).to(dev)
            g5, gray, eq, sb = [o.clone() for o in step(t)]
            e5, egray, eeq, esb = P.configs1_one_image(ops, t)
            assert torch.equal(g5, e5) and torch.equal(gray, egray) and torch.equal(eq, eeq) and torch.equal(sb, esb), f"frame {k}"
        want_gray = CO.bgr2gray_u8(frames[1])
        np.testing.assert_array_equal(g5.cpu().numpy(), CO.gaussian_blur_u8(frames[1], 5))
        np.testing.assert_array_equal(gray.cpu().numpy(), want_gray)
        np.testing.assert_array_equal(eq.cpu().numpy(), CO.equalize_hist_u8(want_gray))
        np.testing.assert_array_equal(sb.cpu().numpy(), CO.sobel_u8(want_gray))
        with pytest.raises(ValueError):
            step(torch.zeros((1080, 1920, 3), dtype=torch.uint8, device=dev))
        # uploading straight into the static input and replaying without arguments is the zero-copy form
        step.inputs[0].copy_(torch.from_numpy(frames[3]))
        out = step()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out[2].cpu().numpy(), CO.equalize_hist_u8(CO.bgr2gray_u8(frames[3])))


def test_lone_frames_take_taller_band_tiles(dev):
    """A launch whose 8-row band tiles would need a short second round of resident tiles (769 ... 960 tiles on 768 slots: one 4K
    frame is 810) takes 9- or 10-row tiles of 16-lane-multiple widths instead: the shapes on both sides of that rule, last bands of
    every length, against the C oracle (5-tap Gaussian alone and with the gray image; the 3 x 3 operators, whose tiles may be 10 rows)."""
    for n, h, w in ((1, 3333, 2048), (3, 719, 3840), (1, 2400, 3840), (1, 2161, 3840), (2, 1081, 3840), (1, 3073, 2048), (1, 3841, 2048)):
        frames = _frames(n, h, w, 17 + h)
        t = torch.from_numpy(frames).to(dev)
        g5, gray = ops.gaussian_blur_gray(t, 5)
        g5b, sharp, sob = ops.gaussian_blur(t, 5), ops.laplacian_sharpen(t), ops.sobel(t)
        for i in range(n):
            want = CO.gaussian_blur_u8(frames[i], 5)
            np.testing.assert_array_equal(g5[i].cpu().numpy(), want, err_msg=f"{n}x{h}x{w} gaussian+gray {i}")
            np.testing.assert_array_equal(g5b[i].cpu().numpy(), want, err_msg=f"{n}x{h}x{w} gaussian {i}")
            np.testing.assert_array_equal(gray[i, :, :, 0].cpu().numpy(), CO.bgr2gray_u8(frames[i]), err_msg=f"{n}x{h}x{w} gray {i}")
            np.testing.assert_array_equal(sharp[i].cpu().numpy(), CO.laplacian_sharpen_u8(frames[i]), err_msg=f"{n}x{h}x{w} sharpen {i}")
            np.testing.assert_array_equal(sob[i].cpu().numpy(), CO.sobel_u8(frames[i]), err_msg=f"{n}x{h}x{w} sobel {i}")


def test_configs1_batch_step_both_forms(dev):
    """ipx.pipelines.configs1_step on a batch: 4K- and 1080p-wide frames take the form that reads the colour batch once (Gaussian +
    gray in one launch), narrower or ragged ones the Gaussian and cvtColor + histogram + tables as separate launches; both against
    the C oracle, and the shape query says which is which."""
    assert ops.gaussian_blur_gray_one_pass(2160, 3840, 5) and ops.gaussian_blur_gray_one_pass(4320, 7680, 3)
    assert ops.gaussian_blur_gray_one_pass(1080, 1920, 5)              # two tiles of 180 lanes
    assert not ops.gaussian_blur_gray_one_pass(720, 1280, 5)           # rows shorter than 4 KiB: the strip walker's
    assert not ops.gaussian_blur_gray_one_pass(2160, 3841, 5)          # rows that are not whole 16-byte lanes
    assert not ops.gaussian_blur_gray_one_pass(2160, 3840, 7)
    for n, h, w in ((3, 70, 3840), (3, 90, 1920), (3, 50, 1280), (2, 33, 500)):
        frames = _frames(n, h, w, 13 + w)
        frames[1][:] = 9                                               # a constant frame
        g5, eq, sb = P.configs1_step(ops, torch.from_numpy(frames).to(dev))
        for i in range(n):
            want_gray = CO.bgr2gray_u8(frames[i])
            np.testing.assert_array_equal(g5[i].cpu().numpy(), CO.gaussian_blur_u8(frames[i], 5), err_msg=f"{n}x{h}x{w} gaussian {i}")
            np.testing.assert_array_equal(eq[i, :, :, 0].cpu().numpy(), CO.equalize_hist_u8(want_gray), err_msg=f"{n}x{h}x{w} equalize {i}")
            np.testing.assert_array_equal(sb[i, :, :, 0].cpu().numpy(), CO.sobel_u8(want_gray), err_msg=f"{n}x{h}x{w} sobel {i}")


def test_configs2_full_pipeline_1080p_batch(dev):
    n, h, w = 8, 1080, 1920
    frames = _frames(n, h, w, 1002)
    got = P.full_pipeline(ops, torch.from_numpy(frames).to(dev))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    tabs = set()
    for i in range(n):
        want = _oracle_full_pipeline(frames[i])
        for name, ref in want.items():
            g = got[name][i]
            np.testing.assert_array_equal(g[..., 0] if ref.ndim == 2 else g, ref, err_msg=f"frame {i} {name}")
        tabs.add(luts.log(2.0, O.hist_u8(want["gray"])).tobytes())
    assert len(tabs) >= 4                                       # the batch really needs one log table per image


def test_configs2_batch_sharder_keeps_input_order(dev):
    """16 frames through BatchSharder.map (image i -> worker i mod G; two workers on the one GPU here): results come
    back in input order and equal the oracle's."""
    n, h, w = 16, 1080, 1920
    frames = _frames(n, h, w, 77)
    sh = BatchSharder([dev, dev])

    def fn(o, t):
        r = P.full_pipeline(o, t[None])
        return r["histogram_eq"][0, :, :, 0], r["log"][0, :, :, 0], r["sharpening"][0]
    res = sh.map(list(frames), fn)
    for i in (0, 1, 2, 7, 8, 15):
        gray = CO.bgr2gray_u8(frames[i])
        np.testing.assert_array_equal(res[i][0], CO.equalize_hist_u8(gray), err_msg=f"frame {i} eq")
        np.testing.assert_array_equal(res[i][1], O.log_u8(gray, 2.0), err_msg=f"frame {i} log")
        np.testing.assert_array_equal(res[i][2], CO.laplacian_sharpen_u8(frames[i]), err_msg=f"frame {i} sharpen")
    # every frame is distinct, so an order mix-up cannot go unnoticed on the frames not checked against the oracle
    sums = [int(r[0].astype(np.int64).sum()) for r in res]
    assert len(set(sums)) == n


@pytest.mark.parametrize("k", [15, 31])
def test_configs3_8k_rgb_large_gaussian(dev, k):
    h, w, r = 4320, 7680, k // 2
    g = torch.Generator(device="cpu").manual_seed(8000 + k)
    img = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, generator=g)
    got = ops.gaussian_blur(img.to(dev), k).cpu().numpy()
    a = img.numpy()
    # row bands (full width: left and right borders included): top, bottom, interior tile seams (64-row tiles)
    for y0, y1 in ((0, 96), (h - 96, h), (2048 - 40, 2048 + 40), (1000, 1080)):
        b0, b1 = max(0, y0 - r), min(h, y1 + r)
        ref = CO.gaussian_blur_u8(a[b0:b1], k)                 # the band's own REFLECT_101 is right only at the image edge
        np.testing.assert_array_equal(got[y0:y1], ref[y0 - b0:y0 - b0 + (y1 - y0)], err_msg=f"k={k} rows {y0}:{y1}")
    # column bands over the full height: 128-byte tile seams in x and both side borders
    for x0, x1 in ((0, 64), (w - 64, w), (3829, 3861)):
        c0, c1 = max(0, x0 - r), min(w, x1 + r)
        ref = CO.gaussian_blur_u8(np.ascontiguousarray(a[:, c0:c1]), k)
        np.testing.assert_array_equal(got[:, x0:x1], ref[:, x0 - c0:x0 - c0 + (x1 - x0)], err_msg=f"k={k} cols {x0}:{x1}")
    # the numpy restatement agrees with the C one on a corner crop (two independent restatements)
    np.testing.assert_array_equal(got[:40, :50], O.gaussian_blur_u8(a[:40 + r, :50 + r], k)[:40, :50])


def test_configs4_32k_gray_row_partitions(dev):
    H = W = 32768
    parts, rows = 8, 4096
    g = torch.Generator(device=dev).manual_seed(32768)
    img = torch.empty((H, W), dtype=torch.uint8, device=dev)
    for p in range(parts):                                      # generated on the device, 128 MiB slabs; low-contrast slabs
        lo, hi = (0, 256) if p % 2 == 0 else (30 + p, 200 + p)
        img[p * rows:(p + 1) * rows] = torch.randint(lo, hi, (rows, W), dtype=torch.uint8, device=dev, generator=g)
    eq, hist, lut = ops.equalize_hist(img, return_aux=True)
    assert int(hist.to(torch.int64).sum()) == H * W == 1 << 30
    # per-partition histograms (what each of 8 GPUs computes) equal torch's own count and sum to the global one
    hsum = torch.zeros(256, dtype=torch.int64, device=dev)
    for p in range(parts):
        hp = ops.hist(img[p * rows:(p + 1) * rows]).to(torch.int64)
        ref = torch.zeros(256, dtype=torch.int64, device=dev)
        for q in range(4):                                      # bincount in 32 MiB pieces (int64 temporaries)
            ref += torch.bincount(img[p * rows + q * 1024:p * rows + (q + 1) * 1024].flatten().to(torch.int64), minlength=256)
        assert torch.equal(hp, ref), f"partition {p} histogram"
        hsum += hp
    assert torch.equal(hsum, hist[0].to(torch.int64))
    hist_np = hsum.cpu().numpy()
    want_lut = O.equalize_lut_from_hist(hist_np)
    np.testing.assert_array_equal(lut[0].cpu().numpy(), want_lut)
    np.testing.assert_array_equal(CO.equalize_lut_from_hist(hist_np.astype(np.uint32)), want_lut)
    lut_from_sum = ops.equalize_lut(hsum.to(torch.int32))       # the table every rank builds after the all-reduce
    whole = {"box3": ops.box3(img), "sobel": ops.sobel(img)}
    for p in range(parts):
        y0, y1 = p * rows, (p + 1) * rows
        part = img[y0:y1]
        top = img[y0 - 1:y0] if p > 0 else None
        bot = img[y1:y1 + 1] if p < parts - 1 else None
        assert torch.equal(ops.apply_lut_dev(part, lut_from_sum), eq[y0:y1]), f"partition {p} equalize"
        for name in ("box3", "sobel"):
            assert torch.equal(getattr(ops, name)(part, top_halo=top, bottom_halo=bot), whole[name][y0:y1]), f"partition {p} {name}"
    # oracle on row bands across two partition seams, the image top and the image bottom (full 32768-pixel rows)
    for y0, y1 in ((4096 - 8, 4096 + 8), (7 * 4096 - 8, 7 * 4096 + 8), (0, 12), (H - 12, H)):
        b0, b1 = max(0, y0 - 1), min(H, y1 + 1)
        band = img[b0:b1].cpu().numpy()
        sl = slice(y0 - b0, y0 - b0 + (y1 - y0))
        np.testing.assert_array_equal(whole["box3"][y0:y1].cpu().numpy(), CO.box3_u8(band)[sl], err_msg=f"box3 rows {y0}:{y1}")
        np.testing.assert_array_equal(whole["sobel"][y0:y1].cpu().numpy(), CO.sobel_u8(band)[sl], err_msg=f"sobel rows {y0}:{y1}")
        np.testing.assert_array_equal(eq[y0:y1].cpu().numpy(), want_lut[img[y0:y1].cpu().numpy()], err_msg=f"equalize rows {y0}:{y1}")


@pytest.mark.parametrize("world", [2, 3])
def test_row_partition_ranks_run_the_hip_library(dev, world):
    """SURVEY.md section 8(e) case 2 rehearsed on ONE GPU: `world` fresh child processes (started, not exec'd), each
    owning a block of rows of one image on cuda:0, exchange halo rows / all-reduce the histogram through
    torch.distributed (gloo; RowPartition stages device tensors through the host) and run libipx on their rows:
    conv3, Gaussian 5 / 7 / 15 / 31, equalize_hist and log_transform equal the single-process whole-image result."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "rowpart_worker.py")
    procs = [subprocess.Popen([sys.executable, "-u", worker, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=200)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and o.strip().endswith("ok"), f"rank {r}:\n{o}"
