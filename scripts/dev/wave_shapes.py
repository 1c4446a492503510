"""dev: wave kernel vs tile kernel (IPX_WAVE=1/0, dev build) over image shapes; HIP-event medians of back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch, ipx.ops as ops
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
shapes = [(1, 512, 512, 3), (1, 1080, 1920, 3), (1, 2160, 3840, 3), (1, 2160, 3840, 1), (8, 1080, 1920, 3), (16, 512, 512, 3), (1, 300, 300, 3), (1, 4320, 7680, 1), (4, 2160, 3840, 4), (1, 4096, 32768, 1)]
for shp in shapes:
    x = torch.randint(0, 256, shp, dtype=torch.uint8).cuda()
    for k in (9, 15, 31):
        r = {}
        for w in ("0", "1"):
            os.environ["IPX_WAVE"] = w
            r[w] = timeit(lambda: ops.gaussian_blur(x, k))
        print(f"{str(shp):24s} k={k:2d}: tile {r['0']:8.1f} us   wave {r['1']:8.1f} us   ratio {r['0'] / r['1']:.2f}", flush=True)
