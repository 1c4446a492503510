"""dev: A/B of the wave-per-strip 9..31-tap kernel (ipx_sepwave.hip) against the tile kernel (ipx_sepmfma.hip) on 8K BGR frames.
Needs the dev build (make dev; IPX_LIB=image-processing_amd/lib_dev/libipx.so): IPX_WAVE=0/1 picks the kernel, IPX_WAVE_TILES the
segment length.  Prints the median HIP-event time per launch and checks that both kernels give the same bytes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch, ipx.ops as ops

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tiles = [t for t in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
ks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [9, 15, 31]
x = torch.randint(0, 256, (frames, 4320, 7680, 3), dtype=torch.uint8).cuda()
for k in ks:
    os.environ["IPX_WAVE"] = "0"
    ref = ops.gaussian_blur(x, k)
    med, mn = timeit(lambda: ops.gaussian_blur(x, k))
    print(f"k={k:2d} frames={frames} tile kernel      : median {med:7.1f} us  min {mn:7.1f} us", flush=True)
    os.environ["IPX_WAVE"] = "1"
    for t in tiles:
        if t != "0": os.environ["IPX_WAVE_TILES"] = t
        else: os.environ.pop("IPX_WAVE_TILES", None)
        got = ops.gaussian_blur(x, k)
        same = bool(torch.equal(got, ref))
        med, mn = timeit(lambda: ops.gaussian_blur(x, k))
        print(f"k={k:2d} frames={frames} wave kernel tiles={t:>3s}: median {med:7.1f} us  min {mn:7.1f} us  same={same}", flush=True)
        if not same:
            d = (got != ref)
            idx = d.nonzero()[:5].tolist()
            print("   mismatches:", int(d.sum()), "first:", idx, flush=True)
