"""dev: launches of the wave kernel under rocprofv3 --kernel-trace, in a known order (printed as JSON lines: label, launches).
Usage: rocprofv3 --kernel-trace ... -- python3 scripts/dev/waveprof.py <frames> <k> <mode:tiles,mode:tiles,...>   (dev build, IPX_LIB set)"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch, ipx.ops as ops
frames, k = int(sys.argv[1]), int(sys.argv[2])
cases = [c.split(":") for c in sys.argv[3].split(",")]
x = torch.randint(0, 256, (frames, 4320, 7680, 3), dtype=torch.uint8).cuda()
os.environ["IPX_WAVE"] = "1"
for mode, tiles in cases:
    os.environ["IPX_WAVE_MODE"] = mode
    if tiles != "0": os.environ["IPX_WAVE_TILES"] = tiles
    else: os.environ.pop("IPX_WAVE_TILES", None)
    for _ in range(6): ops.gaussian_blur(x, k)
    torch.cuda.synchronize()
    print(json.dumps({"mode": mode, "tiles": tiles, "launches": 6}), flush=True)
