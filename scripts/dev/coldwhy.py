"""dev: what makes the 9..31-tap kernel slower after another kernel?  Each case: a predecessor, then blur(one, 31) (one 8K BGR frame), 6 times.
Run under rocprofv3 --kernel-trace; the blur launches are printed per case by coldwhy_sum (order of cases = order of launches)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch, ipx.ops as ops
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
one = torch.randint(0, 256, (1, 4320, 7680, 3), dtype=torch.uint8).cuda()
other = torch.randint(0, 256, (2, 4320, 7680, 3), dtype=torch.uint8).cuda()       # 199 MB
big = torch.randint(0, 256, (6, 4320, 7680, 3), dtype=torch.uint8).cuda()         # 597 MB
outb = torch.empty_like(big)
def none(): pass
def read_other(): other.flatten().view(torch.int64).sum()                                    # reads 199 MB, writes nothing
def read_big(): big.flatten().view(torch.int64).sum()                                        # reads 597 MB: flushes the caches with clean lines
def write_other(): outb[:2].copy_(other)                                           # reads 199 MB, writes 199 MB (dirty lines)
def blur_two(): ops.gaussian_blur(other, k)
cases = (("none", none), ("read 199 MB", read_other), ("read 597 MB", read_big), ("copy 199 MB", write_other), ("blur 2 frames", blur_two))
if len(sys.argv) > 2: cases = [c for c in cases if c[0].split()[0] in sys.argv[2].split(",")]
for name, pre in cases:
    for _ in range(6):
        pre(); ops.gaussian_blur(one, k)
    torch.cuda.synchronize()
    print(json.dumps({"case": name, "launches": 6}), flush=True)
