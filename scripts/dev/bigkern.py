"""Workload for profiling the 9..31-tap kernel: BASELINE configs[3] (one 8K BGR frame, k = 15 and 31) and 2 x 8K, a few launches each."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch, ipx.ops as ops
one = torch.randint(0, 256, (1, 4320, 7680, 3), dtype=torch.uint8).cuda()
two = torch.randint(0, 256, (2, 4320, 7680, 3), dtype=torch.uint8).cuda()
for _ in range(5):
    for k in (9, 15, 31):
        ops.gaussian_blur(one, k); ops.gaussian_blur(two, k)
torch.cuda.synchronize()
