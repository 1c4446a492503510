"""dev: print the MFMA separable kernel's per-phase cycle stamps (IPX_MFMA_PROF=1) for one 8K BGR launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-processing_amd"))
import torch, ipx.ops as ops
b8 = torch.randint(0, 256, (2, 4320, 7680, 3), dtype=torch.uint8).cuda()
for _ in range(3): ops.gaussian_blur(b8, 31)
torch.cuda.synchronize()
os.environ["IPX_MFMA_PROF"] = "1"
ops.gaussian_blur(b8, 31); ops.gaussian_blur(b8, 9)
torch.cuda.synchronize()
