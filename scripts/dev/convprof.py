#!/usr/bin/env python3
"""Small workload for PMC passes: the strip-walker convolutions and a table pass on 24 x 4K, a few launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
dev = torch.device("cuda:0")
B = 24
bgr = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device=dev)
gray = ops.bgr2gray(bgr)
for _ in range(3):
    ops.sobel(gray); ops.box3(gray); ops.gaussian_blur(bgr, 5); ops.gaussian_blur(gray, 5); ops.laplacian_sharpen(bgr)
    ops.apply_lut(gray, list(range(256)))
torch.cuda.synchronize()
