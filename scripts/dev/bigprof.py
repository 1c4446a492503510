#!/usr/bin/env python3
"""Small workload for PMC passes: the 9..31-tap kernel on 2 x 8K BGR."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch
import ipx.ops as ops
b8 = torch.randint(0, 256, (2, 4320, 7680, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    ops.gaussian_blur(b8, 15); ops.gaussian_blur(b8, 31)
torch.cuda.synchronize()
