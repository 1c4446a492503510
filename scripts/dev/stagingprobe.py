"""Dev probe (GPU box): host<->device staging cost for one 4K frame, per-call pin_memory vs a persistent pinned buffer."""
import time
import numpy as np
import torch

dev = torch.device("cuda:0")
for nbytes, name in ((2160 * 3840, "4K gray"), (2160 * 3840 * 3, "4K colour")):
    a = np.random.default_rng(0).integers(0, 256, nbytes, dtype=np.uint8)
    pin_in = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    pin_out = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def up_alloc():
        t = torch.from_numpy(a).pin_memory().to(dev, non_blocking=True)
        torch.cuda.synchronize()
        return t

    def up_pageable():
        t = torch.from_numpy(a).to(dev)
        torch.cuda.synchronize()
        return t

    def up_persist():
        pin_in.numpy()[:] = a
        t = pin_in.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        return t

    def down_pageable():
        return d.to("cpu").numpy()

    def down_persist():
        pin_out.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        return np.array(pin_out.numpy())

    def down_persist_view():
        pin_out.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        return pin_out.numpy()

    for fn in (up_alloc, up_pageable, up_persist, down_pageable, down_persist, down_persist_view):
        for _ in range(3):
            fn()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        print(f"{name:10s} {fn.__name__:18s} {(time.perf_counter() - t0) / 10 * 1e3:7.2f} ms", flush=True)
