"""Multi-GPU layer the reference does not have.

* `BatchSharder`   -- a batch of independent images, image i -> GPU i mod G, one host thread, one HIP stream and
                      pinned staging per GPU, host-side gather in input order.  No collective.  (server-side use)
* `shard_range` / `gather_in_order` -- the same partition for one-process-per-GPU launches (torchrun); the
                      gather is a host-side object gather, never a data-path collective.
* `RowPartition`   -- ONE giant image split into contiguous row blocks over the ranks of a torch.distributed
                      group (RCCL on GPUs): convolutions exchange r halo rows with the two neighbours
                      (send/recv over xGMI), histogram equalization all-reduces the 256-bin histogram (1 KiB) and
                      the log transform all-reduces it too (it needs the global set of levels).  LUT ops need nothing.

The arithmetic goes through an `ops` object with the ipx.ops surface (default: the HIP library).
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Sequence

import numpy as np


def shard_range(n: int, rank: int, world: int) -> range:
    """Images rank `rank` of `world` processes: i with i % world == rank (round-robin)."""
    return range(rank, n, world)


def gather_in_order(local: List, n: int, rank: int, world: int, group=None) -> Optional[List]:
    """Host-side gather of per-rank result lists back into input order (rank 0 gets the list, others None)."""
    import torch.distributed as dist
    if world == 1:
        return list(local)
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(list(local), bucket, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * n
    for r in range(world):
        for j, i in enumerate(shard_range(n, r, world)):
            out[i] = bucket[r][j]
    return out


class BatchSharder:
    """map(images, fn): fn(ops, tensor) -> tensor | tuple of tensors, run on GPU (i mod G) for image i."""

    def __init__(self, devices: Optional[Sequence[str]] = None, ops=None):
        import torch
        self.torch = torch
        if devices is None:
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        if not devices:
            raise RuntimeError("BatchSharder: no GPU visible (there is no CPU fallback)")
        self.devices = [torch.device(d) for d in devices]
        if ops is None:
            from . import _lib, ops as _ops
            _lib.load()
            ops = _ops
        self.ops = ops

    def _worker(self, di: int, images, fn, out, errors):
        torch = self.torch
        dev = self.devices[di]
        try:
            cuda = dev.type == "cuda"
            stream = torch.cuda.Stream(device=dev) if cuda else None
            ctx = torch.cuda.stream(stream) if cuda else _Null()
            with ctx:
                # persistent pinned staging (as GpuBackend in app.py): two upload buffers used in turn -- a buffer is reused
                # only after the event behind its last H2D copy -- and one download buffer; `.pin_memory()` per image is a
                # page-locking system call per upload, a pageable D2H copy of a 4K colour frame takes 5x the pinned one
                up = [None, None]
                up_done = [None, None]
                down = [None]

                def upload(a, k):
                    if not cuda or a.dtype != np.uint8 or a.size == 0:
                        return torch.from_numpy(a).to(dev)
                    j = k & 1
                    if up[j] is None or up[j].numel() < a.size:
                        up[j] = torch.empty(max(a.size, 1 << 20), dtype=torch.uint8).pin_memory()
                    elif up_done[j] is not None:
                        up_done[j].synchronize()
                    view = up[j][:a.size].view(a.shape)
                    view.numpy()[...] = a
                    t = view.to(dev, non_blocking=True)
                    up_done[j] = torch.cuda.Event()
                    up_done[j].record(stream)
                    return t

                def download(t):
                    t = t.contiguous()
                    n = t.numel() * t.element_size()
                    if not cuda or t.dtype != torch.uint8 or n == 0:
                        return t.to("cpu").numpy()
                    if down[0] is None or down[0].numel() < n:
                        down[0] = torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory()
                    view = down[0][:n].view(t.shape)
                    view.copy_(t, non_blocking=True)
                    stream.synchronize()
                    return np.array(view.numpy())                   # the buffer is reused by the next result

                pending = []
                for k, i in enumerate(range(di, len(images), len(self.devices))):
                    res = fn(self.ops, upload(np.ascontiguousarray(images[i]), k))
                    pending.append((i, res))
                for i, res in pending:                              # copies back queue behind the kernels
                    if isinstance(res, tuple):
                        out[i] = tuple(download(r) for r in res)
                    else:
                        out[i] = download(res)
                if cuda:
                    stream.synchronize()
        except BaseException as e:  # noqa: BLE001 - reported to the caller thread
            errors.append((di, e))

    def map(self, images: Sequence[np.ndarray], fn: Callable) -> List:
        out: List = [None] * len(images)
        errors: List = []
        threads = [threading.Thread(target=self._worker, args=(d, images, fn, out, errors), daemon=True)
                   for d in range(min(len(self.devices), max(1, len(images))))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise RuntimeError(f"BatchSharder: device {self.devices[errors[0][0]]} failed: {errors[0][1]!r}") from errors[0][1]
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class RowPartition:
    """Contiguous row block [y0, y1) of an H-row image on each rank of `group`."""

    def __init__(self, height: int, rank: int, world: int, group=None, ops=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.h, self.rank, self.world, self.group = height, rank, world, group
        base, rem = divmod(height, world)
        self.y0 = rank * base + min(rank, rem)
        self.y1 = self.y0 + base + (1 if rank < rem else 0)
        self.min_rows = base                                          # every shard has at least this many rows
        if ops is None:
            from . import _lib, ops as _ops
            _lib.load()
            ops = _ops
        self.ops = ops
        # gloo carries host tensors only: device tensors are staged through the host then (rehearsals on one GPU, CPU
        # tests); RCCL ("nccl") moves device memory directly over xGMI
        self.host_staging = world > 1 and dist.is_initialized() and dist.get_backend(group) == "gloo"

    def _wire(self, t):
        """The tensor as it goes on the wire: contiguous, on the host when the backend cannot carry device memory."""
        t = t.contiguous()
        return t.cpu() if (self.host_staging and t.is_cuda) else t

    def rows(self) -> slice:
        return slice(self.y0, self.y1)

    # -- halo exchange ------------------------------------------------------------------------------
    def exchange_halo(self, local, r: int):
        """Send my first/last r rows to the upper/lower neighbour and receive theirs.  Returns (top, bottom);
        None where the shard touches the image border (REFLECT_101 is applied locally there)."""
        torch, dist = self.torch, self.dist
        if r == 0 or self.world == 1:
            return None, None
        if self.min_rows < r:
            raise ValueError(f"row shards of {self.min_rows} rows are shorter than the halo radius {r}")
        up, down = self.rank - 1, self.rank + 1
        first, last = self._wire(local[:r]), self._wire(local[-r:])
        top = torch.empty_like(first) if up >= 0 else None
        bot = torch.empty_like(last) if down < self.world else None
        ops_ = []
        if up >= 0:
            ops_.append(dist.P2POp(dist.isend, first, up, group=self.group))
            ops_.append(dist.P2POp(dist.irecv, top, up, group=self.group))
        if down < self.world:
            ops_.append(dist.P2POp(dist.isend, last, down, group=self.group))
            ops_.append(dist.P2POp(dist.irecv, bot, down, group=self.group))
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
        if self.host_staging and local.is_cuda:
            top = top.to(local.device) if top is not None else None
            bot = bot.to(local.device) if bot is not None else None
        return top, bot

    def conv3(self, local, name: str):
        """name in {'laplacian_sharpen', 'box3', 'sobel'}."""
        top, bot = self.exchange_halo(local, 1)
        return getattr(self.ops, name)(local, top_halo=top, bottom_halo=bot)

    def gaussian_blur(self, local, k: int):
        top, bot = self.exchange_halo(local, k // 2)
        return self.ops.gaussian_blur(local, k, top_halo=top, bottom_halo=bot)

    # -- global histogram -----------------------------------------------------------------------------
    def global_hist(self, local):
        h = self.ops.hist(local).reshape(256).contiguous()
        if self.world > 1:
            w = self._wire(h)
            self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM, group=self.group)
            h = w.to(h.device) if w is not h else h
        return h

    def equalize_hist(self, local):
        """Local 256-bin histogram -> all-reduce(sum) of 1 KiB -> identical table on every rank -> local apply."""
        lut = self.ops.equalize_lut(self.global_hist(local))
        return self.ops.apply_lut_dev(local, lut)

    def equalize_hist_and_conv3(self, local, name: str):
        """equalize_hist(local) and conv3(local, name) of a gray row block with ONE read of it for the table pass and the filter
        (ipx_conv3_lut_u8): histogram -> all-reduce -> table, halo exchange, then both outputs.  Returns (equalized, filtered)."""
        lut = self.ops.equalize_lut(self.global_hist(local))
        top, bot = self.exchange_halo(local, 1)
        filtered, eq = self.ops.conv3_lut(name, local, lut, top_halo=top, bottom_halo=bot)
        return eq, filtered

    def log_transform(self, local, c: float = 2.0):
        from . import luts
        present = self.global_hist(local).cpu().numpy()
        return self.ops.apply_lut(local, luts.log(c, present))
