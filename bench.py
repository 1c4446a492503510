#!/usr/bin/env python3
"""bench.py -- megapixels/s of the image hot path on MI355X (BASELINE.json metric).

Workload at every N (weak scaling, one process per GPU, no data-path collective): BASELINE.json configs[1],
"3840x2160 RGB image, 5x5 Gaussian + Sobel edge + hist-eq", as a batch of --batch such images already resident
in HBM.  One step = for every image of the batch:
    gauss5  = cv2.GaussianBlur(bgr,(5,5),0)                 (colour, 6 B/px algorithmic)
    gray,eq = cv2.cvtColor(bgr, BGR2GRAY); cv2.equalizeHist(gray)   (4 + 2 B/px as two calls)
    sobel   = 3x3 Sobel magnitude of gray                   (2 B/px)   <- the "conv3x3" of the metric
Three launches per step since round 3, each input read once: the Gaussian AND the gray image from one read of the colour batch
(ipx_gaussian_blur_gray_u8: 3 read + 3 + 1 written per pixel); the gray batch's histograms + equalize tables (1 read); Sobel AND
equalizeHist's table pass on one read of the gray batch (ipx_sobel_lut_u8: 1 read + 2 written) = 11 B/px instead of 14.
--order fused_tail keeps the colour batch's two readers apart, --order sobel_mid is round 2's four-launch form.
value = (ranks x batch x 8.2944 MP) / step time: pixels pushed through the whole pipeline per second.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the 5x5 Gaussian) by its algorithmic
bytes and its duration between two HIP events in every timed step; `kernels` lists every op of the step the same way
(the other two ops from K instrumented steps run after the timed region); `cpu_baseline` times the C oracle (a port, not the
reference: OpenCV is absent) on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))

H, W = 2160, 3840
MP = H * W / 1e6
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(frames: int, passes: int):
    """C oracle (oracle/_build/liboracle.so, OpenMP over rows) on `frames` distinct 4K frames, `passes` times each:
    the same 4-op pipeline as the GPU step, on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import coracle as CO
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(frames)]
    CO.gaussian_blur_u8(imgs[0][:64], 5)        # load + warm the library
    t0 = time.perf_counter()
    done = 0
    for _ in range(passes):
        for im in imgs:
            CO.gaussian_blur_u8(im, 5)
            g = CO.bgr2gray_u8(im)
            CO.sobel_u8(g)
            CO.equalize_hist_u8(g)
            done += 1
        if time.perf_counter() - t0 > 30.0:     # bounded: never more than ~30 s of CPU work
            break
    dt = time.perf_counter() - t0
    return {"value": round(done * MP / dt, 2), "unit": "megapixels/s", "cores": threads, "kind": "port",
            "sample": f"{done} passes over 3840x2160 RGB frames ({frames} distinct), same 4-op pipeline, C oracle "
                      f"(OpenMP, {threads} threads, {dt:.1f} s of CPU work); host reports {os.cpu_count()} logical CPUs. "
                      "A port, not the reference: OpenCV is absent from this image"}


def spawn_ranks(n: int, rank_timeout: float) -> int:
    """Run `n` ranks of this script (one per GPU) as child processes, watch ALL of them, and relay rank 0's output.
    The first rank that exits non-zero ends the run: the others are terminated (then killed) and the exit code is non-zero,
    with that rank's stderr tail -- a rank that dies before the rendezvous must not leave the rest waiting in it.  The same
    after `rank_timeout` seconds without an exit.  An 'address already in use' failure of the rendezvous is retried on a
    new port.  The parent makes no GPU call (children are started with subprocess, nothing is exec'd over an initialised
    process)."""
    import socket
    import subprocess
    import tempfile

    def tail(f, nbytes=3000):
        f.flush(); f.seek(0, os.SEEK_END)
        size = f.tell(); f.seek(max(0, size - nbytes))
        return f.read().decode("utf-8", "replace")

    def stop(procs):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs, outs, errs = [], [], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            outs.append(tempfile.TemporaryFile())
            errs.append(tempfile.TemporaryFile())
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=outs[r], stderr=errs[r]))
        t0 = time.time()
        failed, timed_out = None, False
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > rank_timeout:
                timed_out = True
                break
            time.sleep(0.1)
        if failed is None and not timed_out:
            sys.stdout.write(tail(outs[0], 1 << 22))
            sys.stdout.flush()
            sys.stderr.write(tail(errs[0]))
            return 0
        stop(procs)
        if failed is not None:
            r, c = failed
            text = tail(errs[r])
            if attempt < 2 and ("address already in use" in text.lower() or "EADDRINUSE" in text):
                print(f"bench.py: rendezvous port {port} was taken, retrying on another one", file=sys.stderr)
                continue
            print(f"bench.py: rank {r} exited with code {c}; the other ranks were stopped.  Its stderr ends:\n{text}", file=sys.stderr)
            return 1
        alive = [r for r, c in enumerate(codes) if c is None]
        print(f"bench.py: no result after --rank-timeout {rank_timeout:.0f} s; ranks still running: {alive} (stopped).  "
              f"stderr of rank {alive[0]} ends:\n{tail(errs[alive[0]])}", file=sys.stderr)
        return 1
    return 1


def run_rowpart(args, torch, dist, ops, dev, rank, world):
    """BASELINE.json configs[4]: one S x S gray image, rows split into contiguous blocks over the ranks (SURVEY.md 8(e) case 2).
    One step = cv2.equalizeHist of the whole image (local 256-bin histogram -> all-reduce(sum) of 1 KiB -> the same table on
    every rank -> local table pass) + a 3x3 box filter (one halo row sent to and received from each neighbour: 2 x S bytes per
    rank and direction).  Strong scaling: the image is fixed, value = S*S pixels / step time (max over ranks)."""
    from ipx.sharder import RowPartition
    S = args.rowpart_size
    part = RowPartition(S, rank, world, ops=ops)
    rows = part.y1 - part.y0
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    local = torch.empty((rows, S), dtype=torch.uint8, device=dev)
    for y in range(0, rows, 2048):                             # filled in pieces: a 1 GiB host tensor at once is not needed
        n = min(2048, rows - y)
        local[y:y + n] = torch.randint(0, 256, (n, S), dtype=torch.uint8, generator=g).to(dev)

    fused = args.rowpart_form == "fused"

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        if fused:                                              # histogram -> all-reduce -> table | halo exchange, then the table pass
            lut = ops.equalize_lut(part.global_hist(local))    # and the box filter on ONE read of the block (ipx_conv3_lut_u8)
            if ev is not None:
                ev[1].record()
            top, bot = part.exchange_halo(local, 1)
            bx, eq = ops.conv3_lut("box3", local, lut, top_halo=top, bottom_halo=bot)
        else:
            eq = part.equalize_hist(local)
            if ev is not None:
                ev[1].record()
            bx = part.conv3(local, "box3")
        if ev is not None:
            ev[2].record()
        return eq, bx

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(30):                                        # clock ramp: a FIXED count -- every step holds collectives, so
        step()                                                 # all ranks must run the same number of them
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [None] * world
        dist.all_gather_object(every, dt)                     # (an object gather: works with RCCL and with the gloo rehearsal alike)
        rank_ms = [float(v) / args.steps * 1e3 for v in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    px_local = rows * S
    ops_t = {}
    stages = ((("hist_tables_gray", "hist_partial_kernel (+ 1 KiB all-reduce) + hist_finish_kernel", 1),
               ("box3_gray+equalize_table_pass", "conv3_lut_kernel<BOX> (+ halo rows)", 3)) if fused else
              (("equalize_hist_gray", "hist_partial_kernel+hist_finish_kernel+lut_kernel (+ 1 KiB all-reduce)", 2),
               ("box3_gray", "conv3_kernel<BOX,1> (+ halo rows)", 2)))
    for j, (name, kern, bpp) in enumerate(stages):
        s = sum(events[i][j].elapsed_time(events[i][j + 1]) for i in range(args.steps)) / args.steps * 1e-3
        gbs = px_local * bpp / s / 1e9
        ops_t[name] = {"kernel": kern, "us_per_step_rank0": round(s * 1e6, 1), "algorithmic_B_per_px": bpp, "GBps": round(gbs, 1),
                       "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4), "timing": "HIP events around the op (collectives included) in K instrumented steps, rank 0's rows"}
    dom_name = max(ops_t, key=lambda k: ops_t[k]["us_per_step_rank0"])
    dom = ops_t[dom_name]
    cpu = None
    if args.cpu_sample > 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np
        import coracle as CO
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))
        sample = np.random.default_rng(0).integers(0, 256, (min(4096, S), S), dtype=np.uint8)
        CO.box3_u8(sample[:64])
        t1 = time.perf_counter()
        done = 0
        while time.perf_counter() - t1 < 15.0 and done < args.cpu_sample:   # bounded: ~15 s of CPU work at most
            CO.equalize_hist_u8(sample)
            CO.box3_u8(sample)
            done += 1
        dtc = time.perf_counter() - t1
        cpu = {"value": round(done * sample.size / 1e6 / dtc, 2), "unit": "megapixels/s", "cores": threads, "kind": "port",
               "sample": f"{done} passes of equalizeHist + 3x3 box over a {sample.shape[0]} x {S} gray block, C oracle (OpenMP, {threads} threads, "
                         f"{dtc:.1f} s of CPU work); host reports {os.cpu_count()} logical CPUs.  A port, not the reference: OpenCV is absent from this image"}
    return {
        "metric": "megapixels/sec (whole node) for conv3x3 + hist-eq on 4K RGB; % of HBM roofline",
        "value": round(S * S / 1e6 / (dt / args.steps), 1), "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "ranks_seen": dist.get_world_size() if world > 1 else 1, "ms_per_step_per_rank": [round(v, 3) for v in rank_ms],
        "config": {"workload": f"BASELINE.json configs[4]: one {S} x {S} gray image row-partitioned over the ranks, hist-eq + 3x3 box per step",
                   "form": args.rowpart_form,
                   "rows_per_rank": rows, "image": f"{S}x{S} u8 gray",
                   "messages_per_step_and_rank": f"halo exchange: {S} bytes to and {S} bytes from each neighbour (<= 2 x {S // 1024} KiB each way); "
                                                 "histogram all-reduce(sum): 256 x int32 = 1 KiB",
                   "parallelism": f"{world} x one process per GPU, contiguous row blocks, send/recv of halo rows + all-reduce of the histogram (RCCL)"},
        "roofline": {"bound": "hbm", "kernel": dom["kernel"], "op": dom_name, "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["frac_of_8TBps"], "traffic": None,
                     "traffic_source": "not collected for this workload (profiles/ holds the PMC passes of the pipeline workload)",
                     "algorithmic_bytes_per_launch": px_local * dom["algorithmic_B_per_px"], "launch_us": dom["us_per_step_rank0"]},
        "cpu_baseline": cpu, "kernels": ops_t,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=24, help="4K RGB frames per GPU per step (round 3, fused step: 12: 442 k, 16: 450 k, 20: 476 k, 24: 484 k, 28: 486 k, 32: 488 k, 40: 469 k MP/s; round 2: 8: 353 k, 16: 382 k, 24: 408 k, 32: 410 k, 48: 397 k; beyond ~32 frames the gray batch no longer fits the Infinity Cache between the kernels that write and read it)")
    ap.add_argument("--cpu-sample", type=int, default=250, help="passes of the CPU oracle over 4 frames, ~15 s of CPU work, capped at 30 s (0 = skip)")
    ap.add_argument("--order", default="fused", choices=["fused", "fused_tail", "sobel_mid", "table_first", "gauss_last"],
                    help="the step's launches (same results in every form): fused = Gaussian AND gray from one read of the colour batch "
                         "(ipx_gaussian_blur_gray_u8), histograms + tables of the gray batch, then Sobel and the equalize table pass on "
                         "ONE read of the gray batch (ipx_sobel_lut_u8); fused_tail = Gaussian, BGR->gray+histogram, Sobel + table pass; sobel_mid = the four-launch form of "
                         "round 2 (Gaussian, BGR->gray+histogram, Sobel, table pass); table_first = the round-1 order (equalize finished "
                         "before Sobel); gauss_last = sobel_mid with the Gaussian at the end")
    ap.add_argument("--ops-table", action="store_true", help="also time ops that are not part of the step, back to back")
    ap.add_argument("--rank-timeout", type=float, default=420.0, help="--gpus N without a launcher: seconds before silent ranks are stopped")
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "rowpart"],
                    help="pipeline = BASELINE configs[1] on a resident batch per GPU (weak scaling, the headline line); rowpart = "
                         "BASELINE configs[4]: ONE gray image of --rowpart-size^2 pixels split into contiguous row blocks over the ranks, "
                         "hist-eq (1 KiB histogram all-reduce) + 3x3 box (one halo row exchanged each way) per step (strong scaling)")
    ap.add_argument("--rowpart-size", type=int, default=32768, help="side of the square gray image of --workload rowpart")
    ap.add_argument("--rowpart-form", default="fused", choices=["fused", "separate"],
                    help="--workload rowpart: fused = the equalize table pass and the box filter on one read of the row block "
                         "(ipx_conv3_lut_u8; 4 bytes per pixel and step), separate = equalizeHist, then the filter (5 bytes)")
    args = ap.parse_args()

    # --gpus N without a launcher: start N ranks of this script as CHILD processes, before this process touches the GPU
    # (it never does), wait for them and pass rank 0's JSON line through.  Under torchrun (WORLD_SIZE set) --gpus must agree.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={env_world} set by the launcher")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, args.rank_timeout))

    if os.environ.get("IPX_BENCH_FAIL_RANK") is not None and os.environ.get("IPX_BENCH_FAIL_RANK") == os.environ.get("RANK"):
        sys.exit(3)                                           # test knob: this rank dies before the rendezvous
    if os.environ.get("IPX_BENCH_HANG") and os.environ.get("RANK") is not None:
        time.sleep(3600)                                      # test knob: a silent rank

    import torch
    import torch.distributed as dist
    import ipx._lib as L
    import ipx.ops as ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): IPX_BENCH_DEVICE pins every rank to one card, IPX_BENCH_BACKEND=gloo replaces RCCL
    local = int(os.environ.get("IPX_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("IPX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    L.load()
    L.call("ipx_check_device")

    if args.workload == "rowpart":
        out = run_rowpart(args, torch, dist, ops, dev, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if out is not None:
            print(json.dumps(out))
        return

    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    bgr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g).to(dev)

    STEP_OPS = [  # name, kernels, algorithmic bytes per pixel
        ("gaussian5_bgr", "sep_pow2_band_kernel<3,5,true>", 6),
        ("bgr2gray_hist", "gray_hist_kernel<3>+hist_finish_kernel", 4),      # cvtColor + histogram + equalize table (server.py:245-246, first half)
        ("sobel3_gray", "conv3_kernel<SOBEL,1>", 2),
        ("equalize_table_pass", "lut_kernel", 2),                            # cv2.equalizeHist's second half: dst = table[gray]
        # Sobel and that table pass on one read of the gray batch: 1 byte read + 2 written per pixel (2 + 2 as two launches)
        ("sobel3_gray+equalize_table_pass", "conv3_lut_kernel<SOBEL>", 3),
        # the Gaussian and cvtColor on one read of the colour batch: 3 bytes read, 3 + 1 written per pixel (6 + 4 as two launches)
        ("gaussian5_bgr+bgr2gray", "sep_pow2_band_gray_kernel<5,true>", 7),
        ("hist_tables_gray", "hist_partial_kernel+hist_finish_kernel", 1),   # histograms + equalize tables of the gray batch (1 byte read)
    ]

    ORDERS = {"fused": (5, 6, 4), "fused_tail": (0, 1, 4), "sobel_mid": (0, 1, 2, 3), "table_first": (0, 1, 3, 2), "gauss_last": (1, 2, 3, 0)}
    order = ORDERS[args.order]
    STEP_OPS = [STEP_OPS[i] for i in order]

    DOMINANT = 5 if 5 in order else 0                             # index into the unordered op list: the launch that holds the 5x5 Gaussian

    def step(ev=None, only=None, src=None):
        """One pass of the pipeline over the resident batch; `ev` (optional) receives HIP events at op boundaries,
        recorded on the stream the kernels are launched on (torch's current stream): around every op, or with `only`
        around that op alone (ev[0], ev[1]).  An event between two kernels keeps them from overlapping at their edges
        (measured: 4 events per step cost 2-3 % of the step, scripts/dev/evprobe.py), so the timed region carries only
        the two around the dominant kernel.  cvtColor + equalizeHist is the fused entry point split at its own seam
        (ipx_bgr2gray_equalize_u8 with dst = NULL, then ipx_lut_dev_u8 with the tables it produced): the same kernels as the
        fused call, with Sobel free to run between them, where the gray batch it reads has just been written (step 483 us
        against 495 us, scripts/dev/orderab.py).  --order only changes which launch runs when."""
        res = {}
        src = bgr if src is None else src
        if ev is not None and only is None:
            ev[0].record()
        for j, i in enumerate(order):
            if ev is not None and only == i:
                ev[0].record()
            if i == 0:
                res["g5"] = ops.gaussian_blur(src, 5)
            elif i == 1:
                res["gray"], _, res["lut"] = ops.bgr2gray_hist(src)   # cvtColor, histogram taken while gray is written, equalize tables
            elif i == 2:
                res["sb"] = ops.sobel(res["gray"])
            elif i == 3:
                res["eq"] = ops.apply_lut_dev(res["gray"], res["lut"])
            elif i == 4:
                res["sb"], res["eq"] = ops.sobel_lut(res["gray"], res["lut"])
            elif i == 5:
                res["g5"], res["gray"] = ops.gaussian_blur_gray(src, 5)
            else:
                _, res["lut"] = ops.hist_tables(res["gray"])
            if ev is not None and only is None:
                ev[j + 1].record()
            if ev is not None and only == i:
                ev[1].record()
        return res["g5"], res["sb"], res["eq"]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # clock ramp + allocator warm-up, then the W untimed warm-up steps
    t_end = time.time() + 1.0
    while time.time() < t_end:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    dom_events = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(dom_events[i], only=DOMINANT)
    fence()
    dt = time.perf_counter() - t0
    # the same K steps once more, instrumented around every op (outside the timed region): the per-op table
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)] for _ in range(args.steps)]
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    # the LITERAL configs[1]: ONE 4K image per launch (what a single /api/process request sees); same event scheme
    one = bgr[:1]
    lat_steps = 50
    LAT_OPS = ("gaussian5_bgr+bgr2gray", "hist_tables_gray", "sobel3_gray+equalize_table_pass")

    def step_one(ev=None):
        """the step on ONE frame, three launches: the Gaussian and the gray frame from one read of the colour frame; the gray
        frame's histogram + equalize table (one frame is enough blocks for the table to be built inside the counting launch);
        Sobel and the table pass on one read of the gray frame"""
        if ev is not None:
            ev[0].record()
        _, gray1 = ops.gaussian_blur_gray(one, 5)
        if ev is not None:
            ev[1].record()
        _, lut1 = ops.hist_tables(gray1)
        if ev is not None:
            ev[2].record()
        ops.sobel_lut(gray1, lut1)
        if ev is not None:
            ev[3].record()

    for _ in range(5):
        step_one()
    lat_events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(lat_steps)]
    for i in range(lat_steps):
        step_one(lat_events[i])
    torch.cuda.synchronize()
    # the same one-image step captured once into a HIP graph and replayed (ipx.pipelines.GraphedStep): one enqueue per step
    # instead of three calls through Python and ctypes; "forked" = the Gaussian on a second branch of the graph
    import ipx.pipelines as pipelines
    graph_events = {}
    side = torch.cuda.Stream(device=dev)
    for key, fn in (("step_graph", lambda x: pipelines.configs1_one_image(ops, x)),
                    ("step_graph_forked", lambda x: pipelines.configs1_one_image(ops, x, side))):
        gstep = pipelines.GraphedStep(fn, one)
        for _ in range(5):
            gstep()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(lat_steps)]
        for e in evs:
            e[0].record()
            gstep()
            e[1].record()
        torch.cuda.synchronize()
        graph_events[key] = evs
        if key == "step_graph":                              # a queue of single-frame requests: replays back to back
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(lat_steps):
                gstep()
            e1.record()
            torch.cuda.synchronize()
            graph_stream_us = e0.elapsed_time(e1) / lat_steps * 1e3
        del gstep
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [None] * world
        dist.all_gather_object(every, dt)                     # (an object gather: works with RCCL and with the gloo rehearsal alike)
        rank_ms = [float(v) / args.steps * 1e3 for v in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * MP / (dt / args.steps)

    def time_op(fn, reps=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3          # seconds per launch

    out = None
    if rank == 0:
        px = B * H * W

        def row(kern, bpp, s, how):
            gbs = px * bpp / s / 1e9
            return {"kernel": kern, "us_per_launch": round(s * 1e6, 1), "algorithmic_B_per_px": bpp,
                    "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4), "MPps": round(px / s / 1e6, 1),
                    "timing": how}

        # the dominant kernel: HIP events around it in every TIMED step; the other ops of the step: events around every
        # op in K more steps run right after the timed region
        kernels = {}
        for j, (name, kern, bpp) in enumerate(STEP_OPS):
            if order[j] == DOMINANT:
                s = sum(e[0].elapsed_time(e[1]) for e in dom_events) / args.steps * 1e-3
                kernels[name] = row(kern, bpp, s, "HIP events around the op in every timed step (in-pipeline)")
            else:
                s = sum(events[i][j].elapsed_time(events[i][j + 1]) for i in range(args.steps)) / args.steps * 1e-3
                kernels[name] = row(kern, bpp, s, "HIP events around every op in K instrumented steps after the timed region (in-pipeline)")
        # cvtColor + equalizeHist as ONE op (the round-1 table's key): its two halves added up
        if "equalize_table_pass" in kernels and "bgr2gray_hist" in kernels:
            a, b = kernels["bgr2gray_hist"], kernels["equalize_table_pass"]
            kernels["bgr2gray_equalize"] = row(a["kernel"] + "+" + b["kernel"], 6, (a["us_per_launch"] + b["us_per_launch"]) * 1e-6,
                                               "sum of bgr2gray_hist and equalize_table_pass above")
        latency_b1_us = {}
        for j, name in enumerate(LAT_OPS):
            ts = sorted(lat_events[i][j].elapsed_time(lat_events[i][j + 1]) for i in range(lat_steps))
            latency_b1_us[name] = round(ts[len(ts) // 2] * 1e3, 1)
        ts = sorted(lat_events[i][0].elapsed_time(lat_events[i][3]) for i in range(lat_steps))
        latency_b1_us["step"] = round(ts[len(ts) // 2] * 1e3, 1)
        for key, evs in graph_events.items():
            ts = sorted(e[0].elapsed_time(e[1]) for e in evs)
            latency_b1_us[key] = round(ts[len(ts) // 2] * 1e3, 1)
        latency_b1_us["step_graph_queued"] = round(graph_stream_us, 1)
        latency_b1_us["note"] = ("median over 50 steps of ONE 3840x2160 RGB image per launch, HIP events around every op; "
                                 "one image is 25 MB: launch + ramp dominate, not bandwidth.  step_graph: the same step "
                                 "(same kernels, same pixels: tests/test_gpu_configs.py) captured into a HIP graph and "
                                 "replayed, events around ONE replay on an idle GPU (includes the graph's launch latency); "
                                 "step_graph_forked: the Gaussian as a parallel branch; step_graph_queued: 50 replays "
                                 "enqueued back to back, time per step (a queue of single-frame requests)")
        if args.ops_table:
            gray = ops.bgr2gray(bgr)
            # two non-uniform distributions for hist-eq -- every pixel in one bin (the worst case for LDS
            # atomics that collide) and low contrast, clip(N(128, 10))
            gray_const = torch.full_like(gray, 77)
            gray_low = torch.empty(gray.shape, dtype=torch.float32, device=gray.device).normal_(128.0, 10.0).clamp_(0, 255).to(torch.uint8)
            for _ in range(40):        # the one-image launches above leave the chip idle: its first measurements afterwards read
                step()                 # 10-20 % high (clocks), so bring it back to the timed region's state first
            torch.cuda.synchronize()
            for name, kern, bpp, fn in [
                ("equalize_hist_gray_one_bin", "hist_partial_kernel+hist_finish_kernel+lut_kernel", 2, lambda: ops.equalize_hist(gray_const)),
                ("equalize_hist_gray_low_contrast", "hist_partial_kernel+hist_finish_kernel+lut_kernel", 2, lambda: ops.equalize_hist(gray_low)),
                ("gaussian5+sharpen+gray_bgr", "gauss5_sharpen_gray_band_kernel<true>", 10, lambda: ops.gaussian_sharpen_gray(bgr, 5)),
                ("gaussian5_bgr_b2b", "sep_pow2_band_kernel<3,5,true>", 6, lambda: ops.gaussian_blur(bgr, 5)),
                ("bgr2gray", "gray_kernel<3,false>", 4, lambda: ops.bgr2gray(bgr)),
                ("sobel3_gray_b2b", "conv3_kernel<SOBEL,1>", 2, lambda: ops.sobel(gray)),
                ("box3_gray", "conv3_kernel<BOX,1>", 2, lambda: ops.box3(gray)),
                ("equalize_hist_gray", "hist_partial_kernel+hist_finish_kernel+lut_kernel", 2, lambda: ops.equalize_hist(gray)),
                ("laplacian_sharpen_bgr", "conv3_kernel<SHARPEN,3>", 6, lambda: ops.laplacian_sharpen(bgr)),
                ("lut_gray", "lut_kernel<true>", 2, lambda: ops.apply_lut(gray, list(range(255, -1, -1)))),
                ("lut_bgr", "lut_kernel<true>", 6, lambda: ops.apply_lut(bgr, list(range(255, -1, -1)))),
                # SURVEY.md section 8(f) kernels, 2 B/px algorithmic each (gray in, gray out)
                ("lee_w3", "lee_small_kernel<3>", 2, lambda: ops.lee_filter(gray, 3)),
                ("lee_w5", "lee_small_kernel<5>", 2, lambda: ops.lee_filter(gray, 5)),
                ("lee_w7", "lee_small_kernel<7>", 2, lambda: ops.lee_filter(gray, 7)),
                ("lee_w9", "lee_small_kernel<9>", 2, lambda: ops.lee_filter(gray, 9)),
                ("lee_w11", "lee_wide_kernel<11>", 2, lambda: ops.lee_filter(gray, 11)),
                ("lee_w15", "lee_wide_kernel<15>", 2, lambda: ops.lee_filter(gray, 15)),
                ("lee_w31", "lee_wide_kernel<31>", 2, lambda: ops.lee_filter(gray, 31)),
                ("erode_3", "morph_small_kernel<false,1>", 2, lambda: ops.erode(gray, ops.ellipse_se(3))),
                ("dilate_5", "morph_small_kernel<true,2>", 2, lambda: ops.dilate(gray, ops.ellipse_se(5))),
                ("erode_7", "morph_runs_kernel<false>", 2, lambda: ops.erode(gray, ops.ellipse_se(7))),
                ("dilate_31", "morph_runs_kernel<true>", 2, lambda: ops.dilate(gray, ops.ellipse_se(31))),
                ("median_3", "median3_kernel", 2, lambda: ops.median_blur(gray, 3)),
                ("median_5", "median5_kernel", 2, lambda: ops.median_blur(gray, 5)),
                ("median_7", "median7_kernel", 2, lambda: ops.median_blur(gray, 7)),
                ("median_9", "median9_kernel", 2, lambda: ops.median_blur(gray, 9)),
                ("median_11", "median11_kernel", 2, lambda: ops.median_blur(gray, 11)),
                ("median_15", "median_row_kernel<15>", 2, lambda: ops.median_blur(gray, 15)),
                ("median_31", "median_row_kernel<31>", 2, lambda: ops.median_blur(gray, 31)),
            ]:
                kernels[name] = row(kern, bpp, time_op(fn), "20 back-to-back launches between two HIP events")
        dom_name = max(STEP_OPS, key=lambda o: kernels[o[0]]["us_per_launch"])[0]
        dom = kernels[dom_name]
        # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside this process): read from the committed
        # record of the same kernel at the same batch, and labelled with where it came from
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("batch_per_gpu", 16) == B and tj.get(dom_name) is not None:
                    traffic = tj[dom_name]
                    traffic_source = f"profiles/pmc_traffic.json ({tj.get('source', 'rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes')}; recorded at {tj.get('recorded_at', 'round 1')})"
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom["kernel"], "op": dom_name, "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": dom["frac_of_8TBps"], "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": px * dom["algorithmic_B_per_px"], "launch_us": dom["us_per_launch"],
                    "step_algorithmic_GBps": round(px * sum(o[2] for o in STEP_OPS) / (dt / args.steps) / 1e9, 1)}
        cpu = cpu_baseline(4, args.cpu_sample) if (args.cpu_sample > 0 and world == 1) else None
        out = {
            "metric": "megapixels/sec (whole node) for conv3x3 + hist-eq on 4K RGB; % of HBM roofline",
            "value": round(value, 1), "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if world > 1 else 1, "ms_per_step_per_rank": [round(v, 3) for v in rank_ms],
            "config": {"workload": "BASELINE.json configs[1]: 3840x2160 RGB, 5x5 Gaussian + Sobel (3x3) + hist-eq "
                                   "(incl. BGR->gray), batch resident in HBM",
                       "batch_per_gpu": B, "image": "3840x2160x3 u8 BGR interleaved",
                       "launch_order": [STEP_OPS[j][0] for j in range(len(STEP_OPS))],
                       "parallelism": f"{world} x one process per GPU, images sharded, no collective"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "latency_b1_us": latency_b1_us,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
