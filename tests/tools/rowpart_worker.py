"""One rank of the RowPartition rehearsal on a single GPU (started by tests/test_gpu_configs.py, never collected):
every rank owns a block of rows of the same seeded images on cuda:0, talks to its neighbours through torch.distributed
(gloo here -- device tensors are staged through the host by RowPartition; RCCL on a real multi-GPU node) and runs the
REAL HIP library on its rows.  Each result must equal the rows of the single-process whole-image result, bit for bit.
usage: rowpart_worker.py RANK WORLD PORT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle")]

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ipx._lib as L
        import ipx.ops as ops
        import ipx.pipelines as P
        import oracle as O
        from ipx.sharder import RowPartition
        L.load()
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(2024)
        Hh, Wg, Wc = 1501, 2333, 777                                   # odd sizes: uneven shards, ragged strips
        gray = np.clip(rng.normal(100, 45, (Hh, Wg)), 0, 251).astype(np.uint8)
        gray[Hh // 2:] = np.minimum(gray[Hh // 2:], 170)               # the global maximum lives in rank 0's rows only
        bgr = rng.integers(0, 256, (Hh, Wc, 3), dtype=np.uint8)
        tg, tb = torch.from_numpy(gray).to(dev), torch.from_numpy(bgr).to(dev)
        part = RowPartition(Hh, rank, world)                           # ops = the HIP library
        assert part.host_staging
        lg, lb = tg[part.rows()].clone(), tb[part.rows()].clone()
        # a row-strided local shard (a view into a wider buffer): halo rows arrive densely packed and must be re-laid out
        wide = torch.zeros((part.y1 - part.y0, Wg + 19), dtype=torch.uint8, device=dev)
        wide[:, 3:3 + Wg] = lg
        lg_strided = wide[:, 3:3 + Wg]
        whole = {
            "equalize": ops.equalize_hist(tg), "log": P.log_transform(ops, tg, 2.0),
            "sharpen_bgr": ops.laplacian_sharpen(tb), "sobel": ops.sobel(tg), "box3": ops.box3(tg),
            "gauss5_bgr": ops.gaussian_blur(tb, 5), "gauss31": ops.gaussian_blur(tg, 31), "gauss15_bgr": ops.gaussian_blur(tb, 15),
            "gauss7_strided": ops.gaussian_blur(tg, 7),
        }
        mine = {
            "equalize": part.equalize_hist(lg), "log": part.log_transform(lg, 2.0),
            "sharpen_bgr": part.conv3(lb, "laplacian_sharpen"), "sobel": part.conv3(lg, "sobel"), "box3": part.conv3(lg, "box3"),
            "gauss5_bgr": part.gaussian_blur(lb, 5), "gauss31": part.gaussian_blur(lg, 31), "gauss15_bgr": part.gaussian_blur(lb, 15),
            "gauss7_strided": part.gaussian_blur(lg_strided, 7),
        }
        eq2, bx2 = part.equalize_hist_and_conv3(lg, "box3")            # the table pass and the filter on one read of the block
        eq3, sb3 = part.equalize_hist_and_conv3(lg_strided, "sobel")
        torch.cuda.synchronize()
        bad = [k for k in whole if not torch.equal(mine[k], whole[k][part.rows()])]
        for name, got, want in (("fused.equalize", eq2, whole["equalize"]), ("fused.box3", bx2, whole["box3"]),
                                ("fused.equalize_strided", eq3, whole["equalize"]), ("fused.sobel_strided", sb3, whole["sobel"])):
            if not torch.equal(got, want[part.rows()]):
                bad.append(name)
        gh = part.global_hist(lg)
        if not torch.equal(gh, ops.hist(tg)):
            bad.append("global_hist")
        # and the whole-image results themselves against the oracle on this rank's rows (+ halo)
        y0, y1 = max(0, part.y0 - 1), min(Hh, part.y1 + 1)
        if not np.array_equal(mine["sobel"].cpu().numpy(), O.sobel_u8(gray[y0:y1])[part.y0 - y0:part.y0 - y0 + (part.y1 - part.y0)]):
            bad.append("sobel_vs_oracle")
        if not np.array_equal(mine["equalize"].cpu().numpy(), O.equalize_hist_u8(gray)[part.rows()]):
            bad.append("equalize_vs_oracle")
        if not np.array_equal(mine["log"].cpu().numpy(), O.log_u8(gray, 2.0)[part.rows()]):
            bad.append("log_vs_oracle")
        dist.barrier()
        print(f"rank {rank}/{world} rows {part.y0}:{part.y1} " + ("ok" if not bad else "MISMATCH " + ",".join(bad)), flush=True)
        return 0 if not bad else 1
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
