"""Worker for the world_size-2 gloo test: the multi-GPU sweep plumbing
(phaserotate.lv2_amd/sweep_dist.py) with the per-shard peaks computed by the
CPU oracle in place of the HIP engine (there is no GPU in the CPU test box)."""
import os
import sys

import numpy as np


def run(rank, world, port, blksiz, n_chn, n_frames, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    import orc
    from __graft_entry__ import load_package
    load_package()
    from phaserotate_lv2_amd import sweep_dist
    from conftest import harmonic_stack, xorshift_uniform

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    data = np.stack([0.8 * harmonic_stack(n_frames, f0=110.0 * (1 + 0.37 * c)) +
                     0.1 * xorshift_uniform(n_frames, seed=40 + c) for c in range(n_chn)], axis=1)
    data = data.astype(np.float32)  # [n_frames, n_chn] interleaved

    plan = sweep_dist.plan_shards(n_frames, blksiz, world, align=3)  # the polyphase engine's grid at blksiz 2048
    b0, b1 = plan[rank]
    eng = orc.CliEngine(blksiz, n_chn)

    def block(b):
        blk = np.zeros((blksiz, n_chn), np.float32)
        seg = data[b * blksiz:(b + 1) * blksiz]
        blk[:len(seg)] = seg
        return blk

    if b0 > 0:
        eng.analyze(block(b0 - 1), 1, 0, 1, -1, False)  # empty angle range: state only (halo)
    for b in range(b0, b1):
        eng.analyze(block(b), 0, 360, 1, -1, b == 0)
    if rank == world - 1:
        eng.analyze(np.zeros((blksiz, n_chn), np.float32), 0, 360, 1, -1, False)  # flush block (Q2)
    peaks = torch.from_numpy(eng.peaks().copy())
    sweep_dist.allreduce_peaks(peaks)
    np.save(os.path.join(outdir, "peaks_%d.npy" % rank), peaks.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), a[6])
