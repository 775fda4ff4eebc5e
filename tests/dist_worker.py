"""Worker for the world_size-2 gloo test: the multi-GPU sweep plumbing
(phaserotate.lv2_amd/sweep_dist.py) with the per-shard peaks computed by the
CPU oracle in place of the HIP engine (there is no GPU in the CPU test box)."""
import os
import sys

import numpy as np


def run(rank, world, port, blksiz, n_chn, n_frames, outdir, two_phase=False):
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
    def block(b):
        blk = np.zeros((blksiz, n_chn), np.float32)
        seg = data[b * blksiz:(b + 1) * blksiz]
        blk[:len(seg)] = seg
        return blk

    def shard_peaks(blocks):
        """the oracle's table over the given blocks of this shard (each with the block in front of it as state)"""
        e, prev = orc.CliEngine(blksiz, n_chn), None
        for b in blocks:
            if b > 0 and prev != b - 1:
                e.analyze(block(b - 1), 1, 0, 1, -1, False)  # empty angle range: state only (halo)
            e.analyze(block(b), 0, 360, 1, -1, b == 0)
            prev = b
        if rank == world - 1 and blocks and blocks[-1] == b1 - 1:
            e.analyze(np.zeros((blksiz, n_chn), np.float32), 0, 360, 1, -1, False)  # flush block (Q2)
        return e.peaks().copy()

    if two_phase:
        # sweep_dist.sweep_two_phase -- the protocol the GPU ranks run: seed, MAX all-reduce, main pass, MAX all-reduce -- with a
        # stand-in engine whose two entry points are the oracle (a seed is a LOWER bound: the sampled blocks' peaks scaled down,
        # as the device's seed kernel does; the main pass folds the exact peaks over whatever the table holds)
        calls = []

        class OracleShard:
            def _table(self, ptr):
                import ctypes
                return np.ctypeslib.as_array((ctypes.c_float * (n_chn * 360)).from_address(ptr)).reshape(n_chn, 360)

            def sweep_seed_dev(self, d_in, stride, n, a0, a1, a_s, chn, d_peaks, flags, n_blocks, stream):
                calls.append("seed")
                sample = list(range(b0, b1, max(1, (b1 - b0) // max(1, n_blocks))))
                t = self._table(d_peaks)
                np.maximum(t, np.float32(0.9999) * shard_peaks(sample), out=t)

            def sweep_dev(self, d_in, stride, n, a0, a1, a_s, chn, d_peaks, flags, stream):
                calls.append("main")
                t = self._table(d_peaks)
                np.maximum(t, shard_peaks(list(range(b0, b1))), out=t)

        peaks = torch.zeros((n_chn, 360), dtype=torch.float32)
        x = torch.zeros((n_chn, max(1, (b1 - b0) * blksiz)), dtype=torch.float32)  # (the stand-in reads the file itself)
        n_reduce = [0]

        def counted(t):
            n_reduce[0] += 1
            return sweep_dist.allreduce_peaks(t)
        sweep_dist.sweep_two_phase(OracleShard(), x, peaks, 0, world, 0, (0, 360, 1), allreduce=counted)
        assert calls == ["seed", "main"] and n_reduce[0] == 2, (calls, n_reduce)
    else:
        peaks = torch.from_numpy(shard_peaks(list(range(b0, b1))))
        sweep_dist.allreduce_peaks(peaks)
    np.save(os.path.join(outdir, "peaks_%d.npy" % rank), peaks.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), a[6], len(a) > 7 and a[7] == "two_phase")
