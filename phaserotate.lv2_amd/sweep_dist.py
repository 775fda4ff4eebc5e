"""Multi-GPU plumbing of the CLI min-peak sweep: one process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
CPU tests).

The reference has no distributed code; its only "reduction" is peak_all()'s max
over channel rows (cli/phase-rotate.cc:287-299) after the per-channel threads
join (:441-443).  Here the independent (angle x channel x frame-block) work
items are sharded by frame-block range: rank r sweeps blocks [b0, b1) of the
file for all requested angles, seeded with `history` frames of input in front
of b0 (the previous block for the dry delay + what the OLA tail needs), and the
per-angle maxima are merged with ONE all-reduce(MAX) of float[n_chn][360]
(2 880 B for stereo: latency-bound, the xGMI link bandwidth is irrelevant).
max is associative, commutative and exact in fp32, so sharding cannot change
the table or the argmin.  The first-block rule (Q1) stays with rank 0 and the
zero-pad + flush block (Q2) with the last rank.
"""
import numpy as np

MAXSAMPLE = 360
SUBSAMPLE = 2


def plan_shards(n_frames, blksiz, world, align):
    """Split ceil(n_frames / blksiz) data blocks into `world` contiguous ranges
    [(b0, b1), ...]; earlier ranks get the remainder.  Empty ranges are allowed
    (more ranks than blocks).  Interior boundaries fall on multiples of `align`
    blocks.  `align` is required: pass Engine.shard_align_blocks() -- the device engine works in
    transform blocks of info.run_align frames (3 blocks of 2048 frames, 7 blocks of 1024 frames, ... in
    polyphase form), and only a shard that starts on such a boundary reproduces the single-GPU arithmetic
    bit for bit (so the merged table, and its argmin, are identical, not just within tolerance).
    align=1 cuts anywhere (a CPU oracle as the per-shard engine has no block grid of its own)."""
    if align is None or int(align) < 1:
        raise ValueError("plan_shards: align must be a positive block count (Engine.shard_align_blocks())")
    nb = (n_frames + blksiz - 1) // blksiz
    units = (nb + align - 1) // align
    base, rem = divmod(units, world)
    out, b = [], 0
    for r in range(world):
        k = base + (1 if r < rem else 0)
        e = min(nb, b + k * align)
        out.append((b, e))
        b = e
    return out


def shard_frames(n_frames, blksiz, plan, rank):
    """(first_frame, n_valid_frames) of rank's shard; only the file's last block may be partial."""
    b0, b1 = plan[rank]
    f0 = b0 * blksiz
    f1 = min(b1 * blksiz, n_frames)
    return f0, max(0, f1 - f0)


def _stream_of(x, stream):
    """The stream a call is queued on when the caller names none: torch's CURRENT stream on x's device -- the stream torch's own
    work on `peaks` (zero_(), dist.all_reduce) is ordered on -- for a CUDA tensor; 0 (the engine's own stream: a CPU oracle
    ignores it) otherwise.  ADVICE r5: with the engine's private non-blocking stream as the default, the first all-reduce could
    merge a table the seeding pass was still writing."""
    if stream is not None:
        return stream
    if getattr(x, "is_cuda", False):
        import torch
        return torch.cuda.current_stream(x.device).cuda_stream
    return 0


def prime_history(engine, halo, stream=None):
    """Pass `halo` ([n_chn, H] device tensor: the H frames in front of this
    rank's shard) into the engine's input history, which then equals what a
    single-GPU pass would hold at this point (round 5: a copy kernel,
    prot_engine_prime_dev; until then a Hilbert launch whose output was thrown away)."""
    n_chn, h = halo.shape
    stream = _stream_of(halo, stream)
    if hasattr(engine, "prime_dev"):
        engine.prime_dev(halo.data_ptr(), h, h, stream)
        return halo
    import torch
    scratch = torch.empty_like(halo)
    engine.hilbert_dev(halo.data_ptr(), h, h, scratch.data_ptr(), h, h, 0, stream)
    return scratch  # keep alive until the stream has consumed it


def sweep_two_phase(engine, x, peaks, flags, world, stream=None, ang=(0, 360, 1), allreduce=None, halo=None, fresh=False):
    """One rank's sharded sweep in two phases: seed the pruning bound from a sparse sample of this shard (x: [n_chn, n] device
    tensor), merge the sampled tables over the ranks (MAX all-reduce of `peaks`), sweep against the GLOBAL bound, merge again.
    Bit-identical to a single pass over the whole file (every seeded entry is a peak of real frames; max is exact).
    fresh=True: the pass starts from scratch in ONE call -- engine reset, `peaks` zeroed, the history primed with `halo` ([n_chn, H]
    device tensor: the frames in front of the shard; None for the file's first shard), the seeding pass (prot_sweep_shard_seed_dev).
    stream=None: torch's current stream (where peaks.zero_() and the default all-reduce are ordered); a raw HIP stream handle otherwise --
    then `allreduce` must be ordered on that stream too."""
    allreduce = allreduce or allreduce_peaks
    n_chn, n = x.shape
    stream = _stream_of(x, stream)
    # torch's DEFAULT stream has the handle 0, which the C ABI reads as "the engine's own stream": nothing on the device orders that
    # stream against torch's work then, so the host does (a wait in front of the engine's calls, prot_engine_sync in front of torch's)
    own = stream == 0 and getattr(x, "is_cuda", False)

    def torch_done():
        if own:
            import torch
            torch.cuda.current_stream(x.device).synchronize()

    def engine_done():
        if own:
            engine.sync()

    torch_done()
    if fresh and world > 1 and hasattr(engine, "sweep_shard_seed_dev"):
        h = 0 if halo is None else halo.shape[1]
        engine.sweep_shard_seed_dev(0 if halo is None else halo.data_ptr(), 0 if halo is None else halo.stride(0), h, x.data_ptr(), x.stride(0), n,
                                    ang[0], ang[1], ang[2], peaks.data_ptr(), flags, max(8, 64 // world), stream)
        engine_done()
        allreduce(peaks)
        torch_done()
    elif world > 1:
        if fresh:
            engine.reset()
            peaks.zero_()
            torch_done()
            if halo is not None:
                prime_history(engine, halo, stream)
        engine.sweep_seed_dev(x.data_ptr(), x.stride(0), n, ang[0], ang[1], ang[2], -1, peaks.data_ptr(), flags, max(8, 64 // world), stream)
        engine_done()
        allreduce(peaks)
        torch_done()
    elif fresh:
        engine.reset()
        peaks.zero_()
        torch_done()
        if halo is not None:
            prime_history(engine, halo, stream)
    engine.sweep_dev(x.data_ptr(), x.stride(0), n, ang[0], ang[1], ang[2], -1, peaks.data_ptr(), flags, stream)
    engine_done()
    if world > 1:
        allreduce(peaks)
    return peaks


def angle_shard(ang_start, ang_end, ang_stride, world, rank):
    """Rank's share of the angles one analyze() call visits (loop bounds of cli/phase-rotate.cc:409-428), every world-th of them:
    (start, end, stride) for prot_sweep_dev, or None when the list has fewer than rank + 1 angles.  BASELINE configs[3] words
    the multi-GPU sweep as "angle-sharded" (SURVEY 8e's first option): every rank then runs the Hilbert pass over the WHOLE file --
    72 % of a pass -- so frame-block shards (plan_shards, sweep_two_phase) are what bench.py and the CLI use; this axis is kept
    for files too short to cut into blocks and as a cross-check (tests/test_gpu_multi.py)."""
    first = ang_start + rank * ang_stride
    if rank > 0 and first >= ang_end:
        return None   # (the loop visits `end` itself only when it starts there)
    return first, ang_end, ang_stride * world


def sweep_angle_sharded(engine, x, peaks, flags, world, rank, stream=None, ang=(0, 360, 1), allreduce=None):
    """One rank's angle-sharded sweep: the whole file (x: [n_chn, n] device tensor), this rank's angles, then ONE MAX all-reduce
    of the table (the entries of the other ranks' angles are zero here).  Bit-identical to a single pass: an angle's maximum
    does not depend on which other angles were asked for."""
    allreduce = allreduce or allreduce_peaks
    sub = angle_shard(ang[0], ang[1], ang[2], world, rank)
    n_chn, n = x.shape
    stream = _stream_of(x, stream)
    own = stream == 0 and getattr(x, "is_cuda", False)   # (torch's default stream: see sweep_two_phase)
    if own:
        import torch
        torch.cuda.current_stream(x.device).synchronize()
    if sub is not None:
        engine.sweep_dev(x.data_ptr(), x.stride(0), n, sub[0], sub[1], sub[2], -1, peaks.data_ptr(), flags, stream)
    if own:
        engine.sync()
    if world > 1:
        allreduce(peaks)
    return peaks


def allreduce_peaks(peaks):
    """In-place MAX all-reduce of a [n_chn, 360] peak table (torch tensor)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(peaks, op=dist.ReduceOp.MAX)
    return peaks


def angle_list(ang_start, ang_end, ang_stride):
    """The angles one analyze() call visits (loop bounds of cli/phase-rotate.cc:409-428)."""
    out, angle = [], ang_start
    while angle <= ang_end:
        out.append(angle)
        angle += ang_stride
        if angle >= ang_end:
            break
    return out


def coarse_candidates(peaks, stride, link_chn=False):
    """Candidate minima after the coarse pass (cli/phase-rotate.cc:815-857).
    peaks: [n_chn, 360] float32.  Returns (mins: {angle: [channels]}, p_min, r_zro)."""
    peaks = np.asarray(peaks, np.float32)
    n_chn = peaks.shape[0]
    allp = peaks.max(axis=0)
    mins = {}
    p_min = np.full(n_chn, np.inf, np.float32)
    r_zro = peaks[:, 0].copy()
    for c in range(n_chn):
        row = allp if link_chn else peaks[c]
        sel = row[0:MAXSAMPLE:stride]
        c_min, c_max = np.float32(sel.min()), np.float32(sel.max())
        rng = np.float32(c_max - c_min)
        if rng == 0:
            mins.setdefault(0, []).append(c)
            continue
        if stride > 1:
            rng = np.float32(rng * np.float32(.07))
        else:
            rng = np.float32(0)
            p_min[c] = c_min
        for a in range(0, MAXSAMPLE, stride):
            if row[a] <= np.float32(c_min + rng):
                mins.setdefault(a, []).append(c)
    return dict(sorted(mins.items())), p_min, r_zro


def finish_selection(min_angle, p_min):
    """Phase-distance unwrapping of the per-channel minima (cli/phase-rotate.cc:905-929)."""
    min_angle = [int(a) for a in min_angle]
    found = [bool(p != np.inf) for p in p_min]
    cnt = sum(found)
    avg_rotate = np.float32(sum(a for a, f in zip(min_angle, found) if f)) / np.float32(cnt) if cnt else np.float32(np.nan)
    avg_dist = np.float32(MAXSAMPLE) / np.float32(cnt) if cnt else np.float32(np.inf)
    out = []
    for a, f in zip(min_angle, found):
        if not f:
            out.append(0)
            continue
        if a > 90 * SUBSAMPLE and abs(np.float32(a) - avg_rotate) > avg_dist:
            a -= MAXSAMPLE
        elif avg_rotate > 90 * SUBSAMPLE:
            a -= MAXSAMPLE
        out.append(a)
    return out, found
