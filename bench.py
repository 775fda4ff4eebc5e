#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X phase-rotation hot path.

Metric (BASELINE.json): mono-sample frames/s through the 4096-pt OLA
phase-rotate, % of the HBM roofline at 1 GPU.

Workload (BASELINE.json configs[1], "LV2 mono run(), 4096-pt OLA, 48 kHz"):
the LV2 48 kHz filter (3072-tap windowed Hilbert FIR, latency 1792) on 4096-point
overlap-add FFTs in polyphase form (the FIR's 1536 odd taps applied to the stream read
as complex samples; 5120 new frames per transform pair), over a device-resident mono
stream of 2^27 frames per GPU (512 MiB in + 512 MiB out, beyond the 256 MiB Infinity Cache).  One "step" = one pass of the hot path over that
stream = one kernel launch.  Inputs are resident in HBM before the timed
region; outputs stay in HBM.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by torch.distributed.run, one rank per GPU: every rank runs an
independent stream (the apply path has no exchange step: weak scaling, no
collective); the 360-angle sweep sub-benchmark shards frame-blocks across ranks
and MAX-all-reduces the peak table over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SWEEP_WARM_PASSES = 44    # untimed sweep passes in front of the timed five (>= 60 ms of continuous work, like the headline's pre-roll)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preroll", type=int, default=200,
                    help="untimed launches of the same step BEFORE the W warmup steps of every timed line: after any idle "
                         "time the launch takes ~150 launches (~50 ms of GPU time) to come down from 0.36 to its steady "
                         "0.28 ms (tools/attic/drift_probe.py; DESIGN.md section 5) -- the clocks, not the kernel.  Disclosed in "
                         "the JSON line as preroll_steps.  0 = cold figures")
    ap.add_argument("--frames-log2", type=int, default=27, help="frames per GPU per step (2^k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--sweep-seconds", type=float, default=3600.0, help="sweep: audio seconds (stereo 96 kHz)")
    ap.add_argument("--cfg3-pipelined-child", action="store_true",
                    help="internal: time config 3's pipelined launches in THIS (fresh) process and print one JSON object")
    return ap.parse_args()


def cfg3_pipelined(pkg, torch, np, local_rank, nbuf=256, launches=400):
    """Config 3's launches on a PROT_CFG_PIPELINE engine's own streams (stream = NULL): consecutive launches overlap, the
    history is handed over on the device; three output buffers rotate (three launches are in flight at most); wall clock
    around the series (host launch calls included), closed by prot_engine_sync."""
    n_chn, n = 2, nbuf * 1024
    dev = torch.device("cuda", local_rank)
    rot = np.array([0.0, -1.0, 0.6, 0.8], np.float32)
    x = torch.rand((n_chn, n), device=dev) - 0.5
    ys = [torch.empty_like(x) for _ in range(3)]
    eng = pkg.Engine(pkg.KIND_LV2, 96000, n_chn, device=local_rank, flags=pkg.CFG_PIPELINE)
    for k in range(200):
        eng.apply_dev(x.data_ptr(), n, n, ys[k % 3].data_ptr(), n, n, rot, 0, 0)
    eng.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(launches):
            eng.apply_dev(x.data_ptr(), n, n, ys[k % 3].data_ptr(), n, n, rot, 0, 0)
        eng.sync()
        ms = (time.perf_counter() - t0) / launches * 1e3
        best = ms if best is None else min(best, ms)
    eng.close()
    return best


def cpu_baseline(n_target_s=10.0, sweep_target_s=8.0):
    """The reference-equivalent CPU path timed on this host: the oracle's restated engines on the FFT
    library the reference links -- probed in the order libfftw3f.so.3, libmkl_rt.so (MKL's FFTW3 interface), own
    radix-2 FFT; which one ran is reported.  LV2 run(): mono 48 kHz, 1024-frame buffers, ONE thread (the reference
    processes its channels serially in the RT thread, src/phaserotate.c:841-843) -- this is `value`.  CLI sweep
    (-s 1 -f 2048, stereo): one thread per channel like the reference (cli/phase-rotate.cc:437-443) and an
    all-cores variant (the angles split over the host's cores), on a bounded slice.  Reported baselines only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("MKL_NUM_THREADS", "1")  # the reference's FFTs are single-threaded calls
    import ctypes as C
    import threading
    import numpy as np
    import orc  # checker / baseline leg only
    orc.lib.orc_fft_use_external.restype = C.c_int
    orc.lib.orc_fft_backend_name.restype = C.c_char_p
    kind = orc.lib.orc_fft_use_external(1)
    fft_name = orc.lib.orc_fft_backend_name().decode()
    try:
        rng = np.random.default_rng(1)
        e = orc.Lv2Engine(48000, 1)
        probe = (rng.random(1 << 16, dtype=np.float32) - 0.5)[None, :]
        t0 = time.perf_counter()
        e.run_stream(probe, 90.0, 1024)
        dt = time.perf_counter() - t0
        rate = probe.shape[1] / dt
        n = int(min(max(rate * n_target_s, 1 << 17), 1 << 27)) // 1024 * 1024
        x = (rng.random(n, dtype=np.float32) - 0.5)[None, :]
        e = orc.Lv2Engine(48000, 1)
        t0 = time.perf_counter()
        e.run_stream(x, 90.0, 1024)
        dt = time.perf_counter() - t0
        out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
               "fft": fft_name, "fft_probe_order": "libfftw3f.so.3, libmkl_rt.so, own",
               "sample": "oracle/prot_oracle.c orc_lv2_run (restated src/phaserotate.c run()) on %s, LV2 mono 48 kHz, "
                         "%d frames in 1024-frame buffers, %.1f s" % (fft_name, n, dt),
               "host_cpus": os.cpu_count()}

        # CLI sweep, stereo 96 kHz material, all 360 angles: at blksiz 16384 (the reference's default for the 96 kHz file config 4
        # names -- the bench's lead sweep line) and at -f 2048 (sweep.blksiz_2048), so that each line has its baseline beside it
        def sweep(P, ch, a0, a1):
            o = orc.CliEngine(P, 1)
            o.analyze_file(ch, a0, a1, 1)

        def timed(jobs):
            th = [threading.Thread(target=sweep, args=j) for j in jobs]  # ctypes releases the GIL inside the C call
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return time.perf_counter() - t0

        def sweep_at(P, budget_s):
            nf = max(int(2.5 * 96000) // P, 4) * P
            chans = [(rng.random(nf, dtype=np.float32) - 0.5) * 0.9 for _ in range(2)]
            t1 = timed([(P, chans[0], 0, 360), (P, chans[1], 0, 360)])
            scale = max(1, int(budget_s / max(t1, 1e-3)))
            if scale > 1:  # lengthen the slice towards the time budget
                nf = min(nf * scale, 60 * 96000) // P * P
                chans = [(rng.random(nf, dtype=np.float32) - 0.5) * 0.9 for _ in range(2)]
                t1 = timed([(P, chans[0], 0, 360), (P, chans[1], 0, 360)])
            cores = max(2, min(os.cpu_count() or 2, 64) // 2 * 2)
            per = cores // 2  # threads per channel: contiguous angle ranges (each thread repeats the Hilbert pass, ~10 % of the work)
            edges = [round(i * 360 / per) for i in range(per + 1)]
            tall = timed([(P, chans[c], edges[i], edges[i + 1]) for c in range(2) for i in range(per)])
            return {
                "workload": "cli -s 1 -f %d (360 angles), stereo 96 kHz, %.1f s of audio, restated thr_process + calc_rotated_peak on %s"
                            % (P, nf / 96000.0, fft_name),
                "blksiz": P, "seconds_of_audio": nf / 96000.0,
                "per_channel_threads": {"value": 2 * nf / t1, "unit": "channel-frames/s", "threads": 2, "seconds": t1,
                                        "note": "one thread per channel, as cli/phase-rotate.cc:437-443"},
                "all_cores": {"value": 2 * nf / tall, "unit": "channel-frames/s", "threads": cores, "seconds": tall,
                              "note": "angle ranges over the host's cores (not something the reference does)"}}
        out["cli_sweep"] = sweep_at(16384, sweep_target_s / 2)
        out["cli_sweep"]["blksiz_2048"] = sweep_at(2048, sweep_target_s / 2)
        return out
    finally:
        orc.lib.orc_fft_use_external(0)
        assert kind in (0, 1, 2)


COMPACT_LIMIT = 8192   # the driver keeps an 8 KB tail of stdout: the final line must fit (BENCH_r05.json: a 22.5 KB line was not parsed)


def _r(v, sig=5):
    """floats to `sig` significant digits (the line is a summary; bench_detail.json keeps every digit)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float("%.*g" % (sig, v))
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _kernel_name(label):
    """'prot::olasave8k_dyn_kernel<13, 0> (polyphase form, ...)' -> 'olasave8k_dyn_kernel<13, 0>'"""
    name = (label or "").split(" (")[0]
    return name[len("prot::"):] if name.startswith("prot::") else name


def compact_line(out):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` as objects of numbers
    and names, one summary object per sweep block size and per side benchmark -- no prose.  Everything else (workload texts,
    binding_resource, per-shard arrays, by_buffers_per_launch, notes) stays in the full record: bench_detail.json beside this
    file (and under gpurun_out/ when that directory exists) and on stderr.  Pure function of the full record: tests/test_bench_line.py
    runs it on a recorded one."""
    g = out.get
    cfg = g("config") or {}
    rf = g("roofline") or {}
    line = {k: g(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "preroll_steps", "cold_ms_per_step",
                              "first_call_ms", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": "LV2 mono 48 kHz (BASELINE configs[1]): 3072-tap Hilbert FIR, overlap-save blocks, angle 90 deg, "
                                  "2^%d resident frames per GPU per step" % max(int(cfg.get("frames_per_gpu", 1)).bit_length() - 1, 0),
                      **{k: cfg.get(k) for k in ("fftlen", "hop", "firlen", "latency", "frames_per_gpu", "streams_per_gpu", "parallelism")}}
    line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    line["roofline"].update({"kernel": _kernel_name(rf.get("kernel")), "kernel_ms": rf.get("kernel_ms"),
                             "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
                             "frac_of_achievable": rf.get("frac_of_achievable"),
                             "valu_busy": rf.get("valu_busy"), "lds_busy": rf.get("lds_busy"), "active_fraction": rf.get("active_fraction"),
                             "traffic_from": "profiles/traffic_latest.json (separate --pmc passes, not this run)" if rf.get("traffic") else None})
    cb = g("cpu_baseline")
    if cb:
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "fft", "host_cpus")}
        c["sample"] = (cb.get("sample") or "")[:160]
        cs = cb.get("cli_sweep")
        if cs:
            def one(d):
                return {"blksiz": d.get("blksiz"), "seconds_of_audio": d.get("seconds_of_audio"), "unit": "channel-frames/s",
                        "per_channel_threads": (d.get("per_channel_threads") or {}).get("value"),
                        "all_cores": (d.get("all_cores") or {}).get("value"), "all_cores_threads": (d.get("all_cores") or {}).get("threads")}
            c["cli_sweep"] = one(cs)
            if cs.get("blksiz_2048"):
                c["cli_sweep_2048"] = one(cs["blksiz_2048"])
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    oc = g("oracle_check")
    if oc:
        line["oracle_check"] = {k: oc.get(k) for k in ("ok", "max_abs_diff", "tolerance", "windows")}
    sw = g("sweep")
    if sw:
        def sweep_one(d):
            p8 = d.get("predicted_8gpu") or {}
            nz = d.get("noise_worst_case") or {}
            chk = d.get("oracle_check_10s") or {}
            hi = d.get("host_inclusive") or {}
            o = {"ms": d.get("ms"), "value": d.get("value"), "cold_ms": d.get("cold_ms"), "unpruned_ms": d.get("unpruned_ms"),
                 "frac": (d.get("roofline") or {}).get("frac"),
                 "traffic_ratio": (d.get("roofline") or {}).get("traffic_ratio_to_algorithmic"),
                 "evaluated_fraction": d.get("evaluated_fraction"), "table_identical_unpruned": d.get("table_identical_unpruned"),
                 "argmin_equals_oracle_10s": chk.get("identical"), "min_angle_index": d.get("min_angle_index"),
                 "predicted_speedup": p8.get("predicted_speedup"), "slowest_shard_ms": max(p8["per_shard_ms"]) if p8.get("per_shard_ms") else None,
                 "merged_table_identical": p8.get("merged_table_identical_single_pass")}
            if nz:
                o["noise_ms"] = nz.get("ms")
                o["noise_predicted_speedup"] = (nz.get("predicted_8gpu") or {}).get("predicted_speedup")
            if hi:
                o["host_inclusive_ms"] = hi.get("ms")
            if d.get("per_rank_device_ms") and len(d["per_rank_device_ms"]) > 1:
                o["per_rank_device_ms"] = d["per_rank_device_ms"]
            return o
        line["sweep"] = {"metric": "360-angle min-peak sweep, 60 min stereo 96 kHz (BASELINE configs[3]), resident", "unit": "channel-frames/s",
                         "scaling": "strong", "n_gpus": sw.get("n_gpus"), "seconds_of_audio": sw.get("seconds_of_audio"),
                         "blksiz_16384": sweep_one(sw)}
        for k in ("blksiz_2048", "blksiz_8192"):
            if sw.get(k):
                line["sweep"][k] = sweep_one(sw[k])
        line["sweep"]["prediction"] = ("predicted_speedup = single pass / slowest of eight same-device shard passes; no N > 1 run exists; "
                                       "the two all-reduces are not in it")
    side = {}
    for k in ("block4096", "static_runs", "direct_form", "single_fft_per_hop", "cli_8192", "also"):
        if g(k):
            side["cli_2048" if k == "also" else k] = {"kernel_ms": g(k).get("kernel_ms"), "frac": g(k).get("roofline_frac")}
    if side:
        line["variants"] = side
    if g("fp16"):
        line["fp16"] = {k: g("fp16").get(k) for k in ("value", "kernel", "kernel_ms", "roofline_frac", "vector_pipe_kernel_ms", "vector_pipe_roofline_frac",
                                                        "max_abs_err_vs_fp32", "rms_err_vs_fp32")}
    c3 = g("lv2_stereo_batched")
    if c3:
        line["lv2_stereo_batched"] = {k: c3.get(k) for k in ("ms_per_launch", "roofline_frac", "pipelined_ms_per_launch",
                                                               "pipelined_ms_per_launch_this_process")}
    rh = g("lv2_run_host")
    if rh:
        it = rh.get("instances_in_turn") or {}
        line["lv2_run_host"] = {"us_per_run": rh.get("us_per_run"), "worst_single_call_us": rh.get("worst_single_call_us"),
                                "launch_route_us_per_run": rh.get("launch_route_us_per_run"),
                                "instances_in_turn": {k: {"us_per_run": v.get("us_per_run"), "worst_single_call_us": v.get("worst_single_call_us")}
                                                      for k, v in it.items() if isinstance(v, dict)}}
        for k in ("rt_mode",):
            if rh.get(k):
                line["lv2_run_host"][k] = rh[k]
    fh = g("cli_file_host")
    if fh:
        line["cli_file_host"] = {k: fh.get(k) for k in ("apply_file_ms", "analyze_file_ms")}
    line["detail"] = "bench_detail.json (full record: workload texts, per-shard times, notes); also on stderr"
    line = _r(line)
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= COMPACT_LIMIT:   # never print a line the driver cannot read: drop the side objects first, the contract's keys last
        for k in ("cli_file_host", "variants", "lv2_run_host", "lv2_stereo_batched", "fp16", "oracle_check", "sweep"):
            line.pop(k, None)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt) < COMPACT_LIMIT:
                break
    return txt


def emit(out):
    """full record -> bench_detail.json (+ gpurun_out/) and stderr; the compact line -> the LAST line of stdout"""
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    print(full, file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # PROT_BENCH_DEBUG_ONE_GPU=1: exercise the multi-rank code path on a single-GPU box (all ranks on
    # device 0, gloo instead of RCCL).  Debug only -- its numbers mean nothing.
    one_gpu_debug = os.environ.get("PROT_BENCH_DEBUG_ONE_GPU") == "1"
    if one_gpu_debug:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_debug:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from __graft_entry__ import load_package
    pkg = load_package()
    if args.cfg3_pipelined_child:
        print(json.dumps({"pipelined_ms_per_launch": cfg3_pipelined(pkg, torch, np, local_rank),
                          "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
        return

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # launches go to a non-default torch stream so that torch events bracket them
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream

    n = 1 << args.frames_log2
    gen = torch.Generator(device=dev).manual_seed(0x9E3779B9 ^ rank)
    x = torch.rand(n, generator=gen, device=dev, dtype=torch.float32) - 0.5  # uniform [-0.5, 0.5)
    y = torch.empty_like(x)
    th = -2.0 * np.pi * 90.0 / 360.0
    rot = np.array([np.cos(th), np.sin(th)], np.float32)

    def timed(engine, steps, warmup, n_frames):
        def step():
            engine.apply_dev(x.data_ptr(), n_frames, n_frames, y.data_ptr(), n_frames, n_frames, rot,
                             pkg.F_STATELESS, st)
        for _ in range(args.preroll):  # disclosed in the JSON line as preroll_steps
            step()
        for _ in range(warmup):
            step()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record(ts)
        for _ in range(steps):
            step()
        ev1.record(ts)
        barrier()
        wall = time.perf_counter() - t0
        kern_ms = ev0.elapsed_time(ev1) / steps  # one kernel per step on this stream
        if world > 1:
            t = torch.tensor([wall], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, kern_ms

    def oracle_check(windows=4, W=16384):
        """The timed launch's own output (y: the last step), windows of it against the CPU oracle's LV2 run() fed the frames
        in front of each window (filter memory 3072 + the oracle's start-up ramp): a fast kernel with wrong results must
        not print a number.  Rank 0 only; the checker, never the thing measured."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        rng = np.random.default_rng(123)
        starts = [0, n - W] + [int(v) for v in rng.integers(1 << 16, n - (1 << 16), size=max(0, windows - 2))]
        worst = 0.0
        for t0 in starts:
            lo = max(0, t0 - 16384)
            xs = x[lo:t0 + W].cpu().numpy()[None, :]
            o = orc.Lv2Engine(48000, 1)
            o.activate()
            yo = o.run_stream(xs, 90.0, 1024)[0]
            skip = 2304 if lo == 0 else 0  # Q10: the instance ramps from 0 deg on its first blocks
            got = y[t0 + skip:t0 + W].cpu().numpy()
            worst = max(worst, float(np.abs(got - yo[t0 - lo + skip:]).max()))
        res = {"windows": len(starts), "frames_per_window": W, "max_abs_diff": worst, "tolerance": 2e-6, "ok": worst < 2e-6}
        if not res["ok"]:
            raise SystemExit("bench: the headline launch's output differs from the oracle: %r" % res)
        return res

    eng = pkg.Engine(pkg.KIND_LV2, 48000, 1, device=local_rank)

    def cold_figures(engine, steps):
        """What a caller sees without this file's pre-roll, measured FIRST in the process: the very first
        launch (code object load, first-touch of the output pages), then `steps` launches timed by HIP events with nothing in
        front of them but that one call.  Reported beside the headline; never `value`."""
        def step():
            engine.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        first = (time.perf_counter() - t0) * 1e3
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(ts)
        for _ in range(steps):
            step()
        ev1.record(ts)
        torch.cuda.synchronize()
        return first, ev0.elapsed_time(ev1) / steps

    first_call_ms, cold_ms = cold_figures(eng, args.steps)
    wall, kern_ms = timed(eng, args.steps, args.warmup, n)
    head_check = oracle_check() if rank == 0 else None
    value = world * n * args.steps / wall
    bytes_per_launch = 8.0 * n  # 4 B read + 4 B written per frame
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    rows = int(eng.info.hop) // 512
    is8k = int(eng.info.form) == pkg.FORM_POLY_SAVE8K
    if int(eng.info.form) == pkg.FORM_POLY_SAVE8K:
        rows8 = 16 - int(eng.info.firlen) // 1024
        kname = "olasave8k_dyn_kernel<%d, 0>" % rows8
        klabel = ("prot::%s (polyphase form, overlap-save on 8192-POINT blocks: 512-thread workgroups, two per CU, %d of 16 sample rows new per block, "
                  "table-free centred filter, the radix-2 layer taken by the readers of the LDS exchanges; generated gfx950 assembly, 120 "
                  "hand-allocated VGPRs; round 5: the resident workgroups CLAIM their blocks one at a time -- eight regions, one per XCD, a scalar "
                  "atomic per block -- instead of walking runs planned by the host; launches shorter than 2^24 channel-frames run olasave_kernel "
                  "on 4096-point blocks)" % (kname, rows8))
    elif int(eng.info.form) == pkg.FORM_POLY_SAVE:
        kname = "olasave_kernel<%d, 0>" % rows
        klabel = ("prot::%s (polyphase form, overlap-save: one partition of the odd taps applied centred and table-free -- forward transform at "
                  "the half-shifted index --, %d new sample rows of 16 per block, steady rotation as kernel arguments, dry path from the kept "
                  "input rows; the block loop is generated gfx950 assembly, 120 hand-allocated VGPRs, %d workgroups per CU)"
                  % (kname, rows, int(eng.info.wg_per_cu)))
    else:
        kname = "olapoly_kernel<prot::T32, 1, %d, 0, 3, false, 3>" % rows
        klabel = ("prot::%s (polyphase form, overlap-add: one partition of the odd taps, %d new sample rows per block, steady rotation, dry "
                  "path from registers, packed-fp32 butterflies, %d workgroups per CU)" % (kname, rows, int(eng.info.wg_per_cu)))
    # HBM bytes and SQ counters come from separate rocprofv3 --pmc passes (tools/profile_round.sh, builder-side run on another box of
    # the pool; summaries committed under profiles/): attached only when they are of THIS kernel, and labelled as not of this run
    traffic, pmc = None, {}
    for fn, dst in (("traffic_latest.json", "traffic"), ("pmc_latest.json", "pmc")):
        fp = os.path.join(ROOT, "profiles", fn)
        if os.path.exists(fp) and args.frames_log2 == 27:
            try:
                tj = json.load(open(fp))
                if kname in tj.get("kernel", ""):
                    if dst == "traffic":
                        traffic = tj.get("hbm_bytes_per_launch")
                    else:
                        pmc = {k: tj[k] for k in ("valu_busy", "lds_busy", "active_fraction", "clock_ghz_under_profile") if k in tj}
            except Exception:
                pass
    ACHIEVABLE_GBS = 6290.0  # MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy (this pool: 6.14-6.18, tools/ubench/copy_geom.hip)
    out = {
        "metric": "mono-sample frames/s through 4096-pt OLA phase-rotate",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "preroll_steps": args.preroll,
        "cold_ms_per_step": cold_ms, "cold_roofline_frac": 8.0 * n / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "first_call_ms": first_call_ms,
        "cold_note": "cold_ms_per_step: the first %d launches after the process's very first one, no pre-roll (HIP events); "
                     "first_call_ms: that very first launch, wall clock; `value` is the steady state after preroll_steps + warmup "
                     "untimed launches (clocks settle over ~150 launches: DESIGN.md section 5)" % args.steps,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("LV2 mono 48 kHz: 3072-tap Hilbert FIR (latency 1792) -- the filter of the reference's 4096-pt OLA --, in "
                                "polyphase form (1536 odd taps on the stream read as complex samples) " +
                                ("on 8192-point overlap-save blocks (13312 new frames per block), " if is8k else
                                 "on 4096-point blocks (5120 new frames per block), ") +
                                "angle 90 deg, device-resident stream of 2^%d frames per GPU per step" % args.frames_log2),
                   "partitions": int(eng.info.partitions),
                   # (the kernel that RAN: 8192-point blocks of which 16 - firlen / 1024 rows of 1024 frames are new; the engine's short-launch
                   # form -- prot_info.fftlen / hop: 4096-point blocks, 5120 new frames -- is what run() buffers take)
                   "fftlen": 8192 if is8k else int(eng.info.fftlen),
                   "hop": 1024 * (16 - int(eng.info.firlen) // 1024) if is8k else int(eng.info.hop),
                   "short_launch_fftlen": int(eng.info.fftlen), "short_launch_hop": int(eng.info.hop), "firlen": int(eng.info.firlen),
                   "latency": int(eng.info.latency), "frames_per_gpu": n, "streams_per_gpu": 1,
                   "parallelism": "independent streams x%d" % world},
        "roofline": dict({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "frac_of_achievable": achieved / ACHIEVABLE_GBS,
                          "achievable": ACHIEVABLE_GBS, "traffic": traffic,
                          "source": "achieved / frac: HIP events of this run; traffic, valu_busy, lds_busy, active_fraction: profiles/"
                                    "traffic_latest.json and profiles/pmc_latest.json (rocprofv3 --pmc passes of the same command, "
                                    "builder-side run, NOT this run)",
                          "kernel": klabel, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                          "binding_resource": ("exposed latency inside the workgroup, at a memory system that is two thirds busy: in the steady state "
                                               "of a launch a CU finishes 0.175-0.19 blocks/us (4.75-5.2 TB/s over the chip) with its vector pipes "
                                               "63-69 % busy; of a wave's 19-21 K cycles per block 20-22 % are spent parked at the four workgroup "
                                               "barriers, 17-22 % in the stretch that reads the first exchange and issues the next block's 17 row "
                                               "loads, ~0.3 % waiting for rows (profiles/r05_sv8_stall_table*.json: shader-clock stamps inside the "
                                               "generated loop); the older of a CU's two workgroups takes 7.5 us per block, the younger 17 "
                                               "(oldest-first arbitration) -- with runs planned by the host that left 13 % of the workgroup slots "
                                               "empty at the end of a launch (round 4: 40 blocks/us over the launch against 48.5 in its steady "
                                               "state), with claimed blocks 4 %; what is left is the fill and drain of a pipeline whose stage is "
                                               "12 us long (5 %) and the dependent chains of two 8-wave workgroups per CU (DESIGN.md section 5)" if is8k else
                                               "packed-fp32 VALU issue (4 cycles per v_pk_*_f32, tools/ubench/pk_rate.hip) at the clock the "
                                               "chip holds under this load (1.7 GHz in-kernel, tools/attic/sv_clock.py), with LDS at about half of "
                                               "its cycles: DESIGN.md section 5")}, **pmc),
        "oracle_check": head_check,
    }
    eng.close()

    # The sweep is timed directly behind the headline.  (Found on the last day of round 4 by bisecting the sections: behind the
    # pipelined-LV2 section or the host-file section the sweep passes ran 6-8 % slower -- not a kernel: prot_engine_reset blocked
    # the host once per pass, and that wait costs 0.05-0.25 ms by the state of the runtime's queues once streams have waited on
    # one another.  The reset is stream-ordered now and the order of the sections no longer matters:
    # tools/attic/process_state_probe.py.)
    if not args.no_sweep:
        out["sweep"] = bench_sweep(args, pkg, torch, dist, dev, st, world, rank, local_rank, barrier)

    if not args.no_extra:
        # the metric's literal block size: the same filter on 4096-POINT blocks (PROT_CFG_BLOCK4096: olasave_kernel<10, 0>, four
        # 256-thread workgroups per CU, 5120 new frames per block) -- the headline of round 3 --, and the 8192-point blocks with the
        # runs planned by the host as in round 4 (PROT_CFG_STATIC_RUNS: olasave8k_kernel<13, 0>)
        for key, fl, what in (("block4096", pkg.CFG_BLOCK4096, "4096-point overlap-save blocks (olasave_kernel<10, 0>, 5120 new frames per block, four workgroups per CU)"),
                              ("static_runs", pkg.CFG_STATIC_RUNS, "8192-point blocks, runs planned by the host (olasave8k_kernel<13, 0>: round 4's headline kernel)")):
            engb = pkg.Engine(pkg.KIND_LV2, 48000, 1, device=local_rank, flags=fl)
            wallb, kmsb = timed(engb, args.steps, args.warmup, n)
            out[key] = {"workload": "the headline workload on " + what, "value": world * n * args.steps / wallb, "unit": "frames/s",
                        "roofline_frac": 8.0 * n / (kmsb * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kmsb}
            engb.close()
        # same filter, direct form: two 2048-tap partitions sharing one forward FFT per 2048-frame hop
        eng2 = pkg.Engine(pkg.KIND_LV2, 48000, 1, device=local_rank, flags=pkg.CFG_DIRECT_FORM)
        wall2, kms2 = timed(eng2, args.steps, args.warmup, n)
        out["direct_form"] = {
            "workload": "LV2 mono 48 kHz filter, direct form: two 2048-tap partitions, frequency-domain delay line, 4096 frames "
                        "per transform pair (ola4096p2_kernel, 2 workgroups per CU)",
            "value": world * n * args.steps / wall2, "unit": "frames/s",
            "roofline_frac": 8.0 * n / (kms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kms2}
        eng2.close()
        # CLI default block size at 48 kHz (8192-tap FIR).  Round 4: ONE table-free partition of the 4096 odd taps on 8192-point
        # blocks (olasave8k_kernel<8, 0>: 8 context rows, 8192 new frames per block); rounds 2-3: two 2048-tap partitions with a
        # spectrum table on 4096-point transforms (olapoly_kernel<T32, 2, 8, ...>: 0.356-0.362 ms = 0.37-0.38)
        eng3 = pkg.Engine(pkg.KIND_CLI, 8192, 1, device=local_rank)
        wall3, kms3 = timed(eng3, args.steps, args.warmup, n)
        out["cli_8192"] = {
            "workload": ("CLI engine blksiz 8192 (8192-tap FIR): polyphase, the 4096 odd taps as one table-free partition on 16384-point "
                         "overlap-save blocks (olasave16k_kernel<12, 0>: 12 of 16 rows new)" if int(eng3.info.form) == pkg.FORM_POLY_SAVE16K else
                         "CLI engine blksiz 8192 (8192-tap FIR): polyphase, the 4096 odd taps as one table-free partition on 8192-point "
                         "overlap-save blocks (8 of 16 rows new)" if int(eng3.info.form) == pkg.FORM_POLY_SAVE8K else
                         "CLI engine blksiz 8192 (8192-tap FIR): polyphase, two 2048-tap partitions of the odd taps in one launch"),
            "value": world * n * args.steps / wall3, "unit": "frames/s",
            "roofline_frac": 8.0 * n / (kms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kms3}
        eng3.close()
        # same filter, literal blocking: one zero-padded 4096-pt FFT per 1024-frame hop
        eng1 = pkg.Engine(pkg.KIND_LV2, 48000, 1, device=local_rank, flags=pkg.CFG_SINGLE_FFT_PER_HOP)
        wall1, kms1 = timed(eng1, args.steps, args.warmup, n)
        out["single_fft_per_hop"] = {
            "workload": "LV2 mono 48 kHz filter, one zero-padded 4096-pt FFT per 1024-frame hop (ola4096_kernel<4,...>)",
            "value": world * n * args.steps / wall1, "unit": "frames/s",
            "roofline_frac": 8.0 * n / (kms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kms1}
        eng1.close()
        # the CLI engine A (blksiz 2048: 2048-tap FIR, 4096-pt OLA, hop 2048) on the same stream
        engA = pkg.Engine(pkg.KIND_CLI, 2048, 1, device=local_rank)
        wallA, kmsA = timed(engA, args.steps, args.warmup, n)
        out["also"] = {"workload": "CLI engine blksiz 2048: 2048-tap FIR, 4096-pt OLA, polyphase (1024 odd taps, 6144 new frames per transform pair)",
                       "value": world * n * args.steps / wallA, "unit": "frames/s",
                       "roofline_frac": 8.0 * n / (kmsA * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kmsA}
        engA.close()

    if not args.no_extra:
        # BASELINE config 5: fp16 sample path of the headline workload (4 B per frame)
        engH = pkg.Engine(pkg.KIND_LV2, 48000, 1, device=local_rank)
        xh, yh = x.half(), torch.empty(n, device=dev, dtype=torch.float16)

        def step_h():
            engH.apply_dev_f16(xh.data_ptr(), n, n, yh.data_ptr(), n, n, rot, st)
        for _ in range(args.preroll + args.warmup):  # the same pre-roll as every other timed line
            step_h()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        e0.record(ts)
        for _ in range(args.steps):
            step_h()
        e1.record(ts)
        barrier()
        kmsH = e0.elapsed_time(e1) / args.steps
        y32 = torch.empty_like(x)
        engH.apply_dev(xh.float().data_ptr(), n, n, y32.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
        torch.cuda.synchronize()
        errh = (yh[: 1 << 22].float() - y32[: 1 << 22]).abs()
        # the same launch with the radix-16 passes on the vector pipe (rounds 2-5's kernel: olapoly_kernel<T16>), for the line beside it
        kmsV = None
        try:
            pkg.set_test_option(pkg.TEST_F16_NO_MFMA, 1)
            for _ in range(args.preroll // 2 + args.warmup):
                step_h()
            e0.record(ts)
            for _ in range(args.steps):
                step_h()
            e1.record(ts)
            torch.cuda.synchronize()
            kmsV = e0.elapsed_time(e1) / args.steps
        finally:
            pkg.set_test_option(pkg.TEST_F16_NO_MFMA, 0)
        out["fp16"] = {"workload": "same as config (polyphase overlap-add on 4096-point blocks), fp16 samples (2 B in + 2 B out per frame); round 6: the six radix-16 passes of a "
                                   "transform pair on the MATRIX CORE (prot_f16mfma.hip: v_mfma_f32_32x32x16_f16, fp32 accumulate; twiddles packed half), fp32 overlap-add and rotation",
                       "kernel": "olapolyh_mfma_kernel<10>", "vector_pipe_kernel_ms": kmsV,
                       "vector_pipe_roofline_frac": (4.0 * n / (kmsV * 1e-3) / 1e9 / HBM_PEAK_GBS) if kmsV else None,
                       "value": world * n / (kmsH * 1e-3), "unit": "frames/s", "kernel_ms": kmsH,
                       "roofline_frac": 4.0 * n / (kmsH * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "max_abs_err_vs_fp32": float(errh.max()), "rms_err_vs_fp32": float(torch.sqrt(torch.mean(errh ** 2)))}
        if out["fp16"]["max_abs_err_vs_fp32"] > 0.02:  # half samples of +-0.5 through the filter: 1-2e-3 is the format's noise
            raise SystemExit("bench: the fp16 path's output is wrong: %r" % out["fp16"])
        del xh, yh, y32
        engH.close()

    if not args.no_extra:
        out["lv2_stereo_batched"] = bench_cfg3(args, pkg, torch, np, dev, ts, st, local_rank, barrier, world)
        if rank == 0:
            out["lv2_run_host"] = bench_run_host(pkg, np, local_rank)
            out["cli_file_host"] = bench_file_host(pkg, np, local_rank)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        emit(out)
    if world > 1:
        dist.destroy_process_group()


def bench_cfg3(args, pkg, torch, np, dev, ts, st, local_rank, barrier, world):
    """BASELINE config 3: LV2 stereo, the 96 kHz filter (4096 taps; an 8192-pt FFT in the reference's terms; here the 2048
    odd taps on 4096-pt transforms), 256 buffers of 1024 frames batched per launch (2 x 1 MiB in, 2 x 1 MiB out),
    back-to-back stateful launches on one stream -- and the same with more buffers per launch, which is where the knee
    is: 256 buffers are 64 transform blocks per channel, a fraction of one round of resident workgroups."""
    n_chn, bs = 2, 1024
    rot = np.array([0.0, -1.0, 0.6, 0.8], np.float32)
    rows = []
    for nbuf in (256, 1024, 4096, 16384, 65536):
        n = nbuf * bs
        eng = pkg.Engine(pkg.KIND_LV2, 96000, n_chn, device=local_rank)
        x = torch.rand((n_chn, n), device=dev) - 0.5
        y = torch.empty_like(x)
        launches = 200 if nbuf <= 4096 else 40

        def step():
            eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, 0, st)  # stateful: history carried
        for _ in range(10):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        e0.record(ts)
        for _ in range(launches):
            step()
        e1.record(ts)
        barrier()
        ms = e0.elapsed_time(e1) / launches
        eng.close()
        row = {"buffers_per_launch": nbuf, "ms_per_launch": ms, "value": world * n_chn * n / (ms * 1e-3),
               "roofline_frac": 8.0 * n_chn * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if nbuf == 256:
            # the same launches on a PROT_CFG_PIPELINE engine's own streams.  The HIP runtime multiplexes ALL streams of a
            # process onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams that share a queue do not overlap:
            # measured here as the process is (this bench holds torch's streams beside the engine's), and in a fresh child
            # process with GPU_MAX_HW_QUEUES=8 (the setting a deployment that uses the flag would make; it costs the long
            # launches of the headline ~1 %, which is why this process does not set it)
            row["pipelined_ms_per_launch_this_process"] = cfg3_pipelined(pkg, torch, np, local_rank)
            profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
            if world == 1 and not profiled:  # (no child processes under a profiler)
                import subprocess
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cfg3-pipelined-child"], capture_output=True,
                                       text=True, timeout=300, env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
                    row["pipelined_ms_per_launch"] = json.loads(r.stdout.strip().splitlines()[-1])["pipelined_ms_per_launch"]
                    row["pipelined_env"] = "GPU_MAX_HW_QUEUES=8, own process"
                except Exception as ex:  # the line is informative, never the bench's verdict
                    row["pipelined_error"] = repr(ex)[:200]
            if "pipelined_ms_per_launch" not in row:
                row["pipelined_ms_per_launch"] = row["pipelined_ms_per_launch_this_process"]
                row["pipelined_env"] = "this process, default hardware queues"
            row["pipelined_roofline_frac"] = 8.0 * n_chn * n / (row["pipelined_ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        rows.append(row)
        del x, y
    r0 = rows[0]
    return {"workload": "LV2 stereo, 96 kHz filter (4096 taps: one partition of the 2048 odd taps, overlap-save: one transform pair per "
                        "4096-frame block, no halo), 256 x 1024-frame buffers per channel per launch, stateful back-to-back launches; "
                        "value / ms_per_launch: one stream (the caller's), each launch behind the one before; pipelined_*: the same "
                        "calls on a PROT_CFG_PIPELINE engine's own three streams, history handed over on the device (wall clock "
                        "including the host's launch calls)",
            "value": r0["value"], "unit": "channel-frames/s", "ms_per_launch": r0["ms_per_launch"],
            "roofline_frac": r0["roofline_frac"],
            "pipelined_ms_per_launch": r0.get("pipelined_ms_per_launch"), "pipelined_env": r0.get("pipelined_env"),
            "pipelined_ms_per_launch_this_process": r0.get("pipelined_ms_per_launch_this_process"),
            "pipelined_value": (world * n_chn * 256 * bs / (r0["pipelined_ms_per_launch"] * 1e-3)) if r0.get("pipelined_ms_per_launch") else None,
            "by_buffers_per_launch": rows}


def bench_run_host(pkg, np, local_rank):
    """BASELINE config 2, literal form: the LV2 run() drop-in on HOST buffers, 1024 frames per call.  Default route (round 4):
    the run() SERVER -- a resident one-workgroup kernel takes the call through a pinned mailbox (doorbell, parameters, the port
    block; no launch, no stream synchronisation); `launch_route_us_per_run` = the route of rounds 1-3 (one OLA launch on the
    zero-copy host block + a stream sync per call).  Latency-bound; never `value`."""
    x = (np.random.default_rng(3).random((1, 1024), dtype=np.float32) - 0.5)
    res = {}
    for key, no_server in (("server", 0), ("launch", 1)):
        pkg.set_test_option(pkg.TEST_LV2_NO_SERVER, no_server)
        g = pkg.Lv2Instance(48000, 1, device=local_rank)
        g.activate()
        for _ in range(50):
            g.run(x, 90.0)
        batches = []
        for _ in range(12):  # (batches of 250 calls through the Python wrapper: the median is its steady state; one batch early in a
            t0 = time.perf_counter()  # process's life holds a single stall of tens of milliseconds that the C loop below never shows)
            for _ in range(250):
                g.run(x, 90.0)
            batches.append((time.perf_counter() - t0) / 250)
        res[key] = float(np.median(batches))
        res[key + "_worst"] = float(max(batches))
        res[key + "_stats"] = g.server_stats()
        runs_c = []
        worst_c = 0.0
        for _ in range(8):   # the same calls looped in C: 8 x 500
            runs_c.append(g.time_lv2_run(1024, 500))
            worst_c = max(worst_c, g.last_worst_call[0])
        res[key + "_c"] = float(np.median(runs_c)) * 1e-6
        res[key + "_c_worst"] = worst_c
        g.close()
    pkg.set_test_option(pkg.TEST_LV2_NO_SERVER, 0)
    # what the harness itself adds per call: Lv2Instance.run() = numpy -> ctypes pointer arrays + one ctypes call + a copy of the output;
    # measured on a call that does nothing on the device (zero frames)
    g = pkg.Lv2Instance(48000, 1, device=local_rank)
    g.activate()
    x0 = x[:, :0]
    t0 = time.perf_counter()
    for _ in range(2000):
        g.run(x0, 90.0)
    harness = (time.perf_counter() - t0) / 2000
    g.close()
    # several plugin instances called in turn (round 5: one resident kernel serves every LV2 engine of the process)
    multi = {}
    try:
        for label, specs in (("two_mono_48k", [(48000, 1), (48000, 1)]), ("mono_48k_and_stereo_96k", [(48000, 1), (96000, 2)]),
                             ("four_mono_48k", [(48000, 1)] * 4)):
            gs = [pkg.Lv2Instance(r, c, device=local_rank) for r, c in specs]
            for g in gs:
                g.activate()
            runs_m, worst_m = [], 0.0
            for _ in range(6):
                m, w = pkg.time_lv2_run_multi(gs, 1024, 400)
                runs_m.append(m)
                worst_m = max(worst_m, w)
            multi[label] = {"us_per_run": float(np.median(runs_m)), "worst_single_call_us": worst_m,
                            "server_calls": [g.server_stats()["calls"] for g in gs], "server_kernels_started": sum(g.server_stats()["starts"] for g in gs)}
            for g in gs:
                g.close()
    except Exception as ex:  # noqa: BLE001
        multi["error"] = repr(ex)
    # deferred run() (round 6, prot_lv2_set_deferred): a call posts its block and returns the block before without waiting; calls
    # 150 us apart (a 64-frame period at 48 kHz is 1333 us), each timed by itself, looped in C -- the blocking route the same way beside it
    rt = None
    try:
        g = pkg.Lv2Instance(48000, 1, device=local_rank)
        g.activate()
        mb, wb = g.time_lv2_run_paced(1024, 2000, 150)
        g.lv2_set_deferred(1024)
        md, wd = g.time_lv2_run_paced(1024, 2000, 150)
        rt = dict(g.lv2_deferred_stats(), us_per_run=md, worst_single_call_us=wd, blocking_us_per_run=mb, blocking_worst_single_call_us=wb,
                  extra_latency_frames=1024, calls=2000, gap_us=150)
        g.close()
    except Exception as ex:  # noqa: BLE001
        rt = {"error": repr(ex)[:200]}
    dt = res["server_c"]
    return {"rt_mode": rt, "instances_in_turn": dict(multi, note="1024-frame run() calls on each instance in turn, looped in C (prot_internal_time_lv2_run_multi); us_per_run is per "
                                                   "CALL; every instance is a set of slots of the one resident kernel"),
            "workload": "prot_lv2_run, mono 48 kHz, 1024-frame host buffers, through the run() server (resident kernel; mailbox in device memory "
                        "that the CPU stores into through the PCIe BAR, doorbell, done word in pinned host memory: no launch and no stream sync "
                        "per call), called in a loop from C (prot_internal_time_lv2_run: what an LV2 host's audio thread sees)",
            "us_per_run": dt * 1e6, "frames_per_s": 1024 / dt, "realtime_factor_48k": 1024 / dt / 48000.0,
            "through_python_harness_us_per_run": res["server"] * 1e6,
            "worst_single_call_us": res["server_c_worst"], "launch_route_worst_single_call_us": res["launch_c_worst"],
            "python_harness_worst_batch_us_per_run": res["server_worst"] * 1e6,
            "harness_us_per_call": harness * 1e6,
            "launch_route_us_per_run": res["launch_c"] * 1e6, "launch_route_through_python_harness_us_per_run": res["launch"] * 1e6,
            "note": "us_per_run / launch_route_us_per_run: median of 8 x 500 calls looped in C (worst_single_call_us: the slowest of those 4000 calls); through_python_harness_*: the same calls through "
                    "this harness's numpy / ctypes wrapper (harness_us_per_call of its own; what rounds 1-3 reported: 38-40 us)",
            "server_calls": res["server_stats"]["calls"],
            "server_kernels_started": res["server_stats"]["starts"]}


def bench_file_host(pkg, np, local_rank):
    """The CLI's whole-file calls on HOST buffers (pageable numpy memory, as a file reader hands them
    over): prot_cli_apply_file = upload + de-interleave + OLA + interleave + download, and
    prot_cli_analyze_file = upload + de-interleave + Hilbert pass + 15-angle coarse sweep + reading the
    peak table back.  Host-link-inclusive; never `value`.  Bounded: 2^24 stereo frames (128 MiB each way).
    The output array is allocated and touched BEFORE the timed region: a fresh np.zeros per call costs
    12 ms of page faults here, 3x the call itself (tools/attic/hostlink_probe.py)."""
    n, nc, P = 1 << 24, 2, 8192
    x = (np.random.default_rng(5).random(n * nc, dtype=np.float32) - 0.5)
    out = np.ones_like(x)
    pr = pkg.PhaseRotate(P, nc, device=local_rank)
    pr.apply_file(x[: P * nc * 4], [90, -90])  # allocate staging once

    def med(f, k=5):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[k // 2]

    def ana():
        pr.reset()
        pr.analyze_file(x, 0, 360, 24)
        pr.peak_all(0)
    ta = med(lambda: pr.apply_file(x, [90, -90], out=out))
    tn = med(ana)
    pr.close()
    return {"workload": "CLI blksiz 8192, stereo, 2^24 frames from/to pageable host memory (median of 5, output "
                        "buffer pre-touched)",
            "apply_file_ms": ta * 1e3, "apply_file_channel_frames_per_s": n * nc / ta,
            "apply_file_host_GBps_each_way": n * nc * 4 / ta / 1e9,
            "analyze_file_ms": tn * 1e3, "analyze_file_channel_frames_per_s": n * nc / tn}


def _sweep_signal(torch, np, dev, n_chn, f0, n, seed, kind):
    """Frames [f0, f0 + n) of the synthetic cfg4 material, generated on the device: "harmonic" = an
    asymmetric 32-harmonic stack (110 Hz and 137.5 Hz fundamentals, 0.3 k phase offsets: a real minimum-peak angle)
    under a slow swell + 2 % uniform noise; "noise" = 0.9 x uniform noise (the case the exact pruning cannot help)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    x = (torch.rand((n_chn, n), generator=g, device=dev) - 0.5)
    if kind == "noise":
        return x * 0.9
    x *= 0.04
    L = 4 * 96000  # four seconds: whole cycles of every harmonic
    tt = np.arange(L, dtype=np.float64) / 96000.0
    idx = (torch.arange(n, device=dev, dtype=torch.int64) + int(f0))
    swell_t = idx.double() / float(3600 * 96000)
    for c in range(n_chn):
        fund = 110.0 + 27.5 * c
        pat = sum(np.cos(2 * np.pi * k * fund * tt + 0.3 * k) / k for k in range(1, 33))
        pat = (0.55 / np.abs(pat).max()) * pat
        pat_d = torch.from_numpy(pat.astype(np.float32)).to(dev)
        x[c] += pat_d[idx % L] * (0.75 + 0.25 * torch.sin(2 * np.pi * (3 + c) * swell_t)).float()
    return x


def bench_sweep(args, pkg, torch, dist, dev, st, world, rank, local_rank, barrier):
    """BASELINE config 4 at the reference's default block size for 96 kHz material, 16384 (cli/phase-rotate.cc:749-755: samplerate / 8
    rounded up to a power of two), at -f 2048 (the block size of the headline engine) and at 8192 (the default at 44.1 / 48 kHz)."""
    # (round 5, VERDICT r4 item 3: the LEAD line is the block size the reference itself picks for the 96 kHz file the config names,
    # 16384; -f 2048, the block size of the headline engine and the lead line of rounds 1-4, stands beside it with its own noise and
    # host-inclusive figures)
    main_line = _bench_sweep_at(16384, True, args, pkg, torch, dist, dev, st, world, rank, local_rank, barrier)
    main_line["blksiz_2048"] = _bench_sweep_at(2048, True, args, pkg, torch, dist, dev, st, world, rank, local_rank, barrier)
    # ... and at the reference's default for 44.1 / 48 kHz material, 8192 (the same 96 kHz synthetic material: only the engine differs)
    main_line["blksiz_8192"] = _bench_sweep_at(8192, False, args, pkg, torch, dist, dev, st, world, rank, local_rank, barrier)
    # the sweep against ITS roofline (4 B read per channel-frame): achieved from this run's wall clock, traffic = HBM
    # bytes per channel-frame by counter (tools/sweep_traffic.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, builder-side run
    # on 600 s of the same material, summary committed as profiles/sweep_traffic_latest.json; NOT of this run)
    tr = {}
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "sweep_traffic_latest.json")))
    except Exception:
        pass
    cf = main_line["seconds_of_audio"] * 96000.0 * main_line["channels"]
    for line, key in ((main_line, "harmonic_16384"), (main_line["noise_worst_case"], "noise_16384"), (main_line["blksiz_2048"], "harmonic_2048"),
                      (main_line["blksiz_2048"]["noise_worst_case"], "noise_2048"), (main_line["blksiz_8192"], "harmonic_8192")):
        t = tr.get(key, {})
        ach = 4.0 * cf / (line["ms"] * 1e-3) / 1e9
        line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_channel_frame": 4.0,
                            "traffic_bytes_per_channel_frame": t.get("hbm_bytes_per_channel_frame"),
                            "traffic": (t.get("hbm_bytes_per_channel_frame") * cf) if t.get("hbm_bytes_per_channel_frame") else None,
                            "traffic_ratio_to_algorithmic": t.get("ratio_to_algorithmic"),
                            "source": "achieved: this run's pass; traffic: profiles/sweep_traffic_latest.json (counter passes, builder-side run)"}
    return main_line


def _bench_sweep_at(P, full, args, pkg, torch, dist, dev, st, world, rank, local_rank, barrier):
    """BASELINE config 4: 360-angle (-s 1) min-peak sweep of stereo 96 kHz audio, blksiz 2048 (-f 2048, 4096-pt),
    frame-blocks sharded across the ranks with a history halo, one RCCL MAX all-reduce of the [2][360] peak table.
    Strong scaling: the total amount of audio is fixed.  `value` = resident pass on the harmonic material; the noise
    material (worst case for the pruning) and the host-inclusive pass (shard uploaded from pinned host memory inside
    the timed region, double-buffered against the sweep: SURVEY 8d's multi-GPU wall) are reported beside it."""
    import numpy as np
    from phaserotate_lv2_amd import sweep_dist
    n_chn = 2
    total = int(args.sweep_seconds * 96000) // P * P
    eng = pkg.Engine(pkg.KIND_CLI, P, n_chn, device=local_rank)
    # shard starts on whole device transform blocks: the merged table is bit-identical to a single pass
    plan = sweep_dist.plan_shards(total, P, world, align=eng.shard_align_blocks())
    b0, b1 = plan[rank]
    f0, n = b0 * P, (b1 - b0) * P
    H = int(eng.info.history)
    peaks = torch.zeros((n_chn, 360), device=dev)
    flags = (pkg.F_STREAM_START if rank == 0 else 0) | (pkg.F_STREAM_END if rank == world - 1 else 0)
    res = {}

    def gmax(v):
        if world > 1:
            t = torch.tensor([v], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    for kind in (("harmonic", "noise") if full else ("harmonic",)):
        x = _sweep_signal(torch, np, dev, n_chn, f0, n, 1234, kind) if n > 0 else torch.zeros((n_chn, 0), device=dev)
        # (the halo only primes the history in front of the shard; for this timing any material will do)
        halo = _sweep_signal(torch, np, dev, n_chn, f0 - H, H, 4321, kind).contiguous() if rank > 0 else None

        def one_pass():
            if world > 1:
                # two phases (round 5): every rank seeds the pruning bound from a sparse sample of its shard, the sampled tables are
                # merged, and the main pass lists against the GLOBAL bound (a shard alone lists twice what the whole file lists).
                # Phase one is ONE call: reset, table zeroed, the frames in front of the shard into the history (a copy), seeding pass
                eng.sweep_shard_seed_dev(halo.data_ptr() if rank > 0 else 0, H, H if rank > 0 else 0, x.data_ptr(), n, n, 0, 360, 1,
                                         peaks.data_ptr(), flags, max(8, 64 // world), st)
                dist.all_reduce(peaks, op=dist.ReduceOp.MAX)
            else:
                eng.reset()
                peaks.zero_()
            eng.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, peaks.data_ptr(), flags, st)
            if world > 1:
                dist.all_reduce(peaks, op=dist.ReduceOp.MAX)

        def timed_passes(reps=3):
            one_pass()
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(torch.cuda.current_stream())
            for _ in range(reps):
                one_pass()
            e1.record(torch.cuda.current_stream())
            barrier()
            return gmax((time.perf_counter() - t0) / reps), e0.elapsed_time(e1) / reps

        # `cold`: the first three passes after one untimed pass (what rounds 1-3 reported); `ms` / `value`: five passes after 40
        # more untimed ones -- the same steady state the headline is quoted at (its pre-roll): the clocks settle over tens of
        # milliseconds of continuous work, a 2 ms pass right after an idle period runs ~20 % slower
        cold_wall, cold_dev_ms = timed_passes()
        for _ in range(SWEEP_WARM_PASSES - 4):
            one_pass()
        wall, dev_ms = timed_passes(5)
        pk = peaks.cpu().numpy()
        ev, seen = eng.sweep_stats()
        exs = eng.sweep_stats_ex() if hasattr(eng, "sweep_stats_ex") else {}
        fus = eng.sweep_stats_fused() if hasattr(eng, "sweep_stats_fused") else {"tested": 0, "listed": 0}
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1)
        wall_np, _ = timed_passes()
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0)
        same = bool((peaks.cpu().numpy() == pk).all())
        per_rank = [dev_ms]
        if world > 1:
            t = torch.zeros(world, device=dev, dtype=torch.float64)
            t[rank] = dev_ms
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            per_rank = [float(v) for v in t.cpu()]
        n_ang = 359  # (angle 0 is the plain abs-max of the input, Q4)
        res[kind] = {"ms": wall * 1e3, "value": total * n_chn / wall, "unpruned_ms": wall_np * 1e3,
                     "cold_ms": cold_wall * 1e3, "warmup_passes": SWEEP_WARM_PASSES,
                     # (of the tiles the matrix-core kernel was handed: on the fused route those are the LISTED frames' tiles)
                     "evaluated_fraction": ev / max(seen, 1), "table_identical_unpruned": same,
                     "untested_fraction": exs.get("untested", 0) / max(seen, 1),
                     # SURVEY 8d, "for completeness": the sweep reads 4 B per channel-frame (its Hilbert branch is written and
                     # re-read on the device: 8 B more) and the UNPRUNED evaluation is 359 angles x (mul + fma) on the matrix
                     # core: fraction of the 157 TFLOP/s fp32 matrix peak the unpruned pass reaches, and the HBM fraction of
                     # the pruned pass at its algorithmic 4 B per channel-frame
                     "hbm_frac_algorithmic": total * n_chn * 4.0 / wall / 1e9 / HBM_PEAK_GBS,
                     # the FUSED route (blksiz 2048 / 4096, round 4): the Hilbert pass tests every frame against the call's bound and
                     # writes only the listed (dry, hilbert) pairs -- bytes per channel-frame by construction: 4 read + 16 x listed
                     # fraction (8 written, 8 read back by the matrix-core kernel); the separate passes move 16 (blksiz 16384: 28)
                     "fused": bool(fus["tested"]),
                     "listed_fraction": (fus["listed"] / fus["tested"]) if fus["tested"] else None,
                     "bytes_per_channel_frame_by_construction": (4.0 + 16.0 * fus["listed"] / fus["tested"]) if fus["tested"] else (28.0 if P >= 16384 else 16.0),
                     "unpruned_frac_of_fp32_matrix_peak": total * n_chn * n_ang * 4.0 / wall_np / 157.3e12,
                     "per_rank_device_ms": per_rank,
                     "min_angle_index": [int(pk[c].argmin()) for c in range(n_chn)]}
        if world == 1:
            # What eight GPUs would do, predicted on one: the eight frame-block shards of the same material (plan_shards: whole
            # device transform blocks, history primed in front of each) timed one after the other on this device.  The pruning
            # makes the sweep's time depend on the material, so the shards are not equal; predicted_speedup = single pass /
            # slowest shard.  A PREDICTION (no collective, no host link, one device's clocks), never `value`.
            plan8 = sweep_dist.plan_shards(total, P, 8, align=eng.shard_align_blocks())
            # phase one of all eight shards first (untimed here): the merged table every shard would hold after the first all-reduce
            seeds = []
            for r8, (c0, c1) in enumerate(plan8):
                xs8 = x[:, c0 * P:c1 * P]
                fl8 = (pkg.F_STREAM_START if r8 == 0 else 0) | (pkg.F_STREAM_END if r8 == 7 else 0)
                eng.reset()
                peaks.zero_()
                if r8 > 0:   # (the seeding pass reads the shard's first block too: the history in front of it must be the file's, as in the main pass)
                    h8 = x[:, c0 * P - H:c0 * P].contiguous()
                    eng.prime_dev(h8.data_ptr(), H, H, st)
                eng.sweep_seed_dev(xs8.data_ptr(), x.shape[1], xs8.shape[1], 0, 360, 1, -1, peaks.data_ptr(), fl8, 8, st)
                torch.cuda.synchronize()
                seeds.append(peaks.clone())
            gseed = torch.stack(seeds).amax(0)
            shard_ms, merged = [], torch.zeros_like(peaks)
            for r8, (c0, c1) in enumerate(plan8):
                xs8 = x[:, c0 * P:c1 * P]
                halo8 = x[:, c0 * P - H:c0 * P].contiguous() if r8 > 0 else None   # (the frames really in front of the shard: the tables are compared below)
                fl8 = (pkg.F_STREAM_START if r8 == 0 else 0) | (pkg.F_STREAM_END if r8 == 7 else 0)

                def shard_pass():
                    # what one rank of eight does: reset, prime, its own seeding pass, [all-reduce: here the merged table is copied
                    # in its place], the main pass against the global bound, [all-reduce]
                    eng.reset()
                    peaks.zero_()
                    if r8 > 0:
                        eng.prime_dev(halo8.data_ptr(), H, H, st)
                    eng.sweep_seed_dev(xs8.data_ptr(), x.shape[1], xs8.shape[1], 0, 360, 1, -1, peaks.data_ptr(), fl8, 8, st)
                    peaks.copy_(gseed)
                    eng.sweep_dev(xs8.data_ptr(), x.shape[1], xs8.shape[1], 0, 360, 1, -1, peaks.data_ptr(), fl8, st)
                for _ in range(8):  # (the same steady state as the single pass it is compared with)
                    shard_pass()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
                for _ in range(5):
                    shard_pass()
                e1.record(torch.cuda.current_stream())
                torch.cuda.synchronize()
                shard_ms.append(e0.elapsed_time(e1) / 5)
                merged = torch.maximum(merged, peaks)
            res[kind]["predicted_8gpu"] = {"per_shard_ms": shard_ms, "single_pass_ms": dev_ms,
                                           "predicted_speedup": dev_ms / max(shard_ms),
                                           "two_phase": True,
                                           "merged_table_identical_single_pass": bool((merged.cpu().numpy() == pk).all()),
                                           "note": "PREDICTED from eight same-device shard passes (each: reset, history primed by a copy, its "
                                                   "own seeding pass, the merged seed table copied in where the first all-reduce would write "
                                                   "it, the main pass against that global bound); no collective, no host link; no N > 1 run exists"}
        if kind == "harmonic" and full:
            # host-inclusive pass: the shard comes from pinned host memory inside the timed region, in chunks whose
            # upload overlaps the sweep of the chunk before (two device buffers, a copy stream)
            chunk = eng.shard_align_blocks() * P * 1024
            xh = torch.empty((n_chn, n), dtype=torch.float32, pin_memory=True)
            xh.copy_(x)
            bufs = [torch.empty((n_chn, chunk), device=dev) for _ in range(2)]
            cs = torch.cuda.Stream(device=dev)
            ms_stream = torch.cuda.current_stream()

            def host_pass(h0=0, h1=None, fl_all=None, halo_=None, prime=None, reduce_=True):
                """frames [h0, h1) of this rank's pinned copy (default: all of it) through upload + sweep"""
                h1 = n if h1 is None else h1
                fl_all = flags if fl_all is None else fl_all
                eng.reset()
                peaks.zero_()
                if (rank > 0 if prime is None else prime):
                    sweep_dist.prime_history(eng, halo if halo_ is None else halo_, st)
                evs, done = [None, None], [None, None]
                pos, i = h0, 0
                while pos < h1 or (h1 == h0 and i == 0):
                    m = min(chunk, h1 - pos)
                    b = i & 1
                    with torch.cuda.stream(cs):
                        if done[b] is not None:
                            cs.wait_event(done[b])
                        for c in range(n_chn):  # one contiguous pinned range per channel: a strided 2-D copy takes a slow path
                            bufs[b][c, :m].copy_(xh[c, pos:pos + m], non_blocking=True)
                        evs[b] = cs.record_event()
                    ms_stream.wait_event(evs[b])
                    fl = (fl_all & pkg.F_STREAM_START if pos == h0 else 0) | (fl_all & pkg.F_STREAM_END if pos + m >= h1 else 0)
                    eng.sweep_dev(bufs[b].data_ptr(), chunk, m, 0, 360, 1, -1, peaks.data_ptr(), fl, st)
                    done[b] = ms_stream.record_event()
                    pos += m
                    i += 1
                    if h1 == h0:
                        break
                if world > 1 and reduce_:
                    dist.all_reduce(peaks, op=dist.ReduceOp.MAX)
                return peaks.cpu()  # the reduced table on the host ends the pass

            host_pass()
            barrier()
            t0 = time.perf_counter()
            tab = host_pass()
            barrier()
            hw = gmax(time.perf_counter() - t0)
            res["host_inclusive"] = {"ms": hw * 1e3, "value": total * n_chn / hw, "unit": "channel-frames/s",
                                     "what": "harmonic material; wall from the first H2D copy of each rank's shard (pinned host memory, "
                                             "%d-frame chunks, upload of chunk k+1 under the sweep of chunk k) to the reduced table on the host" % chunk,
                                     "table_identical_resident": bool((tab.numpy() == pk).all()),
                                     "host_GBps_per_rank": n_chn * n * 4 / hw / 1e9}
            if world == 1:
                # SURVEY 8d's multi-GPU wall starts at the first H2D copy: what eight GPUs with a host link EACH would take, predicted
                # on one -- every shard of plan_shards uploaded from pinned host memory and swept (the same chunked, double-buffered
                # pass), one after the other; predicted wall = the slowest shard.  A PREDICTION: one device's link and clocks, no
                # collective (2.9 KB, latency-bound), and a host that really has eight times this link's bandwidth to its memory.
                plan8 = sweep_dist.plan_shards(total, P, 8, align=eng.shard_align_blocks())
                sh = []
                for r8, (c0, c1) in enumerate(plan8):
                    halo8 = _sweep_signal(torch, np, dev, n_chn, c0 * P - H, H, 4321, kind).contiguous() if r8 > 0 else None
                    fl8 = (pkg.F_STREAM_START if r8 == 0 else 0) | (pkg.F_STREAM_END if r8 == 7 else 0)
                    host_pass(c0 * P, c1 * P, fl8, halo8, r8 > 0, False)
                    tms = []
                    for _ in range(3):   # (median of three: a single sample of a 7 ms wall-clock pass strays by milliseconds now and then)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        host_pass(c0 * P, c1 * P, fl8, halo8, r8 > 0, False)
                        tms.append((time.perf_counter() - t0) * 1e3)
                    sh.append(sorted(tms)[1])
                res["host_inclusive"]["predicted_8gpu"] = {"per_shard_ms": sh, "predicted_wall_ms": max(sh), "predicted_speedup": hw * 1e3 / max(sh),
                                                           "note": "PREDICTED from eight same-device host-inclusive shard passes (each with this device's "
                                                                   "host link to itself); no N > 1 run exists"}
            del xh, bufs
        del x

    # parity of the printed quantity on a slice the CPU oracle can do: the first 10 s of the harmonic material
    check = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc  # checker only
        ns = 10 * 96000 // P * P
        xs = _sweep_signal(torch, np, dev, n_chn, 0, ns, 1234, "harmonic")
        e2 = pkg.Engine(pkg.KIND_CLI, P, n_chn, device=local_rank)
        pk2 = torch.zeros((n_chn, 360), device=dev)
        torch.cuda.synchronize()
        e2.sweep_dev(xs.data_ptr(), ns, ns, 0, 360, 1, -1, pk2.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
        e2.sync()
        torch.cuda.synchronize()
        o = orc.CliEngine(P, n_chn)
        o.analyze_file(np.ascontiguousarray(xs.cpu().numpy().T).reshape(-1), 0, 360, 1)
        pg, po = pk2.cpu().numpy(), o.peaks()
        check = {"slice_seconds": ns / 96000.0, "min_angle_index_device": [int(pg[c].argmin()) for c in range(n_chn)],
                 "min_angle_index_oracle": [int(po[c].argmin()) for c in range(n_chn)],
                 "max_table_difference_dB": float(np.abs(20 * np.log10(np.maximum(pg, 1e-30)) - 20 * np.log10(np.maximum(po, 1e-30))).max())}
        check["identical"] = check["min_angle_index_device"] == check["min_angle_index_oracle"]
        if not check["identical"]:
            raise SystemExit("bench sweep: the device's minimum-peak angle differs from the oracle's on the 10 s slice: %r" % check)
        e2.close()
    eng.close()
    h = res["harmonic"]
    engine_txt = (("16384-tap FIR = 8192 odd taps as ONE table-free partition on 16384-point overlap-save blocks (olasave16k_kernel<8, MODE_SWEEP>: "
                   "1024-thread workgroups, one per CU; ONE launch reads the stream once, tests and lists -- rounds 2-4: two accumulating "
                   "olapoly_kernel launches of two 2048-tap partitions each through a workspace in HBM), sweep_eval_kernel evaluates the lists")
                  if P == 16384 else
                  ("8192-tap FIR = 4096 odd taps as ONE table-free partition on 16384-point overlap-save blocks (olasave16k_kernel<12, MODE_SWEEP>, "
                   "12 of 16 rows new; round 4: 8192-point blocks, 8 of 16): the Hilbert pass tests and lists, sweep_eval_kernel evaluates the lists")
                  if P == 8192 else
                  ("2048-tap FIR = 1024 odd taps on 4096-point overlap-save blocks (olasave_kernel<12, MODE_SWEEP>, 12 of 16 rows new): the Hilbert "
                   "pass tests and lists, sweep_eval_kernel evaluates the lists"))
    what = ("the reference's default block size at 88.2 / 96 kHz" if P == 16384 else "the reference's default block size at 44.1 / 48 kHz" if P == 8192 else
            "the block size of the headline engine")
    if not full:
        return {"metric": "channel-frames/s through the 360-angle min-peak sweep (cli -s 1 -f %d: %s)" % (P, what),
                "value": h["value"], "unit": "channel-frames/s", "ms": h["ms"], "cold_ms": h["cold_ms"], "cold_value": h["value"] * h["ms"] / h["cold_ms"],
                "warmup_passes": h["warmup_passes"],
                "unpruned_ms": h["unpruned_ms"],
                "evaluated_fraction": h["evaluated_fraction"], "table_identical_unpruned": h["table_identical_unpruned"],
                "min_angle_index": h["min_angle_index"], "per_rank_device_ms": h["per_rank_device_ms"],
                "predicted_8gpu": h.get("predicted_8gpu"), "oracle_check_10s": check,
                "fused": h["fused"], "listed_fraction": h["listed_fraction"],
                "engine": engine_txt}
    return {"metric": "channel-frames/s through the 360-angle min-peak sweep (cli -s 1 -f %d: %s)" % (P, what),
            "engine": engine_txt, "blksiz": P,
            "signal": "32-harmonic asymmetric stack (110 / 137.5 Hz) under a slow swell + 2 % noise, generated on the device; resident in HBM",
            "value": h["value"], "unit": "channel-frames/s", "ms": h["ms"], "cold_ms": h["cold_ms"], "cold_value": h["value"] * h["ms"] / h["cold_ms"],
                "warmup_passes": h["warmup_passes"],
            "cold_note": "ms / value: five passes after %d untimed ones (the steady state, as the headline's pre-roll: the pass time settles over ~50 ms "
                         "of continuous work, tools/attic/sweep_drift.py); cold_ms: the first " % SWEEP_WARM_PASSES +
                         "three passes after ONE untimed pass (what rounds 1-3 reported as ms)",
            "unpruned_ms": h["unpruned_ms"],
            "evaluated_fraction": h["evaluated_fraction"], "table_identical_unpruned": h["table_identical_unpruned"],
            "min_angle_index": h["min_angle_index"], "per_rank_device_ms": h["per_rank_device_ms"],
            "oracle_check_10s": check,
            "predicted_8gpu": h.get("predicted_8gpu"),
            "noise_worst_case": res["noise"], "host_inclusive": res.get("host_inclusive"),
            "seconds_of_audio": total / 96000.0, "channels": n_chn, "rate": 96000, "scaling": "strong", "n_gpus": world,
            "collective": "all_reduce(MAX) float[2][360] over RCCL" if world > 1 else None}


if __name__ == "__main__":
    main()
