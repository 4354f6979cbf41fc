#!/usr/bin/env python3
"""bench.py -- the null-geodesic hot path on BASELINE.json's headline workload.

A "step" is one pass of the hot path over one frame's worth of rays per GPU: 1024 x 1024 pixels
x 5 samples = 5,242,880 null geodesics (BASELINE.json config 2; camera (1e-4, 0, 30), fov 0.6,
mass 0.5 -> r_s = 1, curve_end 50, directions from the reference's pinhole + MT19937 jitter,
raytracer/RelativisticRenderEngine.py:185-230, seed 42).  Inputs are resident in HBM when the
timed region starts; the timed region is trace + shade/sample-mean and, for N > 1, the single gather of
per-pixel RGBA to rank 0 (asynchronous, overlapping the next frame's trace).  The trace writes WHOLE end records
(exit position and direction, 48 B/ray: what spacetime_ray_cast returns, :307-308 -- the headline since round 5);
--dir-only gives the sky frame's form (exit directions only), reported beside the headline as sky_frame_dir_only.

The timed region -- EXACTLY --steps steps between barrier + synchronise -- is run --reps times (default 5); the headline is the
MEDIAN repetition, every repetition is in the line (ms_per_step_samples) and so is every HIP-event sample of the trace kernel
(roofline.kernel_ms_samples).  Nothing runs on the host between a repetition's barrier and its clock (round 5 built its clock
sampler there and the driver's 28-ms region read 12 % slow: DESIGN.md section 6.1).

roofline.calibration: bhg_peak_probe (a pure v_fma_f64 kernel and one with the step loop's instruction mix, in the trace
kernels' launch geometry) AFTER the timed region (--probe-when), one read of the shader clock per repetition from the main
thread, and a sysfs sampler thread in ONE EXTRA repetition that does not count (its reads slow a region by 5-6 %)
-> frac_of_measured_peak, frac_at_timed_region_clock beside frac (the vendor's 78.6 TFLOP/s).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the headline figures are WEAK scaling --
the frame grows to (1024*nx) x (1024*ny), nx*ny = N, over the same window of directions, so every rank still
traces 5,242,880 rays of the same distribution; 32x32-pixel tiles are dealt to the ranks by cost ranking.  A second
timed region then shards ONE fixed 1024x1024x5 frame over the N ranks (BASELINE.json's metric read as strong
scaling; two frames in flight per rank, the sequential figure beside it) and is reported in the same line as
"strong": {...}.  BHGEO_FORCE_COLLECTIVE=1 makes a single-GPU run take the N > 1 code path (RCCL process group of
one rank, the real asynchronous gather, root-side assembly).

--single-process: the same workloads through the library-owned frame (bhg_frame_*, include/bhgeo.h): ONE process
drives --gpus N devices, no torch.distributed, no PyTorch in the timed path -- what the Blender add-on uses.

roofline.traffic and roofline.valu_insts_per_64_ray_steps are measured in the run itself on one GPU: before this process
touches the GPU it starts three rocprofv3 --pmc child runs of the same command (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU in
separate passes, 3 steps each; live_pmc()); if that fails (or with --live-pmc 0, --lean, --cpu-seconds 0, N > 1) the line
carries the committed profiles/rNN_pmc_summary*.json instead -- roofline.traffic_source says which, and why.

Layout: THIS file holds the command line (parse), the launcher for N > 1 (self_launch), the shard construction
(build_frames), measure() -- ONE timed region, the only place the headline clock runs -- and main(), which assembles the
JSON line.  bench_common.py: constants, Workload (what is traced), Runtime (rank / group / context), Lanes (two frames in
flight), roofline_block.  bench_figures.py: the counters (live_pmc / pmc_traffic), the secondary blocks (time_frame,
strong_predicted, pipelined_figure, host_buffer_figures, cpu_baseline) and the --single-process form.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the pool's host driver supports dmabuf IPC only: without this RCCL between processes fails with hipIpcGetMemHandle:
# invalid argument.  Set before anything initialises the GPU -- also when a launcher, not self_launch(), started this rank.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from bench_common import (BYTES_PER_RAY, BYTES_PER_RAY_DIR, CAM, DISK, EV_EVERY, PEAK_FP64_VALU_TFLOPS, ClockSampler, Lanes, Runtime,  # noqa: F401
                          Workload, emit, grid_for, hbm_copy_probe, merge_clock_samples, roofline_block, run_probes, spread, traced_with_events)
from bench_figures import (counters_for, host_buffer_figures, live_pmc, main_single_process, pipelined_figure,  # noqa: F401
                           pmc_traffic, strong_predicted, time_frame)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5,
                    help="the headline's timed region -- EXACTLY --steps steps between barrier + synchronise -- is run this many times "
                         "back to back; every repetition's ms per step goes into the line (ms_per_step_samples), the headline is "
                         "their MEDIAN (a 20-step region is 28 ms: one number of it cannot tell a slow box from a transient); "
                         "1 = the single region of rounds 1-5")
    ap.add_argument("--probe-when", choices=["after_only", "before_warmup", "before_timed", "off"],
                    default=os.environ.get("BHGEO_PROBE_WHEN", "after_only"),
                    help="where the roofline calibration probes (bhg_peak_probe, power-bound pure-FMA launches, 25 ms) run relative to "
                         "the headline's timed region: after_only (default since round 6: nothing power-hungry in front of a short "
                         "region), before_warmup (round 5: in front of the W warm-up steps, and again afterwards), before_timed, off")
    ap.add_argument("--clock-reads", type=int, default=int(os.environ.get("BHGEO_CLOCK_READS", "1")),
                    help="reads of the shader clock (sysfs hwmon) per headline repetition, from the main thread while the GPU works "
                         "through the enqueued steps; 0 = the headline's repetitions run unobserved (the thread sampler only ever "
                         "runs in the extra repetition behind them)")
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed steps run before the W warm-up steps until this much wall time has passed: the "
                         "GPU's clocks take tens of milliseconds of load to settle (a 20-step timed region right "
                         "after start-up measures 5 %% low); 0 = off")
    ap.add_argument("--regime", choices=["adaptive", "fine", "rk4"], default="adaptive",
                    help="adaptive: DP5(4) rtol 1e-3 atol 1e-6 max_step inf (engine + scipy defaults); "
                         "fine: same with max_step 0.1; rk4: fixed step 0.1")
    ap.add_argument("--rhs", choices=["christoffel", "reduced", "kerr"], default="christoffel",
                    help="kerr = BASELINE.json configs[4]: a/M = 0.9, Boyer-Lindquist Christoffels, same camera")
    ap.add_argument("--workload", choices=["frame", "disk", "orbit"], default="frame",
                    help="frame: BASELINE.json configs[1] (the headline; configs[4] with --rhs kerr); "
                         "disk: configs[2], 1024x1024 + thin disk, a step = the 5 camera inclinations; "
                         "orbit: configs[3], 2048x2048 x16 with a lit sphere orbiting the hole, a step = one "
                         "animation frame (the sphere moves every step); N > 1 shards that one frame (strong scaling)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--dir-only", action="store_true",
                    help="frame workload: have the trace write only the exit directions a sky frame reads (24 B/ray) instead of whole "
                         "end states (x and k, 48 B/ray: spacetime_ray_cast's return values, RelativisticRenderEngine.py:307-308 -- the "
                         "default and the headline since round 5; the direction-only form is reported beside it as sky_frame_dir_only)")
    ap.add_argument("--full-records", action="store_true", help="(the default since round 5; accepted for older scripts)")
    ap.add_argument("--lpt", type=int, default=1, help="1: visit tiles in order of decreasing expected cost")
    ap.add_argument("--visit", choices=["auto", "cost", "row"], default="auto",
                    help="order in which a rank visits its tiles: by decreasing cost (shortens the wave-drain tail: what a "
                         "1/8 shard's 0.2-ms kernel needs), row-major (1 %% faster over a whole frame on one GPU: the cheap "
                         "far-field batches, which claim work every few iterations, stay interleaved with the long rays); "
                         "auto = row on one GPU, cost when the frame is sharded")
    ap.add_argument("--order", choices=["measured", "model", "none"], default="model",
                    help="tile order / dealing: by the attempted steps of one untimed calibration trace of the same frame "
                         "(dist.measured_tile_cost), by the shadow-edge model (plain frame only), or row-major")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline wall time (0 = skip)")
    ap.add_argument("--emulate-shards", type=str, default="2,4,8",
                    help="N = 1, frame workload: also time rank 0's part of a world-N dealing of the SAME fixed frame -- its shard "
                         "traced and shaded into the gather slab, a 1-rank RCCL gather of that slab, the root's assembly of the "
                         "N-rank frame -- for each N listed and report the predicted strong-scaling efficiency T1 / (N T_N) in "
                         "a strong_predicted block; '' = off")
    ap.add_argument("--live-pmc", type=int, default=1,
                    help="1 (default, single GPU, not --lean): measure roofline.traffic and the VALU instruction count NOW, with "
                         "three rocprofv3 --pmc child runs of this same command (3 steps each) started before this process "
                         "touches the GPU; on any failure -- or 0 -- the line replays the committed profiles/ summary")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs: the headline's timed region and nothing after it (no full_records, shard emulation, "
                         "pipelined / host-buffer figures, CPU baseline), so that a profiler's per-kernel averages are the headline's")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives --gpus N devices through the library-owned frame (bhg_frame_*: tile dealing, per-device "
                         "contexts, one gather onto device 0 -- RCCL single-process mode when the devices are distinct); no "
                         "torch.distributed.  BHGEO_DEVICES=0,0 lists the devices explicitly (a repeated index = several contexts "
                         "of one GPU)")
    ap.add_argument("--root-share", type=str, default="auto",
                    help="N > 1: what rank 0 -- which also receives the gather and assembles the frame -- is dealt, as a multiple of "
                         "what every other rank is dealt; auto = from the assembly kernel's and a shard frame's measured times, so "
                         "that all ranks finish together; 1 = equal shares")
    ap.add_argument("--frame-gather", choices=["auto", "copy", "rccl", "peer"], default="auto",
                    help="--single-process: how the devices' pixels reach the first device (bhg_frame_create's gather mode): auto = RCCL "
                         "single-process mode for distinct devices else device-to-device copies; peer = no exchange, every device's "
                         "shade kernel stores straight into the first device's image over xGMI")
    ap.add_argument("--shard", choices=["tiles", "frames", "both"], default="both",
                    help="orbit workload, N > 1: tiles = every frame's tiles over all ranks + one gather per frame (the headline); "
                         "frames = whole frames dealt round-robin to the ranks, no tail, one gather at the end; both = the second "
                         "reported beside the first")
    a = ap.parse_args(argv)
    if a.lean:
        a.cpu_seconds, a.emulate_shards = 0.0, ""
    dw, ds = {"frame": (1024, 5), "disk": (1024, 1), "orbit": (2048, 16)}[a.workload]
    a.width = a.width or dw
    a.height = a.height or a.width
    a.samples = a.samples or ds
    return a


def self_launch(n_gpus):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves, as CHILD processes of a
    parent that has not touched the GPU (never an exec), the way the driver would --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    -- relay their output, print rank 0's JSON line last and exit with the launcher's code."""
    import socket
    import subprocess
    with socket.socket() as s:   # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last_json = None
    for line in p.stdout:
        s = line.strip()
        if s.startswith("{") and s.endswith("}") and '"metric"' in s:
            last_json = s           # held back: it must be the LAST line of our stdout
        else:
            sys.stdout.write(line)
    rc = p.wait()
    sys.stdout.flush()
    if last_json is not None:
        print(last_json, flush=True)
    if rc == 0 and last_json is None:
        rc = 1
    raise SystemExit(rc)


def secondary(out, key, fn):
    """A secondary figure of the line: whatever goes wrong in it is recorded under its key and does not cost the line its
    headline, its roofline block or the CPU baseline."""
    try:
        val = fn()
        if val is not None:
            out[key] = val
    except Exception as e:   # noqa: BLE001
        out[key] = {"error": f"{type(e).__name__}: {e}"}


def build_frames(rt, wl, W, H, S, fov_x, fov_y, pixels, jitter, sky):
    """The DeviceFrames one step passes over (and, disk workload, the batch that traces them with one call)."""
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch, synthetic_sky
    a = wl.a
    frames, batch = [], None
    if a.workload == "disk":
        # traced by ONE library call with per-ray origins (FrameBatch); shaded frame by frame
        batch = FrameBatch(rt.ctx, wl.disk_cameras(), W, H, S, pixels=pixels, jitter=jitter, fov_x=0.9, fov_y=0.9, sampling_seed=42.0)
        frames = batch.frames
        disk_tex = synthetic_sky(1024, 128, seed=3)
        for f in frames:
            f.set_disk(DISK[0], DISK[1], disk_tex)
    else:
        # the trace writes what spacetime_ray_cast returns -- exit position AND direction (:307-308; north_star: "exit
        # position/direction written back"); --dir-only: only the directions a sky frame reads (background_hit, :366-378)
        frames.append(DeviceFrame(rt.ctx, W, H, S, fov_x=fov_x, fov_y=fov_y, sampling_seed=42.0, origin=CAM, pixels=pixels,
                                  jitter=jitter, directions_only=(a.workload == "frame" and a.dir_only)))
    for f in frames:
        f.set_sky(sky)
        f.generate_rays()
    return frames, batch


def measure(rt, wl, sky, nx, ny, ramp, overlap=False, whole_frames=False, reps=1):
    """One timed region over a frame of (width * nx) x (height * ny) pixels sharded over the ranks.
    Returns the figures of this rank (dt already the maximum over ranks).
      overlap (frame workload): two frames in flight -- consecutive frames alternate between two streams / contexts.
      whole_frames (orbit workload, N > 1): whole frames are dealt round-robin to the ranks -- rank r renders frames
        r, r + N, ... of the K-frame animation completely, nothing is exchanged until ONE gather of the ranks' last images
        at the end (an animation's frames are independent: no tail of a short shard launch, no per-frame collective)."""
    from blackhole_geodesic_calculator_amd import dist as bdist
    from blackhole_geodesic_calculator_amd.raygen import python_random_stream
    a, torch, dist, world, rank = wl.a, rt.torch, rt.dist, rt.world, rt.rank
    W, H, S = a.width * nx, a.height * ny, a.samples
    # ---- synthetic input, resident in HBM before the timed region ----------------------------
    # this rank's tiles of the frame (all samples of a pixel together); jitter stream = the
    # reference's random.seed(42) MT19937 doubles; rays generated on device once (the engine
    # re-seeds identically on every render(), so every frame of a static camera traces the same
    # rays); sky = deterministic synthetic equirect image (no dataset: "data": "synthetic").
    # The frame always spans the same window of directions (0.6 x 0.6 in the pinhole's tangent plane): for a
    # non-square rank grid (N = 2, 8) fov_y is widened by nx / ny, because y_render carries the aspect
    # factor H / W (RelativisticRenderEngine.py:197-198); every rank then samples the same distribution of rays
    fov_x, fov_y = 0.6, 0.6 * nx / ny
    shard_world, shard_rank = (1, 0) if whole_frames else (world, rank)     # whole frames: every rank holds the whole frame
    tile_cost = wl.shadow_edge_cost(W, H)
    tile_cost.visit = a.visit if a.visit != "auto" else ("row" if shard_world == 1 else "cost")
    # (the model only for the plain frame: with a disk or the orbiting sphere it is wrong and the order measured 1.5 % slower)
    tcost = tile_cost if (a.lpt and a.order != "none" and a.workload == "frame") else None
    jitter = python_random_stream(42.0, 2 * S * W * H)
    pixels = bdist.rank_pixels(W, H, a.tile, shard_rank, shard_world, tile_cost=tcost)
    frames, batch = build_frames(rt, wl, W, H, S, fov_x, fov_y, pixels, jitter, sky)
    root_share = None
    if a.lpt and a.order == "measured":
        # One untimed calibration trace of the frame prices every tile by the attempted steps of its rays; the tiles
        # are then dealt to the ranks and visited longest-first by THAT (the renderer traces the same pixels sample
        # after sample and frame after frame, :242-250: the last pass prices the next).  Same rays, another order.
        if a.workload == "orbit":
            frames[0].set_objects(*wl.orbit_scene(0))
        (batch or frames[0]).trace(wl.params)
        cost = sum(f.pixel_cost() for f in frames).cpu().numpy()
        tcost = bdist.measured_tile_cost(W, H, a.tile, pixels, cost)
        tcost.visit = tile_cost.visit
        del frames, batch
        pixels = bdist.rank_pixels(W, H, a.tile, shard_rank, shard_world, tile_cost=tcost)
        frames, batch = build_frames(rt, wl, W, H, S, fov_x, fov_y, pixels, jitter, sky)
    if shard_world > 1 and tcost is not None and a.root_share != "1":
        # Rank 0 owns the frame: on top of its shard it receives the gather and assembles W x H pixels per frame.  Dealt a
        # correspondingly smaller shard, all ranks finish together.  Measured here (rank 0, HIP events, untimed): T = trace +
        # shade of one frame of an equal shard, t_root = the assembly kernel over the whole frame; with rank 0 dealt rho
        # times what the others get, rho T_o + t_root = T_o and (N - 1 + rho) T_o = N T give
        # rho = (N T - (N - 1) t_root) / (N T + t_root); rank 0 broadcasts it and the tiles are dealt again.
        if a.root_share == "auto":
            rho = torch.zeros(1, dtype=torch.float64, device="cuda")
            if rank == 0:
                probe = bdist.FrameGatherer(W, H, a.tile, channels=4, dtype=torch.float32, device="cuda", assemble=rt.assemble,
                                            tile_cost=tcost)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                slab = torch.zeros((len(pixels), 4), dtype=torch.float32, device="cuda")
                for rep in range(2):        # (first pass: warm-up)
                    ev[0].record(rt.stream)
                    for _ in range(4):
                        for f in ([batch] if batch is not None else frames):
                            f.trace(wl.params)
                        for f in frames:
                            f.shade_f32(slab, None)
                    ev[1].record(rt.stream)
                    ev[2].record(rt.stream)
                    for _ in range(4):
                        if getattr(probe, "recv_all", None) is not None:
                            rt.assemble(probe.recv_all[0], probe.perm, probe.frame)
                    ev[3].record(rt.stream)
                    torch.cuda.synchronize()
                T_, t_root = ev[0].elapsed_time(ev[1]) / 4.0, ev[2].elapsed_time(ev[3]) / 4.0
                rho[0] = (shard_world * T_ - (shard_world - 1) * t_root) / (shard_world * T_ + t_root)
                del probe, slab
            dist.broadcast(rho, src=0)
            root_share = float(min(1.0, max(0.5, rho.item())))
        else:
            root_share = float(a.root_share)
        if root_share < 0.995:
            tcost.root_share = root_share
            del frames, batch
            pixels = bdist.rank_pixels(W, H, a.tile, shard_rank, shard_world, tile_cost=tcost)
            frames, batch = build_frames(rt, wl, W, H, S, fov_x, fov_y, pixels, jitter, sky)
    del jitter
    fr = frames[0]
    n = sum(f.n for f in frames)

    # frame end: per-pixel RGBA (fp32, what Blender's layer.rect holds) handed to the frame owner.
    # N > 1: ONE gather per frame over RCCL, issued asynchronously so it overlaps the next frame's
    # trace (dist.FrameGatherer: two slabs in rotation; rank 0 puts the slabs into frame order).
    gatherer = bdist.FrameGatherer(W, H, a.tile, channels=4, dtype=torch.float32, device="cuda", assemble=rt.assemble,
                                   tile_cost=tcost, collective=rt.collective and not whole_frames)   # (the gatherer must know the shards' pixel order)
    if not whole_frames:
        assert np.array_equal(gatherer.pixels, pixels)
    kernel_ms = []
    lanes = Lanes(fr, rt.local_rank, two=True) if (overlap and a.workload == "frame") else None
    own_image = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda") if whole_frames else None

    def step(i, timed):
        # HIP events around the trace call of every EV_EVERY-th timed step (whole frames: of this rank's every EV_EVERY-th)
        timed = timed and (i // world if whole_frames else i) % EV_EVERY == 0
        if lanes is not None:
            f, st = lanes[i]
            with torch.cuda.stream(st):
                if timed:
                    traced_with_events(f, wl.params, st, kernel_ms)
                else:
                    f.trace(wl.params)
                gatherer.submit_with(i, f.shade_f32)
            return
        if whole_frames:
            if i % world != rank:
                return                                   # another rank's frame
            fr.set_objects(*wl.orbit_scene(i))
            if timed:
                traced_with_events(fr, wl.params, rt.stream, kernel_ms)
            else:
                fr.trace(wl.params)
            fr.shade_f32(own_image, fr.d_pixels)
            return
        if a.workload == "orbit":
            fr.set_objects(*wl.orbit_scene(i))
        for f in ([batch] if batch is not None else frames):
            if timed:
                traced_with_events(f, wl.params, rt.stream, kernel_ms)
            else:
                f.trace(wl.params)
        for j, f in enumerate(frames):
            # shade + sample mean written as float RGBA straight into the gather slab (N > 1) or, single rank,
            # into the frame image in frame order
            gatherer.submit_with(i * len(frames) + j, f.shade_f32)

    last_images = None

    def barrier(final=False):
        nonlocal last_images
        gatherer.drain()
        if whole_frames and final and world > 1:
            # the animation's ONE exchange: every rank's last image to rank 0
            last_images = [torch.empty_like(own_image) for _ in range(world)] if rank == 0 else None
            dist.gather(own_image, last_images, dst=0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the clock sampler is BUILT here, before anything is timed (see the timed region below)
    sampler = None
    if not overlap and not whole_frames and not a.lean and rank == 0:
        try:
            sampler = ClockSampler(rt.local_rank)
            if not sampler.usable():
                sampler = None
        except Exception:   # noqa: BLE001
            sampler = None
    if ramp > 0:      # clock ramp (untimed, not counted in W or K)
        t_ramp = time.perf_counter()
        while True:
            for i in range(4 * (world if whole_frames else 1)):
                step(i, False)
            torch.cuda.synchronize()
            # every rank must run the same number of frames (each one is a collective): rank 0's clock decides
            go = torch.tensor([1.0 if time.perf_counter() - t_ramp < ramp else 0.0], device="cuda")
            if world > 1:
                dist.broadcast(go, src=0)
            if float(go.item()) == 0.0:
                break
        barrier()
    # roofline calibration (bhg_peak_probe: the fp64 rate THIS box sustains) -- outside the headline's clock.  Since round 6
    # the probes run AFTER the timed region only (--probe-when after_only): they are power-bound pure-FMA launches, and the
    # driver's command times a 28-ms region five warm-up steps behind them (profiles/r06_driver_cmd_ab.log)
    calibrate = not overlap and not whole_frames and not a.lean and a.probe_when != "off"
    calibration = {}
    def probes(when):        # (a probe that fails leaves the line without that part of `calibration`, not without its headline)
        try:
            calibration[when] = run_probes(rt.ctx, rt.local_rank)
        except Exception as e:   # noqa: BLE001
            calibration[when + "_error"] = f"{type(e).__name__}: {e}"
    if calibrate and a.probe_when == "before_warmup":
        probes("before")
        barrier()
    for i in range(a.warmup):
        step(i, False)
    barrier()
    if calibrate and a.probe_when == "before_timed":
        probes("before")
        barrier()
    # The timed region: EXACTLY K steps between barrier + synchronise on both sides -- `reps` times back to back, every
    # repetition with its own wall clock and its own HIP-event samples.  The headline is the MEDIAN repetition; all of them
    # go into the line.  NOTHING runs on the host between the barrier in front of a repetition and its clock: round 5 built
    # its clock sampler right there (a sysfs glob + torch.cuda.get_device_properties, milliseconds with the GPU idle) and
    # the driver's 28-ms region read 12 % slow for it; and the sampler itself -- reads of hwmon files, each one a message
    # to the GPU's power controller -- costs a region 5-6 % while it runs (profiles/r06_matrix_b.log).  So: the headline's
    # repetitions run unobserved but for --clock-reads reads of the clock each (default ONE, mid-region, from the main
    # thread while the GPU works through the enqueued steps), and the thread sampler runs in ONE EXTRA repetition behind them
    # that counts for nothing but the clock figure (its own ms per step is in the line: what sampling costs).
    dts, dts_local, rep_clock, rep_events = [], [], [], []
    for rep in range(max(1, reps)):
        n_ev = len(kernel_ms)
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i, True)
        reads = []
        if sampler is not None and a.clock_reads > 0:
            for q in range(a.clock_reads):
                if q:
                    time.sleep(0.004)
                reads.append(sampler.read_once())
        barrier(final=True)
        dt_rep = time.perf_counter() - t0
        dts_local.append(dt_rep)
        if world > 1:
            tmax = torch.tensor([dt_rep], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_rep = float(tmax.item())
        dts.append(dt_rep)
        reads = [r_ for r_ in reads if r_ is not None]
        rep_clock.append(float(np.mean(reads)) if reads else None)
        rep_events.append(kernel_ms[n_ev:])
    dt = float(np.median(dts))
    sclk, sampled_rep_ms = None, None
    # the extra, SAMPLED repetition (not a headline sample).  Every step is a collective when N > 1: all ranks run it or none
    # does -- rank 0, the only one that samples, says which
    extra = torch.tensor([1.0 if (sampler is not None and calibrate) else 0.0], device="cuda")
    if world > 1:
        dist.broadcast(extra, src=0)
    if float(extra.item()) != 0.0:
        try:
            if sampler is not None:
                sampler.start()
        except Exception:   # noqa: BLE001
            pass
        # (long enough for a clock mean worth the name: >= 0.12 s, i.e. >= 30 samples at the sampler's 4-ms period -- the
        # driver's 20-step region is 28 ms, 7 samples: ADVICE r05)
        n_extra = max(a.steps, int(0.12 / max(dt / a.steps, 1e-6)) + 1)
        if whole_frames:
            n_extra = a.steps
        t0 = time.perf_counter()
        for i in range(n_extra):
            step(i, False)
        barrier(final=True)
        sampled_rep_ms = (time.perf_counter() - t0) / n_extra * 1e3
        try:
            sclk = sampler.stop() if sampler is not None else None
        except Exception:   # noqa: BLE001
            sclk = None
    sclks = rep_clock
    if calibrate:
        probes("after")
        try:
            calibration["hbm_copy_GBps"] = hbm_copy_probe()
        except Exception as e:   # noqa: BLE001
            calibration["hbm_copy_error"] = f"{type(e).__name__}: {e}"

    # the frame really is the frame: rank 0's assembled image against a shade of ITS OWN pixels at their places
    # (outside the timed region; catches a slab / pixel-order mismatch)
    if rank == 0 and len(frames) == 1 and not whole_frames:
        img = gatherer.image().reshape(-1, 4)
        own = torch.empty((fr.P, 4), dtype=torch.float32, device="cuda")
        fr.shade_f32(own)
        torch.cuda.synchronize()
        assert torch.equal(img[fr.d_pixels], own), "assembled frame does not hold rank 0's pixels at their places"

    ray_steps = sum(int(f.d_steps.to(torch.int64).sum().item()) for f in frames)
    # per step: the trace calls of all its frames (HIP events on the stream the library launches on); one sample = one
    # step's call(s).  Every sample goes into the line; kernel_ms is their MEDIAN times the trace kernel's share (below)
    call_samples = [[float(e0.elapsed_time(e1)) for e0, e1 in evs] for evs in rep_events]
    flat = [x for r_ in call_samples for x in r_]
    call_ms = float(np.median(flat)) if flat else float("nan")
    call_ms_mean = float(np.mean(flat)) if flat else float("nan")
    # N > 1: every rank's own clock beside the maximum the headline takes (one all-reduce of a [world, reps + 1] table each rank
    # fills its row of: ms per step of every repetition by ITS wall clock, and its median trace-call time) -- so that a slow
    # rank (a slow box, or rank 0's root share dealt wrongly) can be told from a slow job
    per_rank = None
    if world > 1:
        tab = torch.zeros((world, len(dts_local) + 1), dtype=torch.float64, device="cuda")
        tab[rank, :len(dts_local)] = torch.tensor([d_ / a.steps * 1e3 for d_ in dts_local], dtype=torch.float64)
        tab[rank, -1] = call_ms if flat else 0.0
        dist.all_reduce(tab)
        tab = tab.cpu().numpy()
        per_rank = [dict(rank=r_, ms_per_step_samples=[float(v) for v in tab[r_, :-1]], ms_per_step_median=float(np.median(tab[r_, :-1])),
                         trace_call_ms_median=float(tab[r_, -1])) for r_ in range(world)]
    # the dominant kernel alone: one launch per call finishes every ray (events are located and resumed rays
    # carry on inside trace_*_kernel); Kerr adds a prepare and a finalize launch.  A few extra profiled calls
    # after the timed region (HIP events recorded by the library around prepare | trace on this same stream)
    # give the trace kernel's share of the call, applied to the call time measured inside the timed region
    rt.ctx.set_profiling(True)
    tr = []
    for _ in range(8):
        (batch or fr).trace(wl.params)
        tr.append(rt.ctx.last_pass_ms())
    rt.ctx.set_profiling(False)
    torch.cuda.synchronize()
    # (Kerr: prepare | trace | finalize; the Schwarzschild forms report 0 for the passes they do not have.  A SHARE of
    # the profiled calls, not their absolute times: those calls are synchronous, the GPU idles between them)
    share = float(np.median([t["trace"] / (t["prepare"] + t["trace"] + t["post"]) for t in tr]))
    k_ms = call_ms * share
    frames_mine = len(range(rank, a.steps, world)) if whole_frames else a.steps
    tot = torch.tensor([n * (frames_mine if whole_frames else 1), ray_steps * (frames_mine if whole_frames else 1)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tot)
    if lanes is not None:
        lanes.close()
    return dict(W=W, H=H, S=S, n=n, ray_steps=ray_steps, dt=dt, call_ms=call_ms, k_ms=k_ms, rays_all=float(tot[0].item()),
                steps_all=float(tot[1].item()), launch=rt.ctx.last_launch(), fr=fr, tcost=tcost,
                visit=tile_cost.visit, root_share=root_share, calibration=calibration, sclk=sclk,
                dts=dts, sclks=sclks, call_samples=call_samples, call_ms_mean=call_ms_mean, share=share,
                sampled_rep_ms=sampled_rep_ms, per_rank=per_rank)


# ----------------------------------------------------------------------------------------------------------------------
# The CPU baseline leg: the ONLY place outside tests/ and __graft_entry__.smoke() that touches oracle/
# ----------------------------------------------------------------------------------------------------------------------
def effective_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs under a 16-CPU quota; 128 OpenMP threads there only add throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(k0, cam, a, kw):
    """The C oracle (a port of the algorithm, see oracle/geodesic_oracle.c) timed on this host's
    cores on a bounded sample of the same rays.  Reported baseline only; never the thing shipped."""
    from oracle import oracle as oc
    oc.build()
    cores = min(oc.num_threads(), effective_cores())
    n = len(k0)
    probe = k0[:: max(1, n // 16384)]
    t = time.perf_counter()
    oc.trace(probe, cam, n_threads=cores, **kw)
    rate = len(probe) / (time.perf_counter() - t)
    m = int(min(n, max(len(probe), rate * a.cpu_seconds)))
    stride = max(1, n // m)
    sample = np.ascontiguousarray(k0[::stride])
    t = time.perf_counter()
    o = oc.trace(sample, cam, n_threads=cores, **kw)
    dt = time.perf_counter() - t
    # the same port on ONE core (SURVEY.md section 8d asks for both): a smaller sample of the same rays
    m1 = int(max(256, min(len(sample), len(sample) / dt / max(cores, 1) * 3.0)))   # about 3 s
    s1 = np.ascontiguousarray(sample[:: max(1, len(sample) // m1)])
    t = time.perf_counter()
    o1 = oc.trace(s1, cam, n_threads=1, **kw)
    dt1 = time.perf_counter() - t
    return {
        "value": len(sample) / dt / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "ray_steps_per_s": float(o["n_attempted"].sum()) / dt,
        "sample": f"every {stride}th ray of rank 0's {n} rays ({len(sample)} rays, {dt:.1f} s, OpenMP over rays, {cores} threads = "
                  f"affinity/cgroup-quota cores of {os.cpu_count()} logical CPUs)",
        "single_core": {"value": len(s1) / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                        "ray_steps_per_s": float(o1["n_attempted"].sum()) / dt1,
                        "sample": f"{len(s1)} of those rays, {dt1:.1f} s, one thread"},
    }


def main():
    a = parse()
    if a.single_process:
        return main_single_process(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)     # (before anything touches the GPU)
    live = None
    if a.live_pmc and not a.lean and a.cpu_seconds > 0 and a.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        live = live_pmc(a)      # (child processes; this one has not touched the GPU yet)
    rt = Runtime(a)
    wl = Workload(a)
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
    sky = synthetic_sky(2048, 1024)
    world, rank = rt.world, rt.rank

    nx, ny = grid_for(world) if a.workload != "orbit" else (1, 1)   # orbit: ONE fixed frame over all ranks
    m = measure(rt, wl, sky, nx, ny, a.ramp_seconds, reps=a.reps)
    strong = frames_sharded = None
    per_step = lambda r: r["dt"] / a.steps    # noqa: E731  (dt = the median repetition of the region)
    if world > 1 and a.workload != "orbit":
        # BASELINE.json's metric read as strong scaling: ONE fixed frame of the single-GPU size over all ranks.  The
        # headline of the sharded path has two frames in flight per rank (consecutive frames on alternating streams): a
        # shard's persistent launch ends with a tail of a few tens of microseconds, a quarter of a 1/8 shard's kernel --
        # the next frame's first waves fill it.  The sequential figure is reported beside it.
        # (nothing below has ever run on N > 1 distinct GPUs: an error every rank raises alike -- an API refusing an argument --
        # costs the line its `strong` block, not its headline; an error on one rank only would hang the others at their next
        # collective either way)
        try:
            seq = measure(rt, wl, sky, 1, 1, 0.0)
            head = measure(rt, wl, sky, 1, 1, 0.0, overlap=True) if a.workload == "frame" else seq
        except Exception as e:   # noqa: BLE001
            seq = head = None
            strong = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 and a.workload != "orbit" and strong is None:
        strong = {"value": head["rays_all"] / per_step(head) / 1e6, "unit": "Mrays/s", "ms_per_step": per_step(head) * 1e3,
                  "ray_steps_per_s": head["steps_all"] / per_step(head), "scaling": "strong",
                  "frames_in_flight": 2 if head is not seq else 1,
                  "workload": f"ONE {seq['W']}x{seq['H']} x{seq['S']} frame sharded over {world} GPUs ({seq['n']} rays on rank 0), same K / W, "
                              f"barrier + max-over-ranks timing",
                  "root_share": head["root_share"],
                  "sequential": {"value": seq["rays_all"] / per_step(seq) / 1e6, "unit": "Mrays/s", "ms_per_step": per_step(seq) * 1e3,
                                 "trace_kernel_ms_rank0": seq["k_ms"]}}
        del seq, head
    if world > 1 and a.workload == "orbit" and a.shard in ("frames", "both"):
        # the 100-frame animation's other sharding: whole frames round-robin over the ranks
        try:
            wf = measure(rt, wl, sky, 1, 1, 0.0, whole_frames=True)
        except Exception as e:   # noqa: BLE001
            wf = None
            frames_sharded = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 and a.workload == "orbit" and a.shard in ("frames", "both") and frames_sharded is None:
        frames_sharded = {"value": wf["rays_all"] / wf["dt"] / 1e6, "unit": "Mrays/s", "ms_per_frame": wf["dt"] / a.steps * 1e3,
                          "ray_steps_per_s": wf["steps_all"] / wf["dt"], "trace_call_ms": wf["call_ms"],
                          "what": f"the same {a.steps} animation frames dealt round-robin to the {world} ranks as WHOLE frames (rank r renders "
                                  f"frames r, r + {world}, ...: {wf['n']} rays per launch, no tail of a 1/{world} shard, no per-frame collective), "
                                  f"ONE gather of the ranks' last images at the end, inside the timed region"}
        del wf
    W, H, S, n, ray_steps, dt, call_ms, k_ms = m["W"], m["H"], m["S"], m["n"], m["ray_steps"], m["dt"], m["call_ms"], m["k_ms"]
    rays_all, steps_all, fr = m["rays_all"], m["steps_all"], m["fr"]

    if rank == 0:
        traffic, traffic_source, valu_per_64 = counters_for(a, wl, live, ray_steps)
        bytes_per_ray = BYTES_PER_RAY_DIR if getattr(fr, "_dir_traced", False) else BYTES_PER_RAY
        if a.workload == "disk":
            # the five cameras' frames are ONE trace call with PER-RAY origins: 24 B/ray more read ("24 B in
            # (k0; x0 shared) or 48 B (per-ray x0)") -- up to round 5 the line priced this call at the shared-origin figure
            # and its traffic read 1.34 x the algorithmic bytes for it
            bytes_per_ray += 24
        F = wl.flop
        out = {
            "metric": wl.metric(),
            "value": rays_all / (dt / a.steps) / 1e6,
            "unit": "Mrays/s",
            "ray_steps_per_s": steps_all / (dt / a.steps),
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            # every repetition of the K-step region, in the order they ran (ms per step): the headline is their median
            "ms_per_step_samples": [d_ / a.steps * 1e3 for d_ in m["dts"]],
            "ms_per_step_spread": spread([d_ / a.steps * 1e3 for d_ in m["dts"]]),
            # the shader clock read from sysfs while each repetition ran (--clock-reads reads per repetition, main thread)
            "sclk_mhz_per_repetition": [None if c is None else round(c, 1) for c in m["sclks"]],
            **({"per_rank": m["per_rank"]} if m["per_rank"] else {}),
            "higher_is_better": True,
            "scaling": "strong" if a.workload == "orbit" else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl.describe(W, H, S, world),
                "regime": a.regime, "integrator": "DP5(4) scipy-RK45 controller" if wl.method == "dp54" else "RK4 h=0.1",
                "rtol": 1e-3, "atol": 1e-6, "max_step": (0.1 if a.regime == "fine" else "inf"),
                "rhs_form": a.rhs, "rays_per_gpu": n, "attempted_steps_per_ray": ray_steps / n,
                "tile": a.tile, "tile_order": wl.tile_order_text(m["visit"]),
                "trace_output": "exit directions + flags + step counts (25 + 8 B/ray)" if getattr(fr, "_dir_traced", False) else "end states + flags + step counts (49 + 8 B/ray)",
                "frame_end": "device shade + per-pixel sample mean, written as float RGBA " + ("into the gather slab + 1 async RCCL gather to rank 0 + root-side assembly kernel" if rt.collective else "in frame order"),
                "collective": ("%s gather, %d rank(s)%s" % ("rccl" if rt.backend == "nccl" else rt.backend + " (development aid, ranks sharing a GPU)", world,
                                                           " (BHGEO_FORCE_COLLECTIVE)" if world == 1 else "")) if rt.collective else "none (single rank)",
                "parallelism": f"one process per GPU (torch.distributed), {world} rank(s)",
                "root_share": m["root_share"],
                "launch": m["launch"],
                "timed_region": f"{len(m['dts'])} repetition(s) of EXACTLY {a.steps} steps between barrier + synchronise (max over ranks each), "
                                f"after {a.warmup} warm-up steps; value / ms_per_step = the MEDIAN repetition, every repetition in "
                                f"ms_per_step_samples; calibration probes: {a.probe_when}",
            },
            "roofline": roofline_block(wl, ray_steps, k_ms, call_ms, n, bytes_per_ray, traffic, traffic_source, valu_per_64,
                                       calibration=m["calibration"], sclk=m["sclk"], call_samples=m["call_samples"], share=m["share"],
                                       num_cus=rt.ctx.num_cus, sampled_rep_ms=m["sampled_rep_ms"]),
        }
        if a.workload == "frame":
            out["config"]["north_star_output"] = ("exit directions only (--dir-only)" if getattr(fr, "_dir_traced", False) else
                                                  "full_records: exit position and direction (x, k: 48 B/ray), flags, step counts -- what "
                                                  "spacetime_ray_cast returns (RelativisticRenderEngine.py:307-308)")
        if strong is not None:
            out["strong"] = strong
        if frames_sharded is not None:
            out["frames_sharded"] = frames_sharded
        t1 = None
        if world == 1 and a.workload == "frame" and not a.lean:
            # the other output form of the same frame, same K / W, same clock: whole end states (81 B/ray: what north_star
            # names -- "exit position/direction written back" -- and the headline) against the exit directions alone that a
            # sky frame reads (57 B/ray; background_hit, :366-378)
            other_dir = not getattr(fr, "_dir_traced", False)

            def other_output_form():
                fro = DeviceFrame(rt.ctx, W, H, S, fov_x=fr.fov_x, fov_y=fr.fov_y, sampling_seed=42.0, origin=CAM,
                                  pixels=fr.d_pixels.cpu().numpy(), jitter=np.zeros(2), directions_only=other_dir)
                fro.d_k0 = fr.d_k0
                fro.set_sky(sky)
                ms_f, call_f, steps_f = time_frame(fro, wl.params, a.steps, a.warmup, device=rt.local_rank, ramp=a.ramp_seconds)
                return {
                    "value": n / (ms_f * 1e-3) / 1e6, "unit": "Mrays/s", "ms_per_step": ms_f, "trace_call_ms": call_f,
                    "frac": steps_f * F / (call_f * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                    "algorithmic_bytes_per_ray": BYTES_PER_RAY_DIR if other_dir else BYTES_PER_RAY,
                    "what": ("the same frame and K / W with only the exit directions written (24 B/ray: all a sky frame reads of its rays) and "
                             "shaded from them" if other_dir else
                             "the same frame and K / W with whole end states written (x, k: 48 B/ray) and shaded from them")}
            secondary(out, "sky_frame_dir_only" if other_dir else "full_records", other_output_form)
        if world == 1 and a.workload == "frame" and a.regime == "adaptive" and not a.lean:
            # SURVEY section 8d asks for both step regimes: the fine one (max_step 0.1, ~490 steps per ray) and the fixed-step RK4
            # (h = 0.1, 500 steps per ray) on the same frame, a few steps each (40 and 16 ms per step), same output form
            def regimes():
                import copy
                regs = {}
                for reg in ("fine", "rk4"):
                    a2 = copy.copy(a)
                    a2.regime = reg
                    w2 = Workload(a2)
                    ms_r, call_r, steps_r = time_frame(fr, w2.params, 10, 2, device=rt.local_rank, ramp=0.0)
                    regs[reg] = {"value": n / (ms_r * 1e-3) / 1e6, "unit": "Mrays/s", "ms_per_step": ms_r, "ray_steps_per_s": steps_r / (ms_r * 1e-3),
                                 "attempted_steps_per_ray": steps_r / n, "flop_per_ray_step": w2.flop,
                                 "frac": steps_r * w2.flop / (call_r * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                 "what": ("DP5(4), max_step = 0.1 (the Cam edition's pickle-name value)" if reg == "fine" else "classic RK4, h = 0.1") +
                                         ", 10 timed steps after 2 warm-up steps; frac from the trace call's HIP-event time"}
                fr.trace(wl.params)      # (leave the frame's buffers holding the adaptive result)
                return regs
            secondary(out, "regimes", regimes)
        if world == 1 and a.workload == "frame" and a.emulate_shards.strip():
            def predicted():
                t1_ms, t1_call, _ = time_frame(fr, wl.params, a.steps, a.warmup, device=rt.local_rank, ramp=a.ramp_seconds)
                return strong_predicted(rt, wl, sky, m, (t1_ms, t1_call))
            secondary(out, "strong_predicted", predicted)
        if world == 1 and a.workload == "frame" and a.cpu_seconds > 0:   # (--cpu-seconds 0 = kernels only: profiling runs)
            secondary(out, "pipelined", lambda: pipelined_figure(rt, fr, wl.params, a))
            secondary(out, "host_buffer_call", lambda: host_buffer_figures(rt.ctx, fr, CAM, wl.params, n))
        if a.cpu_seconds > 0 and world == 1:   # the CPU baseline is an N = 1 figure (rank 0's host cores, nothing else running)
            okw = dict(wl.okw)
            if a.workload == "orbit":
                okw["spheres"] = wl.orbit_scene(a.steps - 1)[0]
            secondary(out, "cpu_baseline", lambda: cpu_baseline(fr.d_k0.cpu().numpy(), fr.origin, a, okw))
    rt.close()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
