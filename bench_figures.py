"""bench_figures.py -- everything of the bench line that is NOT the headline's timed region: the PMC counters measured
in the run (live_pmc) or replayed (pmc_traffic), the secondary blocks (full_records via time_frame, strong_predicted,
pipelined, host_buffer_call, cpu_baseline) and the --single-process form.  The headline clock lives in bench.py's
measure() and nowhere else.  See bench.py."""
import json
import os
import sys
import time

import numpy as np

from bench_common import (BYTES_PER_RAY, BYTES_PER_RAY_DIR, CAM, DISK, EV_EVERY, PEAK_FP64_VALU_TFLOPS, ROOT, Lanes, Workload, emit,
                          grid_for, ramp_clocks, roofline_block, traced_with_events)

def live_pmc(a, deadline_s=300.0):
    """HBM bytes and wave-level VALU instructions per launch of the trace kernel, measured in THIS run: three rocprofv3
    child runs of this same command (--lean, 3 timed steps), one counter each -- FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU in
    separate passes, as MI355X_MICROARCH.md prescribes; units KiB, FETCH_SIZE doubled on gfx950 (scripts/summarize_pmc.py
    applies the same corrections to the committed profiles).  Children of a parent that has not touched the GPU yet, each
    in a process group of its own: a child that overruns is killed WITH its descendants (a surviving grandchild would keep
    the GPU busy during the headline's timed region), and the three passes share ONE deadline.
    Returns {"hbm", "valu", "source", "ray_steps"} -- ray_steps: attempted ray-steps per trace launch as the CHILD run
    itself reports them, what its counters are to be normalised with -- or {"error": why}: the committed summary is then
    replayed and roofline.traffic_source says why."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 is not on PATH"}
    # (already under a profiler -- its preloaded library has initialised the GPU in this process and would ride along
    # into the children: leave it to that run)
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ):
        return {"error": "this run is itself under a profiler"}
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--lean", "--live-pmc", "0", "--steps", "3", "--warmup", "1", "--ramp-seconds", "0",
             "--cpu-seconds", "0", "--regime", a.regime, "--rhs", a.rhs, "--workload", a.workload, "--tile", str(a.tile),
             "--order", a.order, "--visit", a.visit, "--lpt", str(a.lpt)]
    for flag, val in (("--width", a.width), ("--height", a.height), ("--samples", a.samples)):
        if val is not None:
            child += [flag, str(val)]
    if getattr(a, "dir_only", False):
        child.append("--dir-only")
    out, child_steps = {}, None
    t_end = time.monotonic() + deadline_s
    tmp = tempfile.mkdtemp(prefix="bhg_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(tmp, ctr)
            left = t_end - time.monotonic()
            if left <= 5.0:
                return {"error": f"the {deadline_s:.0f}-s budget of the three counter passes ran out before {ctr}"}
            p = subprocess.Popen(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + child,
                                 cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                 text=True, start_new_session=True)
            try:
                so, se = p.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)     # the profiler AND the program under it
                except ProcessLookupError:
                    pass
                p.wait()
                return {"error": f"the {ctr} pass overran the budget and was killed with its process group"}
            if p.returncode != 0:
                return {"error": f"the {ctr} pass exited with code {p.returncode}: {(se or '').strip()[-200:]}"}
            for line in (so or "").splitlines():
                line = line.strip()
                if line.startswith("{") and '"metric"' in line:
                    try:
                        child_steps = float(json.loads(line)["roofline"]["ray_steps_per_launch"])
                    except Exception:
                        pass
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == ctr and "trace_" in row.get("Kernel_Name", ""):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return {"error": f"the {ctr} pass produced no counter rows for a trace_ kernel"}
            out[ctr] = (sum(vals) / len(vals), len(vals))
    except Exception as e:       # (anything else: say what, never raise -- the bench line must still come out)
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    hbm = 2.0 * out["FETCH_SIZE"][0] * 1024.0 + out["WRITE_SIZE"][0] * 1024.0
    src = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU, three child runs of this command before the "
           "timed region (means over %d / %d / %d trace launches of the child runs -- warm-up, timed and profiled calls alike; "
           "KiB, FETCH_SIZE x2 on gfx950; normalised with the child run's own ray-steps per launch)"
           % (out["FETCH_SIZE"][1], out["WRITE_SIZE"][1], out["SQ_INSTS_VALU"][1]))
    return {"hbm": hbm, "valu": out["SQ_INSTS_VALU"][0], "source": src, "ray_steps": child_steps}


def pmc_traffic(a, method):
    """HBM bytes per launch of the dominant kernel from the latest committed PMC summary of THIS workload
    (profiles/rNN*_pmc_summary[_<workload>].json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this
    same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The replay used when the live
    measurement is off or failed: (bytes, file name), or (None, None) if there is none for this configuration.  Third
    value: wave-level VALU instructions per launch (SQ_INSTS_VALU pass of the same script), or None."""
    import glob
    if not (a.regime == "adaptive" and method == "dp54" and a.rhs in ("christoffel", "kerr")):
        return None, None, None
    dflt = {"frame": (1024, 5), "disk": (1024, 1), "orbit": (2048, 16)}[a.workload]
    if (a.width, a.height, a.samples) != (dflt[0], dflt[0], dflt[1]):
        return None, None, None
    tag = {"frame": "", "disk": "_disk", "orbit": "_orbit"}[a.workload] + ("_kerr" if a.rhs == "kerr" else "")
    if a.workload == "frame" and a.rhs != "kerr" and getattr(a, "dir_only", False):
        tag = "_dir"       # (the sky frame's direction-only trace writes 24 B/ray less than the full-record headline)
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.json"))
                   if re.fullmatch(r"r\d+_pmc_summary" + re.escape(tag) + r"\.json", os.path.basename(f)))
    if not files:
        return None, None, None
    try:
        s = json.load(open(files[-1]))
        return s.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(files[-1]), s.get("valu_insts_per_launch")
    except Exception:
        return None, None, None


def counters_for(a, wl, live, ray_steps):
    """(traffic, traffic_source, valu per 64 ray-steps): the live measurement, or the committed summary with the reason."""
    traffic, source, valu = pmc_traffic(a, wl.method)
    valu_per_64 = None if not valu else valu * 64.0 / ray_steps
    if live is not None and "error" not in live:
        traffic, source = live["hbm"], live["source"]
        valu_per_64 = live["valu"] * 64.0 / (live["ray_steps"] or ray_steps)
    elif live is not None and source is not None:
        source = f"{source} (replayed: the live measurement failed -- {live['error']})"
    elif live is not None:
        source = f"none (the live measurement failed -- {live['error']} -- and no committed summary matches this configuration)"
    return traffic, source, valu_per_64


def time_frame(fr_, params, steps, warmup, overlap=False, device=0, ramp=0.25, after_shade=None):
    """K timed steps of trace + shade of ONE DeviceFrame on this GPU: (ms per step by the wall clock around a synchronised
    region, trace-call ms by HIP events, attempted ray-steps).  Default: float RGBA written in frame order, no collective.
    after_shade(i, frame, stream) -> None, optional: called in each lane's stream context INSTEAD of the plain shade -- the
    root-side emulation of strong_predicted puts the shard's slab, its gather and the frame assembly there.
    overlap: two frames in flight, alternating between two streams / library contexts (the call times overlap then)."""
    import torch
    lanes_ = Lanes(fr_, device, two=overlap)
    imgs = [torch.zeros((fr_.W * fr_.H, 4), dtype=torch.float32, device="cuda") for _ in range(len(lanes_))]
    evs = []

    def run(k, timed):
        for i in range(k):
            f, st = lanes_[i]
            with torch.cuda.stream(st):
                if timed and i % EV_EVERY == 0:
                    traced_with_events(f, params, st, evs)
                else:
                    f.trace(params)
                if after_shade is not None:
                    after_shade(i, f, st)
                else:
                    f.shade_f32(imgs[i % len(lanes_)], fr_.d_pixels)
    torch.cuda.synchronize()
    ramp_clocks(lambda k: run(k, False), ramp)
    run(warmup, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out_ = dt / steps * 1e3, float(np.mean([x.elapsed_time(y) for x, y in evs])), int(fr_.d_steps.to(torch.int64).sum().item())
    lanes_.close()
    return out_


class EmulatedRoot:
    """Rank 0's frame end of a world-N run, on the one GPU at hand: the shard's float-RGBA slab ([pmax, 4], shade + sample
    mean written straight into it), ONE real collective -- a gather in a process group of one rank (RCCL; the message is
    rank 0's own slab, what every peer would send) issued asynchronously on the backend's stream, two slabs in rotation
    like dist.FrameGatherer -- and the root's assembly of the WHOLE N-rank frame (bhg_assemble_frame_f32_device over
    H x W pixels through the N-rank permutation; the other ranks' slots of the receive block hold whatever they hold: the
    kernel's work does not depend on it)."""

    def __init__(self, rt, W, H, tile, N, tcost):
        import torch
        from blackhole_geodesic_calculator_amd import dist as bdist
        self.rt, self.torch = rt, torch
        pix = [bdist.rank_pixels(W, H, tile, r, N, tile_cost=tcost) for r in range(N)]
        self.P = len(pix[0])
        self.pmax = max(len(p) for p in pix)
        perm = np.empty(H * W, dtype=np.int64)
        for r, p in enumerate(pix):
            perm[p] = r * self.pmax + np.arange(len(p), dtype=np.int64)
        self.perm = torch.from_numpy(perm).cuda()
        self.slabs = [torch.zeros((self.pmax, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        self.recv = [torch.zeros((N * self.pmax, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        self.frame = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
        self.pending = [None, None]
        self.pixels0 = pix[0]

    def finish(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
            self.rt.assemble(self.recv[b], self.perm, self.frame)

    def after_shade(self, i, f, st):
        b = i & 1
        self.finish(b)
        f.shade_f32(self.slabs[b][: self.P], None)
        self.pending[b] = self.rt.dist.gather(self.slabs[b], [self.recv[b][: self.pmax]], dst=0, async_op=True)

    def drain(self):
        for b in (0, 1):
            self.finish(b)


def strong_predicted(rt, wl, sky, m, t1):
    """BASELINE.json's metric is ONE 1024x1024x5 frame over 1, 2, 4, 8 GPUs.  Predicted on the one GPU at hand for rank 0 --
    the slowest rank: it traces its shard like everyone else AND receives and assembles the frame.  For each N: rank 0's
    pixel list of a world-N dealing of that fixed frame (same tiles, same cost order), traced and shaded here;
    `efficiency*` = T_1 / (N T_N) with T_N the shard alone (what round 3 reported), `efficiency_rank0*` with the shard's
    slab going through a real 1-rank RCCL gather and the root's assembly of the whole N-rank frame in FrameGatherer's
    stream order.  What is still left out: the peers' slabs arriving over xGMI (8.4 MB per frame in all, ~55 us of link
    time, overlapped with the next frame's trace)."""
    from blackhole_geodesic_calculator_amd import dist as bdist
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
    from blackhole_geodesic_calculator_amd.raygen import python_random_stream
    a, fr, torch = wl.a, m["fr"], rt.torch
    W, H, S = m["W"], m["H"], m["S"]
    jit = python_random_stream(42.0, 2 * S * W * H)
    t1_ms, t1_call = t1
    t1o_ms, _, _ = time_frame(fr, wl.params, a.steps, a.warmup, overlap=True, device=rt.local_rank, ramp=a.ramp_seconds)
    pred = {"T1_ms_per_step": t1_ms, "T1_trace_call_ms": t1_call, "T1_ms_per_step_two_in_flight": t1o_ms, "shards": {}}
    group_error = None
    try:
        rt.init_group()      # a process group of ONE rank: the gather below is a real collective of the backend
    except Exception as e:  # (no RCCL on this box: the rank-0 figures are then left out, with the reason)
        group_error = f"{type(e).__name__}: {e}"
    tile_cost1 = wl.shadow_edge_cost(W, H)
    tile_cost1.visit = a.visit if a.visit != "auto" else "cost"
    for N in [int(v) for v in a.emulate_shards.split(",") if v.strip()]:
        tc = ((m["tcost"] if a.order == "measured" else tile_cost1) if (a.lpt and a.order != "none") else None)
        pix = bdist.rank_pixels(W, H, a.tile, 0, N, tile_cost=tc)
        frs = DeviceFrame(rt.ctx, W, H, S, fov_x=fr.fov_x, fov_y=fr.fov_y, sampling_seed=42.0, origin=CAM, pixels=pix,
                          jitter=jit, directions_only=fr.directions_only)
        frs.set_sky(sky)
        frs.generate_rays()
        ms_n, call_n, steps_n = time_frame(frs, wl.params, a.steps, a.warmup, device=rt.local_rank, ramp=a.ramp_seconds)
        mso_n, _, _ = time_frame(frs, wl.params, a.steps, a.warmup, overlap=True, device=rt.local_rank, ramp=a.ramp_seconds)
        rec = {"rays": frs.n, "ms_per_step": ms_n, "trace_call_ms": call_n, "attempted_steps_per_ray": steps_n / frs.n,
               "efficiency": t1_ms / (N * ms_n), "efficiency_trace_call": t1_call / (N * call_n),
               "ms_per_step_two_in_flight": mso_n, "efficiency_two_in_flight": t1o_ms / (N * mso_n)}
        if group_error is None:
            root = EmulatedRoot(rt, W, H, a.tile, N, tc)
            assert np.array_equal(root.pixels0, pix)
            r_ms, _, _ = time_frame(frs, wl.params, a.steps, a.warmup, device=rt.local_rank, ramp=a.ramp_seconds, after_shade=root.after_shade)
            root.drain()
            ro_ms, _, _ = time_frame(frs, wl.params, a.steps, a.warmup, overlap=True, device=rt.local_rank, ramp=a.ramp_seconds,
                                     after_shade=root.after_shade)
            root.drain()
            torch.cuda.synchronize()
            rec.update(ms_per_step_rank0=r_ms, ms_per_step_rank0_two_in_flight=ro_ms,
                       efficiency_rank0_sequential=t1_ms / (N * r_ms), efficiency_rank0_equal_shares=t1o_ms / (N * ro_ms))
            del root
            # ... and with rank 0 dealt a smaller shard, so that root and peers finish together (what bench.py does at N > 1,
            # --root-share auto): rho from the two measured times, then rank 0's biased shard WITH the root's work and rank 1's
            # biased shard without, both with two frames in flight; the step is the slower of the two
            t_extra = max(ro_ms - mso_n, 0.0)
            rho = min(1.0, max(0.5, (N * mso_n - (N - 1) * t_extra) / (N * mso_n + t_extra)))
            rec["root_share"] = rho
            if tc is not None and rho < 0.995:
                def tcb(cx, cy, _tc=tc):
                    return _tc(cx, cy)
                tcb.visit, tcb.root_share = tc.visit, rho
                t_b = []
                for r_ in (0, 1):
                    pix_b = bdist.rank_pixels(W, H, a.tile, r_, N, tile_cost=tcb)
                    frb = DeviceFrame(rt.ctx, W, H, S, fov_x=fr.fov_x, fov_y=fr.fov_y, sampling_seed=42.0, origin=CAM, pixels=pix_b,
                                      jitter=jit, directions_only=fr.directions_only)
                    frb.set_sky(sky)
                    frb.generate_rays()
                    rootb = EmulatedRoot(rt, W, H, a.tile, N, tcb) if r_ == 0 else None
                    tb, _, _ = time_frame(frb, wl.params, a.steps, a.warmup, overlap=True, device=rt.local_rank, ramp=a.ramp_seconds,
                                          after_shade=None if rootb is None else rootb.after_shade)
                    if rootb is not None:
                        rootb.drain()
                        torch.cuda.synchronize()
                    t_b.append(tb)
                    del frb, rootb
                rec.update(ms_per_step_rank0_biased=t_b[0], ms_per_step_rank1_biased=t_b[1],
                           efficiency_rank0=t1o_ms / (N * max(t_b)))
            else:
                rec["efficiency_rank0"] = rec["efficiency_rank0_equal_shares"]
        pred["shards"][str(N)] = rec
        del frs
    del jit
    if group_error is not None:
        pred["rank0_error"] = group_error
    pred["what"] = ("rank 0's shard of a world-N dealing of the fixed %dx%d x%d frame on this one GPU; efficiency = T1 / (N T_N).  "
                    "efficiency / _two_in_flight: trace + shade of the shard alone, no collective (round 3's figures).  "
                    "efficiency_rank0_equal_shares (two frames in flight) / efficiency_rank0_sequential: the shard's slab additionally "
                    "goes through a 1-rank %s gather (asynchronous, two slabs in rotation) and the root assembles the WHOLE N-rank "
                    "frame from the receive block -- rank 0's step, the slowest rank's.  efficiency_rank0: the same with rank 0 dealt "
                    "root_share times what the others get (the default of the sharded path, --root-share auto): the slower of rank 0's "
                    "biased shard with the root's work and rank 1's biased shard without.  two_in_flight: "
                    "consecutive frames alternate between two streams / library contexts, the second stream at another priority "
                    "(a hardware queue of its own), so a frame's first waves start while the previous frame's last ones drain; T1 is "
                    "measured the same way" % (W, H, S, "RCCL" if rt.backend == "nccl" else rt.backend))
    return pred


def pipelined_figure(rt, fr, params, a):
    """Consecutive frames of an animation are independent: two frames in flight on two streams (two library
    contexts, each with its own work counters and workspace) let the next frame's waves start while the previous
    launch drains its last batches.  Reported beside `value`, never as `value`: the per-kernel durations the roofline
    figure rests on overlap here and mean something else."""
    import torch
    lanes = Lanes(fr, rt.local_rank, two=True)
    imgs = [torch.zeros((fr.W * fr.H, 4), dtype=torch.float32, device="cuda") for _ in range(2)]

    def run(k):
        for i in range(k):
            f, st = lanes[i]
            with torch.cuda.stream(st):
                f.trace(params)
                f.shade_f32(imgs[i & 1], fr.d_pixels)
    torch.cuda.synchronize()
    ramp_clocks(run, a.ramp_seconds)
    run(a.warmup)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    same = bool(torch.equal(imgs[0], imgs[1]))
    lanes.close()
    return {"value": fr.n / (dt / a.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": dt / a.steps * 1e3, "frames_identical": same,
            "what": f"{a.steps} frames alternating between two streams of different priority (two hardware queues) / two library "
                    f"contexts (trace + shade each), no synchronisation in between"}


def host_buffer_figures(ctx, fr, cam, params, n):
    """The host-buffer entry point (numpy in, numpy out: what the reference's Python caller would use), H2D + trace
    + D2H over PCIe as a chunked pipeline -- reported beside `value`, never as `value`.  Two figures: the adaptor's
    default (k0 a plain numpy array, results in the library's page-locked pool: the copy engines write what the
    caller receives) and everything in plain pageable numpy arrays (results cross a pinned staging ring with
    multi-threaded host copies)."""
    k_host = fr.d_k0.cpu().numpy()
    out = {}
    for key, pinned in (("value", True), ("pageable_results", False)):
        ctx.trace(k_host, cam, params, pinned_results=pinned)          # first call: allocations, page-locking
        best = float("inf")
        for _ in range(3):
            t = time.perf_counter()
            r = ctx.trace(k_host, cam, params, pinned_results=pinned)
            best = min(best, time.perf_counter() - t)
            del r
        out[key] = n / best / 1e6
        out["ms" if pinned else "pageable_results_ms"] = best * 1e3
    # the adaptors' resident-ray path (frame.FrameTracer, camera.RelativisticCamera): rays generated on the device once
    # (bhg_rays_create from the jitter stream), per frame only end_dir + flags come back (bhg_rays_trace)
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.raygen import python_random_stream
    W, H, S = fr.W, fr.H, fr.S
    jit = python_random_stream(42.0, 2 * S * W * H)
    t = time.perf_counter()
    rs = _ffi.RaySet(ctx, W, H, S, fr.fov_x, fr.fov_y, cam, None, jit, False, None)
    t_create = time.perf_counter() - t
    rs.trace(params, want=("end_dir", "flags"))
    best = float("inf")
    for _ in range(3):
        t = time.perf_counter()
        r = rs.trace(params, want=("end_dir", "flags"))
        best = min(best, time.perf_counter() - t)
        del r
    out["resident_rays"] = {"value": rs.n / best / 1e6, "unit": "Mrays/s", "ms": best * 1e3, "rays_create_ms": t_create * 1e3,
                            "what": "bhg_rays_trace over the whole frame, best of 3: rays generated on the device once from the "
                                    "MT19937 jitter stream (rays_create_ms, 16 B/ray up, not in ms), only end_dir + flags (25 B/ray) come back"}
    rs.close()
    # the library-owned frame (bhg_frame_*: what the Blender add-on's device path calls): everything between the jitter
    # stream and the averaged pixels on the GPU, ONE float RGBA array back per frame (16 B/pixel over PCIe)
    sky = fr.d_sky.cpu().numpy()
    fo = _ffi.Frame([ctx.device], W, H, S, fov_x=fr.fov_x, fov_y=fr.fov_y, origin=cam, jitter=jit)
    fo.set_scene(sky)
    img = fo.render(params)
    best = float("inf")
    for _ in range(3):
        t = time.perf_counter()
        fo.render(params, out=img)
        best = min(best, time.perf_counter() - t)
    out["library_frame"] = {"value": W * H * S / best / 1e6, "unit": "Mrays/s", "ms": best * 1e3,
                            "what": "bhg_frame_render into a pageable numpy array, best of 3: rays resident, trace + shade + sample "
                                    "mean on the device, one [H, W, 4] float image back (the add-on's device path; no torch)"}
    # ... into a page-locked array (bhg_host_alloc): the copy engine writes the caller's array, no staging copy
    img_p = ctx.pinned.empty((H, W, 4), np.float32)
    fo.render(params, out=img_p)
    best_p = float("inf")
    for _ in range(3):
        t = time.perf_counter()
        fo.render(params, out=img_p)
        best_p = min(best_p, time.perf_counter() - t)
    out["library_frame"].update(pinned_image_ms=best_p * 1e3, pinned_image_value=W * H * S / best_p / 1e6,
                                pinned_image_identical=bool(np.array_equal(img_p, img)))
    # ... and an animation's form (config 4 is 100 frames): two frame objects on two host threads, each call blocking on its
    # own frame -- one frame's image crosses PCIe while the other frame is traced
    import threading
    fo2 = _ffi.Frame([ctx.device], W, H, S, fov_x=fr.fov_x, fov_y=fr.fov_y, origin=cam, jitter=jit)
    fo2.set_scene(sky)
    img2 = ctx.pinned.empty((H, W, 4), np.float32)
    fo2.render(params, out=img2)
    K2 = 10
    def _animate(f_, o_):
        for _ in range(K2):
            f_.render(params, out=o_)
    th = [threading.Thread(target=_animate, args=(fo, img_p)), threading.Thread(target=_animate, args=(fo2, img2))]
    t = time.perf_counter()
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    dt2 = (time.perf_counter() - t) / (2 * K2)
    out["library_frame"].update(two_frames_ms=dt2 * 1e3, two_frames_value=W * H * S / dt2 / 1e6,
                                two_frames_identical=bool(np.array_equal(img2, img)),
                                two_frames_what=f"{2 * K2} frames, two frame objects on two host threads (page-locked images): wall time per frame")
    fo2.close()
    fo.close()
    # the engine's literal per-ray call (RelativisticRenderEngine.py:293-294: one ray, nr_points_curve = 10000) through the
    # adaptor -- what a caller gets who swaps the integrator object and nothing else
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
    if params.rhs_form != _ffi.RHS_KERR_BL:
        gi = GeodesicIntegratorSchwarzschild(mass=0.5 * params.r_s, time_like=False, verbose=False, context=ctx)
        cam_np = np.asarray(cam, dtype=np.float64)
        for i in range(20):
            gi.calc_trajectory(k_host[i], cam_np, max_step=1e4, curve_end=params.lambda_end, nr_points_curve=10000, verbose=False)
        t = time.perf_counter()
        for i in range(200):
            gi.calc_trajectory(k_host[(i * 2621) % len(k_host)], cam_np, max_step=1e4, curve_end=params.lambda_end, nr_points_curve=10000, verbose=False)
        us = (time.perf_counter() - t) / 200 * 1e6
        out["per_ray_call"] = {"us_per_call": us, "rays_per_s": 1e6 / us, "nr_points_curve": 10000,
                               "what": "GeodesicIntegratorSchwarzschild.calc_trajectory(k0, x0, ..., nr_points_curve=10000), mean of 200 "
                                       "calls: the reference engine's own call, one ray at a time (one wave per ray, samples over the lanes)"}
    out["unit"] = "Mrays/s"
    out["what"] = ("bhg_trace, PCIe-inclusive, best of 3 after one warm-up call: k0 from a pageable numpy array (staged by worker "
                   "threads), H2D || trace || D2H pipelined over 2^20-ray chunks; value: end/flags/n_steps/n_accepted arrive in "
                   "page-locked arrays from the library's pool (the Python adaptor's default); pageable_results: into plain numpy arrays")
    return out


def main_single_process(a):
    """The same workloads through bhg_frame_* (include/bhgeo.h; _ffi.Frame): one process, N devices, no torch.distributed
    and no torch in the timed path.  Weak scaling like the multi-process form (the frame grows with N over the same window
    of directions) plus the fixed frame as `strong`; a step = one bhg_frame_render(..., NULL) per frame of the workload
    (enqueue only: the image stays on device 0); both ends of the timed region wait for every device's stream."""
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.sky import synthetic_sky
    from blackhole_geodesic_calculator_amd.raygen import euler_xyz_matrix, python_random_stream
    wl = Workload(a)
    devices = [int(v) for v in os.environ["BHGEO_DEVICES"].split(",")] if os.environ.get("BHGEO_DEVICES") else list(range(a.gpus))
    N = len(devices)
    sky = synthetic_sky(2048, 1024)
    disk_tex = synthetic_sky(1024, 128, seed=3) if a.workload == "disk" else None
    gmode = {"auto": _ffi.GATHER_AUTO, "copy": _ffi.GATHER_COPY, "rccl": _ffi.GATHER_RCCL, "peer": _ffi.GATHER_PEER}[a.frame_gather]

    shares = []

    def build(nx, ny):
        W, H, S = a.width * nx, a.height * ny, a.samples
        jit = python_random_stream(42.0, 2 * S * W * H)
        frames = []
        if a.workload == "disk":
            for cam in wl.disk_cameras():
                f = _ffi.Frame(devices, W, H, S, fov_x=0.9, fov_y=0.9, origin=cam["origin"], rot=euler_xyz_matrix(cam["rotation_euler"]),
                               jitter=jit, tile=a.tile, gather=gmode)
                f.set_scene(sky, disk=DISK, disk_tex=disk_tex)
                frames.append(f)
        else:
            f = _ffi.Frame(devices, W, H, S, fov_x=0.6, fov_y=0.6 * nx / ny, origin=CAM, jitter=jit, tile=a.tile, gather=gmode)
            if a.workload == "orbit":
                sp, rgb, lamps = wl.orbit_scene(0)
                f.set_scene(sky, spheres=sp, sphere_rgb=rgb, lamps=lamps)
            else:
                f.set_scene(sky)
            frames.append(f)
        del jit
        if a.lpt and a.order != "none" and N > 1:
            # one untimed, profiled calibration render prices the tiles and times the devices' traces and the first device's
            # frame end; the tiles are then re-dealt longest-processing-time-first, the first device a smaller part
            # (rho = (N T - (N - 1) t_root) / (N T + t_root), see measure()) unless the frame end is free (peer stores)
            for f in frames:
                f.set_profiling(True)
                f.render(wl.params, to_host=False)
                tr, t_root = f.last_ms()
                f.set_profiling(False)
                T_ = float(np.mean(tr))
                rho = 1.0 if a.root_share == "1" else (float(a.root_share) if a.root_share != "auto" else
                                                       min(1.0, max(0.5, (N * T_ - (N - 1) * t_root) / (N * T_ + t_root))))
                f.rebalance(root_share=rho)
                shares.append(rho)
        return frames, W, H, S

    def timed(frames, frames_b=None):
        """frames_b: a second set of frame objects (their own contexts, streams and work counters) -- consecutive steps
        alternate between the two sets, so that step i + 1 starts while step i's last waves drain (two frames in flight)."""
        def step(i, profile):
            for f in (frames if (frames_b is None or i % 2 == 0) else frames_b):
                if a.workload == "orbit":
                    sp, rgb, lamps = wl.orbit_scene(i)
                    f.set_scene(None, spheres=sp, sphere_rgb=rgb, lamps=lamps)
                f.set_profiling(profile)
                f.render(wl.params, to_host=False)

        def sync():
            for f in frames + (frames_b or []):
                f.synchronize()
        if a.ramp_seconds > 0:
            t = time.perf_counter()
            while time.perf_counter() - t < a.ramp_seconds:
                for i in range(4):
                    step(i, False)
                sync()
        for i in range(a.warmup):
            step(i, False)
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i, i % EV_EVERY == 0)
        sync()
        dt = time.perf_counter() - t0
        call_ms = np.zeros(N)
        root_ms = 0.0
        for f in frames:       # a step's trace calls: one per frame of the workload, summed; per device
            tr, rm = f.last_ms()
            call_ms += np.array(tr)
            root_ms += rm
        for f in (frames_b or []):
            try:
                f.last_ms()    # (returns the second lane's event pairs to its pool; it may not have had a profiled step)
            except _ffi.BhgError:
                pass
        st = [f.stats() for f in frames]
        return dict(dt=dt, call_ms=call_ms, root_ms=root_ms, rays=sum(s["rays"] for s in st), steps=sum(s["attempted_steps"] for s in st),
                    info=frames[0].info())

    nx, ny = grid_for(N) if a.workload != "orbit" else (1, 1)
    frames, W, H, S = build(nx, ny)
    m = timed(frames)
    pipelined = None
    if not a.lean:
        # two frames in flight: a second set of frame objects, steps alternating between the sets
        fb, _, _, _ = build(nx, ny)
        p_ = timed(frames, fb)
        for f in fb:
            f.close()
        pipelined = {"value": p_["rays"] / (p_["dt"] / a.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": p_["dt"] / a.steps * 1e3,
                     "what": "the same steps alternating between TWO frame objects (each with its own contexts and streams): "
                             "a frame starts while the previous one's last waves drain; never `value`"}
    for f in frames:
        f.close()
    strong = None
    if N > 1 and a.workload != "orbit":
        fs, Ws, Hs, _ = build(1, 1)
        s_ = timed(fs)
        fb, _, _, _ = build(1, 1)
        s2 = timed(fs, fb)
        for f in fs + fb:
            f.close()
        strong = {"value": s_["rays"] / (s_["dt"] / a.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": s_["dt"] / a.steps * 1e3,
                  "ray_steps_per_s": s_["steps"] / (s_["dt"] / a.steps), "scaling": "strong", "trace_call_ms_per_device": [float(v) for v in s_["call_ms"]],
                  "root_gather_assembly_ms": s_["root_ms"],
                  "two_frames_in_flight": {"value": s2["rays"] / (s2["dt"] / a.steps) / 1e6, "ms_per_step": s2["dt"] / a.steps * 1e3,
                                           "what": "steps alternating between two frame objects"},
                  "workload": f"ONE {Ws}x{Hs} x{S} frame sharded over {N} device(s) of one process"}
    # the dominant kernel: the slowest device's trace call (for the Schwarzschild forms the call IS the one trace kernel; Kerr
    # adds its prepare and finalize passes -- the call time is then an upper bound of the kernel's)
    k_ms = float(np.max(m["call_ms"]))
    per_dev_steps = m["steps"] / N
    bytes_per_ray = BYTES_PER_RAY_DIR if m["info"]["directions_only"] else BYTES_PER_RAY
    traffic, traffic_source, valu = pmc_traffic(a, wl.method) if N == 1 else (None, None, None)
    out = {
        "metric": wl.metric(),
        "value": m["rays"] / (m["dt"] / a.steps) / 1e6,
        "unit": "Mrays/s",
        "ray_steps_per_s": m["steps"] / (m["dt"] / a.steps),
        "n_gpus": N,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": m["dt"] / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if a.workload == "orbit" else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl.describe(W, H, S, N),
            "regime": a.regime, "integrator": "DP5(4) scipy-RK45 controller" if wl.method == "dp54" else "RK4 h=0.1",
            "rtol": 1e-3, "atol": 1e-6, "max_step": (0.1 if a.regime == "fine" else "inf"),
            "rhs_form": a.rhs, "rays_per_gpu": m["rays"] // N, "attempted_steps_per_ray": m["steps"] / m["rays"],
            "tile": a.tile, "tile_order": ("re-dealt by the measured cost of one calibration render, visited longest first" if m["info"]["dealt_by_measured_cost"]
                                           else "cyclic dealing, row-major visit"),
            "trace_output": "exit directions + flags + step counts (25 + 8 B/ray)" if m["info"]["directions_only"] else "end states + flags + step counts (49 + 8 B/ray)",
            "frame_end": "device shade + per-pixel sample mean as float RGBA" + (f" into per-device slabs, ONE gather onto device {devices[0]} by {m['info']['gather']}, assembly kernel" if N > 1 else " in frame order"),
            "collective": (f"{m['info']['gather']} (single-process mode), {N} device(s)") if N > 1 else "none (single device)",
            "parallelism": f"ONE process, {N} device(s) {devices}: the library-owned frame (bhg_frame_*), no torch.distributed",
            "root_gather_assembly_ms": m["root_ms"],
            "root_share": shares[0] if shares else None,
            "trace_call_ms_per_device": [float(v) for v in m["call_ms"]],
        },
        "roofline": roofline_block(wl, per_dev_steps, k_ms, k_ms, m["rays"] // N, bytes_per_ray, traffic,
                                   (traffic_source or "none") + " (replayed: the single-process mode does not start counter passes)" if N == 1 else None,
                                   None if not valu else valu * 64.0 / per_dev_steps),
    }
    if pipelined is not None:
        out["pipelined"] = pipelined
    if strong is not None:
        out["strong"] = strong
    emit(out)
