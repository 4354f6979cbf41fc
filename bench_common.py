"""bench_common.py -- what bench.py and bench_figures.py share: the measurement constants, the Workload (what is traced:
parameters, scenes, the strings of the JSON line), the Runtime (this process's place in the job), the two-frames-in-flight
lanes, the roofline block.  See bench.py."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic flop per attempted ray-step: 6 RHS x 44 + 390 bookkeeping
# Kerr (config 5): the generated Boyer-Lindquist RHS is 90 operations (tools/gen_kerr_rhs.py: the structured
# omega / chi form; the sympy-CSE'd contraction it is checked against has 127), sin / cos / each reciprocal counted as one
# -> 6 x 90 + 390 and 4 x 90 + 78
FLOP_PER_STEP = {("dp54", "christoffel"): 654, ("dp54", "reduced"): 468, ("dp54", "kerr"): 930,
                 ("rk4", "christoffel"): 254, ("rk4", "reduced"): 130, ("rk4", "kerr"): 438}
# What the kernels' own evaluation order of the Christoffel form amounts to under the same counting rules (mul / add 1, FMA 2,
# rcp / rsqrt 1): 35 per RHS evaluation + 1 (r at the step's end) instead of SURVEY's 44 -- reported beside the
# accounting figure as roofline.flop_per_ray_step_executed / frac_executed, never instead of it
FLOP_PER_STEP_EXECUTED = {("dp54", "christoffel"): 6 * 35 + 1 + 390, ("rk4", "christoffel"): 4 * 35 + 1 + 78}
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X vector fp64: 256 CU x 128 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0
BYTES_PER_RAY = 24 + 48 + 1 + 4 + 4  # k0 in; end state, flag, n_steps, n_accepted out
BYTES_PER_RAY_DIR = 24 + 24 + 1 + 4 + 4  # direction-only traces (sky frames): the direction half of the end state

EV_EVERY = 4   # HIP event pairs around the trace call of every 4th timed step (an event pair costs 7-9 us of stream time)
DISK = (4.5, 10.5)   # 0.15 .. 0.35 x ratio 30 (tests/golden disk set; LimitedRelativisticRenderEngine.py:283-286)
CAM = np.array([1e-4, 0.0, 30.0])
DISK_INCLINATIONS_DEG = [85.0, 80.0, 60.0, 30.0, 5.0]


def grid_for(n):
    nx = n
    ny = 1
    while nx % 2 == 0 and nx // 2 >= ny * 2:
        nx //= 2
        ny *= 2
    return nx, ny


class Workload:
    """The configuration BASELINE.json names, as parameters, scenes and the strings of the JSON line."""

    def __init__(self, a):
        from blackhole_geodesic_calculator_amd import _ffi
        self.a = a
        self.method = "rk4" if a.regime == "rk4" else "dp54"
        # oracle-style keyword set; the same dict configures the CPU baseline
        self.okw = dict(r_s=1.0, lambda_end=50.0, max_step=(0.1 if a.regime == "fine" else np.inf), rtol=1e-3, atol=1e-6,
                        h_fixed=0.1, method=1 if self.method == "rk4" else 0, rhs_form={"reduced": 1, "kerr": 2}.get(a.rhs, 0),
                        spin=0.45 if a.rhs == "kerr" else 0.0)
        if a.workload == "disk":
            self.okw.update(lambda_end=80.0, r_exit=40.0, disk_r_in=DISK[0], disk_r_out=DISK[1])
        elif a.workload == "orbit":
            self.okw.update(lambda_end=80.0, r_exit=40.0)
        self.params = _ffi.make_params(**self.okw)
        self.flop = FLOP_PER_STEP[(self.method, a.rhs)]
        self.flop_executed = FLOP_PER_STEP_EXECUTED.get((self.method, a.rhs), self.flop)
        self.metric_name = a.rhs == "kerr" and "Kerr" or "Schwarzschild"

    @staticmethod
    def orbit_scene(i):
        # config 4: a sphere of radius 1.5 on a circular orbit of radius 8 r_s, inclined 20 degrees to the line of
        # sight plane, one revolution per 100 frames; lit by one lamp beside the camera
        ph = 2.0 * np.pi * (i % 100) / 100.0
        tilt = np.radians(70.0)
        c = 8.0 * np.array([np.cos(ph), np.sin(ph) * np.cos(tilt), np.sin(ph) * np.sin(tilt)])
        return [[c[0], c[1], c[2], 1.5]], [[1.0, 0.85, 0.7]], [[10.0, 10.0, 30.0, 30.0]]

    @staticmethod
    def disk_cameras():
        # five inclinations of a camera at r = 30 looking at the hole (rotation about y by the inclination)
        return [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in np.radians(DISK_INCLINATIONS_DEG)]

    def shadow_edge_cost(self, W, H):
        """tile_cost(cx, cy): steps per ray peak at the shadow edge (impact parameter b_c = 2.6 r_s -> radius
        b_c / |cam| / fov * width pixels around the frame centre); the model only holds for the plain frame."""
        def tile_cost(cx, cy):
            ax, ay = 0.6 * (cx - W / 2) / W, 0.6 * (cy - H / 2) / H
            return -abs(np.hypot(ax, ay) - 2.598 / 30.0)
        return tile_cost

    def metric(self):
        a = self.a
        if a.workload == "frame":
            return f"Mrays/s (null geodesics traced to curve_end or horizon), 1024x1024x5 {self.metric_name} frame per GPU"
        return {"disk": f"Mrays/s, 1024x1024 {self.metric_name} + thin disk, 5 camera inclinations per step",
                "orbit": "Mrays/s, 2048x2048x16 orbiting-sphere animation frame"}[a.workload]

    def describe(self, W, H, S, world):
        a = self.a
        hole = "Kerr a/M=0.9" if a.rhs == "kerr" else "Schwarzschild"
        if a.workload == "frame":
            return (f"BASELINE.json configs[{4 if a.rhs == 'kerr' else 1}]: {a.width}x{a.height} x{S} multisample {hole} frame per GPU "
                    f"(frame {W}x{H} over {world} GPU(s)), camera (1e-4,0,30), fov 0.6, r_s=1, curve_end=50")
        if a.workload == "disk":
            return (f"BASELINE.json configs[2]: {a.width}x{a.height} x{S} {hole} + thin disk {DISK[0]}..{DISK[1]} r_s, camera r=30 at "
                    f"inclinations 85/80/60/30/5 deg (5 frames per step, one trace call with per-ray origins, shaded per frame), "
                    f"fov 0.9, exit sphere 40, curve_end 80; frame {W}x{H} over {world} GPU(s)")
        return (f"BASELINE.json configs[3]: {W}x{H} x{S} frame of the orbiting-sphere animation (sphere radius 1.5 on an r=8 orbit, "
                f"new position every step, lamp-lit), tiles sharded over {world} GPU(s); camera (1e-4,0,30), fov 0.6, exit sphere 40, "
                f"curve_end 80")

    def tile_order_text(self, visit):
        a = self.a
        if not a.lpt:
            return "row-major"
        if a.order == "measured":
            return "longest first by the attempted steps of an untimed calibration trace"
        if a.order == "model" and a.workload == "frame":
            return "dealt by the shadow-edge model, visited " + ("row-major" if visit == "row" else "longest first")
        return "row-major"


class Runtime:
    """This process's place in the job: rank / world, the process group (if any), its library context."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # development aid: BHGEO_BENCH_BACKEND=gloo exercises the N > 1 code path with several ranks on ONE GPU
        # (RCCL refuses two ranks per device); never used by the driver's runs
        self.backend = os.environ.get("BHGEO_BENCH_BACKEND", "nccl")
        if self.backend != "nccl":
            self.local_rank = self.local_rank % max(torch.cuda.device_count(), 1)
        if self.world != a.gpus and self.world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        torch.cuda.set_device(self.local_rank)
        self.force_collective = os.environ.get("BHGEO_FORCE_COLLECTIVE", "0") == "1"
        self.group_up = False
        if self.world > 1 or self.force_collective:
            self.init_group()
        self.collective = self.world > 1 or self.force_collective
        from blackhole_geodesic_calculator_amd import _ffi
        self.ctx = _ffi.Context(self.local_rank)
        self.stream = torch.cuda.current_stream()

    def init_group(self):
        """The process group (RCCL, or gloo as the development aid); a one-rank group for a single process."""
        if self.group_up:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if self.backend == "nccl":
            self.dist.init_process_group("nccl", device_id=self.torch.device("cuda", self.local_rank))
        else:
            self.dist.init_process_group(self.backend)
        self.group_up = True

    def close(self):
        if self.group_up:
            if self.world > 1:
                self.dist.barrier()
            self.dist.destroy_process_group()
            self.group_up = False

    def assemble(self, slabs, perm, frame):   # rank 0, N > 1: slabs -> frame order in one kernel
        self.ctx.assemble_frame_f32_device(slabs.data_ptr(), perm.data_ptr(), frame.shape[0], frame.data_ptr(),
                                           stream=self.torch.cuda.current_stream().cuda_stream)


def twin_of(fr_, ctx2):
    """A second DeviceFrame over the SAME rays (shared d_k0) with result buffers of its own, on another library
    context (its own work counters): consecutive frames of an animation are independent, so frame i + 1 can be
    traced on a second stream while frame i's last waves drain."""
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
    n_ = fr_.n
    buf = (fr_.d_k0, None if fr_.directions_only else torch.empty((n_, 6), dtype=torch.float64, device="cuda"),
           torch.empty(n_, dtype=torch.uint8, device="cuda"), torch.empty(n_, dtype=torch.int32, device="cuda"),
           torch.empty(n_, dtype=torch.int32, device="cuda"))
    f2 = DeviceFrame(ctx2, fr_.W, fr_.H, fr_.S, fov_x=fr_.fov_x, fov_y=fr_.fov_y, origin=fr_.origin,
                     pixels=None if fr_.d_pixels is None else fr_.d_pixels.cpu().numpy(), jitter=np.zeros(2), buffers=buf,
                     directions_only=fr_.directions_only)
    f2.d_sky, f2.sky_wh = fr_.d_sky, fr_.sky_wh
    return f2


class Lanes:
    """[(frame, stream)] that consecutive steps alternate between.  One lane: the frame on the current stream.  Two: the
    frame and its twin (same rays, own result buffers, own library context) on two streams of DIFFERENT priority -- two
    streams of the same priority share one hardware queue on this ROCm build (rocprofv3 shows one queue id and strictly
    serial kernels, the pair measures exactly like one stream); a stream of another priority gets a queue of its own, and
    only then do the second frame's first waves start while the first frame's last ones drain."""

    def __init__(self, fr, device, two):
        import torch
        from blackhole_geodesic_calculator_amd import _ffi
        self.ctx2 = _ffi.Context(device) if two else None
        if two:
            self.lanes = [(fr, torch.cuda.Stream()), (twin_of(fr, self.ctx2), torch.cuda.Stream(priority=-1))]
        else:
            self.lanes = [(fr, torch.cuda.current_stream())]

    def __len__(self):
        return len(self.lanes)

    def __getitem__(self, i):
        return self.lanes[i % len(self.lanes)]

    def close(self):
        if self.ctx2 is not None:
            self.ctx2.close()
            self.ctx2 = None


def ramp_clocks(run, seconds):
    """Untimed frames until `seconds` of wall time have passed: the GPU's clocks take tens of milliseconds of load to
    settle, and every secondary figure of the line starts after host-side work during which the GPU sat idle (a
    K = 20 / W = 3 run read 5-10 % low without this)."""
    import torch
    t = time.perf_counter()
    while seconds > 0 and time.perf_counter() - t < seconds:
        run(4)
        torch.cuda.synchronize()


def traced_with_events(f, params, stream, sink):
    """f.trace(params) between two HIP events recorded on the stream the library launches on."""
    import torch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    f.trace(params)
    e1.record(stream)
    sink.append((e0, e1))


class ClockSampler:
    """The shader clock while the timed region runs, if the box exposes it WITHOUT a subprocess (a child process after
    this one has initialised the GPU is what the pool forbids; reading a sysfs file is not): hwmon's freq1_input (Hz) of
    the card whose unique_id / index matches, sampled every few milliseconds from a thread -- the main thread sits in a
    blocking synchronise for nearly all of the region, so the sampler costs it nothing.  None where nothing is readable."""

    def __init__(self, device_index=0, period_s=0.004):
        import glob
        self.period = period_s
        self.samples = []
        self.path = None
        self._stop = False
        self._t = None
        # the card of THIS device: by PCI address (a box shows all eight cards of its node in sysfs, one of them is ours)
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*/freq1_input"))
        for c in cards:
            dev = os.path.realpath(c.split("/hwmon/")[0])
            if want is not None and os.path.basename(dev).lower() == want.lower():
                self.path = c
        self.pci = want

    # what else hwmon shows of the same card, sampled at the same instants when readable (round 6: a box can be slow in
    # something other than the shader clock): memory clock, socket power, junction temperature
    EXTRA = {"mclk_mhz": ("freq2_input", 1e-6), "power_w": ("power1_average", 1e-6), "power_input_w": ("power1_input", 1e-6),
             "temp_junction_c": ("temp2_input", 1e-3)}

    def _read(self, path=None, scale=1e-6):
        try:
            with open(path or self.path) as f:
                return float(f.read().strip()) * scale     # Hz -> MHz
        except Exception:
            return None

    def usable(self):
        if os.environ.get("BHGEO_NO_CLOCK_SAMPLER") == "1":
            self.path = None
        if self.path is not None and self._read() is None:
            self.path = None
        return self.path is not None

    def read_once(self):
        """ONE read of the shader clock (MHz) -- no thread; None where nothing is readable."""
        return self._read() if self.path is not None else None

    def start(self):
        """(Re)start the sampling thread: a sampler is built ONCE, before the warm-up steps (its construction globs sysfs and
        asks torch for the device properties: milliseconds of host time during which the GPU would sit idle right in front
        of a timed region), and started only for the repetition it samples."""
        if not self.usable():
            return self
        import threading
        self.samples, self._stop = [], False
        d = os.path.dirname(self.path)
        if not hasattr(self, "extra_paths"):
            self.extra_paths = {k: (os.path.join(d, f), sc) for k, (f, sc) in self.EXTRA.items()
                                if self._read(os.path.join(d, f), sc) is not None}
        self.extra = {k: [] for k in self.extra_paths}

        def loop():
            while not self._stop:
                v = self._read()
                if v is not None:
                    self.samples.append(v)
                for k, (pth, sc) in self.extra_paths.items():
                    x = self._read(pth, sc)
                    if x is not None:
                        self.extra[k].append(x)
                time.sleep(self.period)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()
        return self

    def stop(self):
        self._stop = True
        if self._t is not None:
            self._t.join(timeout=1.0)
        if not self.samples:
            return None
        v = np.array(self.samples)
        out = {"mean_mhz": float(v.mean()), "min_mhz": float(v.min()), "max_mhz": float(v.max()), "samples": int(len(v)),
               "source": self.path, "pci": self.pci}
        for k, xs in getattr(self, "extra", {}).items():
            if xs:
                out[k] = float(np.mean(xs))
        return out


def spread(xs):
    """min / median / max (and their relative spread) of a list of samples."""
    v = np.asarray([x for x in xs if x is not None and np.isfinite(x)], dtype=float)
    if v.size == 0:
        return None
    return {"n": int(v.size), "min": float(v.min()), "median": float(np.median(v)), "max": float(v.max()),
            "rel_spread": float((v.max() - v.min()) / np.median(v))}


MIN_CLOCK_SAMPLES = 16   # below this the clock-scaled peak is not reported (a 28-ms region gives 7 samples at 4 ms)


def merge_clock_samples(sclks):
    """The ClockSampler summaries of the repetitions as one (all samples pooled)."""
    got = [c for c in sclks if isinstance(c, dict)]
    if not got:
        return None
    n = sum(c["samples"] for c in got)
    out = {"mean_mhz": sum(c["mean_mhz"] * c["samples"] for c in got) / n, "min_mhz": min(c["min_mhz"] for c in got),
           "max_mhz": max(c["max_mhz"] for c in got), "samples": int(n), "repetitions": len(got),
           "source": got[0]["source"], "pci": got[0]["pci"]}
    for k in ClockSampler.EXTRA:
        xs = [c[k] for c in got if k in c]
        if xs:
            out[k] = float(np.mean(xs))
    return out


def run_probes(ctx, device_index=0):
    """bhg_peak_probe, both kinds (2-ms launches, median of five each): what THIS box's fp64 pipe delivers.  (Sampling the
    sysfs clock WHILE a probe runs was tried and dropped: reads every 0.5 ms slow the first probe by 11 %,
    profiles/r05_sampler_ab.log; the 4-ms sampling of the timed region measures neutral.)"""
    from blackhole_geodesic_calculator_amd import _ffi
    fma = ctx.peak_probe(_ffi.PROBE_FMA, 2.0)
    mix = ctx.peak_probe(_ffi.PROBE_STEP_MIX, 2.0)
    return {"fp64_fma_tflops": fma["tflops"], "fp64_fma_clock_mhz": fma["fp64_full_rate_clock_mhz"],
            "fma_wave_insts_per_s": fma["valu_wave_insts"] / (fma["ms"] * 1e-3),
            "step_mix_tflops": mix["tflops"], "step_mix_wave_insts_per_s": mix["valu_wave_insts"] / (mix["ms"] * 1e-3)}


def hbm_copy_probe(nbytes=512 << 20, reps=5):
    """Device-to-device copy rate of this box (torch copy_, HIP events; read + write bytes per second): the one thing the
    compute probes do not see -- a box whose memory side is slow."""
    import torch
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return 2.0 * nbytes / (float(np.median(ms)) * 1e-3) / 1e9


def calibration_block(cal, sclk, achieved_tf, valu_per_64, ray_steps, k_ms, num_cus=256):
    """roofline.calibration: the box's own peak beside the vendor's.  cal = {"before": run_probes(), "after": run_probes()}
    around the timed region, or "after" alone (the default since round 6)."""
    if not cal or not ("before" in cal or "after" in cal):
        return None
    b = cal.get("before") or cal["after"]
    a_ = cal.get("after") or cal["before"]
    where = " and ".join(w for w in ("before", "after") if w in cal)
    n_probes = len([w for w in ("before", "after") if w in cal])
    fma = 0.5 * (b["fp64_fma_tflops"] + a_["fp64_fma_tflops"])
    mix_rate = 0.5 * (b["step_mix_wave_insts_per_s"] + a_["step_mix_wave_insts_per_s"])
    out = {
        "fp64_fma_tflops_measured": fma,
        "fp64_fma_tflops_before_after": [b["fp64_fma_tflops"], a_["fp64_fma_tflops"]],
        "fp64_fma_frac_of_vendor_peak": fma / PEAK_FP64_VALU_TFLOPS,
        "sclk_mhz_implied_by_fma_probe": 0.5 * (b["fp64_fma_clock_mhz"] + a_["fp64_fma_clock_mhz"]),
        # the issue ceiling of a kernel with the step loop's instruction mix (16 quarter-rate v_rcp_f64 / v_rsq_f64 per 503):
        # wave-level VALU instructions per second the box sustains on that mix, and the same as TFLOP/s under SURVEY's
        # counting rule (FMA 2, rcp / rsq 1)
        "issue_bound_wave_insts_per_s": mix_rate,
        "issue_bound_tflops": 0.5 * (b["step_mix_tflops"] + a_["step_mix_tflops"]),
        "frac_of_measured_peak": achieved_tf / fma,
        "sclk_mhz_timed_region": sclk if sclk is not None else "omitted: no hwmon freq1_input readable for this device on this box",
        "method": f"bhg_peak_probe (include/bhgeo.h) in this process, {where} the timed region (outside its clock), in the trace "
                  "kernels' launch geometry (1 wave64 per workgroup, 12 waves per CU, no memory traffic): 2-ms launches, median of 5" +
                  ("; figures are the mean of the two probes" if n_probes > 1 else ""),
        "probes": where,
    }
    if "hbm_copy_GBps" in cal:
        out["hbm_copy_GBps"] = cal["hbm_copy_GBps"]      # (512-MiB device-to-device copy after the timed region, read + write)
    if isinstance(sclk, dict) and sclk["samples"] < MIN_CLOCK_SAMPLES:
        out["sclk_mhz_timed_region"] = {**sclk, "note": f"fewer than {MIN_CLOCK_SAMPLES} samples: no clock-scaled peak derived"}
        sclk = None
    if isinstance(sclk, dict):
        # ... and against the vendor's formula at the clock the timed region actually ran at (num_cus CUs x 128 flop per clock):
        # the pure-FMA probe draws more power than the trace kernel and sustains a lower clock than the timed region's
        peak_at_clock = sclk["mean_mhz"] * 1e6 * 128.0 * num_cus / 1e12
        out["peak_tflops_at_timed_region_clock"] = peak_at_clock
        out["frac_at_timed_region_clock"] = achieved_tf / peak_at_clock
    if valu_per_64:
        # the trace kernel's own instruction stream: wave-level VALU instructions per second, against what the probes issue on
        # this box.  (Each kernel runs at the clock the box chooses for it -- the pure-FMA probe draws the most power and
        # clocks lowest -- so a kernel with few stalls can read slightly above 1 against the FMA probe.)
        rate = valu_per_64 * (ray_steps / 64.0) / (k_ms * 1e-3)
        fma_rate = 0.5 * (b["fma_wave_insts_per_s"] + a_["fma_wave_insts_per_s"])
        out["fma_probe_wave_insts_per_s"] = fma_rate
        out["trace_kernel_wave_insts_per_s"] = rate
        out["valu_issue_utilisation"] = rate / fma_rate                    # of a pure v_fma_f64 stream
        out["valu_issue_utilisation_vs_step_mix"] = rate / mix_rate        # of the DP5(4) Christoffel step loop's own mix
        if isinstance(sclk, dict):
            # per clock and SIMD (256 CUs x 4; a full-rate wave64 fp64 instruction occupies its SIMD for 4 clocks: 0.25 at best)
            out["trace_kernel_wave_insts_per_clock_per_simd"] = rate / (sclk["mean_mhz"] * 1e6 * 4.0 * num_cus)
    return out


def roofline_block(wl, ray_steps, k_ms, call_ms, n, bytes_per_ray, traffic, traffic_source, valu_per_64, calibration=None,
                   sclk=None, call_samples=None, share=None, num_cus=256, sampled_rep_ms=None):
    achieved_tf = ray_steps * wl.flop / (k_ms * 1e-3) / 1e12
    cal = calibration_block(calibration, sclk, achieved_tf, valu_per_64, ray_steps, k_ms, num_cus=num_cus)
    extra = {} if cal is None else {"frac_of_measured_peak": cal["frac_of_measured_peak"], "calibration": cal}
    if cal is not None and sampled_rep_ms is not None:
        cal["sampled_repetition_ms_per_step"] = sampled_rep_ms      # the extra repetition the sysfs thread sampled: what sampling costs
        cal["sclk_note"] = ("sclk_mhz_timed_region comes from ONE extra repetition behind the headline's (a sysfs thread, 4-ms period), which "
                            "does not count: the sampler's reads slow a region by 5-6 % while they run (profiles/r06_matrix_b.log)")
    if call_samples:
        # every HIP-event sample of the timed region(s), not their mean: the trace call of every EV_EVERY-th step of each
        # repetition (ms) times the trace kernel's share of a call (1 for the Schwarzschild forms: one launch per call)
        sh = 1.0 if share is None else share
        k_samples = [[x * sh for x in r_] for r_ in call_samples]
        flat = [x for r_ in k_samples for x in r_]
        extra["kernel_ms_samples"] = k_samples
        extra["kernel_ms_spread"] = spread(flat)
        extra["kernel_ms_mean"] = float(np.mean(flat)) if flat else None
        extra["kernel_ms_is"] = (f"median of the {len(flat)} HIP-event samples (every {EV_EVERY}th step of each repetition, on the stream the "
                                 f"library launches on) x the trace kernel's share of a call ({sh:.4f})")
    return {**_roofline_core(wl, ray_steps, k_ms, call_ms, n, bytes_per_ray, traffic, traffic_source, valu_per_64, achieved_tf), **extra}


def _roofline_core(wl, ray_steps, k_ms, call_ms, n, bytes_per_ray, traffic, traffic_source, valu_per_64, achieved_tf):
    return {
        "bound": "valu_fp64",
        "kernel": f"trace_{wl.method}_kernel<{wl.a.rhs}>: ONE launch per trace call integrates every ray to its end "
                  f"(step loop + in-kernel event location and resumption)",
        "achieved": achieved_tf,
        "peak": PEAK_FP64_VALU_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved_tf / PEAK_FP64_VALU_TFLOPS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        # wave-level VALU instructions the whole launch issues (SQ_INSTS_VALU) per 64 attempted ray-steps: step loop +
        # setup + pop + events; the instruction-stream ceiling is F*64 / (2*64*this)
        "valu_insts_per_64_ray_steps": valu_per_64,
        "flop_per_ray_step": wl.flop,
        "flop_per_ray_step_executed": wl.flop_executed,
        "frac_executed": achieved_tf / PEAK_FP64_VALU_TFLOPS * wl.flop_executed / wl.flop,
        "ray_steps_per_launch": ray_steps,
        "kernel_ms": k_ms,
        "trace_call_ms": call_ms,
        "hbm_algorithmic_GBps": n * bytes_per_ray / (k_ms * 1e-3) / 1e9,
        "hbm_frac": n * bytes_per_ray / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        "algorithmic_bytes_per_ray": bytes_per_ray,
    }


def emit(out):
    # the JSON line goes out LAST: RCCL writes a version banner through C stdio, which sits in libc's buffer
    # (stdout is a pipe under the driver) until it is flushed -- flush it first
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
