#!/usr/bin/env python3
"""bench.py -- the null-geodesic hot path on BASELINE.json's headline workload.

A "step" is one pass of the hot path over one frame's worth of rays per GPU: 1024 x 1024 pixels
x 5 samples = 5,242,880 null geodesics (BASELINE.json config 2; camera (1e-4, 0, 30), fov 0.6,
mass 0.5 -> r_s = 1, curve_end 50, directions from the reference's pinhole + MT19937 jitter,
raytracer/RelativisticRenderEngine.py:185-230, seed 42).  Inputs are resident in HBM when the
timed region starts; the timed region is trace + shade/sample-mean and, for N > 1, the single gather of
per-pixel RGBA to rank 0 (asynchronous, overlapping the next frame's trace).

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

Layout of this file: parse() -> Workload (what is traced: parameters, scenes, the strings of the JSON line) -> Runtime
(this process's place in the job) -> measure() (ONE timed region: the only place the headline clock runs) -> the
secondary figures (time_frame, strong_predicted, pipelined_figure, host_buffer_figures, cpu_baseline) -> main().

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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
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
    ap.add_argument("--full-records", action="store_true",
                    help="frame workload: have the trace write whole end states (48 B/ray) instead of the exit directions a sky frame reads")
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
    if a.workload == "orbit" and a.rhs == "kerr":
        ap.error("object spheres are Schwarzschild-only")
    return a


def grid_for(n):
    nx = n
    ny = 1
    while nx % 2 == 0 and nx // 2 >= ny * 2:
        nx //= 2
        ny *= 2
    return nx, ny


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


# ----------------------------------------------------------------------------------------------------------------------
# PMC counters of the dominant kernel, measured in the run itself
# ----------------------------------------------------------------------------------------------------------------------
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
    child = [sys.executable, os.path.abspath(__file__), "--lean", "--live-pmc", "0", "--steps", "3", "--warmup", "1", "--ramp-seconds", "0",
             "--cpu-seconds", "0", "--regime", a.regime, "--rhs", a.rhs, "--workload", a.workload, "--tile", str(a.tile),
             "--order", a.order, "--visit", a.visit, "--lpt", str(a.lpt)]
    for flag, val in (("--width", a.width), ("--height", a.height), ("--samples", a.samples)):
        if val is not None:
            child += [flag, str(val)]
    if a.full_records:
        child.append("--full-records")
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
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.json"))
                   if os.path.basename(f).split("_pmc_summary")[1] == tag + ".json")
    if not files:
        return None, None, None
    try:
        s = json.load(open(files[-1]))
        return s.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(files[-1]), s.get("valu_insts_per_launch")
    except Exception:
        return None, None, None


# ----------------------------------------------------------------------------------------------------------------------
# What is traced
# ----------------------------------------------------------------------------------------------------------------------
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


# ----------------------------------------------------------------------------------------------------------------------
# Two frames in flight: ONE definition of the two-lane setup (measure(overlap=True), time_frame(overlap=True),
# pipelined_figure)
# ----------------------------------------------------------------------------------------------------------------------
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


# ----------------------------------------------------------------------------------------------------------------------
# ONE timed region
# ----------------------------------------------------------------------------------------------------------------------
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
        # a sky frame reads only the exit directions of its rays (background_hit, :366-378): the trace writes those
        # alone (bhg_trace_dir_device) unless --full-records asks for whole end states
        frames.append(DeviceFrame(rt.ctx, W, H, S, fov_x=fov_x, fov_y=fov_y, sampling_seed=42.0, origin=CAM, pixels=pixels,
                                  jitter=jitter, directions_only=(a.workload == "frame" and not a.full_records)))
    for f in frames:
        f.set_sky(sky)
        f.generate_rays()
    return frames, batch


def measure(rt, wl, sky, nx, ny, ramp, overlap=False, whole_frames=False):
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
    for i in range(a.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    barrier(final=True)
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # the frame really is the frame: rank 0's assembled image against a shade of ITS OWN pixels at their places
    # (outside the timed region; catches a slab / pixel-order mismatch)
    if rank == 0 and len(frames) == 1 and not whole_frames:
        img = gatherer.image().reshape(-1, 4)
        own = torch.empty((fr.P, 4), dtype=torch.float32, device="cuda")
        fr.shade_f32(own)
        torch.cuda.synchronize()
        assert torch.equal(img[fr.d_pixels], own), "assembled frame does not hold rank 0's pixels at their places"

    ray_steps = sum(int(f.d_steps.to(torch.int64).sum().item()) for f in frames)
    # per step: the trace calls of all its frames (HIP events on the stream the library launches on)
    n_sampled = len(range(0, a.steps, EV_EVERY))
    if whole_frames:
        n_sampled = max(1, len(kernel_ms))
    call_ms = float(np.sum([e0.elapsed_time(e1) for e0, e1 in kernel_ms])) / max(n_sampled, 1) if kernel_ms else float("nan")
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
                visit=tile_cost.visit, root_share=root_share)


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
    fo.close()
    out["unit"] = "Mrays/s"
    out["what"] = ("bhg_trace, PCIe-inclusive, best of 3 after one warm-up call: k0 from a pageable numpy array (staged by worker "
                   "threads), H2D || trace || D2H pipelined over 2^20-ray chunks; value: end/flags/n_steps/n_accepted arrive in "
                   "page-locked arrays from the library's pool (the Python adaptor's default); pageable_results: into plain numpy arrays")
    return out


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


# ----------------------------------------------------------------------------------------------------------------------
# The JSON line
# ----------------------------------------------------------------------------------------------------------------------
def roofline_block(wl, ray_steps, k_ms, call_ms, n, bytes_per_ray, traffic, traffic_source, valu_per_64):
    achieved_tf = ray_steps * wl.flop / (k_ms * 1e-3) / 1e12
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
    m = measure(rt, wl, sky, nx, ny, a.ramp_seconds)
    strong = frames_sharded = None
    per_step = lambda r: r["dt"] / a.steps    # noqa: E731
    if world > 1 and a.workload != "orbit":
        # BASELINE.json's metric read as strong scaling: ONE fixed frame of the single-GPU size over all ranks.  The
        # headline of the sharded path has two frames in flight per rank (consecutive frames on alternating streams): a
        # shard's persistent launch ends with a tail of a few tens of microseconds, a quarter of a 1/8 shard's kernel --
        # the next frame's first waves fill it.  The sequential figure is reported beside it.
        seq = measure(rt, wl, sky, 1, 1, 0.0)
        head = measure(rt, wl, sky, 1, 1, 0.0, overlap=True) if a.workload == "frame" else seq
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
        wf = measure(rt, wl, sky, 1, 1, 0.0, whole_frames=True)
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
            },
            "roofline": roofline_block(wl, ray_steps, k_ms, call_ms, n, bytes_per_ray, traffic, traffic_source, valu_per_64),
        }
        if strong is not None:
            out["strong"] = strong
        if frames_sharded is not None:
            out["frames_sharded"] = frames_sharded
        t1 = None
        if world == 1 and a.workload == "frame" and getattr(fr, "_dir_traced", False) and not a.lean:
            # what north_star names -- "exit position/direction written back": the same frame with whole end states
            # (81 B/ray) instead of the exit directions a sky frame reads (57 B/ray); same K / W, same clock
            frf = DeviceFrame(rt.ctx, W, H, S, fov_x=fr.fov_x, fov_y=fr.fov_y, sampling_seed=42.0, origin=CAM,
                              pixels=fr.d_pixels.cpu().numpy(), jitter=np.zeros(2), directions_only=False)
            frf.d_k0 = fr.d_k0
            frf.set_sky(sky)
            ms_f, call_f, steps_f = time_frame(frf, wl.params, a.steps, a.warmup, device=rt.local_rank, ramp=a.ramp_seconds)
            out["full_records"] = {"value": n / (ms_f * 1e-3) / 1e6, "unit": "Mrays/s", "ms_per_step": ms_f, "trace_call_ms": call_f,
                                   "frac": steps_f * F / (call_f * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                   "algorithmic_bytes_per_ray": BYTES_PER_RAY,
                                   "what": "the same frame and K / W with whole end states written (x, k: 48 B/ray) and shaded from them"}
            del frf
        if world == 1 and a.workload == "frame" and a.emulate_shards.strip():
            t1_ms, t1_call, _ = time_frame(fr, wl.params, a.steps, a.warmup, device=rt.local_rank, ramp=a.ramp_seconds)
            out["strong_predicted"] = strong_predicted(rt, wl, sky, m, (t1_ms, t1_call))
        if world == 1 and a.workload == "frame" and a.cpu_seconds > 0:   # (--cpu-seconds 0 = kernels only: profiling runs)
            out["pipelined"] = pipelined_figure(rt, fr, wl.params, a)
            out["host_buffer_call"] = host_buffer_figures(rt.ctx, fr, CAM, wl.params, n)
        if a.cpu_seconds > 0 and world == 1:   # the CPU baseline is an N = 1 figure (rank 0's host cores, nothing else running)
            okw = dict(wl.okw)
            if a.workload == "orbit":
                okw["spheres"] = wl.orbit_scene(a.steps - 1)[0]
            out["cpu_baseline"] = cpu_baseline(fr.d_k0.cpu().numpy(), fr.origin, a, okw)
    rt.close()
    if rank == 0:
        emit(out)


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


# ----------------------------------------------------------------------------------------------------------------------
# --single-process: the library-owned frame over N devices of this ONE process
# ----------------------------------------------------------------------------------------------------------------------
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

    def timed(frames):
        def step(i, profile):
            for f in frames:
                if a.workload == "orbit":
                    sp, rgb, lamps = wl.orbit_scene(i)
                    f.set_scene(None, spheres=sp, sphere_rgb=rgb, lamps=lamps)
                f.set_profiling(profile)
                f.render(wl.params, to_host=False)

        def sync():
            for f in frames:
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
        st = [f.stats() for f in frames]
        return dict(dt=dt, call_ms=call_ms, root_ms=root_ms, rays=sum(s["rays"] for s in st), steps=sum(s["attempted_steps"] for s in st),
                    info=frames[0].info())

    nx, ny = grid_for(N) if a.workload != "orbit" else (1, 1)
    frames, W, H, S = build(nx, ny)
    m = timed(frames)
    for f in frames:
        f.close()
    strong = None
    if N > 1 and a.workload != "orbit":
        fs, Ws, Hs, _ = build(1, 1)
        s_ = timed(fs)
        for f in fs:
            f.close()
        strong = {"value": s_["rays"] / (s_["dt"] / a.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": s_["dt"] / a.steps * 1e3,
                  "ray_steps_per_s": s_["steps"] / (s_["dt"] / a.steps), "scaling": "strong", "trace_call_ms_per_device": [float(v) for v in s_["call_ms"]],
                  "root_gather_assembly_ms": s_["root_ms"],
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
    if strong is not None:
        out["strong"] = strong
    emit(out)


if __name__ == "__main__":
    main()
