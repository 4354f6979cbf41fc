#!/usr/bin/env python3
"""first_node_run.py -- everything the first lease of an 8-GPU node is wanted for, in ONE command.

No multi-GPU node has been available to this build: the distinct-device branches of the frame end (hipDeviceEnablePeerAccess,
hipMemcpyPeerAsync, N-rank grouped ncclSend / ncclRecv: csrc/bhgeo_frame.hip) and the torch.distributed gather over RCCL
with N > 1 ranks have never executed.  The parallelism the reference wanted is the commented-out mp.Pool at
raytracer/RelativisticRenderEngine.py:210-216.  This script extracts maximum evidence from a single lease: every step runs
in a FRESH child process (this parent never touches the GPU, nothing is exec'ed over a process that did), writes ONE JSON
line to gpurun_out/node_run.jsonl, and a failure does not stop the steps after it; the exit code is non-zero if any failed.

  (i)   python bench.py --gpus N                      N = 1, 2, 4, 8: one process per GPU, torch.distributed over RCCL
        (+ --workload disk and --workload orbit at the largest N: BASELINE configs 3 and 4)
  (ii)  python bench.py --single-process --gpus N     N = 2, 4, 8 with --frame-gather rccl | copy | peer (N = 1: auto)
  (iii) examples/render_frame.c on devices 0,1,..,N-1 (plain C through the C ABI; compiled here with gcc)
  (iv)  the N-device image against the 1-device image, bit for bit (library-owned frame, every gather mode)
  (v)   the same through torch.distributed: N ranks' gathered frame (disk + objects + exit sphere) == the one-rank frame

Usage:  python scripts/first_node_run.py [--gpus 8] [--steps 100] [--warmup 10] [--out gpurun_out/node_run.jsonl]
        --standin   the same plan on a ONE-GPU box: torch.distributed ranks over gloo sharing the GPU
                    (BHGEO_BENCH_BACKEND=gloo), device lists that repeat device 0 (BHGEO_DEVICES=0,0,..); gather modes
                    RCCL / peer need distinct devices and are skipped.  What tests/test_gpu_rccl.py runs.
        --quick     the N = 1 bench line without the CPU baseline and the host-buffer figures (tests)
        --plan      print the steps as JSON and run nothing (tests/test_host.py checks the plan on the CPU)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "blackhole_geodesic_calculator_amd")

BIT_IDENTITY = r"""
import json, sys
import numpy as np
sys.path.insert(0, {root!r})
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.raygen import python_random_stream
from blackhole_geodesic_calculator_amd.sky import synthetic_sky
devs, mode, W, H, S = {devs!r}, {mode!r}, 1024, 1024, 5
gm = dict(auto=_ffi.GATHER_AUTO, copy=_ffi.GATHER_COPY, rccl=_ffi.GATHER_RCCL, peer=_ffi.GATHER_PEER)[mode]
jit = python_random_stream(42.0, 2 * S * W * H)
sky = synthetic_sky(2048, 1024)
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
imgs, infos = [], []
for d, g in (([devs[0]], _ffi.GATHER_AUTO), (devs, gm)):
    fr = _ffi.Frame(d, W, H, S, fov_x=0.6, fov_y=0.6, jitter=jit, gather=g)
    fr.set_scene(sky)
    imgs.append(fr.render(p).copy())
    if len(d) > 1:
        fr.rebalance(root_share=0.8)             # re-dealt by measured cost, the first device a smaller part: the same image
        imgs.append(fr.render(p).copy())
    infos.append(fr.info())
    fr.close()
same = all(np.array_equal(imgs[0], im) for im in imgs[1:])
print(json.dumps(dict(bit_identical=bool(same), images=len(imgs), gather=infos[-1]["gather"], n_devices=infos[-1]["n_devices"],
                      max_abs_diff=float(max(np.abs(imgs[0] - im).max() for im in imgs[1:])))))
sys.exit(0 if same else 1)
"""


def plan(a):
    """[{name, cmd, env, parse}]: the steps in order."""
    py = sys.executable
    Ns = [n for n in (1, 2, 4, 8) if n <= a.gpus]
    common = ["--steps", str(a.steps), "--warmup", str(a.warmup)]
    steps = []
    # what the node looks like: devices, xGMI links between them, NUMA placement (informational: never fails the run)
    steps.append(dict(name="topology", kind="info", env={}, cmd=["bash", "-c", "rocm-smi --showtopo 2>&1 | tail -60; rocm-smi --showuniqueid 2>&1 | tail -12"]))
    for n in Ns:
        env = {}
        if a.standin and n > 1:
            env["BHGEO_BENCH_BACKEND"] = "gloo"
        # (the N = 1 line of the default command carries strong_predicted: the prediction the N > 1 runs are compared with)
        steps.append(dict(name=f"dist_n{n}", kind="bench", n=n, cmd=[py, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common +
                          (["--cpu-seconds", "0", "--live-pmc", "0"] if n > 1 else
                           (["--live-pmc", "0", "--cpu-seconds", "0", "--emulate-shards", ",".join(str(v) for v in Ns[1:])] if a.quick else ["--live-pmc", "0"])),
                          env=env))
    # ... and BASELINE's other sharded configurations at the largest N: config 3 (disk, five inclinations per step) and config 4
    # (the 2048 x 2048 x 16 orbit frame: ONE frame's tiles over all ranks + whole frames round-robin)
    nmax = Ns[-1]
    if nmax > 1:
        env = {"BHGEO_BENCH_BACKEND": "gloo"} if a.standin else {}
        small = ["--width", "256", "--samples", "2"] if a.quick else []
        for wl_, sw in (("disk", common), ("orbit", ["--steps", str(max(2, a.steps // 4)), "--warmup", "2"])):   # (an orbit step is 16 x a frame step)
            steps.append(dict(name=f"dist_n{nmax}_{wl_}", kind="bench", n=nmax, env=env,
                              cmd=[py, os.path.join(ROOT, "bench.py"), "--gpus", str(nmax), "--workload", wl_] + sw + small +
                                  ["--cpu-seconds", "0", "--live-pmc", "0"]))
    for n in Ns:
        modes = ["auto"] if n == 1 else (["copy"] if a.standin else ["rccl", "copy", "peer"])
        for mode in modes:
            env = {"BHGEO_DEVICES": ",".join(["0"] * n)} if a.standin else {}
            steps.append(dict(name=f"single_n{n}_{mode}", kind="bench", n=n, gather=mode,
                              cmd=[py, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", str(n), "--frame-gather", mode] + common, env=env))
    exe = os.path.join(ROOT, "build", "render_frame")
    steps.append(dict(name="build_render_frame", kind="build",
                      cmd=["gcc", "-std=c99", "-Wall", "-Wextra", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "render_frame.c"), "-L", PKG, "-lbhgeo", "-Wl,-rpath," + PKG, "-lm", "-o", exe], env={}))
    for n in Ns:
        devs = ",".join(["0"] * n) if a.standin else ",".join(str(i) for i in range(n))
        steps.append(dict(name=f"c_example_n{n}", kind="c_example", n=n, cmd=[exe, devs], env={}))
    for n in Ns[1:]:
        for mode in (["copy"] if a.standin else ["rccl", "copy", "peer"]):
            devs = [0] * n if a.standin else list(range(n))
            steps.append(dict(name=f"bit_identity_n{n}_{mode}", kind="bit_identity", n=n, gather=mode,
                              cmd=[py, "-c", BIT_IDENTITY.format(root=ROOT, devs=devs, mode=mode)], env={}))
    # (v) the torch.distributed form of the same question: N ranks (one GPU each, RCCL) gather a frame with disk, objects and
    # exit sphere that must equal the one-rank frame bit for bit (scripts/dist_bit_identity_worker.py)
    for n in Ns[1:]:
        env = {} if a.standin else {"BHG_DISTINCT": "1"}
        steps.append(dict(name=f"dist_bit_identity_n{n}", kind="dist_bit_identity", n=n, env=env,
                          cmd=[py, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                               "--master-port", str(29600 + n), os.path.join(ROOT, "scripts", "dist_bit_identity_worker.py")]))
    return steps


def last_json(text):
    for line in reversed((text or "").splitlines()):
        s = line.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                return json.loads(s)
            except Exception:
                continue
    return None


def summarise(step, rc, out, err, seconds, predicted):
    """One JSON record per step: what ran, how it ended and the figures the next reader wants first."""
    rec = dict(step=step["name"], kind=step["kind"], rc=rc, seconds=round(seconds, 2), cmd=" ".join(step["cmd"][:6]) + (" ..." if len(step["cmd"]) > 6 else ""),
               env=step["env"])
    if rc != 0:
        rec["stderr_tail"] = (err or "")[-600:]
    j = last_json(out)
    if step["kind"] == "bench":
        if j is None or "metric" not in j:
            rec["error"] = "no bench line"
            if rc == 0:
                rec["rc"] = 1
            return rec
        cfg, n = j.get("config", {}), step["n"]
        rec.update(n_gpus=j.get("n_gpus"), value_mrays_s=j.get("value"), ms_per_step=j.get("ms_per_step"), scaling=j.get("scaling"),
                   collective=cfg.get("collective"), root_share=cfg.get("root_share"), frac=(j.get("roofline") or {}).get("frac"),
                   frac_of_measured_peak=(j.get("roofline") or {}).get("frac_of_measured_peak"),
                   frac_at_timed_region_clock=((j.get("roofline") or {}).get("calibration") or {}).get("frac_at_timed_region_clock"))
        # what tells "one box is slow" from "rank 0's root share is wrong" (round 6): every repetition of the timed region, every
        # HIP-event sample of the trace kernel, the clock read while each repetition ran, and every RANK's own clock
        rf = j.get("roofline") or {}
        rec.update(ms_per_step_samples=j.get("ms_per_step_samples"), ms_per_step_spread=j.get("ms_per_step_spread"),
                   sclk_mhz_per_repetition=j.get("sclk_mhz_per_repetition"), kernel_ms=rf.get("kernel_ms"),
                   kernel_ms_samples=rf.get("kernel_ms_samples"), kernel_ms_spread=rf.get("kernel_ms_spread"),
                   per_rank=j.get("per_rank"))
        if j.get("per_rank"):
            med = [r_["ms_per_step_median"] for r_ in j["per_rank"]]
            rec["slowest_rank"] = int(max(range(len(med)), key=med.__getitem__))
            rec["rank_spread"] = (max(med) - min(med)) / max(min(med), 1e-12)
        # RCCL ranks seen: "rccl gather, N rank(s)" (torch.distributed form) / "rccl (single-process mode), N device(s)"
        col = str(cfg.get("collective") or "")
        rec["rccl_ranks_seen"] = n if ("rccl" in col.lower() and str(n) in col) else 0
        if j.get("n_gpus") != n:
            rec["error"] = f"the line says n_gpus = {j.get('n_gpus')}, the step asked for {n}"
            rec["rc"] = rec["rc"] or 1
        st = j.get("strong")
        if st:
            t1 = predicted.get("T1_ms_per_step_two_in_flight") or predicted.get("T1_ms_per_step")
            rec["strong"] = dict(value_mrays_s=st.get("value"), ms_per_step=st.get("ms_per_step"), root_share=st.get("root_share"),
                                 frames_in_flight=st.get("frames_in_flight"))
            if t1 and st.get("ms_per_step"):
                rec["strong"]["efficiency_measured"] = t1 / (n * st["ms_per_step"])
            pr = (predicted.get("shards") or {}).get(str(n))
            if pr:
                rec["strong"]["efficiency_rank0_predicted"] = pr.get("efficiency_rank0")
                rec["strong"]["efficiency_shard_alone_predicted"] = pr.get("efficiency_two_in_flight")
        if j.get("strong_predicted"):
            predicted.clear()
            predicted.update(j["strong_predicted"])
            rec["strong_predicted_rank0"] = {k: v.get("efficiency_rank0") for k, v in j["strong_predicted"].get("shards", {}).items()}
    elif step["kind"] == "info":
        rec["rc"] = 0                      # (informational)
        rec["output_tail"] = (out or "")[-3000:]
    elif step["kind"] == "c_example":
        head = (out or "").split("\n", 1)[0]
        rec["head"] = head[:300]
        rec["signature"] = head.split(":", 1)[1].strip() if ":" in head else None
    elif step["kind"] == "dist_bit_identity":
        ok = "MULTIRANK_OK" in (out or "")
        rec["bit_identical"] = ok
        line = next((l for l in (out or "").splitlines() if "MULTIRANK_OK" in l), "")
        rec["detail"] = line[:300]
        rec["rccl_ranks_seen"] = step["n"] if "'backend': 'nccl'" in line else 0
        if not ok and rc == 0:
            rec["rc"], rec["error"] = 1, "the worker did not report MULTIRANK_OK"
    elif step["kind"] == "bit_identity":
        if j is not None:
            rec.update(j)
        elif rc == 0:
            rec["rc"], rec["error"] = 1, "no result line"
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "node_run.jsonl"))
    ap.add_argument("--standin", action="store_true")
    ap.add_argument("--plan", action="store_true")
    ap.add_argument("--quick", action="store_true", help="the N = 1 line without CPU baseline and host-buffer figures (tests)")
    ap.add_argument("--timeout", type=float, default=900.0, help="per step, seconds")
    ap.add_argument("--only", default="", help="comma-separated step-name prefixes (default: all)")
    a = ap.parse_args(argv)
    steps = plan(a)
    if a.only:
        keep = tuple(s for s in a.only.split(",") if s)
        steps = [s for s in steps if s["name"].startswith(keep) or (s["kind"] == "build" and any(k.startswith("c_example") for k in keep))]
    if a.plan:
        print(json.dumps([dict(name=s["name"], kind=s["kind"], cmd=s["cmd"][:3] + (["<inline>"] if s["kind"] == "bit_identity" else s["cmd"][3:]), env=s["env"]) for s in steps]))
        return 0
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    failed, predicted, c_sigs = [], {}, {}
    with open(a.out, "w") as log:
        for s in steps:
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.update(s["env"])
            t = time.time()
            try:
                r = subprocess.run(s["cmd"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=a.timeout)
                rc, out, err = r.returncode, r.stdout, r.stderr
            except subprocess.TimeoutExpired as e:
                rc, out, err = 124, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), f"timed out after {a.timeout:.0f} s"
            except OSError as e:
                rc, out, err = 127, "", f"{type(e).__name__}: {e}"
            rec = summarise(s, rc, out, err, time.time() - t, predicted)
            if s["kind"] == "c_example" and rec.get("signature"):
                c_sigs[s["n"]] = rec["signature"]
                # rays, steps, checksum of the image: the same however many devices rendered it
                if 1 in c_sigs and c_sigs[1] != rec["signature"]:
                    rec["rc"], rec["error"] = rec["rc"] or 1, f"the {s['n']}-device image differs from the 1-device image"
            log.write(json.dumps(rec) + "\n")
            log.flush()
            print(f"[{rec['step']}] rc={rec['rc']} {rec['seconds']} s " + " ".join(f"{k}={rec[k]}" for k in ("value_mrays_s", "rccl_ranks_seen", "bit_identical", "error") if rec.get(k) is not None), flush=True)
            if rec["rc"] != 0:
                failed.append(rec["step"])
        tail = dict(step="summary", steps=len(steps), failed=failed, rc=1 if failed else 0)
        log.write(json.dumps(tail) + "\n")
    print(json.dumps(tail))
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
