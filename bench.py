#!/usr/bin/env python3
"""Benchmark of the render core: Mpaths/s on the 1080p / 64 spp / depth 8 demo scene.

    python bench.py --gpus N --steps K --warmup W

A step is one whole frame: every pixel of the image traced to completion on the GPU(s), scene
and tables already resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank per
GPU) the frame's 8x8 tiles are dealt round-robin to the ranks, each rank traces its share, and the
tile buffers are gathered on rank 0 over RCCL and laid out as the image there - the one inter-GPU
step of the path.  The frame is fixed as N grows (strong scaling).

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      fp32 VALU roofline of the dominant kernel (pt_k_trace): algorithmic flops per frame
                (event counts of the CPU oracle on the same workload x per-event weights) over the
                kernel's duration measured with HIP events on the launch stream;
  cpu_baseline  the CPU renderer timed on this host's cores on the same frame, scheduled like the
                reference (tile jobs, cores-1 workers).
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pathtracer_amd import api, scenes  # noqa: E402
from pathtracer_amd import dist as ptdist  # noqa: E402

# fp32 operations per event, from the reference source (one add, mul, divide, sqrt or compare each).
FLOP_WEIGHTS = {
    "casts": 70,            # grid entry / traverse init (code/raycast.cpp:619-670)
    "dda_steps": 6,         # spatialGridTraverseNext
    "bsphere_tests": 20,    # lineSphereCollision
    "isect_sphere": 35, "isect_box": 40, "isect_cylinder": 45, "isect_cone": 50,
    "unit_space": 170,      # transformToUnitSpace + transformToGlobalSpace + distance (code/raycast.cpp:750-770)
    "rotated_box": 75,
    "shade_opaque": 40, "shade_oren": 60, "shade_glass": 65,
    "paths": 25 + 6,        # camera ray + accumulate
    "lens": 40,
}
PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X vector fp32 peak: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz x 2 (FMA) x 2 (packed)
PEAK_SCALAR_ISSUE_TFLOPS = 78.6    # the same machine issuing plain (unfused, unpacked) adds and multiplies, one wave64
                                   # instruction per SIMD every 2 cycles: the ceiling of code that must round every
                                   # operation like the reference; profiles/micro/valu_rate measures 58.6 (4 waves per
                                   # SIMD) to 66.2 (8 waves) T lane-ops/s of it
# flop/path of the named workloads as counted by the CPU leg (oracle event counters x FLOP_WEIGHTS); used
# when the CPU leg is skipped (--no-cpu, more than one GPU) so that the roofline block is always filled
# (demo: the whole frame; the others: the same scene and samples per pixel at a reduced image size - the figure moves by
# less than 0.1 % with the size, tests/test_host_stages.py pins the small ones)
KNOWN_FLOPS_PER_PATH = {"demo-1080p-64spp-d8": 259.5531495662587,
                        "glass-1080p-128spp-d32": 3655.3755324074073, "glass-1080p-16spp-d32": 3656.648076322366,
                        "dense100k-4k-256spp-d8": 4271.193952738557, "dense100k-1080p-16spp-d8": 4270.564724633967,
                        "dense10k-1080p-16spp-d8": 2212.231339906449, "mixed9-1080p-64spp-d16": 490.5181691834548}


def flops_from_counters(c):
    unit = c["uneven_sphere"] + c["isect_cylinder"] + c["isect_cone"]
    total = 0
    for k, w in FLOP_WEIGHTS.items():
        if k == "unit_space":
            total += unit * w
        else:
            total += c[k] * w
    return total


def workload(name):
    if name == "demo-1080p-64spp-d8":
        world, objs = scenes.default_scene()
        return world, objs, dict(width=1920, height=1080, spp=64, bounces=8, seed=12345)
    if name == "demo-640x360-16spp-d8":
        world, objs = scenes.default_scene()
        return world, objs, dict(width=640, height=360, spp=16, bounces=8, seed=12345)
    if name == "glass-1080p-128spp-d32":
        world, objs = scenes.glass_scene(seed=3, per_type=16)
        return world, objs, dict(width=1920, height=1080, spp=128, bounces=32, seed=12345)
    if name == "dense100k-4k-256spp-d8":
        world, objs = scenes.dense_scene(n=100000, seed=4)
        return world, objs, dict(width=3840, height=2160, spp=256, bounces=8, seed=12345, grid_hint=64)
    if name == "glass-1080p-16spp-d32":
        world, objs = scenes.glass_scene(seed=3, per_type=16)
        return world, objs, dict(width=1920, height=1080, spp=16, bounces=32, seed=12345)
    if name == "dense100k-1080p-16spp-d8":
        world, objs = scenes.dense_scene(n=100000, seed=4)
        return world, objs, dict(width=1920, height=1080, spp=16, bounces=8, seed=12345, grid_hint=64)
    if name == "dense10k-1080p-16spp-d8":
        world, objs = scenes.dense_scene(n=10000, seed=4, half_extent=90.0, height=45.0)
        return world, objs, dict(width=1920, height=1080, spp=16, bounces=8, seed=12345, grid_hint=32)
    if name == "demo-1080p-progressive":    # interactive mode: every step is one frame of one sample per pixel, accumulated on the device
        world, objs = scenes.default_scene()
        return world, objs, dict(width=1920, height=1080, spp=64, bounces=8, seed=12345, frame_index=0)
    if name == "mixed9-1080p-64spp-d16":    # nine entities of every kind and material: the all-features brute-force kernel
        world, objs = scenes.mixed_scene(seed=11, n=8)
        return world, objs, dict(width=1920, height=1080, spp=64, bounces=16, seed=12345)
    m = re.fullmatch(r"demo-(\d+)x(\d+)-(\d+)spp-d(\d+)", name)
    if m:                                   # the demo scene at any size, for scaling studies
        world, objs = scenes.default_scene()
        return world, objs, dict(width=int(m.group(1)), height=int(m.group(2)), spp=int(m.group(3)), bounces=int(m.group(4)), seed=12345)
    m = re.fullmatch(r"dense100k-(\d+)x(\d+)-(\d+)spp-d(\d+)", name)
    if m:                                   # the 100 000-entity scene at any size (short items as at full size need >= 128 spp)
        world, objs = scenes.dense_scene(n=100000, seed=4)
        return world, objs, dict(width=int(m.group(1)), height=int(m.group(2)), spp=int(m.group(3)), bounces=int(m.group(4)), seed=12345, grid_hint=64)
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(world, objs, kw, budget_paths=160e6):
    """CPU renderer on this host: the reference's own code where oracle/_ref was built, else the
    oracle port; tile jobs in the reference's order on cores-1 worker threads."""
    from oracle import oraclebind
    from oracle.oraclebind import FP, OracleScene
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "1024"))) - 1)
    kw = dict(kw)
    st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    sc = OracleScene(world, objs, st)
    paths = sc.path_count()
    sample = f"whole frame {st.width}x{st.height}"
    if paths > budget_paths:          # keep the leg to tens of seconds: a centred crop of the same frame
        scale = (budget_paths / paths) ** 0.5
        st.width, st.height = max(64, int(st.width * scale) // 16 * 16), max(36, int(st.height * scale) // 9 * 9)
        sc.close()
        sc = OracleScene(world, objs, st)
        paths = sc.path_count()
        sample = f"same scene and spp at {st.width}x{st.height}"
    img = np.zeros((st.height, st.width, 3), np.float32)
    t0 = time.perf_counter()
    sc.lib.pto_render_mt(sc.h, st.seed, api.PT_SEED_PER_JOB, cores, img.ctypes.data_as(FP))
    dt = time.perf_counter() - t0
    counters = sc.counters()
    out = {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
           "sample": f"{sample}, {paths} paths, {dt:.2f} s"}
    if oraclebind.have_ref() and len(objs) < 10000:
        ref = oraclebind.load_ref()
        if hasattr(ref, "ref_render_mt"):
            ref.ref_pcg_seed(0, st.seed)
            ref.ref_tables(None, None)
            oa = api.object_array(objs)
            ref.ref_scene_setup(C.byref(world), oa, len(objs), C.byref(st), 1)
            t0 = time.perf_counter()
            ref.ref_render_mt(st.seed, cores, img.ctypes.data_as(FP))
            dt = time.perf_counter() - t0
            out.update({"value": paths / dt / 1e6, "kind": "reference", "sample": f"{sample}, {paths} paths, {dt:.2f} s"})
    sc.close()
    out["c1_demo_640x360_16spp"] = cpu_c1(world, objs, cores) if len(objs) < 10 else None
    return out, counters, paths


def oracle_flops_per_path(world, objs, kw, budget_paths, threads=4):
    """flop/path of a workload from the oracle's event counters, on the same scene and samples per pixel at an image
    size cut down to about budget_paths paths."""
    from oracle.oraclebind import FP, OracleScene
    kw = dict(kw)
    st = api.default_settings(kw.pop("width"), kw.pop("height"), kw.pop("spp"), kw.pop("bounces"), **kw)
    sc = OracleScene(world, objs, st)
    paths = sc.path_count()
    if paths > budget_paths:
        scale = (budget_paths / paths) ** 0.5
        st.width, st.height = max(64, int(st.width * scale) // 16 * 16), max(36, int(st.height * scale) // 9 * 9)
        sc.close()
        sc = OracleScene(world, objs, st)
    img = np.zeros((st.height, st.width, 3), np.float32)
    sc.lib.pto_render_mt(sc.h, st.seed, api.PT_SEED_PER_JOB, threads, img.ctypes.data_as(FP))
    counters = sc.counters()
    sc.close()
    return flops_from_counters(counters) / max(1, counters["paths"]), counters


def cpu_c1(world, objs, cores):
    """The reference's own CPU-sized case beside it: 640x360, 16 spp, depth 8, with one worker thread and with cores-1."""
    from oracle.oraclebind import FP, OracleScene
    st = api.default_settings(640, 360, 16, 8, seed=12345)
    sc = OracleScene(world, objs, st)
    paths = sc.path_count()
    img = np.zeros((st.height, st.width, 3), np.float32)
    row = {"paths": paths, "kind": "port"}
    for threads in (1, cores):
        t0 = time.perf_counter()
        sc.lib.pto_render_mt(sc.h, st.seed, api.PT_SEED_PER_JOB, threads, img.ctypes.data_as(FP))
        dt = time.perf_counter() - t0
        row[f"seconds_{threads}_threads"] = dt
        row[f"mpaths_per_s_{threads}_threads"] = paths / dt / 1e6
    sc.close()
    return row


def ms_hint(seconds, steps):
    return seconds / max(1, steps) * 1e3


class Env:
    """torch, the process group and this rank's place in it."""

    def __init__(self, gpus):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world_size != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world_size}: launch with torch.distributed.run --nproc-per-node {gpus}")
        # Rehearsal of the multi-rank path on a box with fewer GPUs than ranks: PT_BENCH_BACKEND=gloo lets several ranks
        # share a GPU (RCCL refuses two ranks on one device); the numbers of such a run mean nothing, the code path is the same.
        backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            self.local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.local_rank)
        if self.world_size > 1:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)

    def max_over_ranks(self, value, op="max"):
        if self.world_size == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return float(t.item())


def measure(env, name, steps, warmup, prewarm_enabled=True, ranges=0):
    """Prepares workload `name` on this rank's partition and times `steps` frames after `warmup` untimed ones, bracketed
    by a barrier and a device synchronisation on both sides.  Returns what the JSON line is made of."""
    torch, dist, world_size, rank = env.torch, env.dist, env.world_size, env.rank
    world, objs, kw = workload(name)
    skw = dict(kw)
    skw["sample_ranges"] = ranges
    st = api.default_settings(skw.pop("width"), skw.pop("height"), skw.pop("spp"), skw.pop("bounces"),
                              part_index=rank, part_count=world_size, **skw)
    r = api.Renderer(env.local_rank)
    r.prepare(world, objs, st)
    stats = r.stats()
    part_floats = r.partition_floats()

    stream = torch.cuda.Stream()          # a stream of its own: handle 0 (torch's default stream) would mean "the context's stream" to pt_trace
    if world_size > 1:
        # tile-major partition buffers; rank 0 gathers them and assembles the image on a side stream while
        # the next frame is traced
        pipe = ptdist.FramePipeline([r], [rank], world_size, st.width, st.height, rank, world_size)
        assert part_floats <= pipe.stride
        step = pipe.step
    else:
        image = torch.empty(st.height * st.width * 3, dtype=torch.float32, device="cuda")

        def step():
            r.trace(d_rgb=image.data_ptr(), stream=stream.cuda_stream, sync=False)

    progressive = kw.get("frame_index", -1) >= 0
    if progressive:
        # progressive frames: the frame counter moves on before every trace
        trace_step, counter = step, [0]

        def step():
            r.set_frame(counter[0])
            counter[0] += 1
            trace_step()

    # A GPU that has been idle runs its first half second of work below its clocks (3 % on the demo frame).  Whatever
    # warm-up count was asked for, the device is first kept busy for about that long, the same number of steps on
    # every rank (the steps contain a collective).
    prewarm = 0
    if prewarm_enabled:
        t_warm = time.perf_counter()
        step()
        torch.cuda.synchronize()
        prewarm = 1
        chunk = max(1, min(64, int(0.1 / max(env.max_over_ranks(time.perf_counter() - t_warm), 1e-6))))
        while env.max_over_ranks(time.perf_counter() - t_warm) < 0.6:
            for _ in range(chunk):
                step()
            torch.cuda.synchronize()
            prewarm += chunk
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    r.sync()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r.sync()                                   # folds the HIP events of the K launches into the statistics
    kstats = r.stats()
    return {"world": world, "objs": objs, "kw": kw, "st": st, "r": r, "stats": stats, "kstats": kstats, "prewarm": prewarm,
            "progressive": progressive,
            "dt": env.max_over_ranks(dt), "paths": env.max_over_ranks(stats.paths, "sum"), "kernel_ms": env.max_over_ranks(kstats.dominant_kernel_ms)}


def progressive_block(env, frames=256, warm=32):
    """Configuration 5, the interactive mode: the demo scene at 1080p, one sample per pixel and frame added into the
    accumulation buffer, every frame waited for as a display would (the per-frame latency a viewer sees): median and 99th
    percentile over `frames` frames.  With one rank the frame is pt_set_frame + pt_trace with synchronisation; with
    several, every rank traces its tiles, the tile buffers are gathered on rank 0 and assembled there (one pipeline step
    and a device synchronisation on every rank), and the latency reported is the slowest rank's."""
    torch, world_size, rank = env.torch, env.world_size, env.rank
    world, objs, kw = workload("demo-1080p-progressive")
    skw = dict(kw)
    st = api.default_settings(skw.pop("width"), skw.pop("height"), skw.pop("spp"), skw.pop("bounces"),
                              part_index=rank, part_count=world_size, **skw)
    r = api.Renderer(env.local_rank)
    r.prepare(world, objs, st)
    if world_size > 1:
        pipe = ptdist.FramePipeline([r], [rank], world_size, st.width, st.height, rank, world_size)

        def frame(f):
            r.set_frame(f)
            pipe.step()
            torch.cuda.synchronize()
    else:
        def frame(f):
            r.set_frame(f)
            r.trace()                        # synchronises
    for f in range(warm):
        frame(f)
    lat = np.zeros(frames)
    for f in range(frames):
        t0 = time.perf_counter()
        frame(f)
        lat[f] = (time.perf_counter() - t0) * 1e3
    r.sync()
    paths = env.max_over_ranks(r.stats().paths, "sum")
    kernel_ms = env.max_over_ranks(r.stats().dominant_kernel_ms)
    med, p99, worst = (env.max_over_ranks(v) for v in (np.median(lat), np.percentile(lat, 99), lat.max()))
    r.close()
    return {"workload": "demo-1080p-progressive", "definition": "one sample per pixel per frame, every frame waited for; host clock around the frame",
            "frames": frames, "warmup_frames": warm, "median_ms": med, "p99_ms": p99, "max_ms": worst, "kernel_ms": kernel_ms,
            "paths_per_frame": int(paths), "n_gpus": world_size}


def measured_traffic(name, world_size):
    """HBM bytes per launch of the trace kernel from the committed counter passes (profiles/summarise.py), or None."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None, None
    for entry in [pm] + list(pm.get("also", {}).values()):
        if entry.get("workload") == name and entry.get("n_gpus", 1) == world_size:
            return entry.get("hbm_bytes_per_launch"), entry.get("source")
    return None, None


def roofline_block(name, m, world_size, flops_per_path=None):
    """fp32 VALU roofline of pt_k_trace for one measurement: neither HBM (0.03 TB/s moved) nor MFMA (no dense contraction
    in a path tracer) bounds this kernel, the vector ALUs' instruction issue does, so the peak quoted is the vector fp32 peak."""
    k = m["kstats"]
    roof = {"bound": "valu", "kernel": "pt_k_trace", "achieved": None, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": None, "traffic": None, "kernel_ms": m["kernel_ms"],
            "kernel_ms_min": float(k.trace_ms_min), "kernel_ms_max": float(k.trace_ms_max)}
    if flops_per_path is None:
        flops_per_path = KNOWN_FLOPS_PER_PATH.get(name)
    if flops_per_path is not None and m["kernel_ms"] > 0:
        roof["flops_per_path"] = flops_per_path
        roof["achieved"] = flops_per_path * m["paths"] / world_size / (m["kernel_ms"] * 1e-3) / 1e12
        roof["frac"] = roof["achieved"] / PEAK_FP32_VALU_TFLOPS
        roof["frac_of_scalar_issue_peak"] = roof["achieved"] / PEAK_SCALAR_ISSUE_TFLOPS
    traffic, source = measured_traffic(name, world_size)
    # what the path has to write, whatever the design: twelve bytes a pixel (code/raycast.cpp:1270).  The design writes one
    # 16-byte range sum per pixel and range instead and folds them in a second kernel; traffic_ratio says what that costs.
    algorithmic = 12 * int(k.pixels) if world_size == 1 else None         # (the committed counter passes are one-GPU runs)
    if algorithmic:
        roof["algorithmic_bytes"] = algorithmic
    if traffic is not None:
        roof["traffic"] = traffic
        roof["traffic_source"] = source
        if algorithmic:
            roof["traffic_ratio"] = traffic / algorithmic
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="demo-1080p-64spp-d8")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the untimed steps that bring an idle GPU up to its clocks")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the two grid-scene configurations and the progressive frames that follow the default workload")
    ap.add_argument("--ranges", type=int, default=0, help="pt_settings.sample_ranges (0 = the library's choice)")
    args = ap.parse_args()

    if os.environ.get("PT_BENCH_HANG_TRACE"):          # rehearsals: where every thread stands after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PT_BENCH_HANG_TRACE"]), exit=True)
    env = Env(args.gpus)
    torch, dist, world_size, rank = env.torch, env.dist, env.world_size, env.rank
    m = measure(env, args.workload, args.steps, args.warmup, not args.no_prewarm, args.ranges)
    world, objs, kw, st, r, stats, kstats, prewarm = m["world"], m["objs"], m["kw"], m["st"], m["r"], m["stats"], m["kstats"], m["prewarm"]
    dt, total_paths, kernel_ms = m["dt"], m["paths"], m["kernel_ms"]
    if m["progressive"]:
        args.no_cpu = True

    # The application's view of the same frame (one GPU only): pt_render_start with host pointers in -> done with the
    # image in the caller's host buffer, i.e. set-up, upload, every kernel and the download.  Median of a few shots into
    # a pageable buffer the caller keeps between renders (the reference's malloc'ed AppData::buffer) and into one from
    # pt_image_alloc (page-locked).  The first shot of a context builds the random tables and is not counted.
    end_to_end = None
    if world_size == 1 and kw.get("frame_index", -1) < 0:
        shots = 9 if ms_hint(dt, args.steps) < 50 else 3
        end_to_end = {"definition": "pt_render_start -> done, host scene in, host image out"}
        for label, image in (("pageable", np.zeros((st.height, st.width, 3), np.float32)), ("pinned", r.image_alloc(st.width, st.height))):
            times = []
            for shot in range(shots + 1):
                ts = time.perf_counter()
                r.start(world, objs, st, out=image)
                r.wait()
                times.append((time.perf_counter() - ts) * 1e3)
            times = sorted(times[1:])
            end_to_end[f"ms_{label}"] = times[len(times) // 2]
            end_to_end[f"mpaths_per_s_{label}"] = stats.paths / times[len(times) // 2] / 1e3
            if label == "pinned":
                r.image_free(image)
        end_to_end["shots"] = shots

    # The grid-scene configurations beside the headline one, a few frames each (same partitioning, same pipeline): the
    # glass scene (1080p, 128 spp, depth 32, lens) and the 100 000-entity scene (3840x2160, 256 spp) - the frame that
    # is long enough to scale over eight GPUs.  Only after the default workload; every rank takes part.
    secondary = []
    progressive = None
    if args.workload == "demo-1080p-64spp-d8" and not args.no_secondary:
        r.close()
        r = None
        progressive = progressive_block(env)
        for name, k_steps, k_warm in (("glass-1080p-128spp-d32", 10, 2), ("dense100k-4k-256spp-d8", 3, 1)):
            mm = measure(env, name, k_steps, k_warm, False, 0)
            secondary.append({"workload": name, "steps": k_steps, "warmup": k_warm, "ms_per_step": mm["dt"] / k_steps * 1e3,
                              "value": mm["paths"] * k_steps / mm["dt"] / 1e6, "unit": "Mpaths/s", "paths_per_frame": int(mm["paths"]),
                              "objects": len(mm["objs"]), "roofline": roofline_block(name, mm, world_size)})
            mm["r"].close()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        out = {
            "metric": "Mpaths/s, 1080p 64spp demo scene" if args.workload == "demo-1080p-64spp-d8" else f"Mpaths/s, {args.workload}",
            "value": total_paths * args.steps / dt / 1e6, "unit": "Mpaths/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm,
            "ms_per_step": ms_per_step, "seconds_per_frame": ms_per_step / 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": st.width, "height": st.height, "spp_wanted": st.spp_wanted,
                       "max_bounces": st.max_bounces, "objects": len(objs), "paths_per_frame": int(total_paths),
                       "rays_per_frame_rank0": int(kstats.rays), "partition": f"8x8 tiles round-robin over {world_size} GPU(s)",
                       "seed_policy": "per-sample PCG32 streams"},
        }
        flops_per_path = None
        if not args.no_cpu and world_size == 1:          # the CPU leg belongs to the one-GPU line only
            base, counters, base_paths = cpu_baseline(world, objs, kw)
            out["cpu_baseline"] = base
            flops_per_path = flops_from_counters(counters) / max(1, counters["paths"])
        roof = roofline_block(args.workload, m, world_size, flops_per_path)
        out["roofline"] = roof
        if end_to_end is not None:
            out["end_to_end"] = end_to_end
        if secondary:
            out["secondary"] = secondary
        if progressive is not None:
            out["progressive"] = progressive
        # HBM bytes per launch: what the path writes by definition (12 bytes a pixel), and what this design's trace kernel
        # writes (one 16-byte range sum per pixel and range; pt_k_resolve then reads them and writes the pixels)
        out["hbm_algorithmic_bytes_per_launch"] = int(12 * stats.pixels)
        out["hbm_design_bytes_per_launch"] = int(16 * stats.pixels * api.load_core().pt_sample_ranges(C.byref(st)))
        print(json.dumps(out), flush=True)
    if r is not None:
        r.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
