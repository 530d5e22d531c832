#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C4|C5] [--no-secondary]

One "step" = one pass of the hot path over the whole workload: Renderer::sample of the
lampshade scene in fog (BASELINE config C3, examples/volumetric_pathtrace_lampshade.rs) at
1024x1024 pixels x 256 paths per pixel, scene resident in HBM before the timed region.  With
N > 1 the 32x32 pixel tiles are sharded over one rank per GPU and rank 0 assembles the frame from the owned tiles
of every rank (rpt_gather_frame_device: packed f64 tiles over RCCL, behind the C ABI); the total work is fixed, so
scaling is "strong".  `python bench.py --gpus N` starts its N ranks itself (fresh children through
torch.distributed.run, before this process has touched a GPU); under torch.distributed.run it is one of the
ranks.  Rank 0 prints ONE JSON line.

The headline (`value`, `config`, `roofline`, `cpu_baseline`) is C3.  At N = 1 the same line carries a `secondary`
array with the other BASELINE configurations (C2 cornell 512x512x64, C4 beam x point photon map 1024x1024x256 with
1 M photons, C5 100k-triangle mesh in fog 2048x2048x1024) at 2 steps each, every entry with its own value, ms_per_step,
roofline, issue figures and cpu_baseline.

Figures:
  value                 camera samples per second, whole job, frame resident in HBM at the end of a step
  ms_per_step           pipelined: consecutive steps alternate between two HIP streams (an iterative render's batches)
  wall_clock_s          one step alone: launch -> frame in HBM, strictly one stream, synchronised after every step
  value_host_resident   pipelined steps with the frame copied to the host of rank 0 (SURVEY.md 8d's wall definition)
  roofline              bound "hbm": measured HBM bytes of the dominant kernel / its duration against 8 TB/s.
                        The path is NOT HBM-bound (frac ~ 0.01); what binds it is fp32 VALU issue:
  valu_frac             algorithmic fp32 flops (SURVEY.md 8d's per-primitive counts on the reference's
                        structure x exact device counters) / kernel time / 157.3 TFLOP/s
  valu_issue_frac       wave-level VALU instructions issued / (SIMDs x clk/2 x kernel time), from the PMC profile
  active_lanes          mean fraction of the 64 lanes that are enabled in an issued VALU instruction (PMC), and
  path_lanes            fraction of lanes holding a live path vertex per loop trip (device counters of this run)
A PMC profile is used only if it was taken from exactly the kernel sources in the tree (`_source_digest`).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector
FP64_PEAK_TFLOPS = 78.6    # fp64 vector = half the fp32 vector rate (one wave64 v_fma_f64 per 4 clocks per SIMD)
N_SIMD = 1024              # 256 CUs x 4 SIMDs; one wave64 VALU instruction issues every 2 clocks per SIMD
N_XCD = 8                  # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: / 8 = the launch's clock cycles

SCENE_FILES = {"C2": "examples/cornell.rs", "C3": "examples/volumetric_pathtrace_lampshade.rs",
               "C4": "examples/volumetric_beamphoton_lampshade.rs", "C5": "examples/dragon.rs layout, procedural 100,352-triangle mesh",
               "C5G": "C5's mesh inside a KdTree<Box<dyn Bounded>> of 64 spheres (the shape of examples/fractal_teapots.rs:56 with one large mesh)",
               "C3eps": "examples/volumetric_pathtrace_lampshade.rs, reference-epsilon mode (option epsilon_policy = 1: fp64, rpt's own 1e-12 tests)",
               "C2eps": "examples/cornell.rs, reference-epsilon mode (option epsilon_policy = 1)",
               "C4eps": "examples/volumetric_beamphoton_lampshade.rs, reference-epsilon mode (option epsilon_policy = 1: shooting pass and the surface "
                        "estimate's visibility rays in fp64 with rpt's own 1e-12 tests; maps, k-nearest selection and beam estimate on the fp32 records)"}
WORKLOAD_NAMES = {"C2": "C2 cornell box path trace", "C3": "C3 lampshade-in-fog path trace",
                  "C4": "C4 lampshade beam x point photon map", "C5": "C5 100k-triangle mesh in fog path trace",
                  "C5G": "C5G (not a BASELINE configuration) 100k-triangle mesh in a kd-tree group of 64 spheres, in fog",
                  "C3eps": "C3 lampshade-in-fog path trace, reference-epsilon mode (fp64)", "C2eps": "C2 cornell box path trace, reference-epsilon mode (fp64)",
                  "C4eps": "C4 lampshade beam x point photon map, reference-epsilon mode (fp64 shooting + visibility rays)"}


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as fresh child processes (this
    process has not initialised a GPU -- torch is not even imported yet -- and never will), relay their output and
    exit with their status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def algorithmic_work(stats, n_objects, counters, samples):
    """Per-sample algorithmic bytes/flops of the closest-hit queries, counted
    on the REFERENCE's structure so the figure does not move when the device layout is specialised:
    bytes per ray = 64 B per scene object tested (3x4 inverse affine + type/material words, every
    object is tested: renderer.rs:419-423) + 48 B per triangle tested (single-leaf meshes test all
    their triangles) + 48 B for the accepted hit's normals + 64 B per (two-box) BVH node visited + 48 B per
    BVH triangle tested (exact device counters).  flops per ray follow SURVEY.md section 8d:
    56 per affine map, 30 per cube, 25 per sphere, 12 per plane, 75 per triangle, 20 per node,
    + 300 per path vertex of shading."""
    rays = counters["rays"]
    cubes = stats["cubes"] + stats["aabbs"]
    tris = stats["tris"] + 2 * stats["rects"]
    per_ray_bytes = 64 * n_objects + 48 * tris + 48
    per_ray_flops = (56 + 25) * stats["spheres"] + (56 + 30) * cubes + 12 * stats["planes"] + 75 * tris
    total_bytes = rays * per_ray_bytes + 64 * counters["bvh_nodes"] + 48 * counters["bvh_tris"]
    total_flops = rays * per_ray_flops + 20 * counters["bvh_nodes"] + 75 * counters["bvh_tris"] + 300 * counters["vertices"]
    return total_bytes / samples, total_flops / samples, rays / samples


def reference_flops_per_sample(scene, oc):
    """SURVEY.md section 8d's flop count on what the REFERENCE executes, per camera sample, from the oracle's own counters
    (`oc`: rays, tri_tests, nodes, samples of a short run of the cpu_baseline leg): every object is tested per ray
    (renderer.rs:419-423) -- 56 per `Transformed` ray map, 30 per cube, 25 per sphere, 12 per plane, 20 per mesh for the
    kd-tree's root-box test (kdtree.rs:132-139) --, triangles are tested only behind that box and the tree (75 each, 20 per kd
    node visited), + 300 per sample of shading."""
    from rpt_amd.api import Cube, Mesh, Plane, Sphere, Transformed
    n_xf = n_cube = n_sphere = n_plane = n_mesh = 0
    for o in scene.objects:
        base = o.shape.base()
        n_xf += isinstance(o.shape, Transformed)
        n_cube += isinstance(base, Cube)
        n_sphere += isinstance(base, Sphere)
        n_plane += isinstance(base, Plane)
        n_mesh += isinstance(base, Mesh)
    per_ray = 56 * n_xf + 30 * n_cube + 25 * n_sphere + 12 * n_plane + 20 * n_mesh
    samples = max(oc["samples"], 1)
    flops = (oc["rays"] * per_ray + 75 * oc["tri_tests"] + 20 * oc["nodes"]) / samples + 300.0
    return flops, {"rays": round(oc["rays"] / samples, 3), "object_tests": round(oc["obj_tests"] / samples, 2),
                   "triangle_tests": round(oc["tri_tests"] / samples, 3), "kd_nodes": round(oc["nodes"] / samples, 3),
                   "vertices": round(oc["vertices"] / samples, 3), "flops_per_ray_for_the_object_scan": per_ray}


def source_digest():
    """Digest of the kernel sources the library in the tree was built from (the stamp __graft_entry__.build() keeps)."""
    from __graft_entry__ import library_source_digest
    return library_source_digest()


def pmc_profile(workload, width, height, spp, world, photons=0):
    """The committed rocprofv3 PMC summary of the dominant kernel for exactly this configuration
    (profiles/rNN/pmc_<workload>.json, written by tools/pmc_passes.sh + tools/pmc_collect.py: separate --pmc passes,
    per-launch sums), newest round first.  Returns (pmc, path, current): `current` says whether the profile was taken
    from the very sources in the tree (its _source_digest); (None, None, False) when this configuration was never profiled."""
    digest = source_digest()
    stale = (None, None, False)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{workload}.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        cfg = pmc.get("_config", {})
        if (cfg.get("width"), cfg.get("height"), cfg.get("spp"), cfg.get("n_gpus", 1)) != (width, height, spp, world):
            continue
        if photons and cfg.get("photons") != photons:
            continue
        if pmc.get("_source_digest") == digest:
            return pmc, os.path.relpath(path, ROOT), True
        if stale[0] is None:
            stale = (pmc, os.path.relpath(path, ROOT), False)
    return stale


def roofline_block(kernel, k_ms, k_ms_source, grid_blocks, pmc, pmc_path, pmc_current, compulsory_bytes, model_bytes, gather_type, extra=None):
    """The bench contract's roofline object for the dominant kernel, HBM view: achieved = HBM bytes per launch /
    launch duration.  Bytes are the measured ones (2 x FETCH_SIZE + WRITE_SIZE of the PMC profile of this very
    configuration AND these very sources, MI355X_MICROARCH.md section HBM) when such a profile is committed, else the
    bytes the kernel moves by design (partial-sum slab written + read back, frame written).  `traffic_x1` is
    FETCH_SIZE + WRITE_SIZE: the guide's doubling of FETCH_SIZE is calibrated on wide coalesced streaming reads; for a
    kernel whose reads are 64-byte node / record gathers (`gather_type`) the undoubled figure is the safer reading.
    `compulsory_bytes` is what the algorithm has to move (the fp32 RGB frame)."""
    traffic, traffic_x1, source = None, None, None
    if pmc and pmc_current and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)   # rocprofv3 reports KB
        traffic_x1 = int((pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
        source = f"{pmc_path}: 2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this configuration and these sources (not this run)"
    elif pmc:
        source = f"none: {pmc_path} was taken from other kernel sources than the ones in the tree; `achieved` uses bytes_by_design"
    used = traffic if traffic is not None else model_bytes
    ach = used / (k_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
           "traffic": traffic, "traffic_x1": traffic_x1, "gather_type_reads": bool(gather_type),
           "traffic_source": source or "no PMC profile of this configuration: `achieved` uses bytes_by_design",
           "bytes_by_design": int(model_bytes), "compulsory_bytes": int(compulsory_bytes),
           "kernel": kernel, "kernel_ms": round(k_ms, 3), "kernel_ms_source": k_ms_source, "grid_blocks": grid_blocks,
           "note": "not HBM-bound: scene records are wave-uniform scalar-cache reads; the binding resource is fp32 VALU "
                   "issue -- see valu_frac / valu_issue_frac / active_lanes beside this block"}
    if extra:
        out.update(extra)
    return out


def issue_view(pmc, current):
    """(valu_issue_frac, active_lanes, s_waitcnt share) from the PMC summary: VALU instructions issued against the
    chip's issue slots over the profiled launch (its own GRBM_GUI_ACTIVE cycles give the clock), enabled lanes
    per issued VALU instruction, share of wave cycles spent waiting in s_waitcnt."""
    if not pmc or not current or "SQ_INSTS_VALU" not in pmc or "GRBM_GUI_ACTIVE" not in pmc:
        return None, None, None
    slots = N_SIMD * (pmc["GRBM_GUI_ACTIVE"] / N_XCD) / 2.0
    issue = pmc["SQ_INSTS_VALU"] / slots
    lanes = None
    if pmc.get("SQ_THREAD_CYCLES_VALU") and pmc.get("SQ_ACTIVE_INST_VALU"):
        lanes = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
    wait = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in pmc else None
    rnd = lambda v: None if v is None else round(v, 4)
    return rnd(issue), rnd(lanes), rnd(wait)


def mirror_issue_figures(out):
    """The figures that say what binds the kernel, once more INSIDE the roofline object (a reader that keeps only the
    contract's objects still gets them): fp32 VALU view of the same launch."""
    out["roofline"]["valu"] = {k: out.get(k) for k in ("valu_frac", "valu_tflops", "valu_peak_tflops", "valu_issue_frac", "active_lanes",
                                                      "lane_slot_utilisation", "s_waitcnt_share", "path_lanes", "pmc_source") if k in out}


def cpu_baseline(scene, cam, cfg, width, height, target_seconds=15.0):
    """The fp64 oracle (C++ restatement of rpt's CPU algorithm, literal reference semantics,
    one task per image row like the rayon loop) on the host CPUs this process may use, on a bounded sample."""
    from oracle.pyoracle import OracleScene
    osc = OracleScene(scene)
    threads = usable_cpus()
    t0 = time.perf_counter()
    _, oc = osc.render(cam, width, height, 1, cfg["max_bounces"], seed=0, threads=threads, counters=True)   # (its work counters: reference_flops_per_sample)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(32, target_seconds / max(t1, 1e-3))))
    t0 = time.perf_counter()
    osc.render(cam, width, height, spp, cfg["max_bounces"], seed=0, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": round(width * height * spp / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{width}x{height}x{spp}spp of the same scene (fp64 C++ restatement of rpt's CPU algorithm, "
                  f"{dt:.1f} s, {threads} threads = usable CPUs of {os.cpu_count()} logical)",
    }, oc


def cpu_baseline_photon(scene, cam, cfg, width, height, spp, n_photons, watts, npix=16384):
    """C4 on the CPU: the oracle shoots the same number of photons and builds the map once (the whole of
    photon.rs:656-704), then runs the camera pass on a random pixel subset at 1 sample per pixel.  `value` puts the two
    together for the full frame (the camera time scaled: its cost is linear in pixels and samples) and says so; the
    measured parts are given beside it."""
    import numpy as np
    from oracle.pyoracle import OracleScene
    threads = usable_cpus()
    t0 = time.perf_counter()
    pm = OracleScene(scene).photon_map(n_photons, 1, watts, cfg["gather_size"], cfg["gather_size_volume"], seed=0, robust=0)
    t_map = time.perf_counter() - t0
    pix = np.sort(np.random.default_rng(0).choice(width * height, size=min(npix, width * height), replace=False)).astype(np.uint32)
    t0 = time.perf_counter()
    pm.render(cam, width, height, 1, seed=0, pixels=pix, threads=threads)
    t_pix = time.perf_counter() - t0
    t_cam = t_pix * (width * height / len(pix)) * spp
    return {
        "value": round(width * height * spp / (t_map + t_cam) / 1e6, 4),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "extrapolated": True,
        "map_build_s": round(t_map, 2),
        "camera_pass_Msamples_per_s": round(len(pix) / t_pix / 1e6, 4),
        "sample": f"measured: {n_photons} photons shot + map built ({t_map:.1f} s) and the camera pass on {len(pix)} random pixels x 1 spp "
                  f"({t_pix:.2f} s); `value` scales that camera time to {width}x{height}x{spp} ({t_cam:.0f} s) and adds the map build; "
                  f"fp64 C++ restatement of src/photon.rs, {threads} threads",
    }


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", choices=sorted(SCENE_FILES))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--photons", type=int, default=0, help="C4 only: photons shot per step (default: the config's 1,000,000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only (default at N = 1 with the default workload and "
                                                               "size: C2, C4 and C5 follow at 2 steps each in `secondary`)")
    ap.add_argument("--streams", type=int, default=0, choices=(0, 1, 2),
                    help="HIP streams the consecutive steps alternate between (2, the default: the start of a step overlaps the tail "
                         "of the previous one, as consecutive batches of an iterative render do; 1: strictly one launch at a time)")
    ap.add_argument("--exchange", default="gather", choices=("gather", "reduce"),
                    help="N > 1: how rank 0 gets the frame -- gather of the owned tiles through the library's own RCCL communicator "
                         "(rpt_gather_frame_device, default) or torch.distributed's sum-reduce of zero-padded full frames")
    ap.add_argument("--chunk-spp", type=int, default=0, help="diagnostic: samples per work item (option chunk_spp; 0 = the automatic rule)")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="diagnostic, one GPU: render only rank 0's tiles of an N-rank job (the step time of one rank of N, without the exchange)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) and the frame exchange even with one rank, to rehearse the N > 1 code path")
    ap.add_argument("--dryrun-cpu", action="store_true",
                    help="no GPU: rehearse launch, rendezvous (gloo), frame reduce and the JSON line with an all-zero frame")
    return ap.parse_args(argv)


class Exchange:
    """What a step does with its frame when the job has more than one rank (or --force-dist): assemble it on rank 0.
    "gather": the library's communicator, every transfer enqueued on one exchange stream behind the step's render
    (HIP events order render -> exchange -> next writer of that frame).  "reduce": torch.distributed's async sum-reduce."""

    def __init__(self, torch, dist, mode, width, height, local_rank):
        self.torch, self.dist, self.mode, self.w, self.h = torch, dist, mode, width, height
        self.comm, self.stream, self.fallback = None, None, None
        if mode == "gather":
            from rpt_amd.dist import FrameComm
            try:
                self.comm = FrameComm.from_torch(dist, local_rank)
                ok = 1
            except Exception as e:   # the library's communicator could not be formed on this rank
                self.fallback, ok = f"{type(e).__name__}: {e}", 0
            # every rank takes the same route: the gather only if all of them have a communicator
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if self.comm is not None:
                    self.comm.close()
                    self.comm = None
                self.fallback = self.fallback or "another rank could not form the communicator"
                self.mode = "reduce"
            else:
                self.stream = torch.cuda.Stream()

    def issue(self, frame, on):
        """Behind the render that `on` has just been given; returns what the next user of `frame` has to wait for."""
        torch = self.torch
        if self.mode == "gather":
            ev = torch.cuda.Event()
            ev.record(on)
            self.stream.wait_event(ev)
            self.comm.gather(self.w, self.h, frame.data_ptr(), frame.data_ptr(), self.stream.cuda_stream)
            done = torch.cuda.Event()
            done.record(self.stream)
            return done
        with torch.cuda.stream(on):
            return self.dist.reduce(frame, dst=0, op=self.dist.ReduceOp.SUM, async_op=True)

    def wait(self, handle, on):
        """Make stream `on` wait for a handle of issue()."""
        if handle is None:
            return
        if self.mode == "gather":
            on.wait_event(handle)
        else:
            with self.torch.cuda.stream(on):
                handle.wait()

    def close(self):
        if self.comm is not None:
            self.comm.close()


def measure(workload, args, steps, warmup, streams_opt, torch, dist, rank, local_rank, world, headline):
    """Run one workload and return its fields of the JSON line (rank 0; None elsewhere)."""
    import rpt_amd
    from rpt_amd import Renderer, scenes
    from rpt_amd.dist import photon_map_build_sharded

    eps = workload.endswith("eps")   # the reference-epsilon mode of the same configuration (kernels_f64.hip)
    scene, cam, cfg = scenes.CONFIGS[workload[:-3] if eps else workload]()
    if eps:
        scene.set_option("epsilon_policy", 1)
    width = (args.width if headline else 0) or cfg["width"]
    height = (args.height if headline else 0) or cfg["height"]
    spp = (args.spp if headline else 0) or cfg["spp"]
    rpt_amd.set_option("timing", 1)
    rpt_amd.set_option("counters", 0)
    rpt_amd.set_option("chunk_spp", args.chunk_spp if headline else 0)
    r = Renderer(scene, cam).width(width).height(height).max_bounces(cfg["max_bounces"]).seed(0)
    r.device(local_rank).shard(rank, world)
    if headline and args.emulate_shard > 1 and world == 1:
        r.shard(0, args.emulate_shard)
    photon = "photons" in cfg   # C4: Renderer::photon_render = shoot + build the map + camera pass, every step
    n_photons = 0
    if photon:
        n_photons = (args.photons if headline else 0) or cfg["photons"]
        r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"])
        r.watts(cfg["renderer_watts"] / cfg["photons"] * n_photons)
    # Two HIP streams used alternately -- what an iterative render does with consecutive batches: the first blocks
    # of step k + 1 take over the CUs that the last paths of step k no longer fill (the library keeps a slab + work
    # counter per stream).  The frames form a ring of THREE so that, with N > 1, step k + 2 does not write the frame whose
    # exchange (issued behind step k, running beside step k + 1's persistent grid) may still be reading it.  Every frame
    # carries the handle of whatever touched it last -- its render's event, or its exchange -- and its next writer
    # (another stream) waits for that.
    n_streams = 1 if photon else (streams_opt or 2)   # the photon map is built on the null stream
    n_frames = 3
    frames = [torch.zeros(width * height * 3, dtype=torch.float64, device="cuda") for _ in range(n_frames)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    xchg = Exchange(torch, dist, args.exchange, width, height, local_rank) if dist is not None else None
    last_use = [None] * n_frames     # (kind, handle): "event" = torch.cuda.Event, "xchg" = Exchange handle
    step_no = [0]

    def stream_of(k):
        return torch.cuda.default_stream() if (photon or n_streams == 1) else streams[k % 2]

    def wait_frame(fi, on):
        if last_use[fi] is None:
            return
        kind, h = last_use[fi]
        if kind == "event":
            on.wait_event(h)
        else:
            xchg.wait(h, on)

    def step():
        r._sample_offset = 0
        k = step_no[0]
        step_no[0] += 1
        fi, on = k % n_frames, stream_of(k)
        wait_frame(fi, on)   # the last reader / writer of this frame (another stream, or the exchange)
        with torch.cuda.stream(on):
            st = on.cuda_stream
            if photon:
                if dist is not None:   # shooting sharded by photon index, records all-gathered over RCCL
                    photon_map_build_sharded(r, n_photons, Renderer.PHOTON_POINT_BEAM, rank, world, comm=xchg.comm)
                else:
                    r.photon_map_build(n_photons, Renderer.PHOTON_POINT_BEAM)
                r.photon_sample_device(spp, frames[fi].data_ptr(), st)
            else:
                r.sample_device(spp, frames[fi].data_ptr(), st)
        if xchg is not None:
            last_use[fi] = ("xchg", xchg.issue(frames[fi], on))
        else:
            ev = torch.cuda.Event()
            ev.record(on)
            last_use[fi] = ("event", ev)
        return fi, on

    def drain():
        for fi in range(n_frames):
            if last_use[fi] is not None and last_use[fi][0] == "xchg" and xchg.mode == "reduce":
                last_use[fi][1].wait()
        torch.cuda.synchronize()

    # W untimed steps, and with two streams at least one on each: the scratch of a stream is allocated by its first launch
    for _ in range(max(warmup, n_streams) if warmup else 0):
        step()
    drain()
    if warmup:
        r.timing_mean()   # start the kernel-time measurement at the timed region
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):   # nothing in a step waits for the device: the launches queue up behind each other
        step()
    drain()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    grid_blocks = r.timing()[2]
    k_ms = r.timing_mean()[0]   # HIP events around every timed launch of the dominant kernel, on the stream it ran on
    k_src = "HIP events around each launch of the timed region"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: is the assembled frame the frame?  Rank 0 renders the whole image alone (same seed: the render is a function of
    # scene, seed and pixel) and compares it with the last frame the exchange delivered, bit for bit (the exchange carries the
    # frame's own f64).  Untimed; no collective involved.
    exchange_verified = None
    if xchg is not None and not photon:
        if rank == 0:
            fi_last = (step_no[0] - 1) % n_frames
            whole = torch.zeros_like(frames[fi_last])
            r.shard(0, 1)
            r._sample_offset = 0
            r.sample_device(spp, whole.data_ptr(), 0)
            torch.cuda.synchronize()
            r.shard(rank, world)
            if headline and args.emulate_shard > 1 and world == 1:
                r.shard(0, args.emulate_shard)
            exchange_verified = bool(torch.equal(whole, frames[fi_last])) if not (headline and args.emulate_shard > 1) else None
            del whole
        r.timing_mean()   # (keep that launch out of the kernel-time mean)

    # ---- one step alone: strictly one stream, synchronised after every step (launch -> frame in HBM).  With two launches
    # in flight the HIP events of one also span its wait for the other grid's CUs, so the kernel's own duration comes
    # from these launches as well.
    single_s = None
    if n_streams == 2:
        n1 = max(1, min(steps, 3))
        saved, n_streams = n_streams, 1
        ts = []
        for _ in range(n1):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            drain()
            ts.append(time.perf_counter() - t1)
        n_streams = saved
        single_s = sum(ts) / len(ts)
        k_ms, k_src = r.timing_mean()[0], f"HIP events around {n1} launches that ran alone on one stream, after the timed region (its launches overlap on two)"
    else:
        single_s = elapsed / steps

    # ---- the pipelined steps with the frame delivered to the host of rank 0 ("kernel launch -> framebuffer
    # resident on host"): one pinned 24 B/pixel fp64 copy per step, on the step's stream, behind its render and exchange
    host_elapsed = None
    if headline and (not photon or dist is None):
        host_frame = torch.empty(width * height * 3, dtype=torch.float64, pin_memory=True) if rank == 0 else None
        n_host = max(1, min(steps, 5))
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n_host):
            fi, on = step()
            if rank == 0:
                wait_frame(fi, on)   # the exchange of this frame (N > 1), else its own render: already in stream order
                with torch.cuda.stream(on):
                    host_frame.copy_(frames[fi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(on)
                last_use[fi] = ("event", ev)
        drain()
        if dist is not None:
            dist.barrier()
        host_elapsed = (time.perf_counter() - t0) / n_host
        r.timing_mean()

    samples_per_step = width * height * spp
    ms_per_step = elapsed / steps * 1e3
    value = samples_per_step * steps / elapsed / 1e6
    local_samples = samples_per_step / world          # tiles are sharded evenly over ranks
    n_owned_px = width * height / world
    pmc, pmc_path, pmc_current = pmc_profile(workload, width, height, spp, world, n_photons)
    issue, lanes, wait = issue_view(pmc, pmc_current)
    compulsory = n_owned_px * 12.0   # fp32 RGB per owned pixel, written once
    out = {
        "value": round(value, 3),
        "unit": "Msamples/s",
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms_per_step, 3),
        "ms_per_step_is": "pipelined over two HIP streams" if n_streams == 2 else "one stream",
        "wall_clock_s": round(single_s, 5),
        "dtype": "f64" if eps else "f32",
    }
    if host_elapsed is not None:
        out["value_host_resident"] = round(samples_per_step / host_elapsed / 1e6, 3)
        out["ms_per_step_host_resident"] = round(host_elapsed * 1e3, 3)
    parallelism = f"tile-shard x{world}" + (f", frame {xchg.mode} on rank 0" if xchg is not None else "")
    if exchange_verified is not None:
        out["exchange_verified"] = exchange_verified   # rank 0's own render of the whole frame == the frame assembled from the ranks' tiles
    if xchg is not None and xchg.fallback:
        out["exchange_fallback"] = f"rpt_gather_frame_device unavailable ({xchg.fallback}): torch.distributed sum-reduce used instead"

    # one extra, untimed pass with device counters on (rank-local work) for the algorithmic figures
    r.scene.set_option("timing", 0)
    r.scene.set_option("counters", 1)
    r._sample_offset = 0
    cspp = 1 if photon else min(spp, 16)   # (the camera pass's counters build carries timers: one sample per pixel is enough for per-sample ratios)
    if photon:
        r.photon_sample_device(cspp, frames[0].data_ptr(), 0)
    else:
        r.sample_device(cspp, frames[0].data_ptr(), 0)
    torch.cuda.synchronize()
    cnt = r.counters()
    if xchg is not None:
        xchg.close()
    if rank != 0:
        return None

    stats = r.scene_stats()
    samples_c = max(cnt["samples"], 1)
    if photon:
        # slab: one float4 per (pixel, 64-sample chunk) written by the camera pass and read by the resolve; frame: 24 B/pixel
        n_chunks = (spp + 63) // 64
        model = n_owned_px * (32.0 * n_chunks + 24.0)
        # algorithmic flops of the camera pass per sample: the camera ray against the reference's object list (as in
        # algorithmic_work), 20 per (ray, photon sphere) pair the beam estimate tests + 25 more per accepted pair (kernel
        # weight, transmittance), 40 per gathered surface photon (K = gather_size: cosine, bsdf, power, kernel) -- the
        # visibility rays of the gathered photons (photon.rs:357-361) are NOT counted: the device proves most of them
        # unnecessary, and counting them on the reference's structure would flatter the figure
        cubes, tris = stats["cubes"] + stats["aabbs"], stats["tris"] + 2 * stats["rects"]
        ray_flops = (56 + 25) * stats["spheres"] + (56 + 30) * cubes + 12 * stats["planes"] + 75 * tris
        flops_ps = ray_flops + 300 + (20.0 * cnt["bvh_nodes"] + 25.0 * cnt["bvh_tris"]) / samples_c + 40.0 * cfg["gather_size"]
        ach_tflops = flops_ps * local_samples / (k_ms * 1e-3) / 1e12
        out["config"] = {"workload": f"{WORKLOAD_NAMES[workload]} {width}x{height}x{spp}spp, {n_photons} photons shot + map build + "
                                     f"camera pass per step", "scene": SCENE_FILES[workload], "parallelism": parallelism,
                         "streams": n_streams, "camera_pass_kernel_ms": round(k_ms, 3),
                         "beam_tests_per_sample": round(cnt["bvh_nodes"] / samples_c, 1),
                         "beam_accepted_per_sample": round(cnt["bvh_tris"] / samples_c, 1)}
        kernel = "rptg::photon_query_kernel"
        peak, frac_is = FP32_PEAK_TFLOPS, ("camera pass only (the map build is ~10 % of a step): camera ray on the reference's object list + 20 per "
                                           "tested and 25 per accepted (ray, photon) pair + 40 per gathered surface photon; visibility rays not counted")
        if eps:
            # two kernels per camera pass: the fp32 one (beam estimate, k-nearest selection handed over per sample: gather_size + 2 dwords
            # written, then read) and the fp64 one (camera ray again, one visibility ray per gathered photon, the terms) with its fp64 slab
            kernel = "rptg::photon_query_kernel<EMIT> + rpt64::photon_surface_f64_kernel"
            model += n_owned_px * spp * (cfg["gather_size"] + 2) * 4.0 * 2.0 + n_owned_px * 64.0 * n_chunks
            flops_ps += ray_flops * (1 + cfg["gather_size"])   # here every visibility ray is traced, as in the reference
            ach_tflops = flops_ps * local_samples / (k_ms * 1e-3) / 1e12
            peak, frac_is = FP64_PEAK_TFLOPS, ("camera pass only, both kernels, priced against the fp64 vector peak although the beam estimate runs in fp32: "
                                              "as for C4, plus the fp64 camera ray and one visibility ray per gathered photon on the reference's object list")
        out["roofline"] = roofline_block(kernel, k_ms, k_src, grid_blocks, pmc, pmc_path, pmc_current, compulsory, model, True)
        out.update({"valu_frac": round(ach_tflops / peak, 4), "valu_tflops": round(ach_tflops, 3),
                    "valu_peak_tflops": peak, "algorithmic_flops_per_sample": round(flops_ps, 1),
                    "valu_frac_is": frac_is,
                    "valu_issue_frac": issue, "active_lanes": lanes, "s_waitcnt_share": wait, "pmc_source": pmc_path if pmc_current else None})
        if world == 1 and not args.no_cpu_baseline:
            key = ("photon", workload[:-3] if eps else workload, width, height, spp, n_photons)   # (C4 and C4eps share the leg: the literal oracle either way)
            if key not in _CPU_LEG:
                _CPU_LEG[key] = cpu_baseline_photon(scene, cam, cfg, width, height, spp, n_photons, r.watts_)
            out["cpu_baseline"] = _CPU_LEG[key]
        mirror_issue_figures(out)
        return out

    bytes_ps, flops_all, rays_ps = algorithmic_work(stats, len(scene.objects), cnt, samples_c)
    chunk_spp, n_chunks = r.chunking(spp)
    model = n_owned_px * ((64.0 if eps else 32.0) * n_chunks + 24.0)   # slab written + read (fp64 partial sums in the reference-epsilon mode), fp64 frame written
    peak_tflops = FP64_PEAK_TFLOPS if eps else FP32_PEAK_TFLOPS
    out["config"] = {"workload": f"{WORKLOAD_NAMES[workload]} {width}x{height}x{spp}spp", "scene": SCENE_FILES[workload],
                     "parallelism": parallelism, "streams": n_streams, "rays_per_sample": round(rays_ps, 3),
                     "Mrays_per_s": round(value * rays_ps, 1), "chunk_spp": chunk_spp,
                     # the contract's `value` / `ms_per_step` are the pipelined figures; the walls SURVEY.md 8d defines are here too
                     "value_is": ("pipelined: consecutive steps alternate between two HIP streams and overlap (an iterative render's batches)"
                                  if n_streams == 2 else "one stream, one launch at a time"),
                     "wall_clock_s": round(single_s, 5), "wall_clock_is": "one step alone, launch -> frame in HBM, one stream, synchronised"}
    if host_elapsed is not None:
        out["config"]["value_host_resident"] = out["value_host_resident"]
        out["config"]["ms_per_step_host_resident"] = out["ms_per_step_host_resident"]
        out["config"]["host_resident_is"] = "the pipelined steps with every frame copied to pinned host memory of rank 0 (SURVEY.md 8d: launch -> framebuffer resident on host)"
    out["roofline"] = roofline_block("rpt64::render_f64_kernel" if eps else "rptg::render_kernel", k_ms, k_src, grid_blocks, pmc, pmc_path, pmc_current, compulsory, model,
                                     stats["bvh_nodes"] > 0, {"algorithmic_scene_bytes_per_sample": round(bytes_ps, 1)})
    # The CPU leg first: its work counters are what the algorithmic flops are counted on.
    oc = None
    if world == 1 and not args.no_cpu_baseline:
        key = (workload[:-3] if eps else workload, width, height)
        if key not in _CPU_LEG:
            _CPU_LEG[key] = cpu_baseline(scene, cam, cfg, width, height, 15.0 if headline else 8.0)
        out["cpu_baseline"], oc = _CPU_LEG[key]
        out["config"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    all_tflops = flops_all * local_samples / (k_ms * 1e-3) / 1e12
    if oc is not None:
        flops_ps, per_sample = reference_flops_per_sample(scene, oc)
        ach_tflops = flops_ps * local_samples / (k_ms * 1e-3) / 1e12
        out.update({"valu_frac": round(ach_tflops / peak_tflops, 4), "valu_tflops": round(ach_tflops, 3), "valu_peak_tflops": peak_tflops,
                    "algorithmic_flops_per_sample": round(flops_ps, 1),
                    "algorithmic_flops_source": {
                        "formula": "SURVEY.md 8d: per ray 56 per Transformed object + 30 per cube + 25 per sphere + 12 per plane + 20 per mesh (kd root box), "
                                   "+ 75 per triangle test + 20 per kd node, + 300 per sample of shading",
                        "counts_per_sample": per_sample,
                        "counts_from": f"the oracle's work counters on {width}x{height}x1 spp of this scene, literal policy (the cpu_baseline leg): "
                                       "what the REFERENCE executes -- triangles only behind the kd-tree's root box (kdtree.rs:132-139)"},
                    "valu_frac_is": "algorithmic flops of the reference's own execution / kernel time / the vector peak of the arithmetic type"})
    else:
        out.update({"valu_frac": None, "valu_tflops": None, "valu_peak_tflops": peak_tflops, "algorithmic_flops_per_sample": None,
                    "algorithmic_flops_source": "none in this run: the reference-structure counts come from the oracle's counters, i.e. from the cpu_baseline "
                                                "leg (rank 0 at N = 1 without --no-cpu-baseline)"})
    out.update({"valu_frac_all_objects": round(all_tflops / peak_tflops, 4), "flops_per_sample_all_objects": round(flops_all, 1),
                "valu_frac_all_objects_is": "every object AND every triangle of every single-leaf mesh charged per ray, + 300 per vertex (device counters): "
                                            "an upper bound that flatters the kernel -- not the section-8d figure",
                "valu_issue_frac": issue, "active_lanes": lanes, "s_waitcnt_share": wait, "pmc_source": pmc_path if pmc_current else None,
                "lane_slot_utilisation": None if issue is None or lanes is None else round(issue * lanes, 4),
                "path_lanes": round(cnt["vertices"] / max(1, 64 * cnt["wave_trips"]), 4)})
    mirror_issue_figures(out)
    return out


_CPU_LEG = {}   # (configuration, width, height) -> (cpu_baseline object, oracle counters): C3 and its reference-epsilon twin share one


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become the launcher.  Nothing below this line runs in this process.
        sys.exit(spawn_ranks(args.gpus, argv))

    # Rank 0 prints ONE line on stdout.  Libraries write there too (RCCL prints a version banner when a communicator is formed), so
    # file descriptor 1 points at stderr for the run and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch  # before the HIP library: one shared HIP runtime (rpt_amd/_lib.py)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dryrun_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()   # n_gpus in the line = the ranks the communicator actually formed
    if args.dryrun_cpu:
        return dryrun_cpu(args, dist, rank, world, json_fd)
    torch.cuda.set_device(local_rank)

    head = measure(args.workload, args, args.steps, args.warmup, args.streams, torch, dist, rank, local_rank, world, True)
    out = None
    if rank == 0:
        out = {"metric": "Msamples/sec", "value": head.pop("value"), "unit": head.pop("unit"), "n_gpus": world,
               "steps": head.pop("steps"), "warmup": head.pop("warmup"), "ms_per_step": head.pop("ms_per_step"),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": head.pop("dtype"), "data": "synthetic"}
        out.update(head)
    default_run = args.workload == "C3" and not (args.width or args.height or args.spp or args.chunk_spp or args.emulate_shard)
    if world == 1 and dist is None and default_run and not args.no_secondary:
        secondary = []
        for wl in ("C3eps", "C2", "C2eps", "C4", "C4eps", "C5", "C5G"):
            entry = measure(wl, args, 2, 1, 1, torch, None, rank, local_rank, world, False)
            entry = dict({"workload": wl}, **entry)
            secondary.append(entry)
        out["secondary"] = secondary
        out["secondary_note"] = ("C3eps (and C2eps, C4eps): the headline configuration (and C2, C4) in the reference-epsilon mode (option epsilon_policy = 1: fp64, rpt's own 1e-12 tests -- "
                                 "the mode that is within 1e-3 of the reference per pixel); then the other BASELINE configurations at their configured sizes (and C5G: the scene-tree + parked-mesh-walk flavour, "
                                 "2048x2048x256), 2 timed steps each after 1 warm-up, strictly one stream (ms_per_step = wall_clock_s x 1000), "
                                 "same definitions as the headline")
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dryrun_cpu(args, dist, rank, world, json_fd=1):
    """--dryrun-cpu: everything around the device work -- argument handling, rank launch, rendezvous, tile sharding,
    the per-step frame reduce to rank 0, max-over-ranks timing, the JSON line -- on the CPU with gloo.  Each rank
    writes 1.0 into the pixels of the tiles it owns; the reduced frame must be all ones."""
    import numpy as np
    import torch
    from rpt_amd.api import shard_pixels
    width, height = args.width or 256, args.height or 192
    frame = torch.zeros(height * width, 3, dtype=torch.float64)
    frame[torch.from_numpy(shard_pixels(width, height, rank, world).astype(np.int64))] = 1.0
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = frame.clone()
        if dist is not None:
            dist.reduce(out, dst=0, op=dist.ReduceOp.SUM)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ok = bool(np.all(out.numpy() == 1.0))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps({"metric": "Msamples/sec", "value": 0.0, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "dryrun": True, "frame_assembled": ok,
                          "config": {"workload": f"dry run {width}x{height}: tile shards reduced to rank 0, no device work",
                                     "parallelism": f"tile-shard x{world}"}}) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
