#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]

One "step" = one pass of the hot path over the whole workload: Renderer::sample of the
lampshade scene in fog (BASELINE config C3, examples/volumetric_pathtrace_lampshade.rs) at
1024x1024 pixels x 256 paths per pixel, scene resident in HBM before the timed region.  With
N > 1 (launched by torch.distributed.run, one rank per GPU) the 32x32 pixel tiles are sharded
over the ranks and the frame is assembled on rank 0 with one RCCL sum-reduce per step; the
total work is fixed, so scaling is "strong".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector


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


def hbm_traffic_from_profile(workload, width, height, spp, world):
    """HBM bytes per render_kernel launch from the committed rocprofv3 PMC passes
    (profiles/r01/final_c3_pmc_render_kernel.json: FETCH_SIZE and WRITE_SIZE in KB, collected in
    separate --pmc runs; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM).  Only valid for the
    configuration that was profiled (C3 at full size on one GPU); null otherwise."""
    path = os.path.join(ROOT, "profiles", "r01", "final_c3_pmc_render_kernel.json")
    if (workload, width, height, spp, world) != ("C3", 1024, 1024, 256, 1) or not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    if "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return None
    return int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)


def cpu_baseline(scene, cam, cfg, width, height, target_seconds=15.0):
    """The fp64 oracle (C++ restatement of rpt's CPU algorithm, literal reference semantics,
    one task per image row like the rayon loop) on all host cores, on a bounded sample."""
    from oracle.pyoracle import OracleScene
    osc = OracleScene(scene)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    osc.render(cam, width, height, 1, cfg["max_bounces"], seed=0, threads=cores)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(32, target_seconds / max(t1, 1e-3))))
    t0 = time.perf_counter()
    osc.render(cam, width, height, spp, cfg["max_bounces"], seed=0, threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": round(width * height * spp / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{width}x{height}x{spp}spp of the same scene (fp64 C++ restatement of rpt's CPU algorithm, "
                  f"{dt:.1f} s, {cores} threads)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--photons", type=int, default=0, help="C4 only: photons shot per step (default: the config's 1,000,000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0, choices=(0, 1, 2),
                    help="HIP streams the consecutive steps alternate between (2: the start of a step overlaps the tail of the "
                         "previous one; 0 = 1 stream on one GPU, where the per-launch kernel time is the figure of merit, 2 on several)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) even with one rank, to rehearse the N > 1 code path")
    args = ap.parse_args()

    import torch  # before the HIP library: one shared HIP runtime (rpt_amd/_lib.py)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import numpy as np
    import rpt_amd
    from rpt_amd import Renderer, scenes
    from rpt_amd.dist import photon_map_build_sharded

    scene, cam, cfg = scenes.CONFIGS[args.workload]()
    width = args.width or cfg["width"]
    height = args.height or cfg["height"]
    spp = args.spp or cfg["spp"]
    r = Renderer(scene, cam).width(width).height(height).max_bounces(cfg["max_bounces"]).seed(0)
    r.device(local_rank).shard(rank, world)
    # Two frames on two HIP streams, used alternately -- what an iterative render does with consecutive batches:
    # the first blocks of step k + 1 take over the CUs that the last paths of step k no longer fill (the library
    # keeps a slab + work counter per stream), and for N > 1 the sum-reduce of step k (RCCL, its own stream)
    # overlaps the rendering of step k + 1.  Everything has completed before the closing synchronize + barrier of
    # the timed region.
    frames = [torch.zeros(width * height * 3, dtype=torch.float64, device="cuda") for _ in range(2)]
    if args.streams == 0:
        args.streams = 2 if world > 1 else 1
    streams = [torch.cuda.Stream() for _ in range(2)]
    d_out = frames[0]
    pending = [None, None]
    step_no = [0]
    stream = torch.cuda.current_stream().cuda_stream
    rpt_amd.set_option("timing", 1)

    kernel_ms = []

    photon = "photons" in cfg   # C4: Renderer::photon_render = shoot + build the map + camera pass, every step
    if photon:
        n_photons = args.photons or cfg["photons"]
        r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"])
        r.watts(cfg["renderer_watts"] / cfg["photons"] * n_photons)

    def step():
        r._sample_offset = 0
        slot = step_no[0] % 2 if (args.streams == 2 and not photon) else 0   # the photon map is built on the null stream
        step_no[0] += 1
        frame = frames[slot]
        on = torch.cuda.default_stream() if (photon or args.streams == 1) else streams[slot]
        with torch.cuda.stream(on):
            st = on.cuda_stream
            if pending[slot] is not None:   # the reduce that last used this frame must be done before it is overwritten
                pending[slot].wait()
                pending[slot] = None
            if photon:
                if dist is not None:   # shooting sharded by photon index, records all-gathered over RCCL
                    photon_map_build_sharded(r, n_photons, Renderer.PHOTON_POINT_BEAM, rank, world)
                else:
                    r.photon_map_build(n_photons, Renderer.PHOTON_POINT_BEAM)
                r.photon_sample_device(spp, frame.data_ptr(), st)
            else:
                r.sample_device(spp, frame.data_ptr(), st)
            if dist is not None:
                pending[slot] = dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for i in range(2):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    # W untimed steps, and with two streams at least one on each: the scratch of a stream is allocated by its first launch
    for _ in range(max(args.warmup, args.streams) if args.warmup else 0):
        step()
    drain()
    torch.cuda.synchronize()
    if args.warmup:
        r.timing_mean()   # start the kernel-time measurement at the timed region
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):   # nothing in a step waits for the device: the launches queue up behind each other
        step()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    grid_blocks = r.timing()[2]
    kernel_ms.append(r.timing_mean()[0])   # HIP events around every timed launch, on the stream it ran on
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if photon:
        if rank == 0:
            samples_per_step = width * height * spp
            ms_per_step = elapsed / args.steps * 1e3
            print(json.dumps({
                "metric": "Msamples/sec", "value": round(samples_per_step * args.steps / elapsed / 1e6, 3),
                "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"C4 lampshade beam x point photon map {width}x{height}x{spp}spp, "
                                       f"{n_photons} photons shot + map build + camera pass per step",
                           "scene": "examples/volumetric_beamphoton_lampshade.rs", "parallelism": f"tile-shard x{world}", "streams": args.streams,
                           "camera_pass_kernel_ms": round(float(np.mean(kernel_ms)), 3)},
                "roofline": None}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # one extra, untimed pass with device counters on (rank-local work) for the roofline figures
    rpt_amd.set_option("counters", 1)
    rpt_amd.set_option("timing", 0)
    r._sample_offset = 0
    cspp = min(spp, 16)
    r.sample_device(cspp, d_out.data_ptr(), stream)
    torch.cuda.synchronize()
    cnt = r.counters()
    rpt_amd.set_option("counters", 0)

    if rank == 0:
        samples_per_step = width * height * spp
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step * args.steps / elapsed / 1e6
        stats = r.scene_stats()
        bytes_ps, flops_ps, rays_ps = algorithmic_work(stats, len(scene.objects), cnt, max(cnt["samples"], 1))
        k_ms = float(np.mean(kernel_ms))
        overlapped = args.streams == 2
        if overlapped:
            # two launches are in flight: the events of one span its wait for the CUs the other still holds, so the
            # per-launch figure is the step time (an upper bound of the kernel's own time; the resolve is inside it)
            k_ms = ms_per_step
        local_samples = samples_per_step / world          # tiles are sharded evenly over ranks
        ach_gbs = bytes_ps * local_samples / (k_ms * 1e-3) / 1e9
        ach_tflops = flops_ps * local_samples / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "Msamples/sec",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "wall_clock_s": round(ms_per_step / 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} lampshade-in-fog path trace {width}x{height}x{spp}spp"
                       if args.workload == "C3" else f"{args.workload} {width}x{height}x{spp}spp",
                       "scene": "examples/volumetric_pathtrace_lampshade.rs" if args.workload == "C3" else args.workload,
                       "parallelism": f"tile-shard x{world}", "streams": args.streams, "rays_per_sample": round(rays_ps, 3),
                       "Mrays_per_s": round(value * rays_ps, 1)},
            "roofline": {
                "bound": "hbm",
                "achieved": round(ach_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(ach_gbs / HBM_PEAK_GBS, 4),
                "traffic": hbm_traffic_from_profile(args.workload, width, height, spp, world),
                "kernel": "rptg::render_kernel",
                "kernel_ms": round(k_ms, 3),
                "kernel_ms_source": "step time (launches overlap on two streams)" if overlapped else "HIP events around each launch",
                "grid_blocks": grid_blocks,
                "algorithmic_bytes_per_sample": round(bytes_ps, 1),
                "note": "scene records are wave-uniform and served from the scalar cache, not HBM: the binding "
                        "resource is fp32 VALU issue, see `valu` and DESIGN.md section 5",
                "valu": {"achieved": round(ach_tflops, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tflops / FP32_PEAK_TFLOPS, 4),
                         "algorithmic_flops_per_sample": round(flops_ps, 1)},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cam, cfg, width, height)
            out["config"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
