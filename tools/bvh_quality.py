"""C5's mesh tree built with 16-bin SAH splits throughout (bvh_sweep_below = 0) against exact SAH sweeps for ranges of at most n
triangles: nodes and triangles visited per ray, share of walks that end without a triangle, kernel time.
Usage: python tools/bvh_quality.py [width] [spp] [n ...]"""
import sys
import time

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
values = [int(v) for v in sys.argv[3:]] or [0, 64, 1024, 1 << 20]
rpt_amd.set_option("chunk_spp", 16)
for n in values:
    rpt_amd.set_option("bvh_sweep_below", n)
    scene, cam, cfg = scenes.CONFIGS["C5"]()
    r = Renderer(scene, cam).width(width).height(width).max_bounces(cfg["max_bounces"]).seed(0)
    t0 = time.perf_counter()
    r.sample_array(1)
    commit_s = time.perf_counter() - t0
    rpt_amd.set_option("timing", 1)
    rpt_amd.set_option("counters", 0)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        img = r.sample_array(spp)
        ms.append(r.timing()[0])
    rpt_amd.set_option("counters", 1)
    r._sample_offset = 0
    r.sample_array(8)
    c = r.counters()
    st = r.scene_stats()
    rpt_amd.set_option("counters", 0)
    print(f"bvh_sweep_below={n:8d}: {st['bvh_nodes']} nodes, depth {st['tree_depth']}, commit + first frame {commit_s:5.2f} s; nodes/ray {c['bvh_nodes'] / c['rays']:.3f}, "
          f"triangles/ray {c['bvh_tris'] / c['rays']:.3f}; kernel {min(ms):8.3f} ms for {width}x{width}x{spp}; mean {img.mean():.9f}", flush=True)
rpt_amd.set_option("bvh_sweep_below", 0)
