"""Times examples/fractal_spheres.rs's workload (937 spheres in 5 KdTree groups, 800x600) with the
scene-level BVH and with the linear scan forced.  Usage: python tools/fractal_bench.py [spp]"""
import json
import sys
import time

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = {}
for mode, thresh in (("scene_bvh", 64), ("linear", 1 << 30)):
    rpt_amd.set_option("scene_bvh_min", thresh)
    rpt_amd.set_option("timing", 1)
    scene, cam, cfg = scenes.fractal_spheres()
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(1)
    r.sample_array(2)
    t0 = time.time()
    img = r.sample_array(spp)
    wall = time.time() - t0
    ms = r.timing()
    out[mode] = dict(wall_s=wall, kernel_ms=ms, msamples_per_s=cfg["width"] * cfg["height"] * spp / wall / 1e6,
                     mean=float(img.mean()), stats=r.scene_stats())
print(json.dumps(out, indent=1))
