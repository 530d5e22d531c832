"""Times the two "kd-tree of shapes" examples at their own size (800x600):
  spheres: examples/fractal_spheres.rs (937 spheres in 5 groups), scene BVH vs forced linear scan
  meshes : examples/fractal_teapots.rs layout (937 uses of one 2,304-triangle mesh), instanced vs flattened
Usage: python tools/fractal_bench.py [spp]"""
import json
import sys
import time

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = {}
cases = [("spheres/scene_bvh", scenes.fractal_spheres, 64, 1), ("spheres/linear", scenes.fractal_spheres, 1 << 30, 1),
         ("meshes/instanced", scenes.fractal_meshes, 64, 1), ("meshes/flattened", scenes.fractal_meshes, 64, 0)]
for name, make, thresh, inst in cases:
    rpt_amd.set_option("scene_bvh_min", thresh)
    rpt_amd.set_option("instancing", inst)
    rpt_amd.set_option("timing", 1)
    scene, cam, cfg = make()
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(1)
    t0 = time.time()
    r.sample_array(2)
    first = time.time() - t0
    t0 = time.time()
    img = r.sample_array(spp)
    wall = time.time() - t0
    st = r.scene_stats()
    out[name] = dict(commit_plus_first_s=round(first, 3), wall_s=round(wall, 4), kernel_ms=round(r.timing()[0], 3),
                     msamples_per_s=round(cfg["width"] * cfg["height"] * spp / wall / 1e6, 1), mean=float(img.mean()),
                     scene_bytes=st["scene_bytes"], bvh_tris=st["bvh_tris"], instances=st["instances"])
print(json.dumps(out, indent=1))
