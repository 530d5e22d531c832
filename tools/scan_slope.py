"""Cost of one more scan record: C3 with k extra far-away primitives that no ray reaches (the image is
unchanged), kernel time against k.  Usage: python tools/scan_slope.py [spp]"""
import math
import sys

import numpy as np

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Material, Object, Renderer, cube, polygon, scenes, sphere, vec3  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rpt_amd.set_option("timing", 1)
rpt_amd.set_option("scene_bvh_min", 1 << 30)
white = Material.diffuse(vec3(1, 1, 1))


def extra(kind, j):
    c = vec3(100000.0 + 10.0 * j, -100000.0, 100000.0)
    if kind == "aabb":
        return cube().translate(c)
    if kind == "cube":
        return cube().rotate_y(0.3).translate(c)
    if kind == "sphere":
        return sphere().translate(c)
    if kind == "rect":
        return polygon([c, c + vec3(1, 0, 0), c + vec3(1, 0, 1), c + vec3(0, 0, 1)])
    raise ValueError(kind)


base = None
for kind in ("aabb", "rect", "cube", "sphere"):
    for k in (0, 8, 16):
        scene, cam, cfg = scenes.lampshade()
        for j in range(k):
            scene.add(Object(extra(kind, j)).material(white))
        r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
        r.sample_array(4)
        ms = []
        for _ in range(3):
            r._sample_offset = 0
            img = r.sample_array(spp)
            ms.append(r.timing()[0])
        if base is None:
            base = img
        print(f"{kind:6s} +{k:2d}: kernel {min(ms):8.3f} ms  same_image={np.array_equal(img, base)}  "
              f"{ {n: v for n, v in r.scene_stats().items() if n in ('spheres', 'cubes', 'aabbs', 'rects', 'tris')} }", flush=True)
