"""C5's mesh inside a kd-tree group of 64 spheres (scenes.mesh_among_spheres): the mesh outside the scene tree with parked walks
(default) against the mesh as a leaf of the scene tree, every query walked to completion (option "scene_tree_meshes" = 1).
Usage: python tools/group_mesh_check.py [width] [spp]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rpt_amd.set_option("timing", 1)
rpt_amd.set_option("chunk_spp", 16)
frames = {}
for in_tree in (1, 0):
    rpt_amd.set_option("scene_tree_meshes", in_tree)
    scene, cam, cfg = scenes.mesh_among_spheres()
    r = Renderer(scene, cam).width(width).height(width).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        frames[in_tree] = r.sample_array(spp)
        ms.append(r.timing()[0])
    st = r.scene_stats()
    print(f"scene_tree_meshes={in_tree}: kernel {min(ms):9.3f} ms for {width}x{width}x{spp}; scene tree over {st['scene_bvh_prims']} primitives, "
          f"{st['bvh_tris']} mesh triangles, mean {frames[in_tree].mean():.9f}", flush=True)
d = frames[0] - frames[1]
print(f"parked vs walked to completion: rel RMS {np.sqrt((d ** 2).mean() / (frames[1] ** 2).mean()):.3e}, max abs {np.abs(d).max():.3e}")
