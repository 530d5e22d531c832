import sys, json, time
sys.path.insert(0, '.')
import numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes
scene, cam, cfg = scenes.mesh_in_fog()
r = Renderer(scene, cam).width(512).height(512).max_bounces(cfg["max_bounces"]).seed(0)
rpt_amd.set_option("counters", 1)
t=time.time(); img = r.sample_array(8); print("first call (commit+render) %.2fs" % (time.time()-t))
c = r.counters(); print(c, r.scene_stats())
print("nodes/ray %.1f tris/ray %.1f rays/sample %.2f trips-eff %.3f" % (c["bvh_nodes"]/c["rays"], c["bvh_tris"]/c["rays"], c["rays"]/c["samples"], c["vertices"]/(c["wave_trips"]*64)))
