import sys, time
sys.path.insert(0, '.')
import numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes
n, size, spp = 1000000, 512, 8
sc, cam, cfg = scenes.CONFIGS["C4"]()
watts = 14.65 * n
rpt_amd.set_option("timing", 1)
for K in (20, 0, 1, 50):
    r = Renderer(sc, cam).width(size).height(size).watts(watts).gather_size(K).gather_size_volume(3).seed(0)
    if K == 20: r.photon_map_build(n, 1)
    img = r.photon_sample_array(spp)
    print("gather_size", K, "kernel ms %.1f" % r.timing()[0], flush=True)
