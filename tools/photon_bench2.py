import sys
sys.path.insert(0, '.')
import rpt_amd
from rpt_amd import Renderer, scenes
n, size, spp = 1000000, 512, 8
sc, cam, cfg = scenes.CONFIGS["C4"]()
rpt_amd.set_option("timing", 1)
for K in (20, 0):
    r = Renderer(sc, cam).width(size).height(size).watts(14.65 * n).gather_size(K).gather_size_volume(3).seed(0)
    if K == 20: r.photon_map_build(n, 1)
    for bpc in (0, 1, 2, 4):
        rpt_amd.set_option("blocks_per_cu", bpc)
        r._sample_offset = 0
        img = r.photon_sample_array(spp)
        print("gather_size", K, "blocks_per_cu opt", bpc, "kernel ms %.1f grid %d" % (r.timing()[0], r.timing()[2]), flush=True)
