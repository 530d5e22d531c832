import sys, time
sys.path.insert(0, '.')
import numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes
n = int(sys.argv[1]); size = int(sys.argv[2]); spp = int(sys.argv[3])
sc, cam, cfg = scenes.CONFIGS["C4"]()
watts = 200000.0 / (130 * 105) * n
r = Renderer(sc, cam).width(size).height(size).watts(watts).gather_size(20).gather_size_volume(3).seed(0)
t = time.time(); st = r.photon_map_build(n, 1); print("build", st, "wall %.3fs" % (time.time() - t), flush=True)
v = r.photon_map_download(1)
print("radius: median %.2f mean %.2f p99 %.1f max %.1f" % (np.median(v[:, 9]), v[:, 9].mean(), np.quantile(v[:, 9], 0.99), v[:, 9].max()), flush=True)
rpt_amd.set_option("timing", 1); rpt_amd.set_option("counters", 1)
t = time.time(); img = r.photon_sample_array(spp); dt = time.time() - t
print("query %dx%dx%d: %.3fs -> %.2f Msamples/s; kernel ms %s; mean %s finite %s" % (size, size, spp, dt, size * size * spp / dt / 1e6, r.timing(), img.mean(0), np.isfinite(img).all()), flush=True)
c = r.counters(); print("spheres visited/sample %.0f accepted/sample %.0f" % (c["bvh_nodes"] / max(c["samples"], 1), c["bvh_tris"] / max(c["samples"], 1)))
