"""Kernel times of C5 (detached per-mesh-tree kernel), C5G and C4's camera pass: for A/B builds of the library (RPT_LIB)."""
import sys
sys.path.insert(0, ".")
import rpt_amd
from rpt_amd import Renderer, scenes
rpt_amd.set_option("timing", 1)
for name, width, spp in (("C5", 2048, 128), ("C5G", 2048, 64)):
    scene, cam, cfg = scenes.CONFIGS[name]()
    rpt_amd.set_option("chunk_spp", 32)
    r = Renderer(scene, cam).width(width).height(width).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        img = r.sample_array(spp)
        ms.append(r.timing()[0])
    print(name, f"{width}x{width}x{spp} kernel", round(min(ms), 3), "ms  mean", img.mean(), flush=True)
rpt_amd.set_option("chunk_spp", 0)
scene, cam, cfg = scenes.CONFIGS["C4"]()
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
st = r.photon_map_build(cfg["photons"], Renderer.PHOTON_POINT_BEAM)
ms = []
for _ in range(3):
    r._sample_offset = 0
    img = r.photon_sample_array(256)
    ms.append(r.timing()[0])
print("C4 camera pass", round(min(ms), 3), "ms  mean", img.mean(), " map build us", st["build_us"], "shoot us", st["shoot_us"], flush=True)
