"""C4 camera-pass time with parts of the estimate switched off (diagnostic option "photon_skip").
Usage: python tools/photon_breakdown.py [spp] [photons]"""
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene, cam, cfg = scenes.CONFIGS["C4"]()
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["photons"]
rpt_amd.set_option("timing", 1)
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"] / cfg["photons"] * n)
print(r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM))
for skip, what in ((0, "full"), (1, "no volume (beam) estimate"), (2, "no surface estimate"), (3, "primary rays only"),
                   (6, "beam walk without the per-ray tests"), (9, "surface estimate without visibility scans"),
                   (17, "surface: no second pass"), (33, "surface: collection + ordering only"), (65, "surface: collection only"),
                   (513, "surface: hit record + material only"), (769, "... and no pixel list either")):
    rpt_amd.set_option("photon_skip", skip)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        r.photon_sample_array(spp)
        ms.append(r.timing()[0])
    print(f"skip={skip} {what:28s}: {min(ms):9.3f} ms for {cfg['width']}x{cfg['height']}x{spp}", flush=True)
rpt_amd.set_option("photon_skip", 0)
