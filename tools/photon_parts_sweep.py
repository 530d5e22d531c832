"""C4 camera pass time against the strips per 8x8 block ("photon_parts").  Usage: python tools/photon_parts_sweep.py [spp]"""
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene, cam, cfg = scenes.CONFIGS["C4"]()
n = cfg["photons"]
rpt_amd.set_option("timing", 1)
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
print(r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM))
for parts in (1, 2, 4, 8):
    rpt_amd.set_option("photon_parts", parts)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        r.photon_sample_array(spp)
        ms.append(r.timing()[0])
    print(f"photon_parts={parts}: {min(ms):9.3f} ms for {cfg['width']}x{cfg['height']}x{spp}", flush=True)
rpt_amd.set_option("photon_parts", 4)
