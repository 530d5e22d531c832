import sys
sys.path.insert(0, ".")
import rpt_amd
from rpt_amd import Renderer, scenes
rpt_amd.set_option("timing", 1)
for name in ("C3", "C2"):
    scene, cam, cfg = scenes.CONFIGS[name]()
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(8)
    ms = []
    for _ in range(5):
        r._sample_offset = 0
        img = r.sample_array(cfg["spp"])
        ms.append(r.timing()[0])
    print(name, "kernel", round(min(ms), 3), "ms  mean", img.mean())
