"""Reference-epsilon mode on a scene of many objects (examples/fractal_spheres.rs: 937 spheres in five groups, 800 x 600 x 100):
kernel time against the fp32 path, and the box test's work.  RPT_OPTS as in tools/f64_ab.py."""
import os
import sys
sys.path.insert(0, ".")
from rpt_amd import Renderer, scenes

for eps in (0, 1):
    scene, cam, cfg = scenes.fractal_spheres(5)
    scene.set_option("timing", 1)
    if eps:
        scene.set_option("epsilon_policy", 1)
    for k, v in (kv.split("=") for kv in os.environ.get("RPT_OPTS", "").split(",") if kv):
        scene.set_option(k, int(v))
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(2)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        img = r.sample_array(cfg["spp"])
        ms.append(r.timing()[0])
    print(f"fractal_spheres {cfg['width']}x{cfg['height']}x{cfg['spp']} epsilon_policy={eps}: kernel {min(ms):.2f} ms, mean {img.mean():.6f}", flush=True)
