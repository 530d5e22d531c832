"""Reference-epsilon mode: kernel time of C3 / C2 at their configured sizes and the schedule counters of a smaller render
(objects evaluated per query, evaluation rounds per query, lanes holding a path per trip).  RPT_LIB selects an A/B build."""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
from rpt_amd import Renderer, scenes, _lib

names = sys.argv[1:] or ["C3", "C2"]
for name in names:
    scene, cam, cfg = scenes.CONFIGS[name]()
    scene.set_option("epsilon_policy", 1)
    scene.set_option("timing", 1)
    for k, v in (kv.split("=") for kv in os.environ.get("RPT_OPTS", "").split(",") if kv):
        scene.set_option(k, int(v))
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        img = r.sample_array(cfg["spp"])
        ms.append(r.timing()[0])
    n = cfg["width"] * cfg["height"] * cfg["spp"]
    line = f"{os.environ.get('RPT_LIB', 'default')} {os.environ.get('RPT_OPTS', '')} {name}: kernel {min(ms):.2f} ms = {n / min(ms) / 1e3:.0f} Msamples/s, mean {img.mean():.9g}"
    if os.environ.get("RPT_COUNTERS", "1") == "1":
        scene.set_option("counters", 1)
        scene.set_option("f64_cull", 2)   # (the counters build with the search limits of the plain one)
        r._sample_offset = 0
        r.width(256).height(256).sample_array(16)
        out = (C.c_uint64 * 12)()
        _lib.check(_lib.load().rpt_debug_epsilon_counters(scene._handle, out))
        c = [int(v) for v in out]
        line += (f" | per query: {c[8] / c[0]:.2f} objects evaluated, {c[9] * 64 / c[0]:.2f} x 1/64 rounds; rays/sample {c[0] / c[6]:.2f}, "
                 f"vertices/sample {c[7] / c[6]:.2f}, live lanes/trip {c[11] / max(c[10], 1):.1f}, rounds/trip {c[9] / max(c[10], 1):.2f}")
    print(line, flush=True)
