"""Kernel time against the `pull_batch` option (lanes that wait for a work item before a wave runs its item bookkeeping).
Usage: python tools/pull_batch_sweep.py [workload ...]   (C3 C2 at their own sizes, C5 / C5G at 2048 x 2048 x 64)"""
import hashlib
import sys
sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

rpt_amd.set_option("timing", 1)
for name in (sys.argv[1:] or ["C3", "C2", "C5"]):
    scene, cam, cfg = scenes.CONFIGS[name]()
    big = name in ("C5", "C5G")
    w, h, spp = (2048, 2048, 64) if big else (cfg["width"], cfg["height"], cfg["spp"])
    if big:
        rpt_amd.set_option("chunk_spp", 32)
    r = Renderer(scene, cam).width(w).height(h).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    for pb in (1, 2, 4, 6, 8, 12, 16, 24):
        rpt_amd.set_option("pull_batch", pb)
        ms = []
        for _ in range(3):
            r._sample_offset = 0
            img = r.sample_array(spp)
            ms.append(r.timing()[0])
        print(f"{name} pull_batch {pb:2d}: kernel {min(ms):8.3f} ms   frame sha {hashlib.sha256(img.tobytes()).hexdigest()[:12]}", flush=True)
    rpt_amd.set_option("chunk_spp", 0)
    rpt_amd.set_option("pull_batch", 6)
