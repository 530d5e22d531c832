import sys, time
sys.path.insert(0, '.')
import numpy as np
from rpt_amd import Renderer, scenes
from oracle.pyoracle import OracleScene
n, size, spp = 20000, 64, 4
for name in ["C2", "C4"]:
    sc, cam, cfg = scenes.CONFIGS[name]()
    watts = 200000.0 / (130 * 105) * n
    r = Renderer(sc, cam).width(size).height(size).watts(watts).gather_size(20).gather_size_volume(3).seed(0)
    st = r.photon_map_build(n, 1)
    o = OracleScene(sc)
    pm = o.photon_map(n, 1, watts, 20, 3, seed=0, robust=1)
    img = r.photon_sample_array(spp)
    ref = pm.render(cam, size, size, spp, seed=0)
    d = np.abs(img - ref).sum(axis=1)
    rel = np.sqrt(np.mean((img - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    print(name, st, "mean gpu", img.mean(0), "oracle", ref.mean(0), "relrms %.4f" % rel)
    bad = np.argsort(-d)[:8]
    for b in bad:
        print("   pix", b % size, b // size, "gpu", img[b], "ref", ref[b])
    print("   frac pixels with >1% diff:", (d > 0.01 * (np.abs(ref).sum(axis=1) + 1e-9)).mean())
