"""C4 surface estimate: gathered photons that cannot be blocked (both ends inside the room shell, no record near) skip the
visibility scan; this compares that build with one that always scans (photon_skip 128) and both with the oracle.
Usage: python tools/photon_vis_check.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes
from oracle.pyoracle import OracleScene
scene, cam, cfg = scenes.CONFIGS["C4"]()
n, size, spp = 20000, 64, 16
r = Renderer(scene, cam).width(size).height(size).watts(14.65 * n).seed(3).gather_size(20).gather_size_volume(3)
r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
imgs = []
for skip in (0, 128):
    rpt_amd.set_option("photon_skip", skip)
    r._sample_offset = 0
    imgs.append(r.photon_sample_array(spp))
rpt_amd.set_option("photon_skip", 0)
pm = OracleScene(scene).photon_map(n, 1, 14.65 * n, 20, 3, seed=3, robust=1)
exp = pm.render(cam, size, size, spp, seed=3)
a, b = imgs
for nm, im in (("scan-free", a), ("always-scan", b)):
    print(nm, "rel rms vs oracle", float(np.sqrt(((im - exp) ** 2).mean() / (exp ** 2).mean())), "mean ratio", float(im.mean() / exp.mean()))
d = np.abs(a - b).max(axis=-1)
idx = np.nonzero(d.reshape(-1) > 1e-6 * np.abs(b).max())[0]
print("pixels differing", len(idx))
ea = np.abs(a.reshape(-1, 3)[idx] - exp.reshape(-1, 3)[idx]).sum(axis=1)
eb = np.abs(b.reshape(-1, 3)[idx] - exp.reshape(-1, 3)[idx]).sum(axis=1)
print("scan-free closer to the oracle at", int((ea < eb).sum()), "always-scan closer at", int((eb < ea).sum()))
print("sum abs err at those pixels: scan-free", float(ea.sum()), "always-scan", float(eb.sum()))
