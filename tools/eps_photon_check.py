"""Photon mapping in the reference-epsilon mode against the oracle's literal restatement (robust = 0), with the fp32 policy's
numbers beside it.  Usage: python tools/eps_photon_check.py [workload] [photons] [size] [spp] [kind]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rpt_amd import Renderer, scenes  # noqa: E402
from oracle.pyoracle import OracleScene  # noqa: E402
from tests.util import rel_rms  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
size = int(sys.argv[3]) if len(sys.argv) > 3 else 64
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 16
kind = int(sys.argv[5]) if len(sys.argv) > 5 else 1
watts = 14.65 * n


def gpu(eps):
    scene, cam, cfg = scenes.CONFIGS[name]()
    if eps:
        scene.set_option("epsilon_policy", 1)
    r = Renderer(scene, cam).width(size).height(size).watts(watts).gather_size(20).gather_size_volume(3).seed(7)
    t0 = time.time()
    st = r.photon_map_build(n, kind)
    t1 = time.time()
    img = r.seed(0).photon_sample_array(spp)
    t2 = time.time()
    r.seed(1).photon_sample_array(spp)
    t3 = time.time()
    return st, img, r.photon_map_download(0), r.photon_map_download(1), (t1 - t0, t2 - t1, t3 - t2)


scene, cam, cfg = scenes.CONFIGS[name]()
osc = OracleScene(scene)
res = {}
for robust in (0, 1):
    pm = osc.photon_map(n, kind, watts, 20, 3, seed=7, robust=robust)
    res[robust] = (len(pm.photons(0)), len(pm.photons(1)), pm.render(cam, size, size, spp, seed=0), pm.photons(0))
for eps in (1, 0):
    st, img, ps, pv, tm = gpu(eps)
    print(f"--- device, epsilon_policy = {eps}: stored {st['surface']} surface / {st['volume']} volume photons; build {tm[0]:.3f} s, camera pass {tm[1]:.3f} s, again {tm[2]:.3f} s")
    for robust in (0, 1):
        ns, nv, exp, eph = res[robust]
        print(f"   oracle robust={robust}: {ns} / {nv} photons; rel-RMS {rel_rms(img, exp):.3e}, mean {(img.mean() - exp.mean()) / exp.mean():+.3e}, finite {np.isfinite(img).all()}")
print(f"oracle literal against oracle robust: rel-RMS {rel_rms(res[1][2], res[0][2]):.3e}, mean {(res[1][2].mean() - res[0][2].mean()) / res[0][2].mean():+.3e}")
