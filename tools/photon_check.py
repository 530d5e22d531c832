import sys, time
sys.path.insert(0, '.')
import numpy as np
from rpt_amd import Renderer, scenes
from oracle.pyoracle import OracleScene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sc, cam, cfg = scenes.CONFIGS["C4"]()
watts = 200000.0 / (130 * 105) * n
r = Renderer(sc, cam).width(size).height(size).watts(watts).gather_size(20).gather_size_volume(3).seed(0)
t = time.time(); st = r.photon_map_build(n, 1); print("gpu build", st, "%.2fs" % (time.time() - t))
gs, gv = r.photon_map_download(0), r.photon_map_download(1)
o = OracleScene(sc)
t = time.time(); pm = o.photon_map(n, 1, watts, 20, 3, seed=0, robust=1); print("oracle build %.2fs" % (time.time() - t))
es, ev = pm.photons(0), pm.photons(1)
print("counts gpu", gs.shape[0], gv.shape[0], "oracle", es.shape[0], ev.shape[0])
from scipy.spatial import cKDTree
def cmp(name, g, e):
    tree = cKDTree(g[:, :3].astype(np.float64))
    d, j = tree.query(e[:, :3])
    tol = 1e-4 * (1 + np.abs(e[:, :3]).max(axis=1))
    ok = d < tol
    print(name, "oracle photons matched by position: %.5f (unmatched %d of %d)" % (ok.mean(), (~ok).sum(), len(e)))
    gp, ep = g[j[ok]], e[ok]
    print("   dir err %.2e  power rel err %.2e" % (np.abs(gp[:, 3:6] - ep[:, 3:6]).max(), (np.abs(gp[:, 6:9] - ep[:, 6:9]) / (np.abs(ep[:, 6:9]) + 1e-12)).max()))
    if name == "volume":
        rr = np.abs(gp[:, 9] - ep[:, 9]) / ep[:, 9]
        print("   radius rel err: median %.2e  p99 %.2e  frac>1e-2: %.4f  max %.3f" % (np.median(rr), np.quantile(rr, 0.99), (rr > 1e-2).mean(), rr.max()))
cmp("surface", gs, es); cmp("volume", gv, ev)
t = time.time(); img = r.photon_sample_array(spp); print("gpu query %.2fs" % (time.time() - t), img.mean(0))
t = time.time(); ref = pm.render(cam, size, size, spp, seed=0); print("oracle query %.2fs" % (time.time() - t), ref.mean(0))
print("rel rms", np.sqrt(np.mean((img - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)), "finite", np.isfinite(img).all())
