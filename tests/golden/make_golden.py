"""Generates tests/golden/*.npz from the CPU oracle (oracle/rpt_oracle.cpp).

These are NOT reference outputs: the reference is Rust and cannot be built in this image (DESIGN.md
section 2), and its only test on the path is the colour test reproduced in tests/test_oracle_kat.py.
The fixtures freeze the oracle's own results (robust epsilon policy, fixed seeds) so that
  * a change to the oracle that moves any number is caught on CPU (tests/test_golden.py), and
  * the GPU tests can compare against committed vectors without trusting the oracle build of the day.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleScene  # noqa: E402
from rpt_amd import scenes  # noqa: E402
from tests.util import random_rays  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {  # name -> (scene factory, ray sphere centre, radius, tile size, spp[, number of rays (default 256)])
    "C1": (scenes.spheres, (0.5, 0.0, 1.0), 12.0, 32, 16),
    "C1lit": (scenes.spheres_lit, (0.5, 0.0, 1.0), 12.0, 32, 16),
    "C2": (scenes.cornell, (278.0, 274.0, 280.0), 700.0, 32, 16),
    "C3": (scenes.lampshade, (278.0, 274.0, 280.0), 700.0, 32, 16),
    "C5small": (lambda: scenes.mesh_in_fog(nu=32, nv=32), (0.0, 0.0, 0.0), 4.0, 32, 16),
    # config C5 at the size bench.py --workload C5 walks: 224 x 224 x 2 = 100,352 triangles (the reference's kd-tree in
    # the oracle, the two-box BVH on the device); more rays than the small cases because only ~14 % of them reach the mesh
    "C5": (scenes.mesh_in_fog, (0.0, 0.0, 0.0), 4.0, 64, 16, 4096),
    "fractal": (scenes.fractal_spheres, (0.0, 0.0, 0.0), 4.0, 32, 16),
}


def main():
    only = sys.argv[1:]
    for name, (make, centre, radius, size, spp, *rest) in CASES.items():
        if only and name not in only:
            continue
        scene, cam, cfg = make()
        o = OracleScene(scene)
        rng = np.random.default_rng(2024)
        ro, rd = random_rays(rng, rest[0] if rest else 256, np.array(centre), radius)
        ro, rd = ro.astype(np.float32), rd.astype(np.float32)   # what the device is given
        t, obj, nrm = o.intersect(ro, rd, robust=1)
        bounces = max(cfg["max_bounces"], 2) if name == "fractal" else cfg["max_bounces"]
        img = o.render(cam, size, size, spp, bounces, seed=17, robust=1)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ray_o=ro, ray_d=rd, hit_t=t, hit_obj=obj.astype(np.int32),
                            hit_n=nrm, image=img, size=size, spp=spp, max_bounces=bounces, seed=17)
        print(name, "hits", int((obj >= 0).sum()), "image mean", float(img.mean()))


if __name__ == "__main__":
    main()
