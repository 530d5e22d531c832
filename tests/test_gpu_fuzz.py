"""Randomised scenes (every shape kind, nested groups, shared meshes, rooms of rectangles, planes) against the
oracle: closest hits for all of them, a short render for a few.  Seeds are fixed; each scene exercises whichever
flatten-time paths its content triggers (linear scan, box shell, per-mesh trees, scene BVH, instancing)."""
import numpy as np
import pytest

from rpt_amd import (Camera, KdTree, Light, Material, Medium, Mesh, Object, Renderer, Scene, cube, plane, polygon, scenes,
                     sphere, vec3)
from tests.util import random_rays, rel_rms

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle.pyoracle import OracleScene
    return OracleScene(scene)


def _xf(rng, s, scale=(0.2, 0.8), spread=3.0, rotate=True):
    s = s.scale(rng.uniform(*scale, 3))
    if rotate and rng.random() < 0.6:
        s = s.rotate_y(rng.uniform(0, 6.28)).rotate_x(rng.uniform(0, 6.28))
    return s.translate(rng.uniform(-spread, spread, 3))


def _random_scene(seed):
    rng = np.random.default_rng(seed)
    sc = Scene()
    mats = [Material.diffuse(vec3(*rng.uniform(0.2, 0.9, 3))), Material.specular(vec3(*rng.uniform(0.2, 0.9, 3)), 8.0),
            Material.mirror(), Material.clear(1.5)]
    mesh_a = Mesh(scenes.bumpy_torus(int(rng.integers(5, 14)), int(rng.integers(4, 10))))     # 40..250 triangles
    mesh_small = Mesh(scenes.bumpy_torus(3, 3))                                                # 18 triangles (linear)
    n = int(rng.integers(3, 40)) if seed % 3 else int(rng.integers(60, 140))                    # some scenes pass the BVH threshold
    for i in range(n):
        kind = rng.integers(0, 6)
        base = [sphere(), cube(), cube(), mesh_a, mesh_small, sphere()][kind]
        rotate = not (kind == 2)                                                               # kind 2: axis-aligned boxes
        sc.add(Object(_xf(rng, base, rotate=rotate)).material(mats[int(rng.integers(0, 4))]))
    if seed % 2 == 0:   # a group with a nested group and shared meshes, transformed as a whole
        inner = KdTree([_xf(rng, sphere(), spread=1.0) for _ in range(5)] + [_xf(rng, mesh_a, spread=1.0)])
        kids = [_xf(rng, [sphere(), cube(), mesh_a][int(rng.integers(0, 3))], spread=1.5) for _ in range(12)]
        kids.append(inner.scale(vec3(0.5, 0.5, 0.5)).translate(vec3(0.5, 0.5, 0.0)))
        sc.add(Object(KdTree(kids).rotate_z(0.3).translate(vec3(0.0, 0.0, 1.0))).material(mats[0]))
    if seed % 4 < 2:    # a room: five or six walls that are the faces of one box
        lo, hi = -5.0, 5.0
        c = [[lo, lo, lo], [hi, lo, lo], [hi, hi, lo], [lo, hi, lo], [lo, lo, hi], [hi, lo, hi], [hi, hi, hi], [lo, hi, hi]]
        quads = [(0, 1, 2, 3), (4, 7, 6, 5), (0, 4, 5, 1), (3, 2, 6, 7), (0, 3, 7, 4), (1, 5, 6, 2)]
        for q in quads[:5 + seed % 2]:
            sc.add(Object(polygon([vec3(*c[k]) for k in q])).material(mats[0]))
    else:
        sc.add(Object(plane(vec3(0, 1, 0), -4.0)).material(mats[0]))
    light = polygon([vec3(-1, 4.5, -1), vec3(1, 4.5, -1), vec3(1, 4.5, 1), vec3(-1, 4.5, 1)])
    sc.add(Light.Object(Object(light).material(Material.light(vec3(1, 1, 1), 40.0))))
    sc.add(Light.Ambient(vec3(0.05, 0.05, 0.05)))
    if seed % 5 == 0:
        sc.add(Medium.homogeneous_isotropic(0.02, 0.1))
    return sc, rng


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_closest_hits_match_oracle(seed):
    sc, rng = _random_scene(seed)
    o, d = random_rays(rng, 20000, np.zeros(3), 6.0)
    r = Renderer(sc, Camera())
    t, obj, nrm = r.get_closest_hit(o, d)
    te, obje, nrme = _oracle(sc).intersect(o.astype(np.float32), d.astype(np.float32), robust=1)
    same = obj == obje
    with np.errstate(invalid="ignore"):
        coincident = (obj >= 0) & (obje >= 0) & ~same & (np.abs(t - te) <= 2e-4 * np.abs(te))   # overlapping shapes
    assert (same | coincident).mean() > 0.998, r.scene_stats()
    hit = same & (obje >= 0)
    assert hit.sum() > 2000
    rel = np.abs(t[hit] - te[hit]) / te[hit]
    assert np.quantile(rel, 0.999) < 2e-4
    close = hit & (np.abs(t - te) <= 2e-4 * np.abs(te))
    assert np.quantile(np.abs(nrm[close] - nrme[close]).max(axis=1), 0.995) < 5e-3
    assert np.all(np.isinf(t[same & (obje < 0)]))


@pytest.mark.parametrize("seed", [0, 1, 5, 6])
def test_random_scene_render_matches_oracle(seed):
    sc, rng = _random_scene(seed)
    cam = Camera.look_at(vec3(0.5, 1.0, 4.6), vec3(0, 0, 0), vec3(0, 1, 0), 0.9)
    w, h, spp = 48, 48, 16
    got = Renderer(sc, cam).width(w).height(h).max_bounces(3).seed(seed).sample_array(spp)
    exp = _oracle(sc).render(cam, w, h, spp, 3, seed=seed, robust=1)
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    assert rel_rms(got, exp) < 3e-2                            # specular / glass paths amplify a flipped decision
    assert abs(got.mean() - exp.mean()) / exp.mean() < 1e-2


@pytest.mark.parametrize("seed", [1, 4, 7, 10])
def test_random_scene_render_with_per_mesh_trees_matches_oracle(seed):
    """The same scenes with instancing and the scene-level tree switched off: every mesh use gets its own tree and
    the per-mesh-tree kernel (deferred, resumable walks over several trees per query) renders them."""
    import rpt_amd
    rpt_amd.set_option("instancing", 0)
    rpt_amd.set_option("scene_bvh_min", 1 << 20)
    try:
        sc, rng = _random_scene(seed)
        cam = Camera.look_at(vec3(0.5, 1.0, 4.6), vec3(0, 0, 0), vec3(0, 1, 0), 0.9)
        w, h, spp = 48, 48, 16
        r = Renderer(sc, cam).width(w).height(h).max_bounces(3).seed(seed)
        got = r.sample_array(spp)
        st = r.scene_stats()
    finally:
        rpt_amd.set_option("instancing", 1)
        rpt_amd.set_option("scene_bvh_min", 64)
    assert st["scene_bvh"] == 0 and st["instances"] == 0 and st["bvh_nodes"] > 0
    exp = _oracle(sc).render(cam, w, h, spp, 3, seed=seed, robust=1)
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    assert rel_rms(got, exp) < 3e-2
    assert abs(got.mean() - exp.mean()) / exp.mean() < 1e-2
