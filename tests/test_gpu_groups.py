"""GPU parity for `KdTree<Box<dyn Bounded>>` group shapes (src/kdtree.rs:103-146,
examples/fractal_spheres.rs) and for the scene-level BVH the library builds over many bounded
primitives: against the fp64 oracle's kd-tree and against the library's own linear scan."""
import numpy as np
import pytest

import rpt_amd
from rpt_amd import Camera, KdTree, Material, Mesh, Object, Renderer, Scene, cube, plane, scenes, sphere, vec3
from tests.util import random_rays, rel_rms

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle.pyoracle import OracleScene
    return OracleScene(scene)


@pytest.fixture
def scene_bvh_option():
    """Restores the commit-time option whatever the test sets it to."""
    yield lambda v: rpt_amd.set_option("scene_bvh_min", v)
    rpt_amd.set_option("scene_bvh_min", 64)


def _mixed_group(rng, n):
    kids = []
    for i in range(n):
        base = sphere() if i % 3 else cube()
        s = base.scale(rng.uniform(0.05, 0.3, 3))
        if i % 5 == 0:
            s = s.rotate_y(rng.uniform(0, 3)).rotate_x(rng.uniform(0, 3))
        kids.append(s.translate(rng.uniform(-2, 2, 3)))
    return kids


def test_fractal_spheres_closest_hit_matches_oracle_kdtree():
    scene, cam, cfg = scenes.fractal_spheres()
    r = Renderer(scene, cam)
    st = r.scene_stats()
    assert st["spheres"] == 937 and st["scene_bvh"] == 1 and st["scene_bvh_prims"] == 937
    o, d = random_rays(np.random.default_rng(5), 40000, np.zeros(3), 4.0)
    t, obj, nrm = r.get_closest_hit(o, d)
    te, obje, nrme = _oracle(scene).intersect(o.astype(np.float32), d.astype(np.float32), robust=1)
    same = obj == obje
    assert same.mean() > 0.9995                              # silhouette rays may flip in fp32
    hit = same & (obje >= 0)
    assert sorted(set(obje[obje >= 0].tolist())) == [0, 1, 2, 3, 4, 5]   # every level and the plane are hit
    assert np.max(np.abs(t[hit] - te[hit]) / te[hit]) < 2e-4
    assert np.quantile(np.abs(nrm[hit] - nrme[hit]).max(axis=1), 0.999) < 2e-3
    assert np.all(np.isinf(t[same & (obje < 0)]))


def test_fractal_spheres_render_matches_oracle():
    scene, cam, cfg = scenes.fractal_spheres()
    w, h, spp = 160, 120, 16
    got = Renderer(scene, cam).width(w).height(h).max_bounces(2).seed(4).sample_array(spp)
    exp = _oracle(scene).render(cam, w, h, spp, 2, seed=4, robust=1)
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    assert rel_rms(got, exp) < 5e-3
    assert abs(got.mean() - exp.mean()) / exp.mean() < 2e-3


def test_transformed_group_of_mixed_shapes_matches_flat_scene_and_oracle():
    """A rotated/scaled KdTree of spheres, cubes, a small mesh, a BVH mesh and a nested group:
    (a) equals the oracle's kd-tree, (b) equals the same primitives added as separate objects."""
    rng = np.random.default_rng(11)
    kids = _mixed_group(rng, 120)
    tet = scenes.bumpy_torus(6, 4)                           # 48 triangles -> BVH mesh inside the group
    kids.append(Mesh(tet).scale(vec3(1.5, 1.5, 1.5)).translate(vec3(0.0, 0.5, 0.0)))
    kids.append(Mesh(scenes.bumpy_torus(3, 3)).translate(vec3(1.0, -1.0, 0.5)))   # 18 triangles -> linear prims
    kids.append(KdTree(_mixed_group(rng, 10)).scale(vec3(0.5, 0.5, 0.5)).translate(vec3(0.0, 2.0, 0.0)))
    white = Material.diffuse(vec3(1, 1, 1))

    def wrap(s):
        return s.scale(vec3(1.2, 0.8, 1.0)).rotate_z(0.4).translate(vec3(0.3, 0.0, -0.2))

    grouped = Scene()
    grouped.add(Object(wrap(KdTree(kids))).material(white))
    grouped.add(Object(plane(vec3(0, 1, 0), -3.0)).material(white))
    flat = Scene()
    for k in kids[:-1]:
        flat.add(Object(wrap(k)).material(white))
    for k in kids[-1].base().shapes:
        flat.add(Object(wrap(k.scale(vec3(0.5, 0.5, 0.5)).translate(vec3(0.0, 2.0, 0.0)))).material(white))
    flat.add(Object(plane(vec3(0, 1, 0), -3.0)).material(white))
    o, d = random_rays(rng, 30000, np.zeros(3), 5.0)
    tg, objg, ng = Renderer(grouped, Camera()).get_closest_hit(o, d)
    te, obje, ne = _oracle(grouped).intersect(o.astype(np.float32), d.astype(np.float32), robust=1)
    same = objg == obje
    assert same.mean() > 0.999
    hit = same & (obje >= 0)
    assert (obje == 0).sum() > 3000
    assert np.quantile(np.abs(tg[hit] - te[hit]) / te[hit], 0.999) < 2e-4
    close = hit & (np.abs(tg - te) <= 2e-4 * te)
    assert np.quantile(np.abs(ng[close] - ne[close]).max(axis=1), 0.995) < 5e-3
    # same primitives as separate objects: same distances and normals (ties between overlapping
    # shapes may pick the other primitive, hence the tolerance on t rather than bit equality)
    tf, objf, nf = Renderer(flat, Camera()).get_closest_hit(o, d)
    assert np.array_equal(np.isfinite(tg), np.isfinite(tf))
    fin = np.isfinite(tg)
    assert np.max(np.abs(tg[fin] - tf[fin]) / tf[fin]) < 1e-6
    assert np.array_equal(objg[fin] == 1, objf[fin] == len(flat.objects) - 1)     # the plane


@pytest.mark.parametrize("name,size,spp", [("C2", 96, 16), ("C3", 96, 16)])
def test_forced_scene_bvh_equals_linear_scan(name, size, spp, scene_bvh_option):
    """The scene-level BVH is an acceleration structure only: forcing it on the Cornell-box
    configs (whose ~20 primitives are normally scanned) must give the same image as the scan up
    to exact-tie resolution between coincident surfaces."""
    imgs = []
    for force in (False, True):
        scene_bvh_option(2 if force else 1 << 30)
        scene, cam, cfg = scenes.CONFIGS[name]()
        r = Renderer(scene, cam).width(size).height(size).max_bounces(cfg["max_bounces"]).seed(9)
        assert r.scene_stats()["scene_bvh"] == (1 if force else 0)
        imgs.append(r.sample_array(spp))
    lin, bvh = imgs
    assert np.all(np.isfinite(bvh))
    differing = np.any(lin != bvh, axis=1).mean()
    assert differing < 0.02                                  # paths through exactly coincident surfaces only
    assert rel_rms(bvh, lin) < 2e-2
    assert abs(bvh.mean() - lin.mean()) / lin.mean() < 2e-3


def test_group_errors_are_rejected_at_add():
    with pytest.raises(TypeError):
        KdTree([plane(vec3(0, 1, 0), 0.0)])
    with pytest.raises(ValueError):
        KdTree([])


# ------------------------------------------------------------------ shared meshes (Arc<Mesh>) -> instancing
@pytest.fixture
def instancing_option():
    yield lambda v: rpt_amd.set_option("instancing", v)
    rpt_amd.set_option("instancing", 1)


def test_shared_mesh_is_instanced_and_matches_oracle():
    """examples/fractal_teapots.rs layout, 4 levels = 187 uses of one 2,304-triangle mesh: stored once
    (one local-space tree + 187 instance records), closest hits equal the oracle's kd-tree of kd-trees."""
    scene, cam, cfg = scenes.fractal_meshes(levels=4)
    r = Renderer(scene, cam)
    st = r.scene_stats()
    assert st["instances"] == 187 and st["shared_meshes"] == 1 and st["bvh_tris"] == 2304 and st["scene_bvh"] == 1
    o, d = random_rays(np.random.default_rng(6), 40000, np.zeros(3), 4.0)
    t, obj, nrm = r.get_closest_hit(o, d)
    te, obje, nrme = _oracle(scene).intersect(o.astype(np.float32), d.astype(np.float32), robust=1)
    same = obj == obje
    assert same.mean() > 0.999
    hit = same & (obje >= 0)
    assert sorted(set(obje[obje >= 0].tolist())) == [0, 1, 2, 3, 4]
    rel = np.abs(t[hit] - te[hit]) / te[hit]
    assert np.quantile(rel, 0.999) < 2e-4                    # silhouette edges may pick the neighbouring triangle
    close = hit & (rel_or_inf(t, te) <= 2e-4)
    assert np.quantile(np.abs(nrm[close] - nrme[close]).max(axis=1), 0.999) < 5e-3


def rel_or_inf(t, te):
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.abs(t - te) / te
    return np.where(np.isfinite(r), r, np.inf)


def test_instanced_render_matches_flattened_render_and_oracle(instancing_option):
    scene, cam, cfg = scenes.fractal_meshes(levels=3)
    w, h, spp = 128, 96, 16
    imgs = {}
    for inst in (1, 0):
        instancing_option(inst)
        scene, cam, cfg = scenes.fractal_meshes(levels=3)
        r = Renderer(scene, cam).width(w).height(h).max_bounces(2).seed(12)
        st = r.scene_stats()
        assert st["instances"] == (37 if inst else 0) and st["bvh_tris"] == (2304 if inst else 37 * 2304)
        imgs[inst] = r.sample_array(spp)
    exp = _oracle(scene).render(cam, w, h, spp, 2, seed=12, robust=1)
    assert exp.mean() > 0 and np.all(np.isfinite(imgs[1]))
    assert rel_rms(imgs[1], exp) < 1e-2 and rel_rms(imgs[0], exp) < 1e-2
    assert abs(imgs[1].mean() - exp.mean()) / exp.mean() < 3e-3
    assert rel_rms(imgs[1], imgs[0]) < 1e-2                  # local-space vs world-space triangles: rounding only


def test_full_fractal_of_937_mesh_instances_fits_and_renders():
    """The whole example: 937 instances of one mesh = 2.16 M triangles if flattened, 2,304 when instanced."""
    scene, cam, cfg = scenes.fractal_meshes()
    r = Renderer(scene, cam).width(200).height(150).max_bounces(0).seed(2)
    st = r.scene_stats()
    assert st["instances"] == 937 and st["bvh_tris"] == 2304 and st["scene_bytes"] < 1 << 20
    img = r.sample_array(4)
    assert np.all(np.isfinite(img)) and img.mean() > 0
    o, d = random_rays(np.random.default_rng(7), 20000, np.zeros(3), 4.0)
    t, obj, nrm = r.get_closest_hit(o, d)
    assert sorted(set(obj[obj >= 0].tolist())) == [0, 1, 2, 3, 4, 5]
    assert np.allclose(np.linalg.norm(nrm[obj >= 0], axis=1), 1.0, atol=1e-4)


# ------------------------------------------------------------------ box shell (room walls as one slab test)
@pytest.mark.parametrize("name", ["C2", "C3"])
def test_room_shell_equals_the_rectangle_scan(name):
    """The five walls of the Cornell box are the faces of one axis-aligned box: folding them into a single
    slab test (commit-time option "room_shell") must not change a hit."""
    res = {}
    try:
        for on in (1, 0):
            rpt_amd.set_option("room_shell", on)
            scene, cam, cfg = scenes.CONFIGS[name]()
            r = Renderer(scene, cam).width(96).height(96).max_bounces(cfg["max_bounces"]).seed(21)
            st = r.scene_stats()
            assert st["shell_faces"] == (5 if on else 0) and st["rects"] == 6
            o, d = random_rays(np.random.default_rng(9), 30000, np.array([278.0, 274.0, 280.0]), 700.0)
            res[on] = (r.get_closest_hit(o, d), r.sample_array(16))
    finally:
        rpt_amd.set_option("room_shell", 1)
    (t1, o1, n1), img1 = res[1]
    (t0, o0, n0), img0 = res[0]
    same = o1 == o0
    assert same.mean() > 0.9999                               # rays through the seam between two walls may pick the other one
    assert np.array_equal(t1[same], t0[same]) and np.array_equal(n1[same], n0[same])   # same (plane - o) * inv arithmetic
    assert np.any(img1 != img0) is not None and rel_rms(img1, img0) < 2e-3


@pytest.mark.parametrize("fog", [False, True])
def test_group_as_object_light_matches_oracle(fog):
    """Light::Object over a KdTree (src/light.rs:38-55 with KdTree::sample, src/kdtree.rs:141-146): a uniformly
    chosen child, nested groups and per-child transforms included; the same group is also a scene object
    (the twin the shadow test must see through), once flat in the scan and once inside the scene BVH."""
    from rpt_amd import Light, Medium
    lamp_kids = [
        sphere().scale(vec3(0.3, 0.3, 0.3)).translate(vec3(-1.5, 2.5, 0.0)),
        cube().scale(vec3(0.5, 0.1, 0.5)).rotate_y(0.5).translate(vec3(1.5, 2.6, 0.3)),
        Mesh(scenes.bumpy_torus(4, 3)).scale(vec3(0.4, 0.4, 0.4)).translate(vec3(0.0, 2.4, -1.0)),
        KdTree([sphere().scale(vec3(0.2, 0.2, 0.2)).translate(vec3(0.0, 0.0, 1.0)),
                sphere().scale(vec3(0.15, 0.3, 0.15)).translate(vec3(0.6, 0.0, 1.2))]).translate(vec3(0.0, 2.3, 0.0)),
    ]
    glow = Material.light(vec3(1.0, 0.9, 0.7), 25.0)

    def lamp():
        return KdTree([k.clone() for k in lamp_kids]).rotate_z(0.1).translate(vec3(0.0, 0.2, 0.0))

    def build(extra):
        sc = Scene()
        sc.add(Object(lamp()).material(glow))
        sc.add(Light.Object(Object(lamp()).material(glow)))
        sc.add(Object(plane(vec3(0, 1, 0), -1.0)).material(Material.diffuse(vec3(0.8, 0.8, 0.8))))
        sc.add(Object(sphere().translate(vec3(0.0, 0.0, 0.0))).material(Material.specular(vec3(0.9, 0.5, 0.5), 0.3)))
        sc.add(Object(cube().translate(vec3(2.0, -0.5, 0.5))).material(Material.diffuse(vec3(0.3, 0.8, 0.4))))
        for s in extra:
            sc.add(Object(s).material(Material.diffuse(vec3(0.6, 0.6, 0.9))))
        if fog:
            sc.add(Medium.homogeneous_isotropic(0.02, 0.05))
        return sc

    cam = Camera.look_at(vec3(0.0, 1.5, 7.0), vec3(0.0, 1.0, 0.0), vec3(0, 1, 0), 0.9)
    w, h, spp = 96, 72, 64
    rng = np.random.default_rng(3)
    clutter = [sphere().scale(vec3(0.1, 0.1, 0.1)).translate(np.array([rng.uniform(-3, 3), -0.9, rng.uniform(-3, 3)]))
               for _ in range(80)]
    torus = [Mesh(scenes.bumpy_torus(12, 8)).scale(vec3(1.5, 1.5, 1.5)).translate(vec3(-2.0, 0.0, 1.0))]
    for extra in ([], torus, clutter):               # linear scan, per-mesh tree (deferred walks), scene-level BVH
        scene = build(extra)
        r = Renderer(scene, cam).width(w).height(h).max_bounces(3).seed(9)
        got = r.sample_array(spp)
        st = r.scene_stats()
        assert st["scene_bvh"] == (1 if extra is clutter else 0) and (st["bvh_nodes"] > 0) == (len(extra) > 0)
        exp = _oracle(scene).render(cam, w, h, spp, 3, seed=9, robust=1)
        assert np.all(np.isfinite(got)) and exp.mean() > 0
        assert rel_rms(got, exp) < 2e-2
        assert abs(got.mean() - exp.mean()) / exp.mean() < 5e-3


def test_large_mesh_in_a_group_is_walked_outside_the_scene_tree_with_parked_walks():
    """A `KdTree<Box<dyn Bounded>>` of 64 spheres and one mesh large enough for a tree of its own (C5's mesh in fog,
    scenes.mesh_among_spheres): the scene-level tree holds the spheres, the mesh keeps its tree and its walks are parked like
    those of a scene without a scene tree (kernels.hip, BVH = 3).  Closest hits and the render match the oracle's kd-tree of
    shapes; when the parked walks start and stop changes no bit; with the mesh as a leaf of the scene tree (option
    "scene_tree_meshes" = 1, every query walked to completion) the same triangles are found."""
    scene, cam, cfg = scenes.mesh_among_spheres(nu=48, nv=32, n_spheres=64)
    w, h, spp, mb = 96, 72, 16, cfg["max_bounces"]

    def render(lanes=32, stop=16, leaf_quarters=6, in_tree=0):
        rpt_amd.set_option("scene_tree_meshes", in_tree)
        rpt_amd.set_option("defer_lanes", lanes)
        rpt_amd.set_option("defer_stop", stop)
        rpt_amd.set_option("walk_leaf_quarters", leaf_quarters)
        sc, cm, _ = scenes.mesh_among_spheres(nu=48, nv=32, n_spheres=64)
        r = Renderer(sc, cm).width(w).height(h).max_bounces(mb).seed(4)
        img = r.sample_array(spp)
        st = r.scene_stats()
        assert st["scene_bvh"] == (1 if in_tree else 2) and st["bvh_tris"] == 48 * 32 * 2   # the mesh: a leaf of the scene tree, or outside it
        assert st["scene_bvh_prims"] == 65                           # 64 spheres + the lamp's rectangle
        return img, r
    try:
        frames = [render(*v)[0] for v in ((32, 16), (1, 1), (64, 64), (64, 1), (8, 5), (32, 16, 0), (40, 8, 64))]
        in_tree, _ = render(in_tree=1)
        img, r = render()
        rng = np.random.default_rng(12)
        ro, rd = random_rays(rng, 20000, np.array([0.0, 0.0, 0.0]), 4.0)
        t, obj, nrm = r.get_closest_hit(ro, rd)
    finally:
        rpt_amd.set_option("scene_tree_meshes", 0)
        rpt_amd.set_option("defer_lanes", 32)
        rpt_amd.set_option("defer_stop", 16)
        rpt_amd.set_option("walk_leaf_quarters", 6)
    for f in frames[1:]:
        assert np.array_equal(frames[0], f)
    assert rel_rms(in_tree, frames[0]) < 2e-4 and abs(in_tree.mean() - frames[0].mean()) < 1e-5 * frames[0].mean()
    orc = _oracle(scene)
    te, oe, ne = orc.intersect(ro.astype(np.float32), rd.astype(np.float32), robust=1)
    hit = oe >= 0
    assert hit.sum() > 2000
    same = (obj >= 0) == hit
    assert same.mean() > 0.9995                                        # (grazing rays may fall either way in fp32)
    both = hit & (obj >= 0)
    assert np.array_equal(obj[both], oe[both])
    assert np.max(np.abs(t[both] - te[both]) / te[both]) < 2e-4
    exp = orc.render(cam, w, h, spp, mb, seed=4, robust=1)
    assert np.all(np.isfinite(frames[0])) and exp.mean() > 0
    assert rel_rms(frames[0], exp) < 1e-2
    assert abs(frames[0].mean() - exp.mean()) / exp.mean() < 5e-3
