"""The reference-epsilon mode (option "epsilon_policy" = 1, rpt_amd/csrc/kernels_f64.hip): fp64, generic shapes in scene
order, t_min = 1e-12 (src/renderer.rs:17, 420) and |hit - dist| < 1e-12 (:348, :396) -- against the oracle's LITERAL policy
(robust = 0), which restates the same reference lines on the CPU.  Both are fp64 evaluations of the same formulas with the
same RNG streams, so images agree far below the fp32 path's tolerance; what is asserted for the epsilon semantics themselves
is statistical: the mean radiance (the fp32 path is +0.77 % / +0.16 % off the literal reference on C2 / C3) and the rates of
self-intersections and near-miss shadow rejections."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from rpt_amd import Camera, Light, Material, Medium, Object, Renderer, RptError, Scene, cube, hex_color, plane, scenes, sphere, vec3
from rpt_amd import _lib
from tests.util import rel_rms

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle.pyoracle import OracleScene
    return OracleScene(scene)


def _eps_renderer(scene, cam):
    scene.set_option("epsilon_policy", 1)
    scene.set_option("counters", 1)
    return Renderer(scene, cam)


def _eps_counters(r):
    out = (C.c_uint64 * 12)()
    _lib.check(_lib.load().rpt_debug_epsilon_counters(r.scene._handle, out))
    names = ["rays", "hits", "self_hits", "shadow_tests", "shadow_pass", "shadow_near", "samples", "vertices",
             "objects_evaluated", "evaluation_rounds", "trips", "live_lanes"]
    return dict(zip(names, [int(v) for v in out]))


@pytest.mark.parametrize("name,size,spp", [("C1", 64, 8), ("C2", 96, 32), ("C3", 96, 32)])
def test_small_renders_follow_the_literal_oracle(name, size, spp):
    scene, cam, cfg = scenes.CONFIGS[name]() if name != "C1" else scenes.spheres_lit()
    r = _eps_renderer(scene, cam).width(size).height(size).max_bounces(cfg["max_bounces"]).seed(3)
    got = r.sample_array(spp)
    cnt = _eps_counters(r)
    exp, oc = _oracle(scene).render(cam, size, size, spp, cfg["max_bounces"], seed=3, robust=0, counters=True)
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    # same formulas, same draws: the frames differ only where libm's last bit flips a decision
    assert rel_rms(got, exp) < 2e-3
    assert abs(got.mean() - exp.mean()) < 1e-4 * exp.mean()
    # the work is the reference's: every count within a fraction of a percent
    for k in ("rays", "hits", "samples", "vertices", "shadow_tests", "shadow_pass"):
        assert abs(cnt[k] - oc[k]) <= 2e-3 * max(oc[k], 1) + 2, (k, cnt[k], oc[k])
    # the epsilon semantics themselves: self-intersections at t ~ 1e-11 and shadow tests that miss the light's own surface
    for k in ("self_hits", "shadow_near"):
        assert abs(cnt[k] - oc[k]) <= 0.05 * oc[k] + 3.0 * np.sqrt(oc[k] + 1.0), (k, cnt[k], oc[k])


def test_all_materials_shapes_and_both_media():
    """Phong, mirror and glass, a sphere and a cube light under a transform, a plane, fog and the glowing fog."""
    for fog in (None, "fog", "glow"):
        sc = Scene()
        sc.add(Object(plane(vec3(0, 1, 0), -1.0)).material(Material.diffuse(hex_color(0xCCCCCC))))
        sc.add(Object(sphere().translate(vec3(-1.5, 0.0, 0.0))).material(Material.mirror()))
        sc.add(Object(sphere().scale(vec3(0.8, 1.1, 0.8)).translate(vec3(0.2, 0.1, 0.3))).material(Material.clear(1.5)))
        sc.add(Object(cube().rotate_y(0.5).translate(vec3(1.7, -0.5, 0.2))).material(Material.specular(hex_color(0xE7A94D), 8.0)))
        for shape, col in ((sphere().scale(vec3(0.4, 0.4, 0.4)).translate(vec3(0.0, 3.0, 1.0)), vec3(1, 1, 1)),
                           (cube().scale(vec3(0.6, 0.1, 0.6)).rotate_z(0.3).translate(vec3(-2.0, 2.5, -0.5)), vec3(1.0, 0.6, 0.3))):
            sc.add(Object(shape.clone()).material(Material.light(col, 40.0)))
            sc.add(Light.Object(Object(shape.clone()).material(Material.light(col, 40.0))))
        sc.add(Light.Ambient(vec3(0.02, 0.02, 0.03)))
        if fog == "fog":
            sc.add(Medium.homogeneous_isotropic(0.02, 0.08))
        elif fog == "glow":
            sc.add(Medium.colored_glowing_fog(0.002, 0.004))
        cam = Camera.look_at(vec3(0.0, 1.5, 7.0), vec3(0.0, 0.3, 0.0), vec3(0, 1, 0), 0.7)
        w, h, spp = 80, 60, 16
        got = _eps_renderer(sc, cam).width(w).height(h).max_bounces(4).seed(9).sample_array(spp)
        exp = _oracle(sc).render(cam, w, h, spp, 4, seed=9, robust=0)
        assert np.all(np.isfinite(got)) and exp.mean() > 0
        assert rel_rms(got, exp) < 5e-3, fog
        assert abs(got.mean() - exp.mean()) < 5e-4 * exp.mean(), fog


@pytest.mark.parametrize("name,npix", [("C2", 4096), ("C3", 2048)])
def test_full_size_means_against_the_literal_reference(name, npix):
    """The configured sizes (C2 512x512x64, C3 1024x1024x256): mean radiance within 1e-4 of the oracle's literal policy on a
    random pixel subset, where the fp32 path sits at +7.7e-3 (C2) and +1.6e-3 (C3)."""
    scene, cam, cfg = scenes.CONFIGS[name]()
    w, h, spp, mb = cfg["width"], cfg["height"], cfg["spp"], cfg["max_bounces"]
    r = _eps_renderer(scene, cam).width(w).height(h).max_bounces(mb).seed(11)
    scene.set_option("timing", 1)
    got = r.sample_array(spp)
    cnt = _eps_counters(r)
    scene.set_option("counters", 0)      # (the diagnostic counters are one atomic per ray: time the kernel without them)
    r._sample_offset = 0
    again = r.sample_array(spp)
    ms = r.timing()[0]
    # the plain build evaluates only the objects a ray's fp32 box test keeps and stops searching at the sampled medium
    # distance; the counters build searches everything: the same frame, bit for bit
    assert np.array_equal(again, got)
    pix = np.sort(np.random.default_rng(5).choice(w * h, size=npix, replace=False)).astype(np.uint32)
    lit, oc = _oracle(scene).render(cam, w, h, spp, mb, seed=11, robust=0, pixels=pix, counters=True)
    lit = lit[pix]
    bias = (got[pix].mean() - lit.mean()) / lit.mean()
    rates = {"self_hits_per_hit": cnt["self_hits"] / max(cnt["hits"], 1), "shadow_near_per_test": cnt["shadow_near"] / max(cnt["shadow_tests"], 1),
             "oracle_self_hits_per_hit": oc["self_hits"] / max(oc["hits"], 1), "oracle_shadow_near_per_test": oc["shadow_near"] / max(oc["shadow_tests"], 1)}
    out = {"config": name, "size": [w, h, spp], "pixels": int(npix), "rel_rms_vs_literal_oracle": rel_rms(got[pix], lit),
           "mean_bias_vs_literal": float(bias), "kernel_ms": ms, "Msamples_per_s": w * h * spp / ms / 1e3, **rates}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/epsilon_full_{name}.json", "w") as f:
        json.dump(out, f, indent=1)
    print(out)
    assert abs(bias) < 1e-4
    assert rel_rms(got[pix], lit) < 1e-3    # north_star: per-pixel L2 error < 1e-3 against the CPU path
    # (the device counts the whole frame, the oracle its pixel subset: the small full-frame renders above compare like with
    # like to 5 %; here the rates have to agree up to the subset's sampling error)
    for a, b in (("self_hits_per_hit", "oracle_self_hits_per_hit"), ("shadow_near_per_test", "oracle_shadow_near_per_test")):
        assert abs(rates[a] - rates[b]) <= 0.25 * rates[b] + 1e-5, (a, rates[a], rates[b])


@pytest.mark.parametrize("name,size,spp", [("C1", 64, 4), ("C2", 128, 16), ("C3", 128, 16)])
def test_box_culling_does_not_change_a_bit(name, size, spp):
    """Option "f64_cull" = 0 evaluates every object for every ray in fp64, as the reference's loop does (src/renderer.rs:
    416-425); the default evaluates only the objects whose padded fp32 box the ray can reach before the distance that
    decides the event.  Neither the frame nor the work counters of the counters build may differ."""
    frames, counts = [], []
    for cull in (1, 0):
        scene, cam, cfg = scenes.CONFIGS[name]() if name != "C1" else scenes.spheres_lit()
        scene.set_option("f64_cull", cull)
        r = _eps_renderer(scene, cam).width(size).height(size).max_bounces(cfg["max_bounces"]).seed(21)
        with_counters = r.sample_array(spp)
        cnt = _eps_counters(r)
        scene.set_option("counters", 0)
        r._sample_offset = 0
        frames.append(r.sample_array(spp))
        assert np.array_equal(frames[-1], with_counters)
        counts.append(cnt)
    assert np.array_equal(frames[0], frames[1])
    for k in ("rays", "hits", "self_hits", "shadow_tests", "shadow_pass", "shadow_near", "samples", "vertices"):
        assert counts[0][k] == counts[1][k], k
    assert counts[0]["objects_evaluated"] < counts[1]["objects_evaluated"]


def test_parked_surface_events_do_not_depend_on_the_threshold():
    """In a medium a lane whose query ends in a surface event waits until "f64_surf_batch" lanes of its wave do (or nothing
    else is left to do).  Every lane draws its own numbers in its own order, so the frame keeps its bits."""
    frames = []
    for batch in (1, 5, 16, 64):
        scene, cam, cfg = scenes.CONFIGS["C3"]()
        scene.set_option("epsilon_policy", 1)
        scene.set_option("f64_surf_batch", batch)
        frames.append(Renderer(scene, cam).width(160).height(96).max_bounces(cfg["max_bounces"]).seed(2).sample_array(24))
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])


def _group_scene(kind):
    from rpt_amd import KdTree, Mesh
    sc = Scene()
    sc.add(Object(plane(vec3(0, 1, 0), -1.2)).material(Material.diffuse(hex_color(0xCCCCCC))))
    if kind == "spheres":      # examples/fractal_spheres.rs in small: groups of spheres, one material per level
        scene, cam, cfg = scenes.fractal_spheres(3)
        return scene, cam, cfg["max_bounces"] + 2
    mesh = Mesh(scenes.bumpy_torus(10, 8, major=0.5, minor=0.22, bump=0.1))
    kids = [sphere().scale(vec3(0.35, 0.35, 0.35)).translate(vec3(-1.1, 0.0, 0.2)),
            cube().rotate_y(0.4).scale(vec3(0.6, 0.5, 0.6)).translate(vec3(1.0, -0.3, 0.0)),
            mesh.rotate_x(0.6).translate(vec3(0.0, 0.2, -0.4)),
            mesh.scale(vec3(0.5, 0.5, 0.5)).translate(vec3(0.2, 1.0, 0.6))]     # the same mesh twice (Arc<Mesh>)
    inner = KdTree(kids)
    if kind == "transformed group":
        sc.add(Object(inner.rotate_z(0.3).scale(vec3(1.1, 0.9, 1.0)).translate(vec3(0.1, 0.2, 0.0))).material(Material.specular(hex_color(0xE7A94D), 6.0)))
    else:                      # a kd-tree of kd-trees, both levels transformed
        outer = KdTree([inner.rotate_y(0.7).translate(vec3(-0.4, 0.0, 0.0)), sphere().scale(vec3(0.3, 0.3, 0.3)).translate(vec3(1.9, 0.9, -0.5)),
                        KdTree([cube().scale(vec3(0.3, 0.3, 0.3)).translate(vec3(-1.9, 0.9, 0.4))])])
        sc.add(Object(outer.scale(vec3(0.9, 0.9, 0.9)).translate(vec3(0.0, 0.1, 0.0))).material(Material.diffuse(hex_color(0x7CA3E7))))
    light = cube().scale(vec3(1.2, 0.05, 1.2)).translate(vec3(0.0, 3.0, 0.5))
    sc.add(Object(light.clone()).material(Material.light(vec3(1, 1, 1), 30.0)))
    sc.add(Light.Object(Object(light.clone()).material(Material.light(vec3(1, 1, 1), 30.0))))
    sc.add(Light.Ambient(vec3(0.03, 0.03, 0.03)))
    cam = Camera.look_at(vec3(0.0, 1.6, 6.0), vec3(0.0, 0.2, 0.0), vec3(0, 1, 0), 0.7)
    return sc, cam, 3


@pytest.mark.parametrize("kind", ["spheres", "transformed group", "group of groups"])
def test_kdtree_groups_follow_the_literal_oracle(kind):
    """`KdTree<Box<dyn Bounded>>` objects (src/kdtree.rs:103-146) in the reference-epsilon mode: a group's children are records of
    their own under the group's ray map and bounds test -- against the oracle, which walks rpt's kd-tree of shapes.  The box
    culling must not change a bit here either."""
    sc, cam, mb = _group_scene(kind)
    w, h, spp = 72, 54, 8
    frames = []
    for cull in (1, 0):
        sc.set_option("f64_cull", cull) if cull else None
        s2, c2, _ = _group_scene(kind)
        s2.set_option("epsilon_policy", 1)
        s2.set_option("f64_cull", cull)
        frames.append(Renderer(s2, c2).width(w).height(h).max_bounces(mb).seed(5).sample_array(spp))
    assert np.array_equal(frames[0], frames[1])
    exp = _oracle(sc).render(cam, w, h, spp, mb, seed=5, robust=0)
    assert np.all(np.isfinite(frames[0])) and exp.mean() > 0
    assert rel_rms(frames[0], exp) < 5e-3, kind
    assert abs(frames[0].mean() - exp.mean()) < 5e-4 * exp.mean(), kind


def test_tile_shards_add_up_to_the_frame_bit_for_bit():
    """The mode shards like the fp32 path (32 x 32 tiles, owner (tx + ty) mod n): every rank's frame is zero outside its tiles,
    the shards' sum is the single-GPU frame exactly, for frame sizes that clip tiles too."""
    for name, w, h, n in (("C3", 200, 136, 3), ("C2", 96, 96, 2)):
        scene, cam, cfg = scenes.CONFIGS[name]()
        scene.set_option("epsilon_policy", 1)
        r = Renderer(scene, cam).width(w).height(h).max_bounces(cfg["max_bounces"]).seed(8)
        whole = r.sample_array(6)
        total = np.zeros_like(whole)
        for k in range(n):
            r._sample_offset = 0
            part = r.shard(k, n).sample_array(6)
            assert np.all((part == 0) | (part == whole))      # a rank writes its own pixels only, with the frame's values
            total += part
        assert np.array_equal(total, whole)


def test_hdri_environment_follows_the_literal_oracle():
    """Environment::Hdri (src/environment.rs:3-52) in the reference-epsilon mode: fp64 texels as given, atan2 / acos lookup, glm::mix --
    with and without fog (in a medium the background counts only beyond 400, src/renderer.rs:198-206)."""
    from rpt_amd import Environment
    w, h = 64, 32
    yy, xx = np.mgrid[0:h, 0:w]
    sky = np.stack([0.2 + 0.8 * xx / (w - 1), 0.3 + 0.5 * (1 - yy / (h - 1)), 0.1 + 0.9 * ((xx // 8 + yy // 8) % 2)], axis=-1)
    for fog in (False, True):
        sc = Scene()
        sc.environment = Environment.Hdri(w, h, sky.reshape(-1, 3))
        sc.add(Object(plane(vec3(0, 1, 0), 0.0)).material(Material.diffuse(hex_color(0xCCCCCC))))
        sc.add(Object(sphere().translate(vec3(-1.2, 1, 0))).material(Material.mirror()))
        sc.add(Object(sphere().translate(vec3(1.2, 1, 0))).material(Material.specular(hex_color(0xE7A94D), 20.0)))
        if fog:
            sc.add(Medium.homogeneous_isotropic(0.0005, 0.002))
        cam = Camera.look_at(vec3(0, 2.0, 6.0), vec3(0, 0.8, 0), vec3(0, 1, 0), 0.7)
        size, spp, mb = 64, 16, 3
        got = _eps_renderer(sc, cam).width(size).height(size).max_bounces(mb).seed(12).sample_array(spp)
        exp = _oracle(sc).render(cam, size, size, spp, mb, seed=12, robust=0)
        assert exp.mean() > 0 and np.all(np.isfinite(got))
        assert rel_rms(got, exp) < 2e-3, fog
        assert abs(got.mean() - exp.mean()) < 2e-4 * exp.mean(), fog


@pytest.mark.parametrize("fog", [False, True])
def test_group_as_object_light_follows_the_literal_oracle(fog):
    """Light::Object over a KdTree (src/light.rs:38-55 with KdTree::sample, src/kdtree.rs:141-146): a uniformly chosen child, nested
    groups and per-level transforms included; the same group is also a scene object, the twin the shadow test must meet at exactly
    the sampled distance (|hit - dist| < 1e-12)."""
    from rpt_amd import KdTree, Mesh
    lamp_kids = [
        sphere().scale(vec3(0.3, 0.3, 0.3)).translate(vec3(-1.5, 2.5, 0.0)),
        cube().scale(vec3(0.5, 0.1, 0.5)).rotate_y(0.5).translate(vec3(1.5, 2.6, 0.3)),
        Mesh(scenes.bumpy_torus(4, 3)).scale(vec3(0.4, 0.4, 0.4)).translate(vec3(0.0, 2.4, -1.0)),
        KdTree([sphere().scale(vec3(0.2, 0.2, 0.2)).translate(vec3(0.0, 0.0, 1.0)),
                KdTree([sphere().scale(vec3(0.15, 0.3, 0.15)).translate(vec3(0.6, 0.0, 1.2))]).rotate_x(0.2)]).translate(vec3(0.0, 2.3, 0.0)),
    ]
    glow = Material.light(vec3(1.0, 0.9, 0.7), 25.0)

    def lamp():
        return KdTree([k.clone() for k in lamp_kids]).rotate_z(0.1).translate(vec3(0.0, 0.2, 0.0))

    sc = Scene()
    sc.add(Object(lamp()).material(glow))
    sc.add(Light.Object(Object(lamp()).material(glow)))
    sc.add(Object(plane(vec3(0, 1, 0), -1.0)).material(Material.diffuse(vec3(0.8, 0.8, 0.8))))
    sc.add(Object(sphere().translate(vec3(0.0, 0.0, 0.0))).material(Material.specular(vec3(0.9, 0.5, 0.5), 0.3)))
    sc.add(Object(cube().translate(vec3(2.0, -0.5, 0.5))).material(Material.diffuse(vec3(0.3, 0.8, 0.4))))
    if fog:
        sc.add(Medium.homogeneous_isotropic(0.02, 0.05))
    cam = Camera.look_at(vec3(0.0, 1.5, 7.0), vec3(0.0, 1.0, 0.0), vec3(0, 1, 0), 0.9)
    w, h, spp = 80, 60, 32
    r = _eps_renderer(sc, cam).width(w).height(h).max_bounces(3).seed(9)
    got = r.sample_array(spp)
    cnt = _eps_counters(r)
    exp, oc = _oracle(sc).render(cam, w, h, spp, 3, seed=9, robust=0, counters=True)
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    assert rel_rms(got, exp) < 5e-3 and abs(got.mean() - exp.mean()) < 5e-4 * exp.mean()
    for k in ("rays", "shadow_tests", "shadow_pass"):
        assert abs(cnt[k] - oc[k]) <= 2e-3 * max(oc[k], 1) + 2, (k, cnt[k], oc[k])


def test_what_the_mode_refuses():
    from rpt_amd import KdTree
    deep = sphere()
    for _ in range(4):                                    # groups nested four deep: one more than the records hold
        deep = KdTree([deep, sphere().translate(vec3(3, 0, 0))])
    sc = Scene()
    sc.add(Object(deep).material(Material.diffuse(vec3(1, 1, 1))))
    sc.set_option("epsilon_policy", 1)
    with pytest.raises(RptError):
        Renderer(sc, Camera.look_at(vec3(0, 0, 5), vec3(0, 0, 0), vec3(0, 1, 0), 0.6)).width(8).height(8).sample_array(1)
    sc = Scene()                                           # ... as a Light::Object as well
    sc.add(Object(sphere()).material(Material.diffuse(vec3(1, 1, 1))))
    sc.add(Light.Object(Object(deep.clone()).material(Material.light(vec3(1, 1, 1), 5.0))))
    sc.set_option("epsilon_policy", 1)
    with pytest.raises(RptError):
        Renderer(sc, Camera.look_at(vec3(0, 0, 5), vec3(0, 0, 0), vec3(0, 1, 0), 0.6)).width(8).height(8).sample_array(1)
