"""Photon mapping in the reference-epsilon mode (option "epsilon_policy" = 1): the shooting pass and the surface estimate's
visibility rays run in fp64 with rpt's own epsilons (kernels_f64.hip: t_min = 1e-12, a gathered photon counts unless
`len > hit.time`, src/photon.rs:357-361); maps, k-nearest selection and volume estimates are the fp32 machinery's.  Checked against
the oracle's literal restatement (robust = 0).  The comparison is statistical where photon chains are concerned: whether a photon
meets the surface it has just left again is decided by the last bits of its ray, and the device's libm and the host's differ there,
so the two maps hold different photons wherever a chain met such a self-intersection."""
import json
import os

import numpy as np
import pytest

from rpt_amd import Renderer, RptError, scenes
from tests.util import rel_rms

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle.pyoracle import OracleScene
    return OracleScene(scene)


def _renderers(name, size, n, gather=20, gather_volume=3):
    out = []
    for eps in (1, 0):
        scene, cam, cfg = scenes.CONFIGS[name]()
        if eps:
            scene.set_option("epsilon_policy", 1)
        out.append(Renderer(scene, cam).width(size).height(size).watts(14.65 * n).gather_size(gather).gather_size_volume(gather_volume))
    return out


@pytest.mark.parametrize("name", ["C4", "C2"])
def test_shooting_and_camera_pass_follow_the_literal_reference(name):
    """20 k photons, 64 x 64 x 16 spp, beam x point map.  The fp32 policy's frame lies ~1.1 % above the literal reference's (C4 and
    C2 alike); the reference-epsilon mode's lies within 0.1 % of it (measured: C4 3.8e-3 / -0.10 %, C2 6.6e-3 / +0.04 %), and the rest of
    its distance is that of two maps whose chains parted at the self-intersections (the robust oracle's distance to the literal one,
    with the bias taken out, is 1.4e-2 / 4.9e-2)."""
    n, size, spp = 20000, 64, 16
    r_eps, r_f32 = _renderers(name, size, n)
    scene, cam, cfg = scenes.CONFIGS[name]()
    lit = _oracle(scene).photon_map(n, 1, 14.65 * n, 20, 3, seed=7, robust=0)
    rob = _oracle(scene).photon_map(n, 1, 14.65 * n, 20, 3, seed=7, robust=1)
    st = r_eps.seed(7).photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    st32 = r_f32.seed(7).photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    n_lit, n_rob = [len(lit.photons(0)), len(lit.photons(1))], [len(rob.photons(0)), len(rob.photons(1))]
    print({"device_eps": [st["surface"], st["volume"]], "device_f32": [st32["surface"], st32["volume"]], "literal": n_lit, "robust": n_rob})
    # the self-intersections store a photon a second time: the literal map holds more surface photons than the robust one
    assert n_lit[0] > n_rob[0]
    assert abs(st["surface"] - n_lit[0]) < 0.5 * (n_lit[0] - n_rob[0]) + 3
    assert abs(st["volume"] - n_lit[1]) <= 2e-3 * max(n_lit[1], 1)
    got = r_eps.seed(0).photon_sample_array(spp)
    got32 = r_f32.seed(0).photon_sample_array(spp)
    exp = lit.render(cam, size, size, spp, seed=0)
    exp_rob = rob.render(cam, size, size, spp, seed=0)
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    err, bias = rel_rms(got, exp), (got.mean() - exp.mean()) / exp.mean()
    err32, bias32 = rel_rms(got32, exp), (got32.mean() - exp.mean()) / exp.mean()
    rob_bias = (exp_rob.mean() - exp.mean()) / exp.mean()
    rob_spread = rel_rms(exp_rob / (1.0 + rob_bias), exp)
    print({"eps_vs_literal": [err, bias], "f32_vs_literal": [err32, bias32], "robust_oracle_vs_literal": [rel_rms(exp_rob, exp), rob_bias],
           "robust_oracle_vs_literal_bias_removed": rob_spread})
    assert bias32 > 8e-3                      # what the mode is for
    assert abs(bias) < 2e-3
    assert err < 0.6 * rob_spread + 1e-3


def _device_photons(r):
    """The device's map as the oracle takes photon lists: shooting order; surface positions as the shooting pass holds them (fp64)."""
    ps = r.photon_map_download(0).astype(np.float64)
    ps[:, :3] = r.photon_positions64()
    return ps, r.photon_map_download(1).astype(np.float64)


@pytest.mark.parametrize("name", ["C4", "C2"])
def test_camera_pass_on_the_devices_own_photons(name):
    """The camera pass alone: the oracle builds its maps over the DEVICE's photons (oracle test hook photon_map_from_photons) and runs
    its literal camera pass -- no parted chains between the two frames, only the pass's own arithmetic: the fp32 selection and beam
    estimate, and the visibility rays' lottery wherever a last bit of the query point differs."""
    n, size, spp = 20000, 64, 16
    r_eps, r_f32 = _renderers(name, size, n)
    scene, cam, cfg = scenes.CONFIGS[name]()
    r_eps.seed(7).photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    ps, pv = _device_photons(r_eps)
    assert np.abs(ps[:, :3] - r_eps.photon_map_download(0)[:, :3]).max() < 1e-4      # (the records hold these positions rounded to fp32)
    pm = _oracle(scene).photon_map_from_photons(n, 1, 14.65 * n, 20, 3, ps, pv, robust=0)
    exp = pm.render(cam, size, size, spp, seed=0)
    got = r_eps.seed(0).photon_sample_array(spp)
    err, bias = rel_rms(got, exp), (got.mean() - exp.mean()) / exp.mean()
    # the same photons under the robust policy, for scale: what the epsilon policy alone is worth in the camera pass
    exp_rob = _oracle(scene).photon_map_from_photons(n, 1, 14.65 * n, 20, 3, ps, pv, robust=1).render(cam, size, size, spp, seed=0)
    print({"name": name, "eps_vs_literal_same_photons": [err, bias], "robust_vs_literal_same_photons": [rel_rms(exp_rob, exp), (exp_rob.mean() - exp.mean()) / exp.mean()]})
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    # measured: C4 9.4e-6 / -3e-6, C2 1.2e-4 / -5e-6 (the robust policy on the same photons: 1.4e-2 / +1.1 %, 3.5e-2 / +1.3 %)
    assert err < 5e-4 and abs(bias) < 5e-5
    # The visibility rays between two points of one axis-aligned wall have a direction component that is exactly 0; where such a ray
    # starts in the plane of a cube's face (the boxes stand on the floor), the reference's slab test divides 0 by 0 and reports a hit
    # for a ray that passes beside the cube (src/shape/cube.rs:23-60) -- 2.5 % of the floor's radiance next to the boxes.  The box
    # culling keeps such cubes (kernels_f64.hip, cull32): the frame is the full scan's bit for bit.
    r_eps.scene.set_option("f64_cull", 0)
    r_eps._sample_offset = 0
    assert np.array_equal(got, r_eps.photon_sample_array(spp))


def test_gather_lists_in_global_memory_and_records_outside_lds():
    """The other instantiations: a gather size above what the fp32 pass keeps in LDS (one k-nearest search per lane hands the
    selection over), and a scene whose triangle records do not fit the fp64 kernels' LDS tables (a 960-triangle mesh, a plane, fog) --
    the camera pass over the device's own photons against the oracle's literal pass."""
    from rpt_amd import Camera, Light, Material, Medium, Mesh, Object, Scene, plane, polygon, vec3
    n, size, spp = 20000, 48, 8
    scene, cam, cfg = scenes.CONFIGS["C2"]()
    scene.set_option("epsilon_policy", 1)
    r = Renderer(scene, cam).width(size).height(size).watts(14.65 * n).gather_size(100).gather_size_volume(3).seed(2)
    r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    ps, pv = _device_photons(r)
    scene0, _, _ = scenes.CONFIGS["C2"]()
    exp = _oracle(scene0).photon_map_from_photons(n, 1, 14.65 * n, 100, 3, ps, pv, robust=0).render(cam, size, size, spp, seed=0)
    got = r.seed(0).photon_sample_array(spp)
    print({"gather 100": [rel_rms(got, exp), (got.mean() - exp.mean()) / exp.mean()]})
    assert rel_rms(got, exp) < 5e-4 and abs(got.mean() - exp.mean()) < 5e-5 * exp.mean()

    def mesh_scene():
        sc = Scene()
        sc.add(Object(Mesh(scenes.bumpy_torus(24, 20)).scale(vec3(1.5, 1.5, 1.5)).rotate_x(0.5)).material(Material.diffuse(vec3(0.8, 0.6, 0.3))))
        sc.add(Object(plane(vec3(0, 1, 0), -1.2)).material(Material.diffuse(vec3(0.7, 0.7, 0.7))))
        sc.add(Object(polygon([vec3(1.0, 3.0, -1.0), vec3(1.0, 3.0, 1.0), vec3(-1.0, 3.0, 1.0), vec3(-1.0, 3.0, -1.0)])).material(Material.light(vec3(1, 1, 1), 40.0)))
        sc.add(Light.Object(Object(polygon([vec3(1.0, 3.0, -1.0), vec3(1.0, 3.0, 1.0), vec3(-1.0, 3.0, 1.0), vec3(-1.0, 3.0, -1.0)]))
                            .material(Material.light(vec3(1, 1, 1), 40.0))))
        sc.add(Medium.homogeneous_isotropic(0.02, 0.08))
        return sc
    cam = Camera.look_at(vec3(0.0, 1.5, 5.0), vec3(0.0, 0.0, 0.0), vec3(0, 1, 0), 0.8)
    n, size, spp, watts = 6000, 32, 4, 300.0
    sc = mesh_scene()
    sc.set_option("epsilon_policy", 1)
    r = Renderer(sc, cam).width(size).height(size).watts(watts).gather_size(12).gather_size_volume(3).seed(11)
    st = r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    lit = _oracle(mesh_scene()).photon_map(n, 1, watts, 12, 3, seed=11, robust=0)
    assert abs(st["surface"] - len(lit.photons(0))) <= 0.01 * len(lit.photons(0)) + 5 and abs(st["volume"] - len(lit.photons(1))) <= 0.01 * len(lit.photons(1)) + 5
    ps, pv = _device_photons(r)
    exp = _oracle(mesh_scene()).photon_map_from_photons(n, 1, watts, 12, 3, ps, pv, robust=0).render(cam, size, size, spp, seed=0)
    got = r.seed(0).photon_sample_array(spp)
    print({"mesh in fog": [rel_rms(got, exp), (got.mean() - exp.mean()) / exp.mean()], "photons": [st["surface"], st["volume"]]})
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    assert rel_rms(got, exp) < 2e-3 and abs(got.mean() - exp.mean()) < 5e-4 * exp.mean()


def test_the_frame_does_not_depend_on_the_slices():
    """The camera pass of a call runs slice by slice when the per-sample selections of all its samples would not fit the budget
    (option "f64_photon_slice" forces small slices here): the same samples, added up in another order."""
    n, size = 5000, 48
    r_eps, _ = _renderers("C4", size, n)
    r_eps.seed(3).photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    whole = r_eps.seed(0).photon_sample_array(24)
    for slice_ in (16, 5, 64):
        r_eps.scene.set_option("f64_photon_slice", slice_)
        r_eps._sample_offset = 0
        got = r_eps.photon_sample_array(24)
        rel = np.abs(got - whole) / np.abs(whole)
        print(slice_, "max rel", rel.max(), "entries above 1e-6:", (rel > 1e-6).sum(), "of", rel.size)
        # the partial sums of the volume estimate are fp32 and regroup with the slices (1e-7); measured besides: 2 of 6,912 entries off by
        # 1e-4 -- one photon sphere accepted by one grouping's beam walk and not by the other's
        assert (rel > 2e-6).sum() <= 4 and rel.max() < 1e-3, slice_
    r_eps.scene.set_option("f64_photon_slice", 0)
    r_eps._sample_offset = 0          # ... and two calls with the sample offset moved on
    a = r_eps.photon_sample_array(16)
    b = r_eps.photon_sample_array(8)
    rel = np.abs((16 * a + 8 * b) / 24 - whole) / np.abs(whole)
    assert (rel > 2e-6).sum() <= 4 and rel.max() < 1e-3


def test_the_camera_pass_is_a_function_of_its_inputs():
    n, size, spp = 20000, 64, 16
    r_eps, _ = _renderers("C4", size, n)
    r_eps.seed(7).photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    first = r_eps.seed(0).photon_sample_array(spp)
    for _ in range(4):
        r_eps._sample_offset = 0
        assert np.array_equal(first, r_eps.photon_sample_array(spp))


@pytest.mark.parametrize("kind", [Renderer.PHOTON_MAP, Renderer.PHOTON_BEAM_BEAM])
def test_the_other_two_estimators(kind):
    """photon x photon (the distance drawn first decides between the volume gather and the surface estimate, src/photon.rs:384-438)
    and beam x beam (thinned volume photons, :779-787) through the same two fp64 passes."""
    n, size, spp = 20000, 48, 16
    r_eps, r_f32 = _renderers("C4", size, n, 20, 8)
    scene, cam, cfg = scenes.CONFIGS["C4"]()
    lit = _oracle(scene).photon_map(n, kind, 14.65 * n, 20, 8, seed=5, robust=0)
    r_eps.seed(5).photon_map_build(n, kind)
    got = r_eps.seed(0).photon_sample_array(spp)
    exp = lit.render(cam, size, size, spp, seed=0)
    err, bias = rel_rms(got, exp), (got.mean() - exp.mean()) / exp.mean()
    print({"kind": kind, "eps_vs_literal": [err, bias]})
    assert np.all(np.isfinite(got)) and exp.mean() > 0
    # (the beam x beam map keeps one volume photon in a thousand -- ~40 beams here --, drawn from a side stream in path order: a chain
    # that parts from the oracle's at a self-intersection changes which beams are kept; measured +0.2 % / 5.1e-3, against -5e-5 / 1.3e-3
    # for the photon x photon map)
    if kind == Renderer.PHOTON_BEAM_BEAM:
        assert abs(bias) < 3e-2 and err < 5e-2
    else:
        assert abs(bias) < 2e-3 and err < 5e-3


def test_c4_at_its_configured_size_against_the_literal_reference():
    """Config C4 as the bench runs it (1 M photons, 1024 x 1024 x 256 spp) on a 2,048-pixel subset, against the oracle's literal map
    and camera pass; written to gpurun_out/parity_full_C4eps.json."""
    scene, cam, cfg = scenes.CONFIGS["C4"]()
    scene.set_option("epsilon_policy", 1)
    n, watts, w, h = cfg["photons"], cfg["renderer_watts"], 1024, 1024
    r = Renderer(scene, cam).width(w).height(h).watts(watts).gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).seed(7)
    st = r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    pix = np.sort(np.random.default_rng(5).choice(w * h, size=2048, replace=False)).astype(np.uint32)
    got = r.seed(0).photon_sample_array(256)[pix]
    scene0, _, _ = scenes.CONFIGS["C4"]()
    lit = _oracle(scene0).photon_map(n, 1, watts, cfg["gather_size"], cfg["gather_size_volume"], seed=7, robust=0)
    exp = lit.render(cam, w, h, 256, seed=0, pixels=pix)[pix]
    counts = [len(lit.photons(0)), len(lit.photons(1))]
    err, bias = rel_rms(got, exp), (got.mean() - exp.mean()) / exp.mean()
    # ... and the camera pass alone: the oracle's literal pass over the device's own photons
    ps, pv = _device_photons(r)
    own = _oracle(scene0).photon_map_from_photons(n, 1, watts, cfg["gather_size"], cfg["gather_size_volume"], ps, pv, robust=0)
    exp_own = own.render(cam, w, h, 256, seed=0, pixels=pix)[pix]
    err_own, bias_own = rel_rms(got, exp_own), (got.mean() - exp_own.mean()) / exp_own.mean()
    out = {"config": "C4 (epsilon_policy = 1)", "size": [w, h, 256], "pixels": int(len(pix)), "photons": n,
           "rel_rms_vs_literal_oracle_256spp": err, "mean_bias_vs_literal_256spp": float(bias),
           "camera_pass_alone_rel_rms_vs_literal_oracle_on_the_device_photons": err_own, "camera_pass_alone_mean_bias": float(bias_own),
           "stored_photons_device": [st["surface"], st["volume"]], "stored_photons_literal_oracle": counts}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_full_C4eps.json", "w") as f:
        json.dump(out, f, indent=1)
    print(out)
    assert np.all(np.isfinite(got))
    assert abs(st["surface"] - counts[0]) < 1e-3 * counts[0] and abs(st["volume"] - counts[1]) < 1e-3 * counts[1]
    assert abs(bias) < 2e-3 and err < 6e-3
    assert abs(bias_own) < 5e-5 and err_own < 3e-4


def test_builder_entry_point_and_tile_shards():
    """Renderer::photon_point_query_beam_render (src/photon.rs:642-644) through the mode, and the frame over two tile shards: every rank
    builds the whole map (same seed, same streams), the camera pass of a rank covers its tiles, and the shards add up to the frame."""
    size = 64

    def make(rank=0, count=1):
        s2, c2, _ = scenes.CONFIGS["C4"]()
        s2.set_option("epsilon_policy", 1)
        return Renderer(s2, c2).width(size).height(size).num_samples(2).gather_size(20).gather_size_volume(3) \
            .watts(14.65 * 4000).seed(1).shard(rank, count)
    img = make().photon_point_query_beam_render(4000)
    assert img.shape == (size, size, 3) and img.dtype == np.uint8 and img.max() > 0
    full = make()
    full.photon_map_build(4000, 1)
    whole = full.photon_sample_array(2)
    parts = []
    for rk in range(2):
        r = make(rk, 2)
        r.photon_map_build(4000, 1)
        parts.append(r.photon_sample_array(2))
    assert np.all(parts[0][parts[1] != 0] == 0)               # disjoint tiles
    assert np.array_equal(parts[0] + parts[1], whole)


def test_what_photon_mapping_in_the_mode_refuses():
    scene, cam, cfg = scenes.CONFIGS["C4"]()
    scene.set_option("epsilon_policy", 1)
    r = Renderer(scene, cam).width(16).height(16).watts(1000.0)
    with pytest.raises(RptError):
        r.photon_shoot(1000, Renderer.PHOTON_POINT_BEAM, 0, 2)   # the 48-byte records do not carry the fp64 positions
