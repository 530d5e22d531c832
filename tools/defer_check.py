"""Frames of the per-mesh-tree kernels against frames saved by another build of the library (e.g. before a
restructuring of the kernel).  Two builds differ by fp contraction (median |diff| ~1e-9, a path flips in
~0.1 % of the values); within one build the frame is bit-identical for any schedule
(tests/test_gpu_parity.py::test_deferred_tree_walks_do_not_depend_on_the_schedule).
Usage: python tools/defer_check.py save|check DIR"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpt_amd import Camera, Light, Material, Mesh, Object, Renderer, Scene, plane, scenes, sphere, vec3  # noqa: E402


def frames():
    scene, cam, cfg = scenes.mesh_in_fog(nu=96, nv=96)
    yield "fog", Renderer(scene, cam).width(160).height(120).max_bounces(cfg["max_bounces"]).seed(3).sample_array(48)
    # no medium, two meshes, two object lights with twins, a mirror: max_bounces path
    sc = Scene()
    sc.add(Object(Mesh(scenes.bumpy_torus(40, 24)).scale(vec3(2, 2, 2)).rotate_x(0.6)).material(Material.specular(vec3(0.8, 0.6, 0.3), 0.2)))
    sc.add(Object(Mesh(scenes.bumpy_torus(24, 24)).translate(vec3(1.0, 0.8, 0.5))).material(Material.diffuse(vec3(0.3, 0.6, 0.9))))
    sc.add(Object(plane(vec3(0, 1, 0), -1.0)).material(Material.diffuse(vec3(0.8, 0.8, 0.8))))
    sc.add(Object(sphere().scale(vec3(0.4, 0.4, 0.4)).translate(vec3(-1.2, -0.6, 0.8))).material(Material.mirror()))
    for pos, col in ((vec3(0.0, 3.0, 0.0), vec3(1, 1, 1)), (vec3(2.5, 1.0, 2.0), vec3(1.0, 0.5, 0.2))):
        lamp = Mesh(scenes.bumpy_torus(4, 3)).scale(vec3(0.8, 0.8, 0.8)).translate(pos)
        sc.add(Object(lamp.clone()).material(Material.light(col, 30.0)))
        sc.add(Light.Object(Object(lamp.clone()).material(Material.light(col, 30.0))))
    sc.add(Light.Ambient(vec3(0.02, 0.02, 0.02)))
    cam = Camera.look_at(vec3(0.0, 1.5, 6.0), vec3(0.0, 0.0, 0.0), vec3(0, 1, 0), 0.8)
    r = Renderer(sc, cam).width(160).height(120).max_bounces(4).seed(5)
    yield "surface", r.sample_array(32)
    assert r.scene_stats()["bvh_nodes"] > 0 and r.scene_stats()["scene_bvh"] == 0


mode, out_dir = sys.argv[1], sys.argv[2]
os.makedirs(out_dir, exist_ok=True)
for name, img in frames():
    path = os.path.join(out_dir, f"defer_{name}.npy")
    if mode == "save":
        np.save(path, img)
        print(name, "saved", img.mean())
    else:
        ref = np.load(path)
        same = np.array_equal(ref, img)
        d = np.abs(ref - img)
        print(name, "identical" if same else f"DIFFERENT: max abs {d.max():.3e}, {np.mean(ref != img):.3%} of values, median abs {np.median(d):.3e}, "
              f"> 1e-4: {np.mean(d > 1e-4):.3%}, rel rms {np.sqrt(np.mean(d ** 2) / np.mean(ref ** 2)):.3e}, means {ref.mean():.6f} {img.mean():.6f}")
