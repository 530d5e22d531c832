"""SHA-256 of the fp64 frames of a fixed set of renders: a kernel restructuring that must not change one bit of any
image (scheduling, deferral, work decomposition) is checked by running this before and after and diffing the output.
Usage: python tools/frame_hashes.py [out.json]"""
import hashlib
import json
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
from rpt_amd import Light, Material, Medium, Mesh, Object, Renderer, Scene, Camera, plane, sphere, cube, vec3, scenes  # noqa: E402


def two_lights(fog):
    sc = Scene()
    sc.add(Object(Mesh(scenes.bumpy_torus(40, 24)).scale(vec3(2, 2, 2)).rotate_x(0.6)).material(Material.specular(vec3(0.8, 0.6, 0.3), 0.2)))
    sc.add(Object(Mesh(scenes.bumpy_torus(24, 24)).translate(vec3(1.0, 0.8, 0.5))).material(Material.diffuse(vec3(0.3, 0.6, 0.9))))
    sc.add(Object(plane(vec3(0, 1, 0), -1.0)).material(Material.diffuse(vec3(0.8, 0.8, 0.8))))
    sc.add(Object(sphere().scale(vec3(0.4, 0.4, 0.4)).translate(vec3(-1.5, -0.6, 1.0))).material(Material.mirror()))
    sc.add(Object(cube().scale(vec3(0.5, 0.5, 0.5)).rotate_y(0.5).translate(vec3(1.8, -0.75, 1.5))).material(Material.clear(1.5)))
    for pos, col in ((vec3(0.0, 3.0, 0.0), vec3(1, 1, 1)), (vec3(2.5, 1.0, 2.0), vec3(1.0, 0.5, 0.2))):
        lamp = Mesh(scenes.bumpy_torus(4, 3)).scale(vec3(0.8, 0.8, 0.8)).translate(pos)
        sc.add(Object(lamp.clone()).material(Material.light(col, 30.0)))
        sc.add(Light.Object(Object(lamp.clone()).material(Material.light(col, 30.0))))
    sc.add(Light.Ambient(vec3(0.02, 0.02, 0.02)))
    if fog:
        sc.add(Medium.homogeneous_isotropic(0.02, 0.1))
    return sc, Camera.look_at(vec3(0.0, 1.5, 6.0), vec3(0.0, 0.0, 0.0), vec3(0, 1, 0), 0.8), dict(max_bounces=3)


CASES = {
    "C1lit": (scenes.spheres_lit, 96, 96, 16),
    "C2": (scenes.cornell, 192, 192, 32),
    "C3": (scenes.lampshade, 256, 256, 40),
    "C3_offset": (scenes.lampshade, 100, 70, 7),
    "C5small": (lambda: scenes.mesh_in_fog(nu=48, nv=48), 128, 128, 24),
    "two_lights_fog": (lambda: two_lights(True), 96, 72, 24),
    "two_lights": (lambda: two_lights(False), 96, 72, 24),
    "fractal": (scenes.fractal_spheres, 160, 120, 8),
    "fractal_meshes": (lambda: scenes.fractal_meshes(levels=3), 160, 120, 8),
}


def main():
    out = {}
    for name, (make, w, h, spp) in CASES.items():
        scene, cam, cfg = make()
        mb = max(cfg["max_bounces"], 2) if name.startswith("fractal") else cfg["max_bounces"]
        r = Renderer(scene, cam).width(w).height(h).max_bounces(mb).seed(5)
        img = r.sample_array(spp)
        img2 = r.sample_array(3)    # a second batch continues the sample numbering
        out[name] = {"sha": hashlib.sha256(np.ascontiguousarray(img).tobytes() + np.ascontiguousarray(img2).tobytes()).hexdigest()[:24],
                     "mean": float(img.mean())}
        print(name, out[name], flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
