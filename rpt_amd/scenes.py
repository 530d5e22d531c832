"""The five benchmark / parity configurations of BASELINE.json, written against the mirrored
builder API exactly as the reference's example programs are (constants from the cited lines).

    C1  examples/spheres.rs:12-59                       256x256x16,   max_bounces 6
    C2  examples/cornell.rs:15-83                       512x512x64,   max_bounces 2, Filter::Box(1)
    C3  examples/volumetric_pathtrace_lampshade.rs:16-125  1024x1024x256 (headline)
    C4  examples/volumetric_beamphoton_lampshade.rs     (geometry of C3, other sigma; photon pass = next tier)
    C5  examples/dragon.rs:32-73 layout, procedural ~100k-triangle mesh in fog, 2048x2048x1024
"""
import math

import numpy as np

from .api import (Camera, KdTree, Light, Material, Medium, Mesh, Object, Scene, Transformed, cube, hex_color, plane,
                  polygon, sphere, vec3)


def spheres():
    """C1.  Under the reference's semantics this renders black: the only light is a
    Light::Object that is not in scene.objects (KAT 11)."""
    scene = Scene()
    red = Material.specular(hex_color(0xE78999), 0.1)
    yellow = Material.specular(hex_color(0xE7A94D), 0.1)
    green = Material.specular(hex_color(0xB3E7AA), 0.1)
    blue = Material.specular(hex_color(0x7CA3E7), 0.1)
    grey = Material.specular(hex_color(0xAAAAAA), 0.1)
    light_mtl = Material.light(hex_color(0xFFFFFF), 8.0)
    spheres_ = [
        (vec3(0.5, 4.0, 1.0), red),
        (vec3(3.15, -0.7, 1.5), yellow),
        (vec3(0.1, -2.0, 0.6), green),
        (vec3(-1.7, -0.2, 1.1), blue),
        (vec3(1.2, 0.4, 0.5), grey),
    ]
    scene.add(Object(plane(vec3(0.0, 0.0, 1.0), 0.0)).material(Material.diffuse(hex_color(0xE7E7E7))))
    for pos, mtl in spheres_:
        scene.add(Object(sphere().scale(vec3(pos[2], pos[2], pos[2])).translate(pos)).material(mtl))
    scene.add(Light.Object(
        Object(sphere().scale(vec3(2.0, 2.0, 2.0)).translate(vec3(1.2, -1.5, 8.0))).material(light_mtl)))
    camera = Camera.look_at(vec3(0.7166, -9.2992, 2.8803), vec3(0.8673, 0.2095, 0.9557), vec3(0.0, 0.0, 1.0),
                            0.6911).focus(vec3(0.1, -2.0, 0.6), 0.15)
    return scene, camera, dict(width=256, height=256, spp=16, max_bounces=6, filter=0)


def spheres_lit():
    """C1 geometry with the sphere light ALSO added as an object, so that next-event estimation
    can pass the reference's visibility test (exercises Phong, sphere lights, thin lens)."""
    scene, camera, cfg = spheres()
    light_mtl = Material.light(hex_color(0xFFFFFF), 8.0)
    scene.add(Object(sphere().scale(vec3(2.0, 2.0, 2.0)).translate(vec3(1.2, -1.5, 8.0))).material(light_mtl))
    return scene, camera, cfg


def _cornell_walls(scene, white, red, green):
    floor = polygon([vec3(0.0, 0.0, 0.0), vec3(0.0, 0.0, 559.2), vec3(556.0, 0.0, 559.2), vec3(556.0, 0.0, 0.0)])
    ceiling = polygon([vec3(0.0, 548.9, 0.0), vec3(556.0, 548.9, 0.0), vec3(556.0, 548.9, 559.2), vec3(0.0, 548.9, 559.2)])
    back_wall = polygon([vec3(0.0, 0.0, 559.2), vec3(0.0, 548.9, 559.2), vec3(556.0, 548.9, 559.2), vec3(556.0, 0.0, 559.2)])
    right_wall = polygon([vec3(0.0, 0.0, 0.0), vec3(0.0, 548.9, 0.0), vec3(0.0, 548.9, 559.2), vec3(0.0, 0.0, 559.2)])
    left_wall = polygon([vec3(556.0, 0.0, 0.0), vec3(556.0, 0.0, 559.2), vec3(556.0, 548.9, 559.2), vec3(556.0, 548.9, 0.0)])
    scene.add(Object(floor).material(white))
    scene.add(Object(ceiling).material(white))
    scene.add(Object(back_wall).material(white))
    scene.add(Object(left_wall).material(red))
    scene.add(Object(right_wall).material(green))


def _cornell_camera():
    return Camera(eye=vec3(278.0, 273.0, -800.0), direction=vec3(0.0, 0.0, 1.0), up=vec3(0.0, 1.0, 0.0), fov=0.686)


def cornell():
    """C2."""
    scene = Scene()
    white = Material.diffuse(hex_color(0xAAAAAA))
    red = Material.diffuse(hex_color(0xBC0000))
    green = Material.diffuse(hex_color(0x00BC00))
    light_mtl = Material.light(hex_color(0xFFFEFA), 100.0)
    light_rect = polygon([vec3(343.0, 548.8, 227.0), vec3(343.0, 548.8, 332.0), vec3(213.0, 548.8, 332.0),
                          vec3(213.0, 548.8, 227.0)])
    large_box = cube().scale(vec3(165.0, 330.0, 165.0)).rotate_y(2 * math.pi * (-253.0 / 360.0)).translate(
        vec3(368.0, 165.0, 351.0))
    small_box = sphere().scale(vec3(80.0, 80.0, 80.0)).rotate_y(2 * math.pi * (-197.0 / 360.0)).translate(
        vec3(150.0, 82.5, 450.0))
    _cornell_walls(scene, white, red, green)
    scene.add(Object(large_box).material(white))
    scene.add(Object(small_box).material(white))
    scene.add((light_rect, light_mtl))
    return scene, _cornell_camera(), dict(width=512, height=512, spp=64, max_bounces=2, filter=1)


def lampshade(absorb=0.00005, scat=0.003, watts=150.0):
    """C3 (defaults) / the geometry of C4 (absorb=1e-4, scat=1e-3)."""
    scene = Scene()
    white = Material.diffuse(hex_color(0xAAAAAA))
    red = Material.diffuse(hex_color(0xBC0000))
    yellow = Material.diffuse(hex_color(0xBCBC00))
    green = Material.diffuse(hex_color(0x00BC00))
    light_rect = polygon([vec3(330.0, 548.8, 240.0), vec3(330.0, 548.8, 319.0), vec3(226.0, 548.8, 319.0),
                          vec3(226.0, 548.8, 240.0)])
    height, depth, width = 140.0, 105.0, 130.0
    center = vec3(213.0 + 65.0, 548.0, 227.0 + 55.0)
    front_offset = center + vec3(0.0, 0.0, depth / 2.0)
    left_offset = center + vec3(-width / 2.0, 0.0, 0.0)
    back_offset = center + vec3(0.0, 0.0, -depth / 2.0)
    right_offset = center + vec3(width / 2.0, 0.0, 0.0)
    off = 10.0
    front_shade = cube().scale(vec3(130.0 + off * 2.0, height, off)).translate(front_offset)
    left_shade = cube().scale(vec3(off, height, 105.0 + off * 2.0)).translate(left_offset)
    back_shade = cube().scale(vec3(130.0 + off * 2.0, height, off)).translate(back_offset)
    right_shade = cube().scale(vec3(off, height, 105.0 + off * 2.0)).translate(right_offset)
    large_box = cube().scale(vec3(165.0, 330.0, 165.0)).rotate_y(2 * math.pi * (-253.0 / 360.0)).translate(
        vec3(368.0, 165.0, 351.0))
    small_box = cube().scale(vec3(165.0, 165.0, 165.0)).rotate_y(2 * math.pi * (-197.0 / 360.0)).translate(
        vec3(185.0, 82.5, 169.0))
    _cornell_walls(scene, white, red, green)
    scene.add(Object(large_box).material(white))
    scene.add(Object(small_box).material(white))
    scene.add(Object(right_shade).material(yellow))
    scene.add(Object(left_shade).material(yellow))
    scene.add(Object(front_shade).material(yellow))
    scene.add(Object(back_shade).material(yellow))
    light_mtl = Material.light(hex_color(0xFFFEFA), watts)
    scene.add((light_rect, light_mtl))
    scene.add(Medium.homogeneous_isotropic(absorb, scat))
    return scene, _cornell_camera(), dict(width=1024, height=1024, spp=256, max_bounces=10, filter=0)


def bumpy_torus(nu=224, nv=224, major=0.30, minor=0.12, bump=0.2):
    """Deterministic procedural stand-in for the dragon mesh of examples/dragon.rs:12 (an HTTP
    download): a displaced torus, nu*nv*2 triangles (224*224*2 = 100,352) with smooth vertex
    normals.  Returns an (n, 6, 3) fp64 array."""
    u = np.arange(nu) * (2 * np.pi / nu)
    v = np.arange(nv) * (2 * np.pi / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")

    def pos(uu, vv):
        r = minor * (1.0 + bump * np.sin(7 * uu) * np.cos(5 * vv))
        x = (major + r * np.cos(vv)) * np.cos(uu)
        z = (major + r * np.cos(vv)) * np.sin(uu)
        y = r * np.sin(vv)
        return np.stack([x, y, z], axis=-1)

    p = pos(uu, vv)
    h = 1e-5
    du = (pos(uu + h, vv) - pos(uu - h, vv)) / (2 * h)
    dv = (pos(uu, vv + h) - pos(uu, vv - h)) / (2 * h)
    n = np.cross(dv, du)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    i0 = np.arange(nu)[:, None]
    j0 = np.arange(nv)[None, :]
    i1 = (i0 + 1) % nu
    j1 = (j0 + 1) % nv

    def g(a, i, j):
        return a[np.broadcast_to(i, (nu, nv)), np.broadcast_to(j, (nu, nv))]

    # outward-facing winding: (v2-v1)x(v3-v1) along +n
    t1 = np.stack([g(p, i0, j0), g(p, i0, j1), g(p, i1, j1), g(n, i0, j0), g(n, i0, j1), g(n, i1, j1)], axis=2)
    t2 = np.stack([g(p, i0, j0), g(p, i1, j1), g(p, i1, j0), g(n, i0, j0), g(n, i1, j1), g(n, i1, j0)], axis=2)
    tris = np.concatenate([t1.reshape(-1, 6, 3), t2.reshape(-1, 6, 3)], axis=0)
    return np.ascontiguousarray(tris, dtype=np.float64)


def mesh_in_fog(nu=224, nv=224, absorb=0.005, scat=0.045):
    """C5: dragon.rs layout (mesh x3.4, rotate_y pi/2, Phong 0.1; plane y=-1; camera look_at
    (-2.5,4,6.5) -> origin, fov pi/6) with the procedural mesh, in a homogeneous medium, lit by
    an emissive quad that is also an object (otherwise only the ambient term is non-zero)."""
    scene = Scene()
    mesh = Mesh(bumpy_torus(nu, nv))
    scene.add(Object(mesh.scale(vec3(3.4, 3.4, 3.4)).rotate_y(math.pi / 2)).material(
        Material.specular(hex_color(0xB7CA79), 0.1)))
    scene.add(Object(plane(vec3(0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient(vec3(0.01, 0.01, 0.01)))
    light_rect = polygon([vec3(1.5, 6.0, -1.5), vec3(1.5, 6.0, 1.5), vec3(-1.5, 6.0, 1.5), vec3(-1.5, 6.0, -1.5)])
    scene.add((light_rect, Material.light(vec3(1.0, 1.0, 1.0), 60.0)))
    scene.add(Medium.homogeneous_isotropic(absorb, scat))
    camera = Camera.look_at(vec3(-2.5, 4.0, 6.5), vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0), math.pi / 6)
    return scene, camera, dict(width=2048, height=2048, spp=1024, max_bounces=2, filter=0)


def mesh_among_spheres(nu=224, nv=224, n_spheres=64, absorb=0.005, scat=0.045):
    """C5's scene with the mesh inside a `KdTree<Box<dyn Bounded>>` of n_spheres small spheres (src/kdtree.rs:103-146, the
    shape of examples/fractal_teapots.rs:56 with one large mesh): >= 64 bounded primitives give the scene its scene-level
    tree, and the mesh keeps a tree of its own whose walks are parked (kernels.hip, BVH = 3)."""
    scene, camera, cfg = mesh_in_fog(nu, nv, absorb, scat)
    mesh_obj = scene.objects[0]
    kids = [mesh_obj.shape]
    for i in range(n_spheres):   # a double ring of beads around and above the torus
        a = 2 * math.pi * i / n_spheres
        r, y = (1.75, 0.25) if i % 2 == 0 else (1.2, 0.75)
        kids.append(sphere().scale(vec3(0.07, 0.07, 0.07)).translate(vec3(r * math.cos(a), y, r * math.sin(a))))
    out = Scene()
    out.add(Object(KdTree(kids)).material(mesh_obj.material_))
    for o in scene.objects[1:]:
        out.add(o)
    for l in scene.lights:
        out.add(l)
    for m in scene.media:
        out.add(m)
    return out, camera, dict(cfg, spp=256)


def lampshade_beamphoton():
    """C4: examples/volumetric_beamphoton_lampshade.rs:139-164 (photon_point_query_beam_render)."""
    watts = 200_000.0 / (130.0 * 105.0)
    scene, cam, cfg = lampshade(absorb=0.0001, scat=0.001, watts=watts)
    photons = 1_000_000
    cfg = dict(cfg, photons=photons, gather_size=20, gather_size_volume=3, renderer_watts=watts * photons)
    return scene, cam, cfg


def fractal_spheres(levels=5):
    """examples/fractal_spheres.rs: 937 spheres in five `KdTree<Box<dyn Bounded>>` groups (one
    material per recursion level) over a plane; ambient + directional + point light."""
    colors = [0x264653, 0x2A9D8F, 0xE9C46A, 0xF4A261, 0xE76F51][:levels]
    groups = [[] for _ in colors]

    def gen(p, rad, depth, last_dir):
        groups[depth].append(sphere().scale(vec3(rad, rad, rad)).translate(p))
        if depth == len(groups) - 1:
            return
        disp = rad * 7.0 / 5.0
        steps = [(disp, 0, 0), (-disp, 0, 0), (0, disp, 0), (0, -disp, 0), (0, 0, disp), (0, 0, -disp)]
        for i, st in enumerate(steps):
            if last_dir is None or i != (last_dir ^ 1):
                gen(p + vec3(*st), rad * 2.0 / 5.0, depth + 1, i)

    gen(vec3(0.0, 0.0, 0.0), 1.0, 0, None)
    scene = Scene()
    for i, group in enumerate(groups):
        scene.add(Object(KdTree(group)).material(Material.specular(hex_color(colors[i]), 0.25)))
    scene.add(Object(plane(vec3(0.0, 0.0, 1.0), -6.0)).material(Material.diffuse(hex_color(0xFFCCCC))))
    scene.add(Light.Ambient(vec3(0.02, 0.02, 0.02)))
    d = vec3(0.0, -0.65, -1.0)
    scene.add(Light.Directional(vec3(0.6, 0.6, 0.6), d / np.linalg.norm(d)))
    scene.add(Light.Point(vec3(100.0, 100.0, 100.0), vec3(0.0, 5.0, 5.0)))
    cd, cu = vec3(-0.285714, -0.5, -1.0), vec3(0.0, 1.0, -0.5)
    camera = Camera(eye=vec3(2.0, 3.5, 7.0), direction=cd / np.linalg.norm(cd), up=cu / np.linalg.norm(cu),
                    fov=math.pi / 6.0)
    return scene, camera, dict(width=800, height=600, spp=100, max_bounces=0, filter=0)


def fractal_meshes(levels=5, nu=48, nv=24):
    """examples/fractal_teapots.rs ("a kd-tree of kd-trees"): the layout of fractal_spheres with one
    shared mesh (`Arc<Mesh>`) instead of each sphere.  The teapot OBJ is third-party art and does not
    travel; a 2,304-triangle procedural torus (teapot.obj has 2,256 faces) stands in for it."""
    scene, camera, cfg = fractal_spheres(levels)
    mesh = Mesh(bumpy_torus(nu, nv, major=0.62, minor=0.3, bump=0.15))
    out = Scene()
    for o in scene.objects:
        if isinstance(o.shape.base(), KdTree):
            kids = [Transformed(mesh, k.matrix() @ np.diag([0.5, 0.5, 0.5, 1.0])) for k in o.shape.base().shapes]
            out.add(Object(KdTree(kids)).material(o.material_))
        else:
            out.add(o)
    for l in scene.lights:
        out.add(l)
    return out, camera, cfg


CONFIGS = {
    "C1": spheres,
    "C2": cornell,
    "C3": lampshade,
    "C4": lampshade_beamphoton,
    "C5": mesh_in_fog,
    "C5G": mesh_among_spheres,   # not a BASELINE configuration: C5's mesh inside a kd-tree group (scene tree + parked mesh walks)
}
