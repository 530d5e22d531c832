"""Host-side mirror of rpt's builder API (rpt_amd/api.py): pure numpy, no GPU."""
import math

import numpy as np
import pytest

from rpt_amd import (Buffer, Camera, Filter, Light, Material, Medium, Mesh, Object, Renderer, RptError, Scene,
                     Transformed, Triangle, color_bytes, cube, hex_color, plane, polygon, scenes, sphere, vec3)


def test_colors_work():  # the reference's own unit test, src/color.rs:30-38
    assert list(color_bytes(hex_color(0x000000))) == [0, 0, 0]
    assert list(color_bytes(hex_color(0xFFFFFF))) == [255, 255, 255]
    assert list(color_bytes(hex_color(0xFF0000))) == [255, 0, 0]
    assert list(color_bytes(vec3(0.5, 2.0, -1.0))) == [186, 255, 0]      # truncating `as u8`


def test_transform_chaining_left_multiplies_and_does_not_nest():
    s = cube().scale(vec3(2, 3, 4)).rotate_y(math.pi / 2).translate(vec3(5, 6, 7))
    assert isinstance(s, Transformed) and not isinstance(s.shape, Transformed)
    p = s.m @ np.array([0.5, 0.5, 0.5, 1.0])          # S -> (1,1.5,2); R_y(90deg) -> (2,1.5,-1); T
    assert np.allclose(p[:3], [7.0, 7.5, 6.0])
    r = sphere().rotate(math.pi / 2, vec3(0, 0, 2))   # glm::rotate normalises the axis
    assert np.allclose(r.m[:3, :3] @ [1, 0, 0], [0, 1, 0])
    assert np.allclose(sphere().rotate_x(math.pi / 2).m[:3, :3] @ [0, 1, 0], [0, 0, 1])
    assert np.allclose(sphere().rotate_z(math.pi / 2).m[:3, :3] @ [1, 0, 0], [0, 1, 0])
    m = np.arange(16.0).reshape(4, 4)
    assert np.array_equal(sphere().transform(m).m, m)


def test_polygon_is_a_fan_with_face_normals():
    q = polygon([vec3(0, 0, 0), vec3(1, 0, 0), vec3(1, 1, 0), vec3(0, 1, 0)])
    assert isinstance(q, Mesh) and q.tris.shape == (2, 6, 3)
    assert np.array_equal(q.tris[0, :3], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    assert np.array_equal(q.tris[1, :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(q.tris[:, 3:], np.broadcast_to([0, 0, 1.0], (2, 3, 3)))
    t = Triangle.from_vertices(vec3(0, 0, 0), vec3(0, 0, 2), vec3(3, 0, 0))
    assert np.array_equal(t.n1, [0, 1, 0])


def test_material_constructors_follow_the_reference():
    m = Material.specular(hex_color(0xFFFFFF), 0.1)
    assert m.kind == Material.PHONG and m.shininess == 0.1       # "roughness" is stored as shininess
    assert Material.metallic(vec3(1, 1, 1), 0.4).kind == Material.PHONG
    assert Material.clear(1.5, 0.2).kind == Material.TRANSMISSIVE and Material.clear(1.5, 0.2).ior == 1.5
    lm = Material.light(vec3(1, 1, 1), 8.0)
    assert lm.kind == Material.LAMBERTIAN and lm.emittance() == 8.0
    assert Material.mirror().emittance() == 0.0 and np.array_equal(Material.transmissive(1.3).color(), [0, 0, 0])
    d = Material()
    assert d.kind == Material.LAMBERTIAN and np.array_equal(d.albedo, [0.5, 0.5, 0.5])


def test_scene_add_dispatch_and_light_object_pairs():
    sc = Scene()
    sc.add(Object(sphere()))
    sc.add(Light.Ambient(vec3(0.1, 0.1, 0.1)))
    sc.add(Medium.homogeneous_isotropic(1e-4, 1e-3))
    quad = polygon([vec3(0, 0, 0), vec3(1, 0, 0), vec3(1, 1, 0), vec3(0, 1, 0)])
    sc.add((quad, Material.light(vec3(1, 1, 1), 5.0)))            # scene.rs:57-65
    sc.add((cube().scale(vec3(2, 2, 2)), Material.light(vec3(1, 1, 1), 5.0)))   # scene.rs:67-75
    assert len(sc.objects) == 3 and len(sc.lights) == 3 and len(sc.media) == 1
    assert sc.lights[1].kind == Light.OBJECT and sc.lights[1].object.shape is not sc.objects[1].shape
    with pytest.raises(TypeError):
        sc.add((sphere(), Material.light(vec3(1, 1, 1), 5.0)))    # no SceneAdd impl for (Sphere, Material)
    with pytest.raises(TypeError):
        sc.add("nonsense")
    with pytest.raises(TypeError):
        Object("not a shape")


def test_camera_look_at_and_focus():
    c = Camera.look_at(vec3(0, 0, 5), vec3(0, 0, 0), vec3(0, 1, 1), 0.7)
    assert np.allclose(c.direction, [0, 0, -1]) and np.allclose(c.up, [0, 1, 0]) and c.aperture == 0.0
    c.focus(vec3(3, 0, 1), 0.15)
    assert c.focal_distance == 4.0 and c.aperture == 0.15
    d = Camera()
    assert np.array_equal(d.eye, [0, 0, 10]) and d.fov == math.pi / 6


def test_renderer_defaults_and_builder():
    sc, cam, _ = scenes.cornell()
    r = Renderer(sc, cam)
    assert (r.width_, r.height_, r.exposure_value_, r.max_bounces_, r.num_samples_) == (800, 600, 0.0, 0, 1)
    assert (r.gather_size_, r.gather_size_volume_, r.watts_, r.filter_.radius) == (50, 50, 100.0, 0)
    r2 = r.width(64).height(32).max_bounces(3).num_samples(7).exposure_value(1.5).filter(Filter.Box(2)).seed(9)
    assert r2 is r and (r.width_, r.height_, r.max_bounces_, r.num_samples_, r.seed_) == (64, 32, 3, 7, 9)


def test_buffer_box_filter_matches_the_nested_loops():
    rng = np.random.default_rng(0)
    w, h = 7, 5
    for radius in (0, 1, 2):
        b = Buffer(w, h, Filter.Box(radius))
        batches = [rng.uniform(0, 1, size=(w * h, 3)) for _ in range(3)]
        for s in batches:
            b.add_samples(s)
        got = b._filtered()
        for y in range(h):
            for x in range(w):
                col, cnt = np.zeros(3), 0
                for i in range(max(x - radius, 0), x + radius + 1):       # buffer.rs:79-88
                    for j in range(max(y - radius, 0), y + radius + 1):
                        if i < w and j < h:
                            col += sum(s[j * w + i] for s in batches)
                            cnt += len(batches)
                assert np.allclose(got[y, x], col / cnt)
        assert b.image().shape == (h, w, 3) and b.image().dtype == np.uint8
    with pytest.raises(AssertionError):
        Buffer(2, 2).add_samples(np.zeros((3, 3)))


def test_buffer_variance_is_across_batches():
    b = Buffer(2, 1)
    b.add_samples([[0, 0, 0], [1, 1, 1]])
    b.add_samples([[2, 0, 0], [1, 1, 1]])
    # pixel 0: mean (1,0,0), squared distances 1+1, / (n-1) = 2; pixel 1: 0 -> mean 1.0 (buffer.rs:59-73)
    assert b.variance() == 1.0


def test_config_scenes_have_the_documented_shape():
    sc, cam, cfg = scenes.lampshade()
    assert len(sc.objects) == 12 and len(sc.lights) == 1 and len(sc.media) == 1
    assert sum(o.shape.base().tris.shape[0] for o in sc.objects if isinstance(o.shape.base(), Mesh)) == 12
    assert (cfg["width"], cfg["height"], cfg["spp"]) == (1024, 1024, 256)
    sc, cam, cfg = scenes.cornell()
    assert len(sc.objects) == 8 and cfg["max_bounces"] == 2 and cfg["filter"] == 1
    sc, cam, cfg = scenes.spheres()
    assert len(sc.objects) == 6 and cam.aperture == 0.15 and cfg["max_bounces"] == 6
    tris = scenes.bumpy_torus()
    assert tris.shape == (100352, 6, 3) and np.all(np.isfinite(tris))
    e0, e1 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    gn = np.cross(e0, e1)
    assert np.all(np.linalg.norm(gn, axis=1) > 0)
    assert np.all(np.einsum("ij,ij->i", gn, tris[:, 3]) > 0)      # winding agrees with the vertex normals


def test_load_obj_follows_the_reference_parser(tmp_path):
    import io
    import struct
    from rpt_amd import load_obj, load_stl
    text = """# a quad, a triangle with normals, relative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0.5 0.5
vn 0 0 1
vn 0 1 0
usemtl ignored
f 1 2 3 4
f 1//1 2//1 3//2
f -4/1 -3/1 -2
"""
    m = load_obj(io.StringIO(text))
    assert m.tris.shape == (4, 6, 3)                                   # quad -> 2 (fan), then 1 + 1
    assert np.array_equal(m.tris[0, :3], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    assert np.array_equal(m.tris[1, :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(m.tris[0, 3:], np.broadcast_to([0, 0, 1.0], (3, 3)))      # face normal
    assert np.array_equal(m.tris[2, 3:], [[0, 0, 1], [0, 0, 1], [0, 1, 0]])          # explicit vn
    assert np.array_equal(m.tris[3, :3], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])          # negative indices
    assert np.array_equal(m.tris[3, 3:], np.broadcast_to([0, 0, 1.0], (3, 3)))      # one corner lacks vn -> face normal
    with pytest.raises(ValueError):
        load_obj(io.StringIO("v 0 0 0\nf 1 2 3\n"))
    p = tmp_path / "t.stl"
    p.write_bytes(b"\0" * 80 + struct.pack("<I", 1) + struct.pack("<12f", 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 2, 0) + b"\0\0")
    s = load_stl(str(p))
    assert s.tris.shape == (1, 6, 3) and np.array_equal(s.tris[0, 3], [0, 0, 1])
    q = tmp_path / "a.stl"
    q.write_text("solid x\nfacet normal 0 0 0\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid x\n")
    assert load_stl(str(q)).tris.shape == (1, 6, 3)
    # the C++ mirror's loaders (include/rpt.hpp) parse the same files to the same triangles
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mesh_dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "mesh_dump.cpp"), "-o", exe])
    o = tmp_path / "t.obj"
    o.write_text(text)
    for path, kind, expect in ((o, "obj", m.tris), (p, "stl", s.tris), (q, "stl", load_stl(str(q)).tris)):
        out = subprocess.check_output([exe, str(path), kind], text=True).split()
        assert int(out[0]) == expect.shape[0]
        assert np.array_equal(np.array(out[1:], dtype=np.float64).reshape(-1, 6, 3), expect)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    assert subprocess.run([exe, str(bad), "obj"], capture_output=True).returncode == 1


def test_load_obj_with_mtl_splits_objects_by_usemtl():
    import io
    from rpt_amd import load_obj_with_mtl
    obj = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl a\nf 1 3 4\nusemtl b\nf 1 2 4\n"
    objs = load_obj_with_mtl(io.StringIO(obj), io.StringIO("# lib\nnewmtl a\nnewmtl b\n"))
    assert [o.shape.tris.shape[0] for o in objs] == [2, 1]               # a repeated `usemtl a` does not split (io.rs:123)
    with pytest.raises(NotImplementedError):
        load_obj_with_mtl(io.StringIO(obj), io.StringIO("newmtl a\nKd 1 0 0\n"))    # the reference panics here
    with pytest.raises(ValueError):
        load_obj_with_mtl(io.StringIO(obj), io.StringIO("newmtl a\n"))               # usemtl b not in the library


def test_kdtree_group_lowers_to_a_children_array():
    """KdTree<Box<dyn Bounded>> (kdtree.rs:103-146): children keep their own transforms, groups nest,
    planes are rejected (not Bounded), and a transform on the group stays on the group."""
    import ctypes as C
    from rpt_amd import KdTree, cube, plane, sphere, vec3
    from rpt_amd._lib import ShapeDesc
    from rpt_amd.api import shape_desc
    inner = KdTree([sphere().translate(vec3(0, 1, 0)), cube()])
    g = KdTree([sphere().scale(vec3(2, 2, 2)), inner.translate(vec3(5, 0, 0))]).rotate_y(0.5)
    d, keep = shape_desc(g, ShapeDesc)
    assert d.kind == 4 and d.has_transform == 1 and d.n_children == 2 and d.n_tris == 0
    c0, c1 = d.children[0], d.children[1]
    assert c0.kind == 0 and c0.has_transform == 1 and c0.transform[0] == 2.0
    assert c1.kind == 4 and c1.n_children == 2 and c1.transform[3] == 5.0
    assert c1.children[0].kind == 0 and c1.children[0].transform[7] == 1.0 and c1.children[1].kind == 1
    assert C.sizeof(ShapeDesc) == 200
    with pytest.raises(TypeError):
        KdTree([plane(vec3(0, 1, 0), 0.0)])
    with pytest.raises(ValueError):
        KdTree([])
    scene, cam, cfg = scenes.fractal_spheres()
    assert [len(o.shape.shapes) for o in scene.objects[:5]] == [1, 6, 30, 150, 750]     # fractal_spheres.rs prints these
