"""The C-ABI library: loads, exports every symbol include/rpt_hip.h declares, validates its
arguments without a GPU, and fails loudly (never falls back) when no gfx950 device exists."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from rpt_amd import Material, Object, Renderer, RptError, Scene, cube, plane, polygon, scenes, shard_pixels, sphere, vec3
from rpt_amd import _lib
from rpt_amd.api import material_desc, shape_desc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rpt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rpt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert set(declared) == bound                         # the Python binding covers the whole header
    for name in declared:
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (rpt_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md section 1 shows the Rust `extern "C"` block a maintainer would add: one declaration per function of the
    header, no more, no fewer."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = set(re.findall(r"pub fn (rpt_[a-z0-9_]+)", text))
    assert rust == set(header_functions())


def test_library_is_built_for_gfx950_only():
    data = open(_lib.LIB_PATH, "rb").read()
    # code objects in the fat binary are tagged hipv4-amdgcn-amd-amdhsa--<arch> (bare arch names also
    # occur in rocPRIM's host-side dispatch tables and mean nothing)
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))
    assert targets == {b"gfx950"}


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    h = lib.rpt_scene_create()
    try:
        sd, keep = shape_desc(sphere(), _lib.ShapeDesc)
        md = material_desc(Material.diffuse(vec3(1, 1, 1)), _lib.MaterialDesc)
        assert lib.rpt_scene_add_object(h, C.byref(sd), C.byref(md)) == 0          # returns the object index
        assert lib.rpt_scene_add_object(h, C.byref(sd), C.byref(md)) == 1
        bad = _lib.ShapeDesc()
        bad.kind = 9
        assert lib.rpt_scene_add_object(h, C.byref(bad), C.byref(md)) == -1
        assert b"shape kind" in lib.rpt_last_error()
        sing, _ = shape_desc(cube().scale(vec3(1, 0, 1)), _lib.ShapeDesc)             # singular transform
        assert lib.rpt_scene_add_object(h, C.byref(sing), C.byref(md)) == -1
        empty = _lib.ShapeDesc()
        empty.kind = 3
        assert lib.rpt_scene_add_object(h, C.byref(empty), C.byref(md)) == -1         # mesh without triangles
        pl, _ = shape_desc(plane(vec3(0, 1, 0), 0.0), _lib.ShapeDesc)
        assert lib.rpt_scene_add_light_object(h, C.byref(pl), C.byref(md)) == -1      # Plane::sample is unimplemented!()
        assert b"plane" in lib.rpt_last_error().lower()
        assert lib.rpt_scene_add_medium(h, 7, 1.0, 1.0) == -1
        assert lib.rpt_scene_add_medium(h, 0, 0.0, 0.0) == -1
        out = np.zeros(12)
        prm = _lib.RenderParams(2, 2, 0.0, 1, 0, 1)
        cam = _lib.CameraDesc()
        rc = lib.rpt_render_sample(h, C.byref(cam), C.byref(prm), 1, 0, 0, out.ctypes.data_as(C.c_void_p))
        assert rc == -2 and b"commit" in lib.rpt_last_error()                         # render before commit
        assert lib.rpt_set_option(b"no_such_option", 1) == -1
        # options are per scene: same names, same validation, no effect on the process defaults
        assert lib.rpt_scene_set_option(h, b"chunk_spp", 8) == 0 and lib.rpt_scene_set_option(h, b"defer_lanes", 16) == 0
        assert lib.rpt_scene_set_option(h, b"defer_lanes", 65) == -1 and lib.rpt_scene_set_option(h, b"nope", 1) == -1
        assert lib.rpt_scene_set_option(None, b"chunk_spp", 8) == -1
        c, n = C.c_uint32(), C.c_uint32()
        assert lib.rpt_render_chunking(256, C.byref(c), C.byref(n)) == 0 and (c.value, n.value) == (16, 16)   # the default rule
        assert lib.rpt_scene_render_chunking(h, 256, C.byref(c), C.byref(n)) == 0 and (c.value, n.value) == (8, 32)   # this scene's option
    finally:
        lib.rpt_scene_destroy(h)


def test_shard_tiles_partition_the_frame():
    for (w, h, n) in [(1024, 1024, 8), (100, 70, 3), (33, 65, 4), (64, 64, 1)]:
        parts = [shard_pixels(w, h, r, n) for r in range(n)]
        allpix = np.concatenate(parts)
        assert len(allpix) == w * h and len(np.unique(allpix)) == w * h
        if w % 256 == 0 and h % 256 == 0:
            assert len({len(p) for p in parts}) == 1                                   # perfectly balanced
    lib = _lib.load()
    assert lib.rpt_shard_tiles(64, 64, 3, 2, None, 0) == -1


def test_frame_pack_layout_is_the_concatenation_of_the_shards():
    """rpt_frame_pack_layout (pure host): rank r's block of the gathered buffer holds exactly the tiles rpt_shard_tiles gives
    rank r, the blocks follow each other in rank order and together they are every tile of the frame."""
    from rpt_amd.dist import frame_pack_layout
    lib = _lib.load()
    for (w, h, n) in [(1024, 1024, 8), (2048, 2048, 8), (100, 70, 3), (33, 65, 4), (64, 64, 1), (32, 32, 5)]:
        offs = frame_pack_layout(w, h, n)
        counts = [lib.rpt_shard_tiles(w, h, r, n, None, 0) for r in range(n)]
        assert offs[0] == 0 and [offs[r + 1] - offs[r] for r in range(n)] == counts
        assert offs[n] == ((w + 31) // 32) * ((h + 31) // 32)
    out = (C.c_uint64 * 2)()
    assert lib.rpt_frame_pack_layout(0, 8, 1, out) == -1 and lib.rpt_frame_pack_layout(8, 8, 0, out) == -1
    assert lib.rpt_gather_frame_device(None, 8, 8, None, None, 0, None) == -1
    assert lib.rpt_comm_create(None, 0, 1, 0, None) == -1


@pytest.mark.skipif(_lib.load().rpt_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_a_loud_error_not_a_fallback():
    scene, cam, cfg = scenes.cornell()
    with pytest.raises(RptError):
        Renderer(scene, cam).width(8).height(8).sample_array(1)
