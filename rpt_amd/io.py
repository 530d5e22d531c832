"""Host-only mesh loaders mirroring rpt's `load_obj` / `load_stl` (src/io.rs:28-74, 152-201,
264-364): they only produce the `Mesh` (triangle array) the hot path consumes.

OBJ: `v`, `vn`, `f` with `a`, `a/b`, `a//c`, `a/b/c` indices (1-based, negative = relative to the
end), polygons fan-triangulated from their first vertex, face normals when any corner lacks a `vn`.
`vt`, `mtllib`, `usemtl` and unknown commands are skipped, as in the reference.  `load_obj_with_mtl`
splits the faces into one Object per `usemtl` run; like the reference's `load_mtl` it accepts only bare
`newmtl` entries (default material) and refuses property lines."""
import struct

import numpy as np

from .api import Mesh, Triangle


def _parse_index(value, length):  # io.rs:12-20
    try:
        index = int(value)
    except ValueError:
        return None
    return index - 1 if index > 0 else length + index


def _parse_face(tokens, vertices, normals):  # parse_obj_face, io.rs:164-201
    vi, vni, out = [], [], []
    for vertex in tokens[1:]:
        args = (vertex.split("/") + ["", "", ""])[:3]
        idx = _parse_index(args[0], len(vertices))
        if idx is None or not (0 <= idx < len(vertices)):
            raise ValueError("Invalid vertex index")
        vi.append(idx)
        vni.append(_parse_index(args[2], len(normals)))
    for i in range(1, len(vi) - 1):
        a, b, c = 0, i, i + 1
        v1, v2, v3 = vertices[vi[a]], vertices[vi[b]], vertices[vi[c]]
        if vni[a] is None or vni[b] is None or vni[c] is None:
            out.append(Triangle.from_vertices(v1, v2, v3))
        else:
            out.append(Triangle(v1, v2, v3, normals[vni[a]], normals[vni[b]], normals[vni[c]]))
    return out


def load_obj(file):
    """load_obj (io.rs:28-74).  `file` is a path or an open text file."""
    close = False
    if isinstance(file, (str, bytes)):
        file = open(file, "r")
        close = True
    vertices, normals, tris = [], [], []
    try:
        for raw in file:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            tokens = line.split()
            if tokens[0] == "v":
                vertices.append([float(tokens[1]), float(tokens[2]), float(tokens[3])])
            elif tokens[0] == "vn":
                normals.append([float(tokens[1]), float(tokens[2]), float(tokens[3])])
            elif tokens[0] == "f":
                tris.extend(_parse_face(tokens, vertices, normals))
    finally:
        if close:
            file.close()
    return Mesh(tris)


def _load_mtl(file):
    """load_mtl (io.rs:203-262): `newmtl name` lines create default materials; any property line panics
    in the reference ("MTL loading not implemented"), mirrored as NotImplementedError."""
    from .api import Material
    materials, current = {}, None
    for raw in file:
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        tokens = line.split()
        if tokens[0] == "newmtl":
            current = tokens[1]
            materials.setdefault(current, Material())
        else:
            if current is None:
                raise ValueError("Material was not specified with `newmtl` before properties were added")
            raise NotImplementedError("MTL loading not implemented")
    return materials


def load_obj_with_mtl(obj_file, mtl_file):
    """load_obj_with_mtl (io.rs:84-150): one Object per run of faces between `usemtl` changes."""
    from .api import Material, Object
    opened = []

    def _open(f):
        if isinstance(f, (str, bytes)):
            f = open(f, "r")
            opened.append(f)
        return f

    try:
        materials = _load_mtl(_open(mtl_file))
        vertices, normals, objects, tris = [], [], [], []
        current_material, last_usemtl = Material(), None

        def flush():
            if tris:
                objects.append(Object(Mesh(list(tris))).material(current_material))
                tris.clear()

        for raw in _open(obj_file):
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            tokens = line.split()
            if tokens[0] in ("v", "vn"):
                (vertices if tokens[0] == "v" else normals).append([float(tokens[1]), float(tokens[2]), float(tokens[3])])
            elif tokens[0] == "f":
                tris.extend(_parse_face(tokens, vertices, normals))
            elif tokens[0] == "usemtl":
                if last_usemtl is None or last_usemtl != tokens[1]:
                    flush()
                    if tokens[1] not in materials:
                        raise ValueError(f"Could not found `usemtl {tokens[1]}` in library")
                    current_material = materials[tokens[1]]
                    last_usemtl = tokens[1]
        flush()
        return objects
    finally:
        for f in opened:
            f.close()


def load_stl(file):
    """load_stl (io.rs:264-364): ASCII or binary STL, face normals recomputed from the vertices."""
    data = open(file, "rb").read() if isinstance(file, (str, bytes)) else file.read()
    tris = []
    head = data[:512].lstrip()
    if head.startswith(b"solid") and b"facet" in data[:2048]:
        pts = []
        for line in data.decode("ascii", "replace").splitlines():
            t = line.split()
            if len(t) == 4 and t[0] == "vertex":
                pts.append([float(t[1]), float(t[2]), float(t[3])])
                if len(pts) == 3:
                    tris.append(Triangle.from_vertices(*pts))
                    pts = []
    else:
        (n,) = struct.unpack_from("<I", data, 80)
        off = 84
        for _ in range(n):
            vals = struct.unpack_from("<12f", data, off)
            tris.append(Triangle.from_vertices(vals[3:6], vals[6:9], vals[9:12]))
            off += 50
    return Mesh(tris)
