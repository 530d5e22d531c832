"""Host-side mirror of rpt's builder API (same names, argument meaning and error behaviour),
lowered onto the C ABI of include/rpt_hip.h.  Reference: neevparikh/rpt `src/`:

    Scene / SceneAdd        scene.rs:12-81        Object            object.rs:10-31
    Light                   light.rs:7-19         Material          material.rs:8-97
    Medium                  medium.rs:78-122      Camera            camera.rs:9-62
    sphere/cube/plane/...   shape.rs:288-314      Transformed       shape.rs:102-285
    Renderer                renderer.rs:23-156    Buffer / Filter   buffer.rs:6-108
    hex_color/color_bytes   color.rs:10-24        Environment       environment.rs:56-77

Everything here is scene description and output bookkeeping in numpy fp64; the per-sample work
(`Renderer.sample`, renderer.rs:158-171) runs in the HIP library.  Additions with no
counterpart in the reference: `Renderer.seed(u64)` (the reference seeds from entropy,
renderer.rs:163), `Renderer.shard(rank, count)` (multi-GPU tiles) and `Renderer.device(i)`.
"""
import ctypes as C
import math
import weakref

import numpy as np

from . import _lib
from ._lib import RptError

def shard_pixels(width, height, rank, count):
    """Pixel indices (y*width + x) owned by `rank` of `count` under the renderer's 32x32-tile
    sharding (rpt_shard_tiles), in ascending order."""
    lib = _lib.load()
    n = _lib.check(lib.rpt_shard_tiles(width, height, rank, count, None, 0))
    tiles = np.zeros(max(n, 1), dtype=np.uint32)
    _lib.check(lib.rpt_shard_tiles(width, height, rank, count, tiles.ctypes.data_as(C.c_void_p), n))
    tiles_x = (width + 31) // 32
    ys, xs = np.mgrid[0:32, 0:32]
    out = []
    for t in tiles[:n]:
        x = (int(t) % tiles_x) * 32 + xs
        y = (int(t) // tiles_x) * 32 + ys
        m = (x < width) & (y < height)
        out.append((y[m] * width + x[m]).astype(np.uint32))
    return np.sort(np.concatenate(out)) if out else np.zeros(0, dtype=np.uint32)


_LIVE_SCENES = weakref.WeakSet()   # Scene objects that hold a device handle


def set_option(name, value):
    """Process-wide convenience over the C ABI's per-scene options: sets the default for scenes created from now on
    (rpt_set_option) and the option of every scene this process has on a device (rpt_scene_set_option), so that
    `set_option("counters", 1)` acts on the renderer at hand as it always did.  Options read by rpt_scene_commit
    ("scene_bvh_min", "instancing", "room_shell", "bvh_leaf_max", "bvh_max_depth") only matter before a scene's
    first render; use Scene.set_option to give one scene its own value."""
    lib = _lib.load()
    _lib.check(lib.rpt_set_option(name.encode(), int(value)))
    for sc in list(_LIVE_SCENES):
        if sc._handle is not None:
            _lib.check(lib.rpt_scene_set_option(sc._handle, name.encode(), int(value)))


__all__ = [
    "set_option", "shard_pixels",
    "vec3", "hex_color", "color_bytes", "Sphere", "Cube", "Plane", "Triangle", "Mesh", "KdTree", "Transformed",
    "sphere", "cube", "plane", "polygon", "Material", "Object", "Light", "Medium", "Environment",
    "Scene", "Camera", "Filter", "Buffer", "DeviceBuffer", "Renderer", "RptError",
]


def vec3(x, y, z):
    """glm::vec3"""
    return np.array([x, y, z], dtype=np.float64)


def _v(a):
    a = np.asarray(a, dtype=np.float64)
    if a.shape != (3,):
        raise ValueError("expected a 3-vector")
    return a


# ------------------------------------------------------------------ color.rs
SRGB_GAMMA = 2.2


def hex_color(x):
    """color.rs:10-15: sRGB hex integer -> linear RGB (gamma 2.2)."""
    r = ((x >> 16) & 0xFF) / 255.0
    g = ((x >> 8) & 0xFF) / 255.0
    b = (x & 0xFF) / 255.0
    return vec3(r ** SRGB_GAMMA, g ** SRGB_GAMMA, b ** SRGB_GAMMA)


def color_bytes(color):
    """color.rs:18-24: clamp, gamma 1/2.2, *255, truncating `as u8`.  Accepts (...,3) arrays."""
    c = np.clip(np.asarray(color, dtype=np.float64), 0.0, 1.0) ** (1.0 / SRGB_GAMMA) * 255.0
    c = np.where(np.isnan(c), 0.0, c)
    return c.astype(np.uint8)


# ------------------------------------------------------------------ shapes (shape.rs)
def _translate(v):
    m = np.eye(4)
    m[:3, 3] = _v(v)
    return m


def _scale(v):
    return np.diag(np.append(_v(v), 1.0))


def _rotate(angle, axis):
    # glm::rotate(identity, angle, axis): axis is normalised, right-handed
    a = _v(axis)
    a = a / np.linalg.norm(a)
    c, s = math.cos(angle), math.sin(angle)
    x, y, z = a
    r = np.array([
        [c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
        [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
        [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)],
    ])
    m = np.eye(4)
    m[:3, :3] = r
    return m


class Shape:
    """trait Shape + Transformable (shape.rs:19-26, 179-230)."""
    KIND = -1

    def _wrap(self, m):
        return Transformed(self, m)

    def translate(self, v):
        return self._wrap(_translate(v))

    def scale(self, v):
        return self._wrap(_scale(v))

    def rotate(self, angle, axis):
        return self._wrap(_rotate(angle, axis))

    def rotate_x(self, angle):
        return self._wrap(_rotate(angle, (1, 0, 0)))

    def rotate_y(self, angle):
        return self._wrap(_rotate(angle, (0, 1, 0)))

    def rotate_z(self, angle):
        return self._wrap(_rotate(angle, (0, 0, 1)))

    def transform(self, m):
        return self._wrap(np.asarray(m, dtype=np.float64).reshape(4, 4))

    # -- lowering helpers (shared by the HIP binding and, in tests, the oracle binding)
    def base(self):
        return self

    def matrix(self):
        return None


class Sphere(Shape):
    """Unit sphere at the origin (shape/sphere.rs:10)."""
    KIND = 0


class Cube(Shape):
    """Unit cube centred at the origin (shape/cube.rs:10)."""
    KIND = 1


class Plane(Shape):
    """x . normal = value (shape/plane.rs:7-13)."""
    KIND = 2

    def __init__(self, normal, value):
        self.normal = _v(normal)
        self.value = float(value)


class Triangle:
    """shape/mesh.rs:9-39."""

    def __init__(self, v1, v2, v3, n1, n2, n3):
        self.v1, self.v2, self.v3 = _v(v1), _v(v2), _v(v3)
        self.n1, self.n2, self.n3 = _v(n1), _v(n2), _v(n3)

    @staticmethod
    def from_vertices(v1, v2, v3):
        v1, v2, v3 = _v(v1), _v(v2), _v(v3)
        n = np.cross(v2 - v1, v3 - v1)
        n = n / np.linalg.norm(n)
        return Triangle(v1, v2, v3, n, n, n)


class Mesh(Shape):
    """`Mesh = KdTree<Triangle>` (shape/mesh.rs:103).  Holds an (n, 6, 3) fp64 array:
    v1 v2 v3 n1 n2 n3 per triangle.  The acceleration structure is built in the library."""
    KIND = 3

    def __init__(self, triangles):
        if isinstance(triangles, np.ndarray):
            arr = np.ascontiguousarray(triangles, dtype=np.float64).reshape(-1, 6, 3)
        else:
            arr = np.array([[t.v1, t.v2, t.v3, t.n1, t.n2, t.n3] for t in triangles], dtype=np.float64).reshape(-1, 6, 3)
        self.tris = arr

    def clone(self):
        return Mesh(self.tris.copy())


class KdTree(Shape):
    """`KdTree<Box<dyn Bounded>>` used as one shape (kdtree.rs:103-146; examples/fractal_spheres.rs:45):
    a group of bounded shapes (spheres, cubes, meshes, nested groups, transformed or not) sharing
    one material.  Planes are not `Bounded` and are rejected."""
    KIND = 4

    def __init__(self, shapes):
        self.shapes = list(shapes)
        if not self.shapes:
            raise ValueError("KdTree needs at least one shape")
        for s in self.shapes:
            if isinstance(s.base(), Plane):
                raise TypeError("Plane is not Bounded and cannot be put in a KdTree")

    def clone(self):
        return KdTree(self.shapes)


class Transformed(Shape):
    """shape.rs:102-125.  Chained calls left-multiply and do not nest (shape.rs:232-285)."""

    def __init__(self, shape, m):
        if isinstance(shape, Transformed):
            m = np.asarray(m) @ shape.m
            shape = shape.shape
        self.shape = shape
        self.m = np.asarray(m, dtype=np.float64).reshape(4, 4)

    def _wrap(self, m):
        return Transformed(self.shape, m @ self.m)

    def base(self):
        return self.shape

    def matrix(self):
        return self.m

    def clone(self):
        return Transformed(self.shape, self.m.copy())


def sphere():
    return Sphere()


def cube():
    return Cube()


def plane(normal, value):
    return Plane(normal, value)


def polygon(verts):
    """shape.rs:308-314: fan triangulation from verts[0]."""
    verts = [_v(p) for p in verts]
    return Mesh([Triangle.from_vertices(verts[0], verts[i], verts[i + 1]) for i in range(1, len(verts) - 1)])


# ------------------------------------------------------------------ material.rs
class Material:
    LAMBERTIAN, PHONG, MIRROR, TRANSMISSIVE = 0, 1, 2, 3

    def __init__(self, kind=0, albedo=(0.5, 0.5, 0.5), emittance=0.0, shininess=0.0, ior=1.0):
        self.kind = kind
        self.albedo = _v(albedo)
        self.emittance_ = float(emittance)
        self.shininess = float(shininess)
        self.ior = float(ior)

    # constructors, material.rs:34-97
    @staticmethod
    def diffuse(color):
        return Material(Material.LAMBERTIAN, color)

    @staticmethod
    def specular(color, roughness):
        return Material(Material.PHONG, color, shininess=roughness)  # roughness IS the shininess (sic)

    @staticmethod
    def mirror():
        return Material(Material.MIRROR, (0, 0, 0))

    @staticmethod
    def transmissive(ior):
        return Material(Material.TRANSMISSIVE, (0, 0, 0), ior=ior)

    @staticmethod
    def clear(index, _roughness=0.0):
        return Material(Material.TRANSMISSIVE, (0, 0, 0), ior=index)

    @staticmethod
    def transparent(color, index, _roughness=0.0):
        return Material(Material.TRANSMISSIVE, color, ior=index)

    @staticmethod
    def metallic(color, roughness):
        return Material(Material.PHONG, color, shininess=roughness)

    @staticmethod
    def light(color, emittance):
        return Material(Material.LAMBERTIAN, color, emittance=emittance)

    def emittance(self):  # material.rs:100-106
        return self.emittance_ if self.kind in (0, 1) else 0.0

    def color(self):  # material.rs:107-113
        return self.albedo if self.kind in (0, 1) else vec3(0, 0, 0)


class Object:
    """object.rs:10-31."""

    def __init__(self, shape):
        if not isinstance(shape, Shape):
            raise TypeError("Object::new expects a shape")
        self.shape = shape
        self.material_ = Material()

    def material(self, material):
        self.material_ = material
        return self


class Light:
    """enum Light (light.rs:7-19)."""
    POINT, AMBIENT, DIRECTIONAL, OBJECT = 0, 1, 2, 3

    def __init__(self, kind, color=None, vec=None, obj=None):
        self.kind, self.color, self.vec, self.object = kind, color, vec, obj

    @staticmethod
    def Point(color, location):
        return Light(Light.POINT, _v(color), _v(location))

    @staticmethod
    def Ambient(color):
        return Light(Light.AMBIENT, _v(color))

    @staticmethod
    def Directional(color, direction):
        return Light(Light.DIRECTIONAL, _v(color), _v(direction))

    @staticmethod
    def Object(obj):
        return Light(Light.OBJECT, obj=obj)


class Medium:
    """medium.rs:78-122: the two constructors are the closed set (fields are private)."""
    HOMOGENEOUS_ISOTROPIC, COLORED_GLOWING_FOG = 0, 1

    def __init__(self, kind, absorption, scattering):
        self.kind, self.absorption, self.scattering = kind, float(absorption), float(scattering)

    @staticmethod
    def homogeneous_isotropic(absorption, scattering):
        return Medium(Medium.HOMOGENEOUS_ISOTROPIC, absorption, scattering)

    @staticmethod
    def colored_glowing_fog(absorption, scattering):
        return Medium(Medium.COLORED_GLOWING_FOG, absorption, scattering)


class Environment:
    """environment.rs:3-77: Environment::Color(c) or Environment::Hdri(Hdri::new(width, height, buf))."""

    def __init__(self, color=(0, 0, 0), hdri=None):
        self.color = _v(color)
        self.hdri = hdri          # (height, width, 3) fp64 array or None

    @staticmethod
    def Color(color):
        return Environment(color)

    @staticmethod
    def Hdri(width, height, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float64).reshape(-1, 3)
        assert buf.shape[0] == width * height and width > 0 and height > 0     # Hdri::new, environment.rs:18-22
        return Environment((0, 0, 0), buf.reshape(height, width, 3))


class Scene:
    """scene.rs:12-81."""

    def __init__(self):
        self.objects, self.lights, self.media = [], [], []
        self.environment = Environment((0, 0, 0))
        self._handle = None
        self._options = {}

    @staticmethod
    def new():
        return Scene()

    def add(self, node):
        if self._handle is not None:
            raise RptError("scene is immutable once rendered (committed to the device)")
        if isinstance(node, Object):
            self.objects.append(node)
        elif isinstance(node, Light):
            self.lights.append(node)
        elif isinstance(node, Medium):
            self.media.append(node)
        elif isinstance(node, tuple) and len(node) == 2 and isinstance(node[1], Material):
            # SceneAdd<(Mesh, Material)> / SceneAdd<(Transformed<Cube>, Material)>, scene.rs:57-75:
            # the same geometry becomes an Object AND a Light::Object
            shape, material = node
            ok = isinstance(shape, Mesh) or (isinstance(shape, Transformed) and isinstance(shape.shape, Cube))
            if not ok:
                raise TypeError("SceneAdd is implemented for (Mesh, Material) and (Transformed<Cube>, Material)")
            self.add(Object(shape.clone()).material(material))
            self.add(Light.Object(Object(shape.clone()).material(material)))
        else:
            raise TypeError(f"cannot add {type(node).__name__} to a Scene")

    def set_option(self, name, value):
        """rpt_scene_set_option: this scene's own value of an option (see rpt_hip.h), whatever other scenes use."""
        self._options[name] = int(value)
        if self._handle is not None:
            _lib.check(_lib.load().rpt_scene_set_option(self._handle, name.encode(), int(value)))
        return self

    # ---- lowering onto the C ABI
    def _commit(self, device):
        if self._handle is not None:
            if self._device != device:
                raise RptError("scene already committed to another device")
            return self._handle
        lib = _lib.load()
        h = lib.rpt_scene_create()
        try:
            for o in self.objects:
                sd, keep = shape_desc(o.shape, _lib.ShapeDesc)
                _lib.check(lib.rpt_scene_add_object(h, C.byref(sd), C.byref(material_desc(o.material_, _lib.MaterialDesc))))
            for l in self.lights:
                if l.kind == Light.POINT:
                    _lib.check(lib.rpt_scene_add_light_point(h, _dp(l.color), _dp(l.vec)))
                elif l.kind == Light.AMBIENT:
                    _lib.check(lib.rpt_scene_add_light_ambient(h, _dp(l.color)))
                elif l.kind == Light.DIRECTIONAL:
                    _lib.check(lib.rpt_scene_add_light_directional(h, _dp(l.color), _dp(l.vec)))
                else:
                    sd, keep = shape_desc(l.object.shape, _lib.ShapeDesc)
                    _lib.check(lib.rpt_scene_add_light_object(
                        h, C.byref(sd), C.byref(material_desc(l.object.material_, _lib.MaterialDesc))))
            for m in self.media:
                _lib.check(lib.rpt_scene_add_medium(h, m.kind, m.absorption, m.scattering))
            env = self.environment
            if env.hdri is not None:
                _lib.check(lib.rpt_scene_set_environment_hdri(h, env.hdri.shape[1], env.hdri.shape[0], _dp(env.hdri)))
            else:
                _lib.check(lib.rpt_scene_set_environment_color(h, _dp(env.color)))
            for name, value in self._options.items():
                _lib.check(lib.rpt_scene_set_option(h, name.encode(), value))
            _lib.check(lib.rpt_scene_commit(h, device))
        except Exception:
            lib.rpt_scene_destroy(h)
            raise
        self._handle, self._device = h, device
        _LIVE_SCENES.add(self)
        return h

    def close(self):
        if self._handle is not None:
            _lib.load().rpt_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dp(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.ctypes.data_as(C.POINTER(C.c_double))


def shape_desc(shape, cls):
    """Fill a rpt_shape_desc-shaped ctypes struct `cls` from a Shape.  Returns (desc, keepalive)."""
    base, m = shape.base(), shape.matrix()
    d = cls()
    d.kind = base.KIND
    d.has_transform = 0 if m is None else 1
    flat = (np.eye(4) if m is None else m).reshape(-1)
    for i in range(16):
        d.transform[i] = float(flat[i])
    keep = None
    if isinstance(base, Plane):
        for i in range(3):
            d.plane_normal[i] = float(base.normal[i])
        d.plane_value = base.value
    elif isinstance(base, Mesh):
        keep = np.ascontiguousarray(base.tris, dtype=np.float64)
        d.tris = keep.ctypes.data_as(C.POINTER(C.c_double))
        d.n_tris = keep.shape[0]
    elif isinstance(base, KdTree):
        arr = (cls * len(base.shapes))()
        keep = [arr]
        for i, child in enumerate(base.shapes):
            cd, ck = shape_desc(child, cls)
            C.memmove(C.byref(arr, i * C.sizeof(cls)), C.byref(cd), C.sizeof(cls))
            keep.append(ck)
        d.children = C.cast(arr, C.POINTER(cls))
        d.n_children = len(base.shapes)
    elif not isinstance(base, (Sphere, Cube)):
        raise TypeError(f"unsupported shape {type(base).__name__}")
    d._keep = keep
    return d, keep


def material_desc(mat, cls):
    d = cls()
    d.kind = mat.kind
    for i in range(3):
        d.albedo[i] = float(mat.albedo[i])
    d.emittance = mat.emittance_
    d.shininess = mat.shininess
    d.ior = mat.ior
    return d


# ------------------------------------------------------------------ camera.rs
class Camera:
    def __init__(self, eye=(0.0, 0.0, 10.0), direction=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0),
                 fov=math.pi / 6, aperture=0.0, focal_distance=0.0):
        self.eye, self.direction, self.up = _v(eye), _v(direction), _v(up)
        self.fov, self.aperture, self.focal_distance = float(fov), float(aperture), float(focal_distance)

    @staticmethod
    def look_at(eye, center, up, fov):  # camera.rs:44-56
        eye, center, up = _v(eye), _v(center), _v(up)
        direction = center - eye
        direction = direction / np.linalg.norm(direction)
        up = up - np.dot(up, direction) * direction
        up = up / np.linalg.norm(up)
        return Camera(eye, direction, up, fov)

    def focus(self, focal_point, aperture):  # camera.rs:58-62
        self.focal_distance = float(np.dot(_v(focal_point) - self.eye, self.direction))
        self.aperture = float(aperture)
        return self


def camera_desc(cam, cls):
    d = cls()
    for i in range(3):
        d.eye[i], d.direction[i], d.up[i] = float(cam.eye[i]), float(cam.direction[i]), float(cam.up[i])
    d.fov, d.aperture, d.focal_distance = cam.fov, cam.aperture, cam.focal_distance
    return d


# ------------------------------------------------------------------ buffer.rs
class Filter:
    def __init__(self, radius=0):
        self.radius = int(radius)

    @staticmethod
    def Box(radius):
        return Filter(radius)

    @staticmethod
    def default():
        return Filter(0)


class Buffer:
    """buffer.rs:6-93: one mean colour per pixel per `sample()` call."""

    def __init__(self, width, height, filter=None):
        self.width, self.height = int(width), int(height)
        self.samples = []  # list of (h*w, 3) arrays, one per add_samples call
        self.filter = filter or Filter.default()

    def add_samples(self, samples):
        samples = np.asarray(samples, dtype=np.float64).reshape(-1, 3)
        assert samples.shape[0] == self.width * self.height, "Invalid sample dimension"
        self.samples.append(samples)

    def _filtered(self):
        assert self.samples, "Pixel found with no samples"
        total = np.sum(self.samples, axis=0).reshape(self.height, self.width, 3)
        count = float(len(self.samples))
        r = self.filter.radius
        if r == 0:
            return total / count
        # box window clipped to the image: sum of sums / sum of counts (buffer.rs:75-93)
        pad = np.zeros((self.height + 2 * r, self.width + 2 * r, 3))
        pad[r:r + self.height, r:r + self.width] = total
        cnt = np.zeros((self.height + 2 * r, self.width + 2 * r))
        cnt[r:r + self.height, r:r + self.width] = count
        acc = np.zeros_like(total)
        cacc = np.zeros((self.height, self.width))
        for dy in range(2 * r + 1):
            for dx in range(2 * r + 1):
                acc += pad[dy:dy + self.height, dx:dx + self.width]
                cacc += cnt[dy:dy + self.height, dx:dx + self.width]
        return acc / cacc[..., None]

    def image(self):
        """(h, w, 3) uint8, the analogue of image::RgbImage (buffer.rs:43-56)."""
        return color_bytes(self._filtered())

    def variance(self):
        """buffer.rs:59-73: mean per-pixel sample variance across batches (n-1 dof)."""
        s = np.stack(self.samples)  # (n, hw, 3)
        n = s.shape[0]
        mean = s.mean(axis=0)
        ss = ((s - mean) ** 2).sum(axis=2).sum(axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(np.mean(ss / (n - 1.0)))


class DeviceBuffer:
    """The same Buffer kept on the GPU (rpt_buffer_*): per-pixel running sums, box filter +
    color_bytes and variance computed where the frame is.  What Renderer.render() and
    iterative_render() use; `Buffer` above is the host restatement."""

    def __init__(self, width, height, filter=None, device=0):
        self.width, self.height = int(width), int(height)
        self.filter = filter or Filter.default()
        self.device = int(device)
        self._h = _lib.load().rpt_buffer_create(self.device, self.width, self.height, int(self.filter.radius))
        if not self._h:
            _lib.check(-1)

    def add_samples(self, samples):
        samples = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1, 3)
        assert samples.shape[0] == self.width * self.height, "Invalid sample dimension"
        _lib.check(_lib.load().rpt_buffer_add_samples(self._h, samples.ctypes.data_as(C.c_void_p)))

    def add_samples_device(self, d_rgb_ptr, stream_ptr=0):
        _lib.check(_lib.load().rpt_buffer_add_samples_device(self._h, C.c_void_p(d_rgb_ptr), C.c_void_p(stream_ptr)))

    @property
    def batches(self):
        n = C.c_uint32()
        _lib.check(_lib.load().rpt_buffer_batches(self._h, C.byref(n)))
        return int(n.value)

    def image(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        _lib.check(_lib.load().rpt_buffer_image(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def variance(self):
        v = C.c_double()
        _lib.check(_lib.load().rpt_buffer_variance(self._h, C.byref(v)))
        return float(v.value)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().rpt_buffer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ renderer.rs
class Renderer:
    def __init__(self, scene, camera):
        self.scene, self.camera = scene, camera
        self.width_, self.height_ = 800, 600
        self.exposure_value_ = 0.0
        self.filter_ = Filter.default()
        self.stepsize_ = 0.0
        self.max_bounces_ = 0
        self.num_samples_ = 1
        self.gather_size_, self.gather_size_volume_, self.watts_ = 50, 50, 100.0
        self.seed_ = 0
        self.shard_rank_, self.shard_count_ = 0, 1
        self.device_ = 0
        self._sample_offset = 0

    @staticmethod
    def new(scene, camera):
        return Renderer(scene, camera)

    def width(self, v):
        self.width_ = int(v); return self

    def height(self, v):
        self.height_ = int(v); return self

    def exposure_value(self, v):
        self.exposure_value_ = float(v); return self

    def stepsize(self, v):
        self.stepsize_ = float(v); return self  # stored, never read (renderer.rs:43, 96-99)

    def filter(self, f):
        self.filter_ = f; return self

    def max_bounces(self, v):
        self.max_bounces_ = int(v); return self

    def num_samples(self, v):
        self.num_samples_ = int(v); return self

    def gather_size(self, v):
        self.gather_size_ = int(v); return self

    def gather_size_volume(self, v):
        self.gather_size_volume_ = int(v); return self

    def watts(self, v):
        self.watts_ = float(v); return self

    # additions (documented deviations)
    def seed(self, v):
        self.seed_ = int(v) & 0xFFFFFFFFFFFFFFFF; return self

    def shard(self, rank, count):
        self.shard_rank_, self.shard_count_ = int(rank), int(count); return self

    def device(self, index):
        self.device_ = int(index); return self

    def _params(self):
        p = _lib.RenderParams()
        p.width, p.height = self.width_, self.height_
        p.exposure_value = self.exposure_value_
        p.max_bounces = self.max_bounces_
        p.shard_rank, p.shard_count = self.shard_rank_, self.shard_count_
        return p

    def sample_array(self, iterations):
        """Renderer::sample (renderer.rs:158-171) -> (h*w, 3) fp64 array of per-pixel means."""
        lib = _lib.load()
        h = self.scene._commit(self.device_)
        out = np.empty((self.width_ * self.height_, 3), dtype=np.float64)
        _lib.check(lib.rpt_render_sample(
            h, C.byref(camera_desc(self.camera, _lib.CameraDesc)), C.byref(self._params()), int(iterations),
            C.c_uint64(self.seed_), self._sample_offset, out.ctypes.data_as(C.c_void_p)))
        self._sample_offset += int(iterations)
        return out

    def sample_device(self, iterations, d_out_ptr, stream_ptr=0):
        """Asynchronous variant: d_out_ptr is a device pointer to width*height*3 doubles."""
        lib = _lib.load()
        h = self.scene._commit(self.device_)
        _lib.check(lib.rpt_render_sample_device(
            h, C.byref(camera_desc(self.camera, _lib.CameraDesc)), C.byref(self._params()), int(iterations),
            C.c_uint64(self.seed_), self._sample_offset, C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr)))
        self._sample_offset += int(iterations)

    def sample(self, iterations, buffer):
        """Renderer::sample (renderer.rs:158-171).  A DeviceBuffer receives the batch on the GPU."""
        if isinstance(buffer, DeviceBuffer) and self.shard_count_ == 1:
            _lib.check(_lib.load().rpt_render_into_buffer(
                self.scene._commit(self.device_), C.byref(camera_desc(self.camera, _lib.CameraDesc)),
                C.byref(self._params()), int(iterations), C.c_uint64(self.seed_), self._sample_offset, buffer._h))
            self._sample_offset += int(iterations)
        else:
            buffer.add_samples(self.sample_array(iterations))

    def render(self):
        """renderer.rs:137-141 -> (h, w, 3) uint8 image."""
        buffer = DeviceBuffer(self.width_, self.height_, self.filter_, self.device_)
        self._sample_offset = 0
        self.sample(self.num_samples_, buffer)
        return buffer.image()

    def iterative_render(self, callback_interval, callback):
        """renderer.rs:144-156."""
        buffer = DeviceBuffer(self.width_, self.height_, self.filter_, self.device_)
        self._sample_offset = 0
        iteration = 0
        while iteration < self.num_samples_:
            steps = min(self.num_samples_ - iteration, callback_interval)
            self.sample(steps, buffer)
            iteration += steps
            callback(iteration, buffer)

    def timing(self):
        """(render_ms, resolve_ms, grid_blocks) of the last call, from HIP events on its stream."""
        a, b, g = C.c_double(), C.c_double(), C.c_int32()
        _lib.check(_lib.load().rpt_get_timing(self.scene._handle, C.byref(a), C.byref(b), C.byref(g)))
        return a.value, b.value, g.value

    def timing_mean(self):
        """(mean render_ms, mean resolve_ms, launches) over the timed calls since the previous timing_mean();
        the calls themselves never wait for their events."""
        a, b, n = C.c_double(), C.c_double(), C.c_int32()
        _lib.check(_lib.load().rpt_get_timing_mean(self.scene._handle, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def chunking(self, iterations):
        """(samples per work item, work items per pixel) of this renderer's sample(iterations) calls: the scene's own
        "chunk_spp" option applies (rpt_scene_render_chunking), as in its renders."""
        c, n = C.c_uint32(), C.c_uint32()
        _lib.check(_lib.load().rpt_scene_render_chunking(self.scene._commit(self.device_), int(iterations), C.byref(c), C.byref(n)))
        return int(c.value), int(n.value)

    def scene_stats(self):
        """rpt_scene_stats of the committed scene (flattened-layout record counts and bytes)."""
        out = (C.c_uint64 * 16)()
        _lib.check(_lib.load().rpt_scene_stats(self.scene._commit(self.device_), out))
        names = ["spheres", "cubes", "planes", "tris", "aabbs", "rects", "bvh_tris", "bvh_nodes", "scan_bytes_per_ray",
                 "scene_bytes", "scene_bvh", "scene_bvh_prims", "instances", "shared_meshes", "shell_faces", "tree_depth"]
        return dict(zip(names, [int(v) for v in out]))

    # ---- photon mapping (src/photon.rs:631-720)
    PHOTON_MAP, PHOTON_POINT_BEAM, PHOTON_BEAM_BEAM = 0, 1, 2   # enum PhotonRenderKind

    def photon_map_build(self, photon_count, kind=1):
        """Shooting + map build of Renderer::photon_render (photon.rs:655-704) on the device."""
        lib = _lib.load()
        h = self.scene._commit(self.device_)
        _lib.check(lib.rpt_photon_map_build(h, int(photon_count), int(kind), self.watts_, C.c_uint64(self.seed_)))
        out = (C.c_uint64 * 8)()
        _lib.check(lib.rpt_photon_map_stats(h, out))
        return {"surface": int(out[0]), "volume": int(out[1]), "shot": int(out[2]), "shoot_us": int(out[3]),
                "build_us": int(out[4])}

    def _photon_stats(self):
        out = (C.c_uint64 * 8)()
        _lib.check(_lib.load().rpt_photon_map_stats(self.scene._handle, out))
        return {"surface": int(out[0]), "volume": int(out[1]), "shot": int(out[2]), "shoot_us": int(out[3]),
                "build_us": int(out[4])}

    def photon_shoot(self, photon_count, kind, shard_rank=0, shard_count=1):
        """rpt_photon_shoot: this rank's contiguous block of the shooting loop (photon.rs:656-690);
        returns (surface, volume) record counts.  The records stay on the device (photon_records)."""
        n = (C.c_uint64 * 2)()
        h = self.scene._commit(self.device_)
        _lib.check(_lib.load().rpt_photon_shoot(h, int(photon_count), int(kind), self.watts_, C.c_uint64(self.seed_),
                                                int(shard_rank), int(shard_count), n))
        return int(n[0]), int(n[1])

    def photon_records(self, which):
        """(device pointer, count) of the records of the last photon_shoot; 48 bytes each."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _lib.check(_lib.load().rpt_photon_records(self.scene._handle, int(which), C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def photon_map_from_records(self, photon_count, kind, d_surface, n_surface, d_volume, n_volume):
        """rpt_photon_map_from_records: build the maps from (gathered) device record arrays."""
        h = self.scene._commit(self.device_)
        _lib.check(_lib.load().rpt_photon_map_from_records(h, int(photon_count), int(kind), C.c_void_p(d_surface),
                                                           int(n_surface), C.c_void_p(d_volume), int(n_volume)))
        return self._photon_stats()

    def photon_map_download(self, which):
        """Test hook: (n, 10) float32 photons in shooting order (position, direction, power, radius)."""
        lib = _lib.load()
        h = self.scene._handle
        out = (C.c_uint64 * 8)()
        _lib.check(lib.rpt_photon_map_stats(h, out))
        n = int(out[which])
        arr = np.zeros((n, 10), dtype=np.float32)
        _lib.check(lib.rpt_photon_map_download(h, which, arr.ctypes.data_as(C.c_void_p), n))
        return arr

    def photon_positions64(self):
        """Reference-epsilon mode: the surface photons' fp64 positions, (n, 3), in the order of photon_map_download(0)."""
        n = self._photon_stats()["surface"]
        out = np.zeros((n, 3), dtype=np.float64)
        _lib.check(_lib.load().rpt_debug_photon_positions64(self.scene._handle, out.ctypes.data_as(C.c_void_p), n))
        return out

    def photon_selections(self):
        """Reference-epsilon mode, test hook: the last camera pass's per-sample selections, (pixel slots, gather_size + 2, samples) u32."""
        lib = _lib.load()
        dims = (C.c_uint64 * 3)()
        _lib.check(lib.rpt_debug_photon_selections(self.scene._handle, None, 0, dims))
        out = np.zeros((int(dims[0]), int(dims[1]), int(dims[2])), dtype=np.uint32)
        _lib.check(lib.rpt_debug_photon_selections(self.scene._handle, out.ctypes.data_as(C.c_void_p), out.size, dims))
        return out

    def photon_sample_array(self, num_samples):
        """get_color_with_photon_map over the frame (photon.rs:706-716): (h*w, 3) fp64 means."""
        lib = _lib.load()
        h = self.scene._commit(self.device_)
        out = np.empty((self.width_ * self.height_, 3), dtype=np.float64)
        _lib.check(lib.rpt_photon_render_sample(
            h, C.byref(camera_desc(self.camera, _lib.CameraDesc)), C.byref(self._params()), self.gather_size_,
            self.gather_size_volume_, int(num_samples), C.c_uint64(self.seed_), self._sample_offset,
            out.ctypes.data_as(C.c_void_p)))
        self._sample_offset += int(num_samples)
        return out

    def photon_sample_device(self, num_samples, d_out_ptr, stream_ptr=0):
        lib = _lib.load()
        h = self.scene._commit(self.device_)
        _lib.check(lib.rpt_photon_render_sample_device(
            h, C.byref(camera_desc(self.camera, _lib.CameraDesc)), C.byref(self._params()), self.gather_size_,
            self.gather_size_volume_, int(num_samples), C.c_uint64(self.seed_), self._sample_offset,
            C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr)))
        self._sample_offset += int(num_samples)

    def photon_render(self, photon_count, kind):
        """Renderer::photon_render (photon.rs:655-720) -> (h, w, 3) uint8 image."""
        self.photon_map_build(photon_count, kind)
        buffer = Buffer(self.width_, self.height_, self.filter_)
        self._sample_offset = 0
        buffer.add_samples(self.photon_sample_array(self.num_samples_))
        return buffer.image()

    def photon_point_query_beam_render(self, photon_count):   # photon.rs:642-644
        return self.photon_render(photon_count, Renderer.PHOTON_POINT_BEAM)

    def photon_beam_query_beam_render(self, photon_count):    # photon.rs:646-648
        return self.photon_render(photon_count, Renderer.PHOTON_BEAM_BEAM)

    def photon_map_render(self, photon_count):                # photon.rs:650-652
        return self.photon_render(photon_count, Renderer.PHOTON_MAP)

    def counters(self):
        out = (C.c_uint64 * 8)()
        _lib.check(_lib.load().rpt_get_counters(self.scene._handle, out))
        names = ["samples", "rays", "vertices", "wave_trips", "prim_tests", "bvh_nodes", "bvh_tris", "stack_overflows"]
        return dict(zip(names, [int(v) for v in out]))

    def get_closest_hit(self, origins, dirs):
        """Renderer::get_closest_hit (renderer.rs:416-425), batched: -> (t, object index, normal)."""
        lib = _lib.load()
        h = self.scene._commit(self.device_)
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float32)
        obj = np.empty(n, dtype=np.int32)
        nrm = np.empty((n, 3), dtype=np.float32)
        _lib.check(lib.rpt_intersect_batch(h, n, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                           t.ctypes.data_as(C.c_void_p), obj.ctypes.data_as(C.c_void_p),
                                           nrm.ctypes.data_as(C.c_void_p)))
        return t, obj, nrm
