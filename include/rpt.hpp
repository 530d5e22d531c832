// rpt.hpp — header-only C++17 mirror of rpt's builder API over the C ABI of rpt_hip.h.
//
// rpt is a Rust crate; no Rust toolchain exists in this environment, so the host layer above the
// C ABI is C++ with the reference's names, argument meaning and builder style (methods take the
// object by value and return it, as `fn width(mut self, ..) -> Self` does).  Reference lines:
//   Scene/SceneAdd scene.rs:12-81   Object object.rs:10-31   Light light.rs:7-19
//   Material material.rs:8-97       Medium medium.rs:78-122  Camera camera.rs:9-62
//   shapes shape.rs:102-314         Renderer renderer.rs:23-156   Buffer/Filter buffer.rs:6-108
//   hex_color/color_bytes color.rs:10-24
// Errors: where the reference panics (assert!/expect/unimplemented!), this layer throws rpt::Error.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <istream>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rpt_hip.h"

namespace rpt {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(int rc) {
    if (rc < 0) throw Error(std::string("rpt error ") + std::to_string(rc) + ": " + rpt_last_error());
}

using Vec3 = std::array<double, 3>;
using Color = Vec3;
inline Vec3 vec3(double x, double y, double z) { return {x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a[0], s * a[1], s * a[2]}; }
inline double dot(Vec3 a, Vec3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline Vec3 normalize(Vec3 a) {
    double l = std::sqrt(dot(a, a));
    return {a[0] / l, a[1] / l, a[2] / l};
}

// ---- color.rs
inline Color hex_color(uint32_t x) {
    auto ch = [](uint32_t c) { return std::pow(double(c) / 255.0, 2.2); };
    return {ch((x >> 16) & 0xff), ch((x >> 8) & 0xff), ch(x & 0xff)};
}
inline std::array<uint8_t, 3> color_bytes(const Color& c) {
    std::array<uint8_t, 3> o{};
    for (int i = 0; i < 3; i++) {
        double t = std::pow(std::fmin(std::fmax(c[i], 0.0), 1.0), 1.0 / 2.2) * 255.0;
        o[i] = (t != t || t <= 0.0) ? 0 : (t >= 255.0 ? 255 : uint8_t(t));  // `as u8`: truncating, saturating
    }
    return o;
}

// ---- 4x4 row-major matrices (glm::translate / scale / rotate of the identity)
struct Mat4 {
    std::array<double, 16> m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    Mat4 operator*(const Mat4& o) const {
        Mat4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += m[i * 4 + k] * o.m[k * 4 + j];
                r.m[i * 4 + j] = s;
            }
        return r;
    }
    static Mat4 translation(Vec3 v) {
        Mat4 r;
        r.m[3] = v[0]; r.m[7] = v[1]; r.m[11] = v[2];
        return r;
    }
    static Mat4 scaling(Vec3 v) {
        Mat4 r;
        r.m[0] = v[0]; r.m[5] = v[1]; r.m[10] = v[2];
        return r;
    }
    static Mat4 rotation(double angle, Vec3 axis) {
        Vec3 a = normalize(axis);
        double c = std::cos(angle), s = std::sin(angle), x = a[0], y = a[1], z = a[2];
        Mat4 r;
        r.m = {c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0,
               y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0,
               z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0,
               0, 0, 0, 1};
        return r;
    }
};

// ---- shapes (shape.rs, shape/*.rs).  One value type covers Sphere / Cube / Plane / Mesh and
//      `Transformed<T>`; chained transforms left-multiply and never nest (shape.rs:232-285).
struct Triangle {
    Vec3 v1, v2, v3, n1, n2, n3;
    static Triangle from_vertices(Vec3 a, Vec3 b, Vec3 c) {
        Vec3 n = normalize(cross(b - a, c - a));
        return {a, b, c, n, n, n};
    }
};
class Shape {
  public:
    int kind = RPT_SHAPE_SPHERE;
    bool has_transform = false;
    Mat4 matrix;
    Vec3 plane_normal{0, 0, 0};
    double plane_value = 0;
    std::vector<double> tris;  // 18 doubles per triangle
    std::vector<Shape> children;  // RPT_SHAPE_GROUP: KdTree<Box<dyn Bounded>> (kdtree.rs:103-146)

    Shape translate(Vec3 v) const { return wrap(Mat4::translation(v)); }
    Shape scale(Vec3 v) const { return wrap(Mat4::scaling(v)); }
    Shape rotate(double angle, Vec3 axis) const { return wrap(Mat4::rotation(angle, axis)); }
    Shape rotate_x(double a) const { return wrap(Mat4::rotation(a, {1, 0, 0})); }
    Shape rotate_y(double a) const { return wrap(Mat4::rotation(a, {0, 1, 0})); }
    Shape rotate_z(double a) const { return wrap(Mat4::rotation(a, {0, 0, 1})); }
    Shape transform(const Mat4& t) const { return wrap(t); }

    rpt_shape_desc desc() const {
        rpt_shape_desc d{};
        d.kind = kind;
        d.has_transform = has_transform ? 1 : 0;
        for (int i = 0; i < 16; i++) d.transform[i] = matrix.m[i];
        for (int i = 0; i < 3; i++) d.plane_normal[i] = plane_normal[i];
        d.plane_value = plane_value;
        d.tris = tris.empty() ? nullptr : tris.data();
        d.n_tris = tris.size() / 18;
        child_descs_.clear();
        for (const Shape& c : children) child_descs_.push_back(c.desc());  // valid while *this is alive and unchanged
        d.children = child_descs_.empty() ? nullptr : child_descs_.data();
        d.n_children = child_descs_.size();
        return d;
    }

  private:
    mutable std::vector<rpt_shape_desc> child_descs_;
    Shape wrap(const Mat4& t) const {
        Shape s = *this;
        s.matrix = has_transform ? t * matrix : t;
        s.has_transform = true;
        return s;
    }
};
using Mesh = Shape;  // `Mesh = KdTree<Triangle>` (shape/mesh.rs:103); the accelerator is built in the library
inline Shape sphere() { return Shape{}; }
inline Shape cube() {
    Shape s;
    s.kind = RPT_SHAPE_CUBE;
    return s;
}
inline Shape plane(Vec3 normal, double value) {
    Shape s;
    s.kind = RPT_SHAPE_PLANE;
    s.plane_normal = normal;
    s.plane_value = value;
    return s;
}
inline Shape mesh(const std::vector<Triangle>& ts) {  // Mesh::new
    Shape s;
    s.kind = RPT_SHAPE_MESH;
    for (const Triangle& t : ts)
        for (const Vec3* v : {&t.v1, &t.v2, &t.v3, &t.n1, &t.n2, &t.n3})
            for (double c : *v) s.tris.push_back(c);
    return s;
}
inline Shape kdtree(std::vector<Shape> shapes) {  // KdTree::new over bounded shapes (examples/fractal_spheres.rs:45)
    Shape s;
    s.kind = RPT_SHAPE_GROUP;
    for (const Shape& c : shapes)
        if (c.kind == RPT_SHAPE_PLANE) throw std::invalid_argument("Plane is not Bounded and cannot be put in a KdTree");
    if (shapes.empty()) throw std::invalid_argument("KdTree needs at least one shape");
    s.children = std::move(shapes);
    return s;
}
inline Shape polygon(const std::vector<Vec3>& verts) {  // shape.rs:308-314
    std::vector<Triangle> ts;
    for (size_t i = 1; i + 1 < verts.size(); i++) ts.push_back(Triangle::from_vertices(verts[0], verts[i], verts[i + 1]));
    return mesh(ts);
}

// ---- io.rs: host-only mesh loaders (they only produce the triangle array the hot path consumes)
namespace io_detail {
inline bool parse_index(const std::string& v, size_t length, long& out) {  // io.rs:12-20
    if (v.empty()) return false;
    char* end = nullptr;
    long i = std::strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end != '\0') return false;
    out = i > 0 ? i - 1 : long(length) + i;
    return true;
}
}  // namespace io_detail
// load_obj (io.rs:28-74, faces :164-201): `v`, `vn`, `f` with a, a/b, a//c, a/b/c indices (1-based, negative =
// relative to the end), polygons fan-triangulated, face normals when a corner lacks a `vn`; other commands skipped.
inline Shape load_obj(std::istream& in) {
    std::vector<Vec3> vertices, normals;
    std::vector<Triangle> tris;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string cmd;
        if (!(ls >> cmd) || cmd[0] == '#') continue;
        if (cmd == "v" || cmd == "vn") {
            double x, y, z;
            if (!(ls >> x >> y >> z)) throw Error("Invalid point in .OBJ file");
            (cmd == "v" ? vertices : normals).push_back({x, y, z});
        } else if (cmd == "f") {
            std::vector<long> vi, vni;
            std::string tok;
            while (ls >> tok) {
                std::string part[3];
                size_t k = 0, start = 0;
                for (size_t c = 0; c <= tok.size() && k < 3; c++)
                    if (c == tok.size() || tok[c] == '/') { part[k++] = tok.substr(start, c - start); start = c + 1; }
                long v = 0, n = -1;
                if (!io_detail::parse_index(part[0], vertices.size(), v) || v < 0 || size_t(v) >= vertices.size())
                    throw Error("Invalid vertex index");
                if (!io_detail::parse_index(part[2], normals.size(), n)) n = -1;
                vi.push_back(v);
                vni.push_back(n);
            }
            for (size_t i = 1; i + 1 < vi.size(); i++) {
                const size_t c[3] = {0, i, i + 1};
                const Vec3 &v1 = vertices[vi[c[0]]], &v2 = vertices[vi[c[1]]], &v3 = vertices[vi[c[2]]];
                if (vni[c[0]] < 0 || vni[c[1]] < 0 || vni[c[2]] < 0) tris.push_back(Triangle::from_vertices(v1, v2, v3));
                else tris.push_back({v1, v2, v3, normals.at(vni[c[0]]), normals.at(vni[c[1]]), normals.at(vni[c[2]])});
            }
        }
    }
    return mesh(tris);
}
// load_stl (io.rs:264-364): ASCII or binary STL, face normals recomputed from the vertices.
inline Shape load_stl(std::istream& in) {
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<Triangle> tris;
    const size_t first = data.find_first_not_of(" \t\r\n");
    const bool ascii = first != std::string::npos && data.compare(first, 5, "solid") == 0 &&
                       data.substr(0, std::min<size_t>(data.size(), 2048)).find("facet") != std::string::npos;
    if (ascii) {
        std::istringstream ss(data);
        std::string word;
        std::vector<Vec3> pts;
        while (ss >> word)
            if (word == "vertex") {
                double x, y, z;
                if (!(ss >> x >> y >> z)) throw Error("Invalid vertex in .STL file");
                pts.push_back({x, y, z});
                if (pts.size() == 3) { tris.push_back(Triangle::from_vertices(pts[0], pts[1], pts[2])); pts.clear(); }
            }
    } else {
        if (data.size() < 84) throw Error("Invalid binary .STL file");
        uint32_t n;
        std::memcpy(&n, data.data() + 80, 4);
        if (data.size() < 84 + size_t(n) * 50) throw Error("Truncated binary .STL file");
        for (uint32_t i = 0; i < n; i++) {
            float f[12];
            std::memcpy(f, data.data() + 84 + size_t(i) * 50, 48);
            tris.push_back(Triangle::from_vertices({f[3], f[4], f[5]}, {f[6], f[7], f[8]}, {f[9], f[10], f[11]}));
        }
    }
    return mesh(tris);
}

// ---- material.rs
struct Material {
    int kind = RPT_MAT_LAMBERTIAN;
    Color albedo{0.5, 0.5, 0.5};
    double emittance_ = 0, shininess = 0, ior = 1;
    static Material diffuse(Color c) { return {RPT_MAT_LAMBERTIAN, c, 0, 0, 1}; }
    static Material specular(Color c, double roughness) { return {RPT_MAT_PHONG, c, 0, roughness, 1}; }
    static Material metallic(Color c, double roughness) { return {RPT_MAT_PHONG, c, 0, roughness, 1}; }
    static Material mirror() { return {RPT_MAT_MIRROR, {0, 0, 0}, 0, 0, 1}; }
    static Material transmissive(double ior) { return {RPT_MAT_TRANSMISSIVE, {0, 0, 0}, 0, 0, ior}; }
    static Material clear(double index, double /*roughness*/) { return {RPT_MAT_TRANSMISSIVE, {0, 0, 0}, 0, 0, index}; }
    static Material transparent(Color c, double index, double /*roughness*/) { return {RPT_MAT_TRANSMISSIVE, c, 0, 0, index}; }
    static Material light(Color c, double emittance) { return {RPT_MAT_LAMBERTIAN, c, emittance, 0, 1}; }
    double emittance() const { return kind <= RPT_MAT_PHONG ? emittance_ : 0.0; }
    Color color() const { return kind <= RPT_MAT_PHONG ? albedo : Color{0, 0, 0}; }
    rpt_material desc() const {
        rpt_material m{};
        m.kind = kind;
        for (int i = 0; i < 3; i++) m.albedo[i] = albedo[i];
        m.emittance = emittance_;
        m.shininess = shininess;
        m.ior = ior;
        return m;
    }
};

// ---- object.rs / light.rs / medium.rs / environment.rs
struct Object {
    Shape shape;
    Material material_;
    explicit Object(Shape s) : shape(std::move(s)) {}
    static Object new_(Shape s) { return Object(std::move(s)); }
    Object material(Material m) && {
        material_ = m;
        return std::move(*this);
    }
    Object material(Material m) const& {
        Object o = *this;
        o.material_ = m;
        return o;
    }
};
struct Light {
    enum Kind { POINT, AMBIENT, DIRECTIONAL, OBJECT } kind;
    Color color{0, 0, 0};
    Vec3 vec{0, 0, 0};
    std::vector<rpt::Object> object;  // 0 or 1 element
    static Light Point(Color c, Vec3 location) { return {POINT, c, location, {}}; }
    static Light Ambient(Color c) { return {AMBIENT, c, {0, 0, 0}, {}}; }
    static Light Directional(Color c, Vec3 direction) { return {DIRECTIONAL, c, direction, {}}; }
    static Light Object(rpt::Object o) { return {OBJECT, {0, 0, 0}, {0, 0, 0}, {std::move(o)}}; }
};
struct Medium {
    int kind;
    double absorption, scattering;
    static Medium homogeneous_isotropic(double a, double s) { return {RPT_MEDIUM_HOMOGENEOUS_ISOTROPIC, a, s}; }
    static Medium colored_glowing_fog(double a, double s) { return {RPT_MEDIUM_COLORED_GLOWING_FOG, a, s}; }
};
struct Environment {  // environment.rs:3-77
    Color color{0, 0, 0};
    uint32_t hdri_width = 0, hdri_height = 0;
    std::vector<double> hdri;  // width*height*3, row-major
    static Environment Color_(Color c) { return {c, 0, 0, {}}; }
    static Environment Hdri(uint32_t width, uint32_t height, std::vector<double> buf) {
        if (buf.size() != size_t(width) * height * 3 || width == 0 || height == 0) throw Error("Hdri::new: bad dimensions");
        return {{0, 0, 0}, width, height, std::move(buf)};
    }
};

// ---- scene.rs
class Scene {
  public:
    std::vector<Object> objects;
    std::vector<Light> lights;
    std::vector<Medium> media;
    Environment environment;
    static Scene new_() { return {}; }
    void add(Object o) { objects.push_back(std::move(o)); }
    void add(Light l) { lights.push_back(std::move(l)); }
    void add(Medium m) { media.push_back(m); }
    // SceneAdd<(Mesh, Material)> / SceneAdd<(Transformed<Cube>, Material)>: object AND light (scene.rs:57-75)
    void add(const std::pair<Shape, Material>& m) {
        bool ok = m.first.kind == RPT_SHAPE_MESH || (m.first.kind == RPT_SHAPE_CUBE && m.first.has_transform);
        if (!ok) throw Error("SceneAdd is implemented for (Mesh, Material) and (Transformed<Cube>, Material)");
        add(Object(m.first).material(m.second));
        add(Light::Object(Object(m.first).material(m.second)));
    }
};

// ---- camera.rs
struct Camera {
    Vec3 eye{0, 0, 10}, direction{0, 0, -1}, up{0, 1, 0};
    double fov = 0.52359877559829887308, aperture = 0, focal_distance = 0;
    static Camera look_at(Vec3 eye, Vec3 center, Vec3 up, double fov) {
        Camera c;
        c.eye = eye;
        c.direction = normalize(center - eye);
        c.up = normalize(up - dot(up, c.direction) * c.direction);
        c.fov = fov;
        return c;
    }
    Camera focus(Vec3 focal_point, double aperture_) const {
        Camera c = *this;
        c.focal_distance = dot(focal_point - eye, direction);
        c.aperture = aperture_;
        return c;
    }
    rpt_camera desc() const {
        rpt_camera c{};
        for (int i = 0; i < 3; i++) {
            c.eye[i] = eye[i];
            c.direction[i] = direction[i];
            c.up[i] = up[i];
        }
        c.fov = fov;
        c.aperture = aperture;
        c.focal_distance = focal_distance;
        return c;
    }
};

// ---- buffer.rs
struct Filter {
    uint32_t radius = 0;
    static Filter Box(uint32_t r) { return {r}; }
};
struct RgbImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;  // row-major RGB
};
// buffer.rs:5-97 on the device (rpt_buffer_*): per-pixel running sums; the box filter, color_bytes
// and the variance run where the frame is.
class Buffer {
  public:
    Buffer(uint32_t w, uint32_t h, Filter f, int device = 0) : width(w), height(h), filter(f) {
        h_ = rpt_buffer_create(device, w, h, f.radius);
        if (!h_) throw Error(rpt_last_error());
    }
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    Buffer(Buffer&& o) noexcept : width(o.width), height(o.height), filter(o.filter), h_(o.h_) { o.h_ = nullptr; }
    ~Buffer() {
        if (h_) rpt_buffer_destroy(h_);
    }
    void add_samples(const std::vector<double>& s) {  // buffer.rs:32-40
        if (s.size() != size_t(width) * height * 3) throw Error("Invalid sample dimension");
        if (rpt_buffer_add_samples(h_, s.data()) != RPT_OK) throw Error(rpt_last_error());
    }
    RgbImage image() const {  // buffer.rs:43-56
        RgbImage img{width, height, std::vector<uint8_t>(size_t(width) * height * 3)};
        if (rpt_buffer_image(h_, img.data.data()) != RPT_OK) throw Error(rpt_last_error());
        return img;
    }
    double variance() const {  // buffer.rs:59-73
        double v = 0;
        if (rpt_buffer_variance(h_, &v) != RPT_OK) throw Error(rpt_last_error());
        return v;
    }
    uint32_t batches() const {  // samples[i].len() of the reference (equal for every pixel)
        uint32_t n = 0;
        if (rpt_buffer_batches(h_, &n) != RPT_OK) throw Error(rpt_last_error());
        return n;
    }
    rpt_buffer* raw() const { return h_; }
    uint32_t width, height;
    Filter filter;

  private:
    rpt_buffer* h_ = nullptr;
};

// ---- renderer.rs
class Renderer {
  public:
    Renderer(const Scene& scene, Camera camera) : scene_(scene), camera_(camera) {}
    static Renderer new_(const Scene& scene, Camera camera) { return Renderer(scene, camera); }
    Renderer(const Renderer&) = delete;
    Renderer(Renderer&& o) noexcept : scene_(o.scene_), camera_(o.camera_), p_(o.p_), handle_(o.handle_) { o.handle_ = nullptr; }
    ~Renderer() {
        if (handle_) rpt_scene_destroy(handle_);
    }
    struct Params {
        uint32_t width = 800, height = 600;
        double exposure_value = 0, stepsize = 0;
        Filter filter;
        uint32_t max_bounces = 0, num_samples = 1;
        size_t gather_size = 50, gather_size_volume = 50;
        double watts = 100;
        uint64_t seed = 0;                   // addition: the reference seeds from entropy (renderer.rs:163)
        uint32_t shard_rank = 0, shard_count = 1;
        int device = 0;
    };
#define RPT_BUILDER(name, type) \
    Renderer&& name(type v) && { p_.name = v; return std::move(*this); } \
    Renderer& name(type v) & { p_.name = v; return *this; }
    RPT_BUILDER(width, uint32_t)
    RPT_BUILDER(height, uint32_t)
    RPT_BUILDER(exposure_value, double)
    RPT_BUILDER(stepsize, double)
    RPT_BUILDER(filter, Filter)
    RPT_BUILDER(max_bounces, uint32_t)
    RPT_BUILDER(num_samples, uint32_t)
    RPT_BUILDER(gather_size, size_t)
    RPT_BUILDER(gather_size_volume, size_t)
    RPT_BUILDER(watts, double)
    RPT_BUILDER(seed, uint64_t)
    RPT_BUILDER(device, int)
#undef RPT_BUILDER
    Renderer& shard(uint32_t rank, uint32_t count) {
        p_.shard_rank = rank;
        p_.shard_count = count;
        return *this;
    }
    const Params& params() const { return p_; }

    RgbImage render() {  // renderer.rs:137-141
        Buffer buffer(p_.width, p_.height, p_.filter, p_.device);
        sample_offset_ = 0;
        sample(p_.num_samples, buffer);
        return buffer.image();
    }
    void iterative_render(uint32_t callback_interval, const std::function<void(uint32_t, const Buffer&)>& cb) {
        Buffer buffer(p_.width, p_.height, p_.filter, p_.device);  // renderer.rs:144-156
        sample_offset_ = 0;
        uint32_t iteration = 0;
        while (iteration < p_.num_samples) {
            uint32_t steps = std::min(p_.num_samples - iteration, callback_interval);
            sample(steps, buffer);
            iteration += steps;
            cb(iteration, buffer);
        }
    }
    // ---- photon mapping (photon.rs:631-720)
    enum PhotonRenderKind { PhotonMap = RPT_PHOTON_MAP, PhotonPointBeam = RPT_PHOTON_POINT_BEAM, PhotonBeamBeam = RPT_PHOTON_BEAM_BEAM };
    RgbImage photon_render(size_t photon_count, PhotonRenderKind kind) {  // photon.rs:655-720
        commit();
        check(rpt_photon_map_build(handle_, photon_count, int32_t(kind), p_.watts, p_.seed));
        Buffer buffer(p_.width, p_.height, p_.filter, p_.device);
        std::vector<double> out(size_t(p_.width) * p_.height * 3);
        rpt_camera cam = camera_.desc();
        rpt_render_params rp{p_.width, p_.height, p_.exposure_value, p_.max_bounces, p_.shard_rank, p_.shard_count};
        check(rpt_photon_render_sample(handle_, &cam, &rp, p_.gather_size, p_.gather_size_volume, p_.num_samples, p_.seed, 0,
                                       out.data()));
        buffer.add_samples(out);
        return buffer.image();
    }
    RgbImage photon_point_query_beam_render(size_t n) { return photon_render(n, PhotonPointBeam); }  // :642-644
    RgbImage photon_beam_query_beam_render(size_t n) { return photon_render(n, PhotonBeamBeam); }   // :646-648
    RgbImage photon_map_render(size_t n) { return photon_render(n, PhotonMap); }                     // :650-652

    // One batch of a multi-GPU render: this rank's tiles into the device frame d_frame
    // (width * height * 3 f64 on this renderer's device), then the gather of every rank's tiles onto rank 0, all
    // enqueued on hip_stream.  shard(rank, count) must match the communicator.
    void sample_sharded(uint32_t iterations, class Comm& comm, void* d_frame, void* hip_stream = nullptr);

    // Renderer::sample (renderer.rs:158-171): the call that crosses the C ABI.
    void sample(uint32_t iterations, Buffer& buffer) {  // renderer.rs:158-171; the batch stays on the device
        commit();
        rpt_camera cam = camera_.desc();
        rpt_render_params rp{p_.width, p_.height, p_.exposure_value, p_.max_bounces, p_.shard_rank, p_.shard_count};
        check(rpt_render_into_buffer(handle_, &cam, &rp, iterations, p_.seed, sample_offset_, buffer.raw()));
        sample_offset_ += iterations;
    }

  private:
    void commit() {
        if (handle_) return;
        rpt_scene* h = rpt_scene_create();
        try {
            for (const Object& o : scene_.objects) {
                rpt_shape_desc d = o.shape.desc();
                rpt_material m = o.material_.desc();
                check(rpt_scene_add_object(h, &d, &m));
            }
            for (const Light& l : scene_.lights) {
                switch (l.kind) {
                    case Light::POINT: check(rpt_scene_add_light_point(h, l.color.data(), l.vec.data())); break;
                    case Light::AMBIENT: check(rpt_scene_add_light_ambient(h, l.color.data())); break;
                    case Light::DIRECTIONAL: check(rpt_scene_add_light_directional(h, l.color.data(), l.vec.data())); break;
                    default: {
                        rpt_shape_desc d = l.object.at(0).shape.desc();
                        rpt_material m = l.object.at(0).material_.desc();
                        check(rpt_scene_add_light_object(h, &d, &m));
                    }
                }
            }
            for (const Medium& m : scene_.media) check(rpt_scene_add_medium(h, m.kind, m.absorption, m.scattering));
            if (scene_.environment.hdri_width)
                check(rpt_scene_set_environment_hdri(h, scene_.environment.hdri_width, scene_.environment.hdri_height,
                                                     scene_.environment.hdri.data()));
            else
                check(rpt_scene_set_environment_color(h, scene_.environment.color.data()));
            check(rpt_scene_commit(h, p_.device));
        } catch (...) {
            rpt_scene_destroy(h);
            throw;
        }
        handle_ = h;
    }
    const Scene& scene_;
    Camera camera_;
    Params p_;
    rpt_scene* handle_ = nullptr;
    uint32_t sample_offset_ = 0;
};

// The frame exchange between the GPUs of one node (rpt_comm_*, rpt_gather_frame_device): one process per GPU, rank 0
// assembles the image from the tiles every rank owns.  Rank 0 draws the id (Comm::unique_id) and hands its 128 bytes to
// the other ranks by whatever means the launcher offers.
class Comm {
  public:
    using Id = std::array<unsigned char, RPT_COMM_ID_BYTES>;
    static Id unique_id() {
        Id id{};
        check(rpt_comm_unique_id(id.data()));
        return id;
    }
    Comm(const Id& id, int rank, int n_ranks, int device) { check(rpt_comm_create(id.data(), rank, n_ranks, device, &h_)); }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    ~Comm() { rpt_comm_destroy(h_); }
    int rank() const { int r = 0; check(rpt_comm_rank(h_, &r, nullptr)); return r; }
    int size() const { int n = 0; check(rpt_comm_rank(h_, nullptr, &n)); return n; }
    // Collective: d_shard = this rank's frame, d_frame = where rank 0 assembles (may be d_shard; ignored elsewhere).
    void gather(uint32_t width, uint32_t height, const void* d_shard, void* d_frame, void* hip_stream = nullptr, bool loopback = false) {
        check(rpt_gather_frame_device(h_, width, height, d_shard, d_frame, loopback ? RPT_GATHER_LOOPBACK : 0u, hip_stream));
    }
    rpt_comm* raw() const { return h_; }

  private:
    rpt_comm* h_ = nullptr;
};
inline void Renderer::sample_sharded(uint32_t iterations, Comm& comm, void* d_frame, void* hip_stream) {
    commit();
    rpt_camera cam = camera_.desc();
    rpt_render_params rp{p_.width, p_.height, p_.exposure_value, p_.max_bounces, p_.shard_rank, p_.shard_count};
    check(rpt_render_sample_device(handle_, &cam, &rp, iterations, p_.seed, sample_offset_, d_frame, hip_stream));
    sample_offset_ += iterations;
    comm.gather(p_.width, p_.height, d_frame, d_frame, hip_stream);
}

}  // namespace rpt
