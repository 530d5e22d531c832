// rpt_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// fp64 CPU restatement of the per-pixel-sample hot path of neevparikh/rpt
// (Renderer::sample -> get_color -> trace_ray -> get_closest_hit ->
// Shape::intersect, plus Material / Medium / Light / Camera), written from the
// reference's Rust sources as the parity checker for the HIP path.  Every
// function cites the reference file:line it follows (paths relative to the
// reference root).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, load or call this file.
//
// PARITY STATUS: the reference is Rust and cannot be compiled or run in this
// environment (no cargo/rustc, crates not vendored).  Its own test-suite holds a
// single known-answer test on this path (src/color.rs:30-38, hex_color /
// color_bytes) which this oracle reproduces; everything else on the path is
// "parity unpinned": pinned only by the known-answer tests in tests/ that were
// derived by hand from the cited lines.
//
// Deliberate, documented deviation: the reference seeds StdRng::from_entropy()
// per image row (src/renderer.rs:163) and is therefore not reproducible.  The
// oracle (and the HIP path) draw from a per-(seed, pixel, sample) stream:
// xoshiro128+ seeded through splitmix64, uniform u = (2*(x>>9)+1) * 2^-24 in
// (0,1).  Only the distributions follow the reference (rand 0.8 / rand_distr 0.4
// call sites listed in SURVEY.md Appendix B).
//
// Third-party math the reference calls (nalgebra 0.24 / nalgebra-glm 0.10, not in
// the mount) is restated from its published semantics; see rotation_between().

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace orc {

static const double PI = 3.14159265358979323846264338327950288;
static const double INF = std::numeric_limits<double>::infinity();

// ------------------------------------------------------------------ vectors
struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(double s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline V3 cmul(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double length(V3 a) { return std::sqrt(dot(a, a)); }
// glm::normalize / nalgebra .normalize(): v / ||v|| (component-wise division).
static inline V3 normalize(V3 a) { return a / length(a); }
static inline V3 vmin(V3 a, V3 b) { return V3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
static inline V3 vmax(V3 a, V3 b) { return V3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }

struct M3 {
    double m[3][3];
};
struct M4 {
    double m[4][4];
};
static inline V3 mul(const M3& a, V3 v) {
    return V3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
              a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
// M * (v, w).xyz
static inline V3 mul(const M4& a, V3 v, double w) {
    return V3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z + a.m[0][3] * w,
              a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z + a.m[1][3] * w,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z + a.m[2][3] * w);
}
static double det3(const M3& a) {
    return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) -
           a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
           a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
static M3 inverse3(const M3& a) {
    double d = det3(a);
    M3 r;
    r.m[0][0] = (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) / d;
    r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) / d;
    r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) / d;
    r.m[1][0] = (a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2]) / d;
    r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) / d;
    r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) / d;
    r.m[2][0] = (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]) / d;
    r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) / d;
    r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) / d;
    return r;
}
static M3 transpose3(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
// glm::inverse for a general 4x4 (Gauss-Jordan with partial pivoting, fp64).
static M4 inverse4(const M4& a) {
    double w[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            w[i][j] = a.m[i][j];
            w[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 8; j++) std::swap(w[p][j], w[c][j]);
        double d = w[c][c];
        for (int j = 0; j < 8; j++) w[c][j] /= d;
        for (int r = 0; r < 4; r++)
            if (r != c) {
                double f = w[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 8; j++) w[r][j] -= f * w[c][j];
            }
    }
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = w[i][4 + j];
    return r;
}

// ------------------------------------------------------------------ RNG
// Deviation from src/renderer.rs:163 (StdRng::from_entropy per row): see header.
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
struct Rng {
    uint32_t s0, s1, s2, s3;
    uint64_t draws;
    Rng(uint64_t seed, uint32_t pixel, uint32_t sample) : draws(0) {
        const uint64_t G = 0x9E3779B97F4A7C15ULL;
        uint64_t a = mix64(seed + G);
        uint64_t z = a ^ ((uint64_t(sample) << 32) | uint64_t(pixel));
        uint64_t r0 = mix64(z + G), r1 = mix64(z + 2 * G);
        s0 = uint32_t(r0);
        s1 = uint32_t(r0 >> 32);
        s2 = uint32_t(r1);
        s3 = uint32_t(r1 >> 32);
    }
    inline uint32_t next() {  // xoshiro128+
        uint32_t r = s0 + s3;
        uint32_t t = s1 << 9;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        draws++;
        return r;
    }
    // rng.gen::<f64>() stand-in: (2k+1) * 2^-24, k = top 23 bits; in (0,1), exact in fp32.
    inline double uniform() { return double(((next() >> 9) << 1) | 1u) * (1.0 / 16777216.0); }
    // rng.gen_range(a..b) stand-in.
    inline double range(double a, double b) { return a + (b - a) * uniform(); }
    // rng.sample(Uniform::from(0..n)) stand-in: widening multiply, no rejection.
    inline uint32_t index(uint32_t n) { return uint32_t((uint64_t(next()) * uint64_t(n)) >> 32); }
    // rand_distr::UnitDisc: rejection from the square [-1,1)^2.
    inline void unit_disc(double& x, double& y) {
        for (;;) {
            x = range(-1.0, 1.0);
            y = range(-1.0, 1.0);
            if (x * x + y * y <= 1.0) return;
        }
    }
};

// ------------------------------------------------------------------ counters
struct Counters {
    uint64_t rays = 0;        // get_closest_hit calls
    uint64_t obj_tests = 0;   // object.shape.intersect calls (renderer.rs:420)
    uint64_t nodes = 0;       // kd nodes visited (kdtree.rs:154)
    uint64_t tri_tests = 0;   // Triangle::intersect calls (mesh.rs:50)
    uint64_t hits = 0;        // accepted closest hits
    uint64_t samples = 0;     // camera samples
    uint64_t vertices = 0;    // trace_ray invocations
    uint64_t self_hits = 0;   // diagnostic: accepted hits with t < 1e-9 * (1+|o|)
    uint64_t shadow_tests = 0;
    uint64_t shadow_pass = 0;
    uint64_t shadow_near = 0;  // |hit-dist| in [1e-12, 1e-6*dist): fp64 false rejections
    void add(const Counters& o) {
        rays += o.rays; obj_tests += o.obj_tests; nodes += o.nodes; tri_tests += o.tri_tests;
        hits += o.hits; samples += o.samples; vertices += o.vertices; self_hits += o.self_hits;
        shadow_tests += o.shadow_tests; shadow_pass += o.shadow_pass; shadow_near += o.shadow_near;
    }
};
static thread_local Counters* tl_cnt = nullptr;
#define CNT(field) do { if (tl_cnt) tl_cnt->field++; } while (0)

// ------------------------------------------------------------------ shape.rs
struct Ray {  // src/shape.rs:50-72
    V3 origin, dir;
    V3 at(double t) const { return origin + t * dir; }
    Ray apply_transform(const M4& m) const {  // :65-72, direction NOT renormalised
        Ray r;
        r.origin = mul(m, origin, 1.0);
        r.dir = mul(m, dir, 0.0);
        return r;
    }
};
struct HitRecord {  // src/shape.rs:75-98
    double time = INF;
    V3 normal = V3(0, 0, 0);
};
struct BBox {  // src/kdtree.rs:29-90
    V3 p_min = V3(INF, INF, INF), p_max = V3(-INF, -INF, -INF);
    BBox merge(const BBox& o) const {
        BBox b;
        b.p_min = vmin(p_min, o.p_min);
        b.p_max = vmax(p_max, o.p_max);
        return b;
    }
    void intersect(const Ray& ray, double& tmin, double& tmax) const {  // :57-72
        double x1 = (p_min.x - ray.origin.x) / ray.dir.x, x2 = (p_max.x - ray.origin.x) / ray.dir.x;
        double a1 = std::fmin(x1, x2), a2 = std::fmax(x1, x2);
        double y1 = (p_min.y - ray.origin.y) / ray.dir.y, y2 = (p_max.y - ray.origin.y) / ray.dir.y;
        double b1 = std::fmin(y1, y2), b2 = std::fmax(y1, y2);
        double z1 = (p_min.z - ray.origin.z) / ray.dir.z, z2 = (p_max.z - ray.origin.z) / ray.dir.z;
        double c1 = std::fmin(z1, z2), c2 = std::fmax(z1, z2);
        tmin = std::fmax(std::fmax(a1, b1), c1);
        tmax = std::fmin(std::fmin(a2, b2), c2);
    }
    void split(int axis, double value, BBox& lo, BBox& hi) const {  // :75-89
        lo = *this;
        hi = *this;
        lo.p_max.at(axis) = value;
        hi.p_min.at(axis) = value;
    }
};
struct SurfSample {
    V3 v, n;
    double p;
};
struct Shape {  // trait Shape, src/shape.rs:19-26
    virtual ~Shape() {}
    virtual bool intersect(const Ray& ray, double t_min, HitRecord& rec) const = 0;
    virtual SurfSample sample(const V3& target, Rng& rng) const = 0;
    virtual bool can_sample() const { return true; }
    virtual bool bounding_box(BBox&) const { return false; }  // trait Bounded, src/kdtree.rs:12-15
};

struct Sphere : Shape {  // src/shape/sphere.rs:14-75
    bool bounding_box(BBox& b) const override { b.p_min = V3(-1, -1, -1); b.p_max = V3(1, 1, 1); return true; }
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override {
        double a = dot(ray.dir, ray.dir);
        double b = dot(ray.dir, ray.origin);
        double c = dot(ray.origin, ray.origin) - 1.0;
        double d = b * b - a * c;
        if (std::signbit(d)) return false;  // is_sign_negative (NaN with sign bit too)
        d = std::sqrt(d);
        double t;
        double t_minus = (-b - d) / a;
        if (t_minus < t_min) {
            double t_plus = (-b + d) / a;
            if (t_plus < t_min) return false;
            t = t_plus;
        } else {
            t = t_minus;
        }
        if (t < rec.time) {
            rec.time = t;
            rec.normal = normalize(ray.at(t));
            return true;
        }
        return false;
    }
    SurfSample sample(const V3& target, Rng& rng) const override {  // :53-65
        double x, y;
        rng.unit_disc(x, y);
        double z = std::sqrt(1.0 - x * x - y * y);
        V3 n = normalize(target);
        V3 n1 = std::isnormal(n.x) ? normalize(V3(n.y, -n.x, 0.0)) : normalize(V3(0.0, -n.z, n.y));
        V3 n2 = cross(n1, n);
        V3 p = x * n1 + y * n2 + z * n;
        return SurfSample{p, p, z * (1.0 / PI)};
    }
};

struct Cube : Shape {  // src/shape/cube.rs:12-89
    bool bounding_box(BBox& b) const override { b.p_min = V3(-0.5, -0.5, -0.5); b.p_max = V3(0.5, 0.5, 0.5); return true; }
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override {
        double lo[3], hi[3];
        V3 lon[3], hin[3];
        for (int dim = 0; dim < 3; dim++) {  // compute_interval :23-35
            double x1 = (-0.5 - ray.origin[dim]) / ray.dir[dim];
            double x2 = (0.5 - ray.origin[dim]) / ray.dir[dim];
            V3 x1n(0, 0, 0), x2n(0, 0, 0);
            x1n.at(dim) = -1.0;
            x2n.at(dim) = 1.0;
            if (x1 > x2) {
                std::swap(x1, x2);
                std::swap(x1n, x2n);
            }
            lo[dim] = x1; hi[dim] = x2; lon[dim] = x1n; hin[dim] = x2n;
        }
        double start, end;
        V3 sn, en;
        if (lo[0] > lo[1] && lo[0] > lo[2]) { start = lo[0]; sn = lon[0]; }
        else if (lo[1] > lo[2]) { start = lo[1]; sn = lon[1]; }
        else { start = lo[2]; sn = lon[2]; }
        if (hi[0] < hi[1] && hi[0] < hi[2]) { end = hi[0]; en = hin[0]; }
        else if (hi[1] < hi[2]) { end = hi[1]; en = hin[1]; }
        else { end = hi[2]; en = hin[2]; }
        if (start > end || end < t_min) return false;
        double time; V3 normal;
        if (start < t_min) { time = end; normal = en; } else { time = start; normal = sn; }
        if (time < rec.time) {
            rec.time = time;
            rec.normal = normal;
            return true;
        }
        return false;
    }
    SurfSample sample(const V3&, Rng& rng) const override {  // :76-89
        double a = rng.uniform() - 0.5;
        double b = rng.uniform() - 0.5;
        V3 v, n;
        switch (rng.index(6)) {
            case 0: v = V3(a, b, 0.5); n = V3(0, 0, 1); break;
            case 1: v = V3(a, b, -0.5); n = V3(0, 0, -1); break;
            case 2: v = V3(a, 0.5, b); n = V3(0, 1, 0); break;
            case 3: v = V3(a, -0.5, b); n = V3(0, -1, 0); break;
            case 4: v = V3(0.5, a, b); n = V3(1, 0, 0); break;
            default: v = V3(-0.5, a, b); n = V3(-1, 0, 0); break;
        }
        return SurfSample{v, n, 1.0 / 6.0};
    }
};

struct Plane : Shape {  // src/shape/plane.rs:17-36
    V3 normal;
    double value;
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override {
        double cosine = dot(normal, ray.dir);
        if (std::fabs(cosine) < 1e-8) return false;
        double time = (value - dot(normal, ray.origin)) / cosine;
        if (time >= t_min && time < rec.time) {
            rec.time = time;
            // f64::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives
            double sg = std::signbit(cosine) ? -1.0 : 1.0;
            rec.normal = -(normalize(normal)) * sg;
            return true;
        }
        return false;
    }
    SurfSample sample(const V3&, Rng&) const override { return SurfSample{V3(), V3(), 1.0}; }  // unimplemented!() :34
    bool can_sample() const override { return false; }
};

struct Triangle {  // src/shape/mesh.rs:9-99
    V3 v1, v2, v3, n1, n2, n3;
    BBox bounding_box() const {
        BBox b;
        b.p_min = vmin(vmin(v1, v2), v3);
        b.p_max = vmax(vmax(v1, v2), v3);
        return b;
    }
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const {  // :50-83
        CNT(tri_tests);
        V3 d0 = v2 - v1, d1 = v3 - v1;
        V3 plane_normal = normalize(cross(d0, d1));
        double cosine = dot(plane_normal, ray.dir);
        if (std::fabs(cosine) < 1e-8) return false;
        double time = dot(plane_normal, v1 - ray.origin) / cosine;
        if (time < t_min || time >= rec.time) return false;
        V3 d2 = ray.at(time) - v1;
        double d00 = dot(d0, d0), d01 = dot(d0, d1), d11 = dot(d1, d1);
        double d20 = dot(d2, d0), d21 = dot(d2, d1);
        double denom = d00 * d11 - d01 * d01;
        double v = (d11 * d20 - d01 * d21) / denom;
        double w = (d00 * d21 - d01 * d20) / denom;
        double u = 1.0 - v - w;
        if (u >= 0.0 && v >= 0.0 && w >= 0.0) {
            rec.time = time;
            rec.normal = normalize(u * n1 + v * n2 + w * n3);
            return true;
        }
        return false;
    }
    SurfSample sample(Rng& rng) const {  // :85-99
        double u = rng.uniform(), v = rng.uniform();
        while (u + v > 1.0) {
            u = rng.uniform();
            v = rng.uniform();
        }
        double w = 1.0 - u - v;
        double area = 0.5 * length(cross(v2 - v1, v3 - v1));
        return SurfSample{u * v1 + v * v2 + w * v3, normalize(u * n1 + v * n2 + w * n3), 1.0 / area};
    }
};

// src/kdtree.rs:230-236
struct KdNode {
    int axis = -1;  // -1 leaf, 0/1/2 split
    double value = 0;
    std::unique_ptr<KdNode> left, right;
    std::vector<size_t> indices;
};
static double median(const std::vector<double>& s) {  // src/kdtree.rs:350-358
    size_t n = s.size();
    if (n % 2 == 1) return s[n / 2];
    size_t mid = n / 2;
    return (s[mid] + s[mid - 1]) / 2.0;
}
static std::unique_ptr<KdNode> construct(const std::vector<BBox>& object_boxes, std::vector<size_t> indices) {
    // src/kdtree.rs:238-348 (generic over T: Bounded; only the bounding boxes are needed)
    auto node = std::make_unique<KdNode>();
    if (indices.size() < 16) {
        node->indices = std::move(indices);
        return node;
    }
    std::vector<double> xs, ys, zs;
    std::vector<BBox> bboxs;
    for (size_t index : indices) {
        BBox b = object_boxes[index];
        xs.push_back(b.p_min.x); xs.push_back(b.p_max.x);
        ys.push_back(b.p_min.y); ys.push_back(b.p_max.y);
        zs.push_back(b.p_min.z); zs.push_back(b.p_max.z);
        bboxs.push_back(b);
    }
    std::sort(xs.begin(), xs.end());
    std::sort(ys.begin(), ys.end());
    std::sort(zs.begin(), zs.end());
    double med[3] = {median(xs), median(ys), median(zs)};
    auto partition_score = [&](int dim, double value) {
        size_t left = 0, right = 0;
        for (const BBox& b : bboxs) {
            if (b.p_min[dim] <= value) left++;
            if (b.p_max[dim] >= value) right++;
        }
        return std::max(left, right);
    };
    size_t s[3] = {partition_score(0, med[0]), partition_score(1, med[1]), partition_score(2, med[2])};
    size_t threshold = size_t(double(indices.size()) * 0.85);  // SCORE_THRESHOLD :9
    if (std::min(std::min(s[0], s[1]), s[2]) >= threshold) {
        node->indices = std::move(indices);
        return node;
    }
    int split_dir = -1;
    BBox bounds;
    for (const BBox& b : bboxs) bounds = bounds.merge(b);
    V3 extent = bounds.p_max - bounds.p_min;
    if (extent.x > extent.y && extent.x > extent.z) {
        if (s[0] < threshold) split_dir = 0;
    } else if (extent.y > extent.z) {
        if (s[1] < threshold) split_dir = 1;
    } else if (s[2] < threshold) {
        split_dir = 2;
    }
    if (split_dir == -1) {
        if (s[0] < s[1] && s[0] < s[2]) split_dir = 0;
        else if (s[1] < s[2]) split_dir = 1;
        else split_dir = 2;
    }
    std::vector<size_t> left, right;
    for (size_t i = 0; i < indices.size(); i++) {
        if (bboxs[i].p_min[split_dir] <= med[split_dir]) left.push_back(indices[i]);
        if (bboxs[i].p_max[split_dir] >= med[split_dir]) right.push_back(indices[i]);
    }
    node->axis = split_dir;
    node->value = med[split_dir];
    node->left = construct(object_boxes, std::move(left));
    node->right = construct(object_boxes, std::move(right));
    return node;
}

struct Mesh : Shape {  // KdTree<Triangle>, src/kdtree.rs:103-227, src/shape/mesh.rs:103
    std::vector<Triangle> objects;
    std::unique_ptr<KdNode> root;
    BBox bounds;
    explicit Mesh(std::vector<Triangle> tris) : objects(std::move(tris)) {  // :111-124
        std::vector<size_t> idx(objects.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
        std::vector<BBox> boxes;
        for (const Triangle& t : objects) { boxes.push_back(t.bounding_box()); bounds = bounds.merge(boxes.back()); }
        root = construct(boxes, std::move(idx));
    }
    bool bounding_box(BBox& b) const override { b = bounds; return true; }
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override {  // :132-139
        double b_min, b_max;
        bounds.intersect(ray, b_min, b_max);
        if (std::fmax(b_min, t_min) > std::fmin(b_max, rec.time)) return false;
        return intersect_subtree(*root, bounds, ray, t_min, rec);
    }
    bool intersect_subtree(const KdNode& node, const BBox& bbox, const Ray& ray, double t_min,
                           HitRecord& rec) const {  // :154-226
        CNT(nodes);
        double b_min, b_max;
        bbox.intersect(ray, b_min, b_max);
        if (node.axis < 0) {
            bool result = false;
            for (size_t index : node.indices)
                if (objects[index].intersect(ray, t_min, rec)) result = true;
            return result;
        }
        int ax = node.axis;
        double value = node.value;
        double t_split = (value - ray.origin[ax]) / ray.dir[ax];
        bool left_first = (ray.origin[ax] < value) || (ray.origin[ax] == value && ray.dir[ax] <= 0.0);
        BBox bl, br;
        bbox.split(ax, value, bl, br);
        const KdNode* first = left_first ? node.left.get() : node.right.get();
        const KdNode* second = left_first ? node.right.get() : node.left.get();
        const BBox& b0 = left_first ? bl : br;
        const BBox& b1 = left_first ? br : bl;
        if (t_split > std::fmin(b_max, rec.time) || t_split <= 0.0) {
            return intersect_subtree(*first, b0, ray, t_min, rec);
        } else if (t_split < std::fmax(b_min, t_min)) {
            return intersect_subtree(*second, b1, ray, t_min, rec);
        } else {
            bool h1 = intersect_subtree(*first, b0, ray, t_min, rec);
            if (h1 && rec.time < t_split) return true;
            bool h2 = intersect_subtree(*second, b1, ray, t_split, rec);
            return h1 || h2;
        }
    }
    bool intersect_brute(const Ray& ray, double t_min, HitRecord& rec) const {  // test hook (KAT 7)
        bool r = false;
        for (const Triangle& t : objects)
            if (t.intersect(ray, t_min, rec)) r = true;
        return r;
    }
    SurfSample sample(const V3&, Rng& rng) const override {  // :141-146
        size_t num = objects.size();
        uint32_t index = rng.index(uint32_t(num));
        SurfSample s = objects[index].sample(rng);
        s.p = s.p / double(num);
        return s;
    }
};

struct Transformed : Shape {  // src/shape.rs:102-152
    std::unique_ptr<Shape> shape;
    M4 transform, inverse_transform;
    M3 linear, normal_transform;
    double scale;
    Transformed(std::unique_ptr<Shape> s, const M4& t) : shape(std::move(s)), transform(t) {  // :112-125
        inverse_transform = inverse4(t);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) linear.m[i][j] = t.m[i][j];
        scale = det3(linear);
        normal_transform = transpose3(inverse3(linear));  // glm::inverse_transpose
    }
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override {  // :129-138
        Ray local = ray.apply_transform(inverse_transform);
        if (shape->intersect(local, t_min, rec)) {
            rec.normal = normalize(mul(normal_transform, rec.normal));
            return true;
        }
        return false;
    }
    SurfSample sample(const V3& target, Rng& rng) const override {  // :140-151
        V3 t = mul(inverse_transform, target, 1.0);
        SurfSample s = shape->sample(t, rng);
        V3 new_normal = normalize(mul(normal_transform, s.n));
        double height = dot(mul(linear, s.n), new_normal);
        double base = scale / height;
        return SurfSample{mul(transform, s.v, 1.0), new_normal, s.p / base};
    }
    bool can_sample() const override { return shape->can_sample(); }
    bool bounding_box(BBox& out) const override {  // :154-176: box of the 8 transformed corners
        BBox b;
        if (!shape->bounding_box(b)) return false;
        out = BBox();
        for (int k = 0; k < 8; k++) {
            V3 c((k & 4) ? b.p_max.x : b.p_min.x, (k & 2) ? b.p_max.y : b.p_min.y, (k & 1) ? b.p_max.z : b.p_min.z);
            V3 w = mul(transform, c, 1.0);
            out.p_min = vmin(out.p_min, w);
            out.p_max = vmax(out.p_max, w);
        }
        return true;
    }
};

// KdTree<Box<dyn Bounded>> (src/kdtree.rs:103-227 over arbitrary bounded shapes, e.g.
// examples/fractal_spheres.rs:45): same construction and traversal as Mesh, children are shapes.
struct ShapeKdTree : Shape {
    std::vector<std::unique_ptr<Shape>> objects;
    std::unique_ptr<KdNode> root;
    BBox bounds;
    explicit ShapeKdTree(std::vector<std::unique_ptr<Shape>> children) : objects(std::move(children)) {
        std::vector<size_t> idx(objects.size());
        std::vector<BBox> boxes(objects.size());
        for (size_t i = 0; i < idx.size(); i++) {
            idx[i] = i;
            objects[i]->bounding_box(boxes[i]);
            bounds = bounds.merge(boxes[i]);
        }
        root = construct(boxes, std::move(idx));
    }
    bool bounding_box(BBox& b) const override { b = bounds; return true; }
    bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override {
        double b_min, b_max;
        bounds.intersect(ray, b_min, b_max);
        if (std::fmax(b_min, t_min) > std::fmin(b_max, rec.time)) return false;
        return subtree(*root, bounds, ray, t_min, rec);
    }
    bool subtree(const KdNode& node, const BBox& bbox, const Ray& ray, double t_min, HitRecord& rec) const {
        CNT(nodes);
        double b_min, b_max;
        bbox.intersect(ray, b_min, b_max);
        if (node.axis < 0) {
            bool result = false;
            for (size_t index : node.indices)
                if (objects[index]->intersect(ray, t_min, rec)) result = true;
            return result;
        }
        int ax = node.axis;
        double value = node.value;
        double t_split = (value - ray.origin[ax]) / ray.dir[ax];
        bool left_first = (ray.origin[ax] < value) || (ray.origin[ax] == value && ray.dir[ax] <= 0.0);
        BBox bl, br;
        bbox.split(ax, value, bl, br);
        const KdNode* first = left_first ? node.left.get() : node.right.get();
        const KdNode* second = left_first ? node.right.get() : node.left.get();
        const BBox& b0 = left_first ? bl : br;
        const BBox& b1 = left_first ? br : bl;
        if (t_split > std::fmin(b_max, rec.time) || t_split <= 0.0) return subtree(*first, b0, ray, t_min, rec);
        if (t_split < std::fmax(b_min, t_min)) return subtree(*second, b1, ray, t_min, rec);
        bool h1 = subtree(*first, b0, ray, t_min, rec);
        if (h1 && rec.time < t_split) return true;
        bool h2 = subtree(*second, b1, ray, t_split, rec);
        return h1 || h2;
    }
    SurfSample sample(const V3& target, Rng& rng) const override {  // :141-146
        size_t num = objects.size();
        uint32_t index = rng.index(uint32_t(num));
        SurfSample s = objects[index]->sample(target, rng);
        s.p = s.p / double(num);
        return s;
    }
    bool can_sample() const override {
        for (const auto& o : objects)
            if (!o->can_sample()) return false;
        return true;
    }
};

// ------------------------------------------------------------------ color.rs
static V3 hex_color(uint32_t x) {  // src/color.rs:10-15
    double r = double((x >> 16) & 0xff) / 255.0, g = double((x >> 8) & 0xff) / 255.0, b = double(x & 0xff) / 255.0;
    return V3(std::pow(r, 2.2), std::pow(g, 2.2), std::pow(b, 2.2));
}
static void color_bytes(const V3& c, uint8_t out[3]) {  // src/color.rs:18-24 (`as u8` truncates, saturating)
    double v[3] = {c.x, c.y, c.z};
    for (int i = 0; i < 3; i++) {
        double t = std::pow(std::fmin(std::fmax(v[i], 0.0), 1.0), 1.0 / 2.2) * 255.0;
        out[i] = (t != t) ? 0 : (t <= 0.0 ? 0 : (t >= 255.0 ? 255 : uint8_t(t)));
    }
}

// ------------------------------------------------------------------ material.rs
enum MatKind { LAMBERTIAN = 0, PHONG = 1, MIRROR = 2, TRANSMISSIVE = 3 };
struct Material {  // src/material.rs:8-23
    int kind = LAMBERTIAN;
    V3 albedo = V3(0.5, 0.5, 0.5);
    double emittance_ = 0.0, shininess = 0.0, ior = 1.0;
    double emittance() const { return (kind == LAMBERTIAN || kind == PHONG) ? emittance_ : 0.0; }  // :100-106
    V3 color() const { return (kind == LAMBERTIAN || kind == PHONG) ? albedo : V3(0, 0, 0); }       // :107-113
};
// nalgebra 0.24 Rotation3::rotation_between / UnitQuaternion::rotation_between
// applied to a vector (published semantics; crate source not in the mount):
// normalise a and b; c = a x b; if |c| > f64::EPSILON rotate about c/|c| by
// acos(a.b); else if a.b < 0 -> None; else identity.  `ok` is false for None.
static V3 rotation_between_apply(V3 a, V3 b, V3 v, bool& ok) {
    ok = true;
    double la = length(a), lb = length(b);
    if (!(la > 0.0) || !(lb > 0.0)) return v;  // try_normalize failed -> identity
    V3 na = a / la, nb = b / lb;
    V3 c = cross(na, nb);
    double lc = length(c);
    double cosang = dot(na, nb);
    if (lc > 2.220446049250313e-16) {
        V3 k = c / lc;
        double ang = std::acos(cosang);
        double s = std::sin(ang), co = std::cos(ang);
        // Rodrigues (== from_axis_angle matrix applied to v)
        return co * v + s * cross(k, v) + (1.0 - co) * dot(k, v) * k;
    }
    if (cosang < 0.0) {
        ok = false;
        return v;
    }
    return v;
}
static V3 reflect_vec(V3 i, V3 n) { return i - 2.0 * dot(n, i) * n; }  // glm::reflect_vec

static double snell_solve(double ni, double nt, double c) {  // :144-146
    double r = ni / nt;
    return std::sqrt(1.0 - r * r * (1.0 - c * c));
}
static V3 refract_ray(double ni, double nt, V3 w_i, double ci, double ct, V3 n) {  // :148-157
    return (ni / nt) * (-w_i) + ((ni / nt) * ci - ct) * n;
}
static double schlick(double ni, double nt, double c) {  // :159-162
    double r0 = (ni - nt) / (ni + nt);
    r0 = r0 * r0;
    double m = 1.0 - c;
    return r0 + (1.0 - r0) * (m * m * m * m * m);
}
// Material::sample_f, src/material.rs:166-263.  Returns false for None.
static bool sample_f(const Material& m, const V3& normal, const V3& wo, Rng& rng, V3& wi, double& pdf) {
    switch (m.kind) {
        case LAMBERTIAN: {  // :173-198
            double r1 = rng.uniform(), r2 = rng.uniform();
            double phi = 2.0 * PI * r1;
            double theta = std::acos(std::sqrt(r2));
            pdf = std::cos(theta) / PI;
            V3 dir(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
            bool ok;
            V3 b = rotation_between_apply(V3(0, 1, 0), normal, dir, ok);
            if (!ok) b = rotation_between_apply(V3(0, 1, 0.00000001), normal, dir, ok);  // :186-194
            wi = normalize(b);
            return true;
        }
        case PHONG: {  // :199-220
            double r1 = rng.uniform(), r2 = rng.uniform();
            double phi = 2.0 * PI * r1;
            double theta = std::acos(std::pow(r2, 1.0 / (m.shininess + 1.0)));
            pdf = (m.shininess + 1.0) / (2.0 * PI) * std::pow(std::cos(theta), m.shininess);
            V3 dir(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
            V3 reflected = -reflect_vec(wo, normal);
            bool ok;
            V3 b = rotation_between_apply(V3(0, 1, 0), reflected, dir, ok);  // quat_rotation: None -> identity
            if (!ok) b = dir;
            wi = normalize(b);
            return true;
        }
        case MIRROR: {  // :221
            wi = -reflect_vec(wo, normalize(normal));
            pdf = 1.0;
            return true;
        }
        default: {  // TRANSMISSIVE :222-261
            bool inside = dot(normal, wo) < 0.0;
            V3 nn = inside ? -normal : normal;
            double cos_i = std::fmin(std::fmax(dot(wo, nn), 0.0), 1.0);
            double ni = inside ? m.ior : 1.0, nt = inside ? 1.0 : m.ior;
            double sr = std::fmin(std::fmax(schlick(ni, nt, cos_i), 0.0), 1.0);
            if (rng.uniform() < sr) {
                wi = -reflect_vec(wo, normal);
                pdf = 1.0;
                return true;
            }
            double cos_t = snell_solve(ni, nt, cos_i);
            if (cos_t != cos_t) return false;  // NaN -> total internal reflection -> None
            wi = refract_ray(ni, nt, wo, cos_i, cos_t, nn);
            pdf = 1.0;
            return true;
        }
    }
}
// Material::bsdf, src/material.rs:266-289
static V3 bsdf(const Material& m, const V3& normal, const V3& wo, const V3& wi) {
    double n_dot_wi = dot(normal, wi), n_dot_wo = dot(normal, wo);
    if (std::signbit(n_dot_wi) || std::signbit(n_dot_wo)) return V3(0, 0, 0);  // !is_sign_positive
    switch (m.kind) {
        case LAMBERTIAN: return (1.0 / PI) * m.albedo;
        case PHONG: {
            V3 normalization = m.albedo * ((m.shininess + 2.0) / (2.0 * PI));
            V3 reflected = -normalize(reflect_vec(wi, normal));
            double c = std::fmin(std::fmax(dot(reflected, wo), 0.0), 1.0);
            return normalization * std::pow(c, m.shininess);
        }
        default: return V3(1, 1, 1);
    }
}

// ------------------------------------------------------------------ medium.rs
enum MediumKind { HOMOGENEOUS_ISOTROPIC = 0, COLORED_GLOWING_FOG = 1 };
struct Medium {  // src/medium.rs:9-27,78-122 (closed set: the two constructors)
    int kind;
    double absorption_, scattering_;
    double absorption(const V3&) const { return absorption_; }
    double scattering(const V3&) const { return scattering_; }
    double extinction(const V3&) const { return absorption_ + scattering_; }
    double emission(const V3&) const { return kind == COLORED_GLOWING_FOG ? 10.0 : 0.0; }
    V3 color(const V3& x) const {
        if (kind == COLORED_GLOWING_FOG) return x.y > 250.0 ? hex_color(0xFF0000) : hex_color(0x0000FF);
        return hex_color(0xD2B48C);
    }
    // :85 `1.0 / (4.0 * pi)` vs :110 `1.0 / 4.0 * pi` (sic)
    double phase(const V3&, const V3&) const { return kind == COLORED_GLOWING_FOG ? (1.0 / 4.0 * PI) : 1.0 / (4.0 * PI); }
    void sample_ph(const V3&, Rng& rng, V3& wi, double& p) const {  // :86-94, :111-119
        double a = rng.range(-1.0, 1.0), b = rng.range(-1.0, 1.0), c = rng.range(-1.0, 1.0);
        wi = normalize(V3(a, b, c));
        p = phase(wi, wi);
    }
    double transmittence(const Ray& ray, double t_max) const {  // :126-130
        return std::exp(-(extinction(ray.origin) * t_max));
    }
    void sample_d(const Ray& ray, Rng& rng, double& dist, double& pdf, double& cdf) const {  // :133-146
        double random = rng.range(0.0, 1.0);
        double ext = extinction(ray.origin);
        dist = -std::log(random) / ext;
        double tr = transmittence(ray, dist);
        pdf = ext * tr;
        cdf = 1.0 - tr;
    }
};

// ------------------------------------------------------------------ object / light / camera / scene
struct Object {  // src/object.rs:10-31
    std::unique_ptr<Shape> shape;
    Material material;
};
enum LightKind { L_POINT = 0, L_AMBIENT = 1, L_DIRECTIONAL = 2, L_OBJECT = 3 };
struct Light {  // src/light.rs:7-47
    int kind;
    V3 color, vec;  // vec = location (Point) or direction (Directional)
    Object object;
    int twin = -1;  // oracle-only, used by the `robust` visibility mode
    void illuminate(const V3& world_pos, Rng& rng, V3& intensity, V3& wi, double& dist) const {
        switch (kind) {
            case L_AMBIENT: intensity = color; wi = V3(0, 0, 0); dist = 0.0; return;
            case L_POINT: {
                V3 disp = vec - world_pos;
                double len = length(disp);
                intensity = color / (len * len);
                wi = disp / len;
                dist = len;
                return;
            }
            case L_DIRECTIONAL: intensity = color; wi = -normalize(vec); dist = INF; return;
            default: {
                SurfSample s = object.shape->sample(world_pos, rng);
                V3 disp = s.v - world_pos;
                double len = length(disp);
                double cosine = std::fmax(-dot(disp, s.n), 0.0) / len;
                double surface_area = std::fmax(cosine, 0.0) / (len * len);
                intensity = object.material.color() * object.material.emittance() * surface_area / s.p;
                wi = disp / len;
                dist = len;
                return;
            }
        }
    }
};
struct Camera {  // src/camera.rs:9-82
    V3 eye, direction, up;
    double fov, aperture, focal_distance;
    Ray cast_ray(double x, double y, Rng& rng) const {  // :65-82
        double d = 1.0 / std::tan(fov / 2.0);
        V3 right = normalize(cross(direction, up));
        V3 origin = eye;
        V3 new_dir = d * direction + x * right + y * up;
        if (aperture > 0.0) {
            V3 focal_point = origin + normalize(new_dir) * focal_distance;
            double dx, dy;
            rng.unit_disc(dx, dy);
            origin = origin + (dx * right + dy * up) * aperture;
            new_dir = focal_point - origin;
        }
        return Ray{origin, normalize(new_dir)};
    }
};
struct Scene {  // src/scene.rs:12-24
    std::vector<Object> objects;
    std::vector<Light> lights;
    std::vector<Medium> media;
    V3 environment = V3(0, 0, 0);  // Environment::Color (src/environment.rs:64-77)
    // Environment::Hdri (src/environment.rs:3-52): equirectangular image, bilinear lookup
    uint32_t hdri_w = 0, hdri_h = 0;
    std::vector<V3> hdri;
    V3 env_color(const V3& dir_in) const {
        if (hdri_w == 0) return environment;
        V3 dir = normalize(dir_in);
        double azimuth = std::atan2(dir.z, dir.x) + PI;
        double polar = std::acos(dir.y);
        double x = azimuth / (2.0 * PI) * double(hdri_w - 1);
        double y = polar / PI * double(hdri_h - 1);
        uint32_t x0 = std::min(uint32_t(x), hdri_w - 1), y0 = std::min(uint32_t(y), hdri_h - 1);
        double ax = x - double(x0), ay = y - double(y0);
        // the reference indexes x0+1 / y0+1 unclamped (it panics or wraps exactly where ax or ay is 0)
        uint32_t x1 = std::min(x0 + 1, hdri_w - 1), y1 = std::min(y0 + 1, hdri_h - 1);
        auto mix = [](const V3& a, const V3& b, double t) { return a * (1.0 - t) + b * t; };
        return mix(mix(hdri[size_t(y0) * hdri_w + x0], hdri[size_t(y0) * hdri_w + x1], ax),
                   mix(hdri[size_t(y1) * hdri_w + x0], hdri[size_t(y1) * hdri_w + x1], ax), ay);
    }
};

// ------------------------------------------------------------------ renderer.rs
static const double EPSILON = 1e-12;       // src/renderer.rs:17
static const double FIREFLY_CLAMP = 100.0;  // src/renderer.rs:18

struct RenderParams {
    uint32_t width, height;
    double exposure_value;
    uint32_t max_bounces;
    // oracle-only analysis switch: 0 = literal reference semantics; 1 = "robust"
    // (the semantic equivalents the fp32 HIP path uses: t_min scaled to the ray
    // origin, shadow test by light-twin object id + relative distance tolerance).
    int robust;
};

struct Renderer {
    const Scene& scene;
    Camera camera;
    RenderParams p;

    double t_min_for(const Ray& ray) const {
        if (!p.robust) return EPSILON;
        double m = std::fmax(std::fmax(std::fabs(ray.origin.x), std::fabs(ray.origin.y)), std::fabs(ray.origin.z));
        return 2e-5 * (1.0 + m);
    }
    // src/renderer.rs:416-425
    bool get_closest_hit(const Ray& ray, HitRecord& h, int& obj) const {
        CNT(rays);
        obj = -1;
        double tm = t_min_for(ray);
        for (size_t i = 0; i < scene.objects.size(); i++) {
            CNT(obj_tests);
            if (scene.objects[i].shape->intersect(ray, tm, h)) obj = int(i);
        }
        if (obj >= 0) {
            CNT(hits);
            if (tl_cnt) {
                double m = std::fmax(std::fmax(std::fabs(ray.origin.x), std::fabs(ray.origin.y)), std::fabs(ray.origin.z));
                if (h.time < 1e-9 * (1.0 + m)) tl_cnt->self_hits++;
            }
        }
        return obj >= 0;
    }
    bool shadow_visible(const Light& light, const V3& pos, const V3& wi, double dist_to_light) const {
        Ray ray{pos, wi};
        HitRecord h;
        int obj;
        CNT(shadow_tests);
        if (!get_closest_hit(ray, h, obj)) return false;
        bool pass;
        if (!p.robust) {
            pass = std::fabs(h.time - dist_to_light) < EPSILON;  // src/renderer.rs:348,396
            if (!pass && tl_cnt && std::fabs(h.time - dist_to_light) < 1e-6 * dist_to_light) tl_cnt->shadow_near++;
        } else {
            pass = (obj == light.twin) && std::fabs(h.time - dist_to_light) <= 1e-3 * dist_to_light;
        }
        if (pass) CNT(shadow_pass);
        return pass;
    }
    // src/renderer.rs:362-409
    V3 sample_lights(const Material& material, const V3& pos, const V3& n, const V3& wo, Rng& rng) const {
        V3 color(0, 0, 0);
        for (const Light& light : scene.lights) {
            if (light.kind == L_AMBIENT) {
                color = color + cmul(light.color, material.color());
            } else {
                V3 intensity, wi;
                double dist;
                light.illuminate(pos, rng, intensity, wi, dist);
                if (shadow_visible(light, pos, wi, dist)) {
                    V3 f = bsdf(material, n, wo, wi);
                    color = color + cmul(f, intensity) * dot(wi, n);
                }
            }
        }
        return color;
    }
    // src/renderer.rs:325-359
    V3 sample_lights_for_media(const Medium& medium, const V3& pos, const V3& wo, Rng& rng) const {
        V3 color(0, 0, 0);
        double scat = medium.scattering(pos), ext = medium.extinction(pos);
        V3 medium_color = medium.color(pos);
        for (const Light& light : scene.lights) {
            if (light.kind == L_AMBIENT) {
                color = color + cmul(light.color, medium_color);
            } else {
                V3 intensity, wi;
                double dist;
                light.illuminate(pos, rng, intensity, wi, dist);
                if (shadow_visible(light, pos, wi, dist)) {
                    double ph = medium.phase(wo, wi);
                    color = color + (scat / ext) * cmul(intensity, medium_color) * ph;
                }
            }
        }
        return color;
    }
    // src/renderer.rs:187-322 (recursive, exactly as written)
    V3 trace_ray(const Ray& ray, uint32_t num_bounces, Rng& rng) const {
        CNT(vertices);
        if (!scene.media.empty()) {
            const Medium& medium = scene.media[0];
            const double rr_p = 0.8;
            double d, pdf_d, cdf_d;
            medium.sample_d(ray, rng, d, pdf_d, cdf_d);
            V3 wo = -normalize(ray.dir);
            double max_dist;
            V3 surface_color(0, 0, 0);
            HitRecord h;
            int oi;
            if (!get_closest_hit(ray, h, oi)) {
                const double background_dist = 400.0;
                surface_color = (d >= background_dist) ? scene.env_color(ray.dir) : V3(0, 0, 0);
                max_dist = background_dist;
            } else {
                if (d >= h.time) {
                    V3 world_pos = ray.at(h.time);
                    const Material& material = scene.objects[oi].material;
                    V3 color = (num_bounces == 0) ? material.color() * material.emittance() : V3(0, 0, 0);
                    color = color + sample_lights(material, world_pos, h.normal, wo, rng);
                    if (rng.uniform() < rr_p) {
                        V3 wi;
                        double pdf;
                        if (sample_f(material, h.normal, wo, rng, wi, pdf)) {
                            V3 f = bsdf(material, h.normal, wo, wi);
                            Ray nr{world_pos, wi};
                            V3 indirect = (1.0 / (pdf * rr_p)) * cmul(f, trace_ray(nr, num_bounces + 1, rng)) *
                                          std::fabs(dot(wi, h.normal));
                            color = color + indirect;
                        }
                    }
                    surface_color = color;
                }
                max_dist = h.time;
            }
            if (d < max_dist) {
                V3 collision = ray.at(d);
                double abs_ = medium.absorption(collision);
                double emm = medium.emission(collision);
                V3 medium_color = medium.color(collision);
                double scat = medium.scattering(collision);
                double extinction = abs_ + scat;
                V3 color = (num_bounces == 0) ? emm * medium_color : V3(0, 0, 0);
                color = color + sample_lights_for_media(medium, collision, wo, rng);
                if (rng.uniform() < rr_p) {
                    V3 wi;
                    double ph_p;
                    medium.sample_ph(wo, rng, wi, ph_p);
                    Ray nr{collision, wi};
                    V3 indirect = (scat / extinction) * trace_ray(nr, num_bounces + 1, rng);
                    indirect = indirect / ph_p;
                    indirect = cmul(indirect, medium_color) * medium.phase(wo, wi);
                    indirect = indirect / rr_p;
                    color = color + indirect;
                }
                return color;
            }
            return surface_color;
        }
        HitRecord h;
        int oi;
        if (!get_closest_hit(ray, h, oi)) return scene.env_color(ray.dir);
        V3 world_pos = ray.at(h.time);
        const Material& material = scene.objects[oi].material;
        V3 wo = -normalize(ray.dir);
        V3 color = (num_bounces == 0) ? material.color() * material.emittance() : V3(0, 0, 0);
        color = color + sample_lights(material, world_pos, h.normal, wo, rng);
        if (num_bounces < p.max_bounces) {
            V3 wi;
            double pdf;
            if (sample_f(material, h.normal, wo, rng, wi, pdf)) {
                V3 f = bsdf(material, h.normal, wo, wi);
                Ray nr{world_pos, wi};
                V3 indirect = (1.0 / pdf) * cmul(f, trace_ray(nr, num_bounces + 1, rng)) * std::fabs(dot(wi, h.normal));
                // f64::min returns the non-NaN operand (src/renderer.rs:311-313)
                color.x += std::fmin(indirect.x, FIREFLY_CLAMP);
                color.y += std::fmin(indirect.y, FIREFLY_CLAMP);
                color.z += std::fmin(indirect.z, FIREFLY_CLAMP);
            }
        }
        return color;
    }
    // src/renderer.rs:173-184; the RNG stream is per (pixel, sample) instead of per row.
    V3 get_color(uint32_t x, uint32_t y, uint32_t iterations, uint64_t seed, uint32_t sample_offset) const {
        double dim = double(std::max(p.width, p.height));
        double xn = (double(2 * x + 1) - double(p.width)) / dim;
        double yn = (double(2 * (p.height - y) - 1) - double(p.height)) / dim;
        V3 color(0, 0, 0);
        for (uint32_t s = 0; s < iterations; s++) {
            Rng rng(seed, y * p.width + x, sample_offset + s);
            double dx = rng.range(-1.0 / dim, 1.0 / dim);
            double dy = rng.range(-1.0 / dim, 1.0 / dim);
            CNT(samples);
            color = color + trace_ray(camera.cast_ray(xn + dx, yn + dy, rng), 0, rng);
        }
        return color / double(iterations) * std::pow(2.0, p.exposure_value);
    }
};


// ====================================================================== photon.rs (next tier)
// Photon shooting (src/photon.rs:724-946), the point map for the beam estimate (:204-247), the
// surface estimate (:327-375), the beam x point volume estimate (:439-502) and the point x point
// volume estimate (:384-438), combined as in :595-628.  Third-party crates restated from their
// published semantics: kd-tree 0.4.1 `nearests(q,k)` = the k items of least squared distance;
// bvh 0.6 `traverse(ray)` = every shape whose AABB the half-infinite ray hits (a superset that the
// per-photon distance test below filters, so any conservative traversal yields the same sum).
struct Photon {  // src/photon.rs:24-34
    V3 position, direction, power, starting_position;
};
struct PhotonList {  // src/photon.rs:141-147
    std::vector<Photon> surface, volume;
};
enum PhotonKind { PK_PHOTON_MAP = 0, PK_POINT_BEAM = 1, PK_BEAM_BEAM = 2 };  // src/photon.rs:631-639

// trace_photon, src/photon.rs:803-946.  The recursion is a tail call (nothing but list
// concatenation follows it), so it is a loop here; photons are stored in path order.
static void trace_photon(const Renderer& r, Ray ray, V3 power, Rng& rng, PhotonList& out) {
    const Scene& scene = r.scene;
    for (;;) {
        V3 wo = -normalize(ray.dir);
        HitRecord h;
        int oi;
        bool hit = r.get_closest_hit(ray, h, oi);
        bool in_volume = false;
        double d = 0.0;
        const Medium* medium = scene.media.empty() ? nullptr : &scene.media[0];
        if (medium) {  // :918-944
            double pdf_d, cdf_d;
            medium->sample_d(ray, rng, d, pdf_d, cdf_d);
            in_volume = !hit || d < h.time;
        } else if (!hit) {
            return;
        }
        if (in_volume) {  // trace_in_volume :879-914
            V3 collision = ray.at(d);
            V3 medium_color = medium->color(collision);
            double scat = medium->scattering(collision), extinction = medium->absorption(collision) + scat;
            V3 attenuated = cmul(power, medium_color) * scat / extinction;
            double rr_prob = scat / extinction;
            out.volume.push_back(Photon{collision, wo, power, ray.origin});
            if (rng.uniform() < rr_prob) {
                V3 wi;
                double ph_p;
                medium->sample_ph(wo, rng, wi, ph_p);
                power = attenuated * medium->phase(wo, wi) / ph_p;
                ray = Ray{collision, wi};
                continue;
            }
            return;
        }
        // trace_on_surface :811-875: p_d = 0.7 (diffuse (0.7,0.7,0.7), no specular)
        V3 world_pos = ray.at(h.time);
        const Material& material = scene.objects[oi].material;
        const double p_d = (0.7 + 0.7 + 0.7) / (0.7 + 0.7 + 0.7 + 0.0) * 0.7;
        if (!(rng.uniform() < p_d)) return;  // absorbed: no photon is stored
        V3 wi;
        double pdf;
        if (!sample_f(material, h.normal, wo, rng, wi, pdf)) return;  // total internal reflection: no photons
        V3 f = bsdf(material, h.normal, wo, wi);
        double cosine_term = dot(wi, h.normal) > 0.0 ? dot(wi, h.normal) : 1.0;
        V3 attenuated = cmul(power, f) * cosine_term / pdf / p_d;
        bool is_mirror = material.kind == MIRROR || material.kind == TRANSMISSIVE;  // material.rs:135-141
        if (!is_mirror) out.surface.push_back(Photon{world_pos, wo, power, world_pos});  // `ray` is shadowed by the new ray at :842-846
        power = attenuated;
        ray = Ray{world_pos, wi};
    }
}
// shoot_photon, src/photon.rs:724-799 (first Light::Object only, as written)
static bool shoot_photon(const Renderer& r, double power, Rng& rng, Rng& thin, int kind, PhotonList& out) {
    for (const Light& light : r.scene.lights) {
        if (light.kind != L_OBJECT) continue;
        SurfSample s = light.object.shape->sample(V3(0, 0, 0), rng);
        double phi = 2.0 * PI * rng.uniform();
        double theta = std::acos(1.0 - rng.uniform());
        V3 dir(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
        bool ok;
        V3 direction = rotation_between_apply(V3(0, 1, 0), s.n, dir, ok);
        if (!ok) direction = rotation_between_apply(V3(0, 1, 0.00000001), s.n, dir, ok);
        PhotonList mine;
        trace_photon(r, Ray{s.v, direction}, power * light.object.material.color(), rng, mine);
        for (const Photon& p : mine.surface) out.surface.push_back(p);
        for (Photon p : mine.volume) {
            if (kind == PK_BEAM_BEAM) {  // :779-787 thinning (iid Bernoulli draws; taken from a side stream in path order)
                const double thresh = 0.001;
                if (thin.uniform() < thresh) {
                    p.power = p.power / thresh;
                    out.volume.push_back(p);
                }
            } else {
                out.volume.push_back(p);
            }
        }
        return true;
    }
    return false;  // panic!("Only found non-object lights while photon mapping")
}

// kd-tree over points: only nearests(q, k) is needed.
struct PointKd {
    std::vector<V3> pts;
    std::vector<uint32_t> idx;   // permutation: tree order -> original index
    void build(const std::vector<Photon>& ph) {
        pts.resize(ph.size());
        idx.resize(ph.size());
        for (size_t i = 0; i < ph.size(); i++) { pts[i] = ph[i].position; idx[i] = uint32_t(i); }
        if (!ph.empty()) rec(0, ph.size(), 0);
    }
    void rec(size_t lo, size_t hi, int axis) {
        if (hi - lo <= 1) return;
        size_t mid = (lo + hi) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                         [&](uint32_t a, uint32_t b) { return pts[a][axis] < pts[b][axis]; });
        rec(lo, mid, (axis + 1) % 3);
        rec(mid + 1, hi, (axis + 1) % 3);
    }
    // max-heap of (dist2, index) of size <= k
    void nearests(const V3& q, size_t k, std::vector<std::pair<double, uint32_t>>& heap) const {
        heap.clear();
        if (!idx.empty() && k) search(0, idx.size(), 0, q, k, heap);
        std::sort_heap(heap.begin(), heap.end());
    }
    void search(size_t lo, size_t hi, int axis, const V3& q, size_t k,
                std::vector<std::pair<double, uint32_t>>& heap) const {
        if (lo >= hi) return;
        size_t mid = (lo + hi) / 2;
        uint32_t id = idx[mid];
        V3 dv = pts[id] - q;
        double d2 = dot(dv, dv);
        if (heap.size() < k) { heap.emplace_back(d2, id); std::push_heap(heap.begin(), heap.end()); }
        else if (d2 < heap.front().first) { std::pop_heap(heap.begin(), heap.end()); heap.back() = {d2, id}; std::push_heap(heap.begin(), heap.end()); }
        double delta = q[axis] - pts[id][axis];
        int na = (axis + 1) % 3;
        if (delta < 0) {
            search(lo, mid, na, q, k, heap);
            if (heap.size() < k || delta * delta <= heap.front().first) search(mid + 1, hi, na, q, k, heap);
        } else {
            search(mid + 1, hi, na, q, k, heap);
            if (heap.size() < k || delta * delta <= heap.front().first) search(lo, mid, na, q, k, heap);
        }
    }
};
// Median-split BVH over item boxes (photon spheres or photon beams); visit() is called for every
// item whose own AABB the half-infinite ray hits: the candidate set of bvh 0.6 `traverse`.
struct SphereBvh {
    struct Node { V3 lo, hi; uint32_t left, right, first, count; };
    std::vector<Node> nodes;
    std::vector<uint32_t> order;
    std::vector<V3> ilo, ihi;
    void build(const std::vector<V3>& p, const std::vector<double>& r) {
        std::vector<V3> lo(p.size()), hi(p.size());
        for (size_t i = 0; i < p.size(); i++) { lo[i] = p[i] - V3(r[i], r[i], r[i]); hi[i] = p[i] + V3(r[i], r[i], r[i]); }
        build_boxes(lo, hi);
    }
    void build_boxes(const std::vector<V3>& lo, const std::vector<V3>& hi) {
        ilo = lo; ihi = hi;
        order.resize(lo.size());
        for (size_t i = 0; i < lo.size(); i++) order[i] = uint32_t(i);
        nodes.clear();
        if (!lo.empty()) rec(0, uint32_t(lo.size()));
    }
    V3 centre(uint32_t i) const { return 0.5 * (ilo[i] + ihi[i]); }
    uint32_t rec(uint32_t first, uint32_t count) {
        uint32_t me = uint32_t(nodes.size());
        nodes.push_back(Node{});
        V3 lo(INF, INF, INF), hi(-INF, -INF, -INF), clo(INF, INF, INF), chi(-INF, -INF, -INF);
        for (uint32_t i = first; i < first + count; i++) {
            lo = vmin(lo, ilo[order[i]]); hi = vmax(hi, ihi[order[i]]);
            V3 c = centre(order[i]);
            clo = vmin(clo, c); chi = vmax(chi, c);
        }
        nodes[me].lo = lo; nodes[me].hi = hi; nodes[me].first = first; nodes[me].count = count;
        nodes[me].left = nodes[me].right = 0;
        if (count > 8) {
            V3 e = chi - clo;
            int ax = (e.x > e.y && e.x > e.z) ? 0 : (e.y > e.z ? 1 : 2);
            uint32_t mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return centre(a)[ax] < centre(b)[ax]; });
            uint32_t l = rec(first, mid - first), rr = rec(mid, first + count - mid);
            nodes[me].left = l; nodes[me].right = rr; nodes[me].count = 0;
        }
        return me;
    }
    static bool box_hit(const V3& lo, const V3& hi, const Ray& ray) {
        BBox b; b.p_min = lo; b.p_max = hi;
        double t0, t1;
        b.intersect(ray, t0, t1);
        return t1 >= std::fmax(t0, 0.0);
    }
    template <class F>
    void traverse(const Ray& ray, F&& visit) const {
        if (nodes.empty()) return;
        uint32_t stack[64];
        int sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const Node& n = nodes[stack[--sp]];
            if (!box_hit(n.lo, n.hi, ray)) continue;
            if (n.count) {
                for (uint32_t i = n.first; i < n.first + n.count; i++)
                    if (box_hit(ilo[order[i]], ihi[order[i]], ray)) visit(order[i]);
            } else { stack[sp++] = n.left; stack[sp++] = n.right; }
        }
    }
};

struct PhotonMap {  // src/photon.rs:181-311
    int kind = PK_POINT_BEAM;
    PhotonList list;
    PointKd surface_kd, volume_kd;
    std::vector<V3> sphere_pos;
    std::vector<double> sphere_radius;  // PointMapForBeamEstimate: distance to the 10th nearest volume photon
    SphereBvh bvh;
    void build() {
        surface_kd.build(list.surface);
        volume_kd.build(list.volume);
        if (kind == PK_BEAM_BEAM) {  // :250-305: fixed radius 3, beam = segment from the previous vertex
            size_t n = list.volume.size();
            std::vector<V3> lo(n), hi(n);
            sphere_radius.assign(n, 3.0);
            for (size_t i = 0; i < n; i++) {  // impl Bounded for PhotonBeam, :74-104
                V3 a = list.volume[i].starting_position, b = list.volume[i].position;
                double cx = (a.x - b.x) * (a.x - b.x), cy = (a.y - b.y) * (a.y - b.y), cz = (a.z - b.z) * (a.z - b.z);
                double sum = cx + cy + cz;
                V3 adj(std::sqrt((cy + cz) / sum) * 3.0, std::sqrt((cx + cz) / sum) * 3.0, std::sqrt((cx + cy) / sum) * 3.0);
                lo[i] = vmin(a, b) - adj;
                hi[i] = vmax(a, b) + adj;
            }
            bvh.build_boxes(lo, hi);
        }
        if (kind == PK_POINT_BEAM) {  // :204-247
            size_t n = list.volume.size();
            sphere_pos.resize(n);
            sphere_radius.resize(n);
            std::vector<std::pair<double, uint32_t>> heap;
            for (size_t i = 0; i < n; i++) {
                volume_kd.nearests(list.volume[i].position, 10, heap);
                double m = -1.0;
                for (auto& e : heap) m = std::fmax(m, e.first);
                sphere_pos[i] = list.volume[i].position;
                sphere_radius[i] = std::sqrt(m);
            }
            bvh.build(sphere_pos, sphere_radius);
        }
    }
};

struct PhotonParams {
    uint64_t photon_count;
    int kind;
    double watts;
    uint64_t gather_size, gather_size_volume;
};

// surface_estimate closure, src/photon.rs:327-375
static V3 photon_surface_estimate(const Renderer& r, const PhotonMap& pm, const PhotonParams& pp, const Ray& ray,
                                  const HitRecord& h, const Material& material, const V3& wo) {
    V3 world_pos = ray.at(h.time);
    std::vector<std::pair<double, uint32_t>> near;
    pm.surface_kd.nearests(world_pos, pp.gather_size, near);
    double max_dist_squared = 0.0;
    for (auto& e : near) max_dist_squared = std::fmax(max_dist_squared, e.first);
    V3 color = material.emittance() * material.color();
    for (auto& e : near) {
        const Photon& photon = pm.list.surface[e.second];
        V3 disp = world_pos - photon.position;
        Ray pr{photon.position, normalize(disp)};
        HitRecord ph;
        int oi;
        if (r.get_closest_hit(pr, ph, oi)) {
            double len = length(disp);
            bool blocked;
            if (r.p.robust) {  // policy of the fp32 path: own-tangent-plane hits are not occluders
                V3 hp = pr.at(ph.time) - world_pos;
                bool own_plane = std::fabs(dot(hp, h.normal)) <= 1e-4 * len;
                blocked = !own_plane && ph.time < len * (1.0 - 1e-3);
            } else {
                blocked = len > ph.time;  // :357-361
            }
            if (blocked) continue;
        }
        double c = std::fmin(std::fmax(dot(photon.direction, h.normal), 0.0), 1.0);
        color = color + cmul(bsdf(material, h.normal, wo, photon.direction), photon.power) * c;
    }
    return color * (1.0 / (PI * max_dist_squared));
}
// PhotonMap::estimate_indirect, src/photon.rs:316-628
static V3 photon_estimate_indirect(const Renderer& r, const PhotonMap& pm, const PhotonParams& pp, const Ray& ray,
                                   Rng& rng) {
    const Scene& scene = r.scene;
    V3 wo = -normalize(ray.dir);
    const Medium* medium = scene.media.empty() ? nullptr : &scene.media[0];
    HitRecord h;
    int oi;
    bool hit = r.get_closest_hit(ray, h, oi);
    auto volume_beam = [&](const HitRecord* hp) {  // :439-502
        V3 dummy(0, 0, 0);
        V3 medium_color = medium->color(dummy);
        double extinction = medium->absorption(dummy) + medium->scattering(dummy);
        V3 volume_color(0, 0, 0);
        pm.bvh.traverse(ray, [&](uint32_t i) {
            const Photon& photon = pm.list.volume[i];
            double radius = pm.sphere_radius[i];
            V3 otc = photon.position - ray.origin;
            if (hp && length(otc) > hp->time) return;
            double radius_squared = radius * radius;
            double disk_distance = dot(otc, ray.dir);
            V3 dv = ray.at(disk_distance) - photon.position;
            double distance_squared = dot(dv, dv);
            if (disk_distance > 0.0 && distance_squared < radius_squared) {
                double tmp = 1.0 - distance_squared / radius_squared;
                double weight = (3.0 / PI) * tmp * tmp / radius_squared;
                V3 wi = -photon.direction;
                double transmittance = std::exp(-extinction * disk_distance);
                volume_color = volume_color + transmittance * cmul(photon.power, medium_color) * medium->phase(wi, -ray.dir) * weight;
            }
        });
        return volume_color;
    };
    auto volume_beam_beam = [&](const HitRecord* hp) {  // :503-593
        V3 dummy(0, 0, 0);
        V3 medium_color = medium->color(dummy);
        double extinction = medium->extinction(dummy);
        V3 volume_color(0, 0, 0);
        pm.bvh.traverse(ray, [&](uint32_t i) {
            const Photon& ph = pm.list.volume[i];
            const double radius = pm.sphere_radius[i];
            V3 bstart = ph.starting_position, bend = ph.position;
            V3 bdir = normalize(bend - bstart);
            V3 l = bstart - ray.origin;
            V3 u = normalize(cross(l, bdir));
            V3 n = normalize(cross(bdir, u));
            double t = dot(n, l) / dot(n, ray.dir);
            V3 query_collision = ray.at(t);
            if (hp && t >= hp->time) return;
            double dd = dot(ray.dir, bdir);
            double inv_sin_theta = 1.0 / std::sqrt(std::fmax(0.0, 1.0 - dd * dd));
            double beam_t = dot(bdir, query_collision - bstart);
            double beam_len = length(bend - bstart);
            if (beam_t < 0.0 || beam_t > beam_len) return;
            V3 beam_collision = bstart + beam_t * bdir;
            double dist = length(query_collision - beam_collision);
            if (dist >= radius) return;
            double tmp = 1.0 - dist / radius;
            double k2 = (3.0 / PI) * tmp * tmp;
            V3 color = extinction * cmul(ph.power, medium_color) * medium->phase(-bdir, -ray.dir) * inv_sin_theta *
                       std::exp(-extinction * t) * std::exp(-extinction * beam_t) * k2 / (2.0 * radius);
            volume_color = volume_color + color;
        });
        (void)rng.uniform();  // the debug print's draw, src/photon.rs:587
        return volume_color;
    };
    auto volume_point = [&](const HitRecord* hp, const Material* material) {  // :384-438
        double d, d_pdf, d_cdf;
        medium->sample_d(ray, rng, d, d_pdf, d_cdf);
        if (!hp || d < hp->time) {
            V3 collision = ray.at(d);
            V3 medium_color = medium->color(collision);
            double extinction = medium->absorption(collision) + medium->scattering(collision);
            V3 color(0, 0, 0);
            std::vector<std::pair<double, uint32_t>> near;
            pm.volume_kd.nearests(collision, pp.gather_size_volume, near);
            double max_dist_squared = 0.0;
            for (auto& e : near) max_dist_squared = std::fmax(max_dist_squared, e.first);
            for (auto& e : near)
                color = color + cmul(pm.list.volume[e.second].power, medium_color) * medium->phase(wo, pm.list.volume[e.second].direction);
            color = color / ((4.0 / 3.0) * PI * std::pow(max_dist_squared, 1.5));
            color = color / extinction;
            color = color * medium->transmittence(ray, d);
            color = color / d_pdf;
            return color;
        }
        return photon_surface_estimate(r, pm, pp, ray, *hp, *material, wo) * medium->transmittence(ray, hp->time) / (1.0 - d_cdf);
    };
    if (!hit) {
        if (!medium) return scene.env_color(ray.dir);
        if (pm.kind == PK_BEAM_BEAM) return volume_beam_beam(nullptr);
        return pm.kind == PK_PHOTON_MAP ? volume_point(nullptr, nullptr) : volume_beam(nullptr);
    }
    const Material& material = scene.objects[oi].material;
    if (!medium) return photon_surface_estimate(r, pm, pp, ray, h, material, wo);
    if (pm.kind == PK_PHOTON_MAP) return volume_point(&h, &material);
    V3 volume_color = pm.kind == PK_BEAM_BEAM ? volume_beam_beam(&h) : volume_beam(&h);
    V3 surface_color = photon_surface_estimate(r, pm, pp, ray, h, material, wo) * medium->transmittence(ray, h.time);
    (void)rng.uniform();  // the debug print's draw, src/photon.rs:619
    return surface_color + volume_color;
}

}  // namespace orc

// ====================================================================== C API (ctypes)
using namespace orc;

extern "C" {

struct orc_shape_desc {
    int32_t kind;           // 0 sphere, 1 cube, 2 plane, 3 mesh, 4 kd-tree of bounded shapes
    int32_t has_transform;  // 0: bare shape, 1: Transformed<shape>
    double transform[16];   // row-major 4x4
    double plane_normal[3];
    double plane_value;
    const double* tris;     // n_tris * 18 doubles: v1 v2 v3 n1 n2 n3
    uint64_t n_tris;
    const struct orc_shape_desc* children;  // kind 4: KdTree<Box<dyn Bounded>> of these shapes
    uint64_t n_children;
};
struct orc_material {
    int32_t kind;
    int32_t _pad;
    double albedo[3];
    double emittance, shininess, ior;
};
struct orc_camera {
    double eye[3], direction[3], up[3];
    double fov, aperture, focal_distance;
};
struct orc_params {
    uint32_t width, height;
    double exposure_value;
    uint32_t max_bounces;
    int32_t robust;
};
struct orc_counters {
    uint64_t rays, obj_tests, nodes, tri_tests, hits, samples, vertices, self_hits, shadow_tests, shadow_pass, shadow_near;
};

static V3 v3(const double* p) { return V3(p[0], p[1], p[2]); }

static std::unique_ptr<Shape> make_shape(const orc_shape_desc* d) {
    std::unique_ptr<Shape> s;
    switch (d->kind) {
        case 0: s.reset(new Sphere()); break;
        case 1: s.reset(new Cube()); break;
        case 2: {
            Plane* p = new Plane();
            p->normal = v3(d->plane_normal);
            p->value = d->plane_value;
            s.reset(p);
            break;
        }
        case 3: {
            std::vector<Triangle> tris(d->n_tris);
            for (uint64_t i = 0; i < d->n_tris; i++) {
                const double* t = d->tris + i * 18;
                tris[i] = Triangle{v3(t), v3(t + 3), v3(t + 6), v3(t + 9), v3(t + 12), v3(t + 15)};
            }
            s.reset(new Mesh(std::move(tris)));
            break;
        }
        case 4: {
            std::vector<std::unique_ptr<Shape>> kids;
            for (uint64_t i = 0; i < d->n_children; i++) {
                auto c = make_shape(d->children + i);
                BBox b;
                if (!c || !c->bounding_box(b)) return nullptr;  // only Bounded shapes (no planes)
                kids.push_back(std::move(c));
            }
            if (kids.empty()) return nullptr;
            s.reset(new ShapeKdTree(std::move(kids)));
            break;
        }
        default: return nullptr;
    }
    if (d->has_transform) {
        M4 m;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m.m[i][j] = d->transform[i * 4 + j];
        s.reset(new Transformed(std::move(s), m));
    }
    return s;
}
static Material make_material(const orc_material* m) {
    Material r;
    r.kind = m->kind;
    r.albedo = v3(m->albedo);
    r.emittance_ = m->emittance;
    r.shininess = m->shininess;
    r.ior = m->ior;
    return r;
}

struct orc_scene {
    Scene scene;
};

orc_scene* orc_scene_new() { return new orc_scene(); }
void orc_scene_free(orc_scene* s) { delete s; }
int orc_add_object(orc_scene* s, const orc_shape_desc* d, const orc_material* m) {
    auto sh = make_shape(d);
    if (!sh) return -1;
    Object o;
    o.shape = std::move(sh);
    o.material = make_material(m);
    s->scene.objects.push_back(std::move(o));
    return int(s->scene.objects.size()) - 1;
}
// kind: LightKind; color/vec for point/ambient/directional; shape+material for object lights;
// twin = index of the identical scene object (or -1), used only by the robust mode.
int orc_add_light(orc_scene* s, int kind, const double* color, const double* vec, const orc_shape_desc* d,
                  const orc_material* m, int twin) {
    Light l;
    l.kind = kind;
    if (color) l.color = v3(color);
    if (vec) l.vec = v3(vec);
    if (kind == L_OBJECT) {
        auto sh = make_shape(d);
        if (!sh || !sh->can_sample()) return -1;  // Plane::sample is unimplemented!() in the reference
        l.object.shape = std::move(sh);
        l.object.material = make_material(m);
        l.twin = twin;
    }
    s->scene.lights.push_back(std::move(l));
    return 0;
}
int orc_add_medium(orc_scene* s, int kind, double absorption, double scattering) {
    s->scene.media.push_back(Medium{kind, absorption, scattering});
    return 0;
}
void orc_set_environment(orc_scene* s, const double* rgb) { s->scene.environment = v3(rgb); }
void orc_set_environment_hdri(orc_scene* s, uint32_t w, uint32_t h, const double* rgb) {
    s->scene.hdri_w = w;
    s->scene.hdri_h = h;
    s->scene.hdri.resize(size_t(w) * h);
    for (size_t i = 0; i < size_t(w) * h; i++) s->scene.hdri[i] = v3(rgb + 3 * i);
}

static Camera make_camera(const orc_camera* c) {
    return Camera{v3(c->eye), v3(c->direction), v3(c->up), c->fov, c->aperture, c->focal_distance};
}

// Renderer::sample (src/renderer.rs:158-171): one task per image row, `threads` workers.
// If pixel_list != null only those n_pixels (y*w+x) are rendered (others left untouched).
int orc_render(orc_scene* s, const orc_camera* cam, const orc_params* prm, uint32_t iterations, uint64_t seed,
               uint32_t sample_offset, double* out_rgb, int threads, orc_counters* counters,
               const uint32_t* pixel_list, uint64_t n_pixels) {
    RenderParams rp{prm->width, prm->height, prm->exposure_value, prm->max_bounces, prm->robust};
    Renderer r{s->scene, make_camera(cam), rp};
    if (threads < 1) threads = 1;
    std::vector<Counters> cnts(threads);
    std::atomic<uint64_t> next(0);
    const uint64_t n_rows = pixel_list ? (n_pixels + 63) / 64 : prm->height;
    auto work = [&](int tid) {
        tl_cnt = counters ? &cnts[tid] : nullptr;
        for (;;) {
            uint64_t row = next.fetch_add(1);
            if (row >= n_rows) break;
            if (pixel_list) {
                for (uint64_t i = row * 64; i < std::min(n_pixels, row * 64 + 64); i++) {
                    uint32_t pix = pixel_list[i];
                    V3 c = r.get_color(pix % prm->width, pix / prm->width, iterations, seed, sample_offset);
                    out_rgb[size_t(pix) * 3 + 0] = c.x; out_rgb[size_t(pix) * 3 + 1] = c.y; out_rgb[size_t(pix) * 3 + 2] = c.z;
                }
            } else {
                uint32_t y = uint32_t(row);
                for (uint32_t x = 0; x < prm->width; x++) {
                    V3 c = r.get_color(x, y, iterations, seed, sample_offset);
                    size_t i = (size_t(y) * prm->width + x) * 3;
                    out_rgb[i] = c.x; out_rgb[i + 1] = c.y; out_rgb[i + 2] = c.z;
                }
            }
        }
        tl_cnt = nullptr;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& t : pool) t.join();
    if (counters) {
        Counters tot;
        for (auto& c : cnts) tot.add(c);
        *counters = orc_counters{tot.rays, tot.obj_tests, tot.nodes, tot.tri_tests, tot.hits, tot.samples,
                                 tot.vertices, tot.self_hits, tot.shadow_tests, tot.shadow_pass, tot.shadow_near};
    }
    return 0;
}

// get_closest_hit over n rays; t = +inf and obj = -1 on a miss.  robust selects the t_min policy.
int orc_intersect(orc_scene* s, uint64_t n, const double* o, const double* d, int robust, double* t, int32_t* obj,
                  double* normal) {
    RenderParams rp{1, 1, 0.0, 0, robust};
    Renderer r{s->scene, Camera{}, rp};
    for (uint64_t i = 0; i < n; i++) {
        Ray ray{v3(o + 3 * i), v3(d + 3 * i)};
        HitRecord h;
        int oi;
        bool hit = r.get_closest_hit(ray, h, oi);
        t[i] = hit ? h.time : INF;
        obj[i] = hit ? oi : -1;
        normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z;
    }
    return 0;
}

// ---- unit hooks for known-answer tests
int orc_shape_intersect(const orc_shape_desc* d, const double* o, const double* dir, double t_min, double t_max_in,
                        double* t, double* normal, int brute) {
    auto sh = make_shape(d);
    if (!sh) return -1;
    HitRecord h;
    h.time = t_max_in;
    Ray ray{v3(o), v3(dir)};
    bool hit;
    if (brute && d->kind == 3 && !d->has_transform) hit = static_cast<Mesh*>(sh.get())->intersect_brute(ray, t_min, h);
    else hit = sh->intersect(ray, t_min, h);
    *t = h.time;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    return hit ? 1 : 0;
}
// kd-tree vs brute force over many rays on one mesh (KAT 7); returns number of mismatches.
int64_t orc_mesh_kd_vs_brute(const orc_shape_desc* d, uint64_t n, const double* o, const double* dir, double* t_out) {
    auto sh = make_shape(d);
    if (!sh || d->kind != 3 || d->has_transform) return -1;
    Mesh* m = static_cast<Mesh*>(sh.get());
    int64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) {
        Ray ray{v3(o + 3 * i), v3(dir + 3 * i)};
        HitRecord a, b;
        bool ha = m->intersect(ray, EPSILON, a), hb = m->intersect_brute(ray, EPSILON, b);
        if (ha != hb || (ha && (a.time != b.time || a.normal.x != b.normal.x || a.normal.y != b.normal.y || a.normal.z != b.normal.z))) bad++;
        if (t_out) t_out[i] = ha ? a.time : INF;
    }
    return bad;
}
int orc_shape_sample(const orc_shape_desc* d, const double* target, uint64_t seed, uint32_t pixel, uint32_t sample,
                     double* v, double* n, double* p) {
    auto sh = make_shape(d);
    if (!sh || !sh->can_sample()) return -1;
    Rng rng(seed, pixel, sample);
    SurfSample s = sh->sample(v3(target), rng);
    v[0] = s.v.x; v[1] = s.v.y; v[2] = s.v.z;
    n[0] = s.n.x; n[1] = s.n.y; n[2] = s.n.z;
    *p = s.p;
    return int(rng.draws);
}
void orc_hex_color(uint32_t x, double* rgb) {
    V3 c = hex_color(x);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
void orc_color_bytes(const double* rgb, uint8_t* out) { color_bytes(v3(rgb), out); }
void orc_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out) {
    Rng rng(seed, pixel, sample);
    for (uint32_t i = 0; i < n; i++) out[i] = rng.next();
}
void orc_rng_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out) {
    Rng rng(seed, pixel, sample);
    for (uint32_t i = 0; i < n; i++) out[i] = rng.uniform();
}
// returns 1 for Some, 0 for None; draws = RNG draws consumed
int orc_material_sample_f(const orc_material* m, const double* normal, const double* wo, uint64_t seed, uint32_t pixel,
                          uint32_t sample, double* wi, double* pdf, int* draws) {
    Rng rng(seed, pixel, sample);
    V3 w;
    double p = 0;
    bool ok = sample_f(make_material(m), v3(normal), v3(wo), rng, w, p);
    wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
    *pdf = p;
    if (draws) *draws = int(rng.draws);
    return ok ? 1 : 0;
}
void orc_material_bsdf(const orc_material* m, const double* normal, const double* wo, const double* wi, double* out) {
    V3 f = bsdf(make_material(m), v3(normal), v3(wo), v3(wi));
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
}
void orc_medium_sample_d(int kind, double absorption, double scattering, uint64_t seed, uint32_t pixel, uint32_t sample,
                         double* dist, double* pdf, double* cdf) {
    Medium m{kind, absorption, scattering};
    Rng rng(seed, pixel, sample);
    Ray ray{V3(0, 0, 0), V3(0, 0, 1)};
    m.sample_d(ray, rng, *dist, *pdf, *cdf);
}
void orc_medium_sample_ph(int kind, uint64_t seed, uint32_t pixel, uint32_t sample, double* wi, double* p) {
    Medium m{kind, 0.0, 0.0};
    Rng rng(seed, pixel, sample);
    V3 w;
    m.sample_ph(V3(0, 0, 1), rng, w, *p);
    wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
}
void orc_camera_cast_ray(const orc_camera* c, double x, double y, uint64_t seed, uint32_t pixel, uint32_t sample,
                         double* o, double* d) {
    Rng rng(seed, pixel, sample);
    Ray r = make_camera(c).cast_ray(x, y, rng);
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
    d[0] = r.dir.x; d[1] = r.dir.y; d[2] = r.dir.z;
}
// Pixel -> NDC mapping of get_color (src/renderer.rs:174-176), KAT 13.
void orc_pixel_ndc(uint32_t x, uint32_t y, uint32_t w, uint32_t h, double* xn, double* yn) {
    double dim = double(std::max(w, h));
    *xn = (double(2 * x + 1) - double(w)) / dim;
    *yn = (double(2 * (h - y) - 1) - double(h)) / dim;
}
void orc_light_illuminate(orc_scene* s, int light_index, const double* pos, uint64_t seed, uint32_t pixel, uint32_t sample,
                          double* intensity, double* wi, double* dist) {
    Rng rng(seed, pixel, sample);
    V3 I, w;
    s->scene.lights[light_index].illuminate(v3(pos), rng, I, w, *dist);
    intensity[0] = I.x; intensity[1] = I.y; intensity[2] = I.z;
    wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
}

// ---------------------------------------------------------------------------- photon mapping (next tier)
struct orc_photon_map {
    PhotonMap pm;
    PhotonParams pp;
};
// Renderer::photon_render, shooting + map build (src/photon.rs:655-704).  Photon i draws from
// the stream (seed, pixel = i, sample = 0x80000000).  Returns null when no Light::Object exists.
orc_photon_map* orc_photon_map_build(orc_scene* s, uint64_t photon_count, int kind, double watts, uint64_t gather_size,
                                     uint64_t gather_size_volume, uint64_t seed, int robust) {
    RenderParams rp{1, 1, 0.0, 0, robust};
    Renderer r{s->scene, Camera{}, rp};
    auto* m = new orc_photon_map();
    m->pp = PhotonParams{photon_count, kind, watts, gather_size, gather_size_volume};
    m->pm.kind = kind;
    double power = watts / double(photon_count);
    for (uint64_t i = 0; i < photon_count; i++) {
        Rng rng(seed, uint32_t(i), 0x80000000u + uint32_t(i >> 32));
        Rng thin(seed, uint32_t(i), 0xC0000000u + uint32_t(i >> 32));
        if (!shoot_photon(r, power, rng, thin, kind, m->pm.list)) {
            delete m;
            return nullptr;
        }
    }
    m->pm.build();
    return m;
}
void orc_photon_map_free(orc_photon_map* m) { delete m; }
// Test hook: the maps of PhotonMap::new (src/photon.rs:181-311) over photon lists handed in -- n * 10 doubles each, laid out as
// orc_photon_map_get writes them (position, direction or start of the beam, power, unused) -- instead of lists this oracle shot: lets
// a camera pass be compared on the very photons of another implementation's shooting pass.
orc_photon_map* orc_photon_map_from_photons(uint64_t photon_count, int kind, double watts, uint64_t gather_size, uint64_t gather_size_volume,
                                            const double* surface, uint64_t n_surface, const double* volume, uint64_t n_volume) {
    auto* m = new orc_photon_map();
    m->pp = PhotonParams{photon_count, kind, watts, gather_size, gather_size_volume};
    m->pm.kind = kind;
    auto fill = [&](const double* in, uint64_t n, std::vector<Photon>& out, bool beams) {
        out.resize(n);
        for (uint64_t i = 0; i < n; i++) {
            const double* o = in + i * 10;
            out[i].position = V3(o[0], o[1], o[2]);
            out[i].power = V3(o[6], o[7], o[8]);
            if (beams) {
                out[i].starting_position = V3(o[3], o[4], o[5]);
                out[i].direction = -normalize(out[i].position - out[i].starting_position);
            } else {
                out[i].direction = V3(o[3], o[4], o[5]);
                out[i].starting_position = out[i].position;
            }
        }
    };
    fill(surface, n_surface, m->pm.list.surface, false);
    fill(volume, n_volume, m->pm.list.volume, kind == PK_BEAM_BEAM);
    m->pm.build();
    return m;
}
// which: 0 surface, 1 volume.  out (may be null): n * 10 doubles: position, direction, power, radius (volume, point-beam)
uint64_t orc_photon_map_get(orc_photon_map* m, int which, double* out) {
    const std::vector<Photon>& v = which == 0 ? m->pm.list.surface : m->pm.list.volume;
    if (out)
        for (size_t i = 0; i < v.size(); i++) {
            double* o = out + i * 10;
            o[0] = v[i].position.x; o[1] = v[i].position.y; o[2] = v[i].position.z;
            const bool beam = which == 1 && m->pm.kind == PK_BEAM_BEAM;  // beams: slots 3..5 = start of the beam
            const V3& dd = beam ? v[i].starting_position : v[i].direction;
            o[3] = dd.x; o[4] = dd.y; o[5] = dd.z;
            o[6] = v[i].power.x; o[7] = v[i].power.y; o[8] = v[i].power.z;
            o[9] = (which == 1 && i < m->pm.sphere_radius.size()) ? m->pm.sphere_radius[i] : 0.0;
        }
    return v.size();
}
// get_color_with_photon_map over the frame (src/photon.rs:950-985), one task per row.
int orc_photon_render(orc_scene* s, orc_photon_map* m, const orc_camera* cam, const orc_params* prm, uint32_t num_samples,
                      uint64_t seed, uint32_t sample_offset, double* out_rgb, int threads, const uint32_t* pixel_list,
                      uint64_t n_pixels) {
    RenderParams rp{prm->width, prm->height, prm->exposure_value, prm->max_bounces, prm->robust};
    Renderer r{s->scene, make_camera(cam), rp};
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> next(0);
    const uint64_t total = pixel_list ? n_pixels : uint64_t(prm->width) * prm->height;
    auto work = [&]() {
        for (;;) {
            uint64_t b = next.fetch_add(64);
            if (b >= total) break;
            for (uint64_t i = b; i < std::min(total, b + 64); i++) {
                uint32_t pix = pixel_list ? pixel_list[i] : uint32_t(i);
                uint32_t x = pix % prm->width, y = pix / prm->width;
                double dim = double(std::max(prm->width, prm->height));
                double xn = (double(2 * x + 1) - double(prm->width)) / dim;
                double yn = (double(2 * (prm->height - y) - 1) - double(prm->height)) / dim;
                V3 color(0, 0, 0);
                for (uint32_t k = 0; k < num_samples; k++) {
                    Rng rng(seed, pix, sample_offset + k);
                    double dx = rng.range(-1.0 / dim, 1.0 / dim), dy = rng.range(-1.0 / dim, 1.0 / dim);
                    Ray ray = r.camera.cast_ray(xn + dx, yn + dy, rng);
                    color = color + photon_estimate_indirect(r, m->pm, m->pp, ray, rng);
                }
                color = color / double(num_samples) * std::pow(2.0, rp.exposure_value);
                out_rgb[size_t(pix) * 3] = color.x; out_rgb[size_t(pix) * 3 + 1] = color.y; out_rgb[size_t(pix) * 3 + 2] = color.z;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return 0;
}

}  // extern "C"
