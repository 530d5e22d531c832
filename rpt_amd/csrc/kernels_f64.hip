// kernels_f64.hip — the reference-epsilon mode (option "epsilon_policy" = 1): rpt's sampling path in fp64 with rpt's
// own epsilons, for callers who need the reference's numbers rather than the fp32 path's speed.
//
// Everything here follows the cited reference lines literally: every object of scene.objects is tested per ray, in
// scene order, as the generic shape it is under its own `Transformed` matrices (no flatten-time specialisation, no
// tree: `Mesh` objects are scanned triangle by triangle behind the kd-tree's root box test), t_min = EPSILON = 1e-12
// (src/renderer.rs:17, 420), the light is visible iff |closest hit - distance to the sample| < 1e-12 (:348, :396),
// colours and path state are f64, and the compiler may not contract a*b+c (the self-intersections at t ~ 1e-11 and the
// near-miss shadow rejections that make rpt's images slightly darker are rounding noise of exactly these formulas).
// What is NOT the reference's: the RNG (per-(seed, pixel, sample) xoshiro128+ stream of the fp32 path, DESIGN.md
// section 2, deviation 1 -- the reference seeds from entropy) and the evaluation of the recursion as a loop with the
// carrier min(P + Q x, R) (exact algebra of src/renderer.rs:229-232, 271-280, 308-313; rounding-level differences only).
// One lane = one pixel: its samples are summed in order, as get_color does (:173-184); no partial-sum slab.
#include <hip/hip_runtime.h>

#include "device_core.h"   // Rng (the fp32 path's stream, bit for bit)
#include "f64_layout.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace rpt64 {

#define R64_DEV __device__ __forceinline__

static constexpr double kEps = 1e-12;            // src/renderer.rs:17
static constexpr double kFireflyClamp = 100.0;   // src/renderer.rs:18
static constexpr double kPi = 3.14159265358979323846;
static constexpr double kInf = __builtin_huge_val();

struct D {
    double x, y, z;
};
R64_DEV D mk(double x, double y, double z) { return D{x, y, z}; }
R64_DEV D ld(const double* p) { return D{p[0], p[1], p[2]}; }
R64_DEV D operator+(D a, D b) { return D{a.x + b.x, a.y + b.y, a.z + b.z}; }
R64_DEV D operator-(D a, D b) { return D{a.x - b.x, a.y - b.y, a.z - b.z}; }
R64_DEV D operator-(D a) { return D{-a.x, -a.y, -a.z}; }
R64_DEV D operator*(double s, D a) { return D{s * a.x, s * a.y, s * a.z}; }
R64_DEV D operator*(D a, double s) { return D{a.x * s, a.y * s, a.z * s}; }
R64_DEV D operator*(D a, D b) { return D{a.x * b.x, a.y * b.y, a.z * b.z}; }
R64_DEV D operator/(D a, double s) { return D{a.x / s, a.y / s, a.z / s}; }
R64_DEV double dot(D a, D b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
R64_DEV D cross(D a, D b) { return D{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
R64_DEV double length(D a) { return sqrt(dot(a, a)); }
R64_DEV D normalize(D a) { return a / length(a); }
R64_DEV D vmin(D a, D b) { return D{fmin(a.x, b.x), fmin(a.y, b.y), fmin(a.z, b.z)}; }
R64_DEV bool is_zero(D a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }
R64_DEV D mul3(const double* m, D v) {   // 3 x 3, row-major
    return D{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
R64_DEV D xf_point(const double* m, D p) {   // rows of a 3 x 4: M * (p, 1)
    return D{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
             m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
R64_DEV D xf_dir(const double* m, D d) {     // M * (d, 0)
    return D{m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z, m[8] * d.x + m[9] * d.y + m[10] * d.z};
}

// The fp32 path's RNG stream read as f64: u = (2k+1) 2^-24 is the same number in both precisions.
struct Rng64 {
    rptg::Rng r;
    R64_DEV double uniform() { return double(((r.next() >> 9) << 1) | 1u) * 0x1p-24; }   // rng.gen::<f64>()
    R64_DEV double range(double a, double b) { return a + (b - a) * uniform(); }           // rng.gen_range(a..b)
    R64_DEV uint32_t index(uint32_t n) { return __umulhi(r.next(), n); }                   // Uniform::from(0..n)
    R64_DEV void unit_disc(double& x, double& y) {                                         // rand_distr::UnitDisc
        for (;;) {
            x = range(-1.0, 1.0);
            y = range(-1.0, 1.0);
            if (x * x + y * y <= 1.0) return;
        }
    }
};

struct Hit {   // HitRecord, src/shape.rs:76-99
    double time;
    D normal;
};

// ---------------------------------------------------------------------------- shapes
// Sphere::intersect, src/shape/sphere.rs:14-46
R64_DEV bool hit_sphere(D o, D d, double t_min, Hit& rec) {
    const double a = dot(d, d), b = dot(d, o), c = dot(o, o) - 1.0;
    double disc = b * b - a * c;
    if (__builtin_signbit(disc)) return false;
    disc = sqrt(disc);
    double t = (-b - disc) / a;
    if (t < t_min) {
        t = (-b + disc) / a;
        if (t < t_min) return false;
    }
    if (t < rec.time) {
        rec.time = t;
        rec.normal = normalize(o + t * d);
        return true;
    }
    return false;
}
// Cube::intersect, src/shape/cube.rs:22-74
R64_DEV bool hit_cube(D o, D d, double t_min, Hit& rec) {
    double lo[3], hi[3], sg_lo[3], sg_hi[3];
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double x1 = (-0.5 - oo[k]) / dd[k], x2 = (0.5 - oo[k]) / dd[k];
        double n1 = -1.0, n2 = 1.0;
        if (x1 > x2) { const double t = x1; x1 = x2; x2 = t; n1 = 1.0; n2 = -1.0; }
        lo[k] = x1; hi[k] = x2; sg_lo[k] = n1; sg_hi[k] = n2;
    }
    int as, ae;
    if (lo[0] > lo[1] && lo[0] > lo[2]) as = 0; else if (lo[1] > lo[2]) as = 1; else as = 2;
    if (hi[0] < hi[1] && hi[0] < hi[2]) ae = 0; else if (hi[1] < hi[2]) ae = 1; else ae = 2;
    const double start = as == 0 ? lo[0] : (as == 1 ? lo[1] : lo[2]), end = ae == 0 ? hi[0] : (ae == 1 ? hi[1] : hi[2]);
    if (start > end || end < t_min) return false;
    const bool use_end = start < t_min;
    const double time = use_end ? end : start;
    if (time < rec.time) {
        const int ax = use_end ? ae : as;
        const double sg = use_end ? (ae == 0 ? sg_hi[0] : (ae == 1 ? sg_hi[1] : sg_hi[2])) : (as == 0 ? sg_lo[0] : (as == 1 ? sg_lo[1] : sg_lo[2]));
        rec.time = time;
        rec.normal = mk(ax == 0 ? sg : 0.0, ax == 1 ? sg : 0.0, ax == 2 ? sg : 0.0);
        return true;
    }
    return false;
}
// Plane::intersect, src/shape/plane.rs:17-32
R64_DEV bool hit_plane(const double* pl, D o, D d, double t_min, Hit& rec) {
    const D n = ld(pl);
    const double cosine = dot(n, d);
    if (fabs(cosine) < 1e-8) return false;
    const double time = (pl[3] - dot(n, o)) / cosine;
    if (time >= t_min && time < rec.time) {
        rec.time = time;
        const double sg = cosine > 0.0 ? 1.0 : (cosine < 0.0 ? -1.0 : (__builtin_signbit(cosine) ? -1.0 : 1.0));   // f64::signum
        rec.normal = -(normalize(n)) * sg;
        return true;
    }
    return false;
}
// Triangle::intersect, src/shape/mesh.rs:50-83
R64_DEV bool hit_tri(const Tri& tr, D o, D d, double t_min, Hit& rec) {
    const D v1 = ld(tr.v1);
    const D d0 = ld(tr.v2) - v1, d1 = ld(tr.v3) - v1;
    const D pn = normalize(cross(d0, d1));
    const double cosine = dot(pn, d);
    if (fabs(cosine) < 1e-8) return false;
    const double time = dot(pn, v1 - o) / cosine;
    if (time < t_min || time >= rec.time) return false;
    const D d2 = (o + time * d) - v1;
    const double d00 = dot(d0, d0), d01 = dot(d0, d1), d11 = dot(d1, d1), d20 = dot(d2, d0), d21 = dot(d2, d1);
    const double denom = d00 * d11 - d01 * d01;
    const double v = (d11 * d20 - d01 * d21) / denom, w = (d00 * d21 - d01 * d20) / denom, u = 1.0 - v - w;
    if (u >= 0.0 && v >= 0.0 && w >= 0.0) {
        rec.time = time;
        rec.normal = normalize(u * ld(tr.n1) + v * ld(tr.n2) + w * ld(tr.n3));
        return true;
    }
    return false;
}
// `Mesh = KdTree<Triangle>` (src/shape/mesh.rs:106): KdTree::intersect rejects the ray against the tree's bounds
// (src/kdtree.rs:132-139, BoundingBox::intersect :56-71; f64::min / max ignore a NaN operand like fmin / fmax), then
// finds the closest triangle.  The kd-tree below the root is an acceleration structure: every triangle is tested here.
R64_DEV bool hit_mesh(const Scene& sc, const Shape& s, D o, D d, double t_min, Hit& rec) {
    const double x1 = (s.bmin[0] - o.x) / d.x, x2 = (s.bmax[0] - o.x) / d.x;
    const double y1 = (s.bmin[1] - o.y) / d.y, y2 = (s.bmax[1] - o.y) / d.y;
    const double z1 = (s.bmin[2] - o.z) / d.z, z2 = (s.bmax[2] - o.z) / d.z;
    const double b_min = fmax(fmax(fmin(x1, x2), fmin(y1, y2)), fmin(z1, z2));
    const double b_max = fmin(fmin(fmax(x1, x2), fmax(y1, y2)), fmax(z1, z2));
    if (fmax(b_min, t_min) > fmin(b_max, rec.time)) return false;
    bool any = false;
    for (uint32_t i = 0; i < s.tri_count; i++)
        if (hit_tri(sc.tris[s.tri_first + i], o, d, t_min, rec)) any = true;
    return any;
}
// Shape::intersect of one object, Transformed::intersect (src/shape.rs:128-138) around it when has_xf
R64_DEV bool hit_shape(const Scene& sc, const Shape& s, D o, D d, double t_min, Hit& rec) {
    D ol = o, dl = d;
    if (s.has_xf) {   // Ray::apply_transform, src/shape.rs:65-72 (direction not renormalised: t is shared)
        ol = xf_point(s.inv, o);
        dl = xf_dir(s.inv, d);
    }
    bool h;
    if (s.kind == SH_SPHERE) h = hit_sphere(ol, dl, t_min, rec);
    else if (s.kind == SH_CUBE) h = hit_cube(ol, dl, t_min, rec);
    else if (s.kind == SH_PLANE) h = hit_plane(s.plane, ol, dl, t_min, rec);
    else h = hit_mesh(sc, s, ol, dl, t_min, rec);
    if (h && s.has_xf) rec.normal = normalize(mul3(s.nrm, rec.normal));
    return h;
}
// Renderer::get_closest_hit, src/renderer.rs:416-425
R64_DEV int closest_hit(const Args& a, D o, D d, Hit& rec) {
    const Scene& sc = a.sc;
    rec.time = kInf;
    rec.normal = mk(0, 0, 0);
    int obj = -1;
    for (uint32_t i = 0; i < sc.n_objects; i++)
        if (hit_shape(sc, sc.objects[i].shape, o, d, kEps, rec)) obj = int(i);
    if (a.counters) {
        atomicAdd(&a.counters[0], 1ull);
        if (obj >= 0) {
            atomicAdd(&a.counters[1], 1ull);
            const double m = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
            if (rec.time < 1e-9 * (1.0 + m)) atomicAdd(&a.counters[2], 1ull);   // diagnostic: a hit on the surface the ray starts on
        }
    }
    return obj;
}

// ---------------------------------------------------------------------------- Shape::sample
// of the unit shapes / a mesh in the shape's own space: point v, normal n, pdf p
R64_DEV void sample_local(const Scene& sc, const Shape& s, D target, Rng64& rng, D& v, D& n, double& p) {
    if (s.kind == SH_SPHERE) {   // src/shape/sphere.rs:53-65
        double x, y;
        rng.unit_disc(x, y);
        const double z = sqrt(1.0 - x * x - y * y);
        const D nn = normalize(target);
        const double ax = fabs(nn.x);
        const bool normal_x = ax >= 2.2250738585072014e-308 && ax < kInf;   // f64::is_normal
        const D n1 = normal_x ? normalize(mk(nn.y, -nn.x, 0.0)) : normalize(mk(0.0, -nn.z, nn.y));
        const D n2 = cross(n1, nn);
        v = x * n1 + y * n2 + z * nn;
        n = v;
        p = z * (1.0 / kPi);
    } else if (s.kind == SH_CUBE) {   // src/shape/cube.rs:76-89
        const double aa = rng.uniform() - 0.5, bb = rng.uniform() - 0.5;
        switch (rng.index(6)) {
            case 0: v = mk(aa, bb, 0.5); n = mk(0, 0, 1); break;
            case 1: v = mk(aa, bb, -0.5); n = mk(0, 0, -1); break;
            case 2: v = mk(aa, 0.5, bb); n = mk(0, 1, 0); break;
            case 3: v = mk(aa, -0.5, bb); n = mk(0, -1, 0); break;
            case 4: v = mk(0.5, aa, bb); n = mk(1, 0, 0); break;
            default: v = mk(-0.5, aa, bb); n = mk(-1, 0, 0); break;
        }
        p = 1.0 / 6.0;
    } else {   // KdTree::sample, src/kdtree.rs:141-146, over Triangle::sample, src/shape/mesh.rs:85-99
        const uint32_t idx = rng.index(s.tri_count);
        const Tri& tr = sc.tris[s.tri_first + idx];
        double u = rng.uniform(), vv = rng.uniform();
        while (u + vv > 1.0) {
            u = rng.uniform();
            vv = rng.uniform();
        }
        const double w = 1.0 - u - vv;
        const D v1 = ld(tr.v1), v2 = ld(tr.v2), v3 = ld(tr.v3);
        const double area = 0.5 * length(cross(v2 - v1, v3 - v1));
        v = u * v1 + vv * v2 + w * v3;
        n = normalize(u * ld(tr.n1) + vv * ld(tr.n2) + w * ld(tr.n3));
        p = (1.0 / area) / double(s.tri_count);
    }
}
R64_DEV void sample_shape(const Scene& sc, const Shape& s, D target, Rng64& rng, D& v, D& n, double& p) {
    if (!s.has_xf) return sample_local(sc, s, target, rng, v, n, p);
    // Transformed::sample, src/shape.rs:140-151
    D vl, nl;
    double pl;
    sample_local(sc, s, xf_point(s.inv, target), rng, vl, nl, pl);
    const D new_normal = normalize(mul3(s.nrm, nl));
    const double height = dot(mul3(s.lin, nl), new_normal);
    const double base = s.det / height;
    v = xf_point(s.fwd, vl);
    n = new_normal;
    p = pl / base;
}
// Light::illuminate for Light::Object, src/light.rs:34-45
R64_DEV void illuminate_object(const Scene& sc, const Light& L, D pos, Rng64& rng, D& intensity, D& wi, double& dist) {
    D v, n;
    double p;
    sample_shape(sc, L.obj.shape, pos, rng, v, n, p);
    const D disp = v - pos;
    const double len = length(disp);
    const double cosine = fmax(-dot(disp, n), 0.0) / len;
    const double surface_area = fmax(cosine, 0.0) / (len * len);
    const Mat& m = L.obj.mat;
    const bool has_color = m.kind <= 1;   // Material::color / emittance, src/material.rs:99-113
    const D col = has_color ? ld(m.albedo) : mk(0, 0, 0);
    const double emit = has_color ? m.emittance : 0.0;
    intensity = ((col * emit) * surface_area) / p;
    wi = disp / len;
    dist = len;
}

// ---------------------------------------------------------------------------- materials
R64_DEV D mat_color(const Mat& m) { return m.kind <= 1 ? ld(m.albedo) : mk(0, 0, 0); }
R64_DEV double mat_emit(const Mat& m) { return m.kind <= 1 ? m.emittance : 0.0; }
// nalgebra Rotation3::rotation_between(+Y, b) applied to v: axis normalize(Y x b), angle acos(Y.b); when the axis
// vanishes: identity if Y.b >= 0, else `None` -- Lambertian then retries from (0, 1, 1e-8), a half-turn about +X
// (src/material.rs:186-194); glm::quat_rotation (Phong, :213) falls back to the identity.
R64_DEV D rotate_from_y(D b, D v, bool pi_fallback_x) {
    const double s2 = b.x * b.x + b.z * b.z;
    if (s2 > 0.0) {
        const double s = sqrt(s2);
        const double kx = b.z / s, kz = -b.x / s;   // k = Y x b / |Y x b| = (b.z, 0, -b.x) / s
        const double kv = kx * v.x + kz * v.z;
        const D kxv = mk(-kz * v.y, kz * v.x - kx * v.z, kx * v.y);
        const double c = b.y, omc = 1.0 - c;
        return mk(c * v.x + s * kxv.x + omc * kv * kx, c * v.y + s * kxv.y, c * v.z + s * kxv.z + omc * kv * kz);
    }
    if (b.y < 0.0 && pi_fallback_x) return mk(v.x, -v.y, -v.z);
    return v;
}
R64_DEV D reflect_neg(D w, D n) { return (2.0 * dot(n, w)) * n - w; }   // -glm::reflect_vec(w, n)
// Material::sample_f, src/material.rs:166-263
R64_DEV bool sample_f(const Mat& m, D n, D wo, Rng64& rng, D& wi, double& pdf) {
    if (m.kind == 0) {
        const double r1 = rng.uniform(), r2 = rng.uniform();
        const double phi = 2.0 * kPi * r1, theta = acos(sqrt(r2));
        pdf = cos(theta) / kPi;
        const D dir = mk(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
        wi = normalize(rotate_from_y(normalize(n), dir, true));
        return true;
    }
    if (m.kind == 1) {
        const double r1 = rng.uniform(), r2 = rng.uniform();
        const double phi = 2.0 * kPi * r1, theta = acos(pow(r2, 1.0 / (m.shininess + 1.0)));
        pdf = (m.shininess + 1.0) / (2.0 * kPi) * pow(cos(theta), m.shininess);
        const D dir = mk(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
        const D refl = reflect_neg(wo, n);
        wi = normalize(rotate_from_y(normalize(refl), dir, false));
        return true;
    }
    if (m.kind == 2) {
        wi = reflect_neg(wo, normalize(n));
        pdf = 1.0;
        return true;
    }
    const bool inside = dot(n, wo) < 0.0;
    const D nn = inside ? -n : n;
    const double ci = fmin(fmax(dot(wo, nn), 0.0), 1.0);
    const double ni = inside ? m.ior : 1.0, nt = inside ? 1.0 : m.ior;
    double r0 = (ni - nt) / (ni + nt);
    r0 = r0 * r0;
    const double om = 1.0 - ci;
    const double sr = fmin(fmax(r0 + (1.0 - r0) * (om * om * om * om * om), 0.0), 1.0);
    pdf = 1.0;
    if (rng.uniform() < sr) {
        wi = reflect_neg(wo, n);
        return true;
    }
    const double eta = ni / nt;
    const double k = 1.0 - (eta * eta) * (1.0 - ci * ci);
    if (k < 0.0) return false;   // sqrt -> NaN -> None: total internal reflection
    const double ct = sqrt(k);
    wi = eta * (-wo) + (eta * ci - ct) * nn;
    return true;
}
// Material::bsdf, src/material.rs:266-289
R64_DEV D bsdf(const Mat& m, D n, D wo, D wi) {
    if (__builtin_signbit(dot(n, wi)) || __builtin_signbit(dot(n, wo))) return mk(0, 0, 0);
    if (m.kind == 0) return (1.0 / kPi) * ld(m.albedo);
    if (m.kind == 1) {
        const D normalization = ld(m.albedo) * ((m.shininess + 2.0) / (2.0 * kPi));
        const D refl = -normalize(wi - (2.0 * dot(n, wi)) * n);
        return normalization * pow(fmin(fmax(dot(refl, wo), 0.0), 1.0), m.shininess);
    }
    return mk(1, 1, 1);
}

// ---------------------------------------------------------------------------- the path
// Camera::cast_ray, src/camera.rs:65-82
R64_DEV void cast_ray(const Camera& c, double x, double y, Rng64& rng, D& o, D& d) {
    const D right = ld(c.right), up = ld(c.up);
    o = ld(c.eye);
    D nd = c.d * ld(c.direction) + x * right + y * up;
    if (c.aperture > 0.0) {
        const D focal = o + normalize(nd) * c.focal_distance;
        double dx, dy;
        rng.unit_disc(dx, dy);
        o = o + (dx * right + dy * up) * c.aperture;
        nd = focal - o;
    }
    d = normalize(nd);
}
// The shadow test of sample_lights / sample_lights_for_media (src/renderer.rs:339-348, 386-396)
R64_DEV bool light_visible(const Args& a, D pos, D wi, double dist) {
    Hit h;
    const int obj = closest_hit(a, pos, wi, h);
    if (a.counters) atomicAdd(&a.counters[3], 1ull);
    if (obj < 0) return false;
    const double miss = fabs(h.time - dist);
    if (a.counters) {
        if (miss < kEps) atomicAdd(&a.counters[4], 1ull);
        else if (miss < 1e-6 * dist) atomicAdd(&a.counters[5], 1ull);   // diagnostic: the light's own surface, missed by rounding
    }
    return miss < kEps;
}

// Medium::color (src/medium.rs:80-122): hex_color(0xD2B48C) for homogeneous_isotropic; blue below / red above y = 250 for
// colored_glowing_fog.  The host passes the two colours (color.rs:10-15 evaluated in fp64).
R64_DEV D medium_color(const Args& a, D pos) {
    return (a.sc.medium_kind == 1 && pos.y > 250.0) ? ld(a.medium_color_hi) : ld(a.medium_color);
}

__global__ __launch_bounds__(256) void render_f64_kernel(const Args a) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= a.n_owned) return;
    const uint32_t tile = a.tiles[p >> 10], within = p & 1023u;
    const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const uint32_t x = tx * 32u + (within & 31u), y = ty * 32u + (within >> 5);
    if (x >= a.width || y >= a.height) return;
    const Scene& sc = a.sc;
    const bool medium = sc.has_medium != 0;
    // src/renderer.rs:174-176 (2 * x + 1 and 2 * (h - y) - 1 in u32, like the reference)
    const double xn = (double(2u * x + 1u) - double(a.width)) / a.dim;
    const double yn = (double(2u * (a.height - y) - 1u) - double(a.height)) / a.dim;
    D color = mk(0, 0, 0);
    for (uint32_t s = 0; s < a.iterations; s++) {
        Rng64 rng;
        rng.r.seed(a.seed_mixed, y * a.width + x, a.sample_offset + s);
        const double dx = rng.range(-1.0 / a.dim, 1.0 / a.dim);
        const double dy = rng.range(-1.0 / a.dim, 1.0 / a.dim);
        D ro, rd;
        cast_ray(a.cam, xn + dx, yn + dy, rng, ro, rd);
        if (a.counters) atomicAdd(&a.counters[6], 1ull);
        // trace_ray (src/renderer.rs:187-322) as a loop: the radiance of the path = min(P + Q x, R) per channel, x = what the
        // rest of the path returns (closed under x -> E + min(k x, 100), :308-313; in a medium there is no clamp, R = inf)
        D P = mk(0, 0, 0), Q = mk(1, 1, 1), R = mk(kInf, kInf, kInf);
        uint32_t depth = 0;
        for (;;) {
            if (a.counters) atomicAdd(&a.counters[7], 1ull);
            double dmed = kInf;
            if (medium) dmed = -log(rng.range(0.0, 1.0)) / (sc.absorption + sc.scattering);   // Medium::sample_d, src/medium.rs:133-146
            const D wo = -normalize(rd);
            Hit h;
            const int obj = closest_hit(a, ro, rd, h);
            const bool hit = obj >= 0;
            const bool ev_medium = medium && dmed < (hit ? h.time : 400.0);   // :197-243 (`d >= h.time` is a surface event)
            if (!ev_medium && !hit) {   // :198-206 (in a medium the background counts only beyond 400), :288
                const D env = (!medium || dmed >= 400.0) ? ld(sc.env) : mk(0, 0, 0);
                color = color + vmin(P + Q * env, R);
                break;
            }
            D E, k = mk(0, 0, 0), pos, wi_next = mk(0, 0, 1);
            bool cont = false;
            if (ev_medium) {   // :243-283
                pos = ro + dmed * rd;
                const D mcol = medium_color(a, pos);
                const double scat = sc.scattering, extinction = sc.absorption + sc.scattering;
                const double emm = sc.medium_kind == 1 ? 10.0 : 0.0;
                const double phase = sc.medium_kind == 1 ? 1.0 / 4.0 * kPi : 1.0 / (4.0 * kPi);   // (sic, src/medium.rs:113)
                E = depth == 0 ? emm * mcol : mk(0, 0, 0);
                // sample_lights_for_media, :325-359
                for (uint32_t li = 0; li < sc.n_lights; li++) {
                    const Light& L = sc.lights[li];
                    if (L.kind == LT_AMBIENT) {
                        E = E + ld(L.color) * mcol;
                    } else if (L.kind == LT_OBJECT) {
                        D I, wi;
                        double dist;
                        illuminate_object(sc, L, pos, rng, I, wi, dist);
                        if (light_visible(a, pos, wi, dist)) E = E + ((scat / extinction) * (I * mcol)) * phase;
                    }
                    // Point / Directional: illuminate draws nothing and the test |hit - dist| < 1e-12 can never pass
                    // (dist = the light's position / +inf, src/light.rs:26-33)
                }
                if (rng.uniform() < 0.8) {   // :262-281
                    const double ax = rng.range(-1.0, 1.0), ay = rng.range(-1.0, 1.0), az = rng.range(-1.0, 1.0);
                    wi_next = normalize(mk(ax, ay, az));   // Medium::sample_ph, src/medium.rs:87-93
                    k = ((((scat / extinction) / phase) * mcol) * phase) / 0.8;   // (scat/ext) x / ph_p . color * phase / rr_p, ph_p == phase
                    cont = true;
                }
            } else {   // surface event: :207-237 in a medium, :289-318 without
                pos = ro + h.time * rd;
                const Mat& mat = sc.objects[obj].mat;
                const D n = h.normal;
                E = depth == 0 ? mat_emit(mat) * mat_color(mat) : mk(0, 0, 0);
                // sample_lights, :362-409
                for (uint32_t li = 0; li < sc.n_lights; li++) {
                    const Light& L = sc.lights[li];
                    if (L.kind == LT_AMBIENT) {
                        E = E + ld(L.color) * mat_color(mat);
                    } else if (L.kind == LT_OBJECT) {
                        D I, wi;
                        double dist;
                        illuminate_object(sc, L, pos, rng, I, wi, dist);
                        if (light_visible(a, pos, wi, dist)) E = E + (bsdf(mat, n, wo, wi) * I) * dot(wi, n);
                    }
                }
                const bool go = medium ? (rng.uniform() < 0.8) : (depth < a.max_bounces);   // :222 / :301
                if (go) {
                    double pdf;
                    if (sample_f(mat, n, wo, rng, wi_next, pdf)) {
                        const D f = bsdf(mat, n, wo, wi_next);
                        k = ((1.0 / (medium ? pdf * 0.8 : pdf)) * f) * fabs(dot(wi_next, n));
                        cont = true;
                    }
                }
            }
            P = P + Q * E;
            if (!cont) {   // (a path whose weight has become zero goes on, as the reference's recursion does: same rays, same draws)
                color = color + vmin(P, R);
                break;
            }
            if (!medium) R = vmin(R, P + kFireflyClamp * Q);   // FIREFLY_CLAMP, :311-313
            Q = Q * k;
            ro = pos;
            rd = wi_next;
            depth++;
        }
    }
    const size_t o = (size_t(y) * a.width + x) * 3;
    const double inv = a.scale / double(a.iterations);   // color / iterations * 2^EV, :183
    a.out[o] = color.x * inv;
    a.out[o + 1] = color.y * inv;
    a.out[o + 2] = color.z * inv;
}

}  // namespace rpt64

namespace rptg {
hipError_t launch_render_f64(const rpt64::Args& a, hipStream_t stream) {
    if (!a.n_owned) return hipSuccess;
    hipLaunchKernelGGL(rpt64::render_f64_kernel, dim3((a.n_owned + 255u) / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}
}  // namespace rptg
