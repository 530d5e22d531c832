// kernels_f64.hip — the reference-epsilon mode (option "epsilon_policy" = 1): rpt's sampling path in fp64 with rpt's
// own epsilons, on the fp32 path's machinery (persistent grid, wave-batched work queue, path regeneration, slab).
//
// The ARITHMETIC is the cited reference lines, literally: generic shapes under their own `Transformed` matrices, scene
// order, t_min = EPSILON = 1e-12 (src/renderer.rs:17, 420), the light is visible iff |closest hit - distance to the
// sample| < 1e-12 (:348, :396), colours and path state in f64, no contraction of a*b+c (the self-intersections at
// t ~ 1e-11 and the near-miss shadow rejections that make rpt's images slightly darker are rounding noise of exactly
// these formulas).  What is NOT the reference's: the RNG (per-(seed, pixel, sample) xoshiro128+ stream of the fp32
// path, DESIGN.md section 2, deviation 1), the recursion evaluated as a loop with the carrier min(P + Q x, R) (exact
// algebra of src/renderer.rs:229-232, 271-280, 308-313), and the order of the additions that form a pixel's mean
// (chunks of samples summed separately, then in chunk order: 1e-16 relative).
//
// The SCHEDULE is this file's own (f64_layout.h): a closest-hit query first tests the ray against every object's
// padded world box in fp32 (wave-uniform records from the scalar cache, full rate), then each lane evaluates only its
// own candidates in fp64, in scene order, a different object in every lane, from per-lane records in LDS.  Objects the
// box test drops cannot change the reference's HitRecord, so hit, time and normal are those of the full scan bit for
// bit (`cull` = 0 runs the full scan; tests/test_gpu_epsilon.py compares the two frames for equality).  The box test
// knows the one case in which the reference reports a hit for a ray that misses an object's box: a NaN in its cube test
// (cull32 below).
#include <hip/hip_runtime.h>

#include "device_core.h"   // Rng (the fp32 path's stream, bit for bit)
#include "f64_layout.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace rpt64 {

#define R64_DEV __device__ __forceinline__
#define R64_CONST __attribute__((address_space(4)))
// The kernel's arguments, read from the kernel-argument segment where they are used (scalar loads through the constant
// address space) instead of being held in scalar registers from the kernel's entry: ~100 dwords of arguments would
// otherwise live in VGPR lanes (kernels.hip, "kernel arguments are read where they are used").  Valid in
// render_f64_kernel and what it inlines: `Args` is that kernel's one argument.
#define KA (*rptg::kernarg_args<Args>())
// -DRPT_MARKERS drops "; SECT k" comments into the ISA (static instruction counts per section: tools/f64_sections.py)
#ifdef RPT_MARKERS
#define R64_MARK(k) asm volatile("; SECT " #k ::: "memory")
#else
#define R64_MARK(k)
#endif
// SECT64(k): the marker, and in the counters build the section's wave-level executions (counters[16 + 2k]) and the lanes
// enabled in them (counters[17 + 2k]); tools/f64_sections.py puts them beside the static instruction counts.
#define SECT64(k)                                                                                    \
    do {                                                                                             \
        R64_MARK(k);                                                                                 \
        if (COUNT) {                                                                                 \
            const uint64_t m_ = __ballot(true);                                                      \
            if (mbcnt64(m_) == 0u) {                                                                 \
                atomicAdd(&KA.counters[16 + 2 * (k)], 1ull);                                         \
                atomicAdd(&KA.counters[17 + 2 * (k)], (unsigned long long)__popcll(m_));             \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#ifndef R64_WAVES
#define R64_WAVES 4   // waves per SIMD the kernel is compiled for (128 VGPRs; C3 118.6 ms against 126.7 at 3 and 137.7 at 2, although 4 spills)
#endif

static constexpr double kEps = 1e-12;            // src/renderer.rs:17
static constexpr double kFireflyClamp = 100.0;   // src/renderer.rs:18
static constexpr double kPi = 3.14159265358979323846;
static constexpr double kInf = __builtin_huge_val();

struct D {
    double x, y, z;
};
R64_DEV D mk(double x, double y, double z) { return D{x, y, z}; }
template <class P> R64_DEV D ld(P p) { return D{p[0], p[1], p[2]}; }
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
template <class P> R64_DEV D mul3(P m, D v) {   // 3 x 3, row-major
    return D{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
template <class P> R64_DEV D xf_point(P m, D p) {   // rows of a 3 x 4: M * (p, 1)
    return D{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
             m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
template <class P> R64_DEV D xf_dir(P m, D d) {     // M * (d, 0)
    return D{m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z, m[8] * d.x + m[9] * d.y + m[10] * d.z};
}
// A wave-uniform record read through the constant address space (scalar loads, SGPR operands of the fp64 instructions)
template <class T> R64_DEV const R64_CONST T& uniform_ref(const T* p) { return *(const R64_CONST T*)(uintptr_t)p; }

RPT_DEV uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
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

// ---------------------------------------------------------------------------- the closest-hit query
// HitRecord (src/shape.rs:76-99) of a query.  The normal is formed once, for the winner (`hit_normal`), from the ray, the
// time and `aux`.
struct Query {
    double t;        // HitRecord::time
    int32_t obj;     // index into scene.objects, -1: none
    uint32_t aux;    // cube: axis | negative << 2; plane: sign of the cosine; mesh: triangle (index into trecs)
};
// What Shape::intersect of ONE object finds for a ray when the record it is handed is empty (time = +inf).
struct PairHit {
    double t;        // time of the hit, +inf: none
    double bm;       // a mesh: max(entry into the tree's bounds, t_min); others: -inf (see closest_hit_wave)
    uint32_t aux;
};

// Shape::intersect of object i (Transformed::intersect around it when has_xf, src/shape.rs:128-138; Ray::apply_transform
// does not renormalise the direction, so t is shared) on an empty record.
template <bool COUNT, class RP, class TP>
R64_DEV PairHit eval_pair(RP recs, TP trecs, uint32_t i, D o, D d) {
    SECT64(5);
    PairHit h{kInf, -kInf, 0u};
    const auto& r = recs[i];
    const int32_t kind = r.kind;
    D ol = o, dl = d;
    double bm = -kInf;
    // the groups around the shape, outermost first: Transformed::intersect's ray map (when the group is transformed), then
    // KdTree::intersect's bounds test (src/kdtree.rs:132-139) on the empty record
    for (uint32_t f = 0; f < r.n_frames; f++) {
        const FrameRec& F = KA.sc.frames[r.frame[f]];
        if (KA.sc.fshade[r.frame[f]].has_xf) {
            const D p = xf_point(F.inv, ol), q = xf_dir(F.inv, dl);
            ol = p;
            dl = q;
        }
        const double x1 = (F.b[0] - ol.x) / dl.x, x2 = (F.b[3] - ol.x) / dl.x;
        const double y1 = (F.b[1] - ol.y) / dl.y, y2 = (F.b[4] - ol.y) / dl.y;
        const double z1 = (F.b[2] - ol.z) / dl.z, z2 = (F.b[5] - ol.z) / dl.z;
        const double b_min = fmax(fmax(fmin(x1, x2), fmin(y1, y2)), fmin(z1, z2));
        const double b_max = fmin(fmin(fmax(x1, x2), fmax(y1, y2)), fmax(z1, z2));
        const double g = fmax(b_min, kEps);
        if (g > fmin(b_max, kInf)) return h;
        bm = fmax(bm, g);
    }
    h.bm = bm;
    if (r.has_xf) {
        const D p = xf_point(r.inv, ol), q = xf_dir(r.inv, dl);
        ol = p;
        dl = q;
    }
    if (kind == SH_PLANE) {   // Plane::intersect, src/shape/plane.rs:17-32
        SECT64(6);
        const D n = ld(r.b);
        const double cosine = dot(n, dl);
        if (!(fabs(cosine) < 1e-8)) {
            const double time = (r.b[3] - dot(n, ol)) / cosine;
            if (time >= kEps && time < kInf) {
                h.t = time;
                h.aux = (cosine > 0.0 || (cosine == 0.0 && !__builtin_signbit(cosine))) ? 0u : 1u;   // f64::signum of the cosine
            }
        }
    } else if (kind == SH_SPHERE) {   // Sphere::intersect, src/shape/sphere.rs:14-46
        SECT64(7);
        const double a = dot(dl, dl), b = dot(dl, ol), c = dot(ol, ol) - 1.0;
        double disc = b * b - a * c;
        if (!__builtin_signbit(disc)) {
            disc = sqrt(disc);
            double t = (-b - disc) / a;
            bool ok = true;
            if (t < kEps) {
                t = (-b + disc) / a;
                ok = !(t < kEps);
            }
            if (ok && t < kInf) h.t = t;
        }
    } else {
        SECT64(8);
        // the six slab roots: Cube::intersect (src/shape/cube.rs:23-35, planes at -0.5 / 0.5) and BoundingBox::intersect of
        // a mesh's bounds (src/kdtree.rs:56-71) are the same (plane - o) / d
        const double x1 = (r.b[0] - ol.x) / dl.x, x2 = (r.b[3] - ol.x) / dl.x;
        const double y1 = (r.b[1] - ol.y) / dl.y, y2 = (r.b[4] - ol.y) / dl.y;
        const double z1 = (r.b[2] - ol.z) / dl.z, z2 = (r.b[5] - ol.z) / dl.z;
        if (kind == SH_CUBE) {   // src/shape/cube.rs:36-74
            SECT64(9);
            const bool sx = x1 > x2, sy = y1 > y2, sz = z1 > z2;
            const double lo0 = sx ? x2 : x1, hi0 = sx ? x1 : x2;
            const double lo1 = sy ? y2 : y1, hi1 = sy ? y1 : y2;
            const double lo2 = sz ? z2 : z1, hi2 = sz ? z1 : z2;
            // the normal of an interval end: -1 along the axis for the -0.5 plane, +1 for the 0.5 plane (swapped with the roots)
            const int as = (lo0 > lo1 && lo0 > lo2) ? 0 : (lo1 > lo2 ? 1 : 2);
            const int ae = (hi0 < hi1 && hi0 < hi2) ? 0 : (hi1 < hi2 ? 1 : 2);
            const double start = as == 0 ? lo0 : (as == 1 ? lo1 : lo2), end = ae == 0 ? hi0 : (ae == 1 ? hi1 : hi2);
            if (!(start > end || end < kEps)) {
                const bool use_end = start < kEps;
                const double time = use_end ? end : start;
                if (time < kInf) {
                    const int ax = use_end ? ae : as;
                    const bool swapped = ax == 0 ? sx : (ax == 1 ? sy : sz);
                    const bool negative = use_end ? swapped : !swapped;   // start carries the lower root's normal (-1 unless swapped)
                    h.t = time;
                    h.aux = uint32_t(ax) | (negative ? 4u : 0u);
                }
            }
        } else {
            // `Mesh = KdTree<Triangle>` (src/shape/mesh.rs:106): KdTree::intersect rejects the ray against the tree's bounds
            // (src/kdtree.rs:132-139; f64::min / max ignore a NaN operand like fmin / fmax), then finds the closest triangle.
            // The kd-tree below the root is an acceleration structure: every triangle is tested here.
            SECT64(10);
            const double b_min = fmax(fmax(fmin(x1, x2), fmin(y1, y2)), fmin(z1, z2));
            const double b_max = fmin(fmin(fmax(x1, x2), fmax(y1, y2)), fmax(z1, z2));
            const double own = fmax(b_min, kEps);
            h.bm = fmax(bm, own);
            if (!(own > fmin(b_max, kInf))) {
                for (uint32_t j = 0; j < r.tri_count; j++) {   // Triangle::intersect, src/shape/mesh.rs:50-83
                    SECT64(11);
                    const auto& tr = trecs[r.tri_first + j];
                    const D pn = ld(tr.pn), v1 = ld(tr.v1);
                    const double cosine = dot(pn, dl);
                    if (fabs(cosine) < 1e-8) continue;
                    const double time = dot(pn, v1 - ol) / cosine;
                    if (time < kEps || time >= h.t) continue;
                    SECT64(12);
                    const D d2 = (ol + time * dl) - v1;
                    const double d20 = dot(d2, ld(tr.d0)), d21 = dot(d2, ld(tr.d1));
                    const double v = (tr.d11 * d20 - tr.d01 * d21) / tr.denom, w = (tr.d00 * d21 - tr.d01 * d20) / tr.denom;
                    const double u = 1.0 - v - w;
                    if (u >= 0.0 && v >= 0.0 && w >= 0.0) {
                        h.t = time;
                        h.aux = r.tri_first + j;
                    }
                }
            }
        }
    }
    return h;
}

// The fp32 box test that decides which objects a lane evaluates.  Conservative: boxes are padded at commit (1e-5 of
// their extent and of their coordinates), the ray's origin error is covered by `eo`, and the slab interval is widened
// by 4e-6 relative before it is compared -- three orders of magnitude above what fp32 rounding of the fp64 ray and of
// the slab arithmetic can move it.  A NaN (0 * inf) compares false and keeps the object.
// One exception to "an object whose box the ray misses cannot be hit": a ray that runs exactly parallel to an axis (a direction
// component that is 0 -- a visibility ray between two points of one axis-aligned wall, y = 0 = y) and starts exactly in a face plane of
// a cube makes the reference's slab test divide 0 by 0, and with a NaN in one axis's interval its comparisons no longer hold the other
// axes' entry times (src/shape/cube.rs:23-60): the reference then reports a hit for a ray that passes BESIDE the cube.  (A mesh's or a
// group's bounds gate can be fooled the same way, but only into letting the ray in: the hit itself is a triangle's or a child's, whose own
// box the ray must then cross.)  So a cube (CullBox::slab_test) is also kept when the ray has a zero component in an axis in which its
// origin lies within the padding of one of the box's two faces (a zero component shows as an infinite reciprocal; the extra test runs only in
// waves that hold such a ray).
template <bool ANY_ZERO>
R64_DEV uint32_t cull32(const CullBox* boxes, uint32_t base, uint32_t nb, float ox, float oy, float oz, float ix, float iy, float iz,
                        float eo, float tlim) {
    uint32_t mask = 0u;
    // (the origin's error bound goes into the origin once: lo - eo - o = lo - (o + eo))
    const float oxl = ox + eo, oyl = oy + eo, ozl = oz + eo, oxh = ox - eo, oyh = oy - eo, ozh = oz - eo;
    for (uint32_t j = 0; j < nb; j++) {
        const rptg::F4 lo = rptg::uload(reinterpret_cast<const rptg::F4*>(boxes + base + j));
        const rptg::F4 hi = rptg::uload(reinterpret_cast<const rptg::F4*>(boxes + base + j) + 1);
        if (__float_as_uint(lo.w) != 0u) {   // unbounded (a plane): always evaluated
            mask |= 1u << j;
            continue;
        }
        const float x1 = (lo.x - oxl) * ix, x2 = (hi.x - oxh) * ix;
        const float y1 = (lo.y - oyl) * iy, y2 = (hi.y - oyh) * iy;
        const float z1 = (lo.z - ozl) * iz, z2 = (hi.z - ozh) * iz;
        const float tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
        const float tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
        const float tn_lo = tn - 4e-6f * fabsf(tn), tf_hi = tf + 4e-6f * fabsf(tf);
        bool out = tn_lo > tf_hi || tf_hi < 0.f || tn_lo > tlim;
        if (ANY_ZERO && __float_as_uint(hi.w) != 0u) {   // (wave-uniform: such a ray in the wave, and a cube -- or a union box that holds one)
            // the faces lie within two paddings (a group's child is padded twice) of the stored ones: 1e-5 of the box's size and of its coordinates each
            const float size = fmaxf(fmaxf(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z);
            const float mag = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))), fmaxf(fabsf(lo.z), fabsf(hi.z)));
            const float tol = 5e-5f * (size + mag) + eo;
            const bool zx = fabsf(ix) == __builtin_huge_valf(), zy = fabsf(iy) == __builtin_huge_valf(), zz = fabsf(iz) == __builtin_huge_valf();   // 1 / (+-0)
            const bool on_face = (zx && (fabsf(ox - lo.x) <= tol || fabsf(ox - hi.x) <= tol)) ||
                                 (zy && (fabsf(oy - lo.y) <= tol || fabsf(oy - hi.y) <= tol)) ||
                                 (zz && (fabsf(oz - lo.z) <= tol || fabsf(oz - hi.z) <= tol));
            if (on_face || (__float_as_uint(hi.w) == 2u && (zx || zy || zz))) out = false;
        }
        if (!out) mask |= 1u << j;
    }
    return mask;
}

R64_DEV uint32_t lane_read(uint32_t v, uint32_t src_lane) { return uint32_t(__builtin_amdgcn_ds_bpermute(int(src_lane << 2), int(v))); }
R64_DEV double lane_read(double v, uint32_t src_lane) {
    const uint64_t u = __double_as_longlong(v);
    const uint32_t lo = lane_read(uint32_t(u), src_lane), hi = lane_read(uint32_t(u >> 32), src_lane);
    return __longlong_as_double((long long)((uint64_t(hi) << 32) | lo));
}
R64_DEV D lane_read(D v, uint32_t src_lane) { return mk(lane_read(v.x, src_lane), lane_read(v.y, src_lane), lane_read(v.z, src_lane)); }

// Renderer::get_closest_hit, src/renderer.rs:416-425, for the wave's queries together.  EVERY lane of the wave calls this
// (wave-uniform control flow); `mine`: the lane has a query (o, d) -- `tlim`: hits beyond it do not matter to it (+inf: all
// do) --, the others only work.
//
// A lane's candidates are the objects its fp32 box test keeps; on average a query has about one and a wave about forty,
// unevenly: were every lane to evaluate its own, the wave would run as many rounds as its unluckiest lane has candidates
// with most lanes idle.  So the (ray, object) pairs of the wave are dealt out to its 64 lanes: rank k = the k-th candidate
// of each lane in scene order; whole ranks are packed into a chunk of at most 64 pairs (pair = slot `used + position among
// the rank's lanes`, written to the wave's 64 LDS dwords); a slot lane fetches its pair's ray from the owner
// (ds_bpermute), evaluates the object on an empty record, and the owners collect their results rank by rank, i.e. in scene
// order, and apply them as the reference's loop does: object i replaces the record iff its time is smaller -- and, for a
// mesh, iff KdTree::intersect's bounds test (src/kdtree.rs:132-139), which sees the record's time, lets the ray in:
// !(max(b_min, t_min) > min(b_max, rec.time)); on the empty record that is `bm <= b_max` (else no triangle was tested), so
// with the record's time R it is !(bm > R).  Inside a mesh the triangles run in order on one lane.  The record is
// therefore the reference's after every object, bit for bit.
template <bool COUNT, class RP, class TP>
R64_DEV void closest_hit_wave(RP recs, TP trecs, volatile uint32_t* slots, bool mine, D o, D d, double tlim, Query& q, uint32_t& c_evals,
                              uint32_t& c_rounds) {
    SECT64(3);
    const uint32_t lane = threadIdx.x & 63u;
    q.t = kInf;
    q.obj = -1;
    q.aux = 0u;
    const uint32_t n = KA.sc.n_objects;
    const bool cull = KA.cull != 0u;
    const float ox = float(o.x), oy = float(o.y), oz = float(o.z);
    const float ix = __builtin_amdgcn_rcpf(float(d.x)), iy = __builtin_amdgcn_rcpf(float(d.y)), iz = __builtin_amdgcn_rcpf(float(d.z));
    const float eo = 1e-6f * fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    const float tl = float(tlim) * 1.00001f;
    // direction components that are zero (or too small for fp32: treated alike): see cull32
    // a zero direction component whose origin coordinate lies in a bucket some cube's face marked (Scene::face_bits): see cull32
    auto near_face = [&](int k, float dk, float ok) {
        if (dk != 0.f) return false;
        const float f = (ok - KA.sc.face_base[k]) * KA.sc.face_inv_cell[k];
        return f >= 0.f && f < 64.f && ((KA.sc.face_bits[k] >> uint32_t(f)) & 1ull) != 0ull;
    };
    bool any_zero = false;   // wave-uniform
#ifndef R64_NO_SLAB_EXCEPTION   // (defined in A/B builds only: tools/build_variant.sh)
    if (__ballot(mine && (float(d.x) == 0.f || float(d.y) == 0.f || float(d.z) == 0.f)) != 0ull)
        any_zero = __ballot(mine && (near_face(0, float(d.x), ox) || near_face(1, float(d.y), oy) || near_face(2, float(d.z), oz))) != 0ull;
#endif
    for (uint32_t base = 0; base < n; base += 32u) {
        const uint32_t nb = min(32u, n - base);
        uint32_t mask = 0u;
        if (mine) {
            mask = nb == 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u);
            if (cull) {
                // scenes of many objects: the 32 records' union box first (one test instead of 32 where no ray of the wave comes near)
                if (any_zero) {   // (rare: the loops with the extra test)
                    if (n > 32u && cull32<true>(KA.sc.cull32, base >> 5, 1u, ox, oy, oz, ix, iy, iz, eo, tl) == 0u) mask = 0u;
                    if (__ballot(mask != 0u) != 0ull) mask = mask ? cull32<true>(KA.sc.cull, base, nb, ox, oy, oz, ix, iy, iz, eo, tl) : 0u;
                } else {
                    if (n > 32u && cull32<false>(KA.sc.cull32, base >> 5, 1u, ox, oy, oz, ix, iy, iz, eo, tl) == 0u) mask = 0u;
                    if (__ballot(mask != 0u) != 0ull) mask = mask ? cull32<false>(KA.sc.cull, base, nb, ox, oy, oz, ix, iy, iz, eo, tl) : 0u;
                }
            }
        }
        while (__ballot(mask != 0u) != 0ull) {   // one chunk of whole ranks per iteration
            SECT64(4);
            if (COUNT && lane == 0u) c_rounds++;
            // ---- deal the pairs out
            slots[lane] = 0u;
            uint32_t used = 0u, n_ranks = 0u;   // wave-uniform
            uint32_t mm = mask;
            for (;;) {
                const uint64_t B = __ballot(mm != 0u);
                const uint32_t c = uint32_t(__popcll(B));
                if (c == 0u || used + c > 64u) break;   // (the first rank always fits)
                if (mm != 0u) {
                    slots[used + mbcnt64(B)] = (lane + 1u) | ((base + uint32_t(__builtin_ctz(mm))) << 8);
                    mm &= mm - 1u;
                }
                used += c;
                n_ranks++;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t info = slots[lane];
            // ---- a slot lane takes its pair's ray from the owner and evaluates the object
            const uint32_t owner = info ? (info & 0xFFu) - 1u : lane;
            const D po = lane_read(o, owner), pd = lane_read(d, owner);
            PairHit ph{kInf, -kInf, 0u};
            if (info != 0u) {
                if (COUNT) c_evals++;
                ph = eval_pair<COUNT>(recs, trecs, info >> 8, po, pd);
            }
            // ---- the owners collect, rank by rank = in scene order
            uint32_t cum = 0u;
            for (uint32_t rk = 0; rk < n_ranks; rk++) {
                SECT64(13);
                const uint64_t B = __ballot(mask != 0u);
                const bool has = mask != 0u;
                const uint32_t src = has ? cum + mbcnt64(B) : lane;
                const double t = lane_read(ph.t, src);
                if (__ballot(has && t < q.t) != 0ull) {   // (the rest of the result only where some record may change)
                    const double bm = lane_read(ph.bm, src);
                    const uint32_t aux = lane_read(ph.aux, src);
                    if (has && !(bm > q.t) && t < q.t) {
                        q.t = t;
                        q.obj = int32_t(base + uint32_t(__builtin_ctz(mask)));
                        q.aux = aux;
                    }
                }
                if (has) mask &= mask - 1u;
                cum += uint32_t(__popcll(B));
            }
        }
    }
}

// The winner's world-space normal: what its Shape::intersect stores in the record -- formed here from the ray, the time
// and `aux` by the same operations on the same operands --, then Transformed::intersect's normalize(normal_transform * n)
// (src/shape.rs:135-136).
template <class RP, class TP>
R64_DEV D hit_normal(RP recs, TP trecs, const Query& q, D o, D d) {
    const auto& r = recs[q.obj];
    D n;
    if (r.kind == SH_CUBE) {
        const double sg = (q.aux & 4u) ? -1.0 : 1.0;
        const uint32_t ax = q.aux & 3u;
        n = mk(ax == 0u ? sg : 0.0, ax == 1u ? sg : 0.0, ax == 2u ? sg : 0.0);
    } else if (r.kind == SH_PLANE) {
        n = -(normalize(ld(r.b))) * (q.aux ? -1.0 : 1.0);
    } else {
        D ol = o, dl = d;
        for (uint32_t f = 0; f < r.n_frames; f++)
            if (KA.sc.fshade[r.frame[f]].has_xf) {
                const FrameRec& F = KA.sc.frames[r.frame[f]];
                const D p = xf_point(F.inv, ol), q2 = xf_dir(F.inv, dl);
                ol = p;
                dl = q2;
            }
        if (r.has_xf) {
            const D p = xf_point(r.inv, ol), q2 = xf_dir(r.inv, dl);
            ol = p;
            dl = q2;
        }
        if (r.kind == SH_SPHERE) {
            n = normalize(ol + q.t * dl);   // src/shape/sphere.rs:40
        } else {   // src/shape/mesh.rs:64-80
            const auto& tr = trecs[q.aux];
            const D d2 = (ol + q.t * dl) - ld(tr.v1);
            const double d20 = dot(d2, ld(tr.d0)), d21 = dot(d2, ld(tr.d1));
            const double v = (tr.d11 * d20 - tr.d01 * d21) / tr.denom, w = (tr.d00 * d21 - tr.d01 * d20) / tr.denom;
            const double u = 1.0 - v - w;
            const TriShade& ts = KA.sc.tshade[q.aux];
            n = normalize(u * ld(ts.n1) + v * ld(ts.n2) + w * ld(ts.n3));
        }
    }
    if (r.has_xf) n = normalize(mul3(KA.sc.shade[q.obj].nrm, n));
    for (uint32_t f = r.n_frames; f != 0u; f--) {   // the groups' own Transformed::intersect, innermost first
        const FrameShade& fs = KA.sc.fshade[r.frame[f - 1u]];
        if (fs.has_xf) n = normalize(mul3(fs.nrm, n));
    }
    return n;
}

// ---------------------------------------------------------------------------- Shape::sample
// of the unit shapes / a mesh in the shape's own space: point v, normal n, pdf p.  `s` is a light's shape: wave-uniform.
template <class S>
R64_DEV void sample_local(const S& s, D target, Rng64& rng, D& v, D& n, double& p) {
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
        const uint32_t idx = s.tri_first + rng.index(s.tri_count);
        // `while u + v > 1 { redraw }`: with u = (2k+1) 2^-24 the sum is exact and u + v > 1 <=> ku + kv >= 2^23, so a
        // rejected pair is never converted (same draws)
        uint32_t ku = rng.r.next() >> 9, kv = rng.r.next() >> 9;
        while (ku + kv >= (1u << 23)) {
            ku = rng.r.next() >> 9;
            kv = rng.r.next() >> 9;
        }
        const double u = double((ku << 1) | 1u) * 0x1p-24, vv = double((kv << 1) | 1u) * 0x1p-24;
        const double w = 1.0 - u - vv;
        const Tri& tr = KA.sc.tris[idx];
        v = u * ld(tr.v1) + vv * ld(tr.v2) + w * ld(tr.v3);
        n = normalize(u * ld(tr.n1) + vv * ld(tr.n2) + w * ld(tr.n3));
        p = KA.sc.tri_pdf[idx];   // (1 / area) / len, area = 0.5 |(v2 - v1) x (v3 - v1)|: the triangle's alone, evaluated at commit
    }
}
template <class S>
R64_DEV void sample_shape(const S& s, D target, Rng64& rng, D& v, D& n, double& p) {
    if (!s.has_xf) return sample_local(s, target, rng, v, n, p);
    // Transformed::sample, src/shape.rs:140-151
    D vl, nl;
    double pl;
    sample_local(s, xf_point(s.inv, target), rng, vl, nl, pl);
    const D new_normal = normalize(mul3(s.nrm, nl));
    const double height = dot(mul3(s.lin, nl), new_normal);
    const double base = s.det / height;
    v = xf_point(s.fwd, vl);
    n = new_normal;
    p = pl / base;
}
// Transformed::sample's second half (src/shape.rs:143-150): the sample of the inner shape carried into the outer space
template <class S>
R64_DEV void transformed_sample_out(const S& s, D& v, D& n, double& p) {
    const D new_normal = normalize(mul3(s.nrm, n));
    const double height = dot(mul3(s.lin, n), new_normal);
    const double base = s.det / height;
    v = xf_point(s.fwd, v);
    n = new_normal;
    p = p / base;
}
// Shape::sample of a light's shape, which may be a `KdTree<Box<dyn Bounded>>` (KdTree::sample, src/kdtree.rs:141-146: a uniformly
// chosen child's sample, its pdf divided by the number of children), nested up to three groups deep, every level possibly
// under its own Transformed.  The light's own record is wave-uniform; the children a lane picks are its own (global loads).
template <class S>
R64_DEV void sample_light_shape(const S& top, D target, Rng64& rng, D& v, D& n, double& p) {
    if (top.kind != SH_GROUP) return sample_shape(top, target, rng, v, n, p);
    const Shape* const pool = KA.sc.lshapes;
    D tgt = top.has_xf ? xf_point(top.inv, target) : target;
    const Shape* c1 = pool + top.tri_first + rng.index(top.tri_count);
    const Shape* c2 = nullptr;
    const Shape* c3 = nullptr;
    const Shape* leaf = c1;
    if (c1->kind == SH_GROUP) {
        if (c1->has_xf) tgt = xf_point(c1->inv, tgt);
        c2 = pool + c1->tri_first + rng.index(c1->tri_count);
        leaf = c2;
        if (c2->kind == SH_GROUP) {
            if (c2->has_xf) tgt = xf_point(c2->inv, tgt);
            c3 = pool + c2->tri_first + rng.index(c2->tri_count);
            leaf = c3;   // (a shape: the commit refuses deeper nesting)
        }
    }
    sample_shape(*leaf, tgt, rng, v, n, p);
    // back out: a group divides the pdf by its number of children, then its own Transformed::sample carries the sample outwards
    if (c3) {
        p = p / double(c2->tri_count);
        if (c2->has_xf) transformed_sample_out(*c2, v, n, p);
    }
    if (c2) {
        p = p / double(c1->tri_count);
        if (c1->has_xf) transformed_sample_out(*c1, v, n, p);
    }
    p = p / double(top.tri_count);
    if (top.has_xf) transformed_sample_out(top, v, n, p);
}
// Light::illuminate for Light::Object, src/light.rs:34-45
template <bool GROUPL, class LT>
R64_DEV void illuminate_object(const LT& L, D pos, Rng64& rng, D& intensity, D& wi, double& dist) {
    D v, n;
    double p;
    if constexpr (GROUPL) sample_light_shape(L.shape, pos, rng, v, n, p);
    else sample_shape(L.shape, pos, rng, v, n, p);
    const D disp = v - pos;
    const double len = length(disp);
    const double cosine = fmax(-dot(disp, n), 0.0) / len;
    const double surface_area = fmax(cosine, 0.0) / (len * len);
    const bool has_color = L.mat.kind <= 1;   // Material::color / emittance, src/material.rs:99-113
    const D col = has_color ? ld(L.mat.albedo) : mk(0, 0, 0);
    const double emit = has_color ? L.mat.emittance : 0.0;
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
// Environment::get_color (src/environment.rs:64-77) / Hdri::get_color + bilinear_sample (:25-52).  (x0 + 1, y0 + 1 are clamped to
// the image: the reference indexes them unclamped and panics or wraps exactly where the weight is zero.)
R64_DEV D env_color(D dir_in) {
    const uint32_t w = KA.sc.hdri_w, h = KA.sc.hdri_h;
    if (w == 0u) return ld(KA.sc.env);
    const D dir = normalize(dir_in);
    const double azimuth = atan2(dir.z, dir.x) + kPi;
    const double polar = acos(dir.y);
    const double x = azimuth / (2.0 * kPi) * double(w - 1u);
    const double y = polar / kPi * double(h - 1u);
    // (`as u32` saturates: negative and NaN -> 0)
    const uint32_t xi = x > 0.0 ? (x < 4294967295.0 ? uint32_t(x) : 4294967295u) : 0u, yi = y > 0.0 ? (y < 4294967295.0 ? uint32_t(y) : 4294967295u) : 0u;
    const uint32_t x0 = min(xi, w - 1u), y0 = min(yi, h - 1u);
    const double ax = x - double(x0), ay = y - double(y0);
    const uint32_t x1 = min(x0 + 1u, w - 1u), y1 = min(y0 + 1u, h - 1u);
    const double* const t = KA.sc.hdri;
    auto texel = [&](uint32_t xx, uint32_t yy) { return ld(t + (size_t(yy) * w + xx) * 3u); };
    auto mix = [](D a, D b, double f) { return a * (1.0 - f) + b * f; };   // glm::mix
    return mix(mix(texel(x0, y0), texel(x1, y0), ax), mix(texel(x0, y1), texel(x1, y1), ax), ay);
}
// Medium::color (src/medium.rs:80-122): hex_color(0xD2B48C) for homogeneous_isotropic; blue below / red above y = 250 for
// colored_glowing_fog (the host passes the colours, src/color.rs:10-15 evaluated in fp64)
R64_DEV D medium_color(D pos) {
    return (KA.sc.medium_kind == 1 && pos.y > 250.0) ? ld(KA.medium_color_hi) : ld(KA.medium_color);
}
// Camera::cast_ray, src/camera.rs:65-82
template <class CP>
R64_DEV void cast_ray(const CP& c, double x, double y, Rng64& rng, D& o, D& d) {
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

// LDS of a block, in doubles: [kColsD columns of 256: acc, P, Q, R][kColsU / 2 columns' worth of dword columns: slab slot,
// sample, end, pixel][ObjRec x kLdsObjs][TriRec x kLdsTris][64 dwords per wave: closest_hit_wave's slots]
static constexpr uint32_t kColsD = 12u, kColsU = 4u;
static constexpr uint32_t kTabBase = (kColsD + kColsU / 2u) * 256u;
static constexpr uint32_t kObjDoubles = sizeof(ObjRec) / 8u, kTriDoubles = sizeof(TriRec) / 8u;
static constexpr uint32_t kSlotBase = kTabBase + kLdsObjs * kObjDoubles + kLdsTris * kTriDoubles;   // 4 waves x 64 dwords: the pairs of a chunk
static constexpr uint32_t kLdsDoubles = kSlotBase + 4u * 32u;
static_assert(sizeof(ObjRec) == 176 && sizeof(TriRec) == 128 && sizeof(CullBox) == 32 && sizeof(FrameRec) == 144, "record sizes");
static_assert(kLdsDoubles * 8u * 4u <= 160u * 1024u, "four blocks per CU");

// GROUPL: some Light::Object is a KdTree group (per-lane child choice, sample_light_shape): an instantiation of its own, as in the fp32
// megakernel -- inlined beside the wave-uniform sampler it cost C3 1.3 % through register allocation alone.
template <bool MEDIUM, bool COUNT, bool LDSTAB, bool GROUPL = false>
__global__ __launch_bounds__(256, R64_WAVES) void render_f64_kernel(const Args a_by_value) {
    (void)a_by_value;   // (read through KA)
    extern __shared__ double lds64[];
    double* const cd = lds64 + threadIdx.x;                                                    // [column * 256]
    uint32_t* const cu = reinterpret_cast<uint32_t*>(lds64 + kColsD * 256u) + threadIdx.x;    // [column * 256]
    enum { C_ACC = 0, C_P = 3, C_Q = 6, C_R = 9 };
    enum { U_SLAB = 0, U_S = 1, U_END = 2, U_XY = 3 };
    const ObjRec* recs = KA.sc.recs;
    const TriRec* trecs = KA.sc.trecs;
    if constexpr (LDSTAB) {
        double* const t0 = lds64 + kTabBase;
        const double* const g0 = reinterpret_cast<const double*>(KA.sc.recs);
        const double* const g1 = reinterpret_cast<const double*>(KA.sc.trecs);
        const uint32_t n0 = KA.sc.n_objects * kObjDoubles, n1 = KA.sc.n_obj_tris * kTriDoubles;
        for (uint32_t i = threadIdx.x; i < n0; i += 256u) t0[i] = g0[i];
        for (uint32_t i = threadIdx.x; i < n1; i += 256u) t0[kLdsObjs * kObjDoubles + i] = g1[i];
        __syncthreads();
        recs = reinterpret_cast<const ObjRec*>(t0);
        trecs = reinterpret_cast<const TriRec*>(t0 + kLdsObjs * kObjDoubles);
    }
    volatile uint32_t* const slots = reinterpret_cast<uint32_t*>(lds64 + kSlotBase) + (threadIdx.x >> 6) * 64u;   // this wave's
    auto ldD = [&](int c) { return mk(cd[c * 256], cd[(c + 1) * 256], cd[(c + 2) * 256]); };
    auto stD = [&](int c, D v) { cd[c * 256] = v.x; cd[(c + 1) * 256] = v.y; cd[(c + 2) * 256] = v.z; };

    Rng64 rng;
    rng.r.s0 = rng.r.s1 = rng.r.s2 = rng.r.s3 = 0;
    D ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    uint32_t depth = 0;
    bool alive = true, have_item = false, need_path = true, item_done = true;
    bool parked = false;            // the lane's path waits at a surface event (below)
    Query pq{kInf, -1, 0u};         // ... with this record
    bool drained = false, first_batch = true;                                   // wave-uniform
    uint32_t pool_next = 0, pool_end = 0, pool_chunk = 0, pool_x0 = 0, pool_y0 = 0;   // wave-uniform: the batch of work items at hand
    uint32_t c_rays = 0, c_hits = 0, c_self = 0, c_shadow = 0, c_pass = 0, c_near = 0, c_samples = 0, c_vertices = 0, c_evals = 0,
             c_rounds = 0, c_trips = 0, c_live = 0;

    for (;;) {
        // ---- work distribution (wave-convergent), as in the fp32 megakernel: a wave draws batches of 64 items -- one 8 x 8
        // pixel block of one chunk -- from the global counter with one atomic and hands them to the lanes that finished
        // theirs through a ballot / mbcnt prefix
        bool want = alive && need_path && item_done;
        const uint32_t n_want = uint32_t(__popcll(__ballot(want)));
        if (n_want != 0u && (n_want >= KA.pull_batch || __ballot(alive && !want) == 0ull)) {
            SECT64(0);
            const auto& ka = KA;
            if (want && have_item) {
                double* const sl = ka.slab + size_t(cu[U_SLAB * 256]) * 4u;
                const D acc = ldD(C_ACC);
                reinterpret_cast<double2*>(sl)[0] = make_double2(acc.x, acc.y);
                reinterpret_cast<double2*>(sl)[1] = make_double2(acc.z, 0.0);
                have_item = false;
            }
            for (;;) {
                const uint64_t m = __ballot(want);
                if (m == 0) break;
                if (pool_next == pool_end) {
                    unsigned long long base = ~0ull;
                    if (first_batch) {   // a wave's first batch is its own index (the host starts the counter behind them)
                        base = (unsigned long long)(blockIdx.x * 4u + (threadIdx.x >> 6)) * 64ull;
                        first_batch = false;
                    } else if (!drained && (threadIdx.x & 63u) == 0) {
                        base = atomicAdd(ka.queue, 64ull);
                    }
                    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(base));
                    const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(base >> 32));
                    if (hi != 0 || lo >= ka.n_items) {
                        drained = true;   // queue exhausted: the waiting lanes retire
                        if (want) alive = false;
                        break;
                    }
                    pool_next = lo;
                    pool_end = min(lo + 64u, ka.n_items);
                    pool_chunk = lo / ka.n_owned;
                    const uint32_t p0 = lo - pool_chunk * ka.n_owned;
                    const uint32_t tile = ka.tiles[p0 >> 10], sb = (p0 & 1023u) >> 6;
                    const uint32_t ty = tile / ka.tiles_x, tx = tile - ty * ka.tiles_x;
                    pool_x0 = __builtin_amdgcn_readfirstlane(tx * 32u + (sb & 3u) * 8u);
                    pool_y0 = __builtin_amdgcn_readfirstlane(ty * 32u + (sb >> 2) * 8u);
                    pool_chunk = __builtin_amdgcn_readfirstlane(pool_chunk);
                }
                const uint32_t take = min(uint32_t(__popcll(m)), pool_end - pool_next);
                const uint32_t rank = mbcnt64(m);
                const uint32_t item = pool_next + rank;
                const bool got = want && rank < take;
                pool_next += take;
                if (got) {
                    const uint32_t l = item & 63u;
                    const uint32_t x = pool_x0 + (l & 7u), y = pool_y0 + (l >> 3);
                    if (x < ka.width && y < ka.height) {   // slots of clipped tiles lie outside the image
                        want = false;
                        have_item = true;
                        item_done = false;
                        cu[U_SLAB * 256] = item;
                        stD(C_ACC, mk(0, 0, 0));
                        const uint32_t s0 = pool_chunk * ka.chunk_spp;
                        cu[U_S * 256] = s0;
                        cu[U_END * 256] = min(s0 + ka.chunk_spp, ka.iterations);
                        cu[U_XY * 256] = x | (y << 16);
                    }
                }
            }
        }
        if (need_path && alive && !item_done) {   // the item's next sample: src/renderer.rs:174-181
            SECT64(1);
            const auto& ka = KA;
            const uint32_t s = cu[U_S * 256], xy = cu[U_XY * 256], x = xy & 0xFFFFu, y = xy >> 16;
            rng.r.seed(ka.seed_mixed, y * ka.width + x, ka.sample_offset + s);
            // (2 * x + 1 and 2 * (h - y) - 1 in u32, like the reference)
            const double dim = ka.dim;
            const double xn = (double(2u * x + 1u) - double(ka.width)) / dim;
            const double yn = (double(2u * (ka.height - y) - 1u) - double(ka.height)) / dim;
            const double dx = rng.range(-1.0 / dim, 1.0 / dim);
            const double dy = rng.range(-1.0 / dim, 1.0 / dim);
            cast_ray(ka.cam, xn + dx, yn + dy, rng, ro, rd);
            depth = 0;
            stD(C_P, mk(0, 0, 0));
            stD(C_Q, mk(1, 1, 1));
            if (!MEDIUM) stD(C_R, mk(kInf, kInf, kInf));
            cu[U_S * 256] = s + 1u;
            item_done = s + 1u >= cu[U_END * 256];
            need_path = false;
            if (COUNT) c_samples++;
        }
        if (__ballot(alive) == 0ull) break;
        if (COUNT) {
            if (mbcnt64(__ballot(true)) == 0u) c_trips++;
            if (alive && !need_path) c_live++;
        }
        // ---- one path vertex = one trace_ray invocation (src/renderer.rs:187-322).  The radiance of the path is
        // min(P + Q x, R) per channel, x = what the rest of the path returns (closed under x -> E + min(k x, 100), :308-313;
        // in a medium there is no clamp, R = inf).
        // The vertex's closest-hit queries -- the path's own, then one per object light -- go through ONE copy of the query
        // code, which every lane of the wave enters (closest_hit_wave: lanes without a query work for the others); `sub`
        // counts the queries, wave-uniformly.
        // In a medium most events are medium events, and the code of a surface event -- the winner's normal, the BSDF, the
        // direction sampler with its libm calls: a third of a trip's instructions -- would run in nearly every trip for the three
        // or four lanes of a wave that need it.  A lane whose closest-hit query ends in a surface event therefore PARKS: it keeps
        // its ray and the query's record and sits out until `surf_batch` lanes of the wave are parked or no lane has anything
        // else to do; then they go through the surface code together.  Every lane still draws its own numbers in its own
        // order: the frame does not depend on the threshold (tests/test_gpu_epsilon.py).
        const bool live = alive && !need_path;   // (a lane without a path waits for the wave's next item hand-out)
        const bool fresh = live && !parked;      // a new vertex starts here
        if (COUNT && fresh) c_vertices++;
        const double sigma_t = KA.sc.absorption + KA.sc.scattering;
        double dmed = kInf;
        if (MEDIUM && fresh) {
            SECT64(2);
            dmed = -log(rng.range(0.0, 1.0)) / sigma_t;   // Medium::sample_d, src/medium.rs:133-146
        }
        const bool limits = COUNT ? KA.cull == 2u : KA.cull != 0u;   // (cull = 2: the counters build keeps the limits too -- its counts are then the schedule's, not the reference's)
        // A hit beyond the sampled distance cannot change the event (dmed < t, or a miss with dmed < 400, is a medium event
        // either way, :197-243): the search may end there.  (The counters build searches everything: its counts are the
        // reference's.)
        // the query at hand: origin ro (the vertex once the event is known), direction rd (towards the light sample for a
        // shadow query), and the distance beyond which hits do not matter
        double qlim = (MEDIUM && limits && dmed < 400.0) ? dmed : kInf;
        D E = mk(0, 0, 0), T = mk(0, 0, 0), n = mk(0, 1, 0), wo = mk(0, 0, 0);
        double dist = 0.0;
        int32_t hobj = -1;          // the object of a surface event
        bool ev_medium = false;
        bool active = fresh;        // false: no path, parked, or the path ended at this vertex without an event (miss)
        auto surface_event = [&](const Query& h) {   // :207-216 in a medium, :289-299 without
            SECT64(16);
            hobj = h.obj;
            n = hit_normal(recs, trecs, h, ro, rd);
            ro = ro + h.t * rd;
            wo = -normalize(rd);
            const Mat& mat = KA.sc.shade[hobj].mat;
            E = depth == 0 ? mat_emit(mat) * mat_color(mat) : mk(0, 0, 0);
        };
        uint32_t li = 0;            // wave-uniform: next light to look at
        const uint32_t n_lights = KA.sc.n_lights;
        for (uint32_t sub = 0;; sub++) {
            Query q;
            closest_hit_wave<COUNT>(recs, trecs, slots, active, ro, rd, qlim, q, c_evals, c_rounds);
            if (active) {
                SECT64(14);
                if (sub == 0) {
                    const bool hit = q.obj >= 0;
                    if (COUNT) {
                        c_rays++;
                        if (hit) {
                            c_hits++;
                            const double m = fmax(fmax(fabs(ro.x), fabs(ro.y)), fabs(ro.z));
                            if (q.t < 1e-9 * (1.0 + m)) c_self++;   // diagnostic: a hit on the surface the ray starts on
                        }
                    }
                    ev_medium = MEDIUM && dmed < (hit ? q.t : 400.0);   // :197-243 (`d >= h.time` is a surface event)
                    if (!ev_medium && !hit) {   // :198-206 (in a medium the background counts only beyond 400), :288
                        const D env = (!MEDIUM || dmed >= 400.0) ? env_color(rd) : mk(0, 0, 0);
                        const D v = ldD(C_P) + ldD(C_Q) * env;
                        stD(C_ACC, ldD(C_ACC) + (MEDIUM ? v : vmin(v, ldD(C_R))));
                        need_path = true;
                        active = false;
                    } else if (ev_medium) {   // :243-255
                        SECT64(15);
                        ro = ro + dmed * rd;
                        const double emm = KA.sc.medium_kind == 1 ? 10.0 : 0.0;
                        E = depth == 0 ? emm * medium_color(ro) : mk(0, 0, 0);
                    } else if (MEDIUM) {   // a surface event: waits for company (see above)
                        parked = true;
                        pq = q;
                        active = false;
                    } else {
                        surface_event(q);
                    }
                } else {
                    SECT64(17);
                    // the shadow test (:339-348, :386-396): the closest hit along wi lies at the sampled distance
                    const double miss = fabs(q.t - dist);
                    const bool visible = q.obj >= 0 && miss < kEps;
                    if (COUNT) {
                        c_rays++;
                        c_shadow++;
                        if (q.obj >= 0) {
                            c_hits++;
                            const double m = fmax(fmax(fabs(ro.x), fabs(ro.y)), fabs(ro.z));
                            if (q.t < 1e-9 * (1.0 + m)) c_self++;
                            if (miss < kEps) c_pass++;
                            else if (miss < 1e-6 * dist) c_near++;   // diagnostic: the light's own surface, missed by rounding
                        }
                    }
                    if (visible) E = E + T;
                }
            }
            if (MEDIUM && sub == 0) {
                const uint64_t pk = __ballot(parked);
                if (pk != 0ull && (uint32_t(__popcll(pk)) >= KA.surf_batch || __ballot(active) == 0ull)) {
                    if (parked) {
                        surface_event(pq);
                        parked = false;
                        active = true;
                    }
                }
            }
            // sample_lights_for_media :325-359 / sample_lights :362-409; lights in scene order fix the draw order.  Ambient
            // lights add their term; the next object light is sampled and its shadow query becomes the query at hand.
            bool more = false;
            while (li < n_lights) {
                const auto& L = uniform_ref(&KA.sc.lights[li]);
                li++;
                if (L.kind == LT_AMBIENT) {
                    if (active) E = E + ld(L.color) * (ev_medium ? medium_color(ro) : mat_color(KA.sc.shade[hobj < 0 ? 0 : hobj].mat));
                } else if (L.kind == LT_OBJECT) {
                    if (active) {
                        SECT64(18);
                        D I, wi;
                        illuminate_object<GROUPL>(L, ro, rng, I, wi, dist);
                        // the light's term of E if it proves visible
                        if (ev_medium) {
                            SECT64(19);
                            const double scat = KA.sc.scattering;
                            const double phase = KA.sc.medium_kind == 1 ? 1.0 / 4.0 * kPi : 1.0 / (4.0 * kPi);   // (sic, src/medium.rs:113)
                            T = ((scat / sigma_t) * (I * medium_color(ro))) * phase;
                        } else {
                            SECT64(20);
                            T = (bsdf(KA.sc.shade[hobj].mat, n, wo, wi) * I) * dot(wi, n);
                        }
                        rd = wi;
                        qlim = limits ? dist : kInf;
                    }
                    more = true;
                    break;
                }
                // Point / Directional: illuminate draws nothing and the test |hit - dist| < 1e-12 can never pass
                // (dist = the light's position / +inf, src/light.rs:26-33)
            }
            if (!more || __ballot(active) == 0ull) break;
        }
        if (!active) continue;

        D k = mk(0, 0, 0), wi_next = mk(0, 0, 1);
        bool cont = false;
        if (ev_medium) {   // :262-281
            SECT64(21);
            if (rng.uniform() < 0.8) {
                const double ax = rng.range(-1.0, 1.0), ay = rng.range(-1.0, 1.0), az = rng.range(-1.0, 1.0);
                wi_next = normalize(mk(ax, ay, az));   // Medium::sample_ph, src/medium.rs:87-93
                const double scat = KA.sc.scattering;
                const double phase = KA.sc.medium_kind == 1 ? 1.0 / 4.0 * kPi : 1.0 / (4.0 * kPi);
                k = ((((scat / sigma_t) / phase) * medium_color(ro)) * phase) / 0.8;   // (scat/ext) x / ph_p . color * phase / rr_p, ph_p == phase
                cont = true;
            }
        } else {
            SECT64(22);
            const bool go = MEDIUM ? (rng.uniform() < 0.8) : (depth < KA.max_bounces);   // :222 / :301
            if (go) {
                double pdf;
                const Mat& mat = KA.sc.shade[hobj].mat;
                if (sample_f(mat, n, wo, rng, wi_next, pdf)) {
                    const D f = bsdf(mat, n, wo, wi_next);
                    k = ((1.0 / (MEDIUM ? pdf * 0.8 : pdf)) * f) * fabs(dot(wi_next, n));
                    cont = true;
                }
            }
        }
        SECT64(23);
        const D Q = ldD(C_Q);
        const D P = ldD(C_P) + Q * E;
        if (!cont) {   // (a path whose weight has become zero goes on, as the reference's recursion does: same rays, same draws)
            stD(C_ACC, ldD(C_ACC) + (MEDIUM ? P : vmin(P, ldD(C_R))));
            need_path = true;
            continue;
        }
        stD(C_P, P);
        if (!MEDIUM) stD(C_R, vmin(ldD(C_R), P + kFireflyClamp * Q));   // FIREFLY_CLAMP, :311-313
        stD(C_Q, Q * k);
        rd = wi_next;   // (ro is the vertex already)
        depth++;
    }
    if (COUNT) {
        const uint32_t v[12] = {c_rays, c_hits, c_self, c_shadow, c_pass, c_near, c_samples, c_vertices, c_evals, c_rounds, c_trips, c_live};
        for (int i = 0; i < 12; i++)
            if (v[i]) atomicAdd(&KA.counters[i], (unsigned long long)v[i]);
    }
}

// The frame from the slab: a pixel's chunk sums added in chunk order, / iterations * 2^EV (src/renderer.rs:183).
__global__ __launch_bounds__(256) void resolve_f64_kernel(const Args a, double scale, double* __restrict__ out) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= a.n_owned) return;
    const uint32_t tl = p >> 10, within = p & 1023u, sb = within >> 6, l = within & 63u;
    const uint32_t tile = a.tiles[tl], ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const uint32_t x = tx * 32u + (sb & 3u) * 8u + (l & 7u), y = ty * 32u + (sb >> 2) * 8u + (l >> 3);
    if (x >= a.width || y >= a.height) return;
    double r = 0.0, g = 0.0, b = 0.0;
    for (uint32_t c = 0; c < a.n_chunks; c++) {
        const double2* const sl = reinterpret_cast<const double2*>(a.slab + (size_t(c) * a.n_owned + p) * 4u);
        const double2 v0 = sl[0], v1 = sl[1];
        r += v0.x;
        g += v0.y;
        b += v1.x;
    }
    const size_t o = (size_t(y) * a.width + x) * 3;
    const double inv = scale / double(a.iterations);
    out[o] = r * inv;
    out[o + 1] = g * inv;
    out[o + 2] = b * inv;
}


// ============================================================================ photon mapping with the reference's epsilons
// The parts of src/photon.rs whose outcome hangs on the epsilon policy run here, in fp64 and through closest_hit_wave: the
// shooting pass (a photon that leaves a surface meets that surface again whenever rounding puts the hit beyond t_min = 1e-12,
// and is then scattered a second time at the same spot, src/photon.rs:803-946 with src/renderer.rs:420) and the surface
// estimate's visibility rays (a gathered photon counts unless `len > hit.time`, :357-361: the ray's own far end decides that by
// its last bits).  The maps over the records (LBVH, radii), the k-nearest selection and the volume estimates have no epsilon in
// them and stay photon.hip's, in fp32: its camera pass hands each sample's selection over (SurfArgs64::emit).
#define KS (*rptg::kernarg_args<ShootArgs64>())
#define KQ (*rptg::kernarg_args<SurfArgs64>())
static constexpr uint32_t kPhSlotBase = kLdsObjs * kObjDoubles + kLdsTris * kTriDoubles;   // LDS of these kernels: [tables][4 x 64 slot dwords]
static constexpr uint32_t kPhLdsDoubles = kPhSlotBase + 4u * 32u;

template <bool LDSTAB>
R64_DEV void stage_tables(double* t0, const ObjRec*& recs, const TriRec*& trecs) {
    recs = KA.sc.recs;
    trecs = KA.sc.trecs;
    if constexpr (LDSTAB) {
        const double* const g0 = reinterpret_cast<const double*>(KA.sc.recs);
        const double* const g1 = reinterpret_cast<const double*>(KA.sc.trecs);
        const uint32_t n0 = KA.sc.n_objects * kObjDoubles, n1 = KA.sc.n_obj_tris * kTriDoubles;
        for (uint32_t i = threadIdx.x; i < n0; i += 256u) t0[i] = g0[i];
        for (uint32_t i = threadIdx.x; i < n1; i += 256u) t0[kLdsObjs * kObjDoubles + i] = g1[i];
        __syncthreads();
        recs = reinterpret_cast<const ObjRec*>(t0);
        trecs = reinterpret_cast<const TriRec*>(t0 + kLdsObjs * kObjDoubles);
    }
}

// One photon per lane; a lane whose photon is absorbed takes the next one of the wave's batch (64 per atomic).  WRITE = false
// counts the records of every photon; the write pass retraces the same chains (same streams, and closest_hit_wave's record does
// not depend on which rays share the wave) and stores them at the prefix sums of the counts: the arrays are in shooting order.
template <bool MEDIUM, bool WRITE, bool LDSTAB>
__global__ __launch_bounds__(256, R64_WAVES) void photon_shoot_f64_kernel(const ShootArgs64 by_value) {
    (void)by_value;
    extern __shared__ double lds64[];
    const ObjRec* recs;
    const TriRec* trecs;
    stage_tables<LDSTAB>(lds64, recs, trecs);
    volatile uint32_t* const slots = reinterpret_cast<uint32_t*>(lds64 + kPhSlotBase) + (threadIdx.x >> 6) * 64u;
    const double sigma_t = KA.sc.absorption + KA.sc.scattering;
    Rng64 rng, thin;
    rng.r.s0 = rng.r.s1 = rng.r.s2 = rng.r.s3 = 0;
    thin.r = rng.r;
    D ro = mk(0, 0, 0), rd = mk(0, 0, 1), power = mk(0, 0, 0);
    uint32_t ns = 0, nv = 0, os = 0, ov = 0;
    uint64_t idx = 0;
    bool have = false;
    uint64_t pool_next = 0, pool_end = 0;   // wave-uniform
    bool drained = false;
    uint32_t ce = 0, cr = 0;
    for (;;) {
        // ---- hand-out
        bool want = !have;
        while (!drained) {
            const uint64_t m = __ballot(want);
            if (m == 0ull) break;
            if (pool_next == pool_end) {
                unsigned long long base = 0ull;
                if ((threadIdx.x & 63u) == 0u) base = atomicAdd(KA.queue, 64ull);
                const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(base));
                const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(base >> 32));
                const uint64_t b = (uint64_t(hi) << 32) | lo;
                if (b >= KS.n_photons) {
                    drained = true;
                    break;
                }
                pool_next = b;
                pool_end = b + 64u < KS.n_photons ? b + 64u : KS.n_photons;
            }
            const uint32_t take = min(uint32_t(__popcll(m)), uint32_t(pool_end - pool_next));
            const uint32_t rank = mbcnt64(m);
            if (want && rank < take) {   // shoot_photon, src/photon.rs:724-760
                idx = pool_next + rank;
                const uint64_t g = KS.first_photon + idx;
                rng.r.seed(KA.seed_mixed, uint32_t(g), 0x80000000u + uint32_t(g >> 32));
                thin.r.seed(KA.seed_mixed, uint32_t(g), 0xC0000000u + uint32_t(g >> 32));   // the thinning draws' side stream
                const Light& L = KA.sc.lights[KS.light_index];
                D v, n;
                double p;
                sample_light_shape(L.shape, mk(0, 0, 0), rng, v, n, p);   // :733-734 (the target is a dummy)
                const double phi = 2.0 * kPi * rng.uniform();
                const double theta = acos(1.0 - rng.uniform());
                const D dir = mk(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
                ro = v;
                rd = rotate_from_y(n, dir, true);
                power = KS.power * mat_color(L.mat);
                ns = nv = 0;
                if (WRITE) {
                    os = KS.off_s[idx];
                    ov = KS.off_v[idx];
                }
                have = true;
                want = false;
            }
            pool_next += take;
        }
        if (__ballot(have) == 0ull) break;
        // ---- one segment of every lane's chain: trace_photon, :803-946
        double dmed = kInf;
        if (MEDIUM && have) dmed = -log(rng.range(0.0, 1.0)) / sigma_t;   // Medium::sample_d (the closest-hit query draws nothing)
        Query q;
        closest_hit_wave<false>(recs, trecs, slots, have, ro, rd, dmed, q, ce, cr);   // (a hit beyond the sampled distance changes nothing)
        if (!have) continue;
        const bool hit = q.obj >= 0;
        const D wo = -normalize(rd);
        bool done = false;
        if (MEDIUM && (!hit || dmed < q.t)) {   // trace_in_volume, :879-914
            const D x = ro + dmed * rd;
            const D mcol = medium_color(x);
            const double scat = KA.sc.scattering;
            const bool beams = KS.kind == 2u;
            const bool keep = !beams || thin.uniform() < 0.001;   // :779-787
            if (keep) {
                if (WRITE) {
                    const double boost = beams ? 1.0 / 0.001 : 1.0;
                    PhotonRec32 r;
                    r.pos_r[0] = float(x.x); r.pos_r[1] = float(x.y); r.pos_r[2] = float(x.z); r.pos_r[3] = beams ? 3.f : 0.f;
                    const D dv = beams ? ro : wo;   // beams: where the beam starts
                    r.dir[0] = float(dv.x); r.dir[1] = float(dv.y); r.dir[2] = float(dv.z); r.dir[3] = 0.f;
                    r.pow[0] = float(boost * power.x); r.pow[1] = float(boost * power.y); r.pow[2] = float(boost * power.z); r.pow[3] = 0.f;
                    KS.vol[ov + nv] = r;
                }
                nv++;
            }
            if (rng.uniform() < scat / sigma_t) {
                const double ax = rng.range(-1.0, 1.0), ay = rng.range(-1.0, 1.0), az = rng.range(-1.0, 1.0);
                const double phase = KA.sc.medium_kind == 1 ? 1.0 / 4.0 * kPi : 1.0 / (4.0 * kPi);
                power = ((((power * mcol) * scat) / sigma_t) * phase) / phase;   // attenuated * phase / ph_p
                ro = x;
                rd = normalize(mk(ax, ay, az));
            } else {
                done = true;
            }
        } else if (!hit) {
            done = true;
        } else {   // trace_on_surface, :811-875
            const D n = hit_normal(recs, trecs, q, ro, rd);
            const D x = ro + q.t * rd;
            const Mat& mat = KA.sc.shade[q.obj].mat;
            const double p_d = 0.7;
            D wi = mk(0, 0, 1);
            double pdf = 1.0;
            if (!(rng.uniform() < p_d) || !sample_f(mat, n, wo, rng, wi, pdf)) {
                done = true;   // absorbed, or total internal reflection: nothing is stored
            } else {
                const D f = bsdf(mat, n, wo, wi);
                const double cw = dot(wi, n);
                const double cosine_term = cw > 0.0 ? cw : 1.0;
                if (mat.kind <= 1) {   // !is_mirror(), src/material.rs:135-141
                    if (WRITE) {
                        PhotonRec32 r;
                        r.pos_r[0] = float(x.x); r.pos_r[1] = float(x.y); r.pos_r[2] = float(x.z); r.pos_r[3] = 0.f;
                        r.dir[0] = float(wo.x); r.dir[1] = float(wo.y); r.dir[2] = float(wo.z); r.dir[3] = 0.f;
                        r.pow[0] = float(power.x); r.pow[1] = float(power.y); r.pow[2] = float(power.z); r.pow[3] = 0.f;
                        KS.surf[os + ns] = r;
                        double* const p64 = KS.pos64 + size_t(os + ns) * 3u;
                        p64[0] = x.x; p64[1] = x.y; p64[2] = x.z;
                    }
                    ns++;
                }
                power = (((power * f) * cosine_term) / pdf) / p_d;
                ro = x;
                rd = wi;
            }
        }
        if (done) {
            if (!WRITE) {
                KS.cnt_s[idx] = ns;
                KS.cnt_v[idx] = nv;
            }
            have = false;
        }
    }
}

// One wave per (pixel, 64 samples of it), dealt out statically.  (With a work counter drawn from by lane 0, as in the other persistent
// kernels, this one was not a function of its inputs: ~20 pixels of 4,096 changed from run to run, always items a wave took after its
// first, written by lane 0 alone -- the cause was not isolated; tools/eps_photon_determinism.py is the check.)  Every lane retraces its sample's camera ray with the reference's arithmetic, then the wave
// goes through the lanes' gathered photons rank by rank -- one visibility query per lane and rank (closest_hit_wave; the search may
// end at the query point: only a hit closer than `len` blocks).  The pixel's partial sum: per lane in rank order, then the lanes
// in a fixed butterfly order.
template <bool MEDIUM, bool LDSTAB>
__global__ __launch_bounds__(256, R64_WAVES) void photon_surface_f64_kernel(const SurfArgs64 by_value) {
    (void)by_value;
    extern __shared__ double lds64[];
    const ObjRec* recs;
    const TriRec* trecs;
    stage_tables<LDSTAB>(lds64, recs, trecs);
    volatile uint32_t* const slots = reinterpret_cast<uint32_t*>(lds64 + kPhSlotBase) + (threadIdx.x >> 6) * 64u;
    const uint32_t lane = threadIdx.x & 63u;
    const double sigma_t = KA.sc.absorption + KA.sc.scattering;
    uint32_t ce = 0, cr = 0;
    // (items are dealt out statically, wave by wave: neighbouring pixels cost about the same)
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;
    for (uint32_t lo = wave_id; lo < KA.n_items; lo += n_waves) {
        const uint32_t n_owned = KA.n_owned;
        const uint32_t g = lo / n_owned, p = lo - g * n_owned;
        const uint32_t within = p & 1023u, sb = within >> 6, l = within & 63u;
        const uint32_t tile = KA.tiles[p >> 10], ty = tile / KA.tiles_x, tx = tile - ty * KA.tiles_x;
        const uint32_t x = tx * 32u + (sb & 3u) * 8u + (l & 7u), y = ty * 32u + (sb >> 2) * 8u + (l >> 3);
        if (x >= KA.width || y >= KA.height) continue;   // slots of clipped tiles lie outside the image (wave-uniform)
        const uint32_t s = g * 64u + lane;
        const bool active = s < KA.iterations;
        Rng64 rng;
        rng.r.seed(KA.seed_mixed, y * KA.width + x, KA.sample_offset + s);
        const double dim = KA.dim;
        const double xn = (double(2u * x + 1u) - double(KA.width)) / dim;            // src/renderer.rs:174-176
        const double yn = (double(2u * (KA.height - y) - 1u) - double(KA.height)) / dim;
        const double dx = rng.range(-1.0 / dim, 1.0 / dim);
        const double dy = rng.range(-1.0 / dim, 1.0 / dim);
        D ro, rd;
        cast_ray(KA.cam, xn + dx, yn + dy, rng, ro, rd);
        Query h;
        closest_hit_wave<false>(recs, trecs, slots, active, ro, rd, kInf, h, ce, cr);
        bool surf = active && h.obj >= 0;
        double scale = 1.0;
        if (MEDIUM && active) {
            if (KQ.kind == 0u) {   // the point x point estimate draws the distance first, :384-438
                const double d = -log(rng.range(0.0, 1.0)) / sigma_t;
                if (!surf || d < h.t) surf = false;
                else scale = exp(-(sigma_t * h.t)) / (1.0 - (1.0 - exp(-(sigma_t * d))));   // transmittance / (1 - cdf)
            } else if (surf) {
                scale = exp(-(sigma_t * h.t));   // :610-611
            }
        }
        D n = mk(0, 1, 0), wo = mk(0, 0, 1), xw = ro, color = mk(0, 0, 0);
        int32_t hobj = 0;
        uint32_t cnt = 0;
        const uint32_t stride = KA.iterations;
        const uint32_t* const em = KQ.emit + size_t(p) * (KQ.K + 2u) * stride + s;
        if (surf) {
            n = hit_normal(recs, trecs, h, ro, rd);
            xw = ro + h.t * rd;
            wo = -normalize(rd);
            hobj = h.obj;
            const Mat& mat = KA.sc.shade[hobj].mat;
            color = mat_emit(mat) * mat_color(mat);
            cnt = em[size_t(KQ.K) * stride];
        }
        uint32_t cmax = cnt;
        for (uint32_t off = 32u; off != 0u; off >>= 1) cmax = max(cmax, lane_read(cmax, lane ^ off));
        cmax = __builtin_amdgcn_readfirstlane(cmax);
        for (uint32_t k = 0; k < cmax; k++) {
            const bool has = k < cnt;
            D po = xw, pdir = mk(0, 1, 0), ppow = mk(0, 0, 0), dirn = mk(0, 0, 1);
            double len = 0.0;
            if (has) {
                const PhotonRec32& ph = KQ.s_ph[em[size_t(k) * stride]];
                const double* const p64 = KQ.pos64 + size_t(__float_as_uint(ph.dir[3])) * 3u;
                po = mk(p64[0], p64[1], p64[2]);
                pdir = mk(double(ph.dir[0]), double(ph.dir[1]), double(ph.dir[2]));
                ppow = mk(double(ph.pow[0]), double(ph.pow[1]), double(ph.pow[2]));
                const D disp = xw - po;
                len = length(disp);
                dirn = normalize(disp);
            }
            Query v{kInf, -1, 0u};
            if (!(KQ.skip & 4096u)) closest_hit_wave<false>(recs, trecs, slots, has, po, dirn, (KQ.skip & 16384u) ? kInf : len, v, ce, cr);   // (diagnostic: 16384 = no search limit)
            if (has && !(v.obj >= 0 && len > v.t)) {   // :357-361
                const double c = fmin(fmax(dot(pdir, n), 0.0), 1.0);
                color = color + (bsdf(KA.sc.shade[hobj].mat, n, wo, pdir) * ppow) * c;
            }
        }
        if (surf) {
            const double max_d2 = cnt ? double(__uint_as_float(em[size_t(KQ.K + 1u) * stride])) : 1.0;
            color = (color * (1.0 / (kPi * max_d2))) * scale;
        }
        if (!surf) color = mk(0, 0, 0);
        for (uint32_t off = 32u; off != 0u; off >>= 1) {
            const D o = lane_read(color, lane ^ off);
            color = color + o;
        }
        if (lane == 0u) {
            double* const sl = KA.slab + size_t(lo) * 4u;
            reinterpret_cast<double2*>(sl)[0] = make_double2(color.x, color.y);
            reinterpret_cast<double2*>(sl)[1] = make_double2(color.z, 0.0);
        }
    }
}

// The frame of that camera pass, slice by slice: the fp32 kernel's partial sums (volume estimate, background; [chunk][pixel] float4) plus
// this file's (surface estimate; [group][pixel]), / samples of the whole call * 2^EV, added to what the earlier slices left.
__global__ __launch_bounds__(256) void resolve_photon_f64_kernel(const Args a, const float4* __restrict__ slab32, uint32_t n_chunks32,
                                                                 double scale_over_total, int accumulate, double* __restrict__ out) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= a.n_owned) return;
    const uint32_t tl = p >> 10, within = p & 1023u, sb = within >> 6, l = within & 63u;
    const uint32_t tile = a.tiles[tl], ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const uint32_t x = tx * 32u + (sb & 3u) * 8u + (l & 7u), y = ty * 32u + (sb >> 2) * 8u + (l >> 3);
    if (x >= a.width || y >= a.height) return;
    double r = 0.0, g = 0.0, b = 0.0;
    for (uint32_t c = 0; c < n_chunks32; c++) {
        const float4 v = slab32[size_t(c) * a.n_owned + p];
        r += double(v.x);
        g += double(v.y);
        b += double(v.z);
    }
    for (uint32_t c = 0; c < a.n_chunks; c++) {
        const double2* const sl = reinterpret_cast<const double2*>(a.slab + (size_t(c) * a.n_owned + p) * 4u);
        const double2 v0 = sl[0], v1 = sl[1];
        r += v0.x;
        g += v0.y;
        b += v1.x;
    }
    const size_t o = (size_t(y) * a.width + x) * 3;
    out[o] = (accumulate ? out[o] : 0.0) + r * scale_over_total;
    out[o + 1] = (accumulate ? out[o + 1] : 0.0) + g * scale_over_total;
    out[o + 2] = (accumulate ? out[o + 2] : 0.0) + b * scale_over_total;
}

}  // namespace rpt64

namespace rptg {
template <bool M, bool C>
static hipError_t launch_f64_t(const rpt64::Args& a, int n_blocks, hipStream_t stream) {
    const size_t lds = size_t(rpt64::kLdsDoubles) * 8u;
    const bool tab = a.sc.n_objects <= rpt64::kLdsObjs && a.sc.n_obj_tris <= rpt64::kLdsTris;
    if (a.group_lights) {
        if (tab) hipLaunchKernelGGL((rpt64::render_f64_kernel<M, C, true, true>), dim3(n_blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((rpt64::render_f64_kernel<M, C, false, true>), dim3(n_blocks), dim3(256), lds, stream, a);
    } else {
        if (tab) hipLaunchKernelGGL((rpt64::render_f64_kernel<M, C, true, false>), dim3(n_blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((rpt64::render_f64_kernel<M, C, false, false>), dim3(n_blocks), dim3(256), lds, stream, a);
    }
    return hipGetLastError();
}
hipError_t launch_render_f64(const rpt64::Args& a, int n_blocks, hipStream_t stream) {
    if (!a.n_items) return hipSuccess;
    const bool m = a.sc.has_medium != 0, c = a.counters != nullptr;
    if (m) return c ? launch_f64_t<true, true>(a, n_blocks, stream) : launch_f64_t<true, false>(a, n_blocks, stream);
    return c ? launch_f64_t<false, true>(a, n_blocks, stream) : launch_f64_t<false, false>(a, n_blocks, stream);
}
hipError_t launch_resolve_f64(const rpt64::Args& a, double scale, double* d_out, hipStream_t stream) {
    if (!a.n_owned) return hipSuccess;
    hipLaunchKernelGGL(rpt64::resolve_f64_kernel, dim3((a.n_owned + 255u) / 256u), dim3(256), 0, stream, a, scale, d_out);
    return hipGetLastError();
}
hipError_t render_f64_occupancy(bool medium, int* blocks_per_cu) {
    const size_t lds = size_t(rpt64::kLdsDoubles) * 8u;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(
        blocks_per_cu, medium ? (const void*)rpt64::render_f64_kernel<true, false, true, false> : (const void*)rpt64::render_f64_kernel<false, false, true, false>, 256, lds);
}

template <bool M, bool W>
static hipError_t launch_shoot_f64_t(const rpt64::ShootArgs64& a, int n_blocks, hipStream_t stream) {
    const size_t lds = size_t(rpt64::kPhLdsDoubles) * 8u;
    const bool tab = a.a.sc.n_objects <= rpt64::kLdsObjs && a.a.sc.n_obj_tris <= rpt64::kLdsTris;
    if (tab) hipLaunchKernelGGL((rpt64::photon_shoot_f64_kernel<M, W, true>), dim3(n_blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((rpt64::photon_shoot_f64_kernel<M, W, false>), dim3(n_blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}
hipError_t launch_photon_shoot_f64(const rpt64::ShootArgs64& a, int n_blocks, hipStream_t stream) {
    const bool m = a.a.sc.has_medium != 0, w = a.surf != nullptr || a.vol != nullptr;
    if (m) return w ? launch_shoot_f64_t<true, true>(a, n_blocks, stream) : launch_shoot_f64_t<true, false>(a, n_blocks, stream);
    return w ? launch_shoot_f64_t<false, true>(a, n_blocks, stream) : launch_shoot_f64_t<false, false>(a, n_blocks, stream);
}
hipError_t launch_photon_surface_f64(const rpt64::SurfArgs64& a, int n_blocks, hipStream_t stream) {
    if (!a.a.n_items) return hipSuccess;
    const size_t lds = size_t(rpt64::kPhLdsDoubles) * 8u;
    const bool tab = a.a.sc.n_objects <= rpt64::kLdsObjs && a.a.sc.n_obj_tris <= rpt64::kLdsTris;
    const bool m = a.a.sc.has_medium != 0;
    if (m) {
        if (tab) hipLaunchKernelGGL((rpt64::photon_surface_f64_kernel<true, true>), dim3(n_blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((rpt64::photon_surface_f64_kernel<true, false>), dim3(n_blocks), dim3(256), lds, stream, a);
    } else {
        if (tab) hipLaunchKernelGGL((rpt64::photon_surface_f64_kernel<false, true>), dim3(n_blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((rpt64::photon_surface_f64_kernel<false, false>), dim3(n_blocks), dim3(256), lds, stream, a);
    }
    return hipGetLastError();
}
hipError_t launch_resolve_photon_f64(const rpt64::Args& a, const void* slab32, uint32_t n_chunks32, double scale_over_total, bool accumulate,
                                     double* d_out, hipStream_t stream) {
    if (!a.n_owned) return hipSuccess;
    hipLaunchKernelGGL(rpt64::resolve_photon_f64_kernel, dim3((a.n_owned + 255u) / 256u), dim3(256), 0, stream, a,
                       static_cast<const float4*>(slab32), n_chunks32, scale_over_total, accumulate ? 1 : 0, d_out);
    return hipGetLastError();
}
}  // namespace rptg
