// device_core.h — gfx950 device functions of the path-tracing core (fp32).
// Reference line citations are relative to the rpt source tree (src/...).
#pragma once
#include <hip/hip_runtime.h>
#include "gpu_layout.h"

namespace rptg {

#define RPT_DEV __device__ __forceinline__

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInf = __builtin_huge_valf();

// ------------------------------------------------------------------ vec3
struct V {
    float x, y, z;
};
RPT_DEV V mk(float x, float y, float z) { return V{x, y, z}; }
RPT_DEV V operator+(V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; }
RPT_DEV V operator-(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
RPT_DEV V operator-(V a) { return V{-a.x, -a.y, -a.z}; }
RPT_DEV V operator*(float s, V a) { return V{s * a.x, s * a.y, s * a.z}; }
RPT_DEV V operator*(V a, float s) { return V{s * a.x, s * a.y, s * a.z}; }
RPT_DEV V operator*(V a, V b) { return V{a.x * b.x, a.y * b.y, a.z * b.z}; }
RPT_DEV float dot(V a, V b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RPT_DEV V cross(V a, V b) {
    return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RPT_DEV V fma3(float s, V a, V b) { return V{fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)}; }
RPT_DEV V fma3(V s, V a, V b) { return V{fmaf(s.x, a.x, b.x), fmaf(s.y, a.y, b.y), fmaf(s.z, a.z, b.z)}; }
RPT_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RPT_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_sqrt_f32 as it is (1 ulp, like rcp / rsq above).  sqrtf() is correctly rounded by default under hipcc: one v_sqrt_f32
// plus ~17 instructions of scaling and refinement around it, per call, in the sphere test and the direction samplers.
RPT_DEV float sqrt1(float x) { return __builtin_amdgcn_sqrtf(x); }
RPT_DEV V normalize(V a) { return rsq(dot(a, a)) * a; }
RPT_DEV V vmin(V a, V b) { return V{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
RPT_DEV float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
RPT_DEV float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
RPT_DEV V xyz(const F4& f) { return V{f.x, f.y, f.z}; }
RPT_DEV float dot3w(const F4& r, V p) { return fmaf(r.x, p.x, fmaf(r.y, p.y, fmaf(r.z, p.z, r.w))); }
RPT_DEV float dot3(const F4& r, V p) { return fmaf(r.x, p.x, fmaf(r.y, p.y, r.z * p.z)); }
RPT_DEV bool is_zero(V a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }

// Wave-uniform read of a 16-byte-multiple scene record through the constant address space, so
// hipcc emits scalar loads (s_load_dwordx4/x8 into SGPRs).  A plain load through the generic
// pointer is emitted as a per-lane global_load (12 VGPRs per 48-byte record) because the kernel
// also stores to global memory and the compiler cannot prove the record is never clobbered.
// Only valid when `p` is the same in every lane.
template <class T>
RPT_DEV T uload(const T* p) {
    static_assert(sizeof(T) % 16 == 0, "scene records are 16-byte multiples");
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u4v* CP;
    CP q = (CP)(uintptr_t)p;
    u4v raw[sizeof(T) / 16];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) raw[i] = q[i];
    T out;
    __builtin_memcpy(&out, raw, sizeof(T));  // not a cast: the record's fields are floats, the loads are uint vectors
    return out;
}

// The scene view as it lies in the kernel-argument segment.  EVERY kernel of this library takes its arguments as one
// struct that begins with the SceneView (RenderArgs, QueryArgs, ShootArgs, intersect_kernel's first parameter: static
// asserts next to each), so kernarg + 0 is the view.  Reading a field through this pointer -- constant address space,
// behind an opaque copy of the pointer -- is a scalar load at the place of use; read as a plain kernel argument it is
// hoisted to the kernel's entry and held in a scalar register for the kernel's whole life.
// CONSEQUENCE: the `const SceneView&` parameter that scan_prims, closest_hit, finalize_hit, load_mat, env_color, ... still take
// is NOT what they read -- it only keeps the call sites readable.  A kernel whose first argument is not the view, or a caller
// that builds a modified copy of the view, would silently get the launch's own view: every kernel that calls these functions
// carries a static_assert (or, for intersect_kernel, a comment) that its argument struct begins with the SceneView.
typedef const __attribute__((address_space(4))) SceneView* KernargView;
RPT_DEV KernargView kernarg_scene() {
    KernargView kv = (KernargView)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kv));
    return kv;
}
// A sub-struct of the kernel arguments by value (dword loads through the constant address space).
template <class T>
RPT_DEV T kernarg_load(const __attribute__((address_space(4))) T* p) {
    static_assert(sizeof(T) % 4 == 0, "kernel-argument structs are dword multiples");
    typedef const __attribute__((address_space(4))) uint32_t* Q;
    Q q = (Q)p;
    uint32_t raw[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) raw[i] = q[i];
    T out;
    __builtin_memcpy(&out, raw, sizeof(T));
    return out;
}
// The same for a kernel's whole argument struct T (it lies at kernarg + 0).
template <class T>
RPT_DEV const __attribute__((address_space(4))) T* kernarg_args() {
    typedef const __attribute__((address_space(4))) T* P;
    P p = (P)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// ------------------------------------------------------------------ RNG
// xoshiro128+ seeded through splitmix64 from (seed, pixel, sample); bit-identical to the
// oracle's Rng.  Replaces StdRng::from_entropy() per row (src/renderer.rs:163).
RPT_DEV uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
struct Rng {
    uint32_t s0, s1, s2, s3;
    // `a` = mix64(seed + GOLDEN), computed on the host.
    RPT_DEV void seed(uint64_t a, uint32_t pixel, uint32_t sample) {
        const uint64_t G = 0x9E3779B97F4A7C15ULL;
        uint64_t z = a ^ ((uint64_t(sample) << 32) | uint64_t(pixel));
        uint64_t r0 = mix64(z + G), r1 = mix64(z + 2 * G);
        s0 = uint32_t(r0);
        s1 = uint32_t(r0 >> 32);
        s2 = uint32_t(r1);
        s3 = uint32_t(r1 >> 32);
    }
    RPT_DEV uint32_t next() {
        uint32_t r = s0 + s3;
        uint32_t t = s1 << 9;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = __builtin_rotateleft32(s3, 11);
        return r;
    }
    // (2k+1) * 2^-24 with k = top 23 bits: open interval (0,1), exact in fp32.
    RPT_DEV float uniform() { return float(((next() >> 9) << 1) | 1u) * 0x1p-24f; }
    RPT_DEV float range(float a, float b) { return fmaf(b - a, uniform(), a); }
    RPT_DEV uint32_t index(uint32_t n) { return __umulhi(next(), n); }
    RPT_DEV void unit_disc(float& x, float& y) {  // rand_distr::UnitDisc (rejection)
        for (;;) {
            x = range(-1.f, 1.f);
            y = range(-1.f, 1.f);
            if (fmaf(x, x, y * y) <= 1.f) return;
        }
    }
};

// ------------------------------------------------------------------ primitive tests
// Each returns the hit parameter or a negative value for "no hit in [tmin, tmax)".
// `t` is shared between world and local space because the local direction is not
// renormalised (Ray::apply_transform, src/shape.rs:65-72).

// Unit sphere, src/shape/sphere.rs:14-46.  Same roots as the reference's (-b -/+ sqrt(b^2-ac))/a,
// evaluated through the closest-approach vector so fp32 keeps the discriminant's digits.
RPT_DEV bool sphere_closer(V ol, V dl, float tmin, float tbest, float& t);
RPT_DEV float hit_sphere(V ol, V dl, float tmin) {
    float t;
    return sphere_closer(ol, dl, tmin, __builtin_inff(), t) ? t : -1.f;
}
// The scans' form of the three bounded primitives: "is there an accepted root closer than tbest", one chain of compares
// (a negative discriminant leaves NaN in t, which fails them).  hit_sphere / hit_cube / hit_aabb are these
// with tbest = inf, and the bodies are compiled without fp contraction (their fmas are written out): a scan and a tree
// walk that reach the same primitive then get the same bits for t wherever the body is inlined.
RPT_DEV bool sphere_closer(V ol, V dl, float tmin, float tbest, float& t) {
#pragma clang fp contract(off)
    float a = dot(dl, dl);
    float inv_a = rcp(a);
    float bb = dot(dl, ol) * inv_a;
    V l = fma3(-bb, dl, ol);
    float sq = sqrt1((1.f - dot(l, l)) * inv_a);
    float t0 = -bb - sq;
    t = (t0 < tmin) ? -bb + sq : t0;
    return t >= tmin && t < tbest;
}
RPT_DEV bool slabs_closer(float x1, float x2, float y1, float y2, float z1, float z2, float tmin, float tbest, float& t) {
#pragma clang fp contract(off)
    const float start = max3(fminf(x1, x2), fminf(y1, y2), fminf(z1, z2));
    const float end = min3(fmaxf(x1, x2), fmaxf(y1, y2), fmaxf(z1, z2));
    t = start < tmin ? end : start;
    return !(start > end) && !(end < tmin) && t < tbest;
}
RPT_DEV bool cube_closer(V ol, V dl, float tmin, float tbest, float& t) {
#pragma clang fp contract(off)
    float ix = rcp(dl.x), iy = rcp(dl.y), iz = rcp(dl.z);
    return slabs_closer((-0.5f - ol.x) * ix, (0.5f - ol.x) * ix, (-0.5f - ol.y) * iy, (0.5f - ol.y) * iy, (-0.5f - ol.z) * iz,
                        (0.5f - ol.z) * iz, tmin, tbest, t);
}
RPT_DEV bool aabb_closer(const F4& lo, const F4& hi, V o, V inv, float tmin, float tbest, float& t) {
#pragma clang fp contract(off)
    return slabs_closer((lo.x - o.x) * inv.x, (hi.x - o.x) * inv.x, (lo.y - o.y) * inv.y, (hi.y - o.y) * inv.y,
                        (lo.z - o.z) * inv.z, (hi.z - o.z) * inv.z, tmin, tbest, t);
}
// Unit cube [-1/2,1/2]^3, src/shape/cube.rs:22-74: t of the accepted root (the face is found from the hit point: box_face_normal).
RPT_DEV float hit_cube(V ol, V dl, float tmin) {
    float t;
    return cube_closer(ol, dl, tmin, __builtin_inff(), t) ? t : -1.f;
}
// Plane, src/shape/plane.rs:17-32 (world-space (n, value)).
RPT_DEV float hit_plane(const F4& nv, V o, V d, float tmin) {
    float c = dot3(nv, d);
    if (fabsf(c) < 1e-8f) return -1.f;
    float t = (nv.w - dot3(nv, o)) * rcp(c);
    return (t >= tmin) ? t : -1.f;
}
// Triangle, src/shape/mesh.rs:50-83, with the plane normal and the barycentric functionals
// pre-solved on the host.  Accepts t in [tmin, tmax).
RPT_DEV float hit_tri(const F4& pn, const F4& A, const F4& B, V o, V d, float tmin, float tmax) {
    float c = dot3(pn, d);
    float t = (pn.w - dot3(pn, o)) * rcp(c);
    V p = fma3(t, d, o);
    float v = dot3w(A, p), w = dot3w(B, p);
    float u = 1.f - v - w;
    bool ok = fabsf(c) >= 1e-8f && t >= tmin && t < tmax && u >= 0.f && v >= 0.f && w >= 0.f;
    return ok ? t : -1.f;
}
// Axis-aligned box in world space (a cube under positive scale + translation): the reference's
// local slab test (src/shape/cube.rs:22-74) evaluated in world coordinates, where it yields the
// same entry/exit parameters because t is shared between the two spaces.  `inv` = 1/d.
RPT_DEV float hit_aabb(const F4& lo, const F4& hi, V o, V inv, float tmin) {
    float t;
    return aabb_closer(lo, hi, o, inv, tmin, __builtin_inff(), t) ? t : -1.f;
}
// Axis-aligned rectangle = two coplanar triangles of src/shape/mesh.rs:50-83 (u,v,w >= 0 on one
// of them <=> the point lies in the closed rectangle).  oa/da/ia: origin, direction and 1/direction
// on the rectangle's axis; (ou,du), (ov,dv) on the two in-plane axes.
RPT_DEV float hit_rect(const F4& a, float vmax, float oa, float ia, float ou, float du, float ov, float dv,
                       float tmin, float tmax) {
    float t = (a.x - oa) * ia;
    float pu = fmaf(t, du, ou), pv = fmaf(t, dv, ov);
    bool ok = t >= tmin && t < tmax && pu >= a.y && pu <= a.z && pv >= a.w && pv <= vmax;
    return ok ? t : -1.f;
}
// Box shell (ShellScan): the closest rectangle among the faces of one axis-aligned box = the slab entry if
// that face exists and lies at or beyond tmin, else the slab exit.  Same t = (plane - o) * inv as hit_rect.
RPT_DEV void hit_shell(const ShellScan& s, V o, V inv, float tmin, float& tbest, uint32_t& code) {
    float x1 = (s.lo.x - o.x) * inv.x, x2 = (s.hi.x - o.x) * inv.x;
    float y1 = (s.lo.y - o.y) * inv.y, y2 = (s.hi.y - o.y) * inv.y;
    float z1 = (s.lo.z - o.z) * inv.z, z2 = (s.hi.z - o.z) * inv.z;
    bool sx = x1 > x2, sy = y1 > y2, sz = z1 > z2;
    float xl = sx ? x2 : x1, xh = sx ? x1 : x2;
    float yl = sy ? y2 : y1, yh = sy ? y1 : y2;
    float zl = sz ? z2 : z1, zh = sz ? z1 : z2;
    float start = max3(xl, yl, zl), end = min3(xh, yh, zh);
    // entering through the lo face of an axis unless the direction is negative there (swapped)
    uint32_t in_x = sx ? s.face[1] : s.face[0], out_x = sx ? s.face[0] : s.face[1];
    uint32_t in_y = sy ? s.face[3] : s.face[2], out_y = sy ? s.face[2] : s.face[3];
    uint32_t in_z = sz ? s.face[5] : s.face[4], out_z = sz ? s.face[4] : s.face[5];
    uint32_t c_in = (xl > yl && xl > zl) ? in_x : (yl > zl ? in_y : in_z);
    uint32_t c_out = (xh < yh && xh < zh) ? out_x : (yh < zh ? out_y : out_z);
    bool box = start <= end;
    bool use_in = box && start >= tmin && c_in != CODE_MISS;
    bool use_out = box && end >= tmin && c_out != CODE_MISS;
    float t = use_in ? start : end;
    if ((use_in || use_out) && t < tbest) { tbest = t; code = use_in ? c_in : c_out; }
}
RPT_DEV void to_local(const XfScan& x, V o, V d, V& ol, V& dl) {
    ol = mk(dot3w(x.r0, o), dot3w(x.r1, o), dot3w(x.r2, o));
    dl = mk(dot3(x.r0, d), dot3(x.r1, d), dot3(x.r2, d));
}

// ------------------------------------------------------------------ closest hit
// Renderer::get_closest_hit, src/renderer.rs:416-425: every object is tested, the closest
// accepted hit wins, ties keep the earlier object (strict `<`).  The analytic primitives are
// scanned with a wave-uniform index (scalar loads); BVH meshes are walked per lane.
// Per-lane BVH walk ("while-while"): descend inner nodes until every lane of the wave holds a
// leaf (or is done), then test the leaves' triangles together.  Stack = one LDS column per lane.
RPT_DEV void slab2(const float lo[3], const float hi[3], V o, V inv, float& tn, float& tf) {
    float x1 = (lo[0] - o.x) * inv.x, x2 = (hi[0] - o.x) * inv.x;
    float y1 = (lo[1] - o.y) * inv.y, y2 = (hi[1] - o.y) * inv.y;
    float z1 = (lo[2] - o.z) * inv.z, z2 = (hi[2] - o.z) * inv.z;
    tn = max3(fminf(x1, x2), fminf(y1, y2), fminf(z1, z2));
    tf = min3(fmaxf(x1, x2), fmaxf(y1, y2), fmaxf(z1, z2));
}
// Occluder test of a shadow walk: a hit closer than t_block on anything but the light's twin primitives
// (hit codes tw_lo..tw_hi) decides the shadow test -- the walk may stop without finding the closest hit.
struct AnyHit {
    float t_block;
    uint32_t tw_lo, tw_hi;
    RPT_DEV bool blocks(float t, uint32_t code) const { return t < t_block && !(code >= tw_lo && code <= tw_hi); }
};
template <bool COUNT, bool PRIMS, bool ANY = false>
RPT_DEV void bvh_traverse(const SceneView& sc, uint32_t root, V o, V d, float tmin, float& tbest, uint32_t& code,
                          uint32_t& inst, uint32_t* stk, uint32_t stride, uint32_t cap, uint32_t& c_nodes,
                          uint32_t& c_tris, AnyHit any = AnyHit{-kInf, 1u, 0u});

// One primitive of a BVH_PRIMS leaf (per lane: kinds may differ between lanes).  `stk`/`cap`:
// the part of the lane's stack column above the caller's entries, for the nested walk of an instance.
template <bool COUNT>
RPT_DEV void hit_prim(const SceneView& scene_, uint32_t pc, V o, V d, V inv, float tmin, float& tbest, uint32_t& code,
                      uint32_t& inst, uint32_t* stk, uint32_t stride, uint32_t cap, uint32_t& c_nodes,
                      uint32_t& c_tris) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    const uint32_t kind = pc >> 28, i = pc & 0x0FFFFFFFu;
    float t = -1.f;
    if (kind == K_INST) {
        const InstRec r = sc.inst[i];
        const V ol = mk(dot3w(r.r0, o), dot3w(r.r1, o), dot3w(r.r2, o));
        const V dl = mk(dot3(r.r0, d), dot3(r.r1, d), dot3(r.r2, d));   // not renormalised: t is shared
        uint32_t c2 = CODE_MISS, unused = 0;
        float tb = tbest;
        bvh_traverse<COUNT, false>(scene_, __float_as_uint(r.n1.w), ol, dl, tmin, tb, c2, unused, stk, stride, cap, c_nodes,
                                   c_tris);
        if (c2 != CODE_MISS) { tbest = tb; code = (K_INSTTRI << 28) | (c2 & 0x0FFFFFFFu); inst = i; }
        return;
    }
    if (kind == K_SPHERE) {
        const XfScan x = sc.sph[i];
        V ol, dl;
        to_local(x, o, d, ol, dl);
        t = hit_sphere(ol, dl, tmin);
    } else if (kind == K_CUBE) {
        const XfScan x = sc.cub[i];
        V ol, dl;
        to_local(x, o, d, ol, dl);
        t = hit_cube(ol, dl, tmin);
    } else if (kind == K_AABB) {
        const AabbScan b = sc.aabb[i];
        t = hit_aabb(b.lo, b.hi, o, inv, tmin);
    } else if (kind == K_RECT) {
        const RectScan r = sc.rect[i];
        if (i < sc.n_rect_x) t = hit_rect(r.a, r.b.x, o.x, inv.x, o.y, d.y, o.z, d.z, tmin, tbest);
        else if (i < sc.n_rect_x + sc.n_rect_y) t = hit_rect(r.a, r.b.x, o.y, inv.y, o.z, d.z, o.x, d.x, tmin, tbest);
        else t = hit_rect(r.a, r.b.x, o.z, inv.z, o.x, d.x, o.y, d.y, tmin, tbest);
    } else {  // K_TRI
        const TriScan tr = sc.tri[i];
        t = hit_tri(tr.pn, tr.A, tr.B, o, d, tmin, tbest);
    }
    if (t >= 0.f && t < tbest) { tbest = t; code = pc; }
}

template <bool COUNT, bool PRIMS, bool ANY>
RPT_DEV void bvh_traverse(const SceneView& scene_, uint32_t root, V o, V d, float tmin, float& tbest, uint32_t& code,
                          uint32_t& inst, uint32_t* stk, uint32_t stride, uint32_t cap, uint32_t& c_nodes,
                          uint32_t& c_tris, AnyHit any) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    const BvhNode* nodes = sc.nodes;
    const V inv = mk(rcp(d.x), rcp(d.y), rcp(d.z));
    const uint32_t kDone = 0xFFFFFFFFu;  // a 32-item prim leaf at the last index never occurs
    uint32_t sp = 0;
    uint32_t cur = root;  // a root is always an inner node
    while (cur != kDone) {
        while (!(cur & BVH_LEAF)) {  // kDone has the leaf bit set, so finished lanes fall through
            const BvhNode nd = nodes[cur];
            if (COUNT) c_nodes++;
            float n0, f0, n1, f1;
            slab2(nd.lo0, nd.hi0, o, inv, n0, f0);
            slab2(nd.lo1, nd.hi1, o, inv, n1, f1);
            const bool h0 = fmaxf(n0, tmin) <= fminf(f0, tbest);
            const bool h1 = fmaxf(n1, tmin) <= fminf(f1, tbest);
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                if (sp < cap) { stk[sp * stride] = first0 ? nd.e1 : nd.e0; sp++; }
                else if (COUNT && sc.stack_overflows) atomicAdd(sc.stack_overflows, 1ull);   // (never, by the commit-time depth check)
                cur = first0 ? nd.e0 : nd.e1;
            } else if (h0 || h1) {
                cur = h0 ? nd.e0 : nd.e1;
            } else if (sp) {
                sp--;
                cur = stk[sp * stride];
            } else {
                cur = kDone;
            }
        }
        if (cur != kDone) {
            const uint32_t first = cur & BVH_INDEX_MASK;
            const uint32_t count = ((cur >> 26) & 31u) + 1u;
            if (PRIMS && (cur & BVH_PRIMS)) {
                for (uint32_t i = 0; i < count; i++) {
                    if (COUNT) c_tris++;
                    hit_prim<COUNT>(scene_, sc.pleaf[first + i], o, d, inv, tmin, tbest, code, inst, stk + sp * stride, stride,
                                    cap - sp, c_nodes, c_tris);
                }
            } else {
                for (uint32_t i = 0; i < count; i++) {
                    const TriScan tr = sc.btri[first + i];
                    if (COUNT) c_tris++;
                    float t = hit_tri(tr.pn, tr.A, tr.B, o, d, tmin, tbest);
                    if (t >= 0.f) { tbest = t; code = (K_BVHTRI << 28) | (first + i); }
                }
            }
            if (ANY && code != CODE_MISS && any.blocks(tbest, code)) sp = 0;  // an occluder is known: nothing left to find
            if (sp) {
                sp--;
                cur = stk[sp * stride];
            } else {
                cur = kDone;
            }
        }
    }
}

// The linear scan over the wave-uniform primitive records (everything that is not in a tree).
// MASKED: bit i of `mask` (wave-uniform) says whether bounded record i -- numbered in scan order: spheres, cubes,
// boxes, rectangles, triangles, as in SceneView::pbox -- can be hit at all; planes and the shell are always tested.
template <bool MASKED = false>
RPT_DEV void scan_prims(const SceneView& scene, V o, V d, float tmin, float& tbest, uint32_t& code, uint64_t mask = ~0ull) {
    // The scene view is a kernel argument (every caller passes its kernarg struct): its fields are read HERE, through the
    // constant address space, behind an opaque copy of the pointer.  Read as plain kernel arguments they are all hoisted to
    // the kernel's entry and held in scalar registers for its whole life -- the render kernels have ~110 such values, the
    // compiler parks the overflow in VGPR lanes, and 14 % of their VALU instructions were v_readlane / v_writelane.
    const auto& sc = *kernarg_scene();
    (void)scene;
    uint32_t bit = 0;  // wave-uniform record number
    auto on = [&](uint32_t i) { return !MASKED || ((mask >> ((bit + i) & 63u)) & 1ull) != 0ull; };
    for (uint32_t i = 0; i < sc.n_sph; i++) {
        if (!on(i)) continue;
        const XfScan x = uload(&sc.sph[i]);
        V ol, dl;
        to_local(x, o, d, ol, dl);
        float t;
        if (sphere_closer(ol, dl, tmin, tbest, t)) { tbest = t; code = (K_SPHERE << 28) | i; }
    }
    bit += sc.n_sph;
    {   // two records per iteration: independent instruction streams for the scheduler (unmasked scans)
        uint32_t i = 0;
        if (!MASKED)
            for (; i + 1u < sc.n_cub; i += 2u) {
                const XfScan x0 = uload(&sc.cub[i]), x1 = uload(&sc.cub[i + 1u]);
                V ol0, dl0, ol1, dl1;
                to_local(x0, o, d, ol0, dl0);
                to_local(x1, o, d, ol1, dl1);
                float t0, t1;
                if (cube_closer(ol0, dl0, tmin, tbest, t0)) { tbest = t0; code = (K_CUBE << 28) | i; }
                if (cube_closer(ol1, dl1, tmin, tbest, t1)) { tbest = t1; code = (K_CUBE << 28) | (i + 1u); }
            }
        for (; i < sc.n_cub; i++) {
            if (!on(i)) continue;
            const XfScan x = uload(&sc.cub[i]);
            V ol, dl;
            to_local(x, o, d, ol, dl);
            float t;
            if (cube_closer(ol, dl, tmin, tbest, t)) { tbest = t; code = (K_CUBE << 28) | i; }
        }
    }
    bit += sc.n_cub;
    for (uint32_t i = 0; i < sc.n_pln; i++) {
        const F4 nv = uload(&sc.pln[i]).nv;
        float t = hit_plane(nv, o, d, tmin);
        if (t >= 0.f && t < tbest) { tbest = t; code = (K_PLANE << 28) | i; }
    }
    const uint32_t n_rect = sc.n_rect_x + sc.n_rect_y + sc.n_rect_z;
    if (sc.n_aabb + n_rect + sc.has_shell != 0) {  // wave-uniform: these kinds share one reciprocal direction per ray
        const V inv = mk(rcp(d.x), rcp(d.y), rcp(d.z));
        if (sc.has_shell) hit_shell(uload(sc.shell), o, inv, tmin, tbest, code);
        {
            uint32_t i = 0;
            if (!MASKED)
                for (; i + 1u < sc.n_aabb; i += 2u) {
                    const AabbScan b0 = uload(&sc.aabb[i]), b1 = uload(&sc.aabb[i + 1u]);
                    float t0, t1;
                    if (aabb_closer(b0.lo, b0.hi, o, inv, tmin, tbest, t0)) { tbest = t0; code = (K_AABB << 28) | i; }
                    if (aabb_closer(b1.lo, b1.hi, o, inv, tmin, tbest, t1)) { tbest = t1; code = (K_AABB << 28) | (i + 1u); }
                }
            for (; i < sc.n_aabb; i++) {
                if (!on(i)) continue;
                const AabbScan b = uload(&sc.aabb[i]);
                float t;
                if (aabb_closer(b.lo, b.hi, o, inv, tmin, tbest, t)) { tbest = t; code = (K_AABB << 28) | i; }
            }
        }
        bit += sc.n_aabb;
        uint32_t i = 0;
        for (uint32_t e = sc.n_rect_x; i < e; i++) {
            if (!on(i)) continue;
            const RectScan r = uload(&sc.rect[i]);
            float t = hit_rect(r.a, r.b.x, o.x, inv.x, o.y, d.y, o.z, d.z, tmin, tbest);
            if (t >= 0.f) { tbest = t; code = (K_RECT << 28) | i; }
        }
        for (uint32_t e = sc.n_rect_x + sc.n_rect_y; i < e; i++) {
            if (!on(i)) continue;
            const RectScan r = uload(&sc.rect[i]);
            float t = hit_rect(r.a, r.b.x, o.y, inv.y, o.z, d.z, o.x, d.x, tmin, tbest);
            if (t >= 0.f) { tbest = t; code = (K_RECT << 28) | i; }
        }
        for (uint32_t e = n_rect; i < e; i++) {
            if (!on(i)) continue;
            const RectScan r = uload(&sc.rect[i]);
            float t = hit_rect(r.a, r.b.x, o.z, inv.z, o.x, d.x, o.y, d.y, tmin, tbest);
            if (t >= 0.f) { tbest = t; code = (K_RECT << 28) | i; }
        }
        bit += n_rect;
    } else {
        bit += sc.n_aabb + n_rect;
    }
    for (uint32_t i = 0; i < sc.n_tri; i++) {
        if (!on(i)) continue;
        const TriScan tr = uload(&sc.tri[i]);
        float t = hit_tri(tr.pn, tr.A, tr.B, o, d, tmin, tbest);
        if (t >= 0.f) { tbest = t; code = (K_TRI << 28) | i; }
    }
}
// Which scanned records can a query touch that stays inside the ball (c, r) of each live lane?  Wave-uniform mask
// for scan_prims<true> (the union over the lanes: one lane's ball reaching a box keeps that record for all of them).
// Scenes with more than 64 bounded scan records do not occur (from 64 on, the scene-level tree takes over).
// `touched` (optional): does any record's box reach THIS lane's ball?
RPT_DEV uint64_t scan_mask_for_ball(const SceneView& scene_, bool live, V c, float r, bool* touched = nullptr) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    const uint32_t n = sc.n_sph + sc.n_cub + sc.n_aabb + sc.n_rect_x + sc.n_rect_y + sc.n_rect_z + sc.n_tri;
    if (touched) *touched = n != 0u;
    if (n > 64u) return ~0ull;
    bool mine = false;
    const float r2 = r * r;
    uint64_t mask = 0ull;
    auto reaches = [&](const AabbScan& b) {
        const float dx = fmaxf(fmaxf(b.lo.x - c.x, c.x - b.hi.x), 0.f);
        const float dy = fmaxf(fmaxf(b.lo.y - c.y, c.y - b.hi.y), 0.f);
        const float dz = fmaxf(fmaxf(b.lo.z - c.z, c.z - b.hi.z), 0.f);
        const bool t = live && fmaf(dx, dx, fmaf(dy, dy, dz * dz)) <= r2;
        mine = mine || t;
        return __any(t);
    };
    uint32_t i = 0;
    for (; i + 4u <= n; i += 4u) {   // four records' scalar loads in flight at a time
        const AabbScan b0 = uload(&sc.pbox[i]), b1 = uload(&sc.pbox[i + 1u]), b2 = uload(&sc.pbox[i + 2u]), b3 = uload(&sc.pbox[i + 3u]);
        if (reaches(b0)) mask |= 1ull << i;
        if (reaches(b1)) mask |= 2ull << i;
        if (reaches(b2)) mask |= 4ull << i;
        if (reaches(b3)) mask |= 8ull << i;
    }
    for (; i < n; i++)
        if (reaches(uload(&sc.pbox[i]))) mask |= 1ull << i;
    if (touched) *touched = mine;
    return mask;
}
// Would a walk of the per-mesh trees visit anything?  The two child boxes of every mesh root against the
// interval the scan left (scalar loads: the roots are wave-uniform).
RPT_DEV bool mesh_roots_hit(const SceneView& scene_, V o, V d, float tmin, float tbest) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    const V inv = mk(rcp(d.x), rcp(d.y), rcp(d.z));
    bool need = false;
    for (uint32_t i = 0; i < sc.n_mesh; i++) {
        const MeshRef m = uload(&sc.meshes[i]);
        const BvhNode nd = uload(&sc.nodes[m.root]);
        float n0, f0, n1, f1;
        slab2(nd.lo0, nd.hi0, o, inv, n0, f0);
        slab2(nd.lo1, nd.hi1, o, inv, n1, f1);
        need = need || fmaxf(n0, tmin) <= fminf(f0, tbest) || fmaxf(n1, tmin) <= fminf(f1, tbest);
    }
    return need;
}
template <bool COUNT, bool ANY>
RPT_DEV void walk_meshes(const SceneView& scene_, V o, V d, float tmin, float& tbest, uint32_t& code, uint32_t& inst,
                         uint32_t* stk, uint32_t stride, uint32_t& c_nodes, uint32_t& c_tris, AnyHit any, uint32_t cap = 32u) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    for (uint32_t i = 0; i < sc.n_mesh; i++) {
        const MeshRef m = uload(&sc.meshes[i]);
        if (ANY && code != CODE_MISS && any.blocks(tbest, code)) break;  // (per lane) already occluded
        bvh_traverse<COUNT, false, ANY>(scene_, m.root, o, d, tmin, tbest, code, inst, stk, stride, cap, c_nodes, c_tris, any);
    }
}
// The same walk as a resumable one (deferred walks of the render kernel): the lane's position -- current entry,
// stack height, mesh -- lives in `w` and its stack column in LDS, and the wave leaves the loop as soon as fewer
// than `min_active` lanes are still walking; the stragglers continue with the next batch of parked walks instead
// of holding 60 idle lanes.  w.cur == kWalkDone: nothing left to do.
static const uint32_t kWalkDone = 0xFFFFFFFFu;  // has the leaf bit set; a 32-item prim leaf at the last index never occurs
struct WalkState {
    uint32_t cur, sp, mesh;
};
RPT_DEV WalkState walk_begin(const SceneView&) { return WalkState{uload(&kernarg_scene()->meshes[0]).root, 0u, 0u}; }
// `cap`: rows of the stack column (a mesh tree is at most bvh_max_depth = 20 levels deep: it pushes at most 19 entries).
template <bool COUNT>
// `leaf_quarters`: the lanes reach their next leaf after very different numbers of steps (a few on average, a dozen for the
// slowest of 25), and a descent that waits for the last of them runs most of its steps for a handful of lanes.  So the
// descent pauses as soon as 4 x (lanes waiting at a leaf) >= leaf_quarters x (lanes still descending): the waiting lanes
// test their triangles, the others go on from where they are.  Every lane still performs its own steps in its own order.
RPT_DEV void walk_meshes_resumable(const SceneView& scene_, V o, V d, float tmin, float& tbest, uint32_t& code, uint32_t* stk,
                                   uint32_t stride, WalkState& w, uint32_t min_active, AnyHit any, uint32_t& c_nodes,
                                   uint32_t& c_tris, uint32_t cap = 32u, uint32_t leaf_quarters = 0u, uint32_t* c_wave = nullptr) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    const BvhNode* nodes = sc.nodes;
    const V inv = mk(rcp(d.x), rcp(d.y), rcp(d.z));
    uint32_t cur = w.cur, sp = w.sp, mesh = w.mesh;
    while (uint32_t(__popcll(__ballot(cur != kWalkDone))) >= min_active) {
        for (;;) {  // the descent (kWalkDone has the leaf bit set: finished lanes take no part)
            const bool inner = !(cur & BVH_LEAF);
            const uint32_t n_inner = uint32_t(__popcll(__ballot(inner)));
            if (n_inner == 0u) break;
            if (leaf_quarters != 0u && 4u * uint32_t(__popcll(__ballot(!inner && cur != kWalkDone))) >= leaf_quarters * n_inner) break;
            if (COUNT && c_wave) c_wave[0]++;   // (wave-level: descent steps)
            if (!inner) continue;
            const BvhNode nd = nodes[cur];
            if (COUNT) c_nodes++;
            float n0, f0, n1, f1;
            slab2(nd.lo0, nd.hi0, o, inv, n0, f0);
            slab2(nd.lo1, nd.hi1, o, inv, n1, f1);
            const bool h0 = fmaxf(n0, tmin) <= fminf(f0, tbest);
            const bool h1 = fmaxf(n1, tmin) <= fminf(f1, tbest);
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                if (sp < cap) { stk[sp * stride] = first0 ? nd.e1 : nd.e0; sp++; }
                else if (COUNT && sc.stack_overflows) atomicAdd(sc.stack_overflows, 1ull);   // (never, by the commit-time depth check)
                cur = first0 ? nd.e0 : nd.e1;
            } else if (h0 || h1) {
                cur = h0 ? nd.e0 : nd.e1;
            } else if (sp) {
                sp--;
                cur = stk[sp * stride];
            } else {
                cur = kWalkDone;
            }
        }
        if (COUNT && c_wave) {   // (wave-level: triangle steps of this round = the largest leaf among the lanes at one)
            uint32_t mx = 0;
            for (uint32_t k = 1; k <= 4u; k++)
                if (__ballot((cur & BVH_LEAF) && cur != kWalkDone && ((cur >> 26) & 31u) + 1u >= k) != 0ull) mx = k;
            c_wave[1] += mx;
        }
        if ((cur & BVH_LEAF) && cur != kWalkDone) {   // (lanes that are still descending go on with the next round)
            const uint32_t first = cur & BVH_INDEX_MASK;
            const uint32_t count = ((cur >> 26) & 31u) + 1u;
            TriScan nxt = sc.btri[first];
            for (uint32_t i = 0; i < count; i++) {  // the next triangle is in flight while this one is tested
                const TriScan tr = nxt;
                if (i + 1u < count) nxt = sc.btri[first + i + 1u];
                if (COUNT) c_tris++;
                float t = hit_tri(tr.pn, tr.A, tr.B, o, d, tmin, tbest);
                if (t >= 0.f) { tbest = t; code = (K_BVHTRI << 28) | (first + i); }
            }
            if (code != CODE_MISS && any.blocks(tbest, code)) { sp = 0; mesh = sc.n_mesh; }  // an occluder is known
            if (sp) {
                sp--;
                cur = stk[sp * stride];
            } else {
                cur = kWalkDone;
            }
        }
        if (cur == kWalkDone && mesh + 1u < sc.n_mesh) {  // the next mesh's tree
            mesh++;
            cur = sc.meshes[mesh].root;
        }
    }
    w.cur = cur;
    w.sp = sp;
    w.mesh = mesh;
}

// BVH: 0 = no tree in the scene, 1 = per-mesh trees only, 2 = scene-level tree possible.
template <int BVH, bool COUNT, bool ANY = false>
RPT_DEV void closest_hit(const SceneView& scene_, V o, V d, float tmin, float& tbest, uint32_t& code, uint32_t& inst,
                         uint32_t* stk, uint32_t stride, uint32_t& c_nodes, uint32_t& c_tris,
                         AnyHit any = AnyHit{-kInf, 1u, 0u}) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    if (BVH == 2 && sc.scene_bvh) {  // wave-uniform: planes (unbounded) are scanned, everything else is in the tree
        for (uint32_t i = 0; i < sc.n_pln; i++) {
            const F4 nv = uload(&sc.pln[i]).nv;
            float t = hit_plane(nv, o, d, tmin);
            if (t >= 0.f && t < tbest) { tbest = t; code = (K_PLANE << 28) | i; }
        }
        bvh_traverse<COUNT, true, ANY>(scene_, sc.top_root, o, d, tmin, tbest, code, inst, stk, stride, 32u, c_nodes, c_tris, any);
        if (sc.mesh_deferred) walk_meshes<COUNT, ANY>(scene_, o, d, tmin, tbest, code, inst, stk, stride, c_nodes, c_tris, any);   // (wave-uniform)
        return;
    }
    scan_prims(scene_, o, d, tmin, tbest, code);
    if (BVH) walk_meshes<COUNT, ANY>(scene_, o, d, tmin, tbest, code, inst, stk, stride, c_nodes, c_tris, any);
}

// Everything of a query except the per-mesh trees: the linear scan, or -- BVH = 3: a scene tree whose meshes are walked
// separately (SceneView::mesh_deferred) -- the planes and the scene tree.
template <int BVH, bool COUNT>
RPT_DEV void scan_or_tree(const SceneView& scene_, V o, V d, float tmin, float& tbest, uint32_t& code, uint32_t& inst, uint32_t* stk,
                          uint32_t stride, uint32_t& c_nodes, uint32_t& c_tris) {
    if constexpr (BVH == 3) {
        const auto& sc = *kernarg_scene();
        for (uint32_t i = 0; i < sc.n_pln; i++) {
            const F4 nv = uload(&sc.pln[i]).nv;
            float t = hit_plane(nv, o, d, tmin);
            if (t >= 0.f && t < tbest) { tbest = t; code = (K_PLANE << 28) | i; }
        }
        bvh_traverse<COUNT, true, false>(scene_, sc.top_root, o, d, tmin, tbest, code, inst, stk, stride, 32u, c_nodes, c_tris);
    } else {
        scan_prims(scene_, o, d, tmin, tbest, code);
    }
}

// Outward normal of the face of the axis-aligned box [lo, hi] that the surface point p lies on: the axis whose nearer face
// plane is closest to p.
RPT_DEV V box_face_normal(V p, V lo, V hi) {
    const float xl = fabsf(p.x - lo.x), xh = fabsf(p.x - hi.x), yl = fabsf(p.y - lo.y), yh = fabsf(p.y - hi.y);
    const float zl = fabsf(p.z - lo.z), zh = fabsf(p.z - hi.z);
    const float ex = fminf(xl, xh), ey = fminf(yl, yh), ez = fminf(zl, zh);
    const bool ax0 = ex <= ey && ex <= ez, ax1 = !ax0 && ey <= ez;
    return mk(ax0 ? (xh < xl ? 1.f : -1.f) : 0.f, ax1 ? (yh < yl ? 1.f : -1.f) : 0.f, (!ax0 && !ax1) ? (zh < zl ? 1.f : -1.f) : 0.f);
}
// Normal and object of the winning primitive (per lane).
RPT_DEV void finalize_hit(const SceneView& scene_, V o, V d, float tmin, float t, uint32_t code, uint32_t inst, V& n,
                          uint32_t& obj) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    uint32_t kind = code >> 28, idx = code & 0x0FFFFFFFu;
    if (kind == K_INSTTRI) {  // src/shape/mesh.rs:78 in the instance's space, then src/shape.rs:131-134
        const InstRec r = sc.inst[inst];
        const TriShade s = sc.btri_sh[idx];
        V nl;
        if (s.n2.w != 0.f) {
            nl = xyz(s.n1);
        } else {
            const TriScan tr = sc.btri[idx];
            const V ol = mk(dot3w(r.r0, o), dot3w(r.r1, o), dot3w(r.r2, o));
            const V dl = mk(dot3(r.r0, d), dot3(r.r1, d), dot3(r.r2, d));
            V p = fma3(t, dl, ol);
            float v = dot3w(tr.A, p), w = dot3w(tr.B, p);
            float u = 1.f - v - w;
            nl = normalize(u * xyz(s.n1) + v * xyz(s.n2) + w * xyz(s.n3));
        }
        n = normalize(mk(dot3(r.n0, nl), dot3(r.n1, nl), dot3(r.n2, nl)));
        obj = __float_as_uint(r.n0.w);
        return;
    }
    if (kind == K_SPHERE) {  // src/shape/sphere.rs:40-41 then src/shape.rs:131-134
        const XfScan x = sc.sph[idx];
        const XfShade s = sc.sph_sh[idx];
        V ol, dl;
        to_local(x, o, d, ol, dl);
        V nl = normalize(fma3(t, dl, ol));
        n = (s.r1.w != 0.f) ? normalize(mk(dot3(s.r0, nl), dot3(s.r1, nl), dot3(s.r2, nl))) : nl;
        obj = __float_as_uint(s.r0.w);
    } else if (kind == K_CUBE) {
        const XfScan x = sc.cub[idx];
        const XfShade s = sc.cub_sh[idx];
        V ol, dl;
        to_local(x, o, d, ol, dl);
        // the face the hit point lies on (src/shape/cube.rs:37-55 picks the slab that bounds the root: the same face except
        // within rounding of an edge), from the point instead of a second run of the slab test
        const V nl = box_face_normal(fma3(t, dl, ol), mk(-0.5f, -0.5f, -0.5f), mk(0.5f, 0.5f, 0.5f));
        n = (s.r1.w != 0.f) ? normalize(mk(dot3(s.r0, nl), dot3(s.r1, nl), dot3(s.r2, nl))) : nl;
        obj = __float_as_uint(s.r0.w);
    } else if (kind == K_AABB) {
        const AabbScan b = sc.aabb[idx];
        n = box_face_normal(fma3(t, d, o), xyz(b.lo), xyz(b.hi));   // (as for the cube)
        obj = __float_as_uint(b.lo.w);
    } else if (kind == K_RECT) {
        const F4 s = sc.rect_sh[idx].n_obj;
        n = xyz(s);
        obj = __float_as_uint(s.w);
    } else if (kind == K_PLANE) {  // src/shape/plane.rs:27: -normalize(n) * signum(cos)
        const F4 nv = sc.pln[idx].nv;
        const F4 s = sc.pln_sh[idx].unit_n_obj;
        float c = dot3(nv, d);
        float sg = __builtin_signbit(c) ? 1.f : -1.f;
        n = sg * xyz(s);
        obj = __float_as_uint(s.w);
    } else {  // triangle: src/shape/mesh.rs:78, normals never face-forwarded
        const TriScan* ta = (kind == K_TRI) ? sc.tri : sc.btri;
        const TriShade* sa = (kind == K_TRI) ? sc.tri_sh : sc.btri_sh;
        const TriShade s = sa[idx];
        obj = __float_as_uint(s.n1.w);
        if (s.n2.w != 0.f) {  // flat: n1 == n2 == n3
            n = xyz(s.n1);
        } else {
            const TriScan tr = ta[idx];
            V p = fma3(t, d, o);
            float v = dot3w(tr.A, p), w = dot3w(tr.B, p);
            float u = 1.f - v - w;
            n = normalize(u * xyz(s.n1) + v * xyz(s.n2) + w * xyz(s.n3));
        }
    }
}

// ------------------------------------------------------------------ materials
struct Mat {
    V albedo;
    float emit;
    uint32_t kind;
    float shin, ior;
};
// Small per-scene tables that every lane indexes on its own -- the material of the object it hit, the triangle of the
// light it samples -- staged in LDS by the render kernel (a ds_read instead of a dependent global load in the middle of a
// stage).  A table that does not fit stays in global memory (count 0 here); kernels that stage nothing pass LdsTables{}.
struct LdsTables {
    const F4* mats = nullptr;    // [n_mats] Material records
    uint32_t n_mats = 0;
    const F4* ltris = nullptr;   // [n_ltris] LightTri records
    uint32_t n_ltris = 0;
};
static constexpr uint32_t kLdsMats = 32, kLdsLtris = 8;   // 32 x 32 B + 8 x 96 B = 1.75 KB per block
RPT_DEV Mat load_mat(const SceneView& scene_, uint32_t obj, const LdsTables& tab = LdsTables{}) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    Material m;
    if (obj < tab.n_mats) {   // wave-uniform in effect: either every object's material is staged or none
        m.albedo_emit = tab.mats[2u * obj];
        m.params = tab.mats[2u * obj + 1u];
    } else {
        m = sc.mats[obj];
    }
    return Mat{xyz(m.albedo_emit), m.albedo_emit.w, __float_as_uint(m.params.x), m.params.y, m.params.z};
}
RPT_DEV V mat_color(const Mat& m) { return (m.kind <= 1u) ? m.albedo : mk(0.f, 0.f, 0.f); }   // src/material.rs:107
RPT_DEV float mat_emit(const Mat& m) { return (m.kind <= 1u) ? m.emit : 0.f; }                // src/material.rs:100

// Rotation taking +Y onto `b` applied to v (nalgebra rotation_between(+Y, b)): axis k =
// normalize(Y x b), angle acos(b.y).  `pi_fallback_x`: Lambertian's (0,1,1e-8) retry
// (src/material.rs:186-194) is a half-turn about +X; Phong's quat_rotation falls back to identity.
RPT_DEV V rotate_from_y(V b, V v, bool pi_fallback_x) {
    float s2 = fmaf(b.x, b.x, b.z * b.z);
    if (s2 > 0.f) {
        float is = rsq(s2);
        float s = s2 * is;           // sin(angle)
        float kx = b.z * is, kz = -b.x * is;  // k = (b.z, 0, -b.x)/s
        float kv = kx * v.x + kz * v.z;       // k . v
        // k x v = (-kz*v.y, kz*v.x - kx*v.z, kx*v.y)
        V kxv = mk(-kz * v.y, kz * v.x - kx * v.z, kx * v.y);
        float c = b.y;
        float omc = 1.f - c;
        return mk(fmaf(c, v.x, fmaf(s, kxv.x, omc * kv * kx)), fmaf(c, v.y, s * kxv.y),
                  fmaf(c, v.z, fmaf(s, kxv.z, omc * kv * kz)));
    }
    if (b.y < 0.f && pi_fallback_x) return mk(v.x, -v.y, -v.z);
    return v;
}
RPT_DEV V reflect_neg(V w, V n) { return fma3(2.f * dot(n, w), n, -w); }  // -glm::reflect_vec(w, n)

// Material::sample_f, src/material.rs:166-263.  Returns false for None (total internal reflection).
RPT_DEV bool sample_f(const Mat& m, V n, V wo, Rng& rng, V& wi, float& pdf) {
    if (m.kind == M_LAMBERTIAN) {
        float r1 = rng.uniform(), r2 = rng.uniform();
        float ct = sqrt1(r2);             // cos(acos(sqrt(r2)))
        float st = sqrt1(1.f - r2);
        pdf = ct * kInvPi;
        V dir = mk(st * __builtin_amdgcn_cosf(r1), ct, st * __builtin_amdgcn_sinf(r1));  // phi = 2 pi r1
        wi = normalize(rotate_from_y(n, dir, true));
        return true;
    } else if (m.kind == M_PHONG) {
        float r1 = rng.uniform(), r2 = rng.uniform();
        float ct = __powf(r2, rcp(m.shin + 1.f));
        float st = sqrt1(fmaxf(1.f - ct * ct, 0.f));
        pdf = (m.shin + 1.f) * (0.5f * kInvPi) * __powf(ct, m.shin);
        V dir = mk(st * __builtin_amdgcn_cosf(r1), ct, st * __builtin_amdgcn_sinf(r1));
        V refl = reflect_neg(wo, n);
        wi = normalize(rotate_from_y(normalize(refl), dir, false));
        return true;
    } else if (m.kind == M_MIRROR) {
        wi = reflect_neg(wo, normalize(n));
        pdf = 1.f;
        return true;
    } else {
        bool inside = dot(n, wo) < 0.f;
        V nn = inside ? -n : n;
        float ci = fminf(fmaxf(dot(wo, nn), 0.f), 1.f);
        float ni = inside ? m.ior : 1.f, nt = inside ? 1.f : m.ior;
        float r0 = (ni - nt) * rcp(ni + nt);
        r0 *= r0;
        float mm = 1.f - ci;
        float m2 = mm * mm;
        float sr = fminf(fmaxf(fmaf(1.f - r0, m2 * m2 * mm, r0), 0.f), 1.f);
        pdf = 1.f;
        if (rng.uniform() < sr) {
            wi = reflect_neg(wo, n);
            return true;
        }
        float eta = ni * rcp(nt);
        float k = 1.f - eta * eta * (1.f - ci * ci);
        if (k < 0.f) return false;  // sqrt -> NaN -> None
        float cost = sqrt1(k);
        wi = fma3(eta * ci - cost, nn, -eta * wo);
        return true;
    }
}
// Material::bsdf, src/material.rs:266-289.
RPT_DEV V bsdf(const Mat& m, V n, V wo, V wi) {
    float nwi = dot(n, wi), nwo = dot(n, wo);
    if (__builtin_signbit(nwi) || __builtin_signbit(nwo)) return mk(0.f, 0.f, 0.f);
    if (m.kind == M_LAMBERTIAN) return kInvPi * m.albedo;
    if (m.kind == M_PHONG) {
        V r = -normalize(fma3(-2.f * dot(n, wi), n, wi));
        float c = fminf(fmaxf(dot(r, wo), 0.f), 1.f);
        // powf(0, s) = 0 for s > 0, 1 for s == 0
        float pw = (c > 0.f) ? __powf(c, m.shin) : (m.shin == 0.f ? 1.f : 0.f);
        return ((m.shin + 2.f) * (0.5f * kInvPi) * pw) * m.albedo;
    }
    return mk(1.f, 1.f, 1.f);
}

// ------------------------------------------------------------------ lights
// Light::illuminate for Light::Object, src/light.rs:34-45, over the shape samplers
// (src/kdtree.rs:141-146 + src/shape/mesh.rs:85-99, sphere.rs:53-65, cube.rs:76-89,
//  Transformed::sample src/shape.rs:140-151).
// Shape::sample of the light's shape: point v, normal n, pdf p (per world area).
// The transform of the sampled leaf.  A plain Light::Object is wave-uniform: the whole record sits in SGPRs.
// A leaf picked out of a group differs per lane: its rows are loaded where they are used, so that the 48
// floats never live in VGPRs at once.
template <bool UNIFORM> struct LightXfRows;
template <> struct LightXfRows<true> {
    LightXf x;
    RPT_DEV explicit LightXfRows(const LightXf* p) : x(uload(p)) {}
    RPT_DEV F4 fwd(int r) const { return x.fwd[r]; }
    RPT_DEV F4 inv(int r) const { return x.inv[r]; }
    RPT_DEV F4 nrm(int r) const { return x.nrm[r]; }
    RPT_DEV F4 lin(int r) const { return x.lin[r]; }
};
template <> struct LightXfRows<false> {
    const LightXf* p;
    RPT_DEV explicit LightXfRows(const LightXf* q) : p(q) {}
    RPT_DEV F4 fwd(int r) const { return p->fwd[r]; }
    RPT_DEV F4 inv(int r) const { return p->inv[r]; }
    RPT_DEV F4 nrm(int r) const { return p->nrm[r]; }
    RPT_DEV F4 lin(int r) const { return p->lin[r]; }
};
// One leaf shape of a Light::Object (Sphere/Cube/Mesh::sample under Transformed::sample, src/shape.rs:140-151).
template <bool UNIFORM>
RPT_DEV void sample_light_leaf(const SceneView& scene_, uint32_t shape, uint32_t first, uint32_t count, const LightXf* xp,
                               V pos, Rng& rng, V& v, V& n, float& p, const LdsTables& tab = LdsTables{}) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    V vl, nl;
    const LightXfRows<UNIFORM> x(xp);
    const bool xf = x.nrm(1).w != 0.f;
    const bool mesh = shape == LS_MESH;
    if (mesh) {
        uint32_t idx = rng.index(count);
        LightTri tr;
        if (first + idx < tab.n_ltris) {
            const F4* q = tab.ltris + 6u * (first + idx);
            tr.v1 = q[0]; tr.v2 = q[1]; tr.v3 = q[2]; tr.n1 = q[3]; tr.n2 = q[4]; tr.n3 = q[5];
        } else {
            tr = sc.ltris[first + idx];
        }
        // `while u + v > 1 { redraw }` (src/shape/mesh.rs:89-93) on the draws' integers: with u = (2k+1) 2^-24 the sum is exact
        // in fp32 and u + v > 1 <=> ku + kv >= 2^23, so the rejected pairs are never converted (the wave runs the loop for its
        // unluckiest lane: ~7 rounds for 2 on average)
        uint32_t ku = rng.next() >> 9, kv = rng.next() >> 9;
        while (ku + kv >= (1u << 23)) {
            ku = rng.next() >> 9;
            kv = rng.next() >> 9;
        }
        const float u = float((ku << 1) | 1u) * 0x1p-24f, vv = float((kv << 1) | 1u) * 0x1p-24f;
        float w = 1.f - u - vv;
        vl = u * xyz(tr.v1) + vv * xyz(tr.v2) + w * xyz(tr.v3);  // already world space
        nl = normalize(u * xyz(tr.n1) + vv * xyz(tr.n2) + w * xyz(tr.n3));
        p = tr.v1.w * rcp(float(count));
    } else if (shape == LS_SPHERE) {
        V tl = xf ? mk(dot3w(x.inv(0), pos), dot3w(x.inv(1), pos), dot3w(x.inv(2), pos)) : pos;
        float dx, dy;
        rng.unit_disc(dx, dy);
        float z = sqrt1(fmaxf(1.f - dx * dx - dy * dy, 0.f));
        V nn = normalize(tl);
        bool normal_x = fabsf(nn.x) >= 1.17549435e-38f && fabsf(nn.x) < kInf;
        V n1 = normal_x ? normalize(mk(nn.y, -nn.x, 0.f)) : normalize(mk(0.f, -nn.z, nn.y));
        V n2 = cross(n1, nn);
        vl = dx * n1 + dy * n2 + z * nn;
        nl = vl;
        p = z * kInvPi;
    } else {
        float a = rng.uniform() - 0.5f, b = rng.uniform() - 0.5f;
        uint32_t f = rng.index(6);
        switch (f) {
            case 0: vl = mk(a, b, 0.5f); nl = mk(0, 0, 1); break;
            case 1: vl = mk(a, b, -0.5f); nl = mk(0, 0, -1); break;
            case 2: vl = mk(a, 0.5f, b); nl = mk(0, 1, 0); break;
            case 3: vl = mk(a, -0.5f, b); nl = mk(0, -1, 0); break;
            case 4: vl = mk(0.5f, a, b); nl = mk(1, 0, 0); break;
            default: vl = mk(-0.5f, a, b); nl = mk(-1, 0, 0); break;
        }
        p = 1.f / 6.f;
    }
    if (xf) {
        n = normalize(mk(dot3(x.nrm(0), nl), dot3(x.nrm(1), nl), dot3(x.nrm(2), nl)));
        V ln = mk(dot3(x.lin(0), nl), dot3(x.lin(1), nl), dot3(x.lin(2), nl));
        float height = dot(ln, n);
        float base = x.nrm(0).w * rcp(height);
        v = mesh ? vl : mk(dot3w(x.fwd(0), vl), dot3w(x.fwd(1), vl), dot3w(x.fwd(2), vl));
        p = p * rcp(base);
    } else {
        v = vl;
        n = nl;
    }
}
// Shape::sample of a Light::Object.  A KdTree group (src/kdtree.rs:141-146) samples a uniformly chosen child and
// divides its pdf by the child count, nested groups repeat that: every lane descends to its own leaf.
template <bool GROUPS>
RPT_DEV void sample_light_shape(const SceneView& scene_, const Light& L, V pos, Rng& rng, V& v, V& n, float& p,
                                const LdsTables& tab = LdsTables{}) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    if (!GROUPS || L.shape != LS_GROUP) {  // wave-uniform
        sample_light_leaf<true>(scene_, L.shape, L.first, L.count, &sc.lxf[L.xf], pos, rng, v, n, p, tab);
        return;
    }
    uint32_t shape = LS_GROUP, first = L.first, count = L.count, xfi = 0;
    float pick = 1.f;
    while (shape == LS_GROUP) {
        const uint32_t idx = rng.index(count);
        pick *= rcp(float(count));
        const LightPart part = sc.lparts[first + idx];
        shape = part.shape; first = part.first; count = part.count; xfi = part.xf;
    }
    sample_light_leaf<false>(scene_, shape, first, count, &sc.lxf[xfi], pos, rng, v, n, p, tab);
    p *= pick;
}
template <bool GROUPS>
RPT_DEV void illuminate_object(const SceneView& sc, const Light& L, V pos, Rng& rng, V& intensity, V& wi,
                               float& dist, const LdsTables& tab = LdsTables{}) {
    V v, n;
    float p;
    sample_light_shape<GROUPS>(sc, L, pos, rng, v, n, p, tab);
    V disp = v - pos;
    float len2 = dot(disp, disp);
    float ilen = rsq(len2);
    float cosine = fmaxf(-dot(disp, n), 0.f) * ilen;
    float surface_area = cosine * rcp(len2);
    intensity = (surface_area * rcp(p)) * xyz(L.color);
    wi = ilen * disp;
    dist = len2 * ilen;
}

// ------------------------------------------------------------------ environment
// Environment::get_color, src/environment.rs:72-77; Hdri::get_color / bilinear_sample :25-52.
RPT_DEV V env_color(const SceneView& scene_, V dir) {
    const auto& sc = *kernarg_scene();   // (see kernarg_scene)
    if (sc.hdri_w == 0) return mk(sc.env[0], sc.env[1], sc.env[2]);
    const V d = normalize(dir);
    const float azimuth = atan2f(d.z, d.x) + kPi;
    const float polar = acosf(fminf(fmaxf(d.y, -1.f), 1.f));
    const float x = azimuth * (0.5f * kInvPi) * float(sc.hdri_w - 1u);
    const float y = polar * kInvPi * float(sc.hdri_h - 1u);
    const uint32_t x0 = min(uint32_t(x), sc.hdri_w - 1u), y0 = min(uint32_t(y), sc.hdri_h - 1u);
    const float ax = x - float(x0), ay = y - float(y0);
    const uint32_t x1 = min(x0 + 1u, sc.hdri_w - 1u), y1 = min(y0 + 1u, sc.hdri_h - 1u);  // weight 0 where clamped
    const V c00 = xyz(sc.hdri[y0 * sc.hdri_w + x0]), c01 = xyz(sc.hdri[y0 * sc.hdri_w + x1]);
    const V c10 = xyz(sc.hdri[y1 * sc.hdri_w + x0]), c11 = xyz(sc.hdri[y1 * sc.hdri_w + x1]);
    const V top = (1.f - ax) * c00 + ax * c01, bot = (1.f - ax) * c10 + ax * c11;
    return (1.f - ay) * top + ay * bot;
}

// ------------------------------------------------------------------ camera
// Camera::cast_ray, src/camera.rs:65-82 (cot(fov/2)*direction and `right` hoisted to the host).
RPT_DEV void cast_ray(const CameraG& c, float x, float y, Rng& rng, V& o, V& d) {
    V right = mk(c.right[0], c.right[1], c.right[2]), up = mk(c.up[0], c.up[1], c.up[2]);
    o = mk(c.eye[0], c.eye[1], c.eye[2]);
    V nd = mk(c.ddir[0], c.ddir[1], c.ddir[2]) + x * right + y * up;
    if (c.aperture > 0.f) {
        V focal = fma3(c.focal_distance, normalize(nd), o);
        float dx, dy;
        rng.unit_disc(dx, dy);
        o = o + c.aperture * (dx * right + dy * up);
        nd = focal - o;
    }
    d = normalize(nd);
}

}  // namespace rptg
