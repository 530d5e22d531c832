// Test infrastructure (see hip_stubs.h): builds a set of scenes through the C++ mirror, commits them with the real
// rpt_capi.cpp and prints the flattened-layout statistics plus a checksum of every tree node and scan record.
// tests/test_host_flatten.py compiles it with sanitizers and at several optimisation levels: the output must not
// depend on the compiler (a loop whose result changed with -fno-unroll-loops exposed undefined behaviour once).
#include "hip_stubs.h"
#include "../../rpt_amd/csrc/rpt_capi.cpp"
namespace rptg {
hipError_t launch_render(const RenderArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t render_occupancy(bool, int, int* b, int) { *b = 4; return hipSuccess; }
size_t stream_scratch_bytes_per_block() { return 0; }
int bvh_mode(const SceneView& sc) { return sc.scene_bvh ? 2 : (sc.n_nodes ? 1 : 0); }
hipError_t launch_resolve(const RenderArgs&, double, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_render_f64(const rpt64::Args&, int, hipStream_t) { return hipSuccess; }
hipError_t launch_resolve_f64(const rpt64::Args&, double, double*, hipStream_t) { return hipSuccess; }
hipError_t render_f64_occupancy(bool, int* blocks_per_cu) { *blocks_per_cu = 4; return hipSuccess; }
hipError_t launch_intersect(const SceneView&, uint64_t, const float*, const float*, float*, int32_t*, float*, bool, hipStream_t) { return hipSuccess; }
hipError_t launch_debug_rng(uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_debug_sample_f(const Material&, uint64_t, const float*, const float*, uint64_t, float*, float*, int32_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_debug_bsdf(const Material&, uint64_t, const float*, const float*, const float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_debug_camera(const CameraG&, uint32_t, uint32_t, uint64_t, uint32_t, float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_buffer_add(uint32_t, const double*, double*, double*, hipStream_t) { return hipSuccess; }
hipError_t launch_buffer_image(uint32_t, uint32_t, uint32_t, uint32_t, const double*, uint8_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_buffer_variance(uint32_t, uint32_t, const double*, const double*, double*, hipStream_t) { return hipSuccess; }
}
namespace rpti { void photon_release(void*) {} }

#include <cstdio>
#include "../../include/rpt.hpp"

static uint64_t fnv(const void* p, size_t n, uint64_t h = 0xCBF29CE484222325ULL) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ULL; }
    return h;
}
static void report(const char* name, rpt_scene* s) {
    const int rc = rpt_scene_commit(s, 0);
    uint64_t st[16] = {0};
    rpt_scene_stats(s, st);
    uint64_t h = 0;
    if (rc == 0 && s->arena) h = fnv(s->arena, size_t(st[9]));   // every array of the flattened scene
    std::printf("%s rc=%d", name, rc);
    for (int i = 0; i < 15; i++) std::printf(" %llu", (unsigned long long)st[i]);
    std::printf(" arena=%016llx\n", (unsigned long long)h);
}
static rpt::Shape torus(int nu, int nv, double R, double r) {
    std::vector<rpt::Triangle> ts;
    auto P = [&](int i, int j) {
        const double u = 6.283185307179586 * i / nu, v = 6.283185307179586 * j / nv;
        return rpt::Vec3{(R + r * std::cos(v)) * std::cos(u), r * std::sin(v), (R + r * std::cos(v)) * std::sin(u)};
    };
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            ts.push_back(rpt::Triangle::from_vertices(P(i, j), P(i + 1, j), P(i + 1, j + 1)));
            ts.push_back(rpt::Triangle::from_vertices(P(i, j), P(i + 1, j + 1), P(i, j + 1)));
        }
    return rpt::mesh(ts);
}
static void add(rpt_scene* s, const rpt::Shape& sh, bool light = false) {
    rpt_shape_desc d = sh.desc();
    rpt_material m{};
    m.kind = RPT_MAT_LAMBERTIAN;
    m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.7;
    m.emittance = light ? 10.0 : 0.0;
    if (light) { if (rpt_scene_add_light_object(s, &d, &m) != 0) std::printf("light rejected: %s\n", rpt_last_error()); }
    else if (rpt_scene_add_object(s, &d, &m) < 0) std::printf("object rejected: %s\n", rpt_last_error());
}
int main() {
    using namespace rpt;
    {   // a room of five walls, two boxes (one rotated), a light quad that is also an object: box shell + aabb + cube
        rpt_scene* s = rpt_scene_create();
        const double X = 556, Y = 548.9, Z = 559.2;
        add(s, polygon({{0, 0, 0}, {0, 0, Z}, {X, 0, Z}, {X, 0, 0}}));
        add(s, polygon({{0, Y, 0}, {X, Y, 0}, {X, Y, Z}, {0, Y, Z}}));
        add(s, polygon({{0, 0, Z}, {0, Y, Z}, {X, Y, Z}, {X, 0, Z}}));
        add(s, polygon({{0, 0, 0}, {0, Y, 0}, {0, Y, Z}, {0, 0, Z}}));
        add(s, polygon({{X, 0, 0}, {X, 0, Z}, {X, Y, Z}, {X, Y, 0}}));
        add(s, cube().scale({165, 330, 165}).rotate_y(0.3).translate({368, 165, 351}));
        add(s, cube().scale({165, 165, 165}).translate({185, 82.5, 169}));
        Shape quad = polygon({{213, 548.8, 227}, {343, 548.8, 227}, {343, 548.8, 332}, {213, 548.8, 332}});
        add(s, quad);
        add(s, quad, true);
        report("room", s);
        rpt_scene_destroy(s);
    }
    {   // a mesh with its own tree under a transform, a plane, spheres
        rpt_scene* s = rpt_scene_create();
        add(s, torus(32, 32, 0.3, 0.12).scale({3.4, 3.4, 3.4}).rotate_y(1.5707963267948966));
        add(s, plane({0, 1, 0}, -1.0));
        add(s, sphere().scale({0.5, 0.7, 0.5}).translate({2, 0, 0}));
        add(s, torus(3, 3, 0.3, 0.12).translate({0, 2, 0}));   // 18 triangles: scanned linearly
        report("mesh", s);
        rpt_scene_destroy(s);
    }
    {   // 110 shapes in groups, one mesh shared by 40 of them: scene tree + instancing
        rpt_scene* s = rpt_scene_create();
        Shape shared = torus(12, 8, 0.6, 0.25);
        std::vector<Shape> kids;
        for (int i = 0; i < 40; i++) kids.push_back(shared.scale({0.2, 0.2, 0.2}).rotate_x(0.1 * i).translate({0.5 * (i % 8), 0.4 * (i / 8), 0.1 * i}));
        for (int i = 0; i < 30; i++) kids.push_back(sphere().scale({0.1, 0.1, 0.1}).translate({-0.3 * i, 0.2, 0.1}));
        std::vector<Shape> inner;
        for (int i = 0; i < 40; i++) inner.push_back(cube().scale({0.1, 0.2, 0.1}).rotate_z(0.05 * i).translate({0.1 * i, -1.0, 0.3}));
        kids.push_back(kdtree(inner).translate({0, 0, 2}));
        add(s, kdtree(kids).rotate_y(0.2));
        add(s, plane({0, 0, 1}, -6.0));
        // the nested group once more as a Light::Object: parts tree, a transform and triangles per leaf
        std::vector<Shape> lamp = {sphere().translate({0, 3, 0}), shared.translate({1, 3, 0}), kdtree(inner).scale({2, 2, 2})};
        add(s, kdtree(lamp).rotate_x(0.3), true);
        report("groups", s);
        rpt_scene_destroy(s);
    }
    {   // the reference-epsilon mode's scene (f64_layout.h): generic shapes, groups as records under frames, a shared mesh stored once,
        // per-triangle constants evaluated with the reference's operations (no contraction: every build must find the same bits)
        rpt_scene* s = rpt_scene_create();
        rpt_scene_set_option(s, "epsilon_policy", 1);
        add(s, polygon({{0, 0, 0}, {0, 0, 5.5}, {5.5, 0, 5.5}, {5.5, 0, 0}}));
        add(s, cube().scale({1.65, 3.3, 1.65}).rotate_y(0.3).translate({3.68, 1.65, 3.51}));
        add(s, plane({0, 1, 0}, -0.25));
        Shape shared = torus(9, 7, 0.6, 0.25);
        std::vector<Shape> inner = {sphere().scale({0.3, 0.2, 0.3}).translate({1, 1, 1}), shared.rotate_x(0.7).translate({2, 1, 0.5}), shared.scale({0.5, 0.5, 0.5})};
        std::vector<Shape> outer = {kdtree(inner).rotate_z(0.2).translate({0.1, 0.2, 0.3}), cube().scale({0.3, 0.3, 0.3}).translate({4, 0.5, 1}),
                                    kdtree({sphere().translate({0, 3, 0})})};
        add(s, kdtree(outer).scale({0.9, 1.1, 1.0}));
        Shape quad = polygon({{2.13, 5.4, 2.27}, {3.43, 5.4, 2.27}, {3.43, 5.4, 3.32}, {2.13, 5.4, 3.32}});
        add(s, quad);
        add(s, quad, true);
        const int rc = rpt_scene_commit(s, 0);
        std::printf("epsilon rc=%d records=%u arena64=%016llx bytes=%zu\n", rc, s->view64.n_objects,
                    (unsigned long long)(rc == 0 && s->arena64 ? fnv(s->arena64, s->arena64_bytes) : 0ull), s->arena64_bytes);
        rpt_scene_destroy(s);
    }
    {   // error paths: nothing may be read out of bounds or leak
        rpt_scene* s = rpt_scene_create();
        rpt_shape_desc bad{};
        bad.kind = 9;
        rpt_material m{};
        std::printf("bad kind rc=%d\n", rpt_scene_add_object(s, &bad, &m));
        bad.kind = RPT_SHAPE_MESH;
        std::printf("empty mesh rc=%d\n", rpt_scene_add_object(s, &bad, &m));
        bad.kind = RPT_SHAPE_SPHERE;
        bad.has_transform = 1;   // all-zero matrix
        std::printf("singular rc=%d\n", rpt_scene_add_object(s, &bad, &m));
        report("empty", s);
        std::printf("second commit rc=%d\n", rpt_scene_commit(s, 0));
        rpt_scene_destroy(s);
    }
    return 0;
}
