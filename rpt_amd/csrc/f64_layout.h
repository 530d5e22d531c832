// f64_layout.h — scene layout of the reference-epsilon mode (option "epsilon_policy" = 1; kernels_f64.hip).
//
// The fp32 path replaces rpt's 1e-12 epsilons (src/renderer.rs:17, 348, 396, 420) by tolerances fp32 can resolve and
// specialises primitives at flatten time.  This mode does neither: every object is the reference's generic shape under
// its own `Transformed` matrices, in fp64, tested in scene order with t_min = 1e-12, and the shadow test is
// |hit - dist| < 1e-12.  It exists for callers who need rpt's own numbers (its self-hits and false shadow rejections
// included), not its speed.
#pragma once
#include <stdint.h>

namespace rpt64 {

enum : int32_t { SH_SPHERE = 0, SH_CUBE = 1, SH_PLANE = 2, SH_MESH = 3 };
enum : int32_t { LT_POINT = 0, LT_AMBIENT = 1, LT_DIRECTIONAL = 2, LT_OBJECT = 3 };

// One `Box<dyn Shape>`: unit primitive / plane / mesh, optionally under Transformed<T> (src/shape.rs:102-152).
struct Shape {
    int32_t kind, has_xf;
    uint32_t tri_first, tri_count;   // SH_MESH: triangles in the mesh's own (local) space, in the order they were given
    double inv[12];     // rows of M^-1 (3 x 4)            Transformed::inverse_transform
    double fwd[12];     // rows of M (3 x 4)               Transformed::transform
    double lin[9];      // linear part of M                Transformed::linear
    double nrm[9];      // (linear)^-T                     Transformed::normal_transform
    double det;         // det(linear)                     Transformed::scale
    double plane[4];    // SH_PLANE: normal, value
    double bmin[3], bmax[3];   // SH_MESH: KdTree::bounds (src/kdtree.rs:108-113)
};
struct Tri {   // src/shape/mesh.rs:9-23
    double v1[3], v2[3], v3[3], n1[3], n2[3], n3[3];
};
struct Mat {   // src/material.rs:8-23
    int32_t kind, _pad;
    double albedo[3], emittance, shininess, ior;
};
struct Object {
    Shape shape;
    Mat mat;
};
struct Light {   // src/light.rs:7-19
    int32_t kind, _pad;
    double color[3];   // Ambient / Point / Directional colour
    Object obj;        // Light::Object
};
struct Scene {
    const Object* objects;
    const Tri* tris;
    const Light* lights;
    uint32_t n_objects, n_lights;
    int32_t has_medium, medium_kind;
    double absorption, scattering;
    double env[3];
};
struct Camera {   // src/camera.rs:9-27, with `d` and `right` of cast_ray (:67-68) evaluated once, in fp64, on the host
    double eye[3], direction[3], up[3], right[3];
    double d, aperture, focal_distance;
};
struct Args {
    Scene sc;
    Camera cam;
    uint32_t width, height, iterations, sample_offset, max_bounces;
    uint32_t n_owned, tiles_x, _pad;
    const uint32_t* tiles;
    uint64_t seed_mixed;
    double medium_color[3], medium_color_hi[3];   // Medium::color: hex_color(0xD2B48C), or blue (y <= 250) / red for the glowing fog
    double dim;       // max(width, height) as f64 (src/renderer.rs:174)
    double scale;     // 2^exposure_value
    double* out;      // width * height * 3
    unsigned long long* counters;   // [0] rays [1] accepted hits [2] self hits [3] shadow tests [4] passed [5] near misses [6] samples [7] vertices; or null
};

}  // namespace rpt64
