// f64_layout.h — scene layout of the reference-epsilon mode (option "epsilon_policy" = 1; kernels_f64.hip).
//
// The fp32 path replaces rpt's 1e-12 epsilons (src/renderer.rs:17, 348, 396, 420) by tolerances fp32 can resolve and
// specialises primitives at flatten time.  This mode does neither: every object is the reference's generic shape under
// its own `Transformed` matrices, in fp64, tested in scene order with t_min = 1e-12, and the shadow test is
// |hit - dist| < 1e-12: a caller gets rpt's own numbers, its self-hits and false shadow rejections included.
//
// What is MI355X about it (round 4): the fp64 arithmetic of an object is the reference's, but it is only run for the
// objects a ray can reach.  Every object carries a padded world-space box in fp32 (`CullBox`, read through the scalar
// cache); a lane first tests its ray against all of them at the full fp32 rate and keeps a bit mask of candidates, then
// evaluates its own candidates -- a different object in every lane -- in scene order from per-lane records (`ObjRec`,
// `TriRec`, staged in LDS when the scene is small).  An object the box test rejects cannot change the reference's
// HitRecord (its literal test would miss, or hit no closer than what is already known to decide the event), so the
// closest hit, its time and its normal are bit for bit those of the full scan (tests: culling on = off).
#pragma once
#include <stdint.h>

namespace rpt64 {

enum : int32_t { SH_SPHERE = 0, SH_CUBE = 1, SH_PLANE = 2, SH_MESH = 3, SH_GROUP = 4 };   // (SH_GROUP: in a light's shape tree only)
enum : int32_t { LT_POINT = 0, LT_AMBIENT = 1, LT_DIRECTIONAL = 2, LT_OBJECT = 3 };

// One `Box<dyn Shape>`: unit primitive / plane / mesh, optionally under Transformed<T> (src/shape.rs:102-152).  The full
// record: what Shape::sample of a light needs (wave-uniform, scalar loads).
struct Shape {
    int32_t kind, has_xf;
    uint32_t tri_first, tri_count;   // SH_MESH: triangles in the mesh's own (local) space, in the order they were given;
                                     // SH_GROUP (a `KdTree<Box<dyn Bounded>>` as a Light::Object): its children in Scene::lshapes
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
struct Light {   // src/light.rs:7-19
    int32_t kind, _pad;
    double color[3];   // Ambient / Point / Directional colour
    Shape shape;       // Light::Object
    Mat mat;
};

// ---- what a closest-hit query reads
// Padded world-space bounds of one object in fp32 (planes: unbounded).  32 bytes, wave-uniform: one s_load_dwordx8.
struct CullBox {
    float lo[3];
    uint32_t unbounded;
    float hi[3];
    uint32_t slab_test;   // 1: a cube -- its intersection is a slab test that a 0 / 0 corrupts (kernels_f64.hip, cull32); 2: a union box (Scene::cull32) with a cube in it
};
// What Shape::intersect of one object needs, read per lane (every lane may hold a different object): 176 bytes.
// A shape inside `KdTree<Box<dyn Bounded>>` groups (src/kdtree.rs:103-146) is a record of its own -- the group's kd-tree is an
// acceleration structure, its children are tested like the triangles of a mesh --; `frame` lists the groups around it, outermost
// first: each transforms the ray (when the group itself is `Transformed`) and applies KdTree::intersect's bounds test.
static constexpr uint32_t kMaxFrames = 3u;
struct ObjRec {
    int32_t kind, has_xf;
    uint32_t tri_first, tri_count;
    uint32_t n_frames, frame[kMaxFrames];
    double inv[12];   // rows of M^-1 (3 x 4)
    double b[6];      // SH_MESH: KdTree::bounds min, max; SH_CUBE: -0.5 x 3, 0.5 x 3 (src/shape/cube.rs:25-26); SH_PLANE: normal, value
};
struct FrameRec {     // one group level: 144 bytes
    double inv[12];   // rows of the group's own M^-1 (has_xf)
    double b[6];      // KdTree::bounds of the group (src/kdtree.rs:108-113), in the group's own space
};
struct FrameShade {
    int32_t has_xf, _pad;
    double nrm[9];    // (linear)^-T of the group's own transform
};
// Triangle::intersect (src/shape/mesh.rs:50-83) with everything that depends on the triangle alone evaluated once, on
// the host, by the same IEEE operations in the same order (no contraction): the plane normal normalize(cross(d0, d1)),
// d00, d01, d11 and denom are bit for bit what the reference recomputes per call.  128 bytes.
struct TriRec {
    double v1[3], pn[3], d0[3], d1[3];
    double d00, d01, d11, denom;
};
// ... and what shading the winning object needs (per lane, once per query)
struct ObjShade {
    double nrm[9];   // (linear)^-T
    Mat mat;
};
struct TriShade {
    double n1[3], n2[3], n3[3];
};

struct Scene {
    const CullBox* cull;        // [n_objects]: one per record (a group's children are records of their own)
    const CullBox* cull32;      // [ceil(n_objects / 32)]: the union of the boxes of records 32 g .. 32 g + 31 (unbounded if one of them is)
    const ObjRec* recs;         // [n_objects]
    const ObjShade* shade;      // [n_objects]
    const FrameRec* frames;     // group levels (ObjRec::frame)
    const FrameShade* fshade;
    const TriRec* trecs;        // [n_obj_tris]: the triangles of scene.objects' meshes
    const TriShade* tshade;     // [n_obj_tris]
    const Tri* tris;            // [n_tris]: vertices and normals as given -- the objects' triangles, then those of the lights' meshes (Triangle::sample)
    const double* tri_pdf;      // [n_tris]: (1 / area) / triangles of the mesh (src/shape/mesh.rs:96-98, src/kdtree.rs:141-146)
    const Light* lights;
    const Shape* lshapes;       // the shapes inside the groups of group lights (KdTree::sample picks one per sample, per lane)
    uint32_t n_objects, n_lights, n_tris, n_obj_tris;
    int32_t has_medium, medium_kind;
    double absorption, scattering;
    double env[3];
    const double* hdri;         // Environment::Hdri (src/environment.rs:3-52): width x height x 3, row-major; hdri_w = 0: Environment::Color(env)
    uint32_t hdri_w, hdri_h;
    // Where the cubes' faces lie, per axis, as 64 buckets over [face_base, face_base + 64 / face_inv_cell): a ray with a zero direction
    // component takes part in cull32's extra test only if its origin's coordinate on that axis falls into a marked bucket (every face
    // marks the buckets of its coordinate -+ the test's tolerance, so the lane's own bucket suffices).
    float face_base[3], face_inv_cell[3];
    unsigned long long face_bits[3];
};
struct Camera {   // src/camera.rs:9-27, with `d` and `right` of cast_ray (:67-68) evaluated once, in fp64, on the host
    double eye[3], direction[3], up[3], right[3];
    double d, aperture, focal_distance;
};
static constexpr uint32_t kLdsObjs = 32u, kLdsTris = 32u;   // tables staged in LDS when the scene has at most this many
struct Args {
    Scene sc;
    Camera cam;
    uint32_t width, height, iterations, sample_offset, max_bounces;
    uint32_t n_owned, tiles_x, n_items;
    uint32_t chunk_spp, n_chunks, pull_batch, cull;   // cull: 0 = every object is evaluated for every ray (the plain reference scan)
    uint32_t surf_batch;      // in a medium: lanes of a wave that wait at a surface event before the wave runs the surface code
    uint32_t group_lights;    // some Light::Object is a KdTree group (its own kernel instantiation)
    const uint32_t* tiles;
    uint64_t seed_mixed;
    double medium_color[3], medium_color_hi[3];   // Medium::color: hex_color(0xD2B48C), or blue (y <= 250) / red for the glowing fog
    double dim;       // max(width, height) as f64 (src/renderer.rs:174)
    unsigned long long* queue;      // work counter of the launch
    double* slab;                   // [n_chunks][n_owned][4]: partial sums of (pixel, chunk) items
    unsigned long long* counters;   // [0] rays [1] accepted hits [2] self hits [3] shadow tests [4] passed [5] near misses [6] samples [7] vertices
                                    // [8] objects evaluated [9] evaluation rounds (wave-level) [10] trips (wave-level) [11] live lanes summed over trips; or null
};

// ---- photon mapping in the reference-epsilon mode (kernels_f64.hip; the maps, the k-nearest selection and the volume estimates are
// photon.hip's, on the records in fp32)
struct PhotonRec32 {   // photon.hip's PhotonRec: position + gather radius, direction + index in shooting order (bits), power
    float pos_r[4], dir[4], pow[4];
};
// shoot_photon + trace_photon (src/photon.rs:724-946) with the reference's own epsilons: count pass (surf == null), then write pass.
struct ShootArgs64 {
    Args a;                     // scene, a.queue (photons handed out 64 at a time), a.seed_mixed, a.cull
    uint64_t n_photons;         // photons of THIS launch
    uint64_t first_photon;      // global index of its first photon (the RNG stream key)
    double power;               // watts / photon_count (of the whole map)
    uint32_t light_index;       // the first Light::Object
    uint32_t kind;              // RPT_PHOTON_*: the beam x beam map thins the volume photons and records each beam's start
    uint32_t* cnt_s;            // [n_photons] records per photon (count pass)
    uint32_t* cnt_v;
    const uint32_t* off_s;      // their exclusive prefix sums (write pass)
    const uint32_t* off_v;
    PhotonRec32* surf;
    PhotonRec32* vol;
    double* pos64;              // [3 per surface record, shooting order]: the position as the reference holds it
};
// The surface estimate of the camera pass (src/photon.rs:327-375): the gathered photons' visibility rays and terms, per sample.
struct SurfArgs64 {
    Args a;                     // a.iterations = samples of this slice, a.n_chunks = groups of 64 of them, a.slab [group][n_owned][4]
    const uint32_t* emit;       // [n_owned][K + 2][iterations]: what the fp32 camera pass selected per sample -- K photon indices
                                // (sorted order), their number, the squared distance of the farthest (float bits)
    const PhotonRec32* s_ph;    // the surface map's photons in sorted order
    const double* pos64;        // ShootArgs64::pos64 of the map
    uint32_t K;                 // gather_size
    uint32_t kind;              // RPT_PHOTON_*
    uint32_t skip;              // diagnostic (option "photon_skip"): 4096 = every gathered photon counts (no visibility rays)
};

}  // namespace rpt64
