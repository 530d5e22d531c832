// gpu_layout.h — flattened fp32 scene layout shared by the host flattener and the HIP kernels.
//
// HBM layout (all arrays 16-byte aligned, read-only during a render):
//   * "scan" arrays are walked by every lane in lock-step with a wave-uniform index, so the
//     compiler fetches them with scalar loads (s_load_dwordx4/x8) into SGPRs: 48 B per sphere /
//     cube (inverse affine rows), 16 B per plane, 48 B per triangle.
//   * "shade" arrays are indexed per lane by the winning primitive code after the scan
//     (normal transform, vertex normals, object -> material).
//   * large meshes are a BVH2 (64-byte two-box nodes) + the same 48-byte triangle records, walked per lane.
#pragma once
#include <stdint.h>

namespace rptg {

struct alignas(16) F4 {
    float x, y, z, w;
};

// Primitive code = (kind << 28) | index-within-kind.  0xFFFFFFFF = miss.
enum : uint32_t { K_SPHERE = 0, K_CUBE = 1, K_PLANE = 2, K_TRI = 3, K_BVHTRI = 4, K_AABB = 5, K_RECT = 6,
                  K_INST = 7,      // scene-BVH leaf item: one instance of a shared local-space mesh
                  K_INSTTRI = 8 }; // hit code: triangle of an instanced mesh (the instance travels beside the code)
static const uint32_t CODE_MISS = 0xFFFFFFFFu;

// Sphere / cube scan record: rows of the inverse affine map (world -> unit primitive).
// For a bare (untransformed) shape this is the identity.
struct alignas(16) XfScan {
    F4 r0, r1, r2;  // local = (r.xyz . p) + r.w
};
// Sphere / cube shade record: rows of the normal transform (M^-T of the linear part);
// r0.w = object id (bits), r1.w = 1.0f if wrapped in Transformed (normal is re-normalised).
struct alignas(16) XfShade {
    F4 r0, r1, r2;
};
// Plane: world-space (n, value) with the reference's |n.d| < 1e-8 parallel test; shade record
// holds the unit normal and the object id.
struct alignas(16) PlaneScan {
    F4 nv;
};
struct alignas(16) PlaneShade {
    F4 unit_n_obj;
};
// World-space triangle, pre-solved: t = (pn.w - pn.o) / (pn.d); P = o + t d;
// v = A.P + A.w, w = B.P + B.w, u = 1 - v - w  (same barycentrics as src/shape/mesh.rs:62-73).
struct alignas(16) TriScan {
    F4 pn, A, B;
};
// Vertex normals (already multiplied by M^-T for Transformed<Mesh>); n1.w = object id (bits).
struct alignas(16) TriShade {
    F4 n1, n2, n3;
};
// Flatten-time specialisations (same hits, normals and object ids as the generic records):
//  * AabbScan: a cube whose transform is a positive scale + translation, i.e. an axis-aligned box
//    in world space; tested with the ray's shared 1/d (no per-box affine map, no per-box rcp).
//    lo.w = object id (bits).  The face choice follows src/shape/cube.rs:37-55 unchanged.
//  * RectScan: two coplanar flat triangles of one mesh forming an axis-aligned rectangle
//    (polygon() of 4 corners, e.g. every Cornell wall): plane coordinate c on `axis`, bounds on
//    the other two axes in cyclic order (u = axis+1, v = axis+2).  Records are sorted by axis.
//    b.z = object id (bits), b.w unused; the flat world normal lives in RectShade.
struct alignas(16) AabbScan {
    F4 lo, hi;
};
struct alignas(16) RectScan {
    F4 a;  // c, umin, umax, vmin
    F4 b;  // vmax, -, obj, -
};
// Rectangles that are exactly the faces of one axis-aligned box (the walls of a room: every Cornell-box
// config) are tested together: a line meets a convex box's boundary at its slab entry and exit only, so
// one slab test replaces up to six rectangle tests.  face[2*axis + side] = hit code of the rectangle on
// that face (side 0 = lo, 1 = hi) or CODE_MISS for an open face.
struct alignas(16) ShellScan {
    F4 lo, hi;         // w unused
    uint32_t face[8];  // [6..7] unused
};
struct alignas(16) RectShade {
    F4 n_obj;  // unit normal, object id (bits)
};
// BVH2 node holding BOTH children's boxes (64 B): one dependent load per inner node yields two
// slab tests, the near child is descended first and leaves are referenced directly.
// Child entry: bit 31 = leaf; leaf: bits 26..30 = item count - 1 (1..32), bit 25 = BVH_PRIMS,
// bits 0..24 = first item: an index into `btri` (triangle leaf) or into `pleaf` (BVH_PRIMS: a
// list of primitive codes kind << 28 | index, scene-level BVH only).  Inner: absolute node index.
struct alignas(16) BvhNode {
    float lo0[3];
    uint32_t e0;
    float hi0[3];
    uint32_t pad0;
    float lo1[3];
    uint32_t e1;
    float hi1[3];
    uint32_t pad1;
};
static const uint32_t BVH_LEAF = 0x80000000u;
static const uint32_t BVH_PRIMS = 0x02000000u;
static const uint32_t BVH_INDEX_MASK = 0x01FFFFFFu;
enum : uint32_t { M_LAMBERTIAN = 0, M_PHONG = 1, M_MIRROR = 2, M_TRANSMISSIVE = 3 };
struct alignas(16) Material {
    F4 albedo_emit;  // rgb albedo, emittance
    F4 params;       // kind (bits), shininess, ior, unused
};
// Light record (scene.lights order is preserved: it fixes the RNG draw order).
enum : uint32_t { L_POINT = 0, L_AMBIENT = 1, L_DIRECTIONAL = 2, L_OBJECT = 3 };
enum : uint32_t { LS_SPHERE = 0, LS_CUBE = 1, LS_GROUP = 2, LS_MESH = 3 };
struct alignas(16) Light {
    uint32_t kind;        // L_*
    uint32_t shape;       // LS_* for L_OBJECT
    int32_t twin_object;  // scene object identical to this light's object, or -1 (never visible)
    uint32_t first;       // LS_MESH: first LightTri;  LS_GROUP: first child LightPart
    uint32_t count;       // LS_MESH: triangle count;  LS_GROUP: number of children
    uint32_t xf;          // index into lxf (every object light has one)
    uint32_t twin_lo, twin_hi;  // hit codes of the twin object's primitives when they form one range (lo <= hi)
    F4 color;             // Ambient: colour;  Object: material.color() * material.emittance()
    F4 albedo;            // Object: material.color() (photon power, src/photon.rs:757)
};
// A Light::Object whose shape is a KdTree<Box<dyn Bounded>>: sampling picks a child uniformly (then a
// child of that child, ...: src/kdtree.rs:141-146), so the hierarchy is kept as parts.  A part is a group
// (first/count index `lparts`) or a leaf shape with the transform composed down from the root.
struct alignas(16) LightPart {
    uint32_t shape;  // LS_*
    uint32_t first;  // LS_GROUP: first child part;  LS_MESH: first LightTri
    uint32_t count;  // LS_GROUP: children;          LS_MESH: triangles
    uint32_t xf;     // leaf: index into lxf
};
// Light-mesh triangle: world-space vertices, LOCAL vertex normals and 1/(local area): the
// pdf / normal mapping of Transformed::sample (src/shape.rs:140-151) is applied per sample.
struct alignas(16) LightTri {
    F4 v1, v2, v3;  // v1.w = 1/area_local
    F4 n1, n2, n3;
};
// Sphere / cube light: forward + inverse affine, normal transform, linear part, det.
struct alignas(16) LightXf {
    F4 fwd[3];   // local -> world
    F4 inv[3];   // world -> local
    F4 nrm[3];   // M^-T rows; nrm[0].w = det(linear); nrm[1].w = has_transform
    F4 lin[3];   // linear rows
};
// One instance of a mesh that several shapes share (Arc<Mesh> under different transforms,
// examples/fractal_teapots.rs:17-22): the mesh's triangles and tree are stored once in LOCAL
// space; the ray is mapped into it exactly as Transformed::intersect does (src/shape.rs:129-138).
struct alignas(16) InstRec {
    F4 r0, r1, r2;   // rows of M^-1 (world -> local)
    F4 n0, n1, n2;   // rows of M^-T (normals local -> world); n0.w = object (bits), n1.w = mesh root node (bits)
};
struct alignas(16) MeshRef {   // one BVH-accelerated mesh object
    uint32_t root, tri_base, tri_count, object;   // root: absolute index of the mesh's root node
};

struct SceneView {
    const XfScan* sph;     const XfShade* sph_sh;   uint32_t n_sph;
    const XfScan* cub;     const XfShade* cub_sh;   uint32_t n_cub;
    const PlaneScan* pln;  const PlaneShade* pln_sh; uint32_t n_pln;
    const TriScan* tri;    const TriShade* tri_sh;  uint32_t n_tri;
    const AabbScan* aabb;  uint32_t n_aabb;
    const RectScan* rect;  const RectShade* rect_sh; uint32_t n_rect_x, n_rect_y, n_rect_z;  // sorted by axis
    const ShellScan* shell; uint32_t has_shell;  // rectangles folded into the shell follow the scanned ones in rect_sh
    // World boxes (lo, hi; padded) of the scanned bounded records in scan order -- spheres, cubes, boxes, rectangles,
    // triangles -- for queries that are known to stay inside a small ball (photon-gather visibility rays): records whose
    // box misses the ball are skipped (scan_prims<true>).  Planes and the shell have no entry: always tested.
    const AabbScan* pbox;
    const BvhNode* nodes;  const TriScan* btri;     const TriShade* btri_sh;
    const MeshRef* meshes; uint32_t n_mesh;
    // Scene-level BVH over every bounded primitive and mesh root (built when the scene has many
    // of them, e.g. KdTree<Box<dyn Bounded>> groups): scene_bvh = 1 replaces the linear scans of
    // sph/cub/aabb/rect/tri and the per-mesh walks by one walk from `top_root`.
    const uint32_t* pleaf; uint32_t n_nodes, scene_bvh, top_root;
    // scene_bvh with mesh_deferred = 1: the meshes with trees of their own (`meshes`) are NOT leaves of the scene tree -- a
    // query walks the scene tree for everything else and the mesh trees separately, which lets the render kernel park those
    // walks (the long ones) as it does in scenes without a scene tree.
    uint32_t mesh_deferred;
    const InstRec* inst;   uint32_t n_inst;
    const Material* mats;  uint32_t n_obj;   // one material record per scene object
    const Light* lights;   uint32_t n_lights;
    const LightTri* ltris; const LightXf* lxf; const LightPart* lparts; uint32_t n_lparts;  // n_lparts != 0: some Light::Object is a group
    uint32_t n_ltris;
    // medium (media[0]); has_medium = 0: surface-only branch
    uint32_t has_medium, medium_kind;
    float sigma_a, sigma_s;
    float medium_color[3];      // homogeneous_isotropic colour, or the y <= 250 colour
    float medium_color_hi[3];   // colored_glowing_fog colour for y > 250
    float medium_emission, medium_phase;
    float env[3];
    // Environment::Hdri (src/environment.rs:3-52): hdri_w = 0 means Environment::Color(env)
    const F4* hdri;
    uint32_t hdri_w, hdri_h;
    // Counters builds only: bumped when a tree walk finds its stack full (bvh_traverse then drops the far child -- the commit-
    // time depth check is what rules that out; this is where a violation would show).  rpt_get_counters()[7].
    unsigned long long* stack_overflows;
};

struct CameraG {
    float eye[3], ddir[3], right[3], up[3];  // ddir = cot(fov/2) * direction
    float aperture, focal_distance;
};

}  // namespace rptg
