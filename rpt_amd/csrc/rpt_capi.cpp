// rpt_capi.cpp — host side of the C ABI (include/rpt_hip.h): fp64 scene store, flattening to
// the fp32 device layout (gpu_layout.h), BVH build for large meshes, launches.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rpt_hip.h"
#include "host_internal.h"
#include "kernels.h"
#include "f64_layout.h"

using namespace rptg;

// ---------------------------------------------------------------------------- errors / options
static thread_local std::string g_err;
namespace rpti {
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace rpti
using rpti::fail;
namespace rpti {
uint64_t seed_mix(uint64_t seed) {
    uint64_t x = seed + 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
}  // namespace rpti
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(RPT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));       \
    } while (0)

// Options.  Every scene carries its own set, copied from the process defaults when it is created
// (rpt_scene_create) and changed with rpt_scene_set_option; rpt_set_option changes the defaults, i.e. the scenes
// created afterwards.  Nothing a render or commit reads is process-global, so scenes with different options can
// be driven from different host threads.
struct rpt_options {
    int64_t counters = 0;
    int64_t chunk_spp = 0;          // 0 = auto: ceil(iterations / 16) clamped to [2, 32]
    int64_t blocks_per_cu = 0;      // 0 = occupancy query
    int64_t timing = 0;
    int64_t room_shell = 1;         // fold rectangles that are the faces of one box into a single slab test
    int64_t photon_skip = 0;
    int64_t photon_block_lists = 1;
    int64_t photon_coop_gather = 1; // surface gather of a pixel's samples by the wave together (0: one search per lane)
    int64_t photon_split = 0;       // camera pass of the beam kinds in a medium: volume estimate and surface estimate as two launches (no scratch in
                                    // either, but 112 ms instead of 95 on C4: in one kernel the LDS-bound and the VALU-bound estimate overlap)
    int64_t photon_parts = 4;       // work items per (8x8 pixel block, sample chunk) of the photon camera pass: the block's rows in strips
    int64_t instancing = 1;         // meshes shared by several shapes are stored once and instanced
    int64_t bvh_leaf_max = 4;       // triangles per leaf of a mesh tree (read by rpt_scene_commit)
    int64_t bvh_max_depth = 20;     // a mesh tree deeper than this is rebuilt balanced (read by rpt_scene_commit)
    int64_t bvh_sweep_below = 4096; // ranges of at most this many triangles are split by an exact SAH sweep instead of 16 bins (read by rpt_scene_commit; 0 = bins only)
    int64_t defer_stop = 16;        // still-walking lanes below which a wave leaves the walk (the rest resume later)
    int64_t walk_leaf_quarters = 6; // deferred walks: test the leaves when 4 x (lanes at a leaf) >= this x (lanes still descending); 0 = when all are there
    int64_t defer_lanes = 32;       // parked tree walks per wave that trigger a walk (per-mesh-tree kernels)
    int64_t detach_shadows = 1;     // per-mesh-tree kernels in a medium: 1 = shadow queries that need a tree walk leave their path (0: they park
                                    // the lane), 2 = every tree walk leaves its path ("streamed walks": ring + parked paths in memory)
    int64_t stream_backlog = 48;    // detach_shadows = 2: queries in a wave's ring that trigger a walk session
    int64_t stream_contexts = 1;    // detach_shadows = 2: paths a lane can have waiting in memory (1..6; C5: 272 / 279 / 290 ms for 1 / 2 / 4 at
                                    // 2048x2048x128 -- every parked path is a round trip to the fabric --, 234 with detach_shadows = 1)
    int64_t detach_lanes = 44;      // ... parked primary + queued shadow queries per wave that trigger a walk session
    int64_t pull_batch = 2;         // lanes waiting for a new work item before a wave runs its item bookkeeping (1..64)
    int64_t detach_trigger = 28;    // ... or this many queued shadow queries alone (the queue holds 32)
    int64_t scene_bvh_min = 64;     // bounded primitives + BVH meshes from which the scene-level BVH is built
    int64_t scene_tree_meshes = 0;  // 1: meshes with trees of their own are leaves of the scene tree (every query walks to completion);
                                    // 0: they stay outside it and their walks are parked as in scenes without a scene tree (read by rpt_scene_commit)
    int64_t f64_cull = 1;           // reference-epsilon mode: 1 = a lane evaluates only the objects whose fp32 box its ray can reach (same bits), 0 = every object;
                                    // 2 = as 1, and the counters build keeps the search limits as well (its counters then describe the schedule, not the reference's work)
    int64_t f64_photon_slice = 0;   // reference-epsilon photon camera pass: samples per slice (0: as many whole chunks of 256 as fit 32 GB of per-sample selections)
    int64_t f64_surf_batch = 8;     // reference-epsilon mode, scenes with a medium: lanes of a wave that wait at a surface event before the wave runs the surface code (1..64)
    int64_t epsilon_policy = 0;     // 1: the reference-epsilon mode (read by rpt_scene_commit): fp64, generic shapes, t_min = 1e-12, |hit - dist| < 1e-12
};
static rpt_options g_defaults;
static std::mutex g_defaults_mutex;
static int set_option_in(rpt_options& o, const char* name, int64_t value) {
    if (!name) return fail(RPT_ERR_INVALID, "null option name");
    const std::string s(name);
    if (s == "counters") o.counters = value;
    else if (s == "chunk_spp") { if (value < 0) return fail(RPT_ERR_INVALID, "chunk_spp must be >= 0 (0 = auto)"); o.chunk_spp = value; }
    else if (s == "blocks_per_cu") o.blocks_per_cu = value;
    else if (s == "timing") o.timing = value;
    else if (s == "room_shell") o.room_shell = value;
    else if (s == "photon_skip") o.photon_skip = value;
    else if (s == "photon_block_lists") o.photon_block_lists = value;
    else if (s == "photon_parts") o.photon_parts = value;
    else if (s == "photon_split") {
#ifndef RPT_EXPERIMENTS
        if (value != 0) return fail(RPT_ERR_UNSUPPORTED, "photon_split: a rejected prototype, built with -DRPT_EXPERIMENTS only");
#endif
        o.photon_split = value;
    }
    else if (s == "photon_coop_gather") o.photon_coop_gather = value;
    else if (s == "instancing") o.instancing = value;
    else if (s == "defer_lanes") { if (value < 1 || value > 64) return fail(RPT_ERR_INVALID, "defer_lanes must be 1..64"); o.defer_lanes = value; }
    else if (s == "bvh_leaf_max") { if (value < 1 || value > 16) return fail(RPT_ERR_INVALID, "bvh_leaf_max must be 1..16"); o.bvh_leaf_max = value; }
    else if (s == "bvh_sweep_below") { if (value < 0) return fail(RPT_ERR_INVALID, "bvh_sweep_below must be >= 0"); o.bvh_sweep_below = value; }
    else if (s == "bvh_max_depth") { if (value < 1 || value > 20) return fail(RPT_ERR_INVALID, "bvh_max_depth must be 1..20"); o.bvh_max_depth = value; }
    else if (s == "walk_leaf_quarters") { if (value < 0 || value > 256) return fail(RPT_ERR_INVALID, "walk_leaf_quarters must be 0..256"); o.walk_leaf_quarters = value; }
    else if (s == "detach_shadows") {
        if (value < 0 || value > 2) return fail(RPT_ERR_INVALID, "detach_shadows must be 0, 1 or 2");
#ifndef RPT_EXPERIMENTS
        if (value == 2) return fail(RPT_ERR_UNSUPPORTED, "detach_shadows = 2 (streamed walks): a rejected prototype, built with -DRPT_EXPERIMENTS only");
#endif
        o.detach_shadows = value;
    }
    else if (s == "stream_contexts") { if (value < 1 || value > 6) return fail(RPT_ERR_INVALID, "stream_contexts must be 1..6"); o.stream_contexts = value; }
    else if (s == "stream_backlog") { if (value < 1 || value > 256) return fail(RPT_ERR_INVALID, "stream_backlog must be 1..256"); o.stream_backlog = value; }
    else if (s == "detach_lanes") { if (value < 1 || value > 96) return fail(RPT_ERR_INVALID, "detach_lanes must be 1..96"); o.detach_lanes = value; }
    else if (s == "pull_batch") { if (value < 1 || value > 64) return fail(RPT_ERR_INVALID, "pull_batch must be 1..64"); o.pull_batch = value; }
    else if (s == "detach_trigger") { if (value < 1 || value > 32) return fail(RPT_ERR_INVALID, "detach_trigger must be 1..32"); o.detach_trigger = value; }
    else if (s == "defer_stop") { if (value < 1 || value > 64) return fail(RPT_ERR_INVALID, "defer_stop must be 1..64"); o.defer_stop = value; }
    else if (s == "f64_cull") { if (value < 0 || value > 2) return fail(RPT_ERR_INVALID, "f64_cull must be 0, 1 or 2"); o.f64_cull = value; }
    else if (s == "f64_photon_slice") { if (value < 0 || value > (1 << 20)) return fail(RPT_ERR_INVALID, "f64_photon_slice must be 0..2^20"); o.f64_photon_slice = value; }
    else if (s == "f64_surf_batch") { if (value < 1 || value > 64) return fail(RPT_ERR_INVALID, "f64_surf_batch must be 1..64"); o.f64_surf_batch = value; }
    else if (s == "epsilon_policy") { if (value < 0 || value > 1) return fail(RPT_ERR_INVALID, "epsilon_policy must be 0 or 1"); o.epsilon_policy = value; }
    else if (s == "scene_tree_meshes") o.scene_tree_meshes = value;
    else if (s == "scene_bvh_min") { if (value < 0) return fail(RPT_ERR_INVALID, "scene_bvh_min must be >= 0"); o.scene_bvh_min = value; }
    else return fail(RPT_ERR_INVALID, "unknown option: " + s);
    return RPT_OK;
}

// ---------------------------------------------------------------------------- fp64 helpers
namespace {
struct D3 {
    double x, y, z;
};
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline D3 normalize(D3 a) {
    double l = std::sqrt(dot(a, a));
    return {a.x / l, a.y / l, a.z / l};
}
inline D3 d3(const double* p) { return {p[0], p[1], p[2]}; }
// component by index.  (Not (&v.x)[a]: indexing past the member x is undefined, and clang does drop the
// y / z iterations of such a loop when it is not unrolled.)
inline double comp(const D3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline F4 f4(D3 v, double w) { return F4{float(v.x), float(v.y), float(v.z), float(w)}; }
inline float bits_f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t bits_u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// Derived matrices of Transformed::new (src/shape.rs:112-125), fp64.
struct Xf {
    bool has = false;
    double M[4][4], Minv[4][4], L[3][3], N[3][3], det = 1.0;
    D3 point(D3 p) const { return {M[0][0] * p.x + M[0][1] * p.y + M[0][2] * p.z + M[0][3], M[1][0] * p.x + M[1][1] * p.y + M[1][2] * p.z + M[1][3], M[2][0] * p.x + M[2][1] * p.y + M[2][2] * p.z + M[2][3]}; }
    D3 normal(D3 n) const { return {N[0][0] * n.x + N[0][1] * n.y + N[0][2] * n.z, N[1][0] * n.x + N[1][1] * n.y + N[1][2] * n.z, N[2][0] * n.x + N[2][1] * n.y + N[2][2] * n.z}; }
};
bool invert4(const double a[4][4], double out[4][4]) {
    double w[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            w[i][j] = a[i][j];
            w[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (w[p][c] == 0.0) return false;
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
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i][j] = w[i][4 + j];
    return true;
}
bool finish_xf(Xf& x);
bool make_xf(const rpt_shape_desc& d, Xf& x) {
    x.has = d.has_transform != 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) x.M[i][j] = x.has ? d.transform[i * 4 + j] : (i == j ? 1.0 : 0.0);
    return finish_xf(x);
}
// Transformed<KdTree<..Transformed<T>..>>: the child's matrix under the group's (world = P * C * local).
bool compose_xf(const Xf& parent, const rpt_shape_desc& d, Xf& x) {
    Xf c;
    if (!make_xf(d, c)) return false;
    x.has = parent.has || c.has;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double a = 0.0;
            for (int k = 0; k < 4; k++) a += parent.M[i][k] * c.M[k][j];
            x.M[i][j] = (parent.has && c.has) ? a : (parent.has ? parent.M[i][j] : c.M[i][j]);
        }
    return finish_xf(x);
}
bool finish_xf(Xf& x) {
    if (!invert4(x.M, x.Minv)) return false;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) x.L[i][j] = x.M[i][j];
    const auto& a = x.L;
    x.det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
            a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (x.det == 0.0 || !(x.det == x.det)) return false;
    // inverse transpose of the linear part
    double inv[3][3];
    inv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / x.det;
    inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / x.det;
    inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / x.det;
    inv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / x.det;
    inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / x.det;
    inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / x.det;
    inv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / x.det;
    inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / x.det;
    inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / x.det;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) x.N[i][j] = inv[j][i];
    return true;
}
D3 hex_color(uint32_t v) {  // src/color.rs:10-15
    auto ch = [](uint32_t c) { return std::pow(double(c) / 255.0, 2.2); };
    return {ch((v >> 16) & 0xff), ch((v >> 8) & 0xff), ch(v & 0xff)};
}
uint64_t mix64_(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
uint64_t seed_mix(uint64_t seed) { return rpti::seed_mix(seed); }

// ---------------------------------------------------------------------------- host scene
struct HShape {
    rpt_shape_desc d;            // tris / children pointers rewritten to own storage
    std::shared_ptr<const std::vector<double>> mesh;  // interned per scene: shapes sharing a mesh share this
    const std::vector<double>& T() const {
        static const std::vector<double> none;
        return mesh ? *mesh : none;
    }
    std::vector<HShape> children;  // RPT_SHAPE_GROUP
};
struct HObject {
    HShape shape;
    rpt_material mat;
};
struct HLight {
    int kind;
    double color[3], vec[3];
    HObject obj;
};
struct HMedium {
    int kind;
    double absorption, scattering;
};
bool same_shape(const HShape& a, const HShape& b) {
    if (a.d.kind != b.d.kind || a.d.has_transform != b.d.has_transform) return false;
    if (a.d.has_transform && std::memcmp(a.d.transform, b.d.transform, sizeof(a.d.transform)) != 0) return false;
    if (a.d.kind == RPT_SHAPE_PLANE)
        return std::memcmp(a.d.plane_normal, b.d.plane_normal, 24) == 0 && a.d.plane_value == b.d.plane_value;
    if (a.d.kind == RPT_SHAPE_MESH)
        return a.mesh == b.mesh || (a.T().size() == b.T().size() &&
                                    (a.T().empty() || std::memcmp(a.T().data(), b.T().data(), a.T().size() * 8) == 0));
    if (a.d.kind == RPT_SHAPE_GROUP) {
        if (a.children.size() != b.children.size()) return false;
        for (size_t i = 0; i < a.children.size(); i++)
            if (!same_shape(a.children[i], b.children[i])) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------- BVH build
struct BTri {
    float lo[3], hi[3], c[3];
    uint32_t idx;
};
struct TmpNode {  // build-time node: own box, children adjacent (left, left+1)
    float lo[3];
    uint32_t left_or_first;
    float hi[3];
    uint32_t count;  // 0 = inner
};
struct BvhBuilder {
    std::vector<BTri>& t;
    std::vector<TmpNode>& nodes;
    static constexpr int kBins = 16, kMaxDepth = 28;
    uint32_t leaf_max = 4;
    const std::vector<uint8_t>* solo = nullptr;  // by BTri::idx: items that must be alone in their leaf
    // The traversal keeps at most one stack entry per level and the stack has 32 (device_core.h): a tree that
    // comes out deeper than the caller can afford is rebuilt with `balanced` = object-median splits along the widest
    // centroid axis, whose depth is ceil(log2(n)) whatever the input looks like.
    bool balanced = false;
    int max_depth = 0;
    uint32_t sweep_below = 0;   // ranges of at most this many items: exact SAH sweep over the sorted centroids of each axis
    std::vector<float> sweep_area;
    // Exact SAH split of [first, first + count): the range is left sorted along the best axis, `mid` is the split.
    bool sweep_split(uint32_t first, uint32_t count, uint32_t& mid, float& cost_out) {
        float best = std::numeric_limits<float>::infinity();
        int best_axis = -1;
        uint32_t best_k = 0;
        sweep_area.resize(count);
        for (int a = 0; a < 3; a++) {
            std::sort(t.begin() + first, t.begin() + first + count, [a](const BTri& p, const BTri& q) { return p.c[a] < q.c[a] || (p.c[a] == q.c[a] && p.idx < q.idx); });
            float l0[3], h0[3];
            for (int k = 0; k < 3; k++) { l0[k] = std::numeric_limits<float>::infinity(); h0[k] = -l0[k]; }
            for (uint32_t i = count; i-- > 1;) {   // sweep_area[i] = area of items [i, count)
                for (int k = 0; k < 3; k++) { l0[k] = std::min(l0[k], t[first + i].lo[k]); h0[k] = std::max(h0[k], t[first + i].hi[k]); }
                sweep_area[i] = area(l0, h0);
            }
            for (int k = 0; k < 3; k++) { l0[k] = std::numeric_limits<float>::infinity(); h0[k] = -l0[k]; }
            for (uint32_t i = 1; i < count; i++) {   // left = [0, i), right = [i, count)
                for (int k = 0; k < 3; k++) { l0[k] = std::min(l0[k], t[first + i - 1].lo[k]); h0[k] = std::max(h0[k], t[first + i - 1].hi[k]); }
                const float cost = area(l0, h0) * float(i) + sweep_area[i] * float(count - i);
                if (cost < best) { best = cost; best_axis = a; best_k = i; }
            }
        }
        if (best_axis < 0) return false;
        if (best_axis != 2)
            std::sort(t.begin() + first, t.begin() + first + count, [best_axis](const BTri& p, const BTri& q) { return p.c[best_axis] < q.c[best_axis] || (p.c[best_axis] == q.c[best_axis] && p.idx < q.idx); });
        mid = first + best_k;
        cost_out = best;
        return true;
    }
    bool can_leaf(uint32_t first, uint32_t count) const {
        if (!solo || count == 1) return true;
        for (uint32_t i = first; i < first + count; i++)
            if ((*solo)[t[i].idx]) return false;
        return true;
    }
    void bounds(uint32_t first, uint32_t count, float lo[3], float hi[3], float clo[3], float chi[3]) {
        for (int a = 0; a < 3; a++) {
            lo[a] = clo[a] = std::numeric_limits<float>::infinity();
            hi[a] = chi[a] = -std::numeric_limits<float>::infinity();
        }
        for (uint32_t i = first; i < first + count; i++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::min(lo[a], t[i].lo[a]);
                hi[a] = std::max(hi[a], t[i].hi[a]);
                clo[a] = std::min(clo[a], t[i].c[a]);
                chi[a] = std::max(chi[a], t[i].c[a]);
            }
    }
    static float area(const float lo[3], const float hi[3]) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    void build(uint32_t node, uint32_t first, uint32_t count, int depth) {
        float lo[3], hi[3], clo[3], chi[3];
        bounds(first, count, lo, hi, clo, chi);
        max_depth = std::max(max_depth, depth);
        TmpNode& n = nodes[node];
        for (int a = 0; a < 3; a++) {  // conservative padding for the fp32 slab test
            float pad = 1e-6f * std::max(std::fabs(lo[a]), std::fabs(hi[a])) + 1e-30f;
            n.lo[a] = lo[a] - pad;
            n.hi[a] = hi[a] + pad;
        }
        auto make_leaf = [&]() {
            nodes[node].left_or_first = first;
            nodes[node].count = count;
        };
        const bool leaf_ok = can_leaf(first, count);
        if (leaf_ok && (count <= leaf_max || (depth >= kMaxDepth && count <= 32))) return make_leaf();
        int best_axis = -1, best_bin = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        uint32_t sweep_mid = 0;
        const bool swept = !balanced && count <= sweep_below && depth < kMaxDepth && sweep_split(first, count, sweep_mid, best_cost);
        for (int a = 0; a < 3 && !balanced && !swept; a++) {
            float ext = chi[a] - clo[a];
            if (!(ext > 0.f)) continue;
            uint32_t cnt[kBins] = {0};
            float blo[kBins][3], bhi[kBins][3];
            for (int b = 0; b < kBins; b++)
                for (int k = 0; k < 3; k++) {
                    blo[b][k] = std::numeric_limits<float>::infinity();
                    bhi[b][k] = -std::numeric_limits<float>::infinity();
                }
            float scale = float(kBins) / ext;
            for (uint32_t i = first; i < first + count; i++) {
                int b = std::min(kBins - 1, int((t[i].c[a] - clo[a]) * scale));
                cnt[b]++;
                for (int k = 0; k < 3; k++) {
                    blo[b][k] = std::min(blo[b][k], t[i].lo[k]);
                    bhi[b][k] = std::max(bhi[b][k], t[i].hi[k]);
                }
            }
            float la[kBins], ra[kBins];
            uint32_t lc[kBins], rc[kBins];
            float l0[3], h0[3];
            for (int k = 0; k < 3; k++) { l0[k] = std::numeric_limits<float>::infinity(); h0[k] = -l0[k]; }
            uint32_t c = 0;
            for (int b = 0; b < kBins; b++) {
                c += cnt[b];
                for (int k = 0; k < 3; k++) { l0[k] = std::min(l0[k], blo[b][k]); h0[k] = std::max(h0[k], bhi[b][k]); }
                lc[b] = c;
                la[b] = c ? area(l0, h0) : 0.f;
            }
            for (int k = 0; k < 3; k++) { l0[k] = std::numeric_limits<float>::infinity(); h0[k] = -l0[k]; }
            c = 0;
            for (int b = kBins - 1; b >= 0; b--) {
                c += cnt[b];
                for (int k = 0; k < 3; k++) { l0[k] = std::min(l0[k], blo[b][k]); h0[k] = std::max(h0[k], bhi[b][k]); }
                rc[b] = c;
                ra[b] = c ? area(l0, h0) : 0.f;
            }
            for (int b = 0; b < kBins - 1; b++) {
                if (lc[b] == 0 || rc[b + 1] == 0) continue;
                float cost = la[b] * float(lc[b]) + ra[b + 1] * float(rc[b + 1]);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        uint32_t mid;
        if (swept) {
            float parent_cost = area(lo, hi) * float(count);
            if (best_cost >= parent_cost && count <= 8 && leaf_ok) return make_leaf();
            mid = sweep_mid;
        } else if (balanced) {
            int axis = 0;
            for (int a = 1; a < 3; a++)
                if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
            mid = first + count / 2;
            std::nth_element(t.begin() + first, t.begin() + mid, t.begin() + first + count,
                             [axis](const BTri& p, const BTri& q) { return p.c[axis] < q.c[axis]; });
        } else if (best_axis < 0 || depth >= kMaxDepth) {
            if (count <= 16 && leaf_ok) return make_leaf();
            mid = first + count / 2;  // all centroids coincide (or depth cap): split by index
        } else {
            float parent_cost = area(lo, hi) * float(count);
            if (best_cost >= parent_cost && count <= 8 && leaf_ok) return make_leaf();
            float ext = chi[best_axis] - clo[best_axis];
            float scale = float(kBins) / ext;
            auto it = std::partition(t.begin() + first, t.begin() + first + count, [&](const BTri& x) {
                int b = std::min(kBins - 1, int((x.c[best_axis] - clo[best_axis]) * scale));
                return b <= best_bin;
            });
            mid = uint32_t(it - t.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        uint32_t left = uint32_t(nodes.size());
        nodes.push_back(TmpNode{});
        nodes.push_back(TmpNode{});
        nodes[node].left_or_first = left;
        nodes[node].count = 0;
        build(left, first, mid - first, depth + 1);
        build(left + 1, mid, first + count - mid, depth + 1);
    }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
};

}  // namespace

// ---------------------------------------------------------------------------- rpt_scene
struct rpt_scene {
    std::vector<HObject> objects;
    std::vector<HLight> lights;
    std::vector<HMedium> media;
    double env[3] = {0, 0, 0};
    uint32_t hdri_w = 0, hdri_h = 0;
    std::vector<float> hdri;  // w*h*4
    std::vector<double> hdri64;  // w*h*3: the texels as they were given (the reference-epsilon mode reads these)
    bool committed = false;
    rpt_options opt;  // this scene's options (rpt_scene_set_option; starts as a copy of the process defaults)
    int device = 0;
    int n_cus = 256;
    // device memory
    void* arena = nullptr;
    SceneView view{};
    // per-render cached buffers
    uint32_t* d_tiles = nullptr;
    size_t tiles_cap = 0;
    // What one launch owns until its resolve has run: the slab of partial sums and the work counter.  Two sets, so
    // that a caller who alternates between two streams (consecutive frames of an iterative render) gets launches
    // that overlap -- the next frame's blocks fill the CUs that the last paths of this frame no longer keep busy,
    // ~0.35 ms per launch -- while launches on one stream keep using one set.
    struct LaunchSet {
        float* d_slab = nullptr;
        size_t slab_cap = 0;  // bytes
        unsigned long long* d_queue = nullptr;
        uint32_t* d_stream = nullptr;   // streamed walks (detach = 2): rings and parked paths of the grid's waves
        size_t stream_cap = 0;          // bytes
        hipEvent_t done = nullptr;  // recorded after the resolve of the last launch that used the set
        hipEvent_t launched = nullptr;  // recorded right before its render kernel
        hipStream_t stream = nullptr;
        bool used = false;
    };
    LaunchSet sets[2];
    int cur_set = 0;
    unsigned long long* d_counters = nullptr;
    double* d_out = nullptr;
    size_t out_cap = 0;  // bytes
    uint64_t last_counters[64] = {0};  // [0..7] counters, [8..63] diagnostic trip stamps
    // "timing": an event triple (before render, after render, after resolve) per launch, in a ring; nothing waits
    // for them until rpt_get_timing / rpt_get_timing_mean is called
    static constexpr size_t kTimedLaunches = 1024;
    std::vector<hipEvent_t> evs;
    size_t ev_count = 0;  // timed launches since the last rpt_get_timing_mean
    int last_blocks = 0;
    uint64_t prims_per_ray = 0;
    uint32_t n_twin_lights = 0;  // Light::Objects with a twin among the scene's objects (the ones that can be visible)
    bool twins_scanned = false;  // every Light::Object that can be visible has its twin among the scanned records, as one range of hit codes
    uint64_t stats[16] = {0};
    void* photon = nullptr;  // PhotonMapDev*, owned by photon.hip
    // reference-epsilon mode (option "epsilon_policy" = 1 at commit): the fp64 scene of kernels_f64.hip
    void* arena64 = nullptr;
    size_t arena64_bytes = 0;
    rpt64::Scene view64{};
    double medium_color64[3] = {0, 0, 0}, medium_color_hi64[3] = {0, 0, 0};
    uint64_t last_counters64[64] = {0};   // [0..11] rpt_debug_epsilon_counters, [16 + 2k], [17 + 2k] section k of kernels_f64.hip (executions, lanes)
    // mesh data interned by content (hash -> candidates), so Arc<Mesh>-style sharing survives the C ABI
    std::unordered_map<uint64_t, std::vector<std::shared_ptr<const std::vector<double>>>> mesh_pool;
    // tile cache key
    uint32_t tk_w = 0, tk_h = 0, tk_rank = 0, tk_count = 0, n_tiles = 0, tiles_x = 0;
};

using MeshCallCache = std::unordered_map<const double*, std::pair<uint64_t, std::shared_ptr<const std::vector<double>>>>;
static std::shared_ptr<const std::vector<double>> intern_mesh(rpt_scene* s, const double* p, uint64_t n_tris,
                                                              MeshCallCache& seen) {
    auto it = seen.find(p);  // caller memory cannot change during one add call: same pointer, same data
    if (it != seen.end() && it->second.first == n_tris) return it->second.second;
    const size_t n = size_t(n_tris) * 18;
    uint64_t h = 0xCBF29CE484222325ULL ^ n;
    for (size_t i = 0; i < n; i++) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001B3ULL;
        h ^= h >> 29;
    }
    auto& bucket = s->mesh_pool[h];
    for (auto& c : bucket)
        if (c->size() == n && std::memcmp(c->data(), p, n * 8) == 0) { seen[p] = {n_tris, c}; return c; }
    auto m = std::make_shared<const std::vector<double>>(p, p + n);
    bucket.push_back(m);
    seen[p] = {n_tris, m};
    return m;
}
static bool copy_shape(rpt_scene* s, MeshCallCache& seen, const rpt_shape_desc* d, HShape& out, std::string& why,
                       int depth = 0) {
    if (!d) { why = "null shape"; return false; }
    if (d->kind < 0 || d->kind > RPT_SHAPE_GROUP) { why = "unknown shape kind"; return false; }
    out.d = *d;
    out.mesh.reset();
    out.children.clear();
    if (d->kind == RPT_SHAPE_MESH) {
        if (d->n_tris == 0 || !d->tris) { why = "mesh without triangles"; return false; }
        out.mesh = intern_mesh(s, d->tris, d->n_tris, seen);
    }
    if (d->kind == RPT_SHAPE_GROUP) {
        if (d->n_children == 0 || !d->children) { why = "group without children"; return false; }
        if (depth >= 16) { why = "groups nested too deeply"; return false; }
        out.children.resize(d->n_children);
        for (uint64_t i = 0; i < d->n_children; i++) {
            if (d->children[i].kind == RPT_SHAPE_PLANE) { why = "Plane is not Bounded and cannot be a KdTree child (src/kdtree.rs:12)"; return false; }
            if (!copy_shape(s, seen, d->children + i, out.children[i], why, depth + 1)) return false;
        }
    }
    out.d.tris = nullptr;
    out.d.children = nullptr;
    Xf x;
    if (!make_xf(*d, x)) { why = "singular transform"; return false; }
    return true;
}
static bool check_material(const rpt_material* m, std::string& why) {
    if (!m) { why = "null material"; return false; }
    if (m->kind < 0 || m->kind > RPT_MAT_TRANSMISSIVE) { why = "unknown material kind"; return false; }
    return true;
}

namespace {
struct TmpDev {
    std::vector<void*> ptrs;
    ~TmpDev() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T** p, size_t n) {
        hipError_t e = hipMalloc((void**)p, std::max<size_t>(n * sizeof(T), 16));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};
}  // namespace

extern "C" {

const char* rpt_last_error(void) { return g_err.c_str(); }

int rpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rpt_set_option(const char* name, int64_t value) {
    std::lock_guard<std::mutex> lock(g_defaults_mutex);
    return set_option_in(g_defaults, name, value);
}
int rpt_scene_set_option(rpt_scene* s, const char* name, int64_t value) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    return set_option_in(s->opt, name, value);
}

rpt_scene* rpt_scene_create(void) {
    rpt_scene* s = new rpt_scene();
    std::lock_guard<std::mutex> lock(g_defaults_mutex);
    s->opt = g_defaults;
    return s;
}

// Everything a commit (and the renders after it) put on the device; the pointers are cleared, so this is safe to repeat and
// a commit that fails half-way (the reference-epsilon scene after the fp32 upload) leaves nothing behind.
static void release_device(rpt_scene* s) {
    if (!s->committed && !s->arena && !s->arena64) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->arena); s->arena = nullptr;
    (void)hipFree(s->arena64); s->arena64 = nullptr; s->arena64_bytes = 0;
    (void)hipFree(s->d_tiles); s->d_tiles = nullptr; s->tiles_cap = 0;
    for (auto& ls : s->sets) {
        (void)hipFree(ls.d_slab); ls.d_slab = nullptr; ls.slab_cap = 0;
        (void)hipFree(ls.d_queue); ls.d_queue = nullptr;
        (void)hipFree(ls.d_stream); ls.d_stream = nullptr; ls.stream_cap = 0;
        if (ls.done) (void)hipEventDestroy(ls.done);
        if (ls.launched) (void)hipEventDestroy(ls.launched);
        ls.done = nullptr; ls.launched = nullptr; ls.used = false;
    }
    (void)hipFree(s->d_counters); s->d_counters = nullptr;
    (void)hipFree(s->d_out); s->d_out = nullptr; s->out_cap = 0;
    for (auto& e : s->evs)
        if (e) (void)hipEventDestroy(e);
    s->evs.clear();
    if (s->photon) rpti::photon_release(s->photon);
    s->photon = nullptr;
    s->committed = false;
}

void rpt_scene_destroy(rpt_scene* s) {
    if (!s) return;
    release_device(s);
    delete s;
}

int rpt_scene_add_object(rpt_scene* s, const rpt_shape_desc* d, const rpt_material* m) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (s->committed) return fail(RPT_ERR_STATE, "scene is immutable after rpt_scene_commit");
    std::string why;
    HObject o;
    MeshCallCache seen;
    if (!copy_shape(s, seen, d, o.shape, why) || !check_material(m, why)) return fail(RPT_ERR_INVALID, why);
    o.mat = *m;
    s->objects.push_back(std::move(o));
    return int(s->objects.size()) - 1;
}
static int add_simple_light(rpt_scene* s, int kind, const double* color, const double* vec) {
    if (!s || !color) return fail(RPT_ERR_INVALID, "null argument");
    if (s->committed) return fail(RPT_ERR_STATE, "scene is immutable after rpt_scene_commit");
    HLight l{};
    l.kind = kind;
    std::memcpy(l.color, color, 24);
    if (vec) std::memcpy(l.vec, vec, 24);
    s->lights.push_back(std::move(l));
    return RPT_OK;
}
int rpt_scene_add_light_point(rpt_scene* s, const double color[3], const double location[3]) {
    if (!location) return fail(RPT_ERR_INVALID, "null location");
    return add_simple_light(s, L_POINT, color, location);
}
int rpt_scene_add_light_ambient(rpt_scene* s, const double color[3]) { return add_simple_light(s, L_AMBIENT, color, nullptr); }
int rpt_scene_add_light_directional(rpt_scene* s, const double color[3], const double direction[3]) {
    if (!direction) return fail(RPT_ERR_INVALID, "null direction");
    return add_simple_light(s, L_DIRECTIONAL, color, direction);
}
int rpt_scene_add_light_object(rpt_scene* s, const rpt_shape_desc* d, const rpt_material* m) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (s->committed) return fail(RPT_ERR_STATE, "scene is immutable after rpt_scene_commit");
    std::string why;
    HLight l{};
    l.kind = L_OBJECT;
    MeshCallCache seen;
    if (!copy_shape(s, seen, d, l.obj.shape, why) || !check_material(m, why)) return fail(RPT_ERR_INVALID, why);
    if (d->kind == RPT_SHAPE_PLANE)
        return fail(RPT_ERR_INVALID, "a plane cannot be a Light::Object (Plane::sample is unimplemented in rpt)");
    l.obj.mat = *m;
    s->lights.push_back(std::move(l));
    return RPT_OK;
}
int rpt_scene_add_medium(rpt_scene* s, int32_t kind, double absorption, double scattering) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (s->committed) return fail(RPT_ERR_STATE, "scene is immutable after rpt_scene_commit");
    if (kind != RPT_MEDIUM_HOMOGENEOUS_ISOTROPIC && kind != RPT_MEDIUM_COLORED_GLOWING_FOG)
        return fail(RPT_ERR_INVALID, "unknown medium kind");
    if (!(absorption + scattering > 0.0)) return fail(RPT_ERR_INVALID, "medium extinction must be > 0");
    s->media.push_back(HMedium{kind, absorption, scattering});
    return RPT_OK;
}
int rpt_scene_set_environment_color(rpt_scene* s, const double rgb[3]) {
    if (!s || !rgb) return fail(RPT_ERR_INVALID, "null argument");
    if (s->committed) return fail(RPT_ERR_STATE, "scene is immutable after rpt_scene_commit");
    std::memcpy(s->env, rgb, 24);
    s->hdri_w = s->hdri_h = 0;
    s->hdri.clear();
    s->hdri64.clear();
    return RPT_OK;
}
int rpt_scene_set_environment_hdri(rpt_scene* s, uint32_t width, uint32_t height, const double* rgb) {
    if (!s || !rgb) return fail(RPT_ERR_INVALID, "null argument");
    if (s->committed) return fail(RPT_ERR_STATE, "scene is immutable after rpt_scene_commit");
    if (width == 0 || height == 0) return fail(RPT_ERR_INVALID, "Hdri::new asserts width > 0 && height > 0");
    s->hdri_w = width;
    s->hdri_h = height;
    s->hdri.resize(size_t(width) * height * 4);
    s->hdri64.assign(rgb, rgb + size_t(width) * height * 3);
    for (size_t i = 0; i < size_t(width) * height; i++) {
        s->hdri[4 * i] = float(rgb[3 * i]); s->hdri[4 * i + 1] = float(rgb[3 * i + 1]); s->hdri[4 * i + 2] = float(rgb[3 * i + 2]);
        s->hdri[4 * i + 3] = 0.f;
    }
    return RPT_OK;
}

// ---------------------------------------------------------------------------- commit (flatten + upload)
static const uint64_t kLinearTriMax = 32;  // meshes up to this size are scanned linearly (scalar loads)

static bool axis_aligned_positive(const Xf& x) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (i != j && x.M[i][j] != 0.0) return false;
    return x.M[0][0] > 0.0 && x.M[1][1] > 0.0 && x.M[2][2] > 0.0;
}

// Two consecutive flat, coplanar triangles of one mesh whose union is an axis-aligned rectangle
// (what polygon() produces for a wall): returns the axis (0/1/2) and fills the records, or -1.
static int detect_rect(const double* t0, const double* t1, const Xf& x, uint32_t obj, RectScan& rs, RectShade& rh) {
    for (int k = 1; k < 6; k++)  // all six vertex normals identical (flat, same plane orientation)
        if (std::memcmp(t0 + 9, (k < 3 ? t0 + 9 + 3 * k : t1 + 9 + 3 * (k - 3)), 24) != 0) return -1;
    D3 p[6] = {x.point(d3(t0)), x.point(d3(t0 + 3)), x.point(d3(t0 + 6)), x.point(d3(t1)), x.point(d3(t1 + 3)), x.point(d3(t1 + 6))};
    auto eq = [](const D3& a, const D3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    // the second triangle must share exactly two vertices with the first
    D3 q[4] = {p[0], p[1], p[2], p[0]};
    int shared = 0, extra = -1;
    for (int k = 3; k < 6; k++) {
        bool s = eq(p[k], p[0]) || eq(p[k], p[1]) || eq(p[k], p[2]);
        if (s) shared++;
        else extra = k;
    }
    if (shared != 2 || extra < 0) return -1;
    q[3] = p[extra];
    int lone = -1;  // vertex of the first triangle that is not on the shared edge
    for (int k = 0; k < 3; k++) {
        bool s = false;
        for (int m = 3; m < 6; m++) s = s || eq(p[k], p[m]);
        if (!s) lone = k;
    }
    if (lone < 0) return -1;
    for (int axis = 0; axis < 3; axis++) {
        auto c = [&](const D3& v, int a) { return comp(v, a % 3); };
        double pc = c(q[0], axis);
        if (c(q[1], axis) != pc || c(q[2], axis) != pc || c(q[3], axis) != pc) continue;
        int ua = axis + 1, va = axis + 2;
        double umin = c(q[0], ua), umax = umin, vmin = c(q[0], va), vmax = vmin;
        for (int k = 1; k < 4; k++) {
            umin = std::min(umin, c(q[k], ua)); umax = std::max(umax, c(q[k], ua));
            vmin = std::min(vmin, c(q[k], va)); vmax = std::max(vmax, c(q[k], va));
        }
        if (!(umax > umin && vmax > vmin)) return -1;
        int seen = 0;
        for (int k = 0; k < 4; k++) {
            double u = c(q[k], ua), v = c(q[k], va);
            if ((u != umin && u != umax) || (v != vmin && v != vmax)) return -1;
            seen |= 1 << ((u == umax ? 1 : 0) | (v == vmax ? 2 : 0));
        }
        if (seen != 15) return -1;
        // the unshared vertices must be opposite corners (so the shared edge is the other diagonal)
        const D3 &a = p[lone], &b = q[3];
        if (c(a, ua) == c(b, ua) || c(a, va) == c(b, va)) return -1;
        D3 n = x.has ? x.normal(d3(t0 + 9)) : d3(t0 + 9);
        n = normalize(n);
        rs.a = F4{float(pc), float(umin), float(umax), float(vmin)};
        rs.b = F4{float(vmax), 0.f, bits_f(obj), 0.f};
        rh.n_obj = f4(n, 0);
        rh.n_obj.w = bits_f(obj);
        return axis;
    }
    return -1;
}

static void push_tri(const double* t, const Xf& x, uint32_t obj, std::vector<TriScan>& scan,
                     std::vector<TriShade>& shade) {
    D3 v1 = x.point(d3(t)), v2 = x.point(d3(t + 3)), v3 = x.point(d3(t + 6));
    D3 n1 = d3(t + 9), n2 = d3(t + 12), n3 = d3(t + 15);
    bool flat = std::memcmp(t + 9, t + 12, 24) == 0 && std::memcmp(t + 9, t + 15, 24) == 0;
    D3 w1 = x.has ? x.normal(n1) : n1, w2 = x.has ? x.normal(n2) : n2, w3 = x.has ? x.normal(n3) : n3;
    D3 d0 = v2 - v1, d1 = v3 - v1;
    D3 pn = normalize(cross(d0, d1));
    double d00 = dot(d0, d0), d01 = dot(d0, d1), d11 = dot(d1, d1);
    double denom = d00 * d11 - d01 * d01;
    D3 A = (1.0 / denom) * (d11 * d0 - d01 * d1);
    D3 B = (1.0 / denom) * (d00 * d1 - d01 * d0);
    TriScan ts;
    ts.pn = f4(pn, dot(pn, v1));
    ts.A = f4(A, -dot(A, v1));
    ts.B = f4(B, -dot(B, v1));
    TriShade sh;
    if (flat) {
        sh.n1 = f4(normalize(w1), 0);
        sh.n2 = f4(D3{0, 0, 0}, 1.0);
        sh.n3 = f4(D3{0, 0, 0}, 0);
    } else {
        sh.n1 = f4(w1, 0);
        sh.n2 = f4(w2, 0.0);
        sh.n3 = f4(w3, 0);
    }
    sh.n1.w = bits_f(obj);
    scan.push_back(ts);
    shade.push_back(sh);
}

// ---------------------------------------------------------------------------- commit
// rpt_scene_commit = flatten_objects (records of every object + one tree per large mesh) -> flatten_lights ->
// fold_shell (flatten-time specialisation of the wall rectangles) -> mark_twin_ranges -> build_scene_tree ->
// collect_scan_boxes -> upload.  Everything before `upload` is pure host code over the fp64 scene;
// tests/host/flatten_harness.cpp runs the whole sequence with malloc-backed HIP stubs under ASan / UBSan.
namespace {
struct PBox { float lo[3], hi[3]; };   // world-space box of a bounded primitive
struct Flattener {
    rpt_scene* s;
    explicit Flattener(rpt_scene* scene) : s(scene) {}

    std::vector<XfScan> sph, cub;
    std::vector<XfShade> sph_sh, cub_sh;
    std::vector<PlaneScan> pln;
    std::vector<PlaneShade> pln_sh;
    std::vector<TriScan> tri, btri;
    std::vector<TriShade> tri_sh, btri_sh;
    std::vector<AabbScan> aabb;
    std::vector<RectScan> rect_axis[3];
    std::vector<RectShade> rect_sh_axis[3];
    std::vector<BvhNode> nodes;
    std::vector<MeshRef> meshes;
    std::vector<Material> mats;
    std::vector<Light> lights;
    std::vector<LightTri> ltris;
    std::vector<LightXf> lxf;
    std::vector<LightPart> lparts;

    // world-space boxes of the bounded primitives (scene-level tree, scan boxes)
    std::vector<PBox> box_sph, box_cub, box_tri, box_mesh;
    std::vector<InstRec> insts;
    std::vector<PBox> box_inst;
    std::unordered_map<const std::vector<double>*, uint32_t> mesh_uses;
    std::unordered_map<const std::vector<double>*, std::pair<MeshRef, PBox>> shared;
    int mesh_depth = 0, top_depth = 0;  // deepest mesh tree / the scene-level tree: their sum must fit the walk's stack
    // what the later stages leave
    ShellScan shell{};
    bool has_shell = false;
    std::vector<RectShade> shell_sh;   // shade records of the folded rectangles, in face order
    std::vector<RectScan> rect;        // the scanned rectangles, sorted by axis
    std::vector<RectShade> rect_sh;    // ... their shade records, followed by shell_sh
    std::vector<uint32_t> pleaf;
    uint32_t top_root = 0;
    bool scene_bvh = false;
    bool mesh_deferred = false;   // the scene tree does not hold the meshes that have trees of their own
    std::vector<AabbScan> pbox;

    static PBox xf_box(const Xf& x, bool sphere) {  // unit sphere / unit cube under an affine map
        PBox b;
        for (int i = 0; i < 3; i++) {
            double h = 0.0;
            for (int j = 0; j < 3; j++) h += sphere ? x.M[i][j] * x.M[i][j] : 0.5 * std::fabs(x.M[i][j]);
            if (sphere) h = std::sqrt(h);
            h += 1e-5 * (h + std::fabs(x.M[i][3]));
            b.lo[i] = float(x.M[i][3] - h);
            b.hi[i] = float(x.M[i][3] + h);
        }
        return b;
    }
    static PBox tri_box(const double* t, const Xf& x) {
        PBox b;
        D3 v[3] = {x.point(d3(t)), x.point(d3(t + 3)), x.point(d3(t + 6))};
        for (int a = 0; a < 3; a++) {
            double c0 = comp(v[0], a), c1 = comp(v[1], a), c2 = comp(v[2], a);
            double lo = std::min(c0, std::min(c1, c2)), hi = std::max(c0, std::max(c1, c2));
            b.lo[a] = std::nextafter(float(lo), -std::numeric_limits<float>::infinity());
            b.hi[a] = std::nextafter(float(hi), std::numeric_limits<float>::infinity());
        }
        return b;
    }
    // Triangles of one mesh under `x` (identity for a shared mesh) + its two-box tree, appended to btri / nodes.
    MeshRef add_bvh_mesh(const std::vector<double>& mt, const Xf& x, uint32_t obj, PBox& box) {
        const uint64_t nt = mt.size() / 18;
        std::vector<TriScan> ms;
        std::vector<TriShade> mh;
        ms.reserve(nt);
        mh.reserve(nt);
        std::vector<BTri> bt(nt);
        for (uint64_t i = 0; i < nt; i++) {
            const double* t = &mt[i * 18];
            push_tri(t, x, obj, ms, mh);
            PBox tb = tri_box(t, x);
            for (int a = 0; a < 3; a++) {
                bt[i].lo[a] = tb.lo[a];
                bt[i].hi[a] = tb.hi[a];
                bt[i].c[a] = 0.5f * (tb.lo[a] + tb.hi[a]);
            }
            bt[i].idx = uint32_t(i);
        }
        MeshRef mr;
        mr.root = uint32_t(nodes.size());  // local node 0 is the root
        mr.tri_base = uint32_t(btri.size());
        mr.tri_count = uint32_t(nt);
        mr.object = obj;
        std::vector<TmpNode> tmp;
        tmp.reserve(nt);
        tmp.push_back(TmpNode{});
        int depth = 0;
        {
            BvhBuilder b{bt, tmp};
            b.leaf_max = uint32_t(s->opt.bvh_leaf_max);
            b.sweep_below = uint32_t(std::min<int64_t>(s->opt.bvh_sweep_below, 1 << 30));
            b.build(0, 0, uint32_t(nt), 0);
            depth = b.max_depth;
        }
        if (depth > s->opt.bvh_max_depth) {  // a chain-like SAH tree: the walk's stack could not hold it
            tmp.assign(1, TmpNode{});
            BvhBuilder b{bt, tmp};
            b.leaf_max = uint32_t(s->opt.bvh_leaf_max);
            b.sweep_below = 0;
            b.balanced = true;
            b.build(0, 0, uint32_t(nt), 0);
            depth = b.max_depth;
        }
        mesh_depth = std::max(mesh_depth, depth);
        // convert to two-box nodes: inner tmp node k -> wide node remap[k]
        std::vector<uint32_t> remap(tmp.size(), 0);
        uint32_t n_inner = 0;
        for (size_t k = 0; k < tmp.size(); k++)
            if (tmp[k].count == 0) remap[k] = n_inner++;
        std::vector<BvhNode> local(n_inner);
        auto entry = [&](uint32_t k) -> uint32_t {  // absolute node / triangle indices
            const TmpNode& c = tmp[k];
            if (c.count == 0) return mr.root + remap[k];
            return BVH_LEAF | ((c.count - 1u) << 26) | (mr.tri_base + c.left_or_first);
        };
        for (size_t k = 0; k < tmp.size(); k++) {
            if (tmp[k].count != 0) continue;
            BvhNode& w = local[remap[k]];
            uint32_t l = tmp[k].left_or_first;
            for (int a = 0; a < 3; a++) {
                w.lo0[a] = tmp[l].lo[a]; w.hi0[a] = tmp[l].hi[a];
                w.lo1[a] = tmp[l + 1].lo[a]; w.hi1[a] = tmp[l + 1].hi[a];
            }
            w.e0 = entry(l);
            w.e1 = entry(l + 1);
            w.pad0 = w.pad1 = 0;
        }
        for (uint64_t i = 0; i < nt; i++) {
            btri.push_back(ms[bt[i].idx]);
            btri_sh.push_back(mh[bt[i].idx]);
        }
        nodes.insert(nodes.end(), local.begin(), local.end());
        for (int a = 0; a < 3; a++) { box.lo[a] = tmp[0].lo[a]; box.hi[a] = tmp[0].hi[a]; }
        return mr;
    }
    void emit(const HShape& shape, const Xf& x, uint32_t obj) {
        switch (shape.d.kind) {
            case RPT_SHAPE_GROUP: {  // KdTree<Box<dyn Bounded>>: children become primitives of this object
                for (const HShape& c : shape.children) {
                    Xf cx;
                    compose_xf(x, c.d, cx);
                    emit(c, cx, obj);
                }
                break;
            }
            case RPT_SHAPE_SPHERE:
            case RPT_SHAPE_CUBE: {
                if (shape.d.kind == RPT_SHAPE_CUBE && axis_aligned_positive(x)) {
                    // positive scale + translation only: an axis-aligned box in world space
                    AabbScan b;
                    b.lo = F4{float(x.M[0][3] - 0.5 * x.M[0][0]), float(x.M[1][3] - 0.5 * x.M[1][1]),
                              float(x.M[2][3] - 0.5 * x.M[2][2]), bits_f(obj)};
                    b.hi = F4{float(x.M[0][3] + 0.5 * x.M[0][0]), float(x.M[1][3] + 0.5 * x.M[1][1]),
                              float(x.M[2][3] + 0.5 * x.M[2][2]), 0.f};
                    aabb.push_back(b);
                    break;
                }
                XfScan sc;
                sc.r0 = F4{float(x.Minv[0][0]), float(x.Minv[0][1]), float(x.Minv[0][2]), float(x.Minv[0][3])};
                sc.r1 = F4{float(x.Minv[1][0]), float(x.Minv[1][1]), float(x.Minv[1][2]), float(x.Minv[1][3])};
                sc.r2 = F4{float(x.Minv[2][0]), float(x.Minv[2][1]), float(x.Minv[2][2]), float(x.Minv[2][3])};
                XfShade sh;
                sh.r0 = F4{float(x.N[0][0]), float(x.N[0][1]), float(x.N[0][2]), bits_f(obj)};
                sh.r1 = F4{float(x.N[1][0]), float(x.N[1][1]), float(x.N[1][2]), x.has ? 1.f : 0.f};
                sh.r2 = F4{float(x.N[2][0]), float(x.N[2][1]), float(x.N[2][2]), 0.f};
                if (shape.d.kind == RPT_SHAPE_SPHERE) { sph.push_back(sc); sph_sh.push_back(sh); box_sph.push_back(xf_box(x, true)); }
                else { cub.push_back(sc); cub_sh.push_back(sh); box_cub.push_back(xf_box(x, false)); }
                break;
            }
            case RPT_SHAPE_PLANE: {
                // world-space plane: (M^-T n) . x = value + (M^-T n) . translation
                D3 n = d3(shape.d.plane_normal);
                double value = shape.d.plane_value;
                if (x.has) {
                    D3 nw = x.normal(n);
                    value = value + dot(nw, D3{x.M[0][3], x.M[1][3], x.M[2][3]});
                    n = nw;
                }
                pln.push_back(PlaneScan{f4(n, value)});
                pln_sh.push_back(PlaneShade{f4(normalize(n), 0)});
                pln_sh.back().unit_n_obj.w = bits_f(obj);
                break;
            }
            default: {
                uint64_t nt = shape.T().size() / 18;
                if (nt <= kLinearTriMax) {
                    for (uint64_t i = 0; i < nt; i++) {
                        RectScan rs;
                        RectShade rh;
                        int axis = -1;
                        if (i + 1 < nt) axis = detect_rect(&shape.T()[i * 18], &shape.T()[(i + 1) * 18], x, obj, rs, rh);
                        if (axis >= 0) {
                            rect_axis[axis].push_back(rs);
                            rect_sh_axis[axis].push_back(rh);
                            i++;
                        } else {
                            push_tri(&shape.T()[i * 18], x, obj, tri, tri_sh);
                            box_tri.push_back(tri_box(&shape.T()[i * 18], x));
                        }
                    }
                } else if (s->opt.instancing && mesh_uses[shape.mesh.get()] >= 2) {
                    // shared mesh: one local-space tree, one InstRec per use
                    auto it = shared.find(shape.mesh.get());
                    if (it == shared.end()) {
                        Xf ident;
                        rpt_shape_desc none{};
                        make_xf(none, ident);
                        PBox lb;
                        MeshRef mr = add_bvh_mesh(shape.T(), ident, 0xFFFFFFFFu, lb);
                        it = shared.emplace(shape.mesh.get(), std::make_pair(mr, lb)).first;
                    }
                    const PBox& lb = it->second.second;
                    InstRec r;
                    r.r0 = F4{float(x.Minv[0][0]), float(x.Minv[0][1]), float(x.Minv[0][2]), float(x.Minv[0][3])};
                    r.r1 = F4{float(x.Minv[1][0]), float(x.Minv[1][1]), float(x.Minv[1][2]), float(x.Minv[1][3])};
                    r.r2 = F4{float(x.Minv[2][0]), float(x.Minv[2][1]), float(x.Minv[2][2]), float(x.Minv[2][3])};
                    r.n0 = F4{float(x.N[0][0]), float(x.N[0][1]), float(x.N[0][2]), bits_f(obj)};
                    r.n1 = F4{float(x.N[1][0]), float(x.N[1][1]), float(x.N[1][2]), bits_f(it->second.first.root)};
                    r.n2 = F4{float(x.N[2][0]), float(x.N[2][1]), float(x.N[2][2]), 0.f};
                    PBox wb;
                    for (int k = 0; k < 3; k++) { wb.lo[k] = std::numeric_limits<float>::infinity(); wb.hi[k] = -wb.lo[k]; }
                    for (int c = 0; c < 8; c++) {  // the instance's box: the local box's corners under the affine map
                        D3 q = x.point(D3{(c & 1) ? lb.hi[0] : lb.lo[0], (c & 2) ? lb.hi[1] : lb.lo[1], (c & 4) ? lb.hi[2] : lb.lo[2]});
                        const double qq[3] = {q.x, q.y, q.z};
                        for (int k = 0; k < 3; k++) {
                            double pad = 1e-5 * (std::fabs(qq[k]) + 1e-30);
                            wb.lo[k] = std::min(wb.lo[k], float(qq[k] - pad));
                            wb.hi[k] = std::max(wb.hi[k], float(qq[k] + pad));
                        }
                    }
                    insts.push_back(r);
                    box_inst.push_back(wb);
                } else {
                    PBox mb;
                    meshes.push_back(add_bvh_mesh(shape.T(), x, obj, mb));
                    box_mesh.push_back(mb);
                }
            }
        }
    }
    // (1) the records of every object, and one two-box tree per large mesh
    int flatten_objects() {
        // meshes referenced by more than one shape (Arc<Mesh> in the reference) are instanced
        std::function<void(const HShape&)> count_uses = [&](const HShape& shape) {
            if (shape.d.kind == RPT_SHAPE_MESH && shape.T().size() / 18 > kLinearTriMax) mesh_uses[shape.mesh.get()]++;
            for (const HShape& c : shape.children) count_uses(c);
        };
        for (const HObject& o : s->objects) count_uses(o.shape);
        for (size_t oi = 0; oi < s->objects.size(); oi++) {
            const HObject& o = s->objects[oi];
            Xf x;
            make_xf(o.shape.d, x);
            Material gm;
            gm.albedo_emit = F4{float(o.mat.albedo[0]), float(o.mat.albedo[1]), float(o.mat.albedo[2]), float(o.mat.emittance)};
            gm.params = F4{bits_f(uint32_t(o.mat.kind)), float(o.mat.shininess), float(o.mat.ior), 0.f};
            mats.push_back(gm);
            emit(o.shape, x, uint32_t(oi));
        }
        if (tri.size() >= BVH_INDEX_MASK || btri.size() >= BVH_INDEX_MASK || nodes.size() >= (1u << 30))
            return fail(RPT_ERR_UNSUPPORTED, "too many triangles");
        // the walk's stack column holds 21 entries in the per-mesh-tree kernels (a tree of depth d pushes at most d - 1)
        if (mesh_depth > 20) return fail(RPT_ERR_UNSUPPORTED, "a mesh tree is deeper than 20 levels even when rebuilt balanced (mesh too large)");

        return RPT_OK;
    }
    // (2) the light list in scene order (it fixes the RNG draw order) and the samplers' data
    void flatten_lights() {
        for (const HLight& hl : s->lights) {
            Light L{};
            L.kind = uint32_t(hl.kind);
            L.twin_object = -1;
            L.color = F4{float(hl.color[0]), float(hl.color[1]), float(hl.color[2]), 0.f};
            if (hl.kind == L_OBJECT) {
                const HObject& o = hl.obj;
                for (size_t j = 0; j < s->objects.size(); j++)
                    if (same_shape(o.shape, s->objects[j].shape)) { L.twin_object = int32_t(j); break; }
                // material.color() * material.emittance() (src/light.rs:41, material.rs:100-113)
                bool has = o.mat.kind == RPT_MAT_LAMBERTIAN || o.mat.kind == RPT_MAT_PHONG;
                double e = has ? o.mat.emittance : 0.0;
                L.color = F4{float(has ? o.mat.albedo[0] * e : 0.0), float(has ? o.mat.albedo[1] * e : 0.0),
                             float(has ? o.mat.albedo[2] * e : 0.0), 0.f};
                L.albedo = F4{float(has ? o.mat.albedo[0] : 0.0), float(has ? o.mat.albedo[1] : 0.0),
                              float(has ? o.mat.albedo[2] : 0.0), 0.f};
                // leaf shape -> LightXf (+ triangles); group -> parts tree (children contiguous, nested groups appended)
                std::function<LightPart(const HShape&, const Xf&)> light_part = [&](const HShape& shp, const Xf& x) -> LightPart {
                    LightPart part{};
                    if (shp.d.kind == RPT_SHAPE_GROUP) {
                        part.shape = LS_GROUP;
                        part.first = uint32_t(lparts.size());
                        part.count = uint32_t(shp.children.size());
                        lparts.resize(lparts.size() + shp.children.size());
                        for (size_t c = 0; c < shp.children.size(); c++) {
                            Xf cx;
                            compose_xf(x, shp.children[c].d, cx);
                            const LightPart child = light_part(shp.children[c], cx);  // may grow lparts: index, not reference
                            lparts[part.first + c] = child;
                        }
                        return part;
                    }
                    LightXf gx;
                    for (int r = 0; r < 3; r++) {
                        gx.fwd[r] = F4{float(x.M[r][0]), float(x.M[r][1]), float(x.M[r][2]), float(x.M[r][3])};
                        gx.inv[r] = F4{float(x.Minv[r][0]), float(x.Minv[r][1]), float(x.Minv[r][2]), float(x.Minv[r][3])};
                        gx.nrm[r] = F4{float(x.N[r][0]), float(x.N[r][1]), float(x.N[r][2]), 0.f};
                        gx.lin[r] = F4{float(x.L[r][0]), float(x.L[r][1]), float(x.L[r][2]), 0.f};
                    }
                    gx.nrm[0].w = float(x.det);
                    gx.nrm[1].w = x.has ? 1.f : 0.f;
                    part.xf = uint32_t(lxf.size());
                    lxf.push_back(gx);
                    if (shp.d.kind == RPT_SHAPE_MESH) {
                        part.shape = LS_MESH;
                        part.first = uint32_t(ltris.size());
                        const uint64_t nt = shp.T().size() / 18;
                        part.count = uint32_t(nt);
                        for (uint64_t i = 0; i < nt; i++) {
                            const double* t = &shp.T()[i * 18];
                            D3 a = d3(t), b = d3(t + 3), c = d3(t + 6);
                            D3 cr = cross(b - a, c - a);
                            double area = 0.5 * std::sqrt(dot(cr, cr));  // local-space area (src/shape/mesh.rs:93)
                            LightTri lt;
                            lt.v1 = f4(x.point(a), 1.0 / area);
                            lt.v2 = f4(x.point(b), 0);
                            lt.v3 = f4(x.point(c), 0);
                            lt.n1 = f4(d3(t + 9), 0);   // local normals; Transformed::sample maps them per sample
                            lt.n2 = f4(d3(t + 12), 0);
                            lt.n3 = f4(d3(t + 15), 0);
                            ltris.push_back(lt);
                        }
                    } else {
                        part.shape = shp.d.kind == RPT_SHAPE_SPHERE ? LS_SPHERE : LS_CUBE;
                    }
                    return part;
                };
                Xf x;
                make_xf(o.shape.d, x);
                const LightPart root = light_part(o.shape, x);
                L.shape = root.shape;
                L.first = root.first;
                L.count = root.count;
                L.xf = root.xf;
            }
            lights.push_back(L);
        }

    }
    // (3) flatten-time specialisation of the wall rectangles, then the final rectangle arrays
    void fold_shell() {
        // ---- box shell: rectangles that are exactly the faces of the box around all rectangles (the walls of a
        // room) leave the scanned list and are answered by one slab test; linear-scan scenes only
        const size_t n_items_total = sph.size() + cub.size() + aabb.size() + tri.size() + insts.size() + meshes.size() +
                                     rect_axis[0].size() + rect_axis[1].size() + rect_axis[2].size();
        const bool will_bvh = n_items_total >= size_t(std::max<int64_t>(2, s->opt.scene_bvh_min)) || !insts.empty();
        {
            const float inf = std::numeric_limits<float>::infinity();
            float blo[3] = {inf, inf, inf}, bhi[3] = {-inf, -inf, -inf};
            for (int a = 0; a < 3; a++)
                for (const RectScan& r : rect_axis[a]) {
                    const float lo[3] = {r.a.x, r.a.y, r.a.w}, hi[3] = {r.a.x, r.a.z, r.b.x};  // axis, u, v
                    for (int k = 0; k < 3; k++) {
                        blo[(a + k) % 3] = std::min(blo[(a + k) % 3], lo[k]);
                        bhi[(a + k) % 3] = std::max(bhi[(a + k) % 3], hi[k]);
                    }
                }
            int face_of[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}}, n_faces = 0;  // index into rect_axis[a]
            if (!will_bvh && s->opt.room_shell)
                for (int a = 0; a < 3; a++)
                    for (size_t i = 0; i < rect_axis[a].size(); i++) {
                        const RectScan& r = rect_axis[a][i];
                        const int u = (a + 1) % 3, w = (a + 2) % 3;
                        if (r.a.y != blo[u] || r.a.z != bhi[u] || r.a.w != blo[w] || r.b.x != bhi[w]) continue;
                        for (int side = 0; side < 2; side++)
                            if (r.a.x == (side ? bhi[a] : blo[a]) && face_of[a][side] < 0 && blo[a] < bhi[a]) {
                                face_of[a][side] = int(i);
                                n_faces++;
                                break;
                            }
                    }
            if (n_faces >= 3) {
                has_shell = true;
                shell.lo = F4{blo[0], blo[1], blo[2], 0.f};
                shell.hi = F4{bhi[0], bhi[1], bhi[2], 0.f};
                for (int k = 0; k < 8; k++) shell.face[k] = CODE_MISS;
                size_t n_scanned = 0;
                for (int a = 0; a < 3; a++) n_scanned += rect_axis[a].size();
                n_scanned -= size_t(n_faces);
                for (int a = 0; a < 3; a++) {  // take the members out (higher index first, so indices stay valid)
                    int order[2] = {0, 1};
                    if (face_of[a][0] >= 0 && face_of[a][1] >= 0 && face_of[a][0] < face_of[a][1]) { order[0] = 1; order[1] = 0; }
                    RectShade keep[2];
                    for (int k = 0; k < 2; k++) {
                        const int side = order[k];
                        if (face_of[a][side] < 0) continue;
                        keep[side] = rect_sh_axis[a][size_t(face_of[a][side])];
                        rect_axis[a].erase(rect_axis[a].begin() + face_of[a][side]);
                        rect_sh_axis[a].erase(rect_sh_axis[a].begin() + face_of[a][side]);
                    }
                    for (int side = 0; side < 2; side++)
                        if (face_of[a][side] >= 0) {
                            shell.face[2 * a + side] = (K_RECT << 28) | uint32_t(n_scanned + shell_sh.size());
                            shell_sh.push_back(keep[side]);
                        }
                }
            }
        }
        for (int a = 0; a < 3; a++) {
            rect.insert(rect.end(), rect_axis[a].begin(), rect_axis[a].end());
            rect_sh.insert(rect_sh.end(), rect_sh_axis[a].begin(), rect_sh_axis[a].end());
        }
        rect_sh.insert(rect_sh.end(), shell_sh.begin(), shell_sh.end());  // hit codes of shell faces point here
    }
    // (4) hit-code ranges of the lights' twin objects
    void mark_twin_ranges() {
        // ---- shadow test shortcut: when the primitives of a light's twin object form one contiguous range of
        // hit codes, "the closest hit belongs to the twin" is a range compare instead of a shade-record load
        for (Light& L : lights) {
            L.twin_lo = 1u;
            L.twin_hi = 0u;
            if (L.kind != L_OBJECT || L.twin_object < 0) continue;
            const uint32_t tw = uint32_t(L.twin_object);
            std::vector<uint32_t> codes;
            bool other = false;  // pieces a range cannot describe
            auto same = [&](float w) { return bits_u(w) == tw; };
            for (size_t i = 0; i < sph_sh.size(); i++) if (same(sph_sh[i].r0.w)) codes.push_back((K_SPHERE << 28) | uint32_t(i));
            for (size_t i = 0; i < cub_sh.size(); i++) if (same(cub_sh[i].r0.w)) codes.push_back((K_CUBE << 28) | uint32_t(i));
            for (size_t i = 0; i < pln_sh.size(); i++) if (same(pln_sh[i].unit_n_obj.w)) codes.push_back((K_PLANE << 28) | uint32_t(i));
            for (size_t i = 0; i < tri_sh.size(); i++) if (same(tri_sh[i].n1.w)) codes.push_back((K_TRI << 28) | uint32_t(i));
            for (size_t i = 0; i < aabb.size(); i++) if (same(aabb[i].lo.w)) codes.push_back((K_AABB << 28) | uint32_t(i));
            for (size_t i = 0; i < rect_sh.size(); i++) if (same(rect_sh[i].n_obj.w)) codes.push_back((K_RECT << 28) | uint32_t(i));
            for (const MeshRef& m : meshes) if (m.object == tw) other = true;
            for (const InstRec& r : insts) if (same(r.n0.w)) other = true;
            if (other || codes.empty()) continue;
            bool contiguous = true;
            for (size_t i = 1; i < codes.size(); i++) contiguous = contiguous && codes[i] == codes[i - 1] + 1u;
            if (contiguous) { L.twin_lo = codes.front(); L.twin_hi = codes.back(); }
        }

    }
    // (5) the scene-level tree (many-primitive scenes only)
    int build_scene_tree() {
        // ---- scene-level BVH over bounded primitives and mesh roots (many-primitive scenes only)
        {
            std::vector<BTri> items;
            std::vector<uint32_t> codes;   // by item: primitive code, or K_BVHTRI << 28 | mesh index for a mesh root
            auto add_item = [&](const float lo[3], const float hi[3], uint32_t code) {
                BTri it;
                for (int a = 0; a < 3; a++) { it.lo[a] = lo[a]; it.hi[a] = hi[a]; it.c[a] = 0.5f * (lo[a] + hi[a]); }
                it.idx = uint32_t(codes.size());
                items.push_back(it);
                codes.push_back(code);
            };
            for (size_t i = 0; i < sph.size(); i++) add_item(box_sph[i].lo, box_sph[i].hi, (K_SPHERE << 28) | uint32_t(i));
            for (size_t i = 0; i < cub.size(); i++) add_item(box_cub[i].lo, box_cub[i].hi, (K_CUBE << 28) | uint32_t(i));
            for (size_t i = 0; i < aabb.size(); i++) {
                float lo[3] = {aabb[i].lo.x, aabb[i].lo.y, aabb[i].lo.z}, hi[3] = {aabb[i].hi.x, aabb[i].hi.y, aabb[i].hi.z};
                add_item(lo, hi, (K_AABB << 28) | uint32_t(i));
            }
            {
                size_t i = 0;
                for (int axis = 0; axis < 3; axis++)
                    for (size_t k = 0; k < rect_axis[axis].size(); k++, i++) {
                        const RectScan& r = rect[i];
                        float lo[3], hi[3];
                        lo[axis] = hi[axis] = r.a.x;
                        lo[(axis + 1) % 3] = r.a.y; hi[(axis + 1) % 3] = r.a.z;
                        lo[(axis + 2) % 3] = r.a.w; hi[(axis + 2) % 3] = r.b.x;
                        add_item(lo, hi, (K_RECT << 28) | uint32_t(i));
                    }
            }
            for (size_t i = 0; i < tri.size(); i++) add_item(box_tri[i].lo, box_tri[i].hi, (K_TRI << 28) | uint32_t(i));
            for (size_t i = 0; i < insts.size(); i++) add_item(box_inst[i].lo, box_inst[i].hi, (K_INST << 28) | uint32_t(i));
            std::vector<uint8_t> solo(items.size(), 0);
            // Meshes with trees of their own: leaves of the scene tree (option "scene_tree_meshes" = 1), or -- the default when
            // the tree has anything else to hold -- left outside it: a query then walks the scene tree for the small things and
            // the mesh trees separately, and the render kernel parks the mesh walks (kernels.hip, BVH = 3).
            const size_t n_counted = items.size() + meshes.size();   // (whether a tree is built does not depend on the option)
            mesh_deferred = !s->opt.scene_tree_meshes && !meshes.empty() && items.size() >= 2;
            if (!mesh_deferred)
                for (size_t i = 0; i < meshes.size(); i++) {
                    add_item(box_mesh[i].lo, box_mesh[i].hi, (K_BVHTRI << 28) | uint32_t(i));
                    solo.push_back(1);
                }
            if (n_counted >= size_t(std::max<int64_t>(2, s->opt.scene_bvh_min)) || !insts.empty()) {  // instances live in the tree only
                scene_bvh = true;
                std::vector<TmpNode> tmp;
                tmp.reserve(2 * items.size());
                // a mesh tree hangs below a leaf of this one (spliced in, or walked as an instance on the same stack)
                for (int attempt = 0; attempt < 2; attempt++) {
                    tmp.assign(1, TmpNode{});
                    BvhBuilder b{items, tmp};
                    b.leaf_max = 2;
                    b.solo = &solo;
                    b.balanced = attempt == 1;
                    b.build(0, 0, uint32_t(items.size()), 0);
                    top_depth = b.max_depth + 1;
                    if (top_depth + mesh_depth <= 31) break;
                }
                if (top_depth + mesh_depth > 31)
                    return fail(RPT_ERR_UNSUPPORTED, "scene tree + mesh tree are deeper than the traversal stack (32 levels)");
                std::vector<uint32_t> remap(tmp.size(), 0);
                uint32_t n_inner = 0;
                for (size_t k = 0; k < tmp.size(); k++)
                    if (tmp[k].count == 0) remap[k] = n_inner++;
                top_root = uint32_t(nodes.size());
                std::vector<BvhNode> local(n_inner);
                auto entry = [&](uint32_t k) -> uint32_t {
                    const TmpNode& c = tmp[k];
                    if (c.count == 0) return top_root + remap[k];
                    const uint32_t c0 = codes[items[c.left_or_first].idx];
                    if ((c0 >> 28) == K_BVHTRI) return meshes[c0 & 0x0FFFFFFFu].root;  // always alone in its leaf
                    const uint32_t first = uint32_t(pleaf.size());
                    for (uint32_t i = 0; i < c.count; i++) pleaf.push_back(codes[items[c.left_or_first + i].idx]);
                    return BVH_LEAF | BVH_PRIMS | ((c.count - 1u) << 26) | first;
                };
                for (size_t k = 0; k < tmp.size(); k++) {
                    if (tmp[k].count != 0) continue;
                    BvhNode& w = local[remap[k]];
                    uint32_t l = tmp[k].left_or_first;
                    for (int a = 0; a < 3; a++) {
                        w.lo0[a] = tmp[l].lo[a]; w.hi0[a] = tmp[l].hi[a];
                        w.lo1[a] = tmp[l + 1].lo[a]; w.hi1[a] = tmp[l + 1].hi[a];
                    }
                    w.e0 = entry(l);
                    w.e1 = entry(l + 1);
                    w.pad0 = w.pad1 = 0;
                }
                nodes.insert(nodes.end(), local.begin(), local.end());
                if (pleaf.size() >= BVH_INDEX_MASK) return fail(RPT_ERR_UNSUPPORTED, "too many primitives");
            }
        }
        return RPT_OK;
    }
    // (6) boxes of the scanned records, for ball-limited queries
    void collect_scan_boxes() {
        // world boxes of the scanned bounded records, in scan order (SceneView::pbox)
        {
            auto push_box = [&](const float lo[3], const float hi[3]) {
                AabbScan b;
                float l[3], h[3];
                for (int a = 0; a < 3; a++) {
                    const float pad = 1e-5f * (std::fabs(lo[a]) + std::fabs(hi[a]) + (hi[a] - lo[a])) + 1e-7f;
                    l[a] = lo[a] - pad;
                    h[a] = hi[a] + pad;
                }
                b.lo = F4{l[0], l[1], l[2], 0.f};
                b.hi = F4{h[0], h[1], h[2], 0.f};
                pbox.push_back(b);
            };
            for (const PBox& b : box_sph) push_box(b.lo, b.hi);
            for (const PBox& b : box_cub) push_box(b.lo, b.hi);
            for (const AabbScan& b : aabb) { const float lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z}; push_box(lo, hi); }
            size_t i = 0;
            for (int axis = 0; axis < 3; axis++)
                for (size_t k = 0; k < rect_axis[axis].size(); k++, i++) {   // the scanned rectangles (shell faces are gone)
                    const RectScan& r = rect[i];
                    float lo[3], hi[3];
                    lo[axis] = hi[axis] = r.a.x;
                    lo[(axis + 1) % 3] = r.a.y; hi[(axis + 1) % 3] = r.a.z;
                    lo[(axis + 2) % 3] = r.a.w; hi[(axis + 2) % 3] = r.b.x;
                    push_box(lo, hi);
                }
            for (const PBox& b : box_tri) push_box(b.lo, b.hi);
        }
    }
    // (7) one arena for every array, the SceneView over it, statistics, per-launch scratch
    int upload(int device) {
        // ---- one arena for every array
        auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
        size_t off = 0;
        auto reserve = [&](size_t bytes) {
            size_t o = off;
            off += align(std::max<size_t>(bytes, 16));
            return o;
        };
        size_t o_sph = reserve(sph.size() * sizeof(XfScan)), o_sphs = reserve(sph_sh.size() * sizeof(XfShade));
        size_t o_cub = reserve(cub.size() * sizeof(XfScan)), o_cubs = reserve(cub_sh.size() * sizeof(XfShade));
        size_t o_pln = reserve(pln.size() * sizeof(PlaneScan)), o_plns = reserve(pln_sh.size() * sizeof(PlaneShade));
        size_t o_tri = reserve(tri.size() * sizeof(TriScan)), o_tris = reserve(tri_sh.size() * sizeof(TriShade));
        size_t o_pbox = reserve(pbox.size() * sizeof(AabbScan));
        size_t o_pleaf = reserve(pleaf.size() * sizeof(uint32_t));
        size_t o_shell = reserve(sizeof(ShellScan));
        size_t o_inst = reserve(insts.size() * sizeof(InstRec));
        size_t o_aabb = reserve(aabb.size() * sizeof(AabbScan));
        size_t o_rect = reserve(rect.size() * sizeof(RectScan)), o_rects = reserve(rect_sh.size() * sizeof(RectShade));
        size_t o_nodes = reserve(nodes.size() * sizeof(BvhNode));
        size_t o_btri = reserve(btri.size() * sizeof(TriScan)), o_btris = reserve(btri_sh.size() * sizeof(TriShade));
        size_t o_mesh = reserve(meshes.size() * sizeof(MeshRef));
        size_t o_mats = reserve(mats.size() * sizeof(Material));
        size_t o_lights = reserve(lights.size() * sizeof(Light));
        size_t o_ltris = reserve(ltris.size() * sizeof(LightTri));
        size_t o_lxf = reserve(lxf.size() * sizeof(LightXf));
        size_t o_lparts = reserve(lparts.size() * sizeof(LightPart));
        size_t o_hdri = reserve(s->hdri.size() * sizeof(float));
        std::vector<char> host(off, 0);
        auto put = [&](size_t o, const void* src, size_t bytes) { if (bytes) std::memcpy(host.data() + o, src, bytes); };
        put(o_sph, sph.data(), sph.size() * sizeof(XfScan));       put(o_sphs, sph_sh.data(), sph_sh.size() * sizeof(XfShade));
        put(o_cub, cub.data(), cub.size() * sizeof(XfScan));       put(o_cubs, cub_sh.data(), cub_sh.size() * sizeof(XfShade));
        put(o_pln, pln.data(), pln.size() * sizeof(PlaneScan));    put(o_plns, pln_sh.data(), pln_sh.size() * sizeof(PlaneShade));
        put(o_tri, tri.data(), tri.size() * sizeof(TriScan));      put(o_tris, tri_sh.data(), tri_sh.size() * sizeof(TriShade));
        put(o_aabb, aabb.data(), aabb.size() * sizeof(AabbScan));
        put(o_pbox, pbox.data(), pbox.size() * sizeof(AabbScan));
        put(o_pleaf, pleaf.data(), pleaf.size() * sizeof(uint32_t));
        put(o_shell, &shell, sizeof(ShellScan));
        put(o_inst, insts.data(), insts.size() * sizeof(InstRec));
        put(o_rect, rect.data(), rect.size() * sizeof(RectScan));  put(o_rects, rect_sh.data(), rect_sh.size() * sizeof(RectShade));
        put(o_nodes, nodes.data(), nodes.size() * sizeof(BvhNode));
        put(o_btri, btri.data(), btri.size() * sizeof(TriScan));   put(o_btris, btri_sh.data(), btri_sh.size() * sizeof(TriShade));
        put(o_mesh, meshes.data(), meshes.size() * sizeof(MeshRef));
        put(o_mats, mats.data(), mats.size() * sizeof(Material));
        put(o_lights, lights.data(), lights.size() * sizeof(Light));
        put(o_ltris, ltris.data(), ltris.size() * sizeof(LightTri));
        put(o_lxf, lxf.data(), lxf.size() * sizeof(LightXf));
        put(o_lparts, lparts.data(), lparts.size() * sizeof(LightPart));
        put(o_hdri, s->hdri.data(), s->hdri.size() * sizeof(float));
        HIP_TRY(hipMalloc(&s->arena, off));
        HIP_TRY(hipMemcpy(s->arena, host.data(), off, hipMemcpyHostToDevice));
        char* base = static_cast<char*>(s->arena);
        SceneView& v = s->view;
        v.sph = (const XfScan*)(base + o_sph);      v.sph_sh = (const XfShade*)(base + o_sphs);    v.n_sph = uint32_t(sph.size());
        v.cub = (const XfScan*)(base + o_cub);      v.cub_sh = (const XfShade*)(base + o_cubs);    v.n_cub = uint32_t(cub.size());
        v.pln = (const PlaneScan*)(base + o_pln);   v.pln_sh = (const PlaneShade*)(base + o_plns); v.n_pln = uint32_t(pln.size());
        v.tri = (const TriScan*)(base + o_tri);     v.tri_sh = (const TriShade*)(base + o_tris);   v.n_tri = uint32_t(tri.size());
        v.aabb = (const AabbScan*)(base + o_aabb);  v.n_aabb = uint32_t(aabb.size());
        v.rect = (const RectScan*)(base + o_rect);  v.rect_sh = (const RectShade*)(base + o_rects);
        v.n_rect_x = uint32_t(rect_axis[0].size()); v.n_rect_y = uint32_t(rect_axis[1].size()); v.n_rect_z = uint32_t(rect_axis[2].size());
        v.shell = (const ShellScan*)(base + o_shell); v.has_shell = has_shell ? 1u : 0u;
        v.pbox = (const AabbScan*)(base + o_pbox);
        v.nodes = (const BvhNode*)(base + o_nodes); v.btri = (const TriScan*)(base + o_btri);      v.btri_sh = (const TriShade*)(base + o_btris);
        v.meshes = (const MeshRef*)(base + o_mesh); v.n_mesh = uint32_t(meshes.size());
        v.pleaf = (const uint32_t*)(base + o_pleaf); v.n_nodes = uint32_t(nodes.size());
        v.scene_bvh = scene_bvh ? 1u : 0u;          v.top_root = top_root;
        v.mesh_deferred = (scene_bvh && mesh_deferred) ? 1u : 0u;
        v.inst = (const InstRec*)(base + o_inst);   v.n_inst = uint32_t(insts.size());
        v.mats = (const Material*)(base + o_mats);  v.n_obj = uint32_t(mats.size());
        v.lights = (const Light*)(base + o_lights); v.n_lights = uint32_t(lights.size());
        v.ltris = (const LightTri*)(base + o_ltris); v.lxf = (const LightXf*)(base + o_lxf);
        v.lparts = (const LightPart*)(base + o_lparts); v.n_lparts = uint32_t(lparts.size());
        v.n_ltris = uint32_t(ltris.size());
        v.has_medium = s->media.empty() ? 0u : 1u;
        v.medium_kind = 0;
        v.sigma_a = v.sigma_s = 0.f;
        v.medium_emission = 0.f;
        v.medium_phase = 0.f;
        for (int i = 0; i < 3; i++) v.medium_color[i] = v.medium_color_hi[i] = 0.f;
        if (!s->media.empty()) {  // only media[0] is used (src/renderer.rs:190)
            const HMedium& m = s->media[0];
            v.medium_kind = uint32_t(m.kind);
            v.sigma_a = float(m.absorption);
            v.sigma_s = float(m.scattering);
            const double pi = 3.14159265358979323846;
            if (m.kind == RPT_MEDIUM_HOMOGENEOUS_ISOTROPIC) {  // src/medium.rs:80-96
                D3 c = hex_color(0xD2B48C);
                v.medium_color[0] = v.medium_color_hi[0] = float(c.x);
                v.medium_color[1] = v.medium_color_hi[1] = float(c.y);
                v.medium_color[2] = v.medium_color_hi[2] = float(c.z);
                v.medium_emission = 0.f;
                v.medium_phase = float(1.0 / (4.0 * pi));
            } else {  // colored_glowing_fog, src/medium.rs:99-122 (phase `1.0 / 4.0 * pi`, sic)
                D3 lo = hex_color(0x0000FF), hi = hex_color(0xFF0000);
                v.medium_color[0] = float(lo.x); v.medium_color[1] = float(lo.y); v.medium_color[2] = float(lo.z);
                v.medium_color_hi[0] = float(hi.x); v.medium_color_hi[1] = float(hi.y); v.medium_color_hi[2] = float(hi.z);
                v.medium_emission = 10.f;
                v.medium_phase = float(1.0 / 4.0 * pi);
            }
        }
        for (int i = 0; i < 3; i++) v.env[i] = float(s->env[i]);
        v.hdri = (const F4*)(base + o_hdri);
        v.hdri_w = s->hdri_w;
        v.hdri_h = s->hdri_h;
        s->prims_per_ray = sph.size() + cub.size() + pln.size() + tri.size() + aabb.size() + rect.size() + (has_shell ? 1 : 0);
        s->twins_scanned = true;
        s->n_twin_lights = 0;
        for (const Light& L : lights) {
            if (L.kind == L_OBJECT && L.twin_object >= 0) s->n_twin_lights++;
            if (L.kind == L_OBJECT && L.twin_object >= 0 && !(L.twin_lo <= L.twin_hi)) s->twins_scanned = false;
        }
        s->stats[0] = sph.size(); s->stats[1] = cub.size(); s->stats[2] = pln.size(); s->stats[3] = tri.size();
        s->stats[4] = aabb.size(); s->stats[5] = rect_sh.size(); s->stats[6] = btri.size(); s->stats[7] = nodes.size();
        // scan-record bytes every closest-hit query walks (the uniform part of the algorithmic bytes)
        s->stats[8] = 48 * (sph.size() + cub.size() + tri.size()) + 16 * pln.size() + 32 * (aabb.size() + rect.size());
        s->stats[9] = off;
        s->stats[10] = scene_bvh ? (mesh_deferred ? 2 : 1) : 0;
        s->stats[11] = pleaf.size();
        s->stats[14] = has_shell ? shell_sh.size() : 0;
        s->stats[15] = uint64_t(top_depth + mesh_depth);
        s->stats[12] = insts.size();
        s->stats[13] = shared.size();

        for (auto& ls : s->sets) {
            HIP_TRY(hipMalloc((void**)&ls.d_queue, 256));
            HIP_TRY(hipEventCreateWithFlags(&ls.done, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ls.launched, hipEventDisableTiming));
        }
        HIP_TRY(hipMalloc((void**)&s->d_counters, 64 * sizeof(unsigned long long)));
        s->view.stack_overflows = s->d_counters + 7;
        s->device = device;
        s->committed = true;
        return RPT_OK;
    }
};
}  // namespace

// ---------------------------------------------------------------------------- reference-epsilon mode (f64_layout.h)
// The scene as the reference holds it: scene.objects in order, each a generic shape with the matrices Transformed::new
// derives (src/shape.rs:112-125), meshes as their triangles in local space, materials and lights in fp64.
// (What depends on a triangle alone is evaluated here with the reference's operations in the reference's order; the
// device must find the same bits, so nothing below may be contracted into an fma.)
#pragma clang fp contract(off)
static int check_scene64(const rpt_scene* s) {   // what the mode refuses, before anything is allocated
    std::function<int(const HShape&)> depth = [&](const HShape& h) {   // group levels around the deepest shape
        int d = 0;
        for (const HShape& c : h.children) d = std::max(d, depth(c));
        return d + (h.d.kind == RPT_SHAPE_GROUP ? 1 : 0);
    };
    for (const auto& o : s->objects)
        if (depth(o.shape) > int(rpt64::kMaxFrames))
            return fail(RPT_ERR_UNSUPPORTED, "epsilon_policy = 1 supports KdTree groups nested at most three deep");
    for (const auto& l : s->lights)
        if (l.kind == int(L_OBJECT) && depth(l.obj.shape) > int(rpt64::kMaxFrames))
            return fail(RPT_ERR_UNSUPPORTED, "epsilon_policy = 1 supports KdTree groups nested at most three deep");
    return RPT_OK;
}
static int fill_shape64(const HShape& hs, rpt64::Shape& o, std::vector<rpt64::Tri>& tris, std::vector<double>& tri_pdf,
                        std::unordered_map<const std::vector<double>*, uint32_t>* shared = nullptr) {
    std::memset(&o, 0, sizeof(o));
    o.kind = hs.d.kind;
    Xf x;
    if (!make_xf(hs.d, x)) return fail(RPT_ERR_INVALID, "singular transform");
    o.has_xf = x.has ? 1 : 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) { o.inv[i * 4 + j] = x.Minv[i][j]; o.fwd[i * 4 + j] = x.M[i][j]; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { o.lin[i * 3 + j] = x.L[i][j]; o.nrm[i * 3 + j] = x.N[i][j]; }
    o.det = x.det;
    for (int k = 0; k < 3; k++) o.plane[k] = hs.d.plane_normal[k];
    o.plane[3] = hs.d.plane_value;
    if (hs.d.kind == RPT_SHAPE_MESH) {
        const std::vector<double>& T = hs.T();
        // (an `Arc<Mesh>` used by several shapes keeps one copy of its triangles)
        const auto known = (shared && hs.mesh) ? shared->find(hs.mesh.get()) : decltype(shared->end()){};
        const bool have = shared && hs.mesh && known != shared->end();
        o.tri_first = have ? known->second : uint32_t(tris.size());
        o.tri_count = uint32_t(T.size() / 18);
        if (shared && hs.mesh && !have) (*shared)[hs.mesh.get()] = o.tri_first;
        for (int k = 0; k < 3; k++) { o.bmin[k] = std::numeric_limits<double>::infinity(); o.bmax[k] = -o.bmin[k]; }
        for (size_t t = 0; t < T.size() / 18; t++) {
            rpt64::Tri tr;
            std::memcpy(&tr, T.data() + t * 18, sizeof(tr));
            for (int v = 0; v < 3; v++)   // Triangle::bounding_box merged over the mesh (src/kdtree.rs:108-113)
                for (int k = 0; k < 3; k++) {
                    o.bmin[k] = std::min(o.bmin[k], T[t * 18 + v * 3 + k]);
                    o.bmax[k] = std::max(o.bmax[k], T[t * 18 + v * 3 + k]);
                }
            if (have) continue;
            tris.push_back(tr);
            // Triangle::sample's pdf (src/shape/mesh.rs:96-98) over KdTree::sample's choice (src/kdtree.rs:141-146)
            const double e0[3] = {tr.v2[0] - tr.v1[0], tr.v2[1] - tr.v1[1], tr.v2[2] - tr.v1[2]};
            const double e1[3] = {tr.v3[0] - tr.v1[0], tr.v3[1] - tr.v1[1], tr.v3[2] - tr.v1[2]};
            const double cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2], cz = e0[0] * e1[1] - e0[1] * e1[0];
            const double area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
            tri_pdf.push_back((1.0 / area) / double(o.tri_count));
        }
    }
    return RPT_OK;
}
static void fill_mat64(const rpt_material& m, rpt64::Mat& o) {
    std::memset(&o, 0, sizeof(o));
    o.kind = m.kind;
    for (int k = 0; k < 3; k++) o.albedo[k] = m.albedo[k];
    o.emittance = m.emittance;
    o.shininess = m.shininess;
    o.ior = m.ior;
}
// What Triangle::intersect (src/shape/mesh.rs:50-83) computes from the triangle alone, by the same operations
static rpt64::TriRec tri_rec64(const rpt64::Tri& t) {
    rpt64::TriRec r;
    double d0[3], d1[3];
    for (int k = 0; k < 3; k++) { d0[k] = t.v2[k] - t.v1[k]; d1[k] = t.v3[k] - t.v1[k]; r.v1[k] = t.v1[k]; r.d0[k] = d0[k]; r.d1[k] = d1[k]; }
    const double c[3] = {d0[1] * d1[2] - d0[2] * d1[1], d0[2] * d1[0] - d0[0] * d1[2], d0[0] * d1[1] - d0[1] * d1[0]};
    const double len = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int k = 0; k < 3; k++) r.pn[k] = c[k] / len;
    r.d00 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2];
    r.d01 = d0[0] * d1[0] + d0[1] * d1[1] + d0[2] * d1[2];
    r.d11 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
    r.denom = r.d00 * r.d11 - r.d01 * r.d01;
    return r;
}
// Padded world-space box of an object in fp32 (f64_layout.h, CullBox): the box of the shape's own bounds under its
// matrix, grown by 1e-5 of its size and of its coordinates, rounded outwards.
static rpt64::CullBox cull_box64(const rpt64::Shape& sh) {
    rpt64::CullBox c{};
    if (sh.kind == rpt64::SH_PLANE) { c.unbounded = 1u; return c; }
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        const double h = sh.kind == rpt64::SH_SPHERE ? 1.0 : 0.5;
        lo[k] = sh.kind == rpt64::SH_MESH ? sh.bmin[k] : -h;
        hi[k] = sh.kind == rpt64::SH_MESH ? sh.bmax[k] : h;
    }
    double wlo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, whi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (int corner = 0; corner < 8; corner++) {
        const double p[3] = {corner & 1 ? hi[0] : lo[0], corner & 2 ? hi[1] : lo[1], corner & 4 ? hi[2] : lo[2]};
        for (int i = 0; i < 3; i++) {
            const double w = sh.has_xf ? sh.fwd[i * 4] * p[0] + sh.fwd[i * 4 + 1] * p[1] + sh.fwd[i * 4 + 2] * p[2] + sh.fwd[i * 4 + 3] : p[i];
            wlo[i] = std::min(wlo[i], w);
            whi[i] = std::max(whi[i], w);
        }
    }
    double size = 0.0, mag = 0.0;
    for (int i = 0; i < 3; i++) { size = std::max(size, whi[i] - wlo[i]); mag = std::max(mag, std::max(std::fabs(wlo[i]), std::fabs(whi[i]))); }
    const double pad = 1e-5 * size + 1e-5 * mag + 1e-30;
    bool finite = true;
    for (int i = 0; i < 3; i++) {
        c.lo[i] = std::nextafter(float(wlo[i] - pad), -HUGE_VALF);
        c.hi[i] = std::nextafter(float(whi[i] + pad), HUGE_VALF);
        finite = finite && std::isfinite(c.lo[i]) && std::isfinite(c.hi[i]);
    }
    if (!finite) { c = rpt64::CullBox{}; c.unbounded = 1u; }   // (a box fp32 cannot hold: always evaluated)
    return c;
}
// Bounded::bounding_box of a shape in its parent's space (src/shape/*.rs, src/kdtree.rs:108-113, src/shape.rs:154-176: a
// Transformed shape's box is the box of the eight transformed corners of the inner box).
struct Box64 {
    double lo[3], hi[3];
};
static Box64 bbox64(const HShape& hs) {
    Box64 b;
    for (int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<double>::infinity(); b.hi[k] = -b.lo[k]; }
    if (hs.d.kind == RPT_SHAPE_SPHERE || hs.d.kind == RPT_SHAPE_CUBE) {
        const double h = hs.d.kind == RPT_SHAPE_SPHERE ? 1.0 : 0.5;
        for (int k = 0; k < 3; k++) { b.lo[k] = -h; b.hi[k] = h; }
    } else if (hs.d.kind == RPT_SHAPE_MESH) {
        const std::vector<double>& T = hs.T();
        for (size_t t = 0; t < T.size() / 18; t++)
            for (int v = 0; v < 3; v++)
                for (int k = 0; k < 3; k++) {
                    b.lo[k] = std::min(b.lo[k], T[t * 18 + v * 3 + k]);
                    b.hi[k] = std::max(b.hi[k], T[t * 18 + v * 3 + k]);
                }
    } else if (hs.d.kind == RPT_SHAPE_GROUP) {
        for (const HShape& c : hs.children) {
            const Box64 cb = bbox64(c);
            for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], cb.lo[k]); b.hi[k] = std::max(b.hi[k], cb.hi[k]); }
        }
    }
    if (!hs.d.has_transform) return b;
    Box64 w;
    for (int k = 0; k < 3; k++) { w.lo[k] = std::numeric_limits<double>::infinity(); w.hi[k] = -w.lo[k]; }
    const double* M = hs.d.transform;
    for (int c = 0; c < 8; c++) {
        const double p[3] = {c & 4 ? b.hi[0] : b.lo[0], c & 2 ? b.hi[1] : b.lo[1], c & 1 ? b.hi[2] : b.lo[2]};
        for (int i = 0; i < 3; i++) {
            const double v = M[i * 4] * p[0] + M[i * 4 + 1] * p[1] + M[i * 4 + 2] * p[2] + M[i * 4 + 3] * 1.0;
            w.lo[i] = std::min(w.lo[i], v);
            w.hi[i] = std::max(w.hi[i], v);
        }
    }
    return w;
}
struct Flat64 {
    std::unordered_map<const std::vector<double>*, uint32_t> shared_meshes;
    std::vector<rpt64::CullBox> cull;
    std::vector<rpt64::ObjRec> recs;
    std::vector<rpt64::ObjShade> shade;
    std::vector<rpt64::FrameRec> frames;
    std::vector<rpt64::FrameShade> fshade;
    std::vector<rpt64::Tri> tris;
    std::vector<double> tri_pdf;
};
// One shape of scene.objects[..] into records: a leaf becomes an ObjRec under the group levels in `chain`; a group adds a level and
// recurses.  `outer`: the forward matrices of the transformed levels above, outermost first (for the fp32 world box only).
static int flatten64(const HShape& hs, const rpt_material& mat, std::vector<uint32_t>& chain, std::vector<const double*>& outer, Flat64& F) {
    Xf x;
    if (!make_xf(hs.d, x)) return fail(RPT_ERR_INVALID, "singular transform");
    if (hs.d.kind == RPT_SHAPE_GROUP) {
        rpt64::FrameRec fr;
        rpt64::FrameShade fs;
        std::memset(&fr, 0, sizeof(fr));
        std::memset(&fs, 0, sizeof(fs));
        fs.has_xf = x.has ? 1 : 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) fr.inv[i * 4 + j] = x.Minv[i][j];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) fs.nrm[i * 3 + j] = x.N[i][j];
        HShape inner = hs;   // the group's bounds are those of KdTree::new: its children's boxes merged, in the group's own space
        inner.d.has_transform = 0;
        const Box64 gb = bbox64(inner);
        for (int k = 0; k < 3; k++) { fr.b[k] = gb.lo[k]; fr.b[3 + k] = gb.hi[k]; }
        chain.push_back(uint32_t(F.frames.size()));
        F.frames.push_back(fr);
        F.fshade.push_back(fs);
        if (x.has) outer.push_back(hs.d.transform);
        for (const HShape& c : hs.children)
            if (int rc = flatten64(c, mat, chain, outer, F)) return rc;
        if (x.has) outer.pop_back();
        chain.pop_back();
        return RPT_OK;
    }
    rpt64::Shape sh;
    if (int rc = fill_shape64(hs, sh, F.tris, F.tri_pdf, &F.shared_meshes)) return rc;
    rpt64::ObjRec r;
    std::memset(&r, 0, sizeof(r));
    r.kind = sh.kind; r.has_xf = sh.has_xf; r.tri_first = sh.tri_first; r.tri_count = sh.tri_count;
    r.n_frames = uint32_t(chain.size());
    for (size_t k = 0; k < chain.size(); k++) r.frame[k] = chain[k];
    for (int k = 0; k < 12; k++) r.inv[k] = sh.inv[k];
    for (int k = 0; k < 3; k++) {
        if (sh.kind == rpt64::SH_MESH) { r.b[k] = sh.bmin[k]; r.b[3 + k] = sh.bmax[k]; }
        else if (sh.kind == rpt64::SH_CUBE) { r.b[k] = -0.5; r.b[3 + k] = 0.5; }
        else if (sh.kind == rpt64::SH_PLANE) { r.b[k] = sh.plane[k]; }
    }
    if (sh.kind == rpt64::SH_PLANE) r.b[3] = sh.plane[3];
    // the fp32 world box: the shape's own box under its matrix, then under the groups' matrices, innermost first
    rpt64::CullBox cb = cull_box64(sh);
    cb.slab_test = sh.kind == rpt64::SH_CUBE ? 1u : 0u;
    if (!cb.unbounded && !outer.empty()) {
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = cb.lo[k]; hi[k] = cb.hi[k]; }
        for (size_t lvl = outer.size(); lvl-- > 0;) {
            const double* M = outer[lvl];
            double wlo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, whi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
            for (int c = 0; c < 8; c++) {
                const double p[3] = {c & 1 ? hi[0] : lo[0], c & 2 ? hi[1] : lo[1], c & 4 ? hi[2] : lo[2]};
                for (int i = 0; i < 3; i++) {
                    const double v = M[i * 4] * p[0] + M[i * 4 + 1] * p[1] + M[i * 4 + 2] * p[2] + M[i * 4 + 3];
                    wlo[i] = std::min(wlo[i], v);
                    whi[i] = std::max(whi[i], v);
                }
            }
            for (int k = 0; k < 3; k++) { lo[k] = wlo[k]; hi[k] = whi[k]; }
        }
        double size = 0.0, mag = 0.0;
        for (int i = 0; i < 3; i++) { size = std::max(size, hi[i] - lo[i]); mag = std::max(mag, std::max(std::fabs(lo[i]), std::fabs(hi[i]))); }
        const double pad = 1e-5 * size + 1e-5 * mag + 1e-30;
        bool finite = true;
        for (int i = 0; i < 3; i++) {
            cb.lo[i] = std::nextafter(float(lo[i] - pad), -HUGE_VALF);
            cb.hi[i] = std::nextafter(float(hi[i] + pad), HUGE_VALF);
            finite = finite && std::isfinite(cb.lo[i]) && std::isfinite(cb.hi[i]);
        }
        if (!finite) { cb = rpt64::CullBox{}; cb.unbounded = 1u; }
    }
    rpt64::ObjShade os;
    std::memset(&os, 0, sizeof(os));
    for (int k = 0; k < 9; k++) os.nrm[k] = sh.nrm[k];
    fill_mat64(mat, os.mat);
    F.cull.push_back(cb);
    F.recs.push_back(r);
    F.shade.push_back(os);
    return RPT_OK;
}
// A light's shape tree (a KdTree group as Light::Object): `out` gets the group's record, its children a contiguous block of
// `pool` (their own children further back), so that KdTree::sample's choice is one index computation.
static int fill_light_shape64(const HShape& hs, rpt64::Shape& out, std::vector<rpt64::Shape>& pool, Flat64& F) {
    if (int rc = fill_shape64(hs, out, F.tris, F.tri_pdf, &F.shared_meshes)) return rc;
    if (hs.d.kind != RPT_SHAPE_GROUP) return RPT_OK;
    out.kind = rpt64::SH_GROUP;
    const size_t first = pool.size();
    out.tri_first = uint32_t(first);
    out.tri_count = uint32_t(hs.children.size());
    pool.resize(first + hs.children.size());
    for (size_t c = 0; c < hs.children.size(); c++) {
        rpt64::Shape child;   // (pool may grow while the child's own children are filled in)
        if (int rc = fill_light_shape64(hs.children[c], child, pool, F)) return rc;
        pool[first + c] = child;
    }
    return RPT_OK;
}
static int build_scene64(rpt_scene* s) {
    static_assert(sizeof(rpt64::Tri) == 18 * sizeof(double), "a triangle is its 18 doubles");
    if (int rc = check_scene64(s)) return rc;
    Flat64 F;
    for (size_t i = 0; i < s->objects.size(); i++) {
        std::vector<uint32_t> chain;
        std::vector<const double*> outer;
        if (int rc = flatten64(s->objects[i].shape, s->objects[i].mat, chain, outer, F)) return rc;
    }
    std::vector<rpt64::CullBox>& cull = F.cull;
    std::vector<rpt64::ObjRec>& recs = F.recs;
    std::vector<rpt64::ObjShade>& shade = F.shade;
    std::vector<rpt64::Tri>& tris = F.tris;
    std::vector<double>& tri_pdf = F.tri_pdf;
    const size_t n = recs.size();
    std::vector<rpt64::Light> lights(s->lights.size());
    std::vector<rpt64::Shape> lshapes;
    const size_t n_obj_tris = tris.size();
    std::vector<rpt64::TriRec> trecs(n_obj_tris);
    std::vector<rpt64::TriShade> tshade(n_obj_tris);
    for (size_t t = 0; t < n_obj_tris; t++) {
        trecs[t] = tri_rec64(tris[t]);
        for (int k = 0; k < 3; k++) { tshade[t].n1[k] = tris[t].n1[k]; tshade[t].n2[k] = tris[t].n2[k]; tshade[t].n3[k] = tris[t].n3[k]; }
    }
    for (size_t i = 0; i < s->lights.size(); i++) {
        const HLight& hl = s->lights[i];
        rpt64::Light& L = lights[i];
        std::memset(&L, 0, sizeof(L));
        L.kind = hl.kind;
        for (int k = 0; k < 3; k++) L.color[k] = hl.color[k];
        if (hl.kind == int(L_OBJECT)) {
            if (int rc = fill_light_shape64(hl.obj.shape, L.shape, lshapes, F)) return rc;
            fill_mat64(hl.obj.mat, L.mat);
        }
    }
    std::vector<rpt64::CullBox> cull_groups((n + 31) / 32);
    for (size_t g = 0; g < cull_groups.size(); g++) {
        rpt64::CullBox u{};
        for (int k = 0; k < 3; k++) { u.lo[k] = HUGE_VALF; u.hi[k] = -HUGE_VALF; }
        for (size_t i = 32 * g; i < std::min(n, 32 * g + 32); i++) {
            if (cull[i].unbounded) u.unbounded = 1u;
            if (cull[i].slab_test) u.slab_test = 2u;
            for (int k = 0; k < 3; k++) { u.lo[k] = std::min(u.lo[k], cull[i].lo[k]); u.hi[k] = std::max(u.hi[k], cull[i].hi[k]); }
        }
        if (u.unbounded) for (int k = 0; k < 3; k++) u.lo[k] = u.hi[k] = 0.f;
        cull_groups[g] = u;
    }
    struct Part { const void* src; size_t bytes, off; };
    Part parts[] = {{cull.data(), cull.size() * sizeof(rpt64::CullBox), 0},       {recs.data(), recs.size() * sizeof(rpt64::ObjRec), 0},
                    {shade.data(), shade.size() * sizeof(rpt64::ObjShade), 0},    {trecs.data(), trecs.size() * sizeof(rpt64::TriRec), 0},
                    {tshade.data(), tshade.size() * sizeof(rpt64::TriShade), 0},  {tris.data(), tris.size() * sizeof(rpt64::Tri), 0},
                    {tri_pdf.data(), tri_pdf.size() * sizeof(double), 0},         {lights.data(), lights.size() * sizeof(rpt64::Light), 0},
                    {F.frames.data(), F.frames.size() * sizeof(rpt64::FrameRec), 0}, {F.fshade.data(), F.fshade.size() * sizeof(rpt64::FrameShade), 0},
                    {s->hdri64.data(), s->hdri64.size() * sizeof(double), 0},
                    {cull_groups.data(), cull_groups.size() * sizeof(rpt64::CullBox), 0},
                    {lshapes.data(), lshapes.size() * sizeof(rpt64::Shape), 0}};
    size_t total = 0;
    for (auto& p : parts) { p.off = total; total = (total + p.bytes + 255) & ~size_t(255); }
    total = std::max<size_t>(total, 256);
    HIP_TRY(hipMalloc(&s->arena64, total));
    HIP_TRY(hipMemset(s->arena64, 0, total));   // (the gaps between the arrays too: the host tests checksum the arena)
    s->arena64_bytes = total;
    char* base = static_cast<char*>(s->arena64);
    for (auto& p : parts)
        if (p.bytes) HIP_TRY(hipMemcpy(base + p.off, p.src, p.bytes, hipMemcpyHostToDevice));
    rpt64::Scene& v = s->view64;
    v.cull = reinterpret_cast<const rpt64::CullBox*>(base + parts[0].off);
    v.recs = reinterpret_cast<const rpt64::ObjRec*>(base + parts[1].off);
    v.shade = reinterpret_cast<const rpt64::ObjShade*>(base + parts[2].off);
    v.trecs = reinterpret_cast<const rpt64::TriRec*>(base + parts[3].off);
    v.tshade = reinterpret_cast<const rpt64::TriShade*>(base + parts[4].off);
    v.tris = reinterpret_cast<const rpt64::Tri*>(base + parts[5].off);
    v.tri_pdf = reinterpret_cast<const double*>(base + parts[6].off);
    v.lights = reinterpret_cast<const rpt64::Light*>(base + parts[7].off);
    v.frames = reinterpret_cast<const rpt64::FrameRec*>(base + parts[8].off);
    v.fshade = reinterpret_cast<const rpt64::FrameShade*>(base + parts[9].off);
    v.hdri = reinterpret_cast<const double*>(base + parts[10].off);
    v.cull32 = reinterpret_cast<const rpt64::CullBox*>(base + parts[11].off);
    v.lshapes = reinterpret_cast<const rpt64::Shape*>(base + parts[12].off);
    v.hdri_w = s->hdri64.empty() ? 0u : s->hdri_w;
    v.hdri_h = s->hdri64.empty() ? 0u : s->hdri_h;
    v.n_objects = uint32_t(n);
    // the buckets of the cubes' face coordinates (f64_layout.h, Scene::face_bits; the tolerance is cull32's, kernels_f64.hip)
    for (int k = 0; k < 3; k++) {
        float lo = HUGE_VALF, hi = -HUGE_VALF;
        for (size_t i = 0; i < n; i++)
            if (cull[i].slab_test && !cull[i].unbounded) { lo = std::min(lo, cull[i].lo[k]); hi = std::max(hi, cull[i].hi[k]); }
        v.face_bits[k] = 0;
        v.face_base[k] = 0.f;
        v.face_inv_cell[k] = 0.f;
        if (!(lo <= hi)) continue;   // no cube: no bucket is marked
        float tol_max = 0.f;
        for (size_t i = 0; i < n; i++)
            if (cull[i].slab_test && !cull[i].unbounded) {
                float size = 0.f, mag = 0.f;
                for (int a = 0; a < 3; a++) { size = std::max(size, cull[i].hi[a] - cull[i].lo[a]); mag = std::max({mag, std::fabs(cull[i].lo[a]), std::fabs(cull[i].hi[a])}); }
                tol_max = std::max(tol_max, 5e-5f * (size + mag));
            }
        const float mag_all = std::max(std::fabs(lo), std::fabs(hi));
        const float T = 2.f * tol_max + 4e-6f * mag_all + 1e-30f;   // cull32's tolerance (its `eo` is 1e-6 of the origin, which lies within T of the face), twice
        const float base = lo - 2.f * T, span = (hi + 2.f * T) - base;
        const float inv = 64.f / (span * 1.0001f);
        v.face_base[k] = base;
        v.face_inv_cell[k] = inv;
        auto mark = [&](float c) {
            for (float x : {c - T, c, c + T}) {
                const float f = (x - base) * inv;
                if (f >= 0.f && f < 64.f) v.face_bits[k] |= 1ull << unsigned(f);
            }
            // (c - T and c + T may lie more than one bucket apart when the buckets are narrower than 2 T: mark the run)
            const float f0 = std::max((c - T - base) * inv, 0.f), f1 = std::min((c + T - base) * inv, 63.f);
            for (int b = int(f0); b <= int(f1); b++) v.face_bits[k] |= 1ull << unsigned(b);
        };
        for (size_t i = 0; i < n; i++)
            if (cull[i].slab_test && !cull[i].unbounded) { mark(cull[i].lo[k]); mark(cull[i].hi[k]); }
    }
    v.n_lights = uint32_t(lights.size());
    v.n_tris = uint32_t(tris.size());
    v.n_obj_tris = uint32_t(n_obj_tris);
    v.has_medium = s->media.empty() ? 0 : 1;
    v.medium_kind = 0;
    v.absorption = v.scattering = 0.0;
    if (!s->media.empty()) {   // only media[0] is used (src/renderer.rs:190)
        v.medium_kind = s->media[0].kind;
        v.absorption = s->media[0].absorption;
        v.scattering = s->media[0].scattering;
        const D3 lo = s->media[0].kind == 1 ? hex_color(0x0000FF) : hex_color(0xD2B48C), hi = s->media[0].kind == 1 ? hex_color(0xFF0000) : lo;
        s->medium_color64[0] = lo.x; s->medium_color64[1] = lo.y; s->medium_color64[2] = lo.z;
        s->medium_color_hi64[0] = hi.x; s->medium_color_hi64[1] = hi.y; s->medium_color_hi64[2] = hi.z;
    }
    for (int k = 0; k < 3; k++) v.env[k] = s->env[k];
    return RPT_OK;
}
// The fp64 kernels' common arguments: the scene, and -- for the camera passes -- camera, frame, tiles and chunking as prepare_render
// laid them out for the fp32 machinery (`a`); cam / prm / a are null for a launch without a camera (photon shooting).
extern "C++" void rpti::fill_args64(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, const RenderArgs* a, rpt64::Args& q) {
    q.sc = s->view64;
    for (int i = 0; i < 3; i++) {
        q.medium_color[i] = s->medium_color64[i];
        q.medium_color_hi[i] = s->medium_color_hi64[i];
    }
    q.cull = uint32_t(s->opt.f64_cull);
    q.surf_batch = uint32_t(s->opt.f64_surf_batch);
    q.group_lights = 0u;
    for (const auto& l : s->lights)
        if (l.kind == int(L_OBJECT) && l.obj.shape.d.kind == RPT_SHAPE_GROUP) q.group_lights = 1u;
    if (cam) {
        const D3 dir = d3(cam->direction), up = d3(cam->up);
        const D3 right = normalize(cross(dir, up));   // src/camera.rs:67-68
        for (int i = 0; i < 3; i++) {
            q.cam.eye[i] = cam->eye[i];
            q.cam.direction[i] = cam->direction[i];
            q.cam.up[i] = cam->up[i];
            q.cam.right[i] = comp(right, i);
        }
        q.cam.d = 1.0 / std::tan(cam->fov / 2.0);
        q.cam.aperture = cam->aperture;
        q.cam.focal_distance = cam->focal_distance;
    }
    if (prm) {
        q.width = prm->width; q.height = prm->height;
        q.max_bounces = prm->max_bounces;
        q.dim = double(std::max(prm->width, prm->height));
    }
    if (a) {
        q.iterations = a->iterations; q.sample_offset = a->sample_offset;
        q.n_owned = a->n_owned; q.tiles_x = a->tiles_x; q.tiles = a->tiles; q.n_items = a->n_items;
        q.chunk_spp = a->chunk_spp; q.n_chunks = a->n_chunks;
        q.pull_batch = a->pull_batch;
        q.seed_mixed = a->seed_mixed;
        q.queue = a->queue;
        q.slab = reinterpret_cast<double*>(a->slab);
    }
    q.counters = nullptr;
}
// Renderer::sample in the reference-epsilon mode: the fp32 path's launch scheme (persistent grid over (pixel, chunk)
// items, one launch set per stream) with an fp64 slab.
static int run_render64(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint32_t iterations, uint64_t seed,
                        uint32_t sample_offset, double* d_out, hipStream_t st) {
    RenderArgs a{};   // (tiles, sharding, chunking, launch sets and argument checks are shared with the fp32 path)
    int rc = rpti::prepare_render(s, st, cam, prm, iterations, seed, sample_offset, a, 0, 0, 32);
    if (rc) return rc;
    rpt64::Args q{};
    rpti::fill_args64(s, cam, prm, &a, q);
    q.counters = a.counters;
    int bpc = int(s->opt.blocks_per_cu);
    if (bpc <= 0) {
        HIP_TRY(render_f64_occupancy(q.sc.has_medium != 0, &bpc));
        if (bpc < 1) bpc = 1;
    }
    rc = rpti::run_persistent(s, prm, a, d_out, st, bpc,
                              [&q](const RenderArgs&, int nb, hipStream_t stream) { return launch_render_f64(q, nb, stream); }, true, false,
                              [&q](double scale, double* out, hipStream_t stream) { return launch_resolve_f64(q, scale, out, stream); });
    if (rc) return rc;
    if (q.counters) {
        HIP_TRY(hipStreamSynchronize(st));
        std::memset(s->last_counters, 0, sizeof(s->last_counters));
        HIP_TRY(hipMemcpy(s->last_counters64, q.counters, sizeof(s->last_counters64), hipMemcpyDeviceToHost));
        s->last_counters[0] = s->last_counters64[6];   // rpt_get_counters: samples, rays, vertices
        s->last_counters[1] = s->last_counters64[0];
        s->last_counters[2] = s->last_counters64[7];
        s->last_counters[3] = s->last_counters64[10];   // wave-level loop trips
        for (int i = 0; i < 48; i++) s->last_counters[8 + i] = s->last_counters64[16 + i];   // rpt_debug_section_counters
    }
    return RPT_OK;
}

int rpt_scene_commit(rpt_scene* s, int device) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (s->committed) return fail(RPT_ERR_STATE, "scene already committed");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(RPT_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RPT_ERR_UNSUPPORTED, std::string("built for gfx950 (MI355X), device is ") + prop.gcnArchName);
    s->n_cus = prop.multiProcessorCount;

    if (s->opt.epsilon_policy == 1)
        if (int rc64 = check_scene64(s)) return rc64;   // (before anything is allocated on the device)
    Flattener f(s);
    int rc = f.flatten_objects();
    if (rc) return rc;
    f.flatten_lights();
    f.fold_shell();
    f.mark_twin_ranges();
    rc = f.build_scene_tree();
    if (rc) return rc;
    f.collect_scan_boxes();
    rc = f.upload(device);
    if (rc) return rc;
    if (s->opt.epsilon_policy == 1) {
        rc = build_scene64(s);
        if (rc) { release_device(s); return rc; }
    }
    return RPT_OK;
}

// ---------------------------------------------------------------------------- render
// Tile ownership (host only, no HIP call): 32x32 tiles, tile (tx,ty) belongs to rank (tx+ty) % count.
extern "C" int64_t rpt_shard_tiles(uint32_t width, uint32_t height, uint32_t shard_rank, uint32_t shard_count,
                                   uint32_t* tiles_out, uint64_t capacity) {
    if (shard_count == 0) shard_count = 1;
    if (width == 0 || height == 0 || shard_rank >= shard_count) return fail(RPT_ERR_INVALID, "bad shard arguments");
    uint32_t tiles_x = (width + 31) / 32, tiles_y = (height + 31) / 32;
    int64_t n = 0;
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++)
            if ((tx + ty) % shard_count == shard_rank) {
                if (tiles_out && uint64_t(n) < capacity) tiles_out[n] = ty * tiles_x + tx;
                n++;
            }
    return n;
}

// Samples per work item.  Small items keep the persistent grid's tail short when a GPU owns only 1/8 of the
// tiles; the value depends on `iterations` alone so that the fp32 partial sums, and hence the image bits, do not
// change with the shard count.  At most 16 chunks per pixel (C3: 16 samples per item, 0.27 GB of partial sums per launch instead of the
// 1.07 GB of 4-sample items; measured with consecutive steps on two streams (tools/chunk_pipelined.py): 25.46 / 25.26 / 25.02 / 24.86 ms per
// step for 4 / 8 / 16 / 32 on the whole frame, 3.31 / 3.34 / 3.40 / 3.51 ms on one rank's shard of an 8-GPU job: an item is also a trip
// through the work-pull code for the whole wave, which some lane needs in 95 % of the trips with 8-sample items and 77 % with 16).
static uint32_t chunk_rule(int64_t opt_chunk_spp, uint32_t iterations, uint32_t min_chunk, uint32_t fixed_chunk) {
    if (fixed_chunk) return fixed_chunk;
    const int64_t chunk = opt_chunk_spp > 0 ? opt_chunk_spp
                                              : std::min<int64_t>(32, std::max<int64_t>(std::max<int64_t>(2, min_chunk), (int64_t(iterations) + 15) / 16));
    return uint32_t(std::min<int64_t>(chunk, iterations));
}
int rpt_render_chunking(uint32_t iterations, uint32_t* chunk_spp, uint32_t* n_chunks) {
    if (iterations == 0) return fail(RPT_ERR_INVALID, "empty render");
    int64_t opt;
    {
        std::lock_guard<std::mutex> lock(g_defaults_mutex);
        opt = g_defaults.chunk_spp;
    }
    const uint32_t c = chunk_rule(opt, iterations, 0, 0);
    if (chunk_spp) *chunk_spp = c;
    if (n_chunks) *n_chunks = (iterations + c - 1) / c;
    return RPT_OK;
}

int rpt_scene_render_chunking(rpt_scene* s, uint32_t iterations, uint32_t* chunk_spp, uint32_t* n_chunks) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (iterations == 0) return fail(RPT_ERR_INVALID, "empty render");
    const uint32_t c = chunk_rule(s->opt.chunk_spp, iterations, 0, 0);
    if (chunk_spp) *chunk_spp = c;
    if (n_chunks) *n_chunks = (iterations + c - 1) / c;
    return RPT_OK;
}

extern "C++" int rpti::prepare_render(rpt_scene* s, hipStream_t st, const rpt_camera* cam, const rpt_render_params* prm, uint32_t iterations,
                         uint64_t seed, uint32_t sample_offset, RenderArgs& a, uint32_t min_chunk, uint32_t fixed_chunk, uint32_t slab_item_bytes) {
    if (!s || !cam || !prm) return fail(RPT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RPT_ERR_STATE, "rpt_scene_commit must be called before rendering");
    if (prm->width == 0 || prm->height == 0 || iterations == 0) return fail(RPT_ERR_INVALID, "empty render");
    if (uint64_t(prm->width) * prm->height > (1ull << 31)) return fail(RPT_ERR_INVALID, "image too large");
    uint32_t shard_count = prm->shard_count == 0 ? 1 : prm->shard_count;
    if (prm->shard_rank >= shard_count) return fail(RPT_ERR_INVALID, "shard_rank >= shard_count");
    HIP_TRY(hipSetDevice(s->device));

    a.sc = s->view;
    // Camera::cast_ray constants (src/camera.rs:66-69), fp64 then rounded once
    D3 dir = d3(cam->direction), up = d3(cam->up);
    double dd = 1.0 / std::tan(cam->fov / 2.0);
    D3 right = normalize(cross(dir, up));
    for (int i = 0; i < 3; i++) {
        a.cam.eye[i] = float(cam->eye[i]);
        a.cam.ddir[i] = float(dd * cam->direction[i]);
        a.cam.right[i] = float(comp(right, i));
        a.cam.up[i] = float(cam->up[i]);
    }
    a.cam.aperture = float(cam->aperture);
    a.cam.focal_distance = float(cam->focal_distance);
    a.width = prm->width;
    a.height = prm->height;
    a.inv_dim = float(1.0 / double(std::max(prm->width, prm->height)));
    a.max_bounces = prm->max_bounces;
    a.iterations = iterations;
    a.sample_offset = sample_offset;
    a.chunk_spp = chunk_rule(s->opt.chunk_spp, iterations, min_chunk, fixed_chunk);
    a.n_chunks = (iterations + a.chunk_spp - 1) / a.chunk_spp;
    a.seed_mixed = seed_mix(seed);

    // owned tiles
    uint32_t tiles_x = (prm->width + 31) / 32, tiles_y = (prm->height + 31) / 32;
    (void)tiles_y;
    if (s->tk_w != prm->width || s->tk_h != prm->height || s->tk_rank != prm->shard_rank || s->tk_count != shard_count ||
        !s->d_tiles) {
        std::vector<uint32_t> tiles(size_t(tiles_x) * tiles_y);
        tiles.resize(size_t(rpt_shard_tiles(prm->width, prm->height, prm->shard_rank, shard_count, tiles.data(),
                                            tiles.size())));
        if (tiles.size() > s->tiles_cap) {
            if (s->d_tiles) HIP_TRY(hipFree(s->d_tiles));
            HIP_TRY(hipMalloc((void**)&s->d_tiles, std::max<size_t>(tiles.size(), 1) * 4));
            s->tiles_cap = tiles.size();
        }
        if (!tiles.empty()) HIP_TRY(hipMemcpy(s->d_tiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
        s->tk_w = prm->width; s->tk_h = prm->height; s->tk_rank = prm->shard_rank; s->tk_count = shard_count;
        s->n_tiles = uint32_t(tiles.size());
        s->tiles_x = tiles_x;
    }
    a.tiles = s->d_tiles;
    a.n_tiles = s->n_tiles;
    a.tiles_x = s->tiles_x;
    a.n_owned = a.n_tiles * 1024u;
    uint64_t n_items = uint64_t(a.n_owned) * a.n_chunks;
    if (n_items >= (1ull << 32) - (1ull << 24)) return fail(RPT_ERR_INVALID, "too many work items; raise chunk_spp");
    a.n_items = uint32_t(n_items);
    size_t slab_bytes = std::max<size_t>(size_t(n_items) * slab_item_bytes, 16);
    // the launch set: the one this stream used last, else the other one
    if (s->sets[s->cur_set].used && s->sets[s->cur_set].stream != st) s->cur_set ^= 1;
    rpt_scene::LaunchSet& ls = s->sets[s->cur_set];
    if (ls.used && ls.stream != st) HIP_TRY(hipStreamWaitEvent(st, ls.done, 0));  // a third stream: wait for the set's last launch
    if (slab_bytes > ls.slab_cap) {
        if (ls.d_slab) HIP_TRY(hipFree(ls.d_slab));
        HIP_TRY(hipMalloc((void**)&ls.d_slab, slab_bytes));
        ls.slab_cap = slab_bytes;
    }
    ls.stream = st;
    ls.used = true;
    a.slab = ls.d_slab;
    a.queue = ls.d_queue;
    a.counters = s->opt.counters ? s->d_counters : nullptr;
    a.lds_stack = s->view.n_nodes ? 1u : 0u;
    a.defer_lanes = uint32_t(s->opt.defer_lanes);
    a.defer_stop = uint32_t(std::min(s->opt.defer_stop, s->opt.defer_lanes));
    a.walk_leaf_quarters = uint32_t(s->opt.walk_leaf_quarters);
    // Detached shadow queries: in a medium (a path's radiance is linear in its light terms), per-mesh trees, and every
    // light that can be visible has its twin among the scanned records as one range of hit codes.
    a.detach = 0;
    if (s->opt.detach_shadows && s->view.has_medium && bvh_mode(s->view) == 1 && s->view.n_lparts == 0 && s->twins_scanned)
        a.detach = (s->opt.detach_shadows == 2 && s->n_twin_lights <= 3) ? 2u : 1u;
    a.stream_backlog = uint32_t(s->opt.stream_backlog);
    a.stream_contexts = uint32_t(s->opt.stream_contexts);
    a.n_twin_lights = s->n_twin_lights;
    a.stream_scratch = nullptr;   // (run_render sizes it for the grid)
    a.pull_batch = uint32_t(s->opt.pull_batch);
    a.detach_trigger = uint32_t(s->opt.detach_trigger);
    if (a.detach) {
        a.defer_lanes = uint32_t(s->opt.detach_lanes);
        a.defer_stop = uint32_t(std::min<int64_t>(s->opt.defer_stop, std::min(s->opt.detach_lanes, s->opt.detach_trigger)));   // a session that starts makes progress
    }
    return RPT_OK;
}

// A launch that uses per-scene scratch beyond its launch set (the photon camera pass: candidate lists, overflow
// flag) must not overlap a launch on another stream: wait for whatever the other stream still has in flight.
extern "C++" int rpti::serialize_with_other_streams(rpt_scene* s, hipStream_t st) {
    for (auto& ls : s->sets)
        if (ls.used && ls.stream != st) HIP_TRY(hipStreamWaitEvent(st, ls.done, 0));
    return RPT_OK;
}

extern "C++" int rpti::run_persistent(rpt_scene* s, const rpt_render_params* prm, const RenderArgs& a, double* d_out, hipStream_t st,
                         int blocks_per_cu, const std::function<hipError_t(const RenderArgs&, int, hipStream_t)>& launch,
                         bool indexed_start, bool wave_items, const std::function<hipError_t(double, double*, hipStream_t)>& resolve) {
    rpt_scene::LaunchSet& mine = s->sets[s->sets[0].d_queue == a.queue ? 0 : 1];
    rpt_scene::LaunchSet& other = s->sets[s->sets[0].d_queue == a.queue ? 1 : 0];
    // Two launches that become ready at the same moment would share the CUs block by block, and the half of each
    // grid that finds no room would start after everything else has drained (measured: 31.3 instead of 30.0 ms per
    // step).  So a launch on the second stream is released only once the first stream has reached its kernel: the
    // grids then follow each other, the later one filling the CUs as the blocks of the earlier one retire.
    if (other.used && other.stream != st) HIP_TRY(hipStreamWaitEvent(st, other.launched, 0));
    HIP_TRY(hipMemsetAsync(a.queue, 0, 8, st));
    if (a.counters) HIP_TRY(hipMemsetAsync(a.counters, 0, 512, st));
    uint32_t shard_count = prm->shard_count == 0 ? 1 : prm->shard_count;
    if (shard_count > 1) HIP_TRY(hipMemsetAsync(d_out, 0, size_t(prm->width) * prm->height * 24, st));
    if (a.n_items) {
        uint64_t want = wave_items ? (uint64_t(a.n_items) + 3) / 4 : (uint64_t(a.n_items) + 255) / 256;
        int n_blocks = int(std::min<uint64_t>(uint64_t(s->n_cus) * std::max(blocks_per_cu, 1), want));
        s->last_blocks = n_blocks;
        // indexed_start: every wave of the grid takes the batch with its own index first (render_kernel's work
        // pull), so the counter starts behind those batches
        if (indexed_start) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)a.queue, int(uint32_t(n_blocks) * 4u * 64u), 1, st));
        hipEvent_t* ev = nullptr;
        if (s->opt.timing) {
            const size_t slot = s->ev_count % rpt_scene::kTimedLaunches;
            while (s->evs.size() < 3 * (slot + 1)) {
                hipEvent_t e = nullptr;
                HIP_TRY(hipEventCreate(&e));
                s->evs.push_back(e);
            }
            ev = &s->evs[3 * slot];
            HIP_TRY(hipEventRecord(ev[0], st));
        }
        HIP_TRY(hipEventRecord(mine.launched, st));
        HIP_TRY(launch(a, n_blocks, st));
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        if (resolve) HIP_TRY(resolve(std::pow(2.0, prm->exposure_value), d_out, st));
        else HIP_TRY(launch_resolve(a, std::pow(2.0, prm->exposure_value), d_out, st));
        if (ev) {
            HIP_TRY(hipEventRecord(ev[2], st));
            s->ev_count++;
        }
    }
    HIP_TRY(hipEventRecord(mine.done, st));
    return RPT_OK;
}
static int run_render(rpt_scene* s, const rpt_render_params* prm, const RenderArgs& a_in, double* d_out, hipStream_t st) {
    RenderArgs a = a_in;
    int bpc = int(s->opt.blocks_per_cu);
    if (bpc <= 0) {
        HIP_TRY(render_occupancy(a.sc.has_medium != 0, bvh_mode(a.sc), &bpc, int(a.detach)));
        if (bpc < 1) bpc = 1;
    }
    if (a.detach == 2) {   // the waves' rings and parked paths: per launch set, like the slab (two launches may be in flight)
        rpt_scene::LaunchSet& ls = s->sets[s->sets[0].d_queue == a.queue ? 0 : 1];
        const size_t need = size_t(s->n_cus) * size_t(bpc) * stream_scratch_bytes_per_block();
        if (need > ls.stream_cap) {
            if (ls.d_stream) HIP_TRY(hipFree(ls.d_stream));
            ls.d_stream = nullptr; ls.stream_cap = 0;
            HIP_TRY(hipMalloc((void**)&ls.d_stream, need));
            ls.stream_cap = need;
        }
        a.stream_scratch = ls.d_stream;
    }
    return rpti::run_persistent(s, prm, a, d_out, st, bpc,
                                [](const RenderArgs& ra, int nb, hipStream_t stream) { return launch_render(ra, nb, stream); }, true);
}
extern "C++" rpti::SceneDev rpti::scene_dev(rpt_scene* s) {
    int first = -1;
    for (size_t i = 0; i < s->lights.size(); i++)
        if (s->lights[i].kind == L_OBJECT) { first = int(i); break; }
    return SceneDev{s->committed, s->device, s->n_cus, s->view, first, s->arena64 != nullptr};
}
extern "C++" void*& rpti::photon_slot(rpt_scene* s) { return s->photon; }
extern "C++" int64_t rpti::option_photon_skip(rpt_scene* s) { return s->opt.photon_skip; }
extern "C++" int64_t rpti::option_f64_photon_slice(rpt_scene* s) { return s->opt.f64_photon_slice; }
extern "C++" int64_t rpti::option_photon_block_lists(rpt_scene* s) { return s->opt.photon_block_lists; }
extern "C++" int64_t rpti::option_photon_parts(rpt_scene* s) { return s->opt.photon_parts; }
extern "C++" int64_t rpti::option_photon_split(rpt_scene* s) { return s->opt.photon_split; }
extern "C++" int64_t rpti::option_photon_coop_gather(rpt_scene* s) { return s->opt.photon_coop_gather; }
extern "C++" double* rpti::scratch_out(rpt_scene* s, size_t bytes) {
    if (bytes > s->out_cap) {
        if (s->d_out) (void)hipFree(s->d_out);
        s->d_out = nullptr;
        s->out_cap = 0;
        if (hipMalloc((void**)&s->d_out, bytes) != hipSuccess) return nullptr;
        s->out_cap = bytes;
    }
    return s->d_out;
}
extern "C++" int rpti::fetch_counters(rpt_scene* s, const RenderArgs& a) {
    std::memset(s->last_counters, 0, sizeof(s->last_counters));
    if (a.counters) {
        HIP_TRY(hipMemcpy(s->last_counters, a.counters, 512, hipMemcpyDeviceToHost));
        s->last_counters[4] = s->last_counters[1] * s->prims_per_ray;
    }
    return RPT_OK;
}

int rpt_render_sample_device(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint32_t iterations,
                             uint64_t seed, uint32_t sample_offset, void* d_out_rgb, void* hip_stream) {
    if (!d_out_rgb) return fail(RPT_ERR_INVALID, "null output");
    if (s && s->arena64) return run_render64(s, cam, prm, iterations, seed, sample_offset, static_cast<double*>(d_out_rgb), static_cast<hipStream_t>(hip_stream));
    RenderArgs a{};
    int rc = rpti::prepare_render(s, static_cast<hipStream_t>(hip_stream), cam, prm, iterations, seed, sample_offset, a);
    if (rc) return rc;
    rc = run_render(s, prm, a, static_cast<double*>(d_out_rgb), static_cast<hipStream_t>(hip_stream));
    if (rc) return rc;
    if (a.counters) {
        HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
        return rpti::fetch_counters(s, a);
    }
    return RPT_OK;
}

int rpt_render_sample(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint32_t iterations,
                      uint64_t seed, uint32_t sample_offset, double* out_rgb) {
    if (!out_rgb) return fail(RPT_ERR_INVALID, "null output");
    RenderArgs a{};
    int rc = rpti::prepare_render(s, nullptr, cam, prm, iterations, seed, sample_offset, a);
    if (rc) return rc;
    size_t bytes = size_t(prm->width) * prm->height * 24;
    if (bytes > s->out_cap) {
        if (s->d_out) HIP_TRY(hipFree(s->d_out));
        HIP_TRY(hipMalloc((void**)&s->d_out, bytes));
        s->out_cap = bytes;
    }
    if (s->arena64) {
        rc = run_render64(s, cam, prm, iterations, seed, sample_offset, s->d_out, nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(out_rgb, s->d_out, bytes, hipMemcpyDeviceToHost));
        return RPT_OK;
    }
    rc = run_render(s, prm, a, s->d_out, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_rgb, s->d_out, bytes, hipMemcpyDeviceToHost));
    return rpti::fetch_counters(s, a);
}
int rpt_debug_epsilon_counters(rpt_scene* s, uint64_t out[12]) {
    if (!s || !out) return fail(RPT_ERR_INVALID, "null argument");
    if (!s->arena64) return fail(RPT_ERR_STATE, "the scene was not committed with epsilon_policy = 1");
    for (int i = 0; i < 12; i++) out[i] = s->last_counters64[i];
    return RPT_OK;
}

// ---------------------------------------------------------------------------- Buffer on the device
struct rpt_buffer {
    int device = 0;
    uint32_t width = 0, height = 0, radius = 0, n_batches = 0;
    double *d_sum = nullptr, *d_sumsq = nullptr, *d_stage = nullptr;  // stage: one batch / per-pixel variances
    uint8_t* d_img = nullptr;
};
rpt_buffer* rpt_buffer_create(int device, uint32_t width, uint32_t height, uint32_t filter_radius) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { fail(RPT_ERR_INVALID, "device index out of range"); return nullptr; }
    if (width == 0 || height == 0 || uint64_t(width) * height >= (1ull << 31)) { fail(RPT_ERR_INVALID, "bad buffer size"); return nullptr; }
    auto* b = new rpt_buffer();
    b->device = device;
    b->width = width;
    b->height = height;
    b->radius = filter_radius;
    const size_t n = size_t(width) * height;
    bool ok = hipSetDevice(device) == hipSuccess && hipMalloc((void**)&b->d_sum, n * 24) == hipSuccess &&
              hipMalloc((void**)&b->d_sumsq, n * 8) == hipSuccess && hipMalloc((void**)&b->d_stage, n * 24) == hipSuccess &&
              hipMalloc((void**)&b->d_img, n * 3) == hipSuccess && hipMemset(b->d_sum, 0, n * 24) == hipSuccess &&
              hipMemset(b->d_sumsq, 0, n * 8) == hipSuccess;
    if (!ok) {
        fail(RPT_ERR_DEVICE, "rpt_buffer_create: device allocation failed");
        rpt_buffer_destroy(b);
        return nullptr;
    }
    return b;
}
void rpt_buffer_destroy(rpt_buffer* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)hipFree(b->d_sum); (void)hipFree(b->d_sumsq); (void)hipFree(b->d_stage); (void)hipFree(b->d_img);
    delete b;
}
int rpt_buffer_add_samples_device(rpt_buffer* b, const void* d_rgb, void* hip_stream) {
    if (!b || !d_rgb) return fail(RPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(launch_buffer_add(b->width * b->height, static_cast<const double*>(d_rgb), b->d_sum, b->d_sumsq,
                              static_cast<hipStream_t>(hip_stream)));
    b->n_batches++;
    return RPT_OK;
}
int rpt_buffer_add_samples(rpt_buffer* b, const double* rgb) {
    if (!b || !rgb) return fail(RPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(b->d_stage, rgb, size_t(b->width) * b->height * 24, hipMemcpyHostToDevice));
    return rpt_buffer_add_samples_device(b, b->d_stage, nullptr);
}
int rpt_buffer_image(rpt_buffer* b, uint8_t* out_rgb8) {
    if (!b || !out_rgb8) return fail(RPT_ERR_INVALID, "null argument");
    if (b->n_batches == 0) return fail(RPT_ERR_STATE, "Pixel found with no samples");  // the reference's assert (buffer.rs:89)
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipDeviceSynchronize());  // batches may have been added on other streams
    HIP_TRY(launch_buffer_image(b->width, b->height, b->radius, b->n_batches, b->d_sum, b->d_img, nullptr));
    HIP_TRY(hipMemcpy(out_rgb8, b->d_img, size_t(b->width) * b->height * 3, hipMemcpyDeviceToHost));
    return RPT_OK;
}
int rpt_buffer_variance(rpt_buffer* b, double* out) {
    if (!b || !out) return fail(RPT_ERR_INVALID, "null argument");
    if (b->n_batches < 2) { *out = std::numeric_limits<double>::quiet_NaN(); return RPT_OK; }  // 0/0 in the reference too
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = size_t(b->width) * b->height;
    HIP_TRY(launch_buffer_variance(uint32_t(n), b->n_batches, b->d_sum, b->d_sumsq, b->d_stage, nullptr));
    std::vector<double> v(n);
    HIP_TRY(hipMemcpy(v.data(), b->d_stage, n * 8, hipMemcpyDeviceToHost));
    double acc = 0.0;
    for (double x : v) acc += x;  // pixel order, as buffer.rs:63-72
    *out = acc / double(n);
    return RPT_OK;
}
int rpt_buffer_batches(rpt_buffer* b, uint32_t* n) {
    if (!b || !n) return fail(RPT_ERR_INVALID, "null argument");
    *n = b->n_batches;
    return RPT_OK;
}
// Renderer::sample(&self, iterations, &mut Buffer), src/renderer.rs:158-171: the frame never leaves the device.
int rpt_render_into_buffer(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint32_t iterations,
                           uint64_t seed, uint32_t sample_offset, rpt_buffer* b) {
    if (!b) return fail(RPT_ERR_INVALID, "null buffer");
    if (!s || !prm) return fail(RPT_ERR_INVALID, "null argument");
    if (prm->width != b->width || prm->height != b->height) return fail(RPT_ERR_INVALID, "Invalid sample dimension");  // buffer.rs:33-36
    if (s->committed && s->device != b->device) return fail(RPT_ERR_INVALID, "buffer and scene live on different devices");
    if (s->arena64) {
        int rc64 = run_render64(s, cam, prm, iterations, seed, sample_offset, b->d_stage, nullptr);
        return rc64 ? rc64 : rpt_buffer_add_samples_device(b, b->d_stage, nullptr);
    }
    RenderArgs a{};
    int rc = rpti::prepare_render(s, nullptr, cam, prm, iterations, seed, sample_offset, a);
    if (rc) return rc;
    if (prm->shard_count > 1) HIP_TRY(hipMemsetAsync(b->d_stage, 0, size_t(b->width) * b->height * 24, nullptr));
    rc = run_render(s, prm, a, b->d_stage, nullptr);
    if (rc) return rc;
    rc = rpt_buffer_add_samples_device(b, b->d_stage, nullptr);
    if (rc) return rc;
    if (a.counters) {
        HIP_TRY(hipStreamSynchronize(nullptr));
        return rpti::fetch_counters(s, a);
    }
    return RPT_OK;
}

int rpt_get_timing(rpt_scene* s, double* render_ms, double* resolve_ms, int32_t* grid_blocks) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (!s->ev_count) return fail(RPT_ERR_STATE, "no timed render: rpt_set_option(\"timing\", 1) first");
    const hipEvent_t* ev = &s->evs[3 * ((s->ev_count - 1) % rpt_scene::kTimedLaunches)];
    HIP_TRY(hipEventSynchronize(ev[2]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
    if (render_ms) *render_ms = a;
    if (resolve_ms) *resolve_ms = b;
    if (grid_blocks) *grid_blocks = s->last_blocks;
    return RPT_OK;
}

int rpt_get_timing_mean(rpt_scene* s, double* render_ms, double* resolve_ms, int32_t* launches) {
    if (!s) return fail(RPT_ERR_INVALID, "null scene");
    if (!s->ev_count) return fail(RPT_ERR_STATE, "no timed render: rpt_set_option(\"timing\", 1) first");
    const size_t n = std::min(s->ev_count, rpt_scene::kTimedLaunches);  // the ring keeps the latest launches
    HIP_TRY(hipEventSynchronize(s->evs[3 * ((s->ev_count - 1) % rpt_scene::kTimedLaunches) + 2]));
    double a = 0.0, b = 0.0;
    for (size_t i = 0; i < n; i++) {
        const hipEvent_t* ev = &s->evs[3 * i];
        float x = 0.f, y = 0.f;
        HIP_TRY(hipEventElapsedTime(&x, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&y, ev[1], ev[2]));
        a += x;
        b += y;
    }
    if (render_ms) *render_ms = a / double(n);
    if (resolve_ms) *resolve_ms = b / double(n);
    if (launches) *launches = int32_t(n);
    s->ev_count = 0;
    return RPT_OK;
}

int rpt_scene_stats(rpt_scene* s, uint64_t out[16]) {
    if (!s || !out) return fail(RPT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RPT_ERR_STATE, "rpt_scene_commit must be called first");
    std::memcpy(out, s->stats, sizeof(s->stats));
    return RPT_OK;
}

int rpt_get_counters(rpt_scene* s, uint64_t out[8]) {
    if (!s || !out) return fail(RPT_ERR_INVALID, "null argument");
    std::memcpy(out, s->last_counters, 64);
    return RPT_OK;
}
int rpt_debug_section_counters(rpt_scene* s, uint64_t out[56]) {
    if (!s || !out) return fail(RPT_ERR_INVALID, "null argument");
    std::memcpy(out, s->last_counters + 8, 56 * 8);
    return RPT_OK;
}

// ---------------------------------------------------------------------------- test hooks
int rpt_intersect_batch(rpt_scene* s, uint64_t n, const float* origins, const float* dirs, float* t, int32_t* object,
                        float* normal) {
    if (!s || !origins || !dirs || !t || !object) return fail(RPT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RPT_ERR_STATE, "rpt_scene_commit must be called first");
    if (n == 0) return RPT_OK;
    HIP_TRY(hipSetDevice(s->device));
    TmpDev tmp;
    float *d_o, *d_d, *d_t, *d_n;
    int32_t* d_obj;
    HIP_TRY(tmp.alloc(&d_o, n * 3));
    HIP_TRY(tmp.alloc(&d_d, n * 3));
    HIP_TRY(tmp.alloc(&d_t, n));
    HIP_TRY(tmp.alloc(&d_n, n * 3));
    HIP_TRY(tmp.alloc(&d_obj, n));
    HIP_TRY(hipMemcpy(d_o, origins, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, dirs, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(launch_intersect(s->view, n, d_o, d_d, d_t, d_obj, d_n, s->view.n_nodes != 0, nullptr));
    HIP_TRY(hipMemcpy(t, d_t, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(object, d_obj, n * 4, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(hipMemcpy(normal, d_n, n * 12, hipMemcpyDeviceToHost));
    return RPT_OK;
}

int rpt_debug_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out) {
    if (!out) return fail(RPT_ERR_INVALID, "null argument");
    TmpDev tmp;
    uint32_t* d;
    HIP_TRY(tmp.alloc(&d, n));
    HIP_TRY(launch_debug_rng(seed_mix(seed), pixel, sample, n, d, nullptr));
    HIP_TRY(hipMemcpy(out, d, size_t(n) * 4, hipMemcpyDeviceToHost));
    return RPT_OK;
}
static Material to_gpu_material(const rpt_material* m) {
    Material g;
    g.albedo_emit = F4{float(m->albedo[0]), float(m->albedo[1]), float(m->albedo[2]), float(m->emittance)};
    g.params = F4{bits_f(uint32_t(m->kind)), float(m->shininess), float(m->ior), 0.f};
    return g;
}
int rpt_debug_material_sample_f(const rpt_material* m, uint64_t n, const float* normals, const float* wos, uint64_t seed,
                                float* wi, float* pdf, int32_t* some) {
    std::string why;
    if (!check_material(m, why) || !normals || !wos || !wi || !pdf || !some) return fail(RPT_ERR_INVALID, "bad argument");
    TmpDev tmp;
    float *d_n, *d_wo, *d_wi, *d_pdf;
    int32_t* d_some;
    HIP_TRY(tmp.alloc(&d_n, n * 3));
    HIP_TRY(tmp.alloc(&d_wo, n * 3));
    HIP_TRY(tmp.alloc(&d_wi, n * 3));
    HIP_TRY(tmp.alloc(&d_pdf, n));
    HIP_TRY(tmp.alloc(&d_some, n));
    HIP_TRY(hipMemcpy(d_n, normals, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_wo, wos, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(launch_debug_sample_f(to_gpu_material(m), n, d_n, d_wo, seed_mix(seed), d_wi, d_pdf, d_some, nullptr));
    HIP_TRY(hipMemcpy(wi, d_wi, n * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf, d_pdf, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(some, d_some, n * 4, hipMemcpyDeviceToHost));
    return RPT_OK;
}
int rpt_debug_material_bsdf(const rpt_material* m, uint64_t n, const float* normals, const float* wos, const float* wis,
                            float* out_rgb) {
    std::string why;
    if (!check_material(m, why) || !normals || !wos || !wis || !out_rgb) return fail(RPT_ERR_INVALID, "bad argument");
    TmpDev tmp;
    float *d_n, *d_wo, *d_wi, *d_out;
    HIP_TRY(tmp.alloc(&d_n, n * 3));
    HIP_TRY(tmp.alloc(&d_wo, n * 3));
    HIP_TRY(tmp.alloc(&d_wi, n * 3));
    HIP_TRY(tmp.alloc(&d_out, n * 3));
    HIP_TRY(hipMemcpy(d_n, normals, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_wo, wos, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_wi, wis, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(launch_debug_bsdf(to_gpu_material(m), n, d_n, d_wo, d_wi, d_out, nullptr));
    HIP_TRY(hipMemcpy(out_rgb, d_out, n * 12, hipMemcpyDeviceToHost));
    return RPT_OK;
}
int rpt_debug_camera_rays(const rpt_camera* cam, const rpt_render_params* prm, uint64_t seed, uint32_t sample,
                          float* origins, float* dirs) {
    if (!cam || !prm || !origins || !dirs) return fail(RPT_ERR_INVALID, "null argument");
    CameraG c;
    D3 dir = d3(cam->direction), up = d3(cam->up);
    double dd = 1.0 / std::tan(cam->fov / 2.0);
    D3 right = normalize(cross(dir, up));
    for (int i = 0; i < 3; i++) {
        c.eye[i] = float(cam->eye[i]);
        c.ddir[i] = float(dd * cam->direction[i]);
        c.right[i] = float(comp(right, i));
        c.up[i] = float(cam->up[i]);
    }
    c.aperture = float(cam->aperture);
    c.focal_distance = float(cam->focal_distance);
    size_t n = size_t(prm->width) * prm->height;
    TmpDev tmp;
    float *d_o, *d_d;
    HIP_TRY(tmp.alloc(&d_o, n * 3));
    HIP_TRY(tmp.alloc(&d_d, n * 3));
    HIP_TRY(launch_debug_camera(c, prm->width, prm->height, seed_mix(seed), sample, d_o, d_d, nullptr));
    HIP_TRY(hipMemcpy(origins, d_o, n * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dirs, d_d, n * 12, hipMemcpyDeviceToHost));
    return RPT_OK;
}

}  // extern "C"
