// photon.hip — photon mapping on gfx950 (reference src/photon.rs):
//   * photon shooting (shoot_photon / trace_photon, :724-946) as a two-pass kernel (count, then
//     write at prefix-summed offsets, so the photon arrays are deterministic);
//   * point maps as LBVHs built on the device: 63-bit Morton codes, hipCUB radix sort, Karras'
//     parallel construction, bottom-up refit, packed into the same two-box 64-byte nodes the mesh
//     BVH uses.  They replace the kd-tree 0.4.1 / bvh 0.6 crates: `nearests(q, k)` -> k nearest by
//     squared distance; `traverse(ray)` -> every sphere whose box the ray hits (filtered below);
//   * the per-photon gather radius = distance to the 10th nearest volume photon (:214-232);
//   * the camera pass (get_color_with_photon_map / estimate_indirect, :316-628, :950-985) as a
//     persistent kernel sharing the path tracer's work queue, slab and resolve.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <type_traits>
#include <vector>

#include "device_core.h"
#include "f64_layout.h"
#include "host_internal.h"
#include "kernels.h"
#include "sort_scan.h"

namespace rptg {

struct alignas(16) PhotonRec {
    F4 pos_r;  // position, gather radius (volume photons of the beam map)
    F4 dir;    // direction toward where the photon came from (wo); w = original index (bits)
    F4 pow;    // power
};

// Photon-tree leaf entry: BVH_LEAF | (count - 1) << 26 | first photon (sorted order).  The trees the beam
// walkers use are packed with one photon per leaf; the k-NN trees with up to kKnnLeaf.
static_assert(sizeof(PhotonRec) == sizeof(rpt64::PhotonRec32) && sizeof(PhotonRec) == RPT_PHOTON_RECORD_BYTES, "one record layout");
static constexpr uint32_t PH_LEAF_INDEX = 0x03FFFFFFu;
static constexpr uint32_t kKnnLeaf = 8;

RPT_DEV float ray_tmin_p(V o) { return 2e-5f * (1.f + max3(fabsf(o.x), fabsf(o.y), fabsf(o.z))); }

struct ShootArgs {
    SceneView sc;
    uint64_t n_photons, seed_mixed;  // photons of THIS launch
    uint64_t first_photon;           // global index of its first photon (the RNG stream key)
    float power;  // watts / photon_count (of the whole map)
    uint32_t light_index;
    uint32_t kind;  // RPT_PHOTON_*: beam-beam thins the volume photons and records each beam's start
    uint32_t* cnt_s;
    uint32_t* cnt_v;
    const uint32_t* off_s;  // exclusive prefix sums of cnt_s / cnt_v (record totals are < 2^26)
    const uint32_t* off_v;
    PhotonRec* surf;
    PhotonRec* vol;
};

static_assert(offsetof(ShootArgs, sc) == 0, "kernel arguments begin with the SceneView (kernarg_scene)");
// shoot_photon + trace_photon, src/photon.rs:724-946.  WRITE = false: count only.
template <bool MEDIUM, bool BVH, bool WRITE>
__global__ __launch_bounds__(256) void photon_shoot_kernel(const ShootArgs a) {
    extern __shared__ uint32_t dyn_lds[];
    const SceneView& sc = a.sc;
    uint32_t* stk = BVH ? (dyn_lds + threadIdx.x) : nullptr;
    const float sigma_t = sc.sigma_a + sc.sigma_s;
    const float inv_sigma_t = MEDIUM ? 1.f / sigma_t : 0.f;
    const float albedo_med = MEDIUM ? sc.sigma_s / sigma_t : 0.f;
    const Light L = uload(&sc.lights[a.light_index]);
    for (uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x; i < a.n_photons; i += uint64_t(gridDim.x) * 256u) {
        Rng rng, thin;
        const uint64_t g = a.first_photon + i;
        rng.seed(a.seed_mixed, uint32_t(g), 0x80000000u + uint32_t(g >> 32));
        thin.seed(a.seed_mixed, uint32_t(g), 0xC0000000u + uint32_t(g >> 32));  // thinning side stream
        const bool beams = a.kind == RPT_PHOTON_BEAM_BEAM;
        V ro, n0;
        float p0;
        sample_light_shape<true>(sc, L, mk(0.f, 0.f, 0.f), rng, ro, n0, p0);  // :733-734 (target is a dummy)
        float u1 = rng.uniform(), u2 = rng.uniform();
        float ct = 1.f - u2;                                           // theta = acos(1 - u), :738
        float st = sqrt1(fmaxf(1.f - ct * ct, 0.f));
        V rd = rotate_from_y(n0, mk(st * __builtin_amdgcn_cosf(u1), ct, st * __builtin_amdgcn_sinf(u1)), true);
        V power = a.power * xyz(L.albedo);
        uint32_t ns = 0, nv = 0, c0 = 0, c1 = 0;
        uint64_t os = WRITE ? a.off_s[i] : 0, ov = WRITE ? a.off_v[i] : 0;
        for (;;) {
            const V wo = -normalize(rd);
            const float tmin = ray_tmin_p(ro);
            float t = kInf;
            uint32_t code = CODE_MISS, inst = 0;
            closest_hit<(BVH ? 2 : 0), false>(sc, ro, rd, tmin, t, code, inst, stk, 256, c0, c1);
            const bool hit = code != CODE_MISS;
            bool in_volume = false;
            float d = 0.f;
            if (MEDIUM) {
                float xi = rng.range(0.f, 1.f);
                d = -__logf(xi) * inv_sigma_t;
                in_volume = !hit || d < t;
            } else if (!hit) {
                break;
            }
            if (in_volume) {  // trace_in_volume :879-914
                V x = fma3(d, rd, ro);
                bool hi = sc.medium_kind == 1u && x.y > 250.f;
                V mcol = hi ? mk(sc.medium_color_hi[0], sc.medium_color_hi[1], sc.medium_color_hi[2])
                            : mk(sc.medium_color[0], sc.medium_color[1], sc.medium_color[2]);
                // beam-beam map: keep 0.1 % of the volume photons at 1000x power (src/photon.rs:779-787)
                const bool keep = !beams || thin.uniform() < 0.001f;
                if (keep) {
                    if (WRITE) {
                        const float boost = beams ? 1000.f : 1.f;
                        PhotonRec r;
                        r.pos_r = F4{x.x, x.y, x.z, beams ? 3.f : 0.f};          // beams: fixed radius 3 (:280)
                        r.dir = beams ? F4{ro.x, ro.y, ro.z, 0.f} : F4{wo.x, wo.y, wo.z, 0.f};  // beams: start of the beam
                        r.pow = F4{boost * power.x, boost * power.y, boost * power.z, 0.f};
                        a.vol[ov + nv] = r;
                    }
                    nv++;
                }
                if (!(rng.uniform() < albedo_med)) break;
                float ax = rng.range(-1.f, 1.f), ay = rng.range(-1.f, 1.f), az = rng.range(-1.f, 1.f);
                power = albedo_med * (power * mcol);  // phase / ph_p == 1
                ro = x;
                rd = normalize(mk(ax, ay, az));
                continue;
            }
            // trace_on_surface :811-875, p_d = 0.7
            V n;
            uint32_t obj;
            finalize_hit(sc, ro, rd, tmin, t, code, inst, n, obj);
            const Mat mat = load_mat(sc, obj);
            V x = fma3(t, rd, ro);
            if (!(rng.uniform() < 0.7f)) break;
            V wi;
            float pdf;
            if (!sample_f(mat, n, wo, rng, wi, pdf)) break;
            V f = bsdf(mat, n, wo, wi);
            float cw = dot(wi, n);
            float cosine_term = cw > 0.f ? cw : 1.f;
            if (mat.kind <= M_PHONG) {  // !is_mirror()
                if (WRITE) {
                    PhotonRec r;
                    r.pos_r = F4{x.x, x.y, x.z, 0.f};
                    r.dir = F4{wo.x, wo.y, wo.z, 0.f};
                    r.pow = F4{power.x, power.y, power.z, 0.f};
                    a.surf[os + ns] = r;
                }
                ns++;
            }
            power = (cosine_term * rcp(pdf * 0.7f)) * (power * f);
            ro = x;
            rd = wi;
            // zero-power photons (back-face hits, bsdf == 0) are still traced and stored: they occupy
            // slots of the k-nearest gathers exactly as in the reference
        }
        if (!WRITE) {
            a.cnt_s[i] = ns;
            a.cnt_v[i] = nv;
        }
    }
}

// ------------------------------------------------------------------ LBVH over points
RPT_DEV uint64_t expand21(uint32_t v) {  // spread 21 bits to every third bit
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// Leaf box of a photon record.  mode 0: the point; 1: sphere of radius pos_r.w; 2: photon beam from
// dir.xyz (start) to pos_r.xyz (end) with radius pos_r.w, box as `impl Bounded for PhotonBeam`
// (src/photon.rs:74-104).
RPT_DEV void leaf_box(const PhotonRec& p, int mode, float lo[3], float hi[3]) {
    const F4 q = p.pos_r;
    if (mode == 2) {
        const float a[3] = {p.dir.x, p.dir.y, p.dir.z}, b[3] = {q.x, q.y, q.z};
        const float cx = (a[0] - b[0]) * (a[0] - b[0]), cy = (a[1] - b[1]) * (a[1] - b[1]), cz = (a[2] - b[2]) * (a[2] - b[2]);
        const float is = rcp(cx + cy + cz);
        const float k[3] = {__builtin_sqrtf((cy + cz) * is), __builtin_sqrtf((cx + cz) * is), __builtin_sqrtf((cx + cy) * is)};
        for (int i = 0; i < 3; i++) {
            float adj = k[i] * q.w * (1.f + 1e-6f);
            lo[i] = fminf(a[i], b[i]) - adj;
            hi[i] = fmaxf(a[i], b[i]) + adj;
        }
    } else {
        const float r = mode == 1 ? q.w * (1.f + 1e-6f) : 0.f;
        lo[0] = q.x - r; lo[1] = q.y - r; lo[2] = q.z - r;
        hi[0] = q.x + r; hi[1] = q.y + r; hi[2] = q.z + r;
    }
}
RPT_DEV V record_centre(const PhotonRec& p, int mode) {
    return mode == 2 ? mk(0.5f * (p.pos_r.x + p.dir.x), 0.5f * (p.pos_r.y + p.dir.y), 0.5f * (p.pos_r.z + p.dir.z)) : xyz(p.pos_r);
}
__global__ void bounds_kernel(const PhotonRec* p, uint32_t n, float* lohi /*6, pre-set to +inf/-inf*/, int mode) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    for (; i < n; i += gridDim.x * blockDim.x) {
        V q = record_centre(p[i], mode);
        lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z);
        hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z);
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; k++) {  // float atomics through the ordered-int trick
            int il = __float_as_int(lo[k]), ih = __float_as_int(hi[k]);
            if (il >= 0) atomicMin((int*)&lohi[k], il); else atomicMax((unsigned*)&lohi[k], (unsigned)il);
            if (ih >= 0) atomicMax((int*)&lohi[3 + k], ih); else atomicMin((unsigned*)&lohi[3 + k], (unsigned)ih);
        }
    }
}
__global__ void morton_kernel(const PhotonRec* p, uint32_t n, const float* lohi, uint64_t* keys, uint32_t* vals, int mode) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V q = record_centre(p[i], mode);
    float ex = fmaxf(lohi[3] - lohi[0], 1e-30f), ey = fmaxf(lohi[4] - lohi[1], 1e-30f), ez = fmaxf(lohi[5] - lohi[2], 1e-30f);
    const float s = 2097151.f;
    uint32_t x = uint32_t(fminf(fmaxf((q.x - lohi[0]) / ex * s, 0.f), s));
    uint32_t y = uint32_t(fminf(fmaxf((q.y - lohi[1]) / ey * s, 0.f), s));
    uint32_t z = uint32_t(fminf(fmaxf((q.z - lohi[2]) / ez * s, 0.f), s));
    keys[i] = (expand21(x) << 2) | (expand21(y) << 1) | expand21(z);
    vals[i] = i;
}
// The last pass of the radix sort (sort_scan.h) hands every (position, key, photon index) to this: the sorted key (Karras'
// construction reads them) and the photon's record, its index in shooting order kept in dir.w -- the gather is part of the sort.
struct GatherPhotons {
    const PhotonRec* in;
    PhotonRec* out;
    uint64_t* keys_out;
    __device__ __forceinline__ void operator()(uint32_t dst, uint64_t key, uint32_t val) const {
        PhotonRec r = in[val];
        r.dir.w = __uint_as_float(val);
        out[dst] = r;
        keys_out[dst] = key;
    }
};
RPT_DEV int lcp(const uint64_t* keys, int n, int i, int j) {  // Karras' delta with index tie-break
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll(a ^ b);
    return 64 + __clz(uint32_t(i) ^ uint32_t(j));
}
// One thread per internal node (Karras 2012).  child < 0x80000000: internal index; else leaf | index.
// range_lo/hi: the contiguous run of sorted photons under each internal node (its Karras interval).
__global__ void karras_kernel(const uint64_t* keys, int n, uint32_t* left, uint32_t* right, uint32_t* parent_int,
                              uint32_t* parent_leaf, uint32_t* range_lo, uint32_t* range_hi) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lcp(keys, n, i, i - d);
    int lmax = 2;
    while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lcp(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    uint32_t lc, rc;
    if (min(i, j) == gamma) { lc = BVH_LEAF | uint32_t(gamma); parent_leaf[gamma] = uint32_t(i); }
    else { lc = uint32_t(gamma); parent_int[gamma] = uint32_t(i); }
    if (max(i, j) == gamma + 1) { rc = BVH_LEAF | uint32_t(gamma + 1); parent_leaf[gamma + 1] = uint32_t(i); }
    else { rc = uint32_t(gamma + 1); parent_int[gamma + 1] = uint32_t(i); }
    left[i] = lc;
    right[i] = rc;
    range_lo[i] = uint32_t(min(i, j));
    range_hi[i] = uint32_t(max(i, j));
    if (i == 0) parent_int[0] = 0xFFFFFFFFu;
}
// Boxes of the internal nodes (6 floats each), in two stages.
// Stage 1: a node over at most kRefitDirect photons takes its box straight from them (its Karras interval is a contiguous
// run of the sorted array): read-only data, no ordering between threads -- that is 15 of every 16 nodes.
// Stage 2: the nodes above are done bottom-up, the second thread to reach a node computes it; the walk starts at every
// leaf or stage-1 node whose parent is such a large node.  Synchronisation without L1 invalidations: a finished box is
// released (stores complete in L2) before the parent's counter is bumped, and the thread that goes on reads its children's
// boxes with agent-scope atomic loads, which are served by L2.  (Two __threadfence() per level cost 11.8 ms per 2 M
// photons; the bottom-up pass over ALL nodes 4.2 ms: 19 fabric-level operations per node.)
static constexpr uint32_t kRefitDirect = 32;
__global__ void refit_small_kernel(const PhotonRec* p, int n, const uint32_t* range_lo, const uint32_t* range_hi, float* box,
                                   int use_radius) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const uint32_t first = range_lo[i], last = range_hi[i];
    if (last - first + 1u > kRefitDirect) return;
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    for (uint32_t k = first; k <= last; k++) {
        float l[3], h[3];
        leaf_box(p[k], use_radius, l, h);
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], l[a]); hi[a] = fmaxf(hi[a], h[a]); }
    }
    float* b = box + size_t(i) * 6;
    for (int a = 0; a < 3; a++) { b[a] = lo[a]; b[3 + a] = hi[a]; }
}
__global__ void refit_kernel(const PhotonRec* p, int n, const uint32_t* left, const uint32_t* right,
                             const uint32_t* parent_int, const uint32_t* parent_leaf, const uint32_t* range_lo,
                             const uint32_t* range_hi, uint32_t* flags, float* box, int use_radius) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n - 1) return;
    auto large = [&](uint32_t node) { return range_hi[node] - range_lo[node] + 1u > kRefitDirect; };
    uint32_t node;   // the large node this thread reports to first
    if (t < n) {
        node = parent_leaf[t];
    } else {
        const uint32_t j = uint32_t(t - n);
        if (large(j)) return;   // large nodes are reached from below
        node = parent_int[j];
        if (node == 0xFFFFFFFFu) return;   // a small root: stage 1 did the whole tree
    }
    if (!large(node)) return;   // stage 1 did the parent
    for (;;) {
        if (__hip_atomic_fetch_add(&flags[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // first arrival: the sibling goes on
        float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
        uint32_t ch[2] = {left[node], right[node]};
        for (int c = 0; c < 2; c++) {
            if (ch[c] & BVH_LEAF) {
                float l[3], h[3];
                leaf_box(p[ch[c] & PH_LEAF_INDEX], use_radius, l, h);
                for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], l[k]); hi[k] = fmaxf(hi[k], h[k]); }
            } else {
                float* b = box + size_t(ch[c]) * 6;
                for (int k = 0; k < 3; k++) {
                    lo[k] = fminf(lo[k], __hip_atomic_load(b + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    hi[k] = fmaxf(hi[k], __hip_atomic_load(b + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
            }
        }
        float* b = box + size_t(node) * 6;
        for (int k = 0; k < 3; k++) {
            __hip_atomic_store(b + k, lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(b + 3 + k, hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        uint32_t par = parent_int[node];
        if (par == 0xFFFFFFFFu) return;
        node = par;
    }
}
// leaf_max > 1 collapses every subtree of at most leaf_max photons into one leaf entry (its photons are a
// contiguous run of the sorted array): fewer dependent node loads per k-NN query.  The internal nodes
// below a collapsed entry stay in the array, unreferenced.
// pad0 / pad1 of a photon-tree node: its parent node (0xFFFFFFFF at the root) and which child of it this node is --
// what the stackless wave-level search (knn_walk_wave) climbs back on.
__global__ void pack_kernel(const PhotonRec* p, int n, const uint32_t* left, const uint32_t* right, const float* box,
                            const uint32_t* range_lo, const uint32_t* range_hi, const uint32_t* parent_int, BvhNode* nodes,
                            int use_radius, uint32_t leaf_max) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    BvhNode w;
    uint32_t ch[2] = {left[i], right[i]};
    float lo[2][3], hi[2][3];
    for (int c = 0; c < 2; c++) {
        if (ch[c] & BVH_LEAF) {
            leaf_box(p[ch[c] & PH_LEAF_INDEX], use_radius, lo[c], hi[c]);
        } else {
            const float* b = box + size_t(ch[c]) * 6;
            for (int k = 0; k < 3; k++) { lo[c][k] = b[k]; hi[c][k] = b[3 + k]; }
            const uint32_t first = range_lo[ch[c]], count = range_hi[ch[c]] - first + 1u;
            if (count <= leaf_max) ch[c] = BVH_LEAF | ((count - 1u) << 26) | first;
        }
    }
    for (int k = 0; k < 3; k++) { w.lo0[k] = lo[0][k]; w.hi0[k] = hi[0][k]; w.lo1[k] = lo[1][k]; w.hi1[k] = hi[1][k]; }
    w.e0 = ch[0];
    w.e1 = ch[1];
    const uint32_t par = parent_int[i];
    w.pad0 = par;
    w.pad1 = (par != 0xFFFFFFFFu && left[par] != uint32_t(i)) ? 1u : 0u;
    nodes[i] = w;
}

RPT_DEV float box_dist2(const float lo[3], const float hi[3], V q) {
    float dx = fmaxf(fmaxf(lo[0] - q.x, q.x - hi[0]), 0.f);
    float dy = fmaxf(fmaxf(lo[1] - q.y, q.y - hi[1]), 0.f);
    float dz = fmaxf(fmaxf(lo[2] - q.z, q.z - hi[2]), 0.f);
    return fmaf(dx, dx, fmaf(dy, dy, dz * dz));
}
// k-nearest walk.  visit(index, d2) returns the current pruning bound (d2 of the k-th best, or +inf).
template <class F>
RPT_DEV void knn_walk(const BvhNode* nodes, const PhotonRec* p, uint32_t n, V q, float& bound, F&& visit) {
    if (n == 0) return;
    if (n == 1) {
        V d = xyz(p[0].pos_r) - q;
        bound = visit(0u, dot(d, d));
        return;
    }
    uint32_t stack[64];
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        if (cur & BVH_LEAF) {
            const uint32_t first = cur & PH_LEAF_INDEX, count = ((cur >> 26) & 31u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                V d = xyz(p[first + k].pos_r) - q;
                bound = visit(first + k, dot(d, d));
            }
        } else {
            const BvhNode nd = nodes[cur];
            float d0 = box_dist2(nd.lo0, nd.hi0, q), d1 = box_dist2(nd.lo1, nd.hi1, q);
            bool h0 = d0 <= bound, h1 = d1 <= bound;
            if (h0 && h1) {
                bool first0 = d0 <= d1;
                if (sp < 64) stack[sp++] = first0 ? nd.e1 : nd.e0;
                cur = first0 ? nd.e0 : nd.e1;
                continue;
            }
            if (h0 || h1) {
                cur = h0 ? nd.e0 : nd.e1;
                continue;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}
// The same search for the live lanes of a wave TOGETHER: one traversal whose control flow is wave-uniform -- a
// child is opened when ANY lane's bound reaches its box -- with the node in scalar registers (one s_load instead of
// 64 per-lane loads) and only the distance tests and the visit per lane.  In lock-step SIMD execution the per-lane
// walks already cost the union of their node visits; what this removes is the 64-entry per-lane stack in scratch
// memory (a dependent memory round trip per pop) and the exec-mask bookkeeping of a divergent loop.  It needs no
// stack at all: a node knows its parent (pad0) and which child of it it is (pad1), and the order in which a node's
// children are opened is a function of the node and the query points alone (the child nearer to most lanes first),
// so a node re-read on the way up knows which child it has just finished.  Every lane is offered a superset of the
// photons its own walk would visit, in another order: the K nearest are the same set.  Meant for queries that lie
// close together (the samples of one pixel); every lane that is live at the call stays live throughout.
template <class F>
RPT_DEV void knn_walk_wave(const BvhNode* nodes, const PhotonRec* p, uint32_t n, V q, float& bound, F&& visit) {
    if (n == 0) return;
    auto leaf = [&](uint32_t e) {
        const uint32_t first = e & PH_LEAF_INDEX, count = ((e >> 26) & 31u) + 1u;
        for (uint32_t k = 0; k < count; k++) {
            const F4 pr = uload(reinterpret_cast<const F4*>(p + first + k));   // pos_r is the record's first 16 bytes
            V d = xyz(pr) - q;
            bound = visit(first + k, dot(d, d));
        }
    };
    if (n == 1u) { leaf(BVH_LEAF); return; }
    const uint32_t live = uint32_t(__popcll(__ballot(true)));
    uint32_t cur = 0u, from = 2u;   // wave-uniform: node, and where it was entered from (2: its parent, 0 / 1: back from that child)
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        const BvhNode nd = uload(nodes + cur);
        const float d0 = box_dist2(nd.lo0, nd.hi0, q), d1 = box_dist2(nd.lo1, nd.hi1, q);
        const uint32_t c_first = (2u * uint32_t(__popcll(__ballot(d0 <= d1))) >= live) ? 0u : 1u;
        uint32_t next = from == 2u ? c_first : (from == c_first ? 1u - c_first : 2u);   // 2: nothing left here
        bool descend = false;
        while (next != 2u) {
            const bool hit = __ballot((next == 0u ? d0 : d1) <= bound) != 0ull;   // (the bounds have shrunk since the node was entered)
            const uint32_t e = next == 0u ? nd.e0 : nd.e1;
            if (hit && !(e & BVH_LEAF)) { cur = e; from = 2u; descend = true; break; }
            if (hit) leaf(e);
            next = next == c_first ? 1u - c_first : 2u;
        }
        if (descend) continue;
        if (nd.pad0 == 0xFFFFFFFFu) break;
        from = nd.pad1;
        cur = nd.pad0;
    }
}
// radius = distance to the k-th (k = 10, self included) nearest photon: src/photon.rs:214-232
__global__ __launch_bounds__(256) void knn_radius_kernel(const BvhNode* nodes, PhotonRec* p, uint32_t n, float* radius_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V q = xyz(p[i].pos_r);
    float best[10];
#pragma unroll
    for (int k = 0; k < 10; k++) best[k] = kInf;
    float bound = kInf;
    uint32_t found = 0;
    knn_walk(nodes, p, n, q, bound, [&](uint32_t, float d2) {
        float v = d2;
#pragma unroll
        for (int k = 0; k < 10; k++) {  // sorted insertion, ascending
            float lo = fminf(best[k], v);
            v = fmaxf(best[k], v);
            best[k] = lo;
        }
        found++;
        return best[9];
    });
    float m = 0.f;  // max over the (up to 10) found; +inf slots are unfilled
#pragma unroll
    for (int k = 0; k < 10; k++) m = (best[k] < kInf) ? fmaxf(m, best[k]) : m;
    radius_out[i] = __builtin_sqrtf(m);
}
__global__ void set_radius_kernel(PhotonRec* p, uint32_t n, const float* radius) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i].pos_r.w = radius[i];
}

// ------------------------------------------------------------------ camera pass
struct QueryArgs {
    RenderArgs r;
    const BvhNode* s_nodes; const PhotonRec* s_ph; uint32_t n_s;   // surface photons (points)
    const BvhNode* v_nodes; const PhotonRec* v_ph; uint32_t n_v;   // volume photons (spheres for the beam query)
    uint32_t kind, gather_size, gather_size_volume;
    uint32_t* overflow;  // set to 1 if a beam-walk stack overflowed (the render is then rejected)
    uint32_t region_dwords;  // LDS dwords per wave: max(gather lists, beam stack + staging)
    uint32_t skip;           // diagnostic (rpt_set_option "photon_skip"): 1 = no volume estimate, 2 = no surface estimate
    uint32_t* cand;          // [waves of the grid][cand_cap] candidate photons of each wave's current pixel block
    uint32_t cand_cap;       // 0: no candidate lists (every sample walks the tree)
    uint32_t* gather;        // GG kernels (gather size > kGatherLds): [waves of the grid][2][K][64] gather lists in global memory
    uint32_t parts;          // work items per (8x8 pixel block, sample chunk): the block's rows in 1, 2, 4 or 8 strips
    uint32_t coop_cap;       // candidates the wave-level surface gather may hold in LDS (multiple of 4, <= kCoopCap); 0: one search per lane
    uint32_t* emit;          // EMIT kernels (reference-epsilon mode): [n_owned][gather_size + 2][iterations] per-sample selections, see
                             // rpt64::SurfArgs64::emit -- the surface estimate's rays and terms are kernels_f64.hip's
};

static_assert(offsetof(QueryArgs, r) == 0, "kernel arguments begin with the SceneView (kernarg_scene)");
// Batched wave-cooperative walk (used when the rays of a wave do not form a packet).  Walking one
// node at a time costs one dependent memory round trip per node (~3,400 cycles each with 220 MB of
// nodes + photons far beyond L2: measured 118 ms per-lane, 56 ms wave-uniform for the same pass that
// takes 39 ms batched and 18 ms as a packet).  Here the wave pops up to
// 64 pending entries at once, lane i fetches entry i's node / photon record (64 loads in flight),
// stages it in LDS, and then all lanes test their own ray against each staged record in turn
// (broadcast LDS reads, ballot to decide which children to push).  Pending entries live on a
// wave-private LDS stack of kBeamCap entries; near the cap the walk degrades to depth-first (batch
// of 1), whose extra footprint is bounded by the tree depth.  *overflow is set if even that fails.
static constexpr uint32_t kBeamCap = 1024;
// Largest k-nearest gather whose per-lane (distance, index) lists fit the wave's LDS region (4 waves x [2][K][64]
// dwords in 160 KB next to the 32 KB traversal stack); larger gathers (examples/lighthouse.rs and
// volumetric_photonphoton_lampshade.rs use gather_size 100) keep the same lists in global memory (GG kernels).
static constexpr uint32_t kGatherLds = 56;
static constexpr uint32_t kGatherMax = 1024;
static constexpr uint32_t kCandCap = 4096;  // candidate photons one wave keeps per 8x8 pixel block (global memory)
// Samples of one pixel that a work item of the camera pass handles (four trips of 64 lanes).  The beam x point estimate
// keeps their rays in LDS ([kSuper] x 16 B per wave) and runs with one PHOTON per lane over them.
static constexpr uint32_t kSuper = 256;
static constexpr uint32_t kPendCap = 128;   // culled candidate indices waiting for a full batch of 64 (LDS, per wave)
template <class G, class F>
RPT_DEV void beam_walk_batch(const BvhNode* nodes, const PhotonRec* photons, uint32_t n, bool active, V o, V d,
                             uint32_t* wstack, F4* stage, uint32_t* overflow, G&& prep, F&& visit) {
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const V inv = mk(rcp(d.x), rcp(d.y), rcp(d.z));
    uint32_t count = 1;  // wave-uniform
    if (lane == 0) wstack[0] = (n == 1) ? BVH_LEAF : 0u;
    while (count != 0) {
        const uint32_t b = (count > kBeamCap - 160u) ? 1u : min(count, 64u);
        uint32_t e = 0u;
        if (lane < b) {
            e = wstack[count - b + lane];
            if (e & BVH_LEAF) {
                const PhotonRec ph = prep(photons[e & PH_LEAF_INDEX]);
                stage[lane * 4u + 0u] = ph.pos_r;
                stage[lane * 4u + 1u] = ph.dir;
                stage[lane * 4u + 2u] = ph.pow;
            } else {
                const F4* src = reinterpret_cast<const F4*>(nodes + e);
                const F4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
                stage[lane * 4u + 0u] = a0;
                stage[lane * 4u + 1u] = a1;
                stage[lane * 4u + 2u] = a2;
                stage[lane * 4u + 3u] = a3;
            }
        }
        count -= b;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = 0; j < b; j++) {
            const uint32_t ej = __builtin_amdgcn_readlane(e, j);
            if (ej & BVH_LEAF) {
                PhotonRec ph;
                ph.pos_r = stage[j * 4u + 0u];
                ph.dir = stage[j * 4u + 1u];
                ph.pow = stage[j * 4u + 2u];
                if (active) visit(ph);
            } else {
                const F4 q0 = stage[j * 4u + 0u], q1 = stage[j * 4u + 1u], q2 = stage[j * 4u + 2u], q3 = stage[j * 4u + 3u];
                const float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q1.x, q1.y, q1.z};
                const float lo1[3] = {q2.x, q2.y, q2.z}, hi1[3] = {q3.x, q3.y, q3.z};
                float n0, f0, n1, f1;
                slab2(lo0, hi0, o, inv, n0, f0);
                slab2(lo1, hi1, o, inv, n1, f1);
                const bool a0 = __ballot(active && fmaxf(n0, 0.f) <= f0) != 0;
                const bool a1 = __ballot(active && fmaxf(n1, 0.f) <= f1) != 0;
                if (a0) {
                    if (count < kBeamCap) { if (lane == 0) wstack[count] = __float_as_uint(q0.w); count++; }
                    else if (lane == 0) *overflow = 1u;
                }
                if (a1) {
                    if (count < kBeamCap) { if (lane == 0) wstack[count] = __float_as_uint(q2.w); count++; }
                    else if (lane == 0) *overflow = 1u;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- ray packets: the wave bounds its directions by a four-plane frustum through the common origin
RPT_DEV float wave_min(float v) {
    for (int off = 32; off; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
RPT_DEV float wave_max(float v) {
    for (int off = 32; off; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
RPT_DEV bool box_outside(const float lo[3], const float hi[3], V o, V nrm) {
    // true if the whole box lies on the negative side of the plane through o with inward normal nrm
    V c = mk(0.5f * (lo[0] + hi[0]) - o.x, 0.5f * (lo[1] + hi[1]) - o.y, 0.5f * (lo[2] + hi[2]) - o.z);
    V h = mk(0.5f * (hi[0] - lo[0]), 0.5f * (hi[1] - lo[1]), 0.5f * (hi[2] - lo[2]));
    float reach = fabsf(nrm.x) * h.x + fabsf(nrm.y) * h.y + fabsf(nrm.z) * h.z;
    return dot(nrm, c) + reach < 0.f;
}
// The frustum of a ray packet: common origin o0, central axis m and four side planes through o0 with
// inward normals (not normalised; ll..lt are their lengths).
struct Frustum {
    V o0, m, nl, nr, nb, nt;
    float ll, lr, lb, lt;
    V axis;          // unit direction through the middle of the (u, v) bounds
    float tan_half;  // tangent of the half angle of the cone around `axis` that contains the whole frustum
};
// Bounds `nd` directions per live lane (all from one origin).  false if the lanes do not share an origin or
// a direction is more than 60 degrees off the first lane's.
RPT_DEV bool frustum_from_dirs(bool active, V o, const V* dirs, int nd, Frustum& fr) {
    const uint64_t act = __ballot(active);
    if (act == 0) return false;
    const uint32_t first = uint32_t(__ffsll((unsigned long long)act)) - 1u;
    auto bcast = [&](float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), first)); };
    const V o0 = mk(bcast(o.x), bcast(o.y), bcast(o.z));
    const V m = normalize(mk(bcast(dirs[0].x), bcast(dirs[0].y), bcast(dirs[0].z)));
    // orthonormal basis around m (Duff et al.), perspective coordinates of every live direction
    const float sg = __builtin_copysignf(1.f, m.z);
    const float aa = -1.f / (sg + m.z), bb = m.x * m.y * aa;
    const V u = mk(1.f + sg * m.x * m.x * aa, sg * bb, -sg * m.x), v = mk(bb, sg + m.y * m.y * aa, -m.y);
    bool fits = !active || (o.x == o0.x && o.y == o0.y && o.z == o0.z);
    float ulo = kInf, uhi = -kInf, vlo = kInf, vhi = -kInf;
    for (int k = 0; k < nd; k++) {
        const float dm = dot(dirs[k], m);
        fits = fits && (!active || dm > 0.5f * __builtin_sqrtf(dot(dirs[k], dirs[k])));
        const float idm = rcp(dm);
        const float pu = dot(dirs[k], u) * idm, pv = dot(dirs[k], v) * idm;
        if (active) {
            ulo = fminf(ulo, pu); uhi = fmaxf(uhi, pu);
            vlo = fminf(vlo, pv); vhi = fmaxf(vhi, pv);
        }
    }
    if (__ballot(!fits) != 0) return false;
    const float pad = 1e-5f;
    const float umin = wave_min(ulo) - pad, umax = wave_max(uhi) + pad;
    const float vmin = wave_min(vlo) - pad, vmax = wave_max(vhi) + pad;
    // every field is wave-uniform: moved to scalar registers (22 VGPRs per frustum otherwise)
    auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    auto uni3 = [&](V x) { return mk(uni(x.x), uni(x.y), uni(x.z)); };
    fr.o0 = uni3(o0);
    fr.m = uni3(m);
    fr.nl = uni3(fma3(-umin, m, u)); fr.nr = uni3(fma3(umax, m, -u));
    fr.nb = uni3(fma3(-vmin, m, v)); fr.nt = uni3(fma3(vmax, m, -v));
    // u, v, m are orthonormal: |n| = sqrt(1 + bound^2)
    fr.ll = uni(__builtin_sqrtf(fmaf(umin, umin, 1.f))); fr.lr = uni(__builtin_sqrtf(fmaf(umax, umax, 1.f)));
    fr.lb = uni(__builtin_sqrtf(fmaf(vmin, vmin, 1.f))); fr.lt = uni(__builtin_sqrtf(fmaf(vmax, vmax, 1.f)));
    // bounding cone: axis through the centre of the bounds; its half angle reaches the farthest corner direction
    const float uc = 0.5f * (umin + umax), vc = 0.5f * (vmin + vmax);
    const V ax = normalize(m + uc * u + vc * v);
    float cmin = 1.f;
    for (int k = 0; k < 4; k++) {
        const V cd = normalize(m + ((k & 1) ? umax : umin) * u + ((k & 2) ? vmax : vmin) * v);
        cmin = fminf(cmin, dot(cd, ax));
    }
    fr.axis = uni3(ax);
    fr.tan_half = uni(__builtin_sqrtf(fmaxf(1.f - cmin * cmin, 0.f)) * rcp(fmaxf(cmin, 1e-6f)) * 1.0001f + 1e-7f);
    return true;
}
RPT_DEV bool sphere_outside(const Frustum& fr, const F4& pos_r) {
    const V c = xyz(pos_r) - fr.o0;
    const float r = pos_r.w;
    if (dot(fr.m, c) < -r || dot(fr.nl, c) < -r * fr.ll || dot(fr.nr, c) < -r * fr.lr || dot(fr.nb, c) < -r * fr.lb ||
        dot(fr.nt, c) < -r * fr.lt)
        return true;
    // the four planes leave a square cross-section; a narrow frustum (one pixel) is far better bounded by its cone:
    // outside if the centre is farther from the axis than the cone's radius at that depth plus r / cos(half angle)
    const float z = dot(c, fr.axis);
    if (z <= 0.f) return false;
    const float d2 = fmaxf(dot(c, c) - z * z, 0.f);
    const float reach = fmaf(fr.tan_half, z, r * __builtin_sqrtf(fmaf(fr.tan_half, fr.tan_half, 1.f)));
    return d2 > reach * reach;
}
// Walk of the photon tree against a frustum: each lane culls a DIFFERENT pending node's two child boxes (64
// nodes per instruction stream), survivors are pushed with a ballot prefix onto the wave-private LDS stack.
// `leaves(is_leaf, entry, record)` is called wave-convergently once per batch of up to 64 popped entries;
// SPHERES: a leaf counts only if the photon sphere itself (pos_r = centre, radius) reaches into the frustum
// (its box, which is all the parent node knows, is ~1.5x looser).
template <bool SPHERES, class L>
RPT_DEV void frustum_walk(const BvhNode* nodes, const PhotonRec* photons, uint32_t n, const Frustum& fr, uint32_t* wstack,
                          uint32_t* overflow, L&& leaves) {
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t count = 1;  // wave-uniform
    if (lane == 0) wstack[0] = (n == 1) ? BVH_LEAF : 0u;
    while (count != 0) {
        const uint32_t b = (count > kBeamCap - 160u) ? 1u : min(count, 64u);
        const bool mine = lane < b;
        uint32_t e = 0u;
        bool s0 = false, s1 = false, leaf = false;
        uint32_t c0 = 0u, c1 = 0u;
        PhotonRec raw{};
        if (mine) {
            e = wstack[count - b + lane];
            leaf = (e & BVH_LEAF) != 0u;
            if (leaf) {
                raw = photons[e & PH_LEAF_INDEX];
                if (SPHERES) leaf = !sphere_outside(fr, raw.pos_r);
            } else {
                const BvhNode nd = nodes[e];
                c0 = nd.e0;
                c1 = nd.e1;
                s0 = !(box_outside(nd.lo0, nd.hi0, fr.o0, fr.m) || box_outside(nd.lo0, nd.hi0, fr.o0, fr.nl) ||
                       box_outside(nd.lo0, nd.hi0, fr.o0, fr.nr) || box_outside(nd.lo0, nd.hi0, fr.o0, fr.nb) ||
                       box_outside(nd.lo0, nd.hi0, fr.o0, fr.nt));
                s1 = !(box_outside(nd.lo1, nd.hi1, fr.o0, fr.m) || box_outside(nd.lo1, nd.hi1, fr.o0, fr.nl) ||
                       box_outside(nd.lo1, nd.hi1, fr.o0, fr.nr) || box_outside(nd.lo1, nd.hi1, fr.o0, fr.nb) ||
                       box_outside(nd.lo1, nd.hi1, fr.o0, fr.nt));
            }
        }
        count -= b;
        // wave-aggregated push of the surviving children
        const uint64_t m0 = __ballot(s0), m1 = __ballot(s1);
        const uint32_t n0 = uint32_t(__popcll(m0)), n1 = uint32_t(__popcll(m1));
        if (count + n0 + n1 > kBeamCap) {
            if (lane == 0) *overflow = 1u;
        } else {
            if (s0) wstack[count + __builtin_amdgcn_mbcnt_hi(uint32_t(m0 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m0), 0u))] = c0;
            if (s1) wstack[count + n0 + __builtin_amdgcn_mbcnt_hi(uint32_t(m1 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m1), 0u))] = c1;
            count += n0 + n1;
        }
        __builtin_amdgcn_wave_barrier();
        leaves(leaf, e & PH_LEAF_INDEX, raw);
        __builtin_amdgcn_wave_barrier();
    }
}
// The lanes flagged `take` store their prepared record densely in the staging slots (ballot prefix), then
// every live ray tests every staged record: a counted loop, so several records' LDS reads are in flight.
template <class F>
RPT_DEV void stage_and_test(bool take, const PhotonRec& staged, F4* stage, bool active, F&& visit) {
    const uint64_t lm = __ballot(take);
    const uint32_t n_leaf = uint32_t(__popcll(lm));
    __builtin_amdgcn_wave_barrier();
    if (take) {
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi(uint32_t(lm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(lm), 0u));
        stage[slot * 4u + 0u] = staged.pos_r;
        stage[slot * 4u + 1u] = staged.dir;
        stage[slot * 4u + 2u] = staged.pow;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 2
    for (uint32_t j = 0; j < n_leaf; j++) {
        PhotonRec ph;
        ph.pos_r = stage[j * 4u + 0u];
        ph.dir = stage[j * 4u + 1u];
        ph.pow = stage[j * 4u + 2u];
        if (active) visit(ph);
    }
    __builtin_amdgcn_wave_barrier();
}
// Packet walk: when the live lanes' rays share one origin (pinhole camera, one pixel tile per wave) the
// inner nodes are not tested per ray at all.  Conservative culling + exact per-ray test of every staged
// photon => the sum equals the brute-force sum.  Returns false if the rays do not form a packet.
template <bool SPHERES, class G, class F>
RPT_DEV bool beam_walk_packet(const BvhNode* nodes, const PhotonRec* photons, uint32_t n, bool active, V o, V d,
                              uint32_t* wstack, F4* stage, uint32_t* overflow, G&& prep, F&& visit) {
    if (n == 0) return true;
    if (__ballot(active) == 0) return true;
    Frustum fr;
    if (!frustum_from_dirs(active, o, &d, 1, fr)) return false;
    frustum_walk<SPHERES>(nodes, photons, n, fr, wstack, overflow, [&](bool leaf, uint32_t, const PhotonRec& raw) {
        PhotonRec staged{};
        if (leaf) staged = prep(raw, fr.o0);
        stage_and_test(leaf, staged, stage, active, visit);
    });
    return true;
}
// `guess`: squared radius the search starts with instead of +inf.  If fewer than K photons lie inside it
// the caller repeats the search unbounded, so the result is always the exact K nearest.
// WAVE: the live lanes search together (knn_walk_wave); every live lane of the wave must make the call.
template <bool WAVE = false>
RPT_DEV uint32_t gather_knn(const BvhNode* nodes, const PhotonRec* photons, uint32_t n, V x, uint32_t K, float* gd,
                            uint32_t* gi, float& max_d2, float guess = kInf) {
    uint32_t found = 0;
    float bound = guess, worst = 0.f;
    uint32_t worst_slot = 0;
    if (K > 0) {
        auto walk = [&](auto&& visit) {
            if (WAVE) knn_walk_wave(nodes, photons, n, x, bound, visit);
            else knn_walk(nodes, photons, n, x, bound, visit);
        };
        walk([&](uint32_t idx, float d2) {
            if (found < K) {
                if (d2 > guess) return guess;
                gd[found * 64u] = d2;
                gi[found * 64u] = idx;
                found++;
                if (found < K) return guess;
            } else if (d2 < worst) {
                gd[worst_slot * 64u] = d2;
                gi[worst_slot * 64u] = idx;
            } else {
                return worst;
            }
            worst = -1.f;  // (re)locate the current worst
            for (uint32_t k = 0; k < K; k++) {
                float v = gd[k * 64u];
                if (v > worst) { worst = v; worst_slot = k; }
            }
            return worst;
        });
    }
    max_d2 = 0.f;
    for (uint32_t k = 0; k < found; k++) max_d2 = fmaxf(max_d2, gd[k * 64u]);
    return found;
}

// ---- the surface gather of a pixel's samples, by the wave together.
// The samples of a pixel hit within a footprint of each other, far less than a gather radius apart: their k-nearest sets
// overlap almost entirely.  Instead of 64 searches (or one lock-step search with 64 selections going on inside it) the wave
// 1. collects every photon within reach of the cluster of query points -- a ball around one of them that contains each
//    member's own search ball -- walking the tree with a DIFFERENT pending node per lane (ball_collect: 64 nodes per
//    instruction stream, no dependent load per node and lane);
// 2. orders the candidates by their distance to the cluster's centre (sort_candidates: rank by counting);
// 3. lets every lane pick its K nearest from the LDS list: in that order a lane's list is nearly final once it is full,
//    and the scan stops when no lane can be reached any more.
// Conservative collection + exact per-lane distances => each lane gets exactly its K nearest photons.
static constexpr uint32_t kBallStack = 128;   // pending tree entries of ball_collect (LDS, per wave)
static constexpr uint32_t kCoopCap = 256;     // candidates at most (four per lane in sort_candidates)
static constexpr uint32_t kCoopOverflow = 0xFFFFFFFFu;
// Diagnostic counters of the camera pass (counters build): bumped in memory where they occur, by one lane -- kept in
// registers for the kernel's life they cost the ordinary build a quarter of its VGPR budget's slack.
RPT_DEV void diag_add(unsigned long long* counters, int k, unsigned long long v) {
    if (counters && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) atomicAdd(&counters[k], v);
}
// The deepest inner node below which every photon within sqrt(R2) of c lies: down from the root while only one child's
// box is in reach (c, R2 wave-uniform; the nodes come through scalar loads).  A pixel's query ball is tiny against the
// map: most of a walk from the root is this chain of one-child steps, a dependent load each -- so the wave keeps the end
// of the chain of a LARGER ball around its current query (the anchor) and starts from there until a query leaves it.
RPT_DEV uint32_t ball_anchor(const BvhNode* nodes, uint32_t n, V c, float R2) {
    uint32_t cur = 0u;
    if (n < 2u) return cur;
    for (;;) {
        const BvhNode nd = uload(nodes + cur);
        const bool h0 = box_dist2(nd.lo0, nd.hi0, c) <= R2, h1 = box_dist2(nd.lo1, nd.hi1, c) <= R2;
        const uint32_t both = __builtin_amdgcn_readfirstlane((h0 ? 1u : 0u) | (h1 ? 2u : 0u));
        if (both != 1u && both != 2u) break;
        const uint32_t next = both == 1u ? nd.e0 : nd.e1;
        if (next & BVH_LEAF) break;
        cur = __builtin_amdgcn_readfirstlane(next);
    }
    return cur;
}
// Every photon below the inner node `start` within sqrt(R2) of c, as (position, index) records in cand[0 .. return value);
// the set is a function of (c, R2) alone when `start` is an anchor of a ball that contains this one.  kCoopOverflow if the
// stack or the list does not hold them.  `lscratch`: 64 dwords of wave-private LDS.
// `box_on`, `box_lo`, `box_hi`: the room shell's box (ShellBox); a candidate inside it carries bit 31 in its index word.
static constexpr uint32_t kCandInShell = 0x80000000u;
RPT_DEV uint32_t ball_collect(const BvhNode* nodes, const PhotonRec* p, uint32_t n, uint32_t start, V c, float R2, uint32_t* pstack,
                              uint32_t* lscratch, F4* cand, uint32_t cap, uint32_t& steps, bool box_on, F4 box_lo, F4 box_hi) {
    const uint32_t lane = threadIdx.x & 63u;
    auto prefix = [](uint64_t m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); };
    uint32_t count = 1u, M = 0u;  // wave-uniform
    if (lane == 0u) pstack[0] = (n == 1u) ? BVH_LEAF : start;
    __builtin_amdgcn_wave_barrier();
    while (count != 0u) {
        steps++;
        const uint32_t b = min(count, 64u);
        const bool mine = lane < b;
        uint32_t e = 0u;
        if (mine) e = pstack[count - b + lane];
        count -= b;
        const bool leaf = mine && (e & BVH_LEAF) != 0u;
        bool s0 = false, s1 = false;
        uint32_t c0 = 0u, c1 = 0u;
        if (mine && !leaf) {
            const BvhNode nd = nodes[e];
            c0 = nd.e0;
            c1 = nd.e1;
            s0 = box_dist2(nd.lo0, nd.hi0, c) <= R2;
            s1 = box_dist2(nd.lo1, nd.hi1, c) <= R2;
        }
        const uint64_t m0 = __ballot(s0), m1 = __ballot(s1);
        const uint32_t n0 = uint32_t(__popcll(m0)), n1 = uint32_t(__popcll(m1));
        if (count + n0 + n1 > kBallStack) return kCoopOverflow;
        __builtin_amdgcn_wave_barrier();   // the popped entries are read before their slots are written again
        if (s0) pstack[count + prefix(m0)] = c0;
        if (s1) pstack[count + n0 + prefix(m1)] = c1;
        count += n0 + n1;
        const uint64_t lm = __ballot(leaf);
        if (lm != 0ull) {   // the photons of the popped leaves: eight leaves at a time, one photon per lane, all loads in flight
            const uint32_t nl = uint32_t(__popcll(lm));
            if (leaf) lscratch[prefix(lm)] = e;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t g = 0; g < nl; g += 8u) {
                const uint32_t li = g + (lane >> 3), k = lane & 7u;
                uint32_t le = 0u;
                if (li < nl) le = lscratch[li];
                const uint32_t first = le & PH_LEAF_INDEX, cnt = li < nl ? ((le >> 26) & 31u) + 1u : 0u;
                for (uint32_t k0 = 0; __ballot(k0 + k < cnt) != 0ull; k0 += 8u) {   // (one pass: k-nearest trees have leaves of <= 8)
                    bool take = false;
                    F4 pr{};
                    if (k0 + k < cnt) {
                        pr = p[first + k0 + k].pos_r;
                        const V d = xyz(pr) - c;
                        take = dot(d, d) <= R2;
                    }
                    const uint64_t tm = __ballot(take);
                    const uint32_t nt = uint32_t(__popcll(tm));
                    if (M + nt > cap) return kCoopOverflow;
                    if (take) {
                        const bool inside = !box_on || (pr.x >= box_lo.x && pr.x <= box_hi.x && pr.y >= box_lo.y && pr.y <= box_hi.y &&
                                                        pr.z >= box_lo.z && pr.z <= box_hi.z);
                        cand[M + prefix(tm)] = F4{pr.x, pr.y, pr.z, __uint_as_float((first + k0 + k) | (inside ? kCandInShell : 0u))};
                    }
                    M += nt;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return M;
}
// Orders cand[0 .. M) by squared distance to c (ties keep their list order) and leaves those distances in keys[0 .. M).
// Rank by counting: every lane holds up to four records and counts, for each, the keys below it (broadcast LDS reads).
RPT_DEV void sort_candidates(F4* cand, float* keys, uint32_t M, V c) {
    const uint32_t lane = threadIdx.x & 63u;
    F4 rec[4];
    float key[4];
    uint32_t rank[4];
#pragma unroll
    for (uint32_t r = 0; r < 4u; r++) {
        const uint32_t i = r * 64u + lane;
        rec[r] = F4{0.f, 0.f, 0.f, 0.f};
        key[r] = kInf;
        rank[r] = 0u;
        if (i < M) {
            rec[r] = cand[i];
            const V d = xyz(rec[r]) - c;
            key[r] = dot(d, d);
            keys[i] = key[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 8
    for (uint32_t j = 0; j < M; j++) {
        const float kj = keys[j];
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++) {
            if (r * 64u < M) {   // wave-uniform
                const uint32_t i = r * 64u + lane;
                rank[r] += (kj < key[r] || (kj == key[r] && j < i)) ? 1u : 0u;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t r = 0; r < 4u; r++) {
        const uint32_t i = r * 64u + lane;
        if (i < M) {
            cand[rank[r]] = rec[r];
            keys[rank[r]] = key[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- pieces of the camera pass (photon_query_kernel).  `q`: the kernel's arguments where they lie (kernarg_args).
typedef const __attribute__((address_space(4))) QueryArgs& QueryK;

// The volume estimates with the SAMPLES in the lanes (beam x beam; beam x point when the rays share no origin or the strip
// has no candidate list): the sum of the weighted photon powers along this lane's ray, before the medium's colour.
template <int KIND>
RPT_DEV V volume_estimate_sample_lanes(QueryK q, bool active, V ro, V rd, bool hit, float t, float sigma_t, float phase,
                                       uint32_t* wstack, F4* stage) {
    V vc = mk(0, 0, 0);
    unsigned long long c_leaf = 0, c_acc = 0;   // diagnostic: photons (beams) tested / accepted by this lane's ray
    // The staging lane pre-computes what depends on the photon only: pos_r.w = r^2, pow = power *
    // 3/pi * phase / r^2, pow.w = 1/r^2 (src/photon.rs:474-493: k2(d^2/r^2)/r^2 with k2(x) = 3/pi (1-x)^2).
    auto prep_point = [&](PhotonRec ph) {
        const float r2 = ph.pos_r.w * ph.pos_r.w, ir2 = rcp(r2), kk = (3.f * kInvPi) * phase * ir2;
        ph.pos_r.w = r2;
        ph.pow = F4{ph.pow.x * kk, ph.pow.y * kk, ph.pow.z * kk, ir2};
        return ph;
    };
    auto visit = [&](const PhotonRec& ph) {
        c_leaf++;
        V otc = xyz(ph.pos_r) - ro;
        float disk = dot(otc, rd);
        V dv = fma3(disk, rd, ro) - xyz(ph.pos_r);
        float dist2 = dot(dv, dv);
        bool ok = disk > 0.f && dist2 < ph.pos_r.w && !(hit && dot(otc, otc) > t * t);
        if (ok) {
            c_acc++;
            float tmp = 1.f - dist2 * ph.pow.w;
            float w = tmp * tmp * __expf(-sigma_t * disk);
            vc = fma3(w, xyz(ph.pow), vc);
        }
    };
    auto prep_none = [](const PhotonRec& ph) { return ph; };
    // Packet form (every ray of the wave starts at o0; the same per-photon terms as in the photon-per-lane form
    // above, computed by the staging lane; used when the strip has no candidate list)
    const float t2 = hit ? t * t : kInf;
    auto prep_packet = [&](PhotonRec ph, const V& o0) {
        const V c = xyz(ph.pos_r) - o0;
        const float r2 = ph.pos_r.w * ph.pos_r.w, ir2 = rcp(r2), c2 = dot(c, c), len = sqrt1(c2);
        const float kk = (3.f * kInvPi) * phase * ir2 * __expf(-sigma_t * len);
        ph.pos_r = F4{c.x, c.y, c.z, r2};
        ph.dir = F4{c2, len, ir2, 0.f};
        ph.pow = F4{ph.pow.x * kk, ph.pow.y * kk, ph.pow.z * kk, 0.f};
        return ph;
    };
    auto visit_packet = [&](const PhotonRec& ph) {
        if (q.skip & 4u) return;  // diagnostic: tree walk and staging only
        c_leaf++;
        const V c = xyz(ph.pos_r);
        const float disk = dot(c, rd);
        const V dv = fma3(disk, rd, -c);
        const float dist2 = dot(dv, dv);
        const bool ok = disk > 0.f && dist2 < ph.pos_r.w && ph.dir.x <= t2;
        if (ok) {
            c_acc++;
            const float tmp = 1.f - dist2 * ph.dir.z;
            const float w = tmp * tmp * fmaf(sigma_t, ph.dir.y - disk, 1.f);
            vc = fma3(w, xyz(ph.pow), vc);
        }
    };
    // beam x beam estimate, src/photon.rs:503-593 (equation 38 of Jarosz et al.)
    const V inv_rd = mk(rcp(rd.x), rcp(rd.y), rcp(rd.z));
    auto visit_beam = [&](const PhotonRec& ph) {
        c_leaf++;
        float lo[3], hi[3], tn, tf;
        leaf_box(ph, 2, lo, hi);
        slab2(lo, hi, ro, inv_rd, tn, tf);
        if (!(fmaxf(tn, 0.f) <= tf)) return;  // bvh `traverse`: only beams whose own box the ray hits
        const V bstart = xyz(ph.dir), bend = xyz(ph.pos_r);
        const float radius = ph.pos_r.w;
        const V bvec = bend - bstart;
        const float beam_len = sqrt1(dot(bvec, bvec));
        const V bdir = rcp(beam_len) * bvec;
        const V l = bstart - ro;
        const V u = normalize(cross(l, bdir));
        const V nn = normalize(cross(bdir, u));
        const float tq = dot(nn, l) * rcp(dot(nn, rd));
        const V qc = fma3(tq, rd, ro);
        const float dd = dot(rd, bdir);
        const float beam_t = dot(bdir, qc - bstart);
        const V bc = fma3(beam_t, bdir, bstart);
        const V dq = qc - bc;
        const float dist = sqrt1(dot(dq, dq));
        const bool ok = !(hit && tq >= t) && beam_t >= 0.f && beam_t <= beam_len && dist < radius;
        if (ok) {
            c_acc++;
            const float inv_sin = rsq(fmaxf(0.f, 1.f - dd * dd));
            const float tmp = 1.f - dist * rcp(radius);
            const float w = sigma_t * phase * inv_sin * __expf(-sigma_t * tq) * __expf(-sigma_t * beam_t) *
                            (3.f * kInvPi) * tmp * tmp * rcp(2.f * radius);
            vc = fma3(w, xyz(ph.pow), vc);
        }
    };
    if (KIND == RPT_PHOTON_BEAM_BEAM) {
        if (!beam_walk_packet<false>(q.v_nodes, q.v_ph, q.n_v, active, ro, rd, wstack, stage, q.overflow,
                                     [](const PhotonRec& ph, const V&) { return ph; }, visit_beam))
            beam_walk_batch(q.v_nodes, q.v_ph, q.n_v, active, ro, rd, wstack, stage, q.overflow, prep_none, visit_beam);
    } else {
        if (!beam_walk_packet<true>(q.v_nodes, q.v_ph, q.n_v, active, ro, rd, wstack, stage, q.overflow, prep_packet, visit_packet))
            beam_walk_batch(q.v_nodes, q.v_ph, q.n_v, active, ro, rd, wstack, stage, q.overflow, prep_point, visit);
    }
    if (q.r.counters) { atomicAdd(&q.r.counters[5], c_leaf); atomicAdd(&q.r.counters[6], c_acc); }
    return vc;
}

// Beam x point estimate of one pixel with ONE PHOTON PER LANE (src/photon.rs:439-502): `cand` lists the photon spheres that
// reach into the strip's frustum, `rays` holds the pixel's n_s rays (unit direction, squared hit distance; far2 / near2:
// the largest / smallest of those distances).  Returns this lane's photons' sum over all the rays, before the medium's colour.
RPT_DEV V beam_estimate_photon_lanes(QueryK q, const uint32_t* cand, uint32_t cand_n, float far2, float near2, float xn, float yn,
                                     V cam_right, V cam_up, float sigma_t, uint32_t n_s, const float4* rays, uint32_t* pend_list) {
    const auto& a = q.r;
    const uint32_t lane_ = threadIdx.x & 63u;
    V beam_sum = mk(0, 0, 0);
    // the pixel's own frustum (footprint included) re-culls the strip's candidates
    const float e = a.inv_dim * 1.0001f;
    const V dd = mk(a.cam.ddir[0], a.cam.ddir[1], a.cam.ddir[2]);
    const V eye0 = mk(a.cam.eye[0], a.cam.eye[1], a.cam.eye[2]);  // pinhole: every ray starts here
    const V corners[4] = {dd + (xn - e) * cam_right + (yn - e) * cam_up, dd + (xn + e) * cam_right + (yn - e) * cam_up,
                          dd + (xn - e) * cam_right + (yn + e) * cam_up, dd + (xn + e) * cam_right + (yn + e) * cam_up};
    Frustum fs;
    const bool have_fs = frustum_from_dirs(true, eye0, corners, 4, fs);
    const float phase = a.sc.medium_phase;
    // A batch of nb <= 64 culled candidates: lane j takes pend_list[j].  What depends on the photon and the common
    // origin is computed once per photon -- c = centre - eye, |c|^2, |c|, the power pre-multiplied by 3/pi * phase /
    // r^2 * exp(-sigma_t |c|) -- and per ray exp(-sigma_t s) = exp(-sigma_t |c|) * exp(sigma_t (|c| - s)) with
    // sigma_t (|c| - s) <= sigma_t r^2 / |c| ~ 1e-5, so the second factor is 1 + x to fp32 precision
    // (src/photon.rs:474-493: k2(d^2/r^2)/r^2 with k2(x) = 3/pi (1-x)^2).
    // Two kinds of photons: PLAIN ones lie in front of every ray of the pixel and nearer to the eye than every hit, so
    // their test is the radius test alone (as a clamp of the kernel's argument: no comparison at all); GUARDED ones --
    // behind the eye's plane or inside the shell between the nearest and the farthest hit -- take the full test.
    auto flush = [&](const uint32_t* list, uint32_t nb, auto counting, auto plain) {
        if (lane_ < nb) {
            const PhotonRec raw = q.v_ph[list[lane_]];
            const V c = xyz(raw.pos_r) - eye0;
            const float r2 = raw.pos_r.w * raw.pos_r.w, ir2 = rcp(r2), c2 = dot(c, c), len = sqrt1(c2);
            const float kk = (3.f * kInvPi) * phase * ir2 * __expf(-sigma_t * len);
            const V pw = kk * xyz(raw.pow);
            const float one_plus = fmaf(sigma_t, len, 1.f);
            float wsum = 0.f;   // the photon's weights over the pixel's rays: its power multiplies their sum once
            uint32_t n_ok = 0;
            if (!(q.skip & 4u)) {  // diagnostic: 4 = tree walk, culling and photon preparation only
#pragma unroll 4
                for (uint32_t s = 0; s < n_s; s++) {
                    const float4 ray = rays[s];
                    const V rd = mk(ray.x, ray.y, ray.z);
                    const float disk = dot(c, rd);
                    const V dv = fma3(disk, rd, -c);
                    const float dist2 = dot(dv, dv);
                    if (decltype(plain)::value) {
                        const float tmp = fmaxf(fmaf(-dist2, ir2, 1.f), 0.f);   // 0 from the radius on
                        wsum = fmaf(tmp * tmp, fmaf(-sigma_t, disk, one_plus), wsum);
                        if (decltype(counting)::value) n_ok += tmp > 0.f ? 1u : 0u;
                    } else {
                        const bool ok = disk > 0.f && dist2 < r2 && c2 <= ray.w;
                        const float tmp = fmaf(-dist2, ir2, 1.f);
                        const float w = tmp * tmp * fmaf(-sigma_t, disk, one_plus);
                        wsum += ok ? w : 0.f;
                        if (decltype(counting)::value) n_ok += ok ? 1u : 0u;
                    }
                }
            }
            beam_sum = fma3(wsum, pw, beam_sum);
            if (decltype(counting)::value) { atomicAdd(&a.counters[5], (unsigned long long)n_s); atomicAdd(&a.counters[6], (unsigned long long)n_ok); }
        }
    };
    auto flush_batch = [&](const uint32_t* list, uint32_t nb, auto plain) {
        if (a.counters) flush(list, nb, std::true_type{}, plain);
        else flush(list, nb, std::false_type{}, plain);
    };
    auto prefix = [](uint64_t m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); };
    // a list has received `cnt` more entries: a full batch of 64 goes through the rays, the rest moves to the front
    auto drain = [&](uint32_t* list, uint32_t& pend, auto plain) {
        __builtin_amdgcn_wave_barrier();
        if (pend >= 64u) {
            flush_batch(list, 64u, plain);
            __builtin_amdgcn_wave_barrier();
            const bool mv = lane_ + 64u < pend;
            uint32_t v = 0u;
            if (mv) v = list[64u + lane_];
            __builtin_amdgcn_wave_barrier();
            if (mv) list[lane_] = v;
            pend -= 64u;
            __builtin_amdgcn_wave_barrier();
        }
    };
    uint32_t* const guard_list = pend_list + kPendCap;
    uint32_t pend = 0, pend_g = 0;   // wave-uniform: entries of pend_list / guard_list
    for (uint32_t base = 0; base < cand_n; base += 64u) {  // wave-uniform loop
        bool take = false, plain = false;
        uint32_t idx = 0u;
        if (base + lane_ < cand_n) {
            idx = cand[base + lane_];
            const F4 pr = q.v_ph[idx].pos_r;
            take = !have_fs || !sphere_outside(fs, pr);
            const V cc = xyz(pr) - eye0;
            const float cc2 = dot(cc, cc), along = dot(cc, fs.axis);
            take = take && cc2 <= far2;  // the per-ray test rejects centres beyond the ray's hit
            // in front of every ray of the pixel (within 89.4 degrees of its axis; the pixel's cone is ~1e-3 wide)
            plain = have_fs && cc2 <= near2 && along > 0.f && along * along > 1e-4f * cc2;
        }
        const uint64_t tp = __ballot(take && plain), tg = __ballot(take && !plain);
        if (take && plain) pend_list[pend + prefix(tp)] = idx;
        if (take && !plain) guard_list[pend_g + prefix(tg)] = idx;
        pend += uint32_t(__popcll(tp));
        pend_g += uint32_t(__popcll(tg));
        drain(pend_list, pend, std::true_type{});
        drain(guard_list, pend_g, std::false_type{});
    }
    // what is left of both lists goes through the full test together (a batch costs the same whatever it holds)
    if (lane_ < pend) guard_list[pend_g + lane_] = pend_list[lane_];
    pend_g += pend;
    __builtin_amdgcn_wave_barrier();
    if (pend_g) flush_batch(guard_list, min(pend_g, 64u), std::false_type{});
    if (pend_g > 64u) flush_batch(guard_list + 64, pend_g - 64u, std::false_type{});
    return beam_sum;
}

// ---- surface estimate (src/photon.rs:327-375)
// One lane's surface point and its estimate in the making.
struct SurfaceSample {
    V x, n, wo;
    Mat mat;
    V sc_col;       // emission + the terms of the gathered photons so far
    float max_d2;   // squared distance of the K-th nearest photon, once known
    bool todo;      // still to be served
    uint32_t* em;   // EMIT: this sample's column of QueryArgs::emit (entries `stride` dwords apart)
    uint32_t n_em;  // ... and the photons written to it so far
};
// The room shell's box, a few ulps wider (see `lane_clear` in gather_serve).
struct ShellBox {
    bool on;
    F4 lo, hi;
    RPT_DEV bool holds(V p) const {
        return !on || (p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z);
    }
};
template <bool BVH>
RPT_DEV ShellBox shell_box() {
    const auto& sc = *kernarg_scene();
    ShellBox b{!BVH && sc.has_shell != 0u, F4{}, F4{}};
    if (b.on) {
        const ShellScan sh = uload(sc.shell);
        const float e = 2e-6f * fmaxf(max3(fabsf(sh.lo.x), fabsf(sh.lo.y), fabsf(sh.lo.z)), max3(fabsf(sh.hi.x), fabsf(sh.hi.y), fabsf(sh.hi.z)));
        b.lo = F4{sh.lo.x - e, sh.lo.y - e, sh.lo.z - e, 0.f};
        b.hi = F4{sh.hi.x + e, sh.hi.y + e, sh.hi.z + e, 0.f};
    }
    return b;
}
// The wave-private LDS of the gathers: per-lane [K][64] distance (and index) lists, then the wave-level gather's stack,
// candidate keys and (position, index) records.
struct GatherLds {
    float* gd;
    uint32_t* gi;
    uint32_t* pstack;
    float* keys;
    F4* cl;
};
// What a scene's traversal needs besides the ray: the LDS stack column and two diagnostic counters.
struct WalkScratch {
    uint32_t* stk;
    uint32_t c0, c1;
};
// One gathered photon's term (src/photon.rs:357-371).  Visibility ("something lies between the photon and the query
// point"): only a hit closer than the query point can block, and every point of the segment photon -> x lies within the
// gather radius of x: scanned records whose box misses that ball (of any sample of this pixel: the mask is wave-uniform)
// cannot decide the test and are skipped.  The closest hit below |disp| (1 - 1e-3) is the closest hit of the unbounded
// query whenever that one would block, so the decisions are the same as with the full scan.
// no_scan (wave-uniform): no lane needs the scan; lane_free: this lane's test cannot be blocked whatever a scan finds.
template <bool BVH>
RPT_DEV void add_photon_term(QueryK q, const SceneView& sc_arg, SurfaceSample& s, WalkScratch& ws, V po, V pdir, V ppow,
                             uint64_t vis_mask, bool no_scan, bool lane_free) {
    V disp = s.x - po;
    float len2 = dot(disp, disp);
    float ilen = rsq(len2);
    V pd = ilen * disp;
    float len = len2 * ilen;
    float ts = BVH ? kInf : len * (1.f - 1e-3f);
    uint32_t cs = CODE_MISS, is = 0;
    if (!(q.skip & 8u) && !no_scan) {  // diagnostic: 8 = no visibility scans
        if (BVH) closest_hit<2, false>(sc_arg, po, pd, ray_tmin_p(po), ts, cs, is, ws.stk, 256, ws.c0, ws.c1);
        else scan_prims<true>(sc_arg, po, pd, ray_tmin_p(po), ts, cs, vis_mask);
    }
    // A hit inside the query point's own tangent plane is the grazing ray meeting its own surface: fp64
    // rejects it as parallel (|cos| < 1e-8); fp32 would place it at a random t.  Not an occluder.
    V hp = fma3(ts, pd, po) - s.x;
    bool own_plane = fabsf(dot(hp, s.n)) <= 1e-4f * len;
    bool blocked = !lane_free && cs != CODE_MISS && !own_plane && ts < len * (1.f - 1e-3f);
    if (!blocked || !(len2 > 0.f)) {  // (a query point that coincides with the photon has no ray to trace)
        float c = fminf(fmaxf(dot(pdir, s.n), 0.f), 1.f);
        s.sc_col = fma3(c, bsdf(s.mat, s.n, s.wo, pdir) * ppow, s.sc_col);
    }
}
// Every member lane picks its K nearest out of the M ordered candidates (centre: where `rho` is measured from) inside its
// search radius `guess`, then sums the terms of the photons within its K-th distance, in candidate order.  Lanes that
// found K are done (s.todo, s.max_d2).
template <bool BVH, bool EMIT>
RPT_DEV bool gather_serve(QueryK q, const SceneView& sc_arg, const GatherLds& l, const ShellBox& shell, SurfaceSample& s,
                          WalkScratch& ws, bool member, float guess, float rho, uint32_t M) {
    const auto& a = q.r;
    const auto& sc = a.sc;
    const uint32_t lane_ = threadIdx.x & 63u;
    const uint32_t K = q.gather_size, want_k = min(K, q.n_s);
    float* const gd = l.gd;
    const float* const keys = l.keys;
    const F4* const cl = l.cl;
    const V x = s.x;
    unsigned long long tk = a.counters ? __builtin_amdgcn_s_memtime() : 0ull;   // (diagnostic: clock ticks of selection, mask, terms)
    auto lap = [&](int k) { if (a.counters) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); diag_add(a.counters, 24 + k, t1 - tk); tk = t1; } };
    // -- each member's K nearest distances (list in LDS; entries beyond `guess` do not count)
    uint32_t found = 0, wslot = 0;
    float worst = 0.f;
    const float reach0 = __builtin_sqrtf(guess) + rho;
    float thr = member ? reach0 * reach0 * (1.f + 1e-5f) : -1.f;
    float k_next = M ? keys[0] : 0.f;
    F4 c_next = M ? cl[0] : F4{0.f, 0.f, 0.f, 0.f};
    uint32_t j = 0;
    // The first K candidates fill the lists: as long as every member takes every one of them (the usual case) nothing has
    // to be decided per lane, and the largest entry is known when the list is full.
    for (; j < K && j < M; j++) {
        const float kj = k_next;
        const V d = xyz(c_next) - x;
        const float d2 = dot(d, d);
        if (__ballot(kj <= thr) == 0ull || __ballot(member && d2 > guess) != 0ull) break;   // (the general loop goes on from here)
        {
            const uint32_t jn = min(j + 1u, M - 1u);
            k_next = keys[jn];
            c_next = cl[jn];
        }
        if (a.counters) diag_add(q.r.counters, 13, 1ull);
        if (member) {
            gd[j * 64u] = d2;
            if (d2 > worst) { worst = d2; wslot = j; }
        }
    }
    if (member) found = j;
    if (member && found == K) {
        const float reach = __builtin_sqrtf(worst) + rho;
        thr = reach * reach * (1.f + 1e-5f);
    }
    for (; j < M; j++) {
        const float kj = k_next;
        const F4 cj = c_next;
        {   // the next candidate's LDS reads are in flight while this one is handled
            const uint32_t jn = min(j + 1u, M - 1u);
            k_next = keys[jn];
            c_next = cl[jn];
        }
        if (__ballot(kj <= thr) == 0ull) break;   // no member's ball reaches this far from the centre
        if (a.counters) diag_add(q.r.counters, 13, 1ull);
        const V d = xyz(cj) - x;
        const float d2 = dot(d, d);
        bool changed = false;
        if (member) {
            if (found < K) {
                if (d2 <= guess) {
                    gd[found * 64u] = d2;
                    found++;
                    changed = found == K;
                }
            } else if (d2 < worst) {
                gd[wslot * 64u] = d2;
                changed = true;
            }
            if (changed) {   // (re)locate the current worst: four independent LDS reads per step
                worst = -1.f;
                uint32_t k = 0;
                for (; k + 4u <= K; k += 4u) {
                    const float v0 = gd[k * 64u], v1 = gd[(k + 1u) * 64u], v2 = gd[(k + 2u) * 64u], v3 = gd[(k + 3u) * 64u];
                    const float m01 = fmaxf(v0, v1), m23 = fmaxf(v2, v3), m = fmaxf(m01, m23);
                    if (m > worst) {
                        worst = m;
                        wslot = k + (m == m01 ? (m == v0 ? 0u : 1u) : (m == v2 ? 2u : 3u));
                    }
                }
                for (; k < K; k++) {
                    const float v = gd[k * 64u];
                    if (v > worst) { worst = v; wslot = k; }
                }
                const float reach = __builtin_sqrtf(worst) + rho;
                thr = reach * reach * (1.f + 1e-5f);
            }
        }
        if (a.counters && __ballot(changed) != 0ull) diag_add(q.r.counters, 14, 1ull);
    }
    const bool ok = member && found >= want_k;
    float r2k = found == K ? worst : 0.f;
    if (ok && found < K) for (uint32_t k = 0; k < found; k++) r2k = fmaxf(r2k, gd[k * 64u]);   // (a map of fewer than K photons)
    if (!ok) r2k = 0.f;
    lap(7);
    // -- the terms of the photons within each lane's radius, in candidate order
    uint64_t vis_mask = ~0ull;
    bool touched = true;   // some scanned record comes near this lane's ball
    if (!BVH && !EMIT) vis_mask = scan_mask_for_ball(sc_arg, ok, x, __builtin_sqrtf(r2k) * (1.f + 1e-4f) + 1e-6f, &touched);   // (EMIT: no ray is traced here)
    // A lane's test cannot be blocked when no scanned record comes near its ball, the scene has no plane, and both
    // ends of the segment lie inside the room shell: its faces bound a convex box, a segment between two points of
    // the closed box meets a face at its ends only, and those the search interval leaves out.  (The scan itself is
    // less exact there: next to an edge of the room it reports false crossings, see tools/photon_vis_check.py.)
    // Then the gathered photon is visible by construction; the scan runs only if some lane of the term needs it.
    // A point counts as inside within a few ulps.  The decision is the lane's own: no other lane's geometry enters.
    bool lane_clear = !BVH && !touched && sc.n_pln == 0u && !(q.skip & 128u);   // (diagnostic: 128 = always scan)
    lane_clear = lane_clear && shell.holds(x);
    const float reach2 = __builtin_sqrtf(r2k) + rho;
    const float thr2 = ok ? reach2 * reach2 * (1.f + 1e-5f) : -1.f;
    bool more = __ballot(ok) != 0ull && !(q.skip & 16u);   // diagnostic: 16 = no second pass
    const float thr2_max = wave_max(thr2);
    // Most terms need no visibility scan and sit on diffuse surfaces: what bsdf() returns for a Lambertian surface whenever
    // n.wi >= 0 is a constant of the lane (src/material.rs:268-275; for n.wi < 0 the cosine factor is zero anyway), and the
    // term is cos * (albedo / pi) * power with nothing of the ray to the photon in it.
    const bool lambert = s.mat.kind == M_LAMBERTIAN;
    const V f_diffuse = (lambert && !__builtin_signbit(dot(s.n, s.wo))) ? kInvPi * s.mat.albedo : mk(0.f, 0.f, 0.f);
    const bool all_diffuse = __ballot(ok && !lambert) == 0ull;   // wave-uniform
    lap(8);
    for (uint32_t base = 0; base < M && more; base += 64u) {
        // lane l fetches what the term needs of candidate base + l; the records are then handed round by readlane
        F4 fdir{}, fpow{};
        if (!EMIT && base + lane_ < M && keys[base + lane_] <= thr2_max) {
            const uint32_t idx = __float_as_uint(cl[base + lane_].w) & ~kCandInShell;
            fdir = q.s_ph[idx].dir;
            fpow = q.s_ph[idx].pow;
        }
        const uint32_t nb = min(64u, M - base);
        float k_nx = keys[base];
        F4 c_nx = cl[base];
        for (uint32_t jj = 0; jj < nb; jj++) {
            const uint32_t j = base + jj;
            const float kj = k_nx;
            const V po = xyz(c_nx);
            const bool po_in = (__float_as_uint(c_nx.w) & kCandInShell) != 0u;   // (wave-uniform) the photon lies inside the room shell
            const uint32_t pidx = __float_as_uint(c_nx.w) & ~kCandInShell;
            {
                const uint32_t jn = min(j + 1u, M - 1u);
                k_nx = keys[jn];
                c_nx = cl[jn];
            }
            if (__ballot(kj <= thr2) == 0ull) { more = false; break; }
            if (a.counters) diag_add(q.r.counters, 15, 1ull);
            const V dd = po - x;
            const bool in = ok && dot(dd, dd) <= r2k;
            if (__ballot(in) == 0ull) continue;
            if constexpr (EMIT) {   // the selection is handed over instead of being evaluated (at most K: ties at the K-th distance are cut)
                if (in && s.n_em < K) {
                    s.em[size_t(s.n_em) * a.iterations] = pidx;
                    s.n_em++;
                }
                continue;
            }
            auto rl = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), jj)); };
            const V pdir = mk(rl(fdir.x), rl(fdir.y), rl(fdir.z)), ppow = mk(rl(fpow.x), rl(fpow.y), rl(fpow.z));
            if (a.counters) diag_add(q.r.counters, 16, 1ull);
            const bool lane_free = lane_clear && po_in;
            const bool no_scan = __ballot(in && !lane_free) == 0ull;
            if (a.counters && no_scan) diag_add(q.r.counters, 9, 1ull);
            if (no_scan && all_diffuse) {   // (wave-uniform) visible by construction, diffuse: the term itself
                if (in) {
                    const float c = fminf(fmaxf(dot(pdir, s.n), 0.f), 1.f);
                    s.sc_col = fma3(c, f_diffuse * ppow, s.sc_col);
                }
            } else if (in) {
                add_photon_term<BVH>(q, sc_arg, s, ws, po, pdir, ppow, vis_mask, no_scan, lane_free);
            }
        }
    }
    lap(9);
    if (ok) {
        s.max_d2 = r2k;
        s.todo = false;
    }
    return ok;
}
// The candidates of the pixel's hit points, when they form one cluster: every photon within R of c, ordered by distance to c
// (valid); c = the pixel's first surface point whenever it has one (have_c), which also orders the lists of the later rounds.
struct PixelList {
    bool valid, have_c;
    uint32_t M;
    float R;
    V c;
};
// The wave's anchor (ball_anchor): node, and the ball it was computed for.
struct Anchor {
    V c;
    float R;
    uint32_t node;
};
RPT_DEV void anchor_for(QueryK q, Anchor& anc, V c, float R) {
    const V da = c - anc.c;
    if (!(anc.R > 0.f) || __builtin_sqrtf(dot(da, da)) + R > anc.R) {   // kept while the queries stay inside a ball twice as wide
        anc.c = c;
        anc.R = 2.f * R;
        anc.node = ball_anchor(q.s_nodes, q.n_s, c, anc.R * anc.R);
        if (q.r.counters) diag_add(q.r.counters, 17, 1ull);
    }
}
// Once per pixel: collect and order the candidates of all its samples (prho2: how far, squared, this lane's surface points
// lie from pix.c; prev_r2: the lane's last gather radius, squared).
RPT_DEV void pixel_candidates(QueryK q, const GatherLds& l, const ShellBox& shell, PixelList& pix, Anchor& anc, float prev_r2, float prho2) {
    const float G = wave_max(prev_r2 > 0.f ? 2.f * prev_r2 : 0.f);
    const float rho_max2 = wave_max(prho2);
    if (!(G > 0.f && rho_max2 <= G)) return;
    pix.R = (__builtin_sqrtf(G) + __builtin_sqrtf(rho_max2)) * (1.f + 1e-5f);
    anchor_for(q, anc, pix.c, pix.R);
    uint32_t steps = 0;
    pix.M = ball_collect(q.s_nodes, q.s_ph, q.n_s, anc.node, pix.c, pix.R * pix.R, l.pstack, reinterpret_cast<uint32_t*>(l.keys), l.cl,
                         q.coop_cap, steps, shell.on, shell.lo, shell.hi);
    if (q.r.counters) { diag_add(q.r.counters, 10, (unsigned long long)(steps)); if (pix.M == kCoopOverflow) diag_add(q.r.counters, 12, 1ull); else diag_add(q.r.counters, 11, (unsigned long long)(pix.M)); }
    if (pix.M != kCoopOverflow) {
        if (!(q.skip & 64u)) sort_candidates(l.cl, l.keys, pix.M, pix.c);   // (diagnostic: 64 = collection only)
        pix.valid = true;
    }
}
// The surface gather of one trip by the wave together.  Consecutive samples of a lane fall within a pixel of each other: the
// previous gather radius (squared, doubled) bounds this search; a lane that finds fewer than K photons inside it searches
// again with a larger one.  Round 0 serves the lanes from the pixel's candidate list (a lane's ball has to lie inside the
// collected one); the later rounds collect for clusters of the query points that are left: none, unless the pixel straddles
// an edge, a radius was too small or there is no pixel list.  Lanes it cannot serve keep s.todo.
template <bool BVH, bool EMIT>
RPT_DEV void surface_gather_wave(QueryK q, const SceneView& sc_arg, const GatherLds& l, const ShellBox& shell, SurfaceSample& s,
                                 WalkScratch& ws, PixelList& pix, Anchor& anc, float prev_r2) {
    const auto& a = q.r;
    const V x = s.x;
    float guess = prev_r2 > 0.f ? 2.f * prev_r2 : 0.f;
    {   // a lane without a radius of its own borrows the largest one in the wave
        const float g = wave_max(s.todo ? guess : 0.f);
        if (!(guess > 0.f)) guess = g;
    }
    if (a.counters && __ballot(s.todo) != 0ull) diag_add(q.r.counters, 8, 1ull);
    for (uint32_t round = pix.valid ? 0u : 1u; round < 7u; round++) {
        const uint64_t cm = __ballot(s.todo && guess > 0.f);
        if (cm == 0ull) break;
        bool member;
        float g_use, rho;
        uint32_t M;
        if (round == 0u) {
            const V dx = x - pix.c;
            rho = __builtin_sqrtf(dot(dx, dx));
            const float room = pix.R * (1.f - 2e-5f) - rho;
            g_use = fminf(guess, room > 0.f ? room * room : 0.f);
            member = s.todo && g_use > 0.f;
            M = pix.M;
        } else {
            pix.valid = false;   // (these rounds reuse the list's LDS)
            const uint32_t lead = uint32_t(__ffsll((unsigned long long)cm)) - 1u;
            auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lead)); };
            const V xc = mk(bc(x.x), bc(x.y), bc(x.z));
            const float gc = bc(guess);
            const V dx = x - xc;
            const float rho2 = dot(dx, dx);
            member = s.todo && guess > 0.f && rho2 <= gc;   // within the leader's own search radius
            rho = __builtin_sqrtf(rho2);
            g_use = guess;
            const float R = wave_max(member ? __builtin_sqrtf(guess) + rho : 0.f) * (1.f + 1e-5f);
            anchor_for(q, anc, xc, R);
            uint32_t steps = 0;
            M = ball_collect(q.s_nodes, q.s_ph, q.n_s, anc.node, xc, R * R, l.pstack, reinterpret_cast<uint32_t*>(l.keys), l.cl, q.coop_cap, steps,
                             shell.on, shell.lo, shell.hi);
            if (a.counters) { diag_add(q.r.counters, 10, (unsigned long long)(steps)); if (M == kCoopOverflow) diag_add(q.r.counters, 12, 1ull); else diag_add(q.r.counters, 11, (unsigned long long)(M)); }
            if (M == kCoopOverflow) break;   // the lanes still to do search one by one
            if (q.skip & 64u) { if (member) { s.max_d2 = 0.5f * guess; s.todo = false; } continue; }   // diagnostic: collection only
            // ordered, like the pixel's list, by distance to the pixel's first surface point when there is one: the order
            // of a lane's terms is then the same whichever round serves it
            const V kc = pix.have_c ? pix.c : xc;
            sort_candidates(l.cl, l.keys, M, kc);
            const V dk = x - kc;
            rho = __builtin_sqrtf(dot(dk, dk));
        }
        if (q.skip & 96u) { if (member) { s.max_d2 = 0.5f * guess; s.todo = false; } continue; }   // diagnostic: 32 / 64 = no selection
        const bool ok = gather_serve<BVH, EMIT>(q, sc_arg, l, shell, s, ws, member, g_use, rho, M);
        if (member && !ok) guess = fmaxf(guess, 4.f * g_use);   // too few photons inside: twice the radius next round
        __builtin_amdgcn_wave_barrier();
    }
}
// One search per lane (lists in global memory, no radius to start from, an overfull candidate list); the lanes of the call
// search the tree together (knn_walk_wave).
template <bool BVH, bool EMIT>
RPT_DEV void surface_gather_lane(QueryK q, const SceneView& sc_arg, const GatherLds& l, const ShellBox& shell, SurfaceSample& s,
                                 WalkScratch& ws, float prev_r2) {
    const auto& sc = q.r.sc;
    const uint32_t want_k = min(q.gather_size, q.n_s);
    uint32_t found = 0;
    if (prev_r2 > 0.f) found = gather_knn<true>(q.s_nodes, q.s_ph, q.n_s, s.x, q.gather_size, l.gd, l.gi, s.max_d2, 2.f * prev_r2);
    if (found < want_k || !(prev_r2 > 0.f)) found = gather_knn<true>(q.s_nodes, q.s_ph, q.n_s, s.x, q.gather_size, l.gd, l.gi, s.max_d2);
    uint64_t vis_mask = ~0ull;
    bool touched = true;
    if (!BVH) vis_mask = scan_mask_for_ball(sc_arg, true, s.x, __builtin_sqrtf(s.max_d2) * (1.f + 1e-4f) + 1e-6f, &touched);
    const bool lane_clear = !BVH && !touched && sc.n_pln == 0u && !(q.skip & 128u) && shell.holds(s.x);   // as in gather_serve
    // The K nearest come out of the search in an order that depends on the walk (which other lanes were live, where the
    // previous gather's radius let it start): the terms are summed in photon-index order instead, a function of the set alone.
    for (uint32_t i = 1; i < found; i++) {
        const uint32_t v = l.gi[i * 64u];
        uint32_t j = i;
        for (; j > 0u && l.gi[(j - 1u) * 64u] > v; j--) l.gi[j * 64u] = l.gi[(j - 1u) * 64u];
        l.gi[j * 64u] = v;
    }
    if constexpr (EMIT) {
        for (uint32_t k = 0; k < found; k++) s.em[size_t(k) * q.r.iterations] = l.gi[k * 64u];
        s.n_em = found;
        s.todo = false;
        return;
    }
    for (uint32_t k = 0; k < found; k++) {
        const PhotonRec ph = q.s_ph[l.gi[k * 64u]];
        const bool lane_free = lane_clear && shell.holds(xyz(ph.pos_r));
        add_photon_term<BVH>(q, sc_arg, s, ws, xyz(ph.pos_r), xyz(ph.dir), xyz(ph.pow), vis_mask, __ballot(!lane_free) == 0ull, lane_free);
    }
    s.todo = false;
}

#ifndef RPT_MIN_WAVES_QUERY
#define RPT_MIN_WAVES_QUERY 4   // waves per SIMD the camera-pass kernels are compiled for (128 VGPRs)
#endif
// get_color_with_photon_map / PhotonMap::estimate_indirect for the point-beam map
// (src/photon.rs:950-985, 316-375, 439-502, 595-628).  LDS: per lane gather_size (d2, index) pairs.
// KIND: the PhotonRenderKind of the map (RPT_PHOTON_*).  One instantiation per kind: the three estimators share the
// camera ray and the surface gather, but each drags its own register and scratch needs along (the point x point
// volume gather keeps a 64-entry per-lane stack), which a run-time switch makes every kind pay.
// PHASE: 0 = the whole estimate in one launch; 1 / 2 = the volume estimate / the surface estimate alone, two launches over the
// same work items whose partial sums go to two slabs that resolve_kernel adds (beam kinds in a medium, lists in LDS): each
// kernel then keeps only its own estimate's wave-uniform state, and neither needs scratch memory.
// EMIT (reference-epsilon mode): the surface estimate stops at the selection -- each sample's K nearest photons, their number and
// the K-th squared distance go to q.emit, and photon_surface_f64_kernel (kernels_f64.hip) traces the visibility rays and adds the
// terms with the reference's arithmetic; this kernel's slab then holds the volume estimate (and the background) alone.
template <bool MEDIUM, bool BVH, bool GG, int KIND, int PHASE = 0, bool EMIT = false>
__global__ __launch_bounds__(256, RPT_MIN_WAVES_QUERY) void photon_query_kernel(const QueryArgs q) {
    static_assert(!EMIT || PHASE == 0, "the selection is handed over by the one-launch camera pass");
    static_assert(PHASE == 0 || (MEDIUM && !GG && KIND != RPT_PHOTON_MAP), "the split camera pass: beam estimates in a medium, gather lists in LDS");
    extern __shared__ uint32_t dyn_lds[];
    const RenderArgs& a = q.r;
    const SceneView& sc = a.sc;
    const SceneView& sc_arg = a.sc;   // what the device functions are handed (they read the view from the kernarg segment themselves)
    uint32_t* stk = BVH ? (dyn_lds + threadIdx.x) : nullptr;
    const uint32_t K = max(q.gather_size, q.gather_size_volume);  // LDS columns are sized for the larger gather
    // One wave-private LDS region (after the BVH stack, q.region_dwords each) serves phases that never overlap in time
    // within a wave (dword offsets):
    //   surface gather      [0, K*64) per-lane distance lists, then either [K*64, 2*K*64) per-lane index lists (one search
    //                       per lane) or the wave-level gather's ball-walk stack [kBallStack], candidate keys [coop_cap]
    //                       and (position, index) records [4 * coop_cap] -- the pixel's candidate list lives there from
    //                       pixel_candidates to the pixel's last trip;
    //   beam, photon/lane   [0, 4*kSuper) the pixel's rays, then the two lists of culled candidates [2 * kPendCap];
    //   beam, sample/lane   [0, kBeamCap) the walk's pending entries, then 64 staging slots of 16 dwords.
    const uint32_t lane_ = threadIdx.x & 63u, wave_ = threadIdx.x >> 6;
    uint32_t* region = dyn_lds + (BVH ? 32u * 256u : 0u) + wave_ * q.region_dwords;
    // GG: the lists of a gather larger than kGatherLds live in a per-wave global-memory region, same [k][lane] layout
    uint32_t* const lists = GG ? q.gather + size_t(blockIdx.x * 4u + wave_) * (size_t(K) * 128u) : region;
    float* gd = reinterpret_cast<float*>(lists) + lane_;
    uint32_t* gi = lists + K * 64u + lane_;
    uint32_t* wstack = region;
    F4* stage = reinterpret_cast<F4*>(region + kBeamCap);
    const float sigma_t = sc.sigma_a + sc.sigma_s;
    const V mcol0 = mk(sc.medium_color[0], sc.medium_color[1], sc.medium_color[2]);  // medium.color(dummy_pos = 0)

    uint32_t c0 = 0, c1 = 0;
    float prev_r2 = 0.f;  // squared radius of this lane's previous surface gather
    auto tick = [&]() { return a.counters ? __builtin_amdgcn_s_memtime() : 0ull; };
    Anchor anc{mk(0, 0, 0), 0.f, 0u};   // wave-uniform: the surface gather's anchor (ball_anchor)
    // Work decomposition of the camera pass: a wave takes a strip of rows of one 8x8 pixel block and one chunk of up
    // to kSuper samples at a time and walks through the strip's pixels; a pixel's samples are handled 64 at a time:
    // in each trip the 64 LANES ARE SAMPLES OF ONE PIXEL.  The rays of a trip then differ only by their sub-pixel
    // jitter, so the k-nearest walks of the lanes follow the same path through the tree, and pixel, NDC coordinates
    // and camera basis are wave-uniform.  The sample values of a pixel are summed per lane over the trips, then
    // across the wave in a fixed butterfly order, and stored as that (pixel, chunk)'s partial sum.
    //
    // Beam x point estimate with a pinhole camera: the photon spheres that reach into the STRIP's frustum are found
    // once per work item (one tree walk) and kept in a per-wave list in global memory.  For a pixel, the rays of
    // all its samples are generated first and parked in LDS (direction + squared hit distance, 16 B each); the list
    // is culled against the pixel's own frustum, and then the roles turn around: ONE PHOTON PER LANE, each lane
    // holding what depends on its photon and the eye in registers and looping over the pixel's rays (one broadcast
    // LDS read per test).  The estimate is a sum over (ray, photon) pairs either way; with the samples in the lanes
    // every test read a 48-byte staged record from LDS (the LDS pipe was as busy as the VALU) and the list was culled
    // and staged once per 64 samples instead of once per pixel.
    uint32_t* const cand = q.cand_cap ? q.cand + size_t(blockIdx.x * 4u + wave_) * q.cand_cap : nullptr;
    const bool cand_mode = PHASE != 2 && MEDIUM && KIND == RPT_PHOTON_POINT_BEAM && q.cand_cap != 0u && a.cam.aperture <= 0.f && !(q.skip & 1u);
    uint32_t cand_n = 0;       // wave-uniform
    bool cand_valid = false;   // wave-uniform: the list describes the strip this wave is working on
    const uint32_t n_blocks64 = a.n_owned >> 6;  // 8x8 blocks owned by this rank
    uint32_t pi = 0u, pi_end = 0u, blk = 0, chunk = 0, x0 = 0, y0 = 0, n_s = 0;  // wave-uniform: pixel cursor within the block, item
    const V cam_right = mk(a.cam.right[0], a.cam.right[1], a.cam.right[2]), cam_up = mk(a.cam.up[0], a.cam.up[1], a.cam.up[2]);
    float4* const rays = reinterpret_cast<float4*>(region);   // [kSuper] (direction, squared hit distance) of the pixel's samples
    uint32_t* const pend_list = region + kSuper * 4u;         // [2][kPendCap] culled candidates (plain, guarded) waiting for a full batch
    for (;;) {
        // the kernel's arguments are read per trip, where they are used (kernarg_scene in device_core.h): held in scalar
        // registers since kernel entry they do not fit, and the overflow lives in VGPR lanes
        const auto& q = *kernarg_args<QueryArgs>();
        const auto& a = q.r;
        const auto& sc = a.sc;
        if (pi == pi_end) {  // next work item (wave-uniform)
            unsigned long long got = ~0ull;
            if (lane_ == 0) got = atomicAdd(a.queue, 1ull);
            const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(got));
            const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(got >> 32));
            if (hi != 0 || lo >= a.n_items) break;
            const uint32_t per_chunk = n_blocks64 * q.parts;
            chunk = lo / per_chunk;
            const uint32_t rem = lo - chunk * per_chunk;
            blk = rem / q.parts;
            const uint32_t part = rem - blk * q.parts, per_part = 64u / q.parts;   // whole rows of the block
            const uint32_t tile = a.tiles[blk >> 4], sb = blk & 15u;
            const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
            x0 = __builtin_amdgcn_readfirstlane(tx * 32u + (sb & 3u) * 8u);
            y0 = __builtin_amdgcn_readfirstlane(ty * 32u + (sb >> 2) * 8u);
            n_s = min(kSuper, a.iterations - chunk * kSuper);
            pi = __builtin_amdgcn_readfirstlane(part * per_part);
            pi_end = pi + per_part;
            cand_valid = false;
            // What a lane carries from one pixel's surface gather to the next -- its last gather radius, the wave's anchor -- starts
            // afresh with the work item: which item a wave had before depends on the work queue, and the radius decides in which
            // round, hence in which order, a sample's photon terms are added.  (photon_skip bit 2048 keeps them: diagnostic.)
            if (!(q.skip & 2048u)) {
                prev_r2 = 0.f;
                anc = Anchor{mk(0, 0, 0), 0.f, 0u};
            }
            if (cand_mode) {
                // the four corner directions of the strip (footprints included): every sample ray lies between them
                const float e = a.inv_dim * 1.0001f;
                const uint32_t r0 = pi >> 3, r1 = (pi_end - 1u) >> 3;
                const float xl = (float(2u * x0 + 1u) - float(a.width)) * a.inv_dim - e;
                const float xh = (float(2u * (x0 + 7u) + 1u) - float(a.width)) * a.inv_dim + e;
                const float yh = (float(2u * (a.height - (y0 + r0)) - 1u) - float(a.height)) * a.inv_dim + e;
                const float yl = (float(2u * (a.height - (y0 + r1)) - 1u) - float(a.height)) * a.inv_dim - e;
                const V eye = mk(a.cam.eye[0], a.cam.eye[1], a.cam.eye[2]);
                const V dd = mk(a.cam.ddir[0], a.cam.ddir[1], a.cam.ddir[2]);
                const V corners[4] = {dd + xl * cam_right + yl * cam_up, dd + xh * cam_right + yl * cam_up,
                                      dd + xl * cam_right + yh * cam_up, dd + xh * cam_right + yh * cam_up};
                Frustum fr;
                if (frustum_from_dirs(true, eye, corners, 4, fr)) {
                    uint32_t n_list = 0;  // wave-uniform
                    frustum_walk<true>(q.v_nodes, q.v_ph, q.n_v, fr, wstack, q.overflow, [&](bool leaf, uint32_t idx, const PhotonRec&) {
                        const uint64_t lm = __ballot(leaf);
                        if (leaf) {
                            const uint32_t slot = n_list + __builtin_amdgcn_mbcnt_hi(uint32_t(lm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(lm), 0u));
                            if (slot < q.cand_cap) cand[slot] = idx;
                        }
                        n_list += uint32_t(__popcll(lm));
                    });
                    cand_n = n_list;
                    cand_valid = n_list <= q.cand_cap;  // an overfull list is dropped: those pixels walk the tree
                    // The walk lists the strip's photons in the order its lanes happen to reach them.  A pixel's beam terms are
                    // summed in list order, so the list is put into photon-index order (bitonic sort in the wave's LDS region,
                    // free at this point): a pixel's sum then no longer depends on how the block was cut into strips.
                    if (cand_valid && n_list > 1u && n_list <= q.region_dwords && !(q.skip & 1024u)) {
                        uint32_t n2 = 2u;
                        while (n2 < n_list) n2 <<= 1;
                        if (n2 <= q.region_dwords) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the walk's stores, before other lanes read them
                            for (uint32_t i = lane_; i < n2; i += 64u) region[i] = i < n_list ? cand[i] : 0xFFFFFFFFu;
                            __builtin_amdgcn_wave_barrier();
                            for (uint32_t k = 2u; k <= n2; k <<= 1)
                                for (uint32_t j = k >> 1; j != 0u; j >>= 1) {
                                    for (uint32_t t = lane_; t < (n2 >> 1); t += 64u) {
                                        const uint32_t i = 2u * t - (t & (j - 1u)), ixj = i | j;
                                        const uint32_t va = region[i], vb = region[ixj];
                                        if ((va > vb) == ((i & k) == 0u)) { region[i] = vb; region[ixj] = va; }
                                    }
                                    __builtin_amdgcn_wave_barrier();
                                }
                            for (uint32_t i = lane_; i < n_list; i += 64u) cand[i] = region[i];
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
        }
        // ---- one pixel of the strip per trip of this loop
        const uint32_t px = x0 + (pi & 7u), py = y0 + (pi >> 3);
        const uint32_t slab_idx = chunk * a.n_owned + blk * 64u + pi;
        pi++;
        if (px >= a.width || py >= a.height) continue;  // slots of clipped tiles lie outside the image (wave-uniform)
        const uint32_t pix = py * a.width + px;
        const float xn = (float(2u * px + 1u) - float(a.width)) * a.inv_dim;           // src/renderer.rs:174-176
        const float yn = (float(2u * (a.height - py) - 1u) - float(a.height)) * a.inv_dim;
        const uint32_t n_sub = (n_s + 63u) >> 6;   // trips of 64 samples
        // camera ray of sample (chunk, sub, lane) and its closest hit
        auto gen_ray = [&](uint32_t sub, Rng& rng, V& ro, V& rd, float& tmin, float& t, uint32_t& code, uint32_t& inst) {
            const bool active = sub * 64u + lane_ < n_s;
            ro = mk(0, 0, 0); rd = mk(0, 0, 1); tmin = 0.f; t = kInf; code = CODE_MISS; inst = 0;
            rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
            if (active) {
                rng.seed(a.seed_mixed, pix, a.sample_offset + chunk * kSuper + sub * 64u + lane_);
                float dx = rng.range(-a.inv_dim, a.inv_dim), dy = rng.range(-a.inv_dim, a.inv_dim);
                cast_ray(kernarg_load(&a.cam), xn + dx, yn + dy, rng, ro, rd);
                tmin = ray_tmin_p(ro);
                closest_hit<(BVH ? 2 : 0), false>(sc_arg, ro, rd, tmin, t, code, inst, stk, 256, c0, c1);
            }
            return active;
        };
        // ---- beam x point estimate, one photon per lane (src/photon.rs:439-502)
        V beam_sum = mk(0, 0, 0);   // this lane's photons over all the pixel's rays
        const bool beam_lanes = MEDIUM && KIND == RPT_PHOTON_POINT_BEAM && cand_valid;
        // the surface gather can collect its candidates once for all the pixel's samples (below) when the estimator decides
        // "surface or not" by the hit alone
        // -- and when no beam walk of the trips below needs the list's LDS (it runs with the photons in the lanes, or not at all)
        const bool pix_gather = PHASE != 1 && !GG && q.coop_cap != 0u && q.gather_size != 0u && q.n_s != 0u && KIND != RPT_PHOTON_MAP && !(q.skip & 2u) &&
                                (!MEDIUM || beam_lanes || (q.skip & 1u) != 0u || PHASE == 2);
        float far2 = 0.f, near2 = kInf;
        bool have_xc = false;      // wave-uniform
        V pxc = mk(0, 0, 0);       // wave-uniform: the first surface point among the pixel's samples
        float prho2 = 0.f;         // how far (squared) this lane's surface points lie from it
        unsigned long long tk = tick();
        if (beam_lanes || pix_gather) {
            for (uint32_t sub = 0; sub < n_sub; sub++) {
                Rng rng;
                V ro, rd;
                float tmin, t;
                uint32_t code, inst;
                const bool active = gen_ray(sub, rng, ro, rd, tmin, t, code, inst);
                if (beam_lanes) {
                    const float t2 = !active ? -1.f : (code != CODE_MISS ? t * t : kInf);   // no photon centre lies within a negative distance
                    rays[sub * 64u + lane_] = make_float4(rd.x, rd.y, rd.z, t2);
                    far2 = fmaxf(far2, t2);
                    if (active) near2 = fminf(near2, t2);
                }
                if (pix_gather) {
                    const bool sf = active && code != CODE_MISS;
                    const uint64_t sm = __ballot(sf);
                    if (sm != 0ull) {
                        const V xs = fma3(t, rd, ro);
                        if (!have_xc) {
                            const uint32_t lead = uint32_t(__ffsll((unsigned long long)sm)) - 1u;
                            auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lead)); };
                            pxc = mk(bc(xs.x), bc(xs.y), bc(xs.z));
                            have_xc = true;
                        }
                        const V dxs = xs - pxc;
                        if (sf) prho2 = fmaxf(prho2, dot(dxs, dxs));
                    }
                }
            }
        }
        if (beam_lanes) {
            far2 = wave_max(far2);   // farthest hit of the pixel's samples (inf on a miss)
            near2 = wave_min(near2);  // nearest one
            { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 0, t1 - tk); tk = t1; }   // [0] first pass over the rays
            beam_sum = beam_estimate_photon_lanes(q, cand, cand_n, far2, near2, xn, yn, cam_right, cam_up, sigma_t, n_s, rays, pend_list);
            { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 1, t1 - tk); tk = t1; }   // [1] beam estimate
            __builtin_amdgcn_wave_barrier();   // the gather lists of the surface estimate reuse this LDS
        }
        // ---- the surface gather's candidates, once for all the pixel's samples when their hit points form one cluster: every
        // photon within pix_R of pxc, ordered by distance to pxc.  A lane may then search any ball that lies inside that one.
        GatherLds lds{gd, gi, region + q.gather_size * 64u, nullptr, nullptr};
        lds.keys = reinterpret_cast<float*>(lds.pstack + kBallStack);
        lds.cl = reinterpret_cast<F4*>(lds.keys + q.coop_cap);
        PixelList plist{false, pix_gather && have_xc, 0u, 0.f, pxc};
        const ShellBox shell = shell_box<BVH>();
        if (plist.have_c && !(q.skip & 256u)) pixel_candidates(q, lds, shell, plist, anc, prev_r2, prho2);   // (diagnostic: 256 = no pixel list)
        { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 2, t1 - tk); tk = t1; }   // [2] the pixel's candidate list (+ the first pass when there is no beam estimate)
        // ---- the pixel's samples, 64 per trip; lane = sample
        V pixel_sum = mk(0, 0, 0);
        // (PHASE 1 comes here only for the pixels of a strip without a candidate list: volume estimate with the samples in the lanes)
        for (uint32_t sub = 0; sub < ((PHASE == 1 && beam_lanes) ? 0u : n_sub); sub++) {
        V ro, rd;
        float tmin, t;
        uint32_t code, inst;
        Rng rng;
        const bool active = gen_ray(sub, rng, ro, rd, tmin, t, code, inst);
        { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 3, t1 - tk); tk = t1; }   // [3] second pass over the rays
        const V wo = -normalize(rd);
        if (a.counters) diag_add(a.counters, 0, (unsigned long long)__popcll(__ballot(active)));
        const bool hit = code != CODE_MISS;
        V color = mk(0, 0, 0);
        bool surface_on = hit;
        float surface_scale = 1.f;
        if (MEDIUM && KIND == RPT_PHOTON_MAP) {  // point x point volume estimate, src/photon.rs:384-438
            if (active) {
                const float xi = rng.range(0.f, 1.f);               // Medium::sample_d
                const float dd = -__logf(xi) / sigma_t;
                const float tr_d = __expf(-sigma_t * dd);
                if (!hit || dd < t) {
                    surface_on = false;
                    const V x = fma3(dd, rd, ro);
                    const bool hi = sc.medium_kind == 1u && x.y > 250.f;
                    const V mcol = hi ? mk(sc.medium_color_hi[0], sc.medium_color_hi[1], sc.medium_color_hi[2]) : mcol0;
                    float max_d2;
                    const uint32_t found = gather_knn(q.v_nodes, q.v_ph, q.n_v, x, q.gather_size_volume, gd, gi, max_d2);
                    V sum = mk(0, 0, 0);
                    for (uint32_t k = 0; k < found; k++) sum = sum + xyz(q.v_ph[gi[k * 64u]].pow);
                    // / (4/3 pi r^3) / extinction * transmittance / pdf, pdf = sigma_t * transmittance
                    const float r3 = max_d2 * sqrt1(max_d2);
                    const float scale = sc.medium_phase * rcp((4.f / 3.f) * kPi * r3) * rcp(sigma_t) * tr_d * rcp(sigma_t * tr_d);
                    color = scale * (sum * mcol);
                } else {
                    surface_scale = __expf(-sigma_t * t) * rcp(tr_d);   // transmittance(t) / (1 - cdf), 1 - cdf = T(d)
                }
            }
        } else if (PHASE != 2 && MEDIUM && !(q.skip & 1u) && !beam_lanes) {  // beam estimates with the samples in the lanes
            // (this walk's pending entries and staging slots lie where a pixel's candidate list would: `pix_gather` rules the list out
            // for such pixels -- counted in the counters build should the two ever meet again)
            if (a.counters && plist.valid) diag_add(a.counters, 23, 1ull);
            const V vc = volume_estimate_sample_lanes<KIND>(q, active, ro, rd, hit, t, sigma_t, sc.medium_phase, wstack, stage);
            color = vc * mcol0;
        }
        if (active && !hit && !MEDIUM) color = env_color(sc_arg, rd);  // src/photon.rs:597
        if constexpr (PHASE != 1) {
        // ---- surface estimate, src/photon.rs:327-375
        const bool surf = active && surface_on && !(q.skip & 2u);
        SurfaceSample s{ro, mk(0, 1, 0), wo, Mat{mk(0, 0, 0), 0.f, 0u, 0.f, 0.f}, mk(0, 0, 0), 0.f, surf, nullptr, 0u};
        if constexpr (EMIT)
            s.em = q.emit + size_t(slab_idx - chunk * a.n_owned) * (q.gather_size + 2u) * a.iterations + (chunk * kSuper + sub * 64u + lane_);
        if (surf) {
            uint32_t obj;
            finalize_hit(sc_arg, ro, rd, tmin, t, code, inst, s.n, obj);
            s.mat = load_mat(sc_arg, obj);
            s.x = fma3(t, rd, ro);
        }
        s.sc_col = mat_emit(s.mat) * mat_color(s.mat);
        WalkScratch ws{stk, c0, c1};
        { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 4, t1 - tk); tk = t1; }   // [4] volume estimate with the samples in the lanes, hit record, material
        if (q.skip & 512u) { s.todo = false; s.max_d2 = 1.f; }   // (diagnostic: 512 = hit record and material only)
        if (!GG && q.coop_cap != 0u && q.gather_size != 0u && q.n_s != 0u)
            surface_gather_wave<BVH, EMIT>(q, sc_arg, lds, shell, s, ws, plist, anc, prev_r2);
        if (a.counters) { const uint64_t fm = __ballot(s.todo); if (fm) { diag_add(q.r.counters, 18, 1ull); diag_add(q.r.counters, 19, (unsigned long long)(uint32_t(__popcll(fm)))); } }
        if (__ballot(s.todo) != 0ull) plist.valid = false;   // (the index lists of these searches lie where the pixel's candidate list is)
        if (s.todo) surface_gather_lane<BVH, EMIT>(q, sc_arg, lds, shell, s, ws, prev_r2);
        c0 = ws.c0;
        c1 = ws.c1;
        { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 5, t1 - tk); tk = t1; }   // [5] surface gather
        V sc_col = s.sc_col;
        const float max_d2 = s.max_d2;
        if constexpr (EMIT) {
            if (active) {   // (a sample without a surface estimate hands over an empty selection)
                s.em[size_t(q.gather_size) * a.iterations] = surf ? s.n_em : 0u;
                s.em[size_t(q.gather_size + 1u) * a.iterations] = __float_as_uint(max_d2);
            }
            if (surf) prev_r2 = max_d2;
        } else if (surf) {
            prev_r2 = max_d2;
            sc_col = (kInvPi * rcp(max_d2)) * sc_col;
            if (MEDIUM && KIND == RPT_PHOTON_MAP) sc_col = surface_scale * sc_col;  // :433-435
            else if (MEDIUM) sc_col = __expf(-sigma_t * t) * sc_col;                  // :610-611
            color = color + sc_col;
        }
        }   // PHASE != 1
        if (active) pixel_sum = pixel_sum + color;
        }   // trips of the pixel
        // the pixel's partial sum over this chunk: per lane its samples in trip order plus its photons' beam terms, then
        // the lanes in a fixed butterfly order
        V sum = beam_lanes ? fma3(beam_sum, mcol0, pixel_sum) : pixel_sum;
        for (int off = 32; off; off >>= 1) {
            sum.x += __shfl_xor(sum.x, off);
            sum.y += __shfl_xor(sum.y, off);
            sum.z += __shfl_xor(sum.z, off);
        }
        if (lane_ == 0) reinterpret_cast<float4*>(PHASE == 2 ? a.slab2 : a.slab)[slab_idx] = make_float4(sum.x, sum.y, sum.z, 0.f);
        { const unsigned long long t1 = tick(); diag_add(a.counters, 24 + 6, t1 - tk); tk = t1; }   // [6] the pixel's sum
    }
    // Diagnostic counters (counters build; added where they occur, diag_add): [0] camera samples, [5] / [6] photon spheres visited /
    // accepted; the wave-level surface gather: [8] trips with a gather, [9] photon terms without a visibility scan, [10] steps of
    // the ball walks, [11] candidates, [12] overfull walks, [13] / [14] selection steps / list updates, [15] / [16] candidates looked
    // at / photon terms of the second pass, [17] new anchors, [18] / [19] trips / lanes that searched one by one; [23] trips in which
    // two layouts of the wave's LDS region were live at once (must stay 0); [24..33] clock
    // ticks (100 MHz) per part of a pixel, summed over the waves.
}

}  // namespace rptg

// ============================================================================ host side
using namespace rptg;

namespace {
// Device buffers of the photon maps of one scene, kept from one map to the next.  Renderer::photon_render builds a new map
// per call (src/photon.rs:655-704): ~40 buffers of up to 100 MB each, and hipMalloc / hipFree of that size cost
// milliseconds apiece (a C4 step spent 20 of its 140 ms in them).  A block is handed out again when it is free and fits
// (at most twice the size asked for); blocks that stayed unused for three maps are given back.
struct DevPool {
    struct Block { void* p; size_t cap; bool used; uint32_t idle; };
    std::vector<Block> blocks;
    ~DevPool() { for (auto& b : blocks) (void)hipFree(b.p); }
    hipError_t alloc(void** out, size_t bytes) {
        bytes = std::max<size_t>(bytes, 64);
        int best = -1;
        for (size_t i = 0; i < blocks.size(); i++)
            if (!blocks[i].used && blocks[i].cap >= bytes && blocks[i].cap <= 2 * bytes + (size_t(1) << 20) &&
                (best < 0 || blocks[i].cap < blocks[size_t(best)].cap))
                best = int(i);
        if (best >= 0) {
            blocks[size_t(best)].used = true;
            blocks[size_t(best)].idle = 0;
            *out = blocks[size_t(best)].p;
            return hipSuccess;
        }
        const size_t cap = bytes + bytes / 8;   // photon counts move by a fraction of a percent from map to map
        hipError_t e = hipMalloc(out, cap);
        if (e == hipSuccess) blocks.push_back(Block{*out, cap, true, 0});
        return e;
    }
    void free(void* p) {
        if (!p) return;
        for (auto& b : blocks)
            if (b.p == p) { b.used = false; return; }
        (void)hipFree(p);
    }
    void next_map() {   // called between maps, with the device idle
        for (size_t i = 0; i < blocks.size();) {
            if (!blocks[i].used && ++blocks[i].idle > 3) {
                (void)hipFree(blocks[i].p);
                blocks.erase(blocks.begin() + long(i));
            } else {
                i++;
            }
        }
    }
};
struct DevLbvh {
    BvhNode* nodes = nullptr;
    PhotonRec* sorted = nullptr;
    uint32_t n = 0;
};
struct PhotonMapDev {
    int device = 0;
    int kind = RPT_PHOTON_POINT_BEAM;
    uint64_t photon_count = 0;
    DevLbvh surf, vol;
    bool built = false;
    // records of the last shooting pass in shooting order (kept for rpt_photon_records / all-gather)
    PhotonRec *raw_s = nullptr, *raw_v = nullptr;
    uint64_t n_raw_s = 0, n_raw_v = 0;
    double build_ms[4] = {0, 0, 0, 0};  // shoot, sort+build, radii, total
    uint32_t* d_overflow = nullptr;
    uint32_t* d_cand = nullptr;  // per-wave candidate lists of the camera pass
    size_t cand_words = 0;
    uint32_t* d_gather = nullptr;  // per-wave k-nearest lists of gathers too large for LDS
    size_t gather_words = 0;
    float* d_slab2 = nullptr;      // the surface term's partial sums of the split camera pass
    size_t slab2_bytes = 0;
    // reference-epsilon mode (kernels_f64.hip): the surface photons' fp64 positions in shooting order, the camera pass's per-sample
    // selections of one slice of samples, and the fp64 partial sums of its surface estimate
    double* pos64 = nullptr;
    uint32_t* d_emit = nullptr;
    size_t emit_words = 0;
    double* d_slab64 = nullptr;
    size_t slab64_bytes = 0;
    uint64_t emit_dims[3] = {0, 0, 0};   // of the last slice: owned pixel slots, gather_size + 2, samples
    std::shared_ptr<DevPool> pool = std::make_shared<DevPool>();   // handed on to the scene's next map (fresh_map)
    void release_raw() {
        pool->free(raw_s); pool->free(raw_v);
        raw_s = raw_v = nullptr;
        n_raw_s = n_raw_v = 0;
    }
    void release() {
        (void)hipSetDevice(device);
        pool->free(surf.nodes); pool->free(surf.sorted);
        pool->free(vol.nodes); pool->free(vol.sorted);
        pool->free(d_overflow);
        pool->free(d_cand);
        pool->free(d_gather);
        pool->free(d_slab2);
        d_slab2 = nullptr;
        slab2_bytes = 0;
        pool->free(pos64); pool->free(d_emit); pool->free(d_slab64);
        pos64 = nullptr; d_emit = nullptr; d_slab64 = nullptr;
        emit_words = slab64_bytes = 0;
        release_raw();
        d_overflow = nullptr;
        d_cand = nullptr;
        d_gather = nullptr;
        cand_words = gather_words = 0;
        surf = DevLbvh{};
        vol = DevLbvh{};
        built = false;
    }
};
struct Tmp {   // scratch buffers of one build step, back in the pool when it is over (the step ends with a synchronise)
    DevPool& pool;
    std::vector<void*> ptrs;
    explicit Tmp(DevPool& p) : pool(p) {}
    ~Tmp() {
        for (void* p : ptrs) pool.free(p);
    }
    template <class T>
    hipError_t alloc(T** p, size_t n) {
        hipError_t e = pool.alloc((void**)p, n * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

// Build the LBVH of `n` photons in `raw` (consumed: the sorted copy is kept).  radius_k > 0: also
// compute the k-NN radii and refit the boxes with them (sphere map for the beam query).
// mode 0: point map; 1: point map + k-NN radii, then sphere boxes; 2: beam map (boxes of whole beams)
int build_lbvh(DevPool& pool, PhotonRec* raw, uint32_t n, int mode, DevLbvh& out, hipStream_t st) {
    const bool with_radius = mode == 1;
    const int first_mode = mode == 2 ? 2 : 0;
    out = DevLbvh{};
    out.n = n;
    if (n == 0) return RPT_OK;
    Tmp tmp(pool);
    float* lohi;
    uint64_t *keys, *keys2;
    uint32_t *vals, *vals2, *left, *right, *par_i, *par_l, *flags, *rlo, *rhi;
    float* box;
    RPTI_HIP_TRY(tmp.alloc(&lohi, 6));
    RPTI_HIP_TRY(tmp.alloc(&keys, n));
    RPTI_HIP_TRY(tmp.alloc(&keys2, n));
    RPTI_HIP_TRY(tmp.alloc(&vals, n));
    RPTI_HIP_TRY(tmp.alloc(&vals2, n));
    RPTI_HIP_TRY(tmp.alloc(&left, n));
    RPTI_HIP_TRY(tmp.alloc(&right, n));
    RPTI_HIP_TRY(tmp.alloc(&par_i, n));
    RPTI_HIP_TRY(tmp.alloc(&par_l, n));
    RPTI_HIP_TRY(tmp.alloc(&flags, n));
    RPTI_HIP_TRY(tmp.alloc(&rlo, n));
    RPTI_HIP_TRY(tmp.alloc(&rhi, n));
    RPTI_HIP_TRY(tmp.alloc(&box, size_t(n) * 6));
    const float inf = std::numeric_limits<float>::infinity();
    float init[6] = {inf, inf, inf, -inf, -inf, -inf};
    RPTI_HIP_TRY(hipMemcpyAsync(lohi, init, sizeof(init), hipMemcpyHostToDevice, st));
    uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(bounds_kernel, dim3(std::min(blocks, 1024u)), dim3(256), 0, st, raw, n, lohi, first_mode);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(256), 0, st, raw, n, lohi, keys, vals, first_mode);
    // stable radix sort of (Morton key, photon index), eight 8-bit passes; the last one gathers the records (sort_scan.h).  Equal keys
    // keep their shooting order, so the sorted array -- and with it the tree -- is a function of the photons alone.
    char* temp;
    RPTI_HIP_TRY(tmp.alloc(&temp, ss::rsort_temp_bytes(n)));
    RPTI_HIP_TRY(pool.alloc((void**)&out.sorted, size_t(n) * sizeof(PhotonRec)));
    RPTI_HIP_TRY(ss::radix_sort_pairs(keys, vals, keys2, vals2, n, 8u, temp, GatherPhotons{raw, out.sorted, keys}, st));   // (pass 7 reads keys2 / vals2: keys is free for the sorted keys)
    if (n >= 2) {
        RPTI_HIP_TRY(pool.alloc((void**)&out.nodes, size_t(n - 1) * sizeof(BvhNode)));
        hipLaunchKernelGGL(karras_kernel, dim3(blocks), dim3(256), 0, st, keys, int(n), left, right, par_i, par_l, rlo, rhi);
        RPTI_HIP_TRY(hipMemsetAsync(flags, 0, size_t(n) * 4, st));
        hipLaunchKernelGGL(refit_small_kernel, dim3(blocks), dim3(256), 0, st, out.sorted, int(n), rlo, rhi, box, first_mode);
        hipLaunchKernelGGL(refit_kernel, dim3(2u * blocks), dim3(256), 0, st, out.sorted, int(n), left, right, par_i, par_l, rlo, rhi, flags, box, first_mode);
        // point trees are walked by the k-NN search only (radii, gathers): collapsed leaves
        hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, out.sorted, int(n), left, right, box, rlo, rhi, par_i, out.nodes,
                           first_mode, first_mode == 0 ? kKnnLeaf : 1u);
    }
    if (with_radius) {
        float* radius;
        RPTI_HIP_TRY(tmp.alloc(&radius, n));
        hipLaunchKernelGGL(knn_radius_kernel, dim3(blocks), dim3(256), 0, st, out.nodes, out.sorted, n, radius);
        hipLaunchKernelGGL(set_radius_kernel, dim3(blocks), dim3(256), 0, st, out.sorted, n, radius);
        if (n >= 2) {
            RPTI_HIP_TRY(hipMemsetAsync(flags, 0, size_t(n) * 4, st));
            hipLaunchKernelGGL(refit_small_kernel, dim3(blocks), dim3(256), 0, st, out.sorted, int(n), rlo, rhi, box, 1);
            hipLaunchKernelGGL(refit_kernel, dim3(2u * blocks), dim3(256), 0, st, out.sorted, int(n), left, right, par_i, par_l, rlo, rhi, flags, box, 1);
            hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, out.sorted, int(n), left, right, box, rlo, rhi, par_i, out.nodes, 1,
                               1u);  // sphere tree for the beam walkers: one photon per leaf
        }
    }
    RPTI_HIP_TRY(hipGetLastError());
    RPTI_HIP_TRY(hipStreamSynchronize(st));
    return RPT_OK;
}

template <bool W>
hipError_t launch_shoot(const ShootArgs& a, bool medium, bool bvh, int blocks, hipStream_t st) {
    size_t lds = bvh ? 32u * 256u * 4u : 0;
    if (medium) {
        if (bvh) hipLaunchKernelGGL((photon_shoot_kernel<true, true, W>), dim3(blocks), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((photon_shoot_kernel<true, false, W>), dim3(blocks), dim3(256), lds, st, a);
    } else {
        if (bvh) hipLaunchKernelGGL((photon_shoot_kernel<false, true, W>), dim3(blocks), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((photon_shoot_kernel<false, false, W>), dim3(blocks), dim3(256), lds, st, a);
    }
    return hipGetLastError();
}
}  // namespace

void rpti::photon_release(void* p) {
    auto* m = static_cast<PhotonMapDev*>(p);
    if (m) {
        m->release();
        delete m;
    }
}

// The split camera pass (PHASE 1, then 2): same grid, same work items; the work counter starts again in between.  A measured-
// slower prototype (112 ms against 95 on C4): its eight kernel flavours are built with -DRPT_EXPERIMENTS only.
#ifdef RPT_EXPERIMENTS
template <bool B>
static void launch_query_split(const QueryArgs& q, int kind, int nb, size_t lds, hipStream_t st) {
    const dim3 g(nb), b(256);
    if (kind == RPT_PHOTON_BEAM_BEAM) hipLaunchKernelGGL((photon_query_kernel<true, B, false, RPT_PHOTON_BEAM_BEAM, 1>), g, b, lds, st, q);
    else hipLaunchKernelGGL((photon_query_kernel<true, B, false, RPT_PHOTON_POINT_BEAM, 1>), g, b, lds, st, q);
    (void)hipMemsetAsync(q.r.queue, 0, 8, st);
    if (kind == RPT_PHOTON_BEAM_BEAM) hipLaunchKernelGGL((photon_query_kernel<true, B, false, RPT_PHOTON_BEAM_BEAM, 2>), g, b, lds, st, q);
    else hipLaunchKernelGGL((photon_query_kernel<true, B, false, RPT_PHOTON_POINT_BEAM, 2>), g, b, lds, st, q);
}
#endif
template <bool M, bool B, bool G>
static void launch_query_k(const QueryArgs& q, int kind, int nb, size_t lds, hipStream_t st) {
    const dim3 g(nb), b(256);
#ifdef RPT_EXPERIMENTS
    if constexpr (M && !G) {
        if (q.r.slab2 && kind != RPT_PHOTON_MAP) return launch_query_split<B>(q, kind, nb, lds, st);
    }
#endif
    if (kind == RPT_PHOTON_MAP) hipLaunchKernelGGL((photon_query_kernel<M, B, G, RPT_PHOTON_MAP>), g, b, lds, st, q);
    else if (kind == RPT_PHOTON_BEAM_BEAM) hipLaunchKernelGGL((photon_query_kernel<M, B, G, RPT_PHOTON_BEAM_BEAM>), g, b, lds, st, q);
    else hipLaunchKernelGGL((photon_query_kernel<M, B, G, RPT_PHOTON_POINT_BEAM>), g, b, lds, st, q);
}
static void launch_query(const QueryArgs& q, bool medium, bool bvh, bool gg, int kind, int nb, size_t lds, hipStream_t st) {
    const int sel = (medium ? 4 : 0) | (bvh ? 2 : 0) | (gg ? 1 : 0);
    switch (sel) {
        case 0: launch_query_k<false, false, false>(q, kind, nb, lds, st); break;
        case 1: launch_query_k<false, false, true>(q, kind, nb, lds, st); break;
        case 2: launch_query_k<false, true, false>(q, kind, nb, lds, st); break;
        case 3: launch_query_k<false, true, true>(q, kind, nb, lds, st); break;
        case 4: launch_query_k<true, false, false>(q, kind, nb, lds, st); break;
        case 5: launch_query_k<true, false, true>(q, kind, nb, lds, st); break;
        case 6: launch_query_k<true, true, false>(q, kind, nb, lds, st); break;
        default: launch_query_k<true, true, true>(q, kind, nb, lds, st); break;
    }
}

// The camera pass that hands its selections over (reference-epsilon mode)
template <bool M, bool B, bool G>
static void launch_query_emit_k(const QueryArgs& q, int kind, int nb, size_t lds, hipStream_t st) {
    const dim3 g(nb), b(256);
    if (kind == RPT_PHOTON_MAP) hipLaunchKernelGGL((photon_query_kernel<M, B, G, RPT_PHOTON_MAP, 0, true>), g, b, lds, st, q);
    else if (kind == RPT_PHOTON_BEAM_BEAM) hipLaunchKernelGGL((photon_query_kernel<M, B, G, RPT_PHOTON_BEAM_BEAM, 0, true>), g, b, lds, st, q);
    else hipLaunchKernelGGL((photon_query_kernel<M, B, G, RPT_PHOTON_POINT_BEAM, 0, true>), g, b, lds, st, q);
}
static void launch_query_emit(const QueryArgs& q, bool medium, bool bvh, bool gg, int kind, int nb, size_t lds, hipStream_t st) {
    const int sel = (medium ? 4 : 0) | (bvh ? 2 : 0) | (gg ? 1 : 0);
    switch (sel) {
        case 0: launch_query_emit_k<false, false, false>(q, kind, nb, lds, st); break;
        case 1: launch_query_emit_k<false, false, true>(q, kind, nb, lds, st); break;
        case 2: launch_query_emit_k<false, true, false>(q, kind, nb, lds, st); break;
        case 3: launch_query_emit_k<false, true, true>(q, kind, nb, lds, st); break;
        case 4: launch_query_emit_k<true, false, false>(q, kind, nb, lds, st); break;
        case 5: launch_query_emit_k<true, false, true>(q, kind, nb, lds, st); break;
        case 6: launch_query_emit_k<true, true, false>(q, kind, nb, lds, st); break;
        default: launch_query_emit_k<true, true, true>(q, kind, nb, lds, st); break;
    }
}

extern "C" {

// The shooting pass of the reference-epsilon mode: same two passes, same streams, same record arrays; the chains are traced by
// kernels_f64.hip (t_min = 1e-12, fp64), which also keeps every surface photon's position in fp64 (pm->pos64).
static int shoot_range64(rpt_scene* s, PhotonMapDev* pm, uint64_t photon_count, uint64_t first, uint64_t n, int32_t kind,
                         double watts, uint64_t seed) {
    rpti::SceneDev sd = rpti::scene_dev(s);
    hipStream_t st = nullptr;
    hipEvent_t e0, e1;
    RPTI_HIP_TRY(hipEventCreate(&e0));
    RPTI_HIP_TRY(hipEventCreate(&e1));
    Tmp tmp(*pm->pool);
    rpt64::ShootArgs64 a{};
    rpti::fill_args64(s, nullptr, nullptr, nullptr, a.a);
    a.a.seed_mixed = rpti::seed_mix(seed);
    a.n_photons = n;
    a.first_photon = first;
    a.power = watts / double(photon_count);
    a.light_index = uint32_t(sd.first_object_light);
    a.kind = uint32_t(kind);
    pm->release_raw();
    pm->pool->free(pm->pos64);
    pm->pos64 = nullptr;
    if (n == 0) return RPT_OK;
    RPTI_HIP_TRY(tmp.alloc(&a.cnt_s, n));
    RPTI_HIP_TRY(tmp.alloc(&a.cnt_v, n));
    RPTI_HIP_TRY(tmp.alloc(&a.a.queue, 1));
    const int blocks = int(std::min<uint64_t>((n + 255) / 256, uint64_t(sd.n_cus) * 4));
    RPTI_HIP_TRY(hipEventRecord(e0, st));
    RPTI_HIP_TRY(hipMemsetAsync(a.a.queue, 0, 8, st));
    RPTI_HIP_TRY(launch_photon_shoot_f64(a, blocks, st));
    uint32_t *d_os, *d_ov;
    RPTI_HIP_TRY(tmp.alloc(&d_os, n));
    RPTI_HIP_TRY(tmp.alloc(&d_ov, n));
    char* scan_tmp;
    RPTI_HIP_TRY(tmp.alloc(&scan_tmp, std::max<size_t>(ss::scan2_temp_bytes(uint32_t(n)), 8)));
    unsigned long long* d_tot;
    RPTI_HIP_TRY(tmp.alloc(&d_tot, 2));
    RPTI_HIP_TRY(ss::exclusive_scan2(a.cnt_s, a.cnt_v, uint32_t(n), d_os, d_ov, d_tot, scan_tmp, st));
    unsigned long long tot[2] = {0, 0};
    RPTI_HIP_TRY(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, st));
    RPTI_HIP_TRY(hipStreamSynchronize(st));
    const uint64_t ts = tot[0], tv = tot[1];
    if (ts >= (1ull << 26) || tv >= (1ull << 26)) return rpti::fail(RPT_ERR_UNSUPPORTED, "too many photons (2^26 records per map)");
    RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->raw_s, ts * sizeof(PhotonRec)));
    RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->raw_v, tv * sizeof(PhotonRec)));
    RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->pos64, ts * 24u));
    pm->n_raw_s = ts;
    pm->n_raw_v = tv;
    a.off_s = d_os;
    a.off_v = d_ov;
    a.surf = reinterpret_cast<rpt64::PhotonRec32*>(pm->raw_s);
    a.vol = reinterpret_cast<rpt64::PhotonRec32*>(pm->raw_v);
    a.pos64 = pm->pos64;
    RPTI_HIP_TRY(hipMemsetAsync(a.a.queue, 0, 8, st));
    RPTI_HIP_TRY(launch_photon_shoot_f64(a, blocks, st));
    RPTI_HIP_TRY(hipEventRecord(e1, st));
    RPTI_HIP_TRY(hipEventSynchronize(e1));
    float m0 = 0;
    (void)hipEventElapsedTime(&m0, e0, e1);
    pm->build_ms[0] = m0;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RPT_OK;
}

// Shooting pass for photons [first, first + n) of a map of `photon_count` photons: count, prefix, write.
// Leaves the records in pm->raw_s / raw_v in shooting order.
static int shoot_range(rpt_scene* s, PhotonMapDev* pm, uint64_t photon_count, uint64_t first, uint64_t n, int32_t kind,
                       double watts, uint64_t seed) {
    rpti::SceneDev sd = rpti::scene_dev(s);
    if (sd.epsilon64) return shoot_range64(s, pm, photon_count, first, n, kind, watts, seed);
    hipStream_t st = nullptr;
    hipEvent_t e0, e1;
    RPTI_HIP_TRY(hipEventCreate(&e0));
    RPTI_HIP_TRY(hipEventCreate(&e1));
    Tmp tmp(*pm->pool);
    ShootArgs a{};
    a.sc = sd.view;
    a.n_photons = n;
    a.first_photon = first;
    a.seed_mixed = rpti::seed_mix(seed);
    a.power = float(watts / double(photon_count));
    a.light_index = uint32_t(sd.first_object_light);
    a.kind = uint32_t(kind);
    pm->release_raw();
    if (n == 0) return RPT_OK;
    RPTI_HIP_TRY(tmp.alloc(&a.cnt_s, n));
    RPTI_HIP_TRY(tmp.alloc(&a.cnt_v, n));
    const bool medium = sd.view.has_medium != 0, bvh = sd.view.n_nodes != 0;
    int blocks = int(std::min<uint64_t>((n + 255) / 256, uint64_t(sd.n_cus) * 8));
    RPTI_HIP_TRY(hipEventRecord(e0, st));
    RPTI_HIP_TRY(launch_shoot<false>(a, medium, bvh, blocks, st));
    // offsets = exclusive prefix sums of the per-photon record counts, on the device
    uint32_t *d_os, *d_ov;
    RPTI_HIP_TRY(tmp.alloc(&d_os, n));
    RPTI_HIP_TRY(tmp.alloc(&d_ov, n));
    // (totals in 64 bits: the 32-bit offsets are only used once the totals are known to fit)
    char* scan_tmp;
    RPTI_HIP_TRY(tmp.alloc(&scan_tmp, std::max<size_t>(ss::scan2_temp_bytes(uint32_t(n)), 8)));
    unsigned long long* d_tot;
    RPTI_HIP_TRY(tmp.alloc(&d_tot, 2));
    RPTI_HIP_TRY(ss::exclusive_scan2(a.cnt_s, a.cnt_v, uint32_t(n), d_os, d_ov, d_tot, scan_tmp, st));
    unsigned long long tot[2] = {0, 0};
    RPTI_HIP_TRY(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, st));
    RPTI_HIP_TRY(hipStreamSynchronize(st));
    const uint64_t ts = tot[0], tv = tot[1];
    if (ts >= (1ull << 26) || tv >= (1ull << 26)) return rpti::fail(RPT_ERR_UNSUPPORTED, "too many photons (2^26 records per map)");
    RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->raw_s, ts * sizeof(PhotonRec)));
    RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->raw_v, tv * sizeof(PhotonRec)));
    pm->n_raw_s = ts;
    pm->n_raw_v = tv;
    a.off_s = d_os;
    a.off_v = d_ov;
    a.surf = pm->raw_s;
    a.vol = pm->raw_v;
    RPTI_HIP_TRY(launch_shoot<true>(a, medium, bvh, blocks, st));
    RPTI_HIP_TRY(hipEventRecord(e1, st));
    RPTI_HIP_TRY(hipEventSynchronize(e1));
    float m0 = 0;
    (void)hipEventElapsedTime(&m0, e0, e1);
    pm->build_ms[0] = m0;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RPT_OK;
}

// Sort + LBVH (+ radii) over record arrays that live on this scene's device.
static int build_maps(PhotonMapDev* pm, const PhotonRec* d_s, uint64_t n_s, const PhotonRec* d_v, uint64_t n_v) {
    if (n_s >= (1ull << 26) || n_v >= (1ull << 26)) return rpti::fail(RPT_ERR_UNSUPPORTED, "too many photons (2^26 records per map)");
    hipStream_t st = nullptr;
    hipEvent_t e1, e2;
    RPTI_HIP_TRY(hipEventCreate(&e1));
    RPTI_HIP_TRY(hipEventCreate(&e2));
    RPTI_HIP_TRY(hipEventRecord(e1, st));
    const int kind = pm->kind;
    int rc = build_lbvh(*pm->pool, const_cast<PhotonRec*>(d_s), uint32_t(n_s), 0, pm->surf, st);
    if (rc == RPT_OK)
        rc = build_lbvh(*pm->pool, const_cast<PhotonRec*>(d_v), uint32_t(n_v),
                        kind == RPT_PHOTON_POINT_BEAM ? 1 : (kind == RPT_PHOTON_BEAM_BEAM ? 2 : 0), pm->vol, st);
    if (rc != RPT_OK) return rc;
    RPTI_HIP_TRY(hipEventRecord(e2, st));
    RPTI_HIP_TRY(hipEventSynchronize(e2));
    float m1 = 0;
    (void)hipEventElapsedTime(&m1, e1, e2);
    pm->build_ms[1] = m1;
    pm->build_ms[3] = pm->build_ms[0] + m1;
    (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    pm->built = true;
    return RPT_OK;
}

static int photon_args_ok(rpt_scene* s, uint64_t photon_count, int32_t kind) {
    if (!s) return rpti::fail(RPT_ERR_INVALID, "null scene");
    rpti::SceneDev sd = rpti::scene_dev(s);
    if (!sd.committed) return rpti::fail(RPT_ERR_STATE, "rpt_scene_commit must be called first");
    if (photon_count == 0) return rpti::fail(RPT_ERR_INVALID, "photon_count must be > 0");
    if (kind != RPT_PHOTON_POINT_BEAM && kind != RPT_PHOTON_MAP && kind != RPT_PHOTON_BEAM_BEAM)
        return rpti::fail(RPT_ERR_INVALID, "unknown PhotonRenderKind");
    if (sd.first_object_light < 0)
        return rpti::fail(RPT_ERR_INVALID, "Only found non-object lights while photon mapping");  // the reference's panic
    RPTI_HIP_TRY(hipSetDevice(sd.device));
    return RPT_OK;
}
static PhotonMapDev* fresh_map(rpt_scene* s, uint64_t photon_count, int32_t kind) {
    void*& slot = rpti::photon_slot(s);
    std::shared_ptr<DevPool> pool;
    if (slot) {
        // the old map's buffers go back to the pool and may be handed out at once: nothing may still be reading them
        // (hipFree used to wait for the device here)
        (void)hipSetDevice(rpti::scene_dev(s).device);
        (void)hipDeviceSynchronize();
        pool = static_cast<PhotonMapDev*>(slot)->pool;
        rpti::photon_release(slot);
        slot = nullptr;
        pool->next_map();
    }
    auto* pm = new PhotonMapDev();
    if (pool) pm->pool = pool;
    pm->device = rpti::scene_dev(s).device;
    pm->kind = kind;
    pm->photon_count = photon_count;
    slot = pm;
    return pm;
}
static void drop_map(rpt_scene* s) {
    void*& slot = rpti::photon_slot(s);
    rpti::photon_release(slot);
    slot = nullptr;
}

int rpt_photon_map_build(rpt_scene* s, uint64_t photon_count, int32_t kind, double watts, uint64_t seed) {
    int rc = photon_args_ok(s, photon_count, kind);
    if (rc) return rc;
    PhotonMapDev* pm = fresh_map(s, photon_count, kind);
    rc = shoot_range(s, pm, photon_count, 0, photon_count, kind, watts, seed);
    if (rc == RPT_OK) rc = build_maps(pm, pm->raw_s, pm->n_raw_s, pm->raw_v, pm->n_raw_v);
    if (rc != RPT_OK) { drop_map(s); return rc; }
    pm->release_raw();
    return RPT_OK;
}

int rpt_photon_shoot(rpt_scene* s, uint64_t photon_count, int32_t kind, double watts, uint64_t seed, uint32_t shard_rank,
                     uint32_t shard_count, uint64_t n_out[2]) {
    int rc = photon_args_ok(s, photon_count, kind);
    if (rc) return rc;
    if (rpti::scene_dev(s).epsilon64) return rpti::fail(RPT_ERR_UNSUPPORTED, "the reference-epsilon mode keeps the surface photons' positions in fp64 beside the 48-byte records: build the map with rpt_photon_map_build (on every rank)");
    if (shard_count == 0) shard_count = 1;
    if (shard_rank >= shard_count) return rpti::fail(RPT_ERR_INVALID, "shard_rank must be < shard_count");
    // contiguous blocks, so that the shards concatenated in rank order ARE the single-GPU record arrays
    const uint64_t first = photon_count * shard_rank / shard_count, last = photon_count * (uint64_t(shard_rank) + 1) / shard_count;
    PhotonMapDev* pm = fresh_map(s, photon_count, kind);
    rc = shoot_range(s, pm, photon_count, first, last - first, kind, watts, seed);
    if (rc != RPT_OK) { drop_map(s); return rc; }
    if (n_out) { n_out[0] = pm->n_raw_s; n_out[1] = pm->n_raw_v; }
    return RPT_OK;
}

int rpt_photon_records(rpt_scene* s, int32_t which, void** d_records, uint64_t* n) {
    if (!s || !d_records || !n) return rpti::fail(RPT_ERR_INVALID, "null argument");
    auto* pm = static_cast<PhotonMapDev*>(rpti::photon_slot(s));
    if (!pm || pm->built) return rpti::fail(RPT_ERR_STATE, "no shot photons: call rpt_photon_shoot first");
    *d_records = which == 0 ? pm->raw_s : pm->raw_v;
    *n = which == 0 ? pm->n_raw_s : pm->n_raw_v;
    return RPT_OK;
}

int rpt_photon_map_from_records(rpt_scene* s, uint64_t photon_count, int32_t kind, const void* d_surface, uint64_t n_surface,
                                const void* d_volume, uint64_t n_volume) {
    int rc = photon_args_ok(s, photon_count, kind);
    if (rc) return rc;
    if ((n_surface && !d_surface) || (n_volume && !d_volume)) return rpti::fail(RPT_ERR_INVALID, "null record array");
    if (rpti::scene_dev(s).epsilon64) return rpti::fail(RPT_ERR_UNSUPPORTED, "the reference-epsilon mode keeps the surface photons' positions in fp64 beside the 48-byte records: build the map with rpt_photon_map_build (on every rank)");
    auto* old = static_cast<PhotonMapDev*>(rpti::photon_slot(s));
    // the arrays may be this scene's own shot records: keep them alive until the maps are built
    PhotonMapDev keep;
    if (old) { keep.raw_s = old->raw_s; keep.raw_v = old->raw_v; keep.build_ms[0] = old->build_ms[0]; old->raw_s = old->raw_v = nullptr; }
    PhotonMapDev* pm = fresh_map(s, photon_count, kind);
    pm->build_ms[0] = keep.build_ms[0];
    rc = build_maps(pm, static_cast<const PhotonRec*>(d_surface), n_surface, static_cast<const PhotonRec*>(d_volume), n_volume);
    pm->pool->free(keep.raw_s); pm->pool->free(keep.raw_v);
    if (rc != RPT_OK) { drop_map(s); return rc; }
    return RPT_OK;
}

int rpt_photon_map_stats(rpt_scene* s, uint64_t out[8]) {
    if (!s || !out) return rpti::fail(RPT_ERR_INVALID, "null argument");
    auto* pm = static_cast<PhotonMapDev*>(rpti::photon_slot(s));
    if (!pm || !pm->built) return rpti::fail(RPT_ERR_STATE, "no photon map: call rpt_photon_map_build first");
    out[0] = pm->surf.n;
    out[1] = pm->vol.n;
    out[2] = pm->photon_count;
    out[3] = uint64_t(pm->build_ms[0] * 1000.0);  // microseconds: shooting (both passes)
    out[4] = uint64_t(pm->build_ms[1] * 1000.0);  // microseconds: sort + LBVH + radii
    out[5] = out[6] = out[7] = 0;
    return RPT_OK;
}

// which: 0 surface, 1 volume.  out: n * 10 floats in ORIGINAL (shooting) order: position, direction,
// power, radius.
int rpt_photon_map_download(rpt_scene* s, int32_t which, float* out, uint64_t capacity) {
    if (!s || !out) return rpti::fail(RPT_ERR_INVALID, "null argument");
    auto* pm = static_cast<PhotonMapDev*>(rpti::photon_slot(s));
    if (!pm || !pm->built) return rpti::fail(RPT_ERR_STATE, "no photon map: call rpt_photon_map_build first");
    const DevLbvh& l = which == 0 ? pm->surf : pm->vol;
    if (capacity < l.n) return rpti::fail(RPT_ERR_INVALID, "output buffer too small");
    std::vector<PhotonRec> h(l.n);
    if (l.n) RPTI_HIP_TRY(hipMemcpy(h.data(), l.sorted, size_t(l.n) * sizeof(PhotonRec), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < l.n; i++) {
        uint32_t orig;
        std::memcpy(&orig, &h[i].dir.w, 4);
        float* o = out + size_t(orig) * 10;
        o[0] = h[i].pos_r.x; o[1] = h[i].pos_r.y; o[2] = h[i].pos_r.z;
        o[3] = h[i].dir.x; o[4] = h[i].dir.y; o[5] = h[i].dir.z;
        o[6] = h[i].pow.x; o[7] = h[i].pow.y; o[8] = h[i].pow.z;
        o[9] = h[i].pos_r.w;
    }
    return RPT_OK;
}

// Reference-epsilon mode: the camera pass runs over the call's samples slice by slice (the selections of one slice are
// [n_owned][gather_size + 2][slice] dwords); `total` = samples of the whole call, `first`: the frame starts with this slice.
struct EpsSlice {
    uint32_t total;
    bool first;
};
static int photon_render_impl(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint64_t gather_size,
                              uint64_t gather_size_volume, uint32_t num_samples, uint64_t seed, uint32_t sample_offset,
                              double* d_out, hipStream_t st, bool sync_counters, const EpsSlice* eps = nullptr) {
    auto* pm = s ? static_cast<PhotonMapDev*>(rpti::photon_slot(s)) : nullptr;
    if (!pm || !pm->built) return rpti::fail(RPT_ERR_STATE, "no photon map: call rpt_photon_map_build first");
    const uint64_t gather_max = pm->kind == RPT_PHOTON_MAP ? std::max(gather_size, gather_size_volume) : gather_size;
    if (gather_max > kGatherMax) return rpti::fail(RPT_ERR_UNSUPPORTED, "gather sizes above 1024 are not supported");
    const bool gg = gather_max > kGatherLds;  // lists in global memory
    const uint64_t gather_lds = gg ? 0 : gather_max;
    QueryArgs q{};
    // Work items of the camera pass are wave-level: (a strip of rows of an 8x8 pixel block, chunk of up to kSuper
    // samples); the chunking is fixed here (whatever the "chunk_spp" option says) and prepare_render sizes the slab
    // [n_chunks][n_owned] for it.  How many strips a block is cut into ("photon_parts") changes the order in which a pixel's beam
    // terms are added (the strip's candidate list), i.e. the last bits of the image (1e-6), nothing else.
    int rc = rpti::prepare_render(s, st, cam, prm, num_samples, seed, sample_offset, q.r, 0, kSuper);
    if (rc) return rc;
    rc = rpti::serialize_with_other_streams(s, st);  // candidate lists and the overflow flag exist once per scene
    if (rc) return rc;
    const int64_t parts = rpti::option_photon_parts(s);
    if (parts != 1 && parts != 2 && parts != 4 && parts != 8) return rpti::fail(RPT_ERR_INVALID, "photon_parts must be 1, 2, 4 or 8");
    q.parts = uint32_t(parts);
    const uint64_t n_items = uint64_t(q.r.n_owned / 64u) * q.parts * q.r.n_chunks;
    if (n_items >= (1ull << 32) - (1ull << 24)) return rpti::fail(RPT_ERR_INVALID, "too many work items");
    q.r.n_items = uint32_t(n_items);
    q.s_nodes = pm->surf.nodes; q.s_ph = pm->surf.sorted; q.n_s = pm->surf.n;
    q.v_nodes = pm->vol.nodes; q.v_ph = pm->vol.sorted; q.n_v = pm->vol.n;
    q.kind = uint32_t(pm->kind);
    q.skip = uint32_t(rpti::option_photon_skip(s));
    q.gather_size = uint32_t(gather_size);
    q.gather_size_volume = pm->kind == RPT_PHOTON_MAP ? uint32_t(gather_size_volume) : 0u;  // only the point-point estimate gathers in the volume
    const bool medium = q.r.sc.has_medium != 0, bvh = q.r.sc.n_nodes != 0;
    // the wave-level surface gather: [K][64] distances, the ball walk's stack, candidate keys + (position, index) records
    const size_t coop_base = size_t(gather_size) * 64u + kBallStack;
    q.region_dwords = uint32_t(std::max<size_t>({size_t(gather_lds) * 64u * 2u, size_t(kBeamCap) + 64u * 16u, size_t(kSuper) * 4u + 2u * kPendCap,
                                                 gg ? 0u : coop_base + 5u * 160u}));
    q.coop_cap = (gg || !rpti::option_photon_coop_gather(s)) ? 0u : uint32_t(std::min<size_t>(kCoopCap, ((q.region_dwords - coop_base) / 5u) & ~size_t(3)));
    const size_t lds = (bvh ? 32u * 256u * 4u : 0u) + 4u * size_t(q.region_dwords) * 4u;
    if (!pm->d_overflow) RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->d_overflow, 64));
    RPTI_HIP_TRY(hipMemsetAsync(pm->d_overflow, 0, 4, st));
    q.overflow = pm->d_overflow;
    rpt64::SurfArgs64 sa{};
    if (eps) {
        if (!pm->pos64 && pm->surf.n) return rpti::fail(RPT_ERR_STATE, "this photon map was not shot in the reference-epsilon mode");
        const size_t words = std::max<size_t>(size_t(q.r.n_owned) * (size_t(gather_size) + 2u) * num_samples, 16u);
        if (words > pm->emit_words) {
            pm->pool->free(pm->d_emit);
            pm->d_emit = nullptr;
            pm->emit_words = 0;
            RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->d_emit, words * 4u));
            pm->emit_words = words;
        }
        q.emit = pm->d_emit;
        pm->emit_dims[0] = q.r.n_owned; pm->emit_dims[1] = gather_size + 2u; pm->emit_dims[2] = num_samples;
        const uint32_t n_groups = (num_samples + 63u) / 64u;
        const size_t bytes = std::max<size_t>(size_t(n_groups) * q.r.n_owned * 32u, 32u);
        if (bytes > pm->slab64_bytes) {
            pm->pool->free(pm->d_slab64);
            pm->d_slab64 = nullptr;
            pm->slab64_bytes = 0;
            RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->d_slab64, bytes));
            pm->slab64_bytes = bytes;
        }
        rpti::fill_args64(s, cam, prm, &q.r, sa.a);
        sa.a.n_chunks = n_groups;                    // of 64 samples: one work item per (group, pixel)
        sa.a.n_items = n_groups * q.r.n_owned;       // (< 2^32: n_owned * n_chunks of kSuper was checked, and a chunk holds four groups)
        sa.a.slab = pm->d_slab64;
        sa.emit = pm->d_emit;
        sa.s_ph = reinterpret_cast<const rpt64::PhotonRec32*>(pm->surf.sorted);
        sa.pos64 = pm->pos64;
        sa.K = uint32_t(gather_size);
        sa.kind = uint32_t(pm->kind);
        sa.skip = q.skip;
        if (uint64_t(n_groups) * q.r.n_owned >= (1ull << 32)) return rpti::fail(RPT_ERR_INVALID, "too many work items");
    }
    auto launch = [&](const RenderArgs& ra, int nb, hipStream_t stream) -> hipError_t {
        QueryArgs qq = q;
        qq.r = ra;
        if (!eps) {
            launch_query(qq, medium, bvh, gg, pm->kind, nb, lds, stream);
            return hipGetLastError();
        }
        launch_query_emit(qq, medium, bvh, gg, pm->kind, nb, lds, stream);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_photon_surface_f64(sa, rpti::scene_dev(s).n_cus * 4, stream);
    };
    std::function<hipError_t(double, double*, hipStream_t)> resolve;
    if (eps)
        resolve = [&](double scale, double* out, hipStream_t stream) {
            return launch_resolve_photon_f64(sa.a, q.r.slab, q.r.n_chunks, scale / double(eps->total), !eps->first, out, stream);
        };
    int bpc = int(std::max<size_t>(1, std::min<size_t>(4, (160u * 1024u) / std::max<size_t>(lds, 1))));
    if (gg) {  // one [2][K][64]-dword region per wave of the largest grid run_persistent may launch
        const size_t words = size_t(rpti::scene_dev(s).n_cus) * size_t(bpc) * 4u * size_t(gather_max) * 128u;
        if (words > pm->gather_words) {
            pm->pool->free(pm->d_gather);
            pm->d_gather = nullptr;
            pm->gather_words = 0;
            RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->d_gather, words * 4u));
            pm->gather_words = words;
        }
        q.gather = pm->d_gather;
    }
    if (pm->kind == RPT_PHOTON_POINT_BEAM && medium && pm->vol.n && rpti::option_photon_block_lists(s)) {
        const size_t words = size_t(rpti::scene_dev(s).n_cus) * size_t(bpc) * 4u * kCandCap;
        if (words > pm->cand_words) {
            pm->pool->free(pm->d_cand);
            pm->d_cand = nullptr;
            pm->cand_words = 0;
            RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->d_cand, words * 4u));
            pm->cand_words = words;
        }
        q.cand = pm->d_cand;
        q.cand_cap = kCandCap;
    }
    // the split camera pass (option "photon_split"; off by default -- two launches cannot overlap the two estimates the way
    // one kernel's mix of waves does: 112 ms against 95 on C4): a second slab for the surface term
    q.r.slab2 = nullptr;
#ifndef RPT_EXPERIMENTS
    if (rpti::option_photon_split(s)) return rpti::fail(RPT_ERR_UNSUPPORTED, "photon_split: a rejected prototype, built with -DRPT_EXPERIMENTS only");
#endif
    if (medium && !gg && pm->kind != RPT_PHOTON_MAP && rpti::option_photon_split(s)) {
        const size_t bytes = std::max<size_t>(size_t(q.r.n_chunks) * q.r.n_owned * 16u, 16u);
        if (bytes > pm->slab2_bytes) {
            pm->pool->free(pm->d_slab2);
            pm->d_slab2 = nullptr;
            pm->slab2_bytes = 0;
            RPTI_HIP_TRY(pm->pool->alloc((void**)&pm->d_slab2, bytes));
            pm->slab2_bytes = bytes;
        }
        q.r.slab2 = pm->d_slab2;
    }
    rc = rpti::run_persistent(s, prm, q.r, d_out, st, bpc, launch, false, true, resolve);
    if (rc == RPT_OK && sync_counters) {
        uint32_t ov = 0;
        RPTI_HIP_TRY(hipMemcpyAsync(&ov, pm->d_overflow, 4, hipMemcpyDeviceToHost, st));
        RPTI_HIP_TRY(hipStreamSynchronize(st));
        if (ov) return rpti::fail(RPT_ERR_UNSUPPORTED, "photon beam walk: traversal stack overflow (photon tree too deep)");
        if (q.r.counters) rc = rpti::fetch_counters(s, q.r);
    }
    return rc;
}

// Budget of the per-sample selections the reference-epsilon camera pass keeps per slice (bytes of device memory).
static constexpr size_t kEmitBudget = size_t(32) << 30;
static int photon_render_any(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint64_t gather_size,
                             uint64_t gather_size_volume, uint32_t num_samples, uint64_t seed, uint32_t sample_offset, double* d_out,
                             hipStream_t st) {
    if (!s || !rpti::scene_dev(s).epsilon64)
        return photon_render_impl(s, cam, prm, gather_size, gather_size_volume, num_samples, seed, sample_offset, d_out, st, true);
    if (!prm || num_samples == 0) return rpti::fail(RPT_ERR_INVALID, "empty render");
    if (prm->shard_count > 1) {   // (run_persistent clears a sharded frame per launch: the slices would not add up)
        const size_t per_sample = size_t(prm->width) * prm->height * (size_t(gather_size) + 2u) * 4u;
        if (per_sample * num_samples > kEmitBudget)
            return rpti::fail(RPT_ERR_UNSUPPORTED, "reference-epsilon photon camera pass on a sharded frame: too many samples for one slice; render in several calls");
    }
    // slices: whole chunks of kSuper samples while the selections fit the budget, else fewer samples
    const size_t per_sample = std::max<size_t>(size_t((prm->width + 31u) / 32u) * ((prm->height + 31u) / 32u) * 1024u * (size_t(gather_size) + 2u) * 4u, 1);
    uint32_t slice = uint32_t(std::min<uint64_t>(num_samples, std::max<uint64_t>(kEmitBudget / per_sample, 1)));
    if (slice >= kSuper) slice -= slice % kSuper;
    if (const int64_t forced = rpti::option_f64_photon_slice(s)) slice = uint32_t(std::min<int64_t>(forced, slice));
    for (uint32_t done = 0; done < num_samples; done += slice) {
        const uint32_t n = std::min(slice, num_samples - done);
        const EpsSlice e{num_samples, done == 0};
        const int rc = photon_render_impl(s, cam, prm, gather_size, gather_size_volume, n, seed, sample_offset + done, d_out, st, true, &e);
        if (rc) return rc;
    }
    return RPT_OK;
}

int rpt_photon_render_sample(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, uint64_t gather_size,
                             uint64_t gather_size_volume, uint32_t num_samples, uint64_t seed, uint32_t sample_offset,
                             double* out_rgb) {
    if (!s || !cam || !prm || !out_rgb) return rpti::fail(RPT_ERR_INVALID, "null argument");
    size_t bytes = size_t(prm->width) * prm->height * 24;
    double* d_out = rpti::scratch_out(s, bytes);
    if (!d_out) return rpti::fail(RPT_ERR_DEVICE, "out of device memory");
    int rc = photon_render_any(s, cam, prm, gather_size, gather_size_volume, num_samples, seed, sample_offset, d_out, nullptr);
    if (rc) return rc;
    RPTI_HIP_TRY(hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost));
    return RPT_OK;
}

int rpt_photon_render_sample_device(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm,
                                    uint64_t gather_size, uint64_t gather_size_volume, uint32_t num_samples, uint64_t seed,
                                    uint32_t sample_offset, void* d_out_rgb, void* hip_stream) {
    if (!s || !cam || !prm || !d_out_rgb) return rpti::fail(RPT_ERR_INVALID, "null argument");
    return photon_render_any(s, cam, prm, gather_size, gather_size_volume, num_samples, seed, sample_offset,
                             static_cast<double*>(d_out_rgb), static_cast<hipStream_t>(hip_stream));
}


int rpt_debug_photon_positions64(rpt_scene* s, double* out, uint64_t capacity) {
    if (!s || !out) return rpti::fail(RPT_ERR_INVALID, "null argument");
    auto* pm = static_cast<PhotonMapDev*>(rpti::photon_slot(s));
    if (!pm || !pm->built) return rpti::fail(RPT_ERR_STATE, "no photon map: call rpt_photon_map_build first");
    if (pm->surf.n && !pm->pos64) return rpti::fail(RPT_ERR_STATE, "this photon map was not shot in the reference-epsilon mode");
    if (capacity < pm->surf.n) return rpti::fail(RPT_ERR_INVALID, "output buffer too small");
    if (pm->surf.n) RPTI_HIP_TRY(hipMemcpy(out, pm->pos64, size_t(pm->surf.n) * 24u, hipMemcpyDeviceToHost));
    return RPT_OK;
}

int rpt_debug_photon_selections(rpt_scene* s, uint32_t* out, uint64_t capacity_words, uint64_t dims[3]) {
    if (!s || !dims) return rpti::fail(RPT_ERR_INVALID, "null argument");
    auto* pm = static_cast<PhotonMapDev*>(rpti::photon_slot(s));
    if (!pm || !pm->d_emit) return rpti::fail(RPT_ERR_STATE, "no camera pass in the reference-epsilon mode yet");
    for (int i = 0; i < 3; i++) dims[i] = pm->emit_dims[i];
    const uint64_t words = dims[0] * dims[1] * dims[2];
    if (!out) return RPT_OK;
    if (capacity_words < words) return rpti::fail(RPT_ERR_INVALID, "output buffer too small");
    RPTI_HIP_TRY(hipSetDevice(pm->device));
    RPTI_HIP_TRY(hipDeviceSynchronize());
    RPTI_HIP_TRY(hipMemcpy(out, pm->d_emit, words * 4u, hipMemcpyDeviceToHost));
    return RPT_OK;
}

// ---- test hooks of sort_scan.h (host arrays in and out)
int rpt_debug_radix_sort(uint64_t n, const uint64_t* keys, uint64_t* keys_out, uint32_t* order_out) {
    if (n && (!keys || !keys_out || !order_out)) return rpti::fail(RPT_ERR_INVALID, "null argument");
    if (n >= (1ull << 31)) return rpti::fail(RPT_ERR_INVALID, "too many pairs");
    if (n == 0) return RPT_OK;
    const uint32_t m = uint32_t(n);
    uint64_t *k0 = nullptr, *k1 = nullptr;
    uint32_t *v0 = nullptr, *v1 = nullptr;
    char* temp = nullptr;
    auto cleanup = [&]() { (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(temp); };
    std::vector<uint32_t> iota(m);
    for (uint32_t i = 0; i < m; i++) iota[i] = i;
    hipError_t e = hipMalloc((void**)&k0, n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&k1, n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&v0, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v1, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&temp, ss::rsort_temp_bytes(m));
    if (e == hipSuccess) e = hipMemcpy(k0, keys, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v0, iota.data(), n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ss::radix_sort_pairs(k0, v0, k1, v1, m, 8u, temp, ss::StorePair{k0, v0}, nullptr);   // (pass 7 reads k1 / v1)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(keys_out, k0, n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(order_out, v0, n * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return rpti::fail(RPT_ERR_DEVICE, std::string("rpt_debug_radix_sort: ") + hipGetErrorString(e));
    return RPT_OK;
}
int rpt_debug_exclusive_scan2(uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out_a, uint32_t* out_b, uint64_t totals[2]) {
    if (!totals || (n && (!a || !b || !out_a || !out_b))) return rpti::fail(RPT_ERR_INVALID, "null argument");
    if (n >= (1ull << 31)) return rpti::fail(RPT_ERR_INVALID, "too many values");
    const uint32_t m = uint32_t(n);
    uint32_t *da = nullptr, *db = nullptr, *oa = nullptr, *ob = nullptr;
    unsigned long long* dt = nullptr;
    char* temp = nullptr;
    auto cleanup = [&]() { (void)hipFree(da); (void)hipFree(db); (void)hipFree(oa); (void)hipFree(ob); (void)hipFree(dt); (void)hipFree(temp); };
    const size_t bytes = std::max<size_t>(n * 4, 4);
    hipError_t e = hipMalloc((void**)&da, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&db, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&oa, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&ob, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&dt, 16);
    if (e == hipSuccess) e = hipMalloc((void**)&temp, std::max<size_t>(ss::scan2_temp_bytes(m), 8));
    if (e == hipSuccess && n) e = hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) e = hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ss::exclusive_scan2(da, db, m, oa, ob, dt, temp, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && n) e = hipMemcpy(out_a, oa, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && n) e = hipMemcpy(out_b, ob, n * 4, hipMemcpyDeviceToHost);
    unsigned long long t[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return rpti::fail(RPT_ERR_DEVICE, std::string("rpt_debug_exclusive_scan2: ") + hipGetErrorString(e));
    totals[0] = t[0];
    totals[1] = t[1];
    return RPT_OK;
}

}  // extern "C"
