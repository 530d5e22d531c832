// kernels.hip — HIP kernels for gfx950 (MI355X): the persistent path-tracing megakernel,
// the slab resolve, the batched closest-hit test hook and small device self-test kernels.
//
// Megakernel structure (replaces the rayon row loop + recursive trace_ray of
// src/renderer.rs:158-322):
//   * one lane = one in-flight path; a lane owns a work item (pixel, chunk of <= chunk_spp
//     samples), loops over the chunk's samples and, when the item is done, pulls the next one
//     from a global counter with ONE wave-aggregated atomic (ballot + mbcnt prefix);
//   * the recursion is flattened into a single loop: every trip processes exactly one path
//     vertex for every live lane (closest hit -> event -> next-event estimation -> bounce), and
//     a lane whose path ended regenerates its next camera ray in the same trip, so lanes of a
//     wave never wait for the longest path;
//   * radiance is carried forward as h(x) = min(P + Q*x, R) per channel, which is closed under
//     composition with the reference's per-vertex map x -> E + min(k*x, 100)
//     (src/renderer.rs:308-313); in a medium there is no clamp (R = +inf) and it reduces to
//     the usual throughput form (src/renderer.rs:229-232, 271-280);
//   * partial sums go to slab[chunk][pixel] (fp32) and are summed in fp64 in chunk order by
//     resolve_kernel, so the image is bit-identical for any grid size, schedule or GPU count.
#include "device_core.h"
#include "kernels.h"

namespace rptg {

RPT_DEV uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}
RPT_DEV uint32_t code_object(const SceneView& sc, uint32_t code, uint32_t inst) {
    uint32_t kind = code >> 28, idx = code & 0x0FFFFFFFu;
    float w;
    if (kind == K_INSTTRI) w = sc.inst[inst].n0.w;
    else if (kind == K_SPHERE) w = sc.sph_sh[idx].r0.w;
    else if (kind == K_CUBE) w = sc.cub_sh[idx].r0.w;
    else if (kind == K_PLANE) w = sc.pln_sh[idx].unit_n_obj.w;
    else if (kind == K_TRI) w = sc.tri_sh[idx].n1.w;
    else if (kind == K_AABB) w = sc.aabb[idx].lo.w;
    else if (kind == K_RECT) w = sc.rect_sh[idx].n_obj.w;
    else w = sc.btri_sh[idx].n1.w;
    return __float_as_uint(w);
}
// t_min stand-in for the reference's EPSILON = 1e-12 (src/renderer.rs:17, 420), which is below
// fp32 resolution: scaled to the magnitude of the ray origin.
RPT_DEV float ray_tmin(V o) { return 2e-5f * (1.f + max3(fabsf(o.x), fabsf(o.y), fabsf(o.z))); }

RPT_DEV void item_pixel(const RenderArgs& a, uint32_t p, uint32_t& x, uint32_t& y) {
    uint32_t tl = p >> 10, within = p & 1023u, sb = within >> 6, l = within & 63u;
    uint32_t tile = a.tiles[tl];
    uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    x = tx * 32u + (sb & 3u) * 8u + (l & 7u);
    y = ty * 32u + (sb >> 2) * 8u + (l >> 3);
}

#ifndef RPT_MIN_WAVES
#define RPT_MIN_WAVES 4       // waves per SIMD the BVH instantiations are compiled for (128 VGPRs)
#endif
#ifndef RPT_MIN_WAVES_MESH
#define RPT_MIN_WAVES_MESH 4  // per-mesh-tree instantiations (BVH = 1)
#endif
#ifndef RPT_MIN_WAVES_SCENE_MESH
#define RPT_MIN_WAVES_SCENE_MESH 3  // scene tree + per-mesh trees (BVH = 3): at 4 (128 VGPRs) the walk of the scene tree inlined into the parked-walk
                                    // machine costs 112 B/lane of scratch -- a footprint of 3.7 MB per XCD against 4 MB of L2: 647 GB of writes per
                                    // C5G frame -- and 253.9 ms at 2048x2048x64; 3 (168 VGPRs, no scratch): 219.5
#endif
#ifndef RPT_MIN_WAVES_STREAM
#define RPT_MIN_WAVES_STREAM 4  // per-mesh-tree instantiation with streamed walks (DETACH = 2); 3 (168 VGPRs) has no scratch
#endif
#ifndef RPT_MIN_WAVES_SCAN
#define RPT_MIN_WAVES_SCAN 6  // linear-scan instantiations (BVH = 0): 80 VGPRs.  Without a medium no scratch; in a medium 24 B/lane (9 MB over
                              // the chip: it stays in L2).  C3 21.4 ms against 22.6 at 5 waves (96 VGPRs), C2 1.11 against 1.17; 7 waves
                              // (72 VGPRs, 52 / 28 B of scratch) 21.5 / 1.09: no further gain.  (Before the build dropped SLP vectorisation
                              // the medium flavour spilled 84 B/lane at 6 waves -- 32 GB of HBM writes per C3 launch -- and 5 was the setting.)
#endif
#ifndef RPT_MIN_WAVES_SCAN_GROUPS
#define RPT_MIN_WAVES_SCAN_GROUPS 5  // ... with group lights or counters (GROUPS / COUNT): 64 / 48 B of scratch at 6 waves, none at 5
#endif
// LDS of the per-mesh-tree kernel with detached shadow queries (render_kernel<true, 1, *, false, true>), in dwords per block
// of 256 lanes: [stack rows][5 state rows][staged tables][pend: 256][acc: 3 x 256 u64][4 wave queues].  40,320 B: four
// blocks per CU fit the 160 KB.
static constexpr uint32_t kDetachStackRows = 20u;   // a mesh tree is at most 20 levels deep: it pushes at most 19 entries
static constexpr uint32_t kQCap = 32u;              // detached shadow queries a wave can hold
static constexpr uint32_t kQFields = 11u;           // origin, direction, end of the interval, contribution if visible, owner lane
static constexpr uint32_t kQWaveDwords = 16u + kQFields * kQCap;   // [0]: slots in use, [1]: slots being walked (bit masks); fields from [16], field-major
static_assert(kQCap == 32u, "one mask bit per slot");
static constexpr uint32_t kTabDwords = (2u * 32u + 6u * 8u) * 4u;   // LdsTables: 32 materials + 8 light triangles
static constexpr uint32_t kDetachPendBase = (kDetachStackRows + 5u) * 256u + kTabDwords;
static constexpr uint32_t kDetachAccBase = kDetachPendBase + 256u;
static constexpr uint32_t kDetachQBase = kDetachAccBase + 3u * 256u * 2u;
static constexpr uint32_t kDetachDwords = kDetachQBase + 4u * kQWaveDwords;
static_assert(kDetachAccBase % 2u == 0u, "the 64-bit accumulators are 8-byte aligned");
// DETACH = 2 ("streamed walks"): per wave, in global memory (RenderArgs::stream_scratch; only this wave ever touches its
// part), a FIFO ring of queries and kCtxMax parked path contexts per lane.
#ifndef RPT_STREAM_RING
#define RPT_STREAM_RING 512
#endif
static constexpr uint32_t kRingEntries = RPT_STREAM_RING;   // queries a wave can hold; a session is forced before the ring could overflow
static constexpr uint32_t kRingDwords = 12u;     // origin, direction | end of the interval, 3 words by kind | meta, pad (3 x 16 B)
static constexpr uint32_t kCtxMax = 6u;          // parked paths per lane (RenderArgs::stream_contexts of them are used)
static constexpr uint32_t kCtxFields = 20u;      // origin, direction, P, Q, RNG state, depth, medium distance, answer (t, code)
static constexpr uint32_t kWaveScratchDwords = kRingEntries * kRingDwords + kCtxMax * kCtxFields * 64u;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
static_assert(kDetachDwords * 4u * 4u <= 160u * 1024u, "four blocks per CU");
// Radiance -> unsigned 32.32 fixed point (negative values and NaN -> 0, values from 2^32 up saturate).  Sums of such
// numbers do not depend on the order of the additions: that is what lets a detached shadow query add its term to
// its pixel whenever its walk happens to run.
RPT_DEV unsigned long long to_fixed(float v) {
    v = fminf(fmaxf(v, 0.f), 4294967040.f);
    const uint32_t hi = uint32_t(v);
    const uint32_t lo = uint32_t((v - float(hi)) * 4294967296.f);   // exact: the fraction has at most 24 bits
    return ((unsigned long long)hi << 32) | lo;
}
RPT_DEV float from_fixed(unsigned long long x) { return float(double(x) * 0x1p-32); }

// Diagnostic sections of the megakernel (COUNT build): per section, counters[8 + 2k] counts wave-level
// executions and counters[9 + 2k] the lanes active in them (lane utilisation of divergent code).
// -DRPT_MARKERS additionally drops "; SECT k" comments into the ISA for static instruction counts.
#ifdef RPT_MARKERS
#define RPT_MARK(k) asm volatile("; SECT " #k ::: "memory")
#else
#define RPT_MARK(k)
#endif
#ifdef RPT_SECT_CLOCKS
#define SECT(k) do { } while (0)   // (the clocks build keeps its counts per wave in LDS: global atomics would be what it measures)
#else
#define SECT(k)                                                                                  \
    do {                                                                                         \
        RPT_MARK(k);                                                                             \
        if (COUNT) {                                                                             \
            const uint64_t m_ = __ballot(true);                                                  \
            if (mbcnt64(m_) == 0) {                                                              \
                atomicAdd(&a.counters[8 + 2 * (k)], 1ull);                                       \
                atomicAdd(&a.counters[9 + 2 * (k)], (unsigned long long)__popcll(m_));           \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#endif
// SECTK: the same inside render_kernel's body.  -DRPT_SECT_CLOCKS (a diagnostic build of its own, tools/sect_clocks.py): the
// lanes slot of a section then holds wave clock ticks (s_memtime) instead -- the time from reaching that section until the
// wave reaches the next section point, stalls included.
#ifdef RPT_SECT_CLOCKS
#define SECTK(k)                                                                                 \
    do {                                                                                         \
        if (COUNT) {                                                                             \
            const uint64_t now_ = __builtin_amdgcn_s_memtime();                                  \
            if (mbcnt64(__ballot(true)) == 0) {                                                  \
                sect_lds_[threadIdx.x >> 6][2 * (k)] += 1ull;                                    \
                sect_lds_[threadIdx.x >> 6][2 * sect_prev_ + 1] += now_ - sect_clk_;             \
            }                                                                                    \
            sect_prev_ = (k);                                                                    \
            sect_clk_ = now_;                                                                    \
        }                                                                                        \
    } while (0)
#else
#define SECTK(k) SECT(k)
#endif
// ---- The stages of a path vertex, shared by the two loop bodies of render_kernel (the parked-walk state machine of the
// per-mesh-tree flavour and the lock-step body of the others): what they compute is one thing, when they run another.
// Distance sample of a new vertex and the interval its closest-hit query has to search (Medium::sample_d,
// src/medium.rs:133-146).  A hit beyond the sampled medium distance cannot change the event (dmed < t, or a miss with
// dmed < 400, is a medium event either way, src/renderer.rs:197-243): the search ends there, which culls most of a
// tree walk in fog.  The margin keeps the `dmed < t` comparison below the one that decides.
template <bool MEDIUM>
RPT_DEV void stage_distance(Rng& rng, float inv_sigma_t, float& dmed, float& t) {
    dmed = kInf;
    if (MEDIUM) {
        float xi = rng.range(0.f, 1.f);
        dmed = -__logf(xi) * inv_sigma_t;
    }
    t = (MEDIUM && dmed < 400.f) ? dmed * (1.f + 1e-6f) : kInf;
}
// The event at a medium point or a surface hit: position, what shading needs, emission (src/renderer.rs:207-216,
// 243-255, 289-299).
template <bool COUNT>
RPT_DEV void stage_event(const RenderArgs& a, const LdsTables& tab, V ro, V rd, uint32_t depth, bool medium, float dmed, float t,
                         uint32_t code, uint32_t inst, V& x, V& n, V& mcol, Mat& mat, V& E) {
    const SceneView& sc = a.sc;
    if (medium) {
        SECT(5);
        x = fma3(dmed, rd, ro);
        bool hi = sc.medium_kind == 1u && x.y > 250.f;
        mcol = hi ? mk(sc.medium_color_hi[0], sc.medium_color_hi[1], sc.medium_color_hi[2])
                  : mk(sc.medium_color[0], sc.medium_color[1], sc.medium_color[2]);
        E = (depth == 0) ? sc.medium_emission * mcol : mk(0, 0, 0);
    } else {
        uint32_t obj;
        SECT(6);
        finalize_hit(sc, ro, rd, ray_tmin(ro), t, code, inst, n, obj);
        mat = load_mat(sc, obj, tab);
        x = fma3(t, rd, ro);
        E = (depth == 0) ? mat_emit(mat) * mat_color(mat) : mk(0, 0, 0);
    }
}
// The shadow test of an object light and its term of E (src/renderer.rs:347-353, 395-404).  Reference: contributes
// iff the closest hit along wi lies at dist_to_light (|hit - dist| < 1e-12).  fp32 equivalent: the closest hit
// belongs to the scene object that IS this light, at the sampled distance (rel. tol 1e-3).
RPT_DEV void stage_light_term(const SceneView& sc, const Light& L, float albedo_med, V rd, bool medium, float ts, uint32_t cs,
                              uint32_t is, float dist, V I, V wi, V n, V mcol, const Mat& mat, V& E) {
    const bool twin = (L.twin_lo <= L.twin_hi) ? (cs >= L.twin_lo && cs <= L.twin_hi)   // wave-uniform choice
                                               : (cs != CODE_MISS && code_object(sc, cs, is) == uint32_t(L.twin_object));
    if (cs != CODE_MISS && ts >= dist * (1.f - 1e-3f) && twin) {
        if (medium) {
            E = fma3(albedo_med * sc.medium_phase, I * mcol, E);
        } else {
            V f = bsdf(mat, n, -normalize(rd), wi);   // (wo is only needed at surface events: derived where used)
            E = fma3(dot(wi, n), f * I, E);
        }
    }
}
// Continue or end: Russian roulette / max_bounces, phase or BSDF sample, path weight (src/renderer.rs:222-232, 262-281, 301-313)
template <bool MEDIUM, bool COUNT>
RPT_DEV bool stage_bounce(const RenderArgs& a, float albedo_med, V rd, uint32_t depth, bool medium, V n, V mcol, const Mat& mat,
                          Rng& rng, V& wi, V& k) {
    bool bounce;
    SECT(10);
    if (medium) {
        SECT(11);
        bounce = rng.uniform() < 0.8f;
        if (bounce) {
            float ax = rng.range(-1.f, 1.f), ay = rng.range(-1.f, 1.f), az = rng.range(-1.f, 1.f);
            wi = normalize(mk(ax, ay, az));        // src/medium.rs:87-93 (cube, then normalise)
            k = (albedo_med * 1.25f) * mcol;       // (scat/ext) / ph_p * phase / rr_p, ph_p == phase
        }
    } else {
        SECT(12);
        bounce = MEDIUM ? (rng.uniform() < 0.8f) : (depth < a.max_bounces);  // :222 / :301
        if (bounce) {
            float pdf;
            SECT(13);
            const V wo = -normalize(rd);
            bounce = sample_f(mat, n, wo, rng, wi, pdf);
            if (bounce) {
                V f = bsdf(mat, n, wo, wi);
                float wgt = fabsf(dot(wi, n)) * rcp(MEDIUM ? pdf * 0.8f : pdf);
                k = wgt * f;
            }
        }
    }
    return bounce && !is_zero(k);
}

// GROUPS: some Light::Object is a KdTree group (per-lane leaf sampler).  A separate instantiation: the extra
// sampler copy costs the plain kernels 6 % through register allocation alone, and a call costs 6x.
// DETACH (per-mesh-tree kernels in a medium): shadow queries that need a tree walk leave their path (see the loop body).
// DETACH = 2: primary queries leave as well -- their paths wait in memory and the lane goes on with another one.
template <bool MEDIUM, int BVH, bool COUNT, bool GROUPS = false, int DETACH = 0>
__global__ __launch_bounds__(256, BVH == 0 ? ((GROUPS || COUNT) ? RPT_MIN_WAVES_SCAN_GROUPS : RPT_MIN_WAVES_SCAN) : BVH == 3 ? RPT_MIN_WAVES_SCENE_MESH : BVH == 1 ? (DETACH == 2 ? RPT_MIN_WAVES_STREAM : RPT_MIN_WAVES_MESH) : RPT_MIN_WAVES)
void render_kernel(const RenderArgs a) {
#ifdef RPT_SECT_CLOCKS
    __shared__ unsigned long long sect_lds_[4][56];   // per wave: [2k] visits of section k, [2k + 1] its ticks
    if ((threadIdx.x & 63u) < 56u) sect_lds_[threadIdx.x >> 6][threadIdx.x & 63u] = 0ull;
    uint64_t sect_clk_ = __builtin_amdgcn_s_memtime();
    uint32_t sect_prev_ = 27u;   // (time before the first section point)
#endif
    static_assert(DETACH == 0 || (MEDIUM && BVH == 1 && !GROUPS), "detached tree walks: per-mesh-tree kernels in a medium only");
    extern __shared__ uint32_t dyn_lds[];
    const SceneView& sc = a.sc;
    uint32_t* stk = BVH ? (dyn_lds + threadIdx.x) : nullptr;
    const uint32_t stride = 256;

    const float sigma_t = sc.sigma_a + sc.sigma_s;
    const float inv_sigma_t = MEDIUM ? 1.f / sigma_t : 0.f;
    const float albedo_med = MEDIUM ? sc.sigma_s / sigma_t : 0.f;

    Rng rng;
    rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    V ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    V P = mk(0, 0, 0), Q = mk(1, 1, 1), Rc = mk(kInf, kInf, kInf);
    // Per-lane state that is touched only when a path ends or a sample starts (the item's partial sum, its slab
    // slot and sample range, the pixel) lives in LDS in the scan instantiations: they are compiled for 5 waves
    // per SIMD (96 VGPRs), LDS is otherwise unused there, and a value parked in LDS is a spill that costs one
    // ds instruction instead of a trip to L2 (register spills of that build ran at 39 GB of HBM writes per launch).
    // The tree-walking instantiations (128 VGPRs, 32 KB of LDS stack per block) park only the five values that
    // are read once per sample: 4 blocks x (32 + 5) KB still fit the CU's 160 KB.
    // rows of the traversal stack: scene tree + mesh tree need up to 32; a mesh tree alone is at most 20 levels deep
    constexpr uint32_t kStackRows = BVH == 1 ? (DETACH ? kDetachStackRows : 21u) : 32u;
    constexpr uint32_t kStateBase = BVH ? kStackRows * 256u : 0u;  // dwords: after the traversal stack
    uint32_t* const ls = dyn_lds + kStateBase + threadIdx.x;  // [slots][256] dwords, one column per lane
    enum { S_SLAB = 0, S_END = 1, S_PIX = 2, S_XN = 3, S_YN = 4, S_S = 5, S_ACC = 6, S_P = 9, S_Q = 12, S_RC = 15, S_ROWS = 18 };
    static_assert(S_RC + 3 == S_ROWS, "the staged tables begin right behind the state rows");
    // Material and light-triangle tables of small scenes, staged once per block behind the lane state
    constexpr uint32_t kTabBase = kStateBase + (BVH == 0 ? uint32_t(S_ROWS) : 5u) * 256u;   // dwords
    LdsTables tab;
    {
        const F4* const t4 = reinterpret_cast<const F4*>(dyn_lds + kTabBase);
        F4* const w4 = reinterpret_cast<F4*>(dyn_lds + kTabBase);
        const uint32_t nm = sc.n_obj <= kLdsMats ? sc.n_obj : 0u, nl = sc.n_ltris <= kLdsLtris ? sc.n_ltris : 0u;
        for (uint32_t i = threadIdx.x; i < 2u * nm; i += 256u) w4[i] = reinterpret_cast<const F4*>(sc.mats)[i];
        for (uint32_t i = threadIdx.x; i < 6u * nl; i += 256u) w4[2u * kLdsMats + i] = reinterpret_cast<const F4*>(sc.ltris)[i];
        __syncthreads();
        tab.mats = t4; tab.n_mats = nm;
        tab.ltris = t4 + 2u * kLdsMats; tab.n_ltris = nl;
    }
    auto in_lds = [](int k) { return BVH == 0 || k <= S_YN; };
    V acc_r = mk(0, 0, 0);
    uint32_t slab_idx_r = 0, s_r = 0, s_end_r = 0, pix_r = 0;
    float xn_r = 0.f, yn_r = 0.f;
    auto ldu = [&](int k, uint32_t reg) { return in_lds(k) ? ls[k * 256] : reg; };
    auto ldf = [&](int k, float reg) { return in_lds(k) ? __uint_as_float(ls[k * 256]) : reg; };
    auto stu = [&](int k, uint32_t& reg, uint32_t v) { if (in_lds(k)) ls[k * 256] = v; else reg = v; };
    auto stf = [&](int k, float& reg, float v) { if (in_lds(k)) ls[k * 256] = __float_as_uint(v); else reg = v; };
    // DETACH: the item's sum is kept in fixed point in LDS (acc64[channel][lane]) where other lanes of the wave can add to
    // it; pend = detached shadow queries of the lane's item that are still in the wave's queue
    volatile uint32_t* const ls_pend = dyn_lds + (DETACH ? kDetachPendBase : 0u);                      // [256]
    unsigned long long* const acc64 = reinterpret_cast<unsigned long long*>(dyn_lds + (DETACH ? kDetachAccBase : 0u));   // [3][256]
    volatile uint32_t* const wq = dyn_lds + (DETACH ? kDetachQBase + (threadIdx.x >> 6) * kQWaveDwords : 0u);   // this wave's queue (DETACH = 2: its header words only)
    auto acc_add = [&](V v) {
        if constexpr (DETACH) {
            atomicAdd(&acc64[threadIdx.x], to_fixed(v.x));
            atomicAdd(&acc64[256u + threadIdx.x], to_fixed(v.y));
            atomicAdd(&acc64[512u + threadIdx.x], to_fixed(v.z));
        } else {
            stf(S_ACC + 0, acc_r.x, ldf(S_ACC + 0, acc_r.x) + v.x);
            stf(S_ACC + 1, acc_r.y, ldf(S_ACC + 1, acc_r.y) + v.y);
            stf(S_ACC + 2, acc_r.z, ldf(S_ACC + 2, acc_r.z) + v.z);
        }
    };
    if constexpr (DETACH) {
        if ((threadIdx.x & 63u) < 16u) wq[threadIdx.x & 63u] = 0u;   // (DETACH = 2 keeps head, tail, low-water mark and the ready masks there)
        ls_pend[threadIdx.x] = 0u;
    }
    // The radiance carrier (P, Q, Rc) is read and written once per vertex: in the scan instantiations it lives in LDS
    // as well, which takes 6 (9 without a medium) long-lived values out of the 96-VGPR budget.
    auto ldv = [&](int k, V reg) { return BVH == 0 ? mk(__uint_as_float(ls[k * 256]), __uint_as_float(ls[(k + 1) * 256]), __uint_as_float(ls[(k + 2) * 256])) : reg; };
    auto stv = [&](int k, V& reg, V v) {
        if (BVH == 0) { ls[k * 256] = __float_as_uint(v.x); ls[(k + 1) * 256] = __float_as_uint(v.y); ls[(k + 2) * 256] = __float_as_uint(v.z); }
        else reg = v;
    };
    uint32_t depth = 0;
    bool alive = true, have_item = false, need_path = true;
    bool item_done = true;  // the lane's item has no samples left (s >= s_end)
    bool drained = false;  // wave-uniform: the global queue is exhausted
    bool first_batch = true;  // wave-uniform
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform cursor into the current batch of work items
    uint32_t pool_chunk = 0, pool_x0 = 0, pool_y0 = 0;  // wave-uniform: the batch's chunk and 8x8 block origin

    uint32_t c_samples = 0, c_rays = 0, c_vertices = 0, c_trips = 0, c_nodes = 0, c_btris = 0;
    uint32_t c_nodes_primary = 0;   // (detached kernel) tree nodes visited by the parked primary walks in the sessions
    uint32_t c_wave[2] = {0, 0}, w_tot[2] = {0, 0};   // steps of the deferred walks (descent, triangles): of the walk at hand as its lanes count them; wave-level totals

    // ---- per-mesh-tree flavours (BVH == 1; BVH == 3: the same beside a scene tree that holds everything else): deferred walks.  In fog most rays never reach a mesh, a few walk
    // a hundred nodes, and a wave walks as long as its slowest lane: 43 wave-level node steps per query for 3.4
    // nodes per ray (8 % of the lanes busy) on C5.  So a query is split: the scan and a test of the mesh roots'
    // child boxes run at once; a lane that really has to walk parks the query and waits (phase PH_WAIT*) while the
    // rest of the wave goes on with its paths; when `defer_lanes` lanes wait (or nothing else can run) they walk
    // together.  Every lane still computes exactly the same sequence: the image is bit-identical.
    enum : uint32_t { PH_NEW = 0, PH_HAVEP = 1, PH_LIGHT = 2, PH_HAVES = 3, PH_WAITP = 4, PH_WAITS = 5 };
    uint32_t phase = PH_NEW, q_code = CODE_MISS, q_inst = 0, v_li = 0;
    float q_t = kInf, v_dmed = kInf, v_dist = 0.f;
    V v_x = mk(0, 0, 0), v_n = mk(0, 1, 0), v_mcol = mk(0, 0, 0), v_E = mk(0, 0, 0), v_I = mk(0, 0, 0), v_wi = mk(0, 0, 1);
    Mat v_mat = Mat{mk(0, 0, 0), 0.f, 0u, 0.f, 0.f};
    bool v_medium = false;
    WalkState walk{kWalkDone, 0u, 0u};
    uint32_t fuse = 0u;   // (DETACH) trips of this wave
    constexpr uint32_t kNoEntry = 0xFFFFFFFFu;
    uint32_t h_e = kNoEntry;   // (DETACH) queue entry whose unfinished walk this lane holds
    // (DETACH = 2) this wave's memory: the ring of queries and the parked path contexts [context][field][lane]
    uint32_t* const ring_base = DETACH == 2 ? a.stream_scratch + size_t(blockIdx.x * 4u + (threadIdx.x >> 6)) * kWaveScratchDwords : nullptr;
    uint32_t* const ctx_base = DETACH == 2 ? ring_base + kRingEntries * kRingDwords : nullptr;
    uint32_t parked_mask = 0u;   // contexts of this lane that wait in memory
    bool did_work = false;       // this lane took a step in this trip

    for (;;) {
        // ---- work distribution (wave-convergent).  A wave draws batches of 64 items from the global
        // 64-bit counter with ONE atomic per batch and hands them to the lanes that finished their
        // item through a ballot/mbcnt prefix; the batch cursor lives in wave-uniform registers.
        // (One atomic per lane-pull saturated the counter at ~80 M dequeues/s: profiles/r01.)
        bool want = alive && need_path && item_done;
        // DETACH: an item is written out once the last of its detached shadow queries has been answered
        if constexpr (DETACH != 0) want = want && (!have_item || ls_pend[threadIdx.x] == 0u);
        if constexpr (DETACH == 2) {
            // a parked path of this lane whose walk has been answered comes back first (see the loop body)
            if (alive && need_path && parked_mask != 0u) {
                const uint32_t lane = threadIdx.x & 63u;
                uint32_t rc = kCtxMax;
                for (uint32_t c = 0; c < a.stream_contexts; c++) {
                    const uint32_t rdy = wq[4u + 2u * c + (lane >> 5)];
                    if (rc == kCtxMax && ((parked_mask >> c) & 1u) && ((rdy >> (lane & 31u)) & 1u)) rc = c;
                }
                if (rc != kCtxMax) {
                    const uint32_t* const cx = ctx_base + rc * kCtxFields * 64u + lane;
                    auto ldn = [&](uint32_t f) { return __builtin_nontemporal_load(cx + f * 64u); };
                    ro = mk(__uint_as_float(ldn(0)), __uint_as_float(ldn(1)), __uint_as_float(ldn(2)));
                    rd = mk(__uint_as_float(ldn(3)), __uint_as_float(ldn(4)), __uint_as_float(ldn(5)));
                    P = mk(__uint_as_float(ldn(6)), __uint_as_float(ldn(7)), __uint_as_float(ldn(8)));
                    Q = mk(__uint_as_float(ldn(9)), __uint_as_float(ldn(10)), __uint_as_float(ldn(11)));
                    rng.s0 = ldn(12); rng.s1 = ldn(13); rng.s2 = ldn(14); rng.s3 = ldn(15);
                    depth = ldn(16);
                    v_dmed = __uint_as_float(ldn(17));
                    q_t = __uint_as_float(ldn(18));
                    q_code = ldn(19);
                    parked_mask &= ~(1u << rc);
                    atomicAnd(const_cast<uint32_t*>(wq) + 4u + 2u * rc + (lane >> 5), ~(1u << (lane & 31u)));
                    phase = PH_HAVEP;
                    need_path = false;
                    did_work = true;
                }
            }
            // (a lane that has just taken a path back is not looking for an item, whatever it was a moment ago)
            want = want && need_path && parked_mask == 0u;
        }
        // The bookkeeping below costs the wave the same whether one lane or sixty-four want an item, so lanes wait until
        // `pull_batch` of them do -- or until no lane of the wave has anything else to do.  Which lane renders an item does
        // not change the item's sum (same frame bits for every setting).  Measured (tools/pull_batch_sweep.py): C3 21.78 ms
        // at 1, 21.49 at 2, 21.51 at 4, 21.9 at 8, 24.7 at 24; C5 flat up to 2, then slower -- waiting lanes cost more than
        // the saved visits beyond 2, which is the default.
        const uint32_t n_want = uint32_t(__popcll(__ballot(want)));
        if (n_want != 0u && (n_want >= a.pull_batch || __ballot(alive && !want) == 0ull)) {
            const auto& ka = *kernarg_args<RenderArgs>();   // item bookkeeping reads its arguments here, not from registers held since kernel entry
            SECTK(0);
            if (want && have_item) {
                if constexpr (DETACH) {
                    const volatile unsigned long long* av = acc64;
                    reinterpret_cast<float4*>(ka.slab)[ldu(S_SLAB, slab_idx_r)] =
                        make_float4(from_fixed(av[threadIdx.x]), from_fixed(av[256u + threadIdx.x]), from_fixed(av[512u + threadIdx.x]), 0.f);
                } else {
                    reinterpret_cast<float4*>(ka.slab)[ldu(S_SLAB, slab_idx_r)] =
                        make_float4(ldf(S_ACC + 0, acc_r.x), ldf(S_ACC + 1, acc_r.y), ldf(S_ACC + 2, acc_r.z), 0.f);
                }
                have_item = false;
            }
            for (;;) {
                const uint64_t m = __ballot(want);
                if (m == 0) break;
                if (pool_next == pool_end) {
                    // once this wave has seen the queue run dry it never touches the counter again: at the end
                    // of a launch every lane of every wave retires through here, and 4096 waves x 64 atomics on
                    // one address (~88 dequeues/us) used to cost ~1 ms per launch
                    // The first batch of a wave is its own index (the host starts the counter behind them): no
                    // 5120-way pile-up on the counter when the grid starts.  (Reading the counter before the
                    // atomic, to spare the last one per wave, made the launch 2.6x slower: an sc1 load of a
                    // line under atomic traffic is far more expensive than the atomic it saves.)
                    unsigned long long base = ~0ull;
                    if (first_batch) {
                        base = (unsigned long long)(blockIdx.x * 4u + (threadIdx.x >> 6)) * 64ull;
                        first_batch = false;
                    } else if (!drained && (threadIdx.x & 63u) == 0) {
                        base = atomicAdd(ka.queue, 64ull);
                    }
                    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(base));
                    const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(base >> 32));
                    if (hi != 0 || lo >= ka.n_items) {
                        drained = true;  // queue exhausted: the waiting lanes retire
                        if (want) alive = false;
                        break;
                    }
                    pool_next = lo;
                    pool_end = min(lo + 64u, ka.n_items);
                    // a batch is 64-aligned and n_owned is a multiple of 1024: its 64 items are one 8x8 pixel
                    // block of one chunk, so the decode (two divisions, a table load) is done once, wave-uniformly
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
                    const uint32_t chunk = pool_chunk, l = item & 63u;
                    const uint32_t x = pool_x0 + (l & 7u), y = pool_y0 + (l >> 3);
                    if (x < ka.width && y < ka.height) {  // slots of clipped tiles lie outside the image
                        want = false;
                        have_item = true;
                        item_done = false;
                        stu(S_SLAB, slab_idx_r, item);
                        if constexpr (DETACH) {
                            volatile unsigned long long* av = acc64;
                            av[threadIdx.x] = 0ull; av[256u + threadIdx.x] = 0ull; av[512u + threadIdx.x] = 0ull;
                        } else {
                            stf(S_ACC + 0, acc_r.x, 0.f);
                            stf(S_ACC + 1, acc_r.y, 0.f);
                            stf(S_ACC + 2, acc_r.z, 0.f);
                        }
                        const uint32_t s0 = chunk * ka.chunk_spp;
                        stu(S_S, s_r, s0);
                        stu(S_END, s_end_r, min(s0 + ka.chunk_spp, ka.iterations));
                        stu(S_PIX, pix_r, y * ka.width + x);
                        // src/renderer.rs:174-176
                        stf(S_XN, xn_r, (float(2u * x + 1u) - float(ka.width)) * ka.inv_dim);
                        stf(S_YN, yn_r, (float(2u * (ka.height - y) - 1u) - float(ka.height)) * ka.inv_dim);
                    }
                }
            }
        }
        // (a lane whose item is finished has no sample to start: it waits for the wave's next item hand-out, or -- DETACH -- for
        // the item's last shadow answers; DETACH = 2: nor has a lane with every context parked)
        if (need_path && alive && !item_done && !(DETACH == 2 && uint32_t(__popc(parked_mask)) >= a.stream_contexts)) {
            if (alive) {  // src/renderer.rs:179-181
                const auto& ka = *kernarg_args<RenderArgs>();
                SECTK(1);
                const uint32_t s = ldu(S_S, s_r);
                rng.seed(ka.seed_mixed, ldu(S_PIX, pix_r), ka.sample_offset + s);
                float dx = rng.range(-ka.inv_dim, ka.inv_dim);
                float dy = rng.range(-ka.inv_dim, ka.inv_dim);
                const CameraG cam = kernarg_load(&ka.cam);
                cast_ray(cam, ldf(S_XN, xn_r) + dx, ldf(S_YN, yn_r) + dy, rng, ro, rd);
                depth = 0;
                if constexpr (BVH == 0) {
                    stv(S_P, P, mk(0, 0, 0));
                    stv(S_Q, Q, mk(1, 1, 1));
                    if (!MEDIUM) stv(S_RC, Rc, mk(kInf, kInf, kInf));
                } else {
                    P = mk(0, 0, 0);
                    Q = mk(1, 1, 1);
                    Rc = mk(kInf, kInf, kInf);
                }
                stu(S_S, s_r, s + 1u);
                item_done = s + 1u >= ldu(S_END, s_end_r);
                need_path = false;
                if (DETACH == 2) { phase = PH_NEW; did_work = true; }
                if (COUNT) c_samples++;
            }
        }
        if (!__any(alive)) break;
        if (COUNT && (threadIdx.x & 63u) == 0) c_trips++;
        if constexpr (DETACH == 2) {
            // ---- per-mesh-tree flavour in a medium, every tree walk streamed.  A query that has to walk a tree -- primary or
            // shadow -- is written to a ring of this wave in memory and leaves its lane.  A shadow query carries its
            // contribution and is added to its item's sum when answered (as with DETACH = 1).  The path of a primary
            // query is parked: its state (20 dwords) goes to one of the lane's kCtxMax context slots in memory, and the lane
            // goes on with another of its parked paths whose answer has arrived, or starts the item's next sample.  So a
            // wave holds up to 4 x 64 paths for its 64 lanes, the stages run for the lanes that have a path to advance,
            // and a walk session has every lane walking: all 64 take queries from the ring and take the next one when
            // theirs is done, until the ring is empty.  Paths are independent (own RNG state) and the item sums are fixed
            // point, so the image does not depend on the order in which anything is answered: still bit-identical.
            // ---- A: a new path vertex: distance sample, scan, do the trees matter?
            if (alive && !need_path && phase == PH_NEW) {
                if (COUNT) c_vertices++;
                SECTK(2);
                did_work = true;
                stage_distance<MEDIUM>(rng, inv_sigma_t, v_dmed, q_t);
                const float tmin = ray_tmin(ro);
                q_code = CODE_MISS;
                scan_prims(sc, ro, rd, tmin, q_t, q_code);
                if (COUNT) c_rays++;
                phase = PH_HAVEP;
                if (mesh_roots_hit(sc, ro, rd, tmin, q_t)) {
                    SECTK(21);
                    const uint32_t lane = threadIdx.x & 63u;
                    const uint32_t c = uint32_t(__builtin_ctz(~parked_mask));   // a free context (the caller of this stage had one)
                    // the query
                    const uint64_t m = __ballot(true);
                    uint32_t base = 0u;
                    if (mbcnt64(m) == 0u) base = atomicAdd(const_cast<uint32_t*>(wq), uint32_t(__popcll(m)));
                    base = __builtin_amdgcn_readfirstlane(base);
                    uint32_t* const e = ring_base + ((base + mbcnt64(m)) % kRingEntries) * kRingDwords;
                    reinterpret_cast<f4v*>(e)[0] = f4v{ro.x, ro.y, ro.z, rd.x};
                    reinterpret_cast<f4v*>(e)[1] = f4v{rd.y, rd.z, q_t, __uint_as_float(q_code)};
                    reinterpret_cast<u4v*>(e)[2] = u4v{0u, 0u, lane | (c << 6) | (1u << 12), 0u};
                    // the path
                    uint32_t* const cx = ctx_base + c * kCtxFields * 64u + lane;
                    cx[0 * 64] = __float_as_uint(ro.x); cx[1 * 64] = __float_as_uint(ro.y); cx[2 * 64] = __float_as_uint(ro.z);
                    cx[3 * 64] = __float_as_uint(rd.x); cx[4 * 64] = __float_as_uint(rd.y); cx[5 * 64] = __float_as_uint(rd.z);
                    cx[6 * 64] = __float_as_uint(P.x); cx[7 * 64] = __float_as_uint(P.y); cx[8 * 64] = __float_as_uint(P.z);
                    cx[9 * 64] = __float_as_uint(Q.x); cx[10 * 64] = __float_as_uint(Q.y); cx[11 * 64] = __float_as_uint(Q.z);
                    cx[12 * 64] = rng.s0; cx[13 * 64] = rng.s1; cx[14 * 64] = rng.s2; cx[15 * 64] = rng.s3;
                    cx[16 * 64] = depth;
                    cx[17 * 64] = __float_as_uint(v_dmed);
                    parked_mask |= 1u << c;
                    need_path = true;
                    phase = PH_NEW;
                }
            }
            // ---- B, L, C: event, next-event estimation, continue or end
            if (alive && !need_path && phase == PH_HAVEP) {
                SECTK(3);
                did_work = true;
                const bool hit = q_code != CODE_MISS;
                const bool ev_medium = v_dmed < (hit ? q_t : 400.f);
                phase = PH_NEW;
                if (!ev_medium && !hit) {
                    SECTK(4);
                    acc_add(fma3(Q, env_color(sc, rd), P));
                    need_path = true;
                } else {
                    V x, n = mk(0, 1, 0), mcol = mk(0, 0, 0), E;
                    Mat mat = Mat{mk(0, 0, 0), 0.f, 0u, 0.f, 0.f};
                    stage_event<COUNT>(a, tab, ro, rd, depth, ev_medium, v_dmed, q_t, q_code, 0u, x, n, mcol, mat, E);
                    for (uint32_t li = 0; li < sc.n_lights; li++) {
                        const Light L = uload(&sc.lights[li]);
                        if (L.kind == L_AMBIENT) {
                            E = fma3(xyz(L.color), ev_medium ? mcol : mat_color(mat), E);
                        } else if (L.kind == L_OBJECT) {
                            V I, wi;
                            float dist;
                            SECTK(7);
                            illuminate_object<false>(sc, L, x, rng, I, wi, dist, tab);
                            if (L.twin_object >= 0) {
                                SECTK(8);
                                const float tm = ray_tmin(x);
                                float ts = dist * (1.f + 1e-3f);
                                uint32_t cs = CODE_MISS;
                                scan_prims(sc, x, wi, tm, ts, cs);
                                if (COUNT) c_rays++;
                                const bool twin = (L.twin_lo <= L.twin_hi) ? (cs >= L.twin_lo && cs <= L.twin_hi)
                                                                           : (cs != CODE_MISS && code_object(sc, cs, 0u) == uint32_t(L.twin_object));
                                if (cs != CODE_MISS && ts >= dist * (1.f - 1e-3f) && twin) {   // (see DETACH = 1)
                                    SECTK(9);
                                    V T;
                                    if (ev_medium) {
                                        T = (albedo_med * sc.medium_phase) * (I * mcol);
                                    } else {
                                        const V f = bsdf(mat, n, -normalize(rd), wi);
                                        T = dot(wi, n) * (f * I);
                                    }
                                    if (!mesh_roots_hit(sc, x, wi, tm, ts)) {
                                        E = E + T;
                                    } else {
                                        SECTK(22);
                                        const V cand = Q * T;
                                        const uint64_t m = __ballot(true);
                                        uint32_t base = 0u;
                                        if (mbcnt64(m) == 0u) base = atomicAdd(const_cast<uint32_t*>(wq), uint32_t(__popcll(m)));
                                        base = __builtin_amdgcn_readfirstlane(base);
                                        uint32_t* const e = ring_base + ((base + mbcnt64(m)) % kRingEntries) * kRingDwords;
                                        reinterpret_cast<f4v*>(e)[0] = f4v{x.x, x.y, x.z, wi.x};
                                        reinterpret_cast<f4v*>(e)[1] = f4v{wi.y, wi.z, ts, cand.x};
                                        reinterpret_cast<u4v*>(e)[2] = u4v{__float_as_uint(cand.y), __float_as_uint(cand.z), threadIdx.x & 63u, 0u};
                                        ls_pend[threadIdx.x] = ls_pend[threadIdx.x] + 1u;
                                    }
                                }
                            }
                        }
                    }
                    V wi = mk(0, 0, 1), k = mk(0, 0, 0);
                    const bool cont = stage_bounce<MEDIUM, COUNT>(a, albedo_med, rd, depth, ev_medium, n, mcol, mat, rng, wi, k);
                    SECTK(14);
                    P = fma3(Q, E, P);
                    if (cont) {
                        Q = Q * k;
                        ro = x;
                        rd = wi;
                        depth++;
                    } else {
                        acc_add(P);
                        need_path = true;
                    }
                }
            }
            // ---- the walk session: every lane takes queries from the ring.  wq[0] = entries ever written (head), wq[1] = entries
            // ever taken (tail), wq[2] = the oldest entry a lane still holds unfinished (low-water mark: entries from there on
            // must not be overwritten), wq[4 + 2 c ..]: lanes whose parked context c has been answered.
            {
                const uint32_t head = __builtin_amdgcn_readfirstlane(wq[0]);
                uint32_t tail = __builtin_amdgcn_readfirstlane(wq[1]);
                const uint32_t low = __builtin_amdgcn_readfirstlane(wq[2]);
                const bool susp = h_e != kNoEntry;
                const uint32_t n_work = (head - tail) + uint32_t(__popcll(__ballot(susp)));
                const bool worked = __ballot(did_work) != 0ull;
                did_work = false;
                // room for what one more trip can write (a primary query per lane + a shadow query per lane and twin light)
                const bool tight = head - low > kRingEntries - 64u * (1u + a.n_twin_lights);
                if (n_work == 0u) {
                    // nothing to walk: if nothing else moved either, some bookkeeping disagrees -- let the items go rather than spin
                    if (!worked && alive && need_path && have_item) { ls_pend[threadIdx.x] = 0u; parked_mask = 0u; }
                } else if (head - tail >= a.stream_backlog || tight || !worked) {
                    if (COUNT) c_wave[0] = c_wave[1] = 0;
                    SECTK(15);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this trip's entries are in memory before any lane reads them
                    const uint32_t min_active = (tight || !worked) ? 1u : a.defer_stop;
                    const BvhNode* const nodes = sc.nodes;
                    const uint32_t root0 = uload(&sc.meshes[0]).root;
                    // the walk at hand of this lane (entry h_e): ray, interval, closest hit so far; what the answer needs
                    V wo = mk(0, 0, 0), wd = mk(0, 0, 1), inv = mk(0, 0, 0);
                    float wt = 0.f, wtmin = 0.f;
                    uint32_t wc = CODE_MISS, aux0 = 0u, aux1 = 0u, aux2 = 0u, meta = 0u;
                    uint32_t cur = kWalkDone, sp = 0u, mesh = 0u;
                    auto load_entry = [&](uint32_t idx) {
                        const uint32_t* const e = ring_base + (idx % kRingEntries) * kRingDwords;
                        const f4v e0 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(e));
                        const f4v e1 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(e) + 1);
                        const u4v e2 = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(e) + 2);
                        wo = mk(e0.x, e0.y, e0.z);
                        wd = mk(e0.w, e1.x, e1.y);
                        wt = e1.z;
                        aux0 = __float_as_uint(e1.w); aux1 = e2.x; aux2 = e2.y; meta = e2.z;
                        wc = (meta >> 12) ? aux0 : CODE_MISS;   // a primary query starts from the scan's hit
                        wtmin = ray_tmin(wo);
                        inv = mk(rcp(wd.x), rcp(wd.y), rcp(wd.z));
                    };
                    if (susp) {
                        load_entry(h_e);
                        cur = walk.cur; sp = walk.sp; mesh = walk.mesh;
                    }
                    for (;;) {
                        if (tail != head) {   // lanes without a walk take the next entries
                            const bool free_lane = cur == kWalkDone;
                            const uint64_t m = __ballot(free_lane);
                            if (m != 0ull) {
                                const uint32_t take = min(uint32_t(__popcll(m)), head - tail), rank = mbcnt64(m);
                                if (free_lane && rank < take) {
                                    h_e = tail + rank;
                                    load_entry(h_e);
                                    cur = root0; sp = 0u; mesh = 0u;
                                }
                                tail += take;
                            }
                        }
                        const uint64_t act = __ballot(cur != kWalkDone);
                        if (act == 0ull) break;
                        if (tail == head && uint32_t(__popcll(act)) < min_active) break;   // the few that are left go on in the next session
                        for (;;) {  // the descent (kWalkDone has the leaf bit set: finished lanes take no part)
                            const bool inner = !(cur & BVH_LEAF);
                            const uint32_t n_inner = uint32_t(__popcll(__ballot(inner)));
                            if (n_inner == 0u) break;
                            if (a.walk_leaf_quarters != 0u && 4u * uint32_t(__popcll(__ballot(!inner && cur != kWalkDone))) >= a.walk_leaf_quarters * n_inner) break;
                            if (COUNT) c_wave[0]++;
                            if (!inner) continue;
                            const BvhNode nd = nodes[cur];
                            if (COUNT) c_nodes++;
                            float n0, f0, n1, f1;
                            slab2(nd.lo0, nd.hi0, wo, inv, n0, f0);
                            slab2(nd.lo1, nd.hi1, wo, inv, n1, f1);
                            const bool h0 = fmaxf(n0, wtmin) <= fminf(f0, wt);
                            const bool h1 = fmaxf(n1, wtmin) <= fminf(f1, wt);
                            if (h0 && h1) {
                                const bool first0 = n0 <= n1;
                                if (sp < kStackRows) { stk[sp * stride] = first0 ? nd.e1 : nd.e0; sp++; }
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
                        if (COUNT) {
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
                                if (COUNT) c_btris++;
                                const float t = hit_tri(tr.pn, tr.A, tr.B, wo, wd, wtmin, wt);
                                if (t >= 0.f) { wt = t; wc = (K_BVHTRI << 28) | (first + i); }
                            }
                            // a shadow query is answered by its first triangle
                            if ((meta >> 12) == 0u && wc != CODE_MISS) { sp = 0u; mesh = sc.n_mesh; }
                            if (sp) {
                                sp--;
                                cur = stk[sp * stride];
                            } else {
                                cur = kWalkDone;
                            }
                        }
                        if (cur == kWalkDone && mesh + 1u < sc.n_mesh && h_e != kNoEntry) {  // the next mesh's tree
                            mesh++;
                            cur = sc.meshes[mesh].root;
                        }
                        if (cur == kWalkDone && h_e != kNoEntry) {   // answered
                            const uint32_t owner = meta & 63u;
                            if ((meta >> 12) == 0u) {   // shadow: visible iff no tree holds a triangle in its interval
                                if (COUNT) { SECTK(18); }
                                if (wc == CODE_MISS) {
                                    const uint32_t ol = (threadIdx.x & ~63u) | owner;
                                    atomicAdd(&acc64[ol], to_fixed(__uint_as_float(aux0)));
                                    atomicAdd(&acc64[256u + ol], to_fixed(__uint_as_float(aux1)));
                                    atomicAdd(&acc64[512u + ol], to_fixed(__uint_as_float(aux2)));
                                }
                                atomicSub(const_cast<uint32_t*>(ls_pend) + ((threadIdx.x & ~63u) | owner), 1u);
                            } else {   // primary: the closest hit goes to the parked path, which becomes ready
                                SECTK(16);
                                if ((wc >> 28) == K_BVHTRI) { SECTK(17); }
                                const uint32_t c = (meta >> 6) & 15u;
                                uint32_t* const cx = ctx_base + c * kCtxFields * 64u + owner;
                                cx[18 * 64] = __float_as_uint(wt);
                                cx[19 * 64] = wc;
                                // (the path is looked for at the top of the next trip, behind the fence that ends this session)
                                atomicOr(const_cast<uint32_t*>(wq) + 4u + 2u * c + (owner >> 5), 1u << (owner & 31u));
                            }
                            h_e = kNoEntry;
                        }
                    }
                    if (h_e != kNoEntry) {   // unfinished: the walk goes on in the next session from where it stands
                        walk = WalkState{cur, sp, mesh};
                        if (meta >> 12) {   // (a primary query's closest hit so far travels in its entry)
                            uint32_t* const e = ring_base + (h_e % kRingEntries) * kRingDwords;
                            e[6] = __float_as_uint(wt);
                            e[7] = wc;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every answer is in memory before any path is taken back
                    // the oldest entry still in some lane's hands (entries are taken in order: everything before it is answered)
                    uint32_t lw = h_e != kNoEntry ? tail - h_e : 0u;   // distance back from the tail
                    for (int off = 32; off; off >>= 1) lw = max(lw, uint32_t(__shfl_xor(int(lw), off)));
                    if ((threadIdx.x & 63u) == 0u) { wq[1] = tail; wq[2] = tail - lw; }
                    if (COUNT) {
                        for (int k = 0; k < 2; k++) {
                            uint32_t v = c_wave[k];
                            for (int off = 32; off; off >>= 1) v = max(v, uint32_t(__shfl_xor(int(v), off)));
                            w_tot[k] += v;
                        }
                    }
                }
            }
            if (++fuse == 0x01000000u) break;   // (no wave of a valid launch comes near 2^24 trips: the loop cannot spin for ever)
            continue;
        }
        if constexpr (DETACH == 1) {
            // ---- per-mesh-tree flavour in a medium, shadow queries detached.  A primary query that has to walk a tree
            // parks its lane as below (PH_WAITP).  A shadow query that has to walk leaves its path instead: in a medium
            // the radiance of a path is linear in every light term (no firefly clamp, src/renderer.rs:229-232), so the
            // term enters the pixel's sum as Q x term if the light is visible, whenever that becomes known, and the path
            // goes on at once.  The query (ray, interval, contribution, owner lane) waits in a queue of this wave in LDS;
            // in a walk session the lanes with a parked primary query walk their own and every other lane of the wave
            // takes a queued shadow query, so a session starts with up to 64 walks and refills lanes from the queue as
            // they finish.  The item's sum is kept in fixed point: it does not depend on when the terms arrive, so the
            // image is still bit-identical for every schedule.
            // ---- A: a new path vertex: distance sample, scan, do the trees matter?
            if (alive && !need_path && phase == PH_NEW) {
                if (COUNT) c_vertices++;
                SECTK(2);
                stage_distance<MEDIUM>(rng, inv_sigma_t, v_dmed, q_t);
                const float tmin = ray_tmin(ro);
                q_code = CODE_MISS;
                scan_prims(sc, ro, rd, tmin, q_t, q_code);
                if (COUNT) c_rays++;
                phase = mesh_roots_hit(sc, ro, rd, tmin, q_t) ? PH_WAITP : PH_HAVEP;
                if (h_e == kNoEntry) walk = walk_begin(sc);   // (else `walk` is the entry's walk; this query begins at the root when that one is through)
            }
            // ---- B, L, C: event, next-event estimation, continue or end -- in one go, nothing waits in between
            if (alive && phase == PH_HAVEP) {
                SECTK(3);
                const bool hit = q_code != CODE_MISS;
                const bool ev_medium = v_dmed < (hit ? q_t : 400.f);
                phase = PH_NEW;
                if (!ev_medium && !hit) {
                    SECTK(4);
                    acc_add(fma3(Q, env_color(sc, rd), P));
                    need_path = true;
                } else {
                    V x, n = mk(0, 1, 0), mcol = mk(0, 0, 0), E;
                    Mat mat = Mat{mk(0, 0, 0), 0.f, 0u, 0.f, 0.f};
                    stage_event<COUNT>(a, tab, ro, rd, depth, ev_medium, v_dmed, q_t, q_code, 0u, x, n, mcol, mat, E);
                    for (uint32_t li = 0; li < sc.n_lights; li++) {
                        const Light L = uload(&sc.lights[li]);
                        if (L.kind == L_AMBIENT) {
                            E = fma3(xyz(L.color), ev_medium ? mcol : mat_color(mat), E);
                        } else if (L.kind == L_OBJECT) {
                            V I, wi;
                            float dist;
                            SECTK(7);
                            illuminate_object<false>(sc, L, x, rng, I, wi, dist, tab);
                            if (L.twin_object >= 0) {
                                SECTK(8);
                                const float tm = ray_tmin(x);
                                float ts = dist * (1.f + 1e-3f);
                                uint32_t cs = CODE_MISS;
                                scan_prims(sc, x, wi, tm, ts, cs);
                                if (COUNT) c_rays++;
                                // the light's twin lies among the scanned records (the host selects this kernel only then): the
                                // light can be visible only if the scan's closest hit is the twin at the sampled distance, and
                                // then it is visible unless a tree holds a triangle in front of that hit
                                const bool twin = (L.twin_lo <= L.twin_hi) ? (cs >= L.twin_lo && cs <= L.twin_hi)
                                                                           : (cs != CODE_MISS && code_object(sc, cs, 0u) == uint32_t(L.twin_object));
                                if (cs != CODE_MISS && ts >= dist * (1.f - 1e-3f) && twin) {
                                    SECTK(9);
                                    V T;
                                    if (ev_medium) {
                                        T = (albedo_med * sc.medium_phase) * (I * mcol);
                                    } else {
                                        const V f = bsdf(mat, n, -normalize(rd), wi);
                                        T = dot(wi, n) * (f * I);
                                    }
                                    if (!mesh_roots_hit(sc, x, wi, tm, ts)) {
                                        E = E + T;
                                    } else {
                                        SECTK(21);
                                        const V cand = Q * T;
                                        const uint64_t m = __ballot(true);   // the lanes that detach a query now
                                        uint32_t fr = ~wq[0];                // free slots (every lane reads before any lane claims)
                                        for (uint32_t i = mbcnt64(m); i != 0u; i--) fr &= fr - 1u;   // ... this lane's: the rank-th of them
                                        if (fr != 0u) {
                                            const uint32_t e = uint32_t(__builtin_ctz(fr));
                                            volatile uint32_t* const q = wq + 16u + e;
                                            q[0u * kQCap] = __float_as_uint(x.x); q[1u * kQCap] = __float_as_uint(x.y); q[2u * kQCap] = __float_as_uint(x.z);
                                            q[3u * kQCap] = __float_as_uint(wi.x); q[4u * kQCap] = __float_as_uint(wi.y); q[5u * kQCap] = __float_as_uint(wi.z);
                                            q[6u * kQCap] = __float_as_uint(ts);
                                            q[7u * kQCap] = __float_as_uint(cand.x); q[8u * kQCap] = __float_as_uint(cand.y); q[9u * kQCap] = __float_as_uint(cand.z);
                                            q[10u * kQCap] = threadIdx.x;
                                            atomicOr(const_cast<uint32_t*>(wq), 1u << e);
                                            ls_pend[threadIdx.x] = ls_pend[threadIdx.x] + 1u;
                                        } else {   // every slot is taken (a session is due as soon as detach_trigger entries wait): walk here
                                            SECTK(22);
                                            if (h_e != kNoEntry) {   // this lane's stack column holds a suspended shadow walk: that one starts again later, from the root
                                                atomicAnd(const_cast<uint32_t*>(wq) + 1, ~(1u << h_e));
                                                h_e = kNoEntry;
                                            }
                                            uint32_t cw = CODE_MISS, is = 0u;   // (the scan's hit is the end of the interval)
                                            walk_meshes<COUNT, true>(sc, x, wi, tm, ts, cw, is, stk, stride, c_nodes, c_btris, AnyHit{kInf, 1u, 0u}, kStackRows);
                                            if (cw == CODE_MISS) acc_add(cand);
                                        }
                                    }
                                }
                            }
                        }
                    }
                    V wi = mk(0, 0, 1), k = mk(0, 0, 0);
                    const bool cont = stage_bounce<MEDIUM, COUNT>(a, albedo_med, rd, depth, ev_medium, n, mcol, mat, rng, wi, k);
                    SECTK(14);
                    P = fma3(Q, E, P);
                    if (cont) {
                        Q = Q * k;
                        ro = x;
                        rd = wi;
                        depth++;
                    } else {
                        acc_add(P);
                        need_path = true;
                    }
                }
            }
            // ---- the walks: parked primary queries + the queue.  A queue entry is a slot (wq[0]: slots in use, wq[1]: slots some
            // lane is walking or holds suspended); the walk of a shadow query can be left unfinished like a parked primary walk:
            // the lane keeps the entry's index (h_e), the position in `walk` and the stack in its LDS column, the ray stays in
            // the slot.  A lane's column holds at most one unfinished walk, its own or an entry's; its own parked query starts
            // only after the entry's walk is through.
            {
                const bool waitp = alive && phase == PH_WAITP;
                const bool susp = h_e != kNoEntry;
                const uint32_t used = __builtin_amdgcn_readfirstlane(wq[0]), claimed = __builtin_amdgcn_readfirstlane(wq[1]);
                uint32_t avail = used & ~claimed;   // entries nobody has taken yet (wave-uniform)
                const uint32_t n_new = uint32_t(__popc(avail));
                const uint32_t n_work = uint32_t(__popcll(__ballot(waitp || susp))) + n_new;
                const bool stalled = alive && need_path && item_done && have_item && ls_pend[threadIdx.x] != 0u;
                // (a lane whose item is finished can do nothing either: it waits for answers -- stalled -- or for the wave's next
                // item hand-out, which in turn waits for the stalled lanes when fewer than pull_batch lanes want an item)
                const bool idle = __ballot(alive && !waitp && !(need_path && item_done)) == 0ull;   // nothing else this wave could do
                if (n_work == 0u) {
                    // nothing to walk.  (A lane can only be stalled while the queue holds a query of its item; should the
                    // bookkeeping ever disagree, the lane writes its item out rather than spin.)
                    if (idle && stalled) ls_pend[threadIdx.x] = 0u;
                } else if (n_work >= a.defer_lanes || n_new >= a.detach_trigger || idle) {
                    if (COUNT) c_wave[0] = c_wave[1] = 0;
                    SECTK(15);
                    const uint32_t min_active = idle ? 1u : a.defer_stop;
                    const BvhNode* const nodes = sc.nodes;
                    const uint32_t root0 = uload(&sc.meshes[0]).root;
                    // the walk at hand of this lane: 0 none, 1 its own parked primary query, 2 the entry h_e
                    uint32_t mode = susp ? 2u : (waitp ? 1u : 0u);
                    V wo = ro, wd = rd;
                    float wt = q_t;
                    uint32_t wc = q_code;
                    uint32_t cur = mode ? walk.cur : kWalkDone, sp = mode ? walk.sp : 0u, mesh = mode ? walk.mesh : 0u;
                    if (susp) {
                        const volatile uint32_t* const q = wq + 16u + h_e;
                        wo = mk(__uint_as_float(q[0u * kQCap]), __uint_as_float(q[1u * kQCap]), __uint_as_float(q[2u * kQCap]));
                        wd = mk(__uint_as_float(q[3u * kQCap]), __uint_as_float(q[4u * kQCap]), __uint_as_float(q[5u * kQCap]));
                        wt = __uint_as_float(q[6u * kQCap]);
                        wc = CODE_MISS;
                    }
                    float wtmin = ray_tmin(wo);
                    V inv = mk(rcp(wd.x), rcp(wd.y), rcp(wd.z));
                    for (;;) {
                        if (avail != 0u) {   // lanes without a walk take entries
                            const bool free_lane = cur == kWalkDone;
                            const uint64_t m = __ballot(free_lane);
                            if (m != 0ull) {
                                const uint32_t take = min(uint32_t(__popcll(m)), uint32_t(__popc(avail))), rank = mbcnt64(m);
                                if (free_lane && rank < take) {
                                    uint32_t f = avail;
                                    for (uint32_t i = rank; i != 0u; i--) f &= f - 1u;
                                    h_e = uint32_t(__builtin_ctz(f));
                                    atomicOr(const_cast<uint32_t*>(wq) + 1, 1u << h_e);
                                    const volatile uint32_t* const q = wq + 16u + h_e;
                                    wo = mk(__uint_as_float(q[0u * kQCap]), __uint_as_float(q[1u * kQCap]), __uint_as_float(q[2u * kQCap]));
                                    wd = mk(__uint_as_float(q[3u * kQCap]), __uint_as_float(q[4u * kQCap]), __uint_as_float(q[5u * kQCap]));
                                    wt = __uint_as_float(q[6u * kQCap]);
                                    wtmin = ray_tmin(wo);
                                    wc = CODE_MISS;
                                    inv = mk(rcp(wd.x), rcp(wd.y), rcp(wd.z));
                                    cur = root0; sp = 0u; mesh = 0u;
                                    mode = 2u;
                                }
                                for (uint32_t i = take; i != 0u; i--) avail &= avail - 1u;
                            }
                        }
                        const uint64_t act = __ballot(cur != kWalkDone);
                        if (act == 0ull) break;
                        if (uint32_t(__popcll(act)) < min_active) break;   // too few are left to walk for: they go on in the next session
                        for (;;) {  // the descent (kWalkDone has the leaf bit set: finished lanes take no part)
                            const bool inner = !(cur & BVH_LEAF);
                            const uint32_t n_inner = uint32_t(__popcll(__ballot(inner)));
                            if (n_inner == 0u) break;
                            if (a.walk_leaf_quarters != 0u && 4u * uint32_t(__popcll(__ballot(!inner && cur != kWalkDone))) >= a.walk_leaf_quarters * n_inner) break;
                            if (COUNT) c_wave[0]++;
                            if (!inner) continue;
                            const BvhNode nd = nodes[cur];
                            if (COUNT) { c_nodes++; if (mode == 1u) c_nodes_primary++; }
                            float n0, f0, n1, f1;
                            slab2(nd.lo0, nd.hi0, wo, inv, n0, f0);
                            slab2(nd.lo1, nd.hi1, wo, inv, n1, f1);
                            const bool h0 = fmaxf(n0, wtmin) <= fminf(f0, wt);
                            const bool h1 = fmaxf(n1, wtmin) <= fminf(f1, wt);
                            if (h0 && h1) {
                                const bool first0 = n0 <= n1;
                                if (sp < kStackRows) { stk[sp * stride] = first0 ? nd.e1 : nd.e0; sp++; }
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
                        if (COUNT) {
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
                                if (COUNT) c_btris++;
                                const float t = hit_tri(tr.pn, tr.A, tr.B, wo, wd, wtmin, wt);
                                if (t >= 0.f) { wt = t; wc = (K_BVHTRI << 28) | (first + i); }
                            }
                            // a shadow query is answered by its first triangle
                            if (mode == 2u && wc != CODE_MISS) { sp = 0u; mesh = sc.n_mesh; }
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
                        if (cur == kWalkDone) {
                            if (mode == 2u) {   // a shadow query is answered: visible iff no tree holds a triangle in its interval
                                const volatile uint32_t* const q = wq + 16u + h_e;
                                const uint32_t owner = q[10u * kQCap];
                                if (COUNT) { SECTK(18); }
                                if (wc == CODE_MISS) {
                                    atomicAdd(&acc64[owner], to_fixed(__uint_as_float(q[7u * kQCap])));
                                    atomicAdd(&acc64[256u + owner], to_fixed(__uint_as_float(q[8u * kQCap])));
                                    atomicAdd(&acc64[512u + owner], to_fixed(__uint_as_float(q[9u * kQCap])));
                                }
                                atomicSub(const_cast<uint32_t*>(ls_pend) + owner, 1u);
                                atomicAnd(const_cast<uint32_t*>(wq), ~(1u << h_e));
                                atomicAnd(const_cast<uint32_t*>(wq) + 1, ~(1u << h_e));
                                h_e = kNoEntry;
                                mode = 0u;
                                if (waitp && phase == PH_WAITP) {   // now this lane's own parked query (not begun: its column was taken)
                                    wo = ro; wd = rd; wt = q_t; wc = q_code;
                                    wtmin = ray_tmin(wo);
                                    inv = mk(rcp(wd.x), rcp(wd.y), rcp(wd.z));
                                    cur = root0; sp = 0u; mesh = 0u;
                                    mode = 1u;
                                }
                            } else if (mode == 1u) {
                                SECTK(16);
                                if ((wc >> 28) == K_BVHTRI) { SECTK(17); }
                                mode = 0u;
                                phase = PH_HAVEP;
                                q_t = wt;
                                q_code = wc;
                            }
                        }
                    }
                    if (mode != 0u) walk = WalkState{cur, sp, mesh};   // an unfinished walk goes on in the next session
                    if (mode == 1u) { q_t = wt; q_code = wc; }
                    if (COUNT) {
                        for (int k = 0; k < 2; k++) {
                            uint32_t v = c_wave[k];
                            for (int off = 32; off; off >>= 1) v = max(v, uint32_t(__shfl_xor(int(v), off)));
                            w_tot[k] += v;
                        }
                    }
                }
            }
            if (++fuse == 0x01000000u) break;   // (no wave of a valid launch comes near 2^24 trips: the loop cannot spin for ever)
            continue;
        }
        if constexpr (BVH == 1 || BVH == 3) {
            // ---- A: a new path vertex: distance sample, scan (BVH = 3: walk of the scene tree), do the mesh trees matter?
            // (a lane without a path waits for the wave's next item hand-out)
            if (alive && !need_path && phase == PH_NEW) {
                if (COUNT) c_vertices++;
                SECTK(2);
                stage_distance<MEDIUM>(rng, inv_sigma_t, v_dmed, q_t);
                const float tmin = ray_tmin(ro);
                q_code = CODE_MISS;
                q_inst = 0;
                scan_or_tree<BVH, COUNT>(sc, ro, rd, tmin, q_t, q_code, q_inst, stk, stride, c_nodes, c_btris);
                if (COUNT) c_rays++;
                phase = mesh_roots_hit(sc, ro, rd, tmin, q_t) ? PH_WAITP : PH_HAVEP;
                walk = walk_begin(sc);
            }
            // ---- B: the event (src/renderer.rs:197-243, 288-299)
            if (alive && phase == PH_HAVEP) {
                SECTK(3);
                const bool hit = q_code != CODE_MISS;
                v_medium = MEDIUM && (v_dmed < (hit ? q_t : 400.f));
                if (!v_medium && !hit) {
                    SECTK(4);
                    acc_add(vmin(fma3(Q, env_color(sc, rd), P), Rc));
                    need_path = true;
                    phase = PH_NEW;
                } else {
                    stage_event<COUNT>(a, tab, ro, rd, depth, v_medium, v_dmed, q_t, q_code, q_inst, v_x, v_n, v_mcol, v_mat, v_E);
                    v_li = 0;
                    phase = PH_LIGHT;
                }
            }
            // ---- L: next-event estimation, one light per iteration; a lane joins at the light it stands at
            for (uint32_t l = 0; l < sc.n_lights; l++) {
                const bool pre = alive && phase == PH_LIGHT && v_li == l;
                if (!__any(pre || (alive && phase == PH_HAVES && v_li == l))) continue;
                const Light L = uload(&sc.lights[l]);
                if (pre) {
                    bool next = true;
                    if (L.kind == L_AMBIENT) {
                        v_E = fma3(xyz(L.color), v_medium ? v_mcol : mat_color(v_mat), v_E);
                    } else if (L.kind == L_OBJECT) {
                        SECTK(7);
                        illuminate_object<GROUPS>(sc, L, v_x, rng, v_I, v_wi, v_dist, tab);
                        if (L.twin_object >= 0) {  // the shadow query (see the undeferred body)
                            SECTK(8);
                            const float tm = ray_tmin(v_x);
                            q_t = v_dist * (1.f + 1e-3f);
                            q_code = CODE_MISS;
                            q_inst = 0;
                            scan_or_tree<BVH, COUNT>(sc, v_x, v_wi, tm, q_t, q_code, q_inst, stk, stride, c_nodes, c_btris);
                            if (COUNT) c_rays++;
                            const AnyHit any{L.twin_lo <= L.twin_hi ? v_dist * (1.f - 1e-3f) : -kInf, L.twin_lo, L.twin_hi};
                            const bool blocked = q_code != CODE_MISS && any.blocks(q_t, q_code);
                            phase = (!blocked && mesh_roots_hit(sc, v_x, v_wi, tm, q_t)) ? PH_WAITS : PH_HAVES;
                            walk = walk_begin(sc);
                            next = false;
                        }
                    }
                    if (next) v_li = l + 1u;
                }
                if (alive && phase == PH_HAVES && v_li == l) {
                    SECTK(9);
                    stage_light_term(sc, L, albedo_med, rd, v_medium, q_t, q_code, q_inst, v_dist, v_I, v_wi, v_n, v_mcol, v_mat, v_E);
                    v_li = l + 1u;
                    phase = PH_LIGHT;
                }
            }
            // ---- C: continue or end the path
            if (alive && phase == PH_LIGHT && v_li >= sc.n_lights) {
                V wi = mk(0, 0, 1), k = mk(0, 0, 0);
                const bool cont = stage_bounce<MEDIUM, COUNT>(a, albedo_med, rd, depth, v_medium, v_n, v_mcol, v_mat, rng, wi, k);
                SECTK(14);
                P = fma3(Q, v_E, P);
                if (cont) {
                    if (!MEDIUM) Rc = vmin(Rc, fma3(100.f, Q, P));
                    Q = Q * k;
                    ro = v_x;
                    rd = wi;
                    depth++;
                } else {
                    acc_add(vmin(P, Rc));
                    need_path = true;
                }
                phase = PH_NEW;
            }
            // ---- the parked walks: start when enough lanes wait, stop when most of them are through
            const bool waiting = alive && phase >= PH_WAITP;
            const uint64_t wm = __ballot(waiting);
            const bool idle = __ballot(alive && !waiting && !(need_path && item_done)) == 0;  // nothing else this wave could do (lanes between items wait for the hand-out)
            if (wm != 0 && (uint32_t(__popcll(wm)) >= a.defer_lanes || idle)) {
                if (COUNT) c_wave[0] = c_wave[1] = 0;
                if (waiting) {
                    SECTK(15);
                    const bool shadow = phase == PH_WAITS;
                    const V qo = shadow ? v_x : ro, qd = shadow ? v_wi : rd;
                    AnyHit any{-kInf, 1u, 0u};  // a primary query: nothing blocks
                    if (shadow) {
                        const uint32_t lo = sc.lights[v_li].twin_lo, hi = sc.lights[v_li].twin_hi;
                        any = AnyHit{lo <= hi ? v_dist * (1.f - 1e-3f) : -kInf, lo, hi};
                    }
                    walk_meshes_resumable<COUNT>(sc, qo, qd, ray_tmin(qo), q_t, q_code, stk, stride, walk,
                                                 idle ? 1u : a.defer_stop, any, c_nodes, c_btris, kStackRows, a.walk_leaf_quarters, c_wave);
                    if (walk.cur == kWalkDone) {
                        phase = shadow ? PH_HAVES : PH_HAVEP;
                        SECTK(16);
                        if ((q_code >> 28) == K_BVHTRI) { SECTK(17); }
                        if (shadow) { SECTK(18); }
                    }
                }
                if (COUNT) {   // every lane that walked counted the same steps
                    for (int k = 0; k < 2; k++) {
                        uint32_t v = c_wave[k];
                        for (int off = 32; off; off >>= 1) v = max(v, uint32_t(__shfl_xor(int(v), off)));
                        w_tot[k] += v;
                    }
                }
            }
            continue;
        }
        if (!alive || need_path) continue;   // (a lane without a path waits for the wave's next item hand-out)

        // ---- one path vertex (one trace_ray invocation, src/renderer.rs:187-322)
        if (COUNT) c_vertices++;
        SECTK(2);
        float dmed, t;
        stage_distance<MEDIUM>(rng, inv_sigma_t, dmed, t);
        const float tmin = ray_tmin(ro);
        uint32_t code = CODE_MISS, inst = 0;
        closest_hit<BVH, COUNT>(sc, ro, rd, tmin, t, code, inst, stk, stride, c_nodes, c_btris);
        if (COUNT) c_rays++;
        SECTK(3);
        const bool hit = code != CODE_MISS;

        const bool ev_medium = MEDIUM && (dmed < (hit ? t : 400.f));  // src/renderer.rs:197-243
        const bool ev_surface = !ev_medium && hit;
        if (!ev_medium && !ev_surface) {  // miss: environment (src/renderer.rs:198-206, 288)
            SECTK(4);
            if constexpr (BVH == 0) {
                const V v = fma3(ldv(S_Q, Q), env_color(sc, rd), ldv(S_P, P));
                acc_add(MEDIUM ? v : vmin(v, ldv(S_RC, Rc)));   // in a medium Rc stays +inf (no firefly clamp, src/renderer.rs:229-232)
            } else {
                acc_add(vmin(fma3(Q, env_color(sc, rd), P), Rc));
            }
            need_path = true;
            continue;
        }

        V x, n = mk(0, 1, 0), mcol = mk(0, 0, 0), E;
        Mat mat = Mat{mk(0, 0, 0), 0.f, 0u, 0.f, 0.f};
        stage_event<COUNT>(a, tab, ro, rd, depth, ev_medium, dmed, t, code, inst, x, n, mcol, mat, E);

        // ---- next-event estimation: sample_lights / sample_lights_for_media
        //      (src/renderer.rs:362-409 / 325-359); lights in scene order fix the draw order.
        for (uint32_t li = 0; li < sc.n_lights; li++) {
            const Light L = uload(&sc.lights[li]);
            if (L.kind == L_AMBIENT) {
                E = fma3(xyz(L.color), ev_medium ? mcol : mat_color(mat), E);
            } else if (L.kind == L_OBJECT) {
                V I, wi;
                float dist;
                SECTK(7);
                illuminate_object<GROUPS>(sc, L, x, rng, I, wi, dist, tab);
                if (L.twin_object >= 0) {
                    SECTK(8);
                    float ts = dist * (1.f + 1e-3f);
                    uint32_t cs = CODE_MISS, is = 0;
                    // tree-walking scenes: any hit in front of the light on something other than its twin settles the test
                    const bool range = L.twin_lo <= L.twin_hi;
                    closest_hit<BVH, COUNT, BVH != 0>(sc, x, wi, ray_tmin(x), ts, cs, is, stk, stride, c_nodes, c_btris,
                                                     AnyHit{range ? dist * (1.f - 1e-3f) : -kInf, L.twin_lo, L.twin_hi});
                    if (COUNT) c_rays++;
                    SECTK(9);
                    stage_light_term(sc, L, albedo_med, rd, ev_medium, ts, cs, is, dist, I, wi, n, mcol, mat, E);
                }
            }
            // Point / Directional lights can never satisfy the reference's test (dist is the
            // light position / +inf, src/light.rs:26-33): no draws, no contribution.
        }

        // ---- continue or end the path
        V wi = mk(0, 0, 1), k = mk(0, 0, 0);
        const bool cont = stage_bounce<MEDIUM, COUNT>(a, albedo_med, rd, depth, ev_medium, n, mcol, mat, rng, wi, k);
        SECTK(14);
        if constexpr (BVH == 0) {
            const V q = ldv(S_Q, Q);
            const V pn = fma3(q, E, ldv(S_P, P));
            if (cont) {
                stv(S_P, P, pn);
                if (!MEDIUM) stv(S_RC, Rc, vmin(ldv(S_RC, Rc), fma3(100.f, q, pn)));  // FIREFLY_CLAMP, src/renderer.rs:311-313
                stv(S_Q, Q, q * k);
                ro = x;
                rd = wi;
                depth++;
            } else {
                acc_add(MEDIUM ? pn : vmin(pn, ldv(S_RC, Rc)));
                need_path = true;
            }
        } else {
            P = fma3(Q, E, P);
            if (cont) {
                if (!MEDIUM) Rc = vmin(Rc, fma3(100.f, Q, P));  // FIREFLY_CLAMP, src/renderer.rs:311-313
                Q = Q * k;
                ro = x;
                rd = wi;
                depth++;
            } else {
                acc_add(vmin(P, Rc));
                need_path = true;
            }
        }
    }

    if (COUNT) {
#ifdef RPT_SECT_CLOCKS
        if ((threadIdx.x & 63u) < 56u) atomicAdd(&a.counters[8 + (threadIdx.x & 63u)], sect_lds_[threadIdx.x >> 6][threadIdx.x & 63u]);
#endif
        atomicAdd(&a.counters[0], (unsigned long long)c_samples);
        atomicAdd(&a.counters[1], (unsigned long long)c_rays);
        atomicAdd(&a.counters[2], (unsigned long long)c_vertices);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&a.counters[3], (unsigned long long)c_trips);
        if ((threadIdx.x & 63u) == 0) { atomicAdd(&a.counters[46], (unsigned long long)w_tot[0]); atomicAdd(&a.counters[47], (unsigned long long)w_tot[1]); }
        atomicAdd(&a.counters[5], (unsigned long long)c_nodes);
        if (c_nodes_primary) atomicAdd(&a.counters[61], (unsigned long long)c_nodes_primary);
        atomicAdd(&a.counters[6], (unsigned long long)c_btris);
    }
}

__global__ __launch_bounds__(256) void resolve_kernel(const RenderArgs a, double scale, double* __restrict__ out) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= a.n_owned) return;
    uint32_t x, y;
    item_pixel(a, p, x, y);
    if (x >= a.width || y >= a.height) return;
    double r = 0.0, g = 0.0, b = 0.0;
    const float4* slab = reinterpret_cast<const float4*>(a.slab);
    for (uint32_t c = 0; c < a.n_chunks; c++) {
        float4 v = slab[size_t(c) * a.n_owned + p];
        r += double(v.x);
        g += double(v.y);
        b += double(v.z);
        if (a.slab2) {
            const float4 w = reinterpret_cast<const float4*>(a.slab2)[size_t(c) * a.n_owned + p];
            r += double(w.x);
            g += double(w.y);
            b += double(w.z);
        }
    }
    size_t o = (size_t(y) * a.width + x) * 3;
    double inv = scale / double(a.iterations);
    out[o] = r * inv;
    out[o + 1] = g * inv;
    out[o + 2] = b * inv;
}

// ------------------------------------------------------------------ Buffer (src/buffer.rs)
// add_samples (:32-40): one mean per pixel per batch.  The per-pixel Vec<Color> is kept as its
// running sum (in push order, as `iter().sum()` adds it) and the sum of squared magnitudes.
__global__ __launch_bounds__(256) void buffer_add_kernel(uint32_t n_pixels, const double* __restrict__ batch,
                                                         double* __restrict__ sum, double* __restrict__ sumsq) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pixels) return;
    double r = batch[3 * size_t(p)], g = batch[3 * size_t(p) + 1], b = batch[3 * size_t(p) + 2];
    sum[3 * size_t(p)] += r;
    sum[3 * size_t(p) + 1] += g;
    sum[3 * size_t(p) + 2] += b;
    sumsq[p] += r * r + g * g + b * b;
}
// image() (:42-57) = get_filtered_color (:76-97, window summed x-outer / y-inner over the clipped
// (2r+1)^2 pixels, divided by the number of samples in it) then color_bytes (src/color.rs:18-24).
__global__ __launch_bounds__(256) void buffer_image_kernel(uint32_t width, uint32_t height, uint32_t radius, uint32_t n_batches,
                                                           const double* __restrict__ sum, uint8_t* __restrict__ out) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= width * height) return;
    uint32_t x = p % width, y = p / width;
    double r = 0.0, g = 0.0, b = 0.0;
    uint64_t count = 0;
    uint32_t x0 = x >= radius ? x - radius : 0u, y0 = y >= radius ? y - radius : 0u;
    for (uint32_t i = x0; i <= x + radius; i++)
        for (uint32_t j = y0; j <= y + radius; j++)
            if (i < width && j < height) {
                size_t q = size_t(j) * width + i;
                r += sum[3 * q];
                g += sum[3 * q + 1];
                b += sum[3 * q + 2];
                count += n_batches;
            }
    double c[3] = {r / double(count), g / double(count), b / double(count)};
    for (int k = 0; k < 3; k++) {
        double v = fmin(fmax(c[k], 0.0), 1.0);               // NaN clamps to 0 like f64::clamp + `as u8`
        out[3 * size_t(p) + k] = uint8_t(pow(v, 1.0 / 2.2) * 255.0);
    }
}
// variance() (:60-74), per pixel: sum |s - mean|^2 / (n - 1) from the running sums.
__global__ __launch_bounds__(256) void buffer_variance_kernel(uint32_t n_pixels, uint32_t n_batches, const double* __restrict__ sum,
                                                              const double* __restrict__ sumsq, double* __restrict__ out) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pixels) return;
    double n = double(n_batches);
    double mr = sum[3 * size_t(p)] / n, mg = sum[3 * size_t(p) + 1] / n, mb = sum[3 * size_t(p) + 2] / n;
    double ss = sumsq[p] - n * (mr * mr + mg * mg + mb * mb);
    out[p] = fmax(ss, 0.0) / (n - 1.0);
}

// ------------------------------------------------------------------ frame exchange (rpt_gather_frame_device)
// Element e of the packed block = channel e % 3 of pixel (e / 3) % 1024 (row-major in the tile) of tile e / 3072 of
// the list: consecutive lanes touch consecutive doubles on both sides (96 per tile row).
template <bool PACK>
__global__ __launch_bounds__(256) void frame_tiles_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                          const uint32_t* __restrict__ tiles, uint32_t n_tiles, uint32_t tiles_x,
                                                          uint32_t width, uint32_t height) {
    const uint64_t e = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (e >= uint64_t(n_tiles) * 3072u) return;
    const uint32_t t = uint32_t(e / 3072u), r = uint32_t(e % 3072u), p = r / 3u, c = r - 3u * p;
    const uint32_t tile = tiles[t], ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t x = tx * 32u + (p & 31u), y = ty * 32u + (p >> 5);
    const bool inside = x < width && y < height;
    const size_t f = (size_t(y) * width + x) * 3u + c;
    if (PACK) dst[e] = inside ? src[f] : 0.0;
    else if (inside) dst[f] = src[e];
}

// (The SceneView MUST stay this kernel's first parameter: the device functions read the view from the kernel-argument segment at
// offset 0 -- kernarg_scene() in device_core.h -- whatever reference they are handed.)
template <int BVH>
__global__ __launch_bounds__(256) void intersect_kernel(const SceneView sc, uint64_t n, const float* __restrict__ o,
                                                        const float* __restrict__ d, float* __restrict__ t_out,
                                                        int32_t* __restrict__ obj_out, float* __restrict__ n_out) {
    extern __shared__ uint32_t dyn_lds[];
    uint32_t* stk = BVH ? (dyn_lds + threadIdx.x) : nullptr;
    uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= n) return;
    V ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    float tmin = ray_tmin(ro), t = kInf;
    uint32_t code = CODE_MISS, inst = 0, c0 = 0, c1 = 0;
    closest_hit<BVH, false>(sc, ro, rd, tmin, t, code, inst, stk, 256, c0, c1);
    V nn = mk(0, 0, 0);
    uint32_t obj = 0xFFFFFFFFu;
    if (code != CODE_MISS) finalize_hit(sc, ro, rd, tmin, t, code, inst, nn, obj);
    t_out[i] = t;
    obj_out[i] = int32_t(obj);
    if (n_out) {
        n_out[3 * i] = nn.x;
        n_out[3 * i + 1] = nn.y;
        n_out[3 * i + 2] = nn.z;
    }
}

__global__ void debug_rng_kernel(uint64_t seed_mixed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Rng r;
    r.seed(seed_mixed, pixel, sample);
    for (uint32_t i = 0; i < n; i++) out[i] = r.next();
}
RPT_DEV Mat mat_from(const Material& m) {
    return Mat{xyz(m.albedo_emit), m.albedo_emit.w, __float_as_uint(m.params.x), m.params.y, m.params.z};
}
__global__ void debug_sample_f_kernel(const Material m, uint64_t n, const float* nrm, const float* wo, uint64_t seed_mixed,
                                      float* wi, float* pdf, int32_t* some) {
    uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r;
    r.seed(seed_mixed, uint32_t(i), 0);
    V w = mk(0, 0, 0);
    float p = 0.f;
    bool ok = sample_f(mat_from(m), mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]),
                       mk(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), r, w, p);
    wi[3 * i] = w.x; wi[3 * i + 1] = w.y; wi[3 * i + 2] = w.z;
    pdf[i] = p;
    some[i] = ok ? 1 : 0;
}
__global__ void debug_bsdf_kernel(const Material m, uint64_t n, const float* nrm, const float* wo, const float* wi,
                                  float* out) {
    uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V f = bsdf(mat_from(m), mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), mk(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]),
               mk(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
    out[3 * i] = f.x; out[3 * i + 1] = f.y; out[3 * i + 2] = f.z;
}
__global__ void debug_camera_kernel(const CameraG cam, uint32_t w, uint32_t h, uint64_t seed_mixed, uint32_t sample,
                                    float* o, float* d) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    uint32_t x = i % w, y = i / w;
    float inv_dim = 1.f / float(max(w, h));
    float xn = (float(2u * x + 1u) - float(w)) * inv_dim;
    float yn = (float(2u * (h - y) - 1u) - float(h)) * inv_dim;
    Rng r;
    r.seed(seed_mixed, i, sample);
    float dx = r.range(-inv_dim, inv_dim), dy = r.range(-inv_dim, inv_dim);
    V ro, rd;
    cast_ray(cam, xn + dx, yn + dy, r, ro, rd);
    o[3 * i] = ro.x; o[3 * i + 1] = ro.y; o[3 * i + 2] = ro.z;
    d[3 * i] = rd.x; d[3 * i + 1] = rd.y; d[3 * i + 2] = rd.z;
}

// ------------------------------------------------------------------ launchers
static constexpr size_t kStackBytes = 32u * 256u * sizeof(uint32_t);
static constexpr size_t kTabBytes = (2u * 32u + 6u * 8u) * 16u;   // LdsTables: 32 materials + 8 light triangles
static constexpr size_t kStateBytes = 18u * 256u * sizeof(uint32_t) + kTabBytes;      // LDS-resident lane state of the scan instantiations
static constexpr size_t kStateBytesBvh = 5u * 256u * sizeof(uint32_t) + kTabBytes;   // ... of the tree-walking ones (behind the stack)
static constexpr size_t kMeshTreeBytes = (21u + 5u) * 256u * sizeof(uint32_t) + kTabBytes;   // per-mesh-tree kernels: 21 stack rows + lane state
static constexpr size_t kDetachBytes = kDetachDwords * sizeof(uint32_t);                      // ... with detached shadow queries
static_assert(kTabBytes == kTabDwords * 4u, "one table layout");

template <bool M, int B, bool C>
static hipError_t launch_render_t(const RenderArgs& a, int n_blocks, hipStream_t stream) {
    const size_t lds = B == 1 ? kMeshTreeBytes : B ? kStackBytes + kStateBytesBvh : kStateBytes;
    if constexpr (M && B == 1) {
#ifdef RPT_EXPERIMENTS   // streamed walks: a measured-slower prototype (297 against 237 ms), not in the default build
        if (a.detach == 2 && !a.sc.n_lparts) {
            hipLaunchKernelGGL((render_kernel<true, 1, C, false, 2>), dim3(n_blocks), dim3(256), kDetachBytes, stream, a);
            return hipGetLastError();
        }
#endif
        if (a.detach && !a.sc.n_lparts) {
            hipLaunchKernelGGL((render_kernel<true, 1, C, false, 1>), dim3(n_blocks), dim3(256), kDetachBytes, stream, a);
            return hipGetLastError();
        }
    }
    // group lights have no counters build: the section counters stay zero for such scenes
    if (a.sc.n_lparts) hipLaunchKernelGGL((render_kernel<M, B, false, true>), dim3(n_blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((render_kernel<M, B, C>), dim3(n_blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}
template <bool M, bool C>
static hipError_t launch_render_b(const RenderArgs& a, int bvh, int n_blocks, hipStream_t stream) {
    if (bvh == 3) return launch_render_t<M, 3, C>(a, n_blocks, stream);
    if (bvh == 2) return launch_render_t<M, 2, C>(a, n_blocks, stream);
    if (bvh == 1) return launch_render_t<M, 1, C>(a, n_blocks, stream);
    return launch_render_t<M, 0, C>(a, n_blocks, stream);
}
int bvh_mode(const SceneView& sc) { return sc.scene_bvh ? ((sc.mesh_deferred && sc.n_mesh) ? 3 : 2) : (sc.n_nodes ? 1 : 0); }
hipError_t launch_render(const RenderArgs& a, int n_blocks, hipStream_t stream) {
    const bool m = a.sc.has_medium != 0, c = a.counters != nullptr;
    const int b = bvh_mode(a.sc);
    if (m) return c ? launch_render_b<true, true>(a, b, n_blocks, stream) : launch_render_b<true, false>(a, b, n_blocks, stream);
    return c ? launch_render_b<false, true>(a, b, n_blocks, stream) : launch_render_b<false, false>(a, b, n_blocks, stream);
}
size_t stream_scratch_bytes_per_block() { return size_t(kWaveScratchDwords) * 4u * 4u; }
hipError_t render_occupancy(bool medium, int bvh, int* blocks_per_cu, int detach) {
#ifdef RPT_EXPERIMENTS
    if (detach == 2 && medium && bvh == 1)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)render_kernel<true, 1, false, false, 2>, 256, kDetachBytes);
#endif
    if (detach && medium && bvh == 1)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)render_kernel<true, 1, false, false, 1>, 256, kDetachBytes);
    const void* f;
    if (bvh == 3) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, medium ? (const void*)render_kernel<true, 3, false> : (const void*)render_kernel<false, 3, false>,
                                                                      256, kStackBytes + kStateBytesBvh);
    if (medium) f = bvh == 2 ? (const void*)render_kernel<true, 2, false> : bvh == 1 ? (const void*)render_kernel<true, 1, false> : (const void*)render_kernel<true, 0, false>;
    else f = bvh == 2 ? (const void*)render_kernel<false, 2, false> : bvh == 1 ? (const void*)render_kernel<false, 1, false> : (const void*)render_kernel<false, 0, false>;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, f, 256, bvh == 1 ? kMeshTreeBytes : bvh ? kStackBytes + kStateBytesBvh : kStateBytes);
}
hipError_t launch_resolve(const RenderArgs& a, double scale, double* d_out, hipStream_t stream) {
    uint32_t blocks = (a.n_owned + 255u) / 256u;
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, a, scale, d_out);
    return hipGetLastError();
}
hipError_t launch_buffer_add(uint32_t n_pixels, const double* d_batch, double* d_sum, double* d_sumsq, hipStream_t st) {
    hipLaunchKernelGGL(buffer_add_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, st, n_pixels, d_batch, d_sum, d_sumsq);
    return hipGetLastError();
}
hipError_t launch_buffer_image(uint32_t w, uint32_t h, uint32_t radius, uint32_t n_batches, const double* d_sum, uint8_t* d_out,
                               hipStream_t st) {
    hipLaunchKernelGGL(buffer_image_kernel, dim3((w * h + 255) / 256), dim3(256), 0, st, w, h, radius, n_batches, d_sum, d_out);
    return hipGetLastError();
}
hipError_t launch_buffer_variance(uint32_t n_pixels, uint32_t n_batches, const double* d_sum, const double* d_sumsq, double* d_out,
                                  hipStream_t st) {
    hipLaunchKernelGGL(buffer_variance_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, st, n_pixels, n_batches, d_sum, d_sumsq,
                       d_out);
    return hipGetLastError();
}
hipError_t launch_frame_pack(const double* d_frame, double* d_packed, const uint32_t* d_tiles, uint32_t n_tiles, uint32_t tiles_x,
                             uint32_t width, uint32_t height, hipStream_t st) {
    if (!n_tiles) return hipSuccess;
    hipLaunchKernelGGL(frame_tiles_kernel<true>, dim3(n_tiles * 12u), dim3(256), 0, st, d_frame, d_packed, d_tiles, n_tiles, tiles_x, width, height);
    return hipGetLastError();
}
hipError_t launch_frame_unpack(const double* d_packed, double* d_frame, const uint32_t* d_tiles, uint32_t n_tiles, uint32_t tiles_x,
                               uint32_t width, uint32_t height, hipStream_t st) {
    if (!n_tiles) return hipSuccess;
    hipLaunchKernelGGL(frame_tiles_kernel<false>, dim3(n_tiles * 12u), dim3(256), 0, st, d_packed, d_frame, d_tiles, n_tiles, tiles_x, width, height);
    return hipGetLastError();
}
hipError_t launch_intersect(const SceneView& sc, uint64_t n, const float* d_o, const float* d_d, float* d_t,
                            int32_t* d_obj, float* d_n, bool bvh, hipStream_t stream) {
    uint32_t blocks = uint32_t((n + 255) / 256);
    if (bvh) hipLaunchKernelGGL(intersect_kernel<2>, dim3(blocks), dim3(256), kStackBytes, stream, sc, n, d_o, d_d, d_t, d_obj, d_n);
    else hipLaunchKernelGGL(intersect_kernel<0>, dim3(blocks), dim3(256), 0, stream, sc, n, d_o, d_d, d_t, d_obj, d_n);
    return hipGetLastError();
}
hipError_t launch_debug_rng(uint64_t seed_mixed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* d_out,
                            hipStream_t stream) {
    hipLaunchKernelGGL(debug_rng_kernel, dim3(1), dim3(64), 0, stream, seed_mixed, pixel, sample, n, d_out);
    return hipGetLastError();
}
hipError_t launch_debug_sample_f(const Material& m, uint64_t n, const float* d_n, const float* d_wo,
                                 uint64_t seed_mixed, float* d_wi, float* d_pdf, int32_t* d_some, hipStream_t s) {
    hipLaunchKernelGGL(debug_sample_f_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, s, m, n, d_n, d_wo, seed_mixed,
                       d_wi, d_pdf, d_some);
    return hipGetLastError();
}
hipError_t launch_debug_bsdf(const Material& m, uint64_t n, const float* d_n, const float* d_wo, const float* d_wi,
                             float* d_out, hipStream_t s) {
    hipLaunchKernelGGL(debug_bsdf_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, s, m, n, d_n, d_wo, d_wi, d_out);
    return hipGetLastError();
}
hipError_t launch_debug_camera(const CameraG& cam, uint32_t w, uint32_t h, uint64_t seed_mixed, uint32_t sample,
                               float* d_o, float* d_d, hipStream_t s) {
    hipLaunchKernelGGL(debug_camera_kernel, dim3((w * h + 255) / 256), dim3(256), 0, s, cam, w, h, seed_mixed, sample, d_o, d_d);
    return hipGetLastError();
}

}  // namespace rptg
