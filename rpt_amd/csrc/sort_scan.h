// sort_scan.h — device-wide stable radix sort of (63-bit key, 32-bit value) pairs and a two-array exclusive prefix sum, written
// for gfx950 (wave64), for the photon-map build (src/photon.rs:193-305 is where the reference builds its maps; here: Morton
// keys -> sort -> LBVH).  Header-only (kernels are templates / inline): included by photon.hip.
//
// Sort: least-significant-digit first, 8-bit digits, eight passes over ping-pong buffers.  A pass is three launches:
//   1. rsort_hist_kernel     one block per tile of 2,048 pairs: the tile's digit histogram -> table[digit][tile]
//   2. rsort_rowscan_kernel  one block per digit: exclusive prefix of its row over the tiles, and the digit's total
//   3. rsort_scatter_kernel  one block per tile: every pair's stable rank inside the tile -- 256 consecutive pairs at a time; in a
//                            wave the lanes holding the same digit find each other with eight ballots (one per digit bit), the
//                            four waves add their per-digit counts in wave order through LDS -- then pair -> digit base +
//                            tiles before + rank.  The last pass can hand each pair to a functor instead of storing it (the
//                            photon build gathers the 48-byte records there: no separate gather launch, no second read of the
//                            permutation).
// Every pass is stable, so equal keys keep their input order: the result is THE sorted sequence of (key, input position), the
// same on every run and for every grid.  No chained scan across blocks, no spinning: every launch runs to completion on its own.
//
// Scan: two arrays of per-photon record counts -> exclusive prefix sums + 64-bit totals, three launches (tile sums, one block
// over the tile sums, tiles again).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rptg {
namespace ss {

static constexpr uint32_t kThreads = 256u, kItems = 8u, kTile = kThreads * kItems;   // 2,048 pairs per block
static constexpr uint32_t kDigits = 256u;

__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}

// ---- sort
__global__ __launch_bounds__(256) void rsort_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t n_tiles,
                                                         uint32_t* __restrict__ table) {
    __shared__ uint32_t hist[kDigits];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kTile;
#pragma unroll
    for (uint32_t k = 0; k < kItems; k++) {
        const uint32_t i = base + k * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&hist[uint32_t(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    table[size_t(threadIdx.x) * n_tiles + blockIdx.x] = hist[threadIdx.x];
}

// Exclusive scan of one value per thread over a 256-thread block; returns the thread's prefix, *total = the block's sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sums /*[4] LDS*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t up = uint32_t(__shfl_up(int(inc), off));
        if (lane >= off) inc += up;
    }
    if (lane == 63u) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t before = 0u, all = 0u;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t s = wave_sums[w];
        if (w < wave) before += s;
        all += s;
    }
    __syncthreads();   // (wave_sums may be reused by the caller's next scan)
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(256) void rsort_rowscan_kernel(uint32_t* __restrict__ table, uint32_t n_tiles, uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t wave_sums[4];
    uint32_t* const row = table + size_t(blockIdx.x) * n_tiles;
    uint32_t running = 0u;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += kThreads) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < n_tiles ? row[t] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, wave_sums, &total);
        if (t < n_tiles) row[t] = running + ex;
        running += total;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = running;
}

struct StorePair {   // the ordinary pass: the pair goes to the other buffer
    uint64_t* keys_out;
    uint32_t* vals_out;
    __device__ __forceinline__ void operator()(uint32_t dst, uint64_t key, uint32_t val) const {
        keys_out[dst] = key;
        vals_out[dst] = val;
    }
};

template <class Emit>
__global__ __launch_bounds__(256) void rsort_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                            uint32_t shift, uint32_t n_tiles, const uint32_t* __restrict__ table,
                                                            const uint32_t* __restrict__ digit_total, Emit emit) {
    __shared__ uint32_t base[kDigits];        // where this tile's pairs of a digit begin in the output
    __shared__ uint32_t counter[kDigits];     // pairs of a digit in the rounds of this tile so far
    __shared__ uint32_t wave_count[4][kDigits];
    __shared__ uint32_t wave_sums[4];
    {   // digit base = exclusive prefix of the digit totals + this digit's pairs in the tiles before
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(digit_total[threadIdx.x], wave_sums, &total);
        base[threadIdx.x] = ex + table[size_t(threadIdx.x) * n_tiles + blockIdx.x];
        counter[threadIdx.x] = 0u;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile0 = blockIdx.x * kTile;
    for (uint32_t k = 0; k < kItems; k++) {
        const uint32_t i = tile0 + k * kThreads + threadIdx.x;
        const bool has = i < n;
        const uint64_t key = has ? keys[i] : 0ull;
        const uint32_t val = has ? vals[i] : 0u;
        const uint32_t d = uint32_t(key >> shift) & 255u;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) wave_count[w][threadIdx.x] = 0u;
        __syncthreads();
        // the lanes of this wave with the same digit (and a pair)
        uint64_t same = __ballot(has);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const uint64_t m = __ballot(((d >> b) & 1u) != 0u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank_in_wave = lanes_below(same);
        if (has && rank_in_wave == 0u) wave_count[wave][d] = uint32_t(__popcll(same));
        __syncthreads();
        uint32_t dst = 0u;
        if (has) {
            uint32_t before = counter[d];
            for (uint32_t w = 0; w < wave; w++) before += wave_count[w][d];
            dst = base[d] + before + rank_in_wave;
        }
        __syncthreads();
        counter[threadIdx.x] += wave_count[0][threadIdx.x] + wave_count[1][threadIdx.x] + wave_count[2][threadIdx.x] + wave_count[3][threadIdx.x];
        if (has) emit(dst, key, val);
        (void)lane;
    }
}

inline uint32_t rsort_tiles(uint32_t n) { return (n + kTile - 1u) / kTile; }
inline size_t rsort_temp_bytes(uint32_t n) { return (size_t(kDigits) * rsort_tiles(n) + kDigits) * sizeof(uint32_t); }

// Sorts n pairs by the low 8 * passes bits of the keys, stably.  keys / vals and keys2 / vals2 are ping-pong buffers: after an even
// number of passes the result is in keys / vals.  `last`: functor that receives (position, key, value) of every pair in the last
// pass instead of the store (pass StorePair{...} for an ordinary sort).
template <class Emit>
inline hipError_t radix_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* keys2, uint32_t* vals2, uint32_t n, uint32_t passes, void* temp,
                                   Emit last, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t n_tiles = rsort_tiles(n);
    uint32_t* const table = static_cast<uint32_t*>(temp);
    uint32_t* const digit_total = table + size_t(kDigits) * n_tiles;
    for (uint32_t p = 0; p < passes; p++) {
        const uint64_t* kin = (p & 1u) ? keys2 : keys;
        const uint32_t* vin = (p & 1u) ? vals2 : vals;
        hipLaunchKernelGGL(rsort_hist_kernel, dim3(n_tiles), dim3(kThreads), 0, st, kin, n, 8u * p, n_tiles, table);
        hipLaunchKernelGGL(rsort_rowscan_kernel, dim3(kDigits), dim3(kThreads), 0, st, table, n_tiles, digit_total);
        if (p + 1u == passes)
            hipLaunchKernelGGL((rsort_scatter_kernel<Emit>), dim3(n_tiles), dim3(kThreads), 0, st, kin, vin, n, 8u * p, n_tiles, table, digit_total, last);
        else
            hipLaunchKernelGGL((rsort_scatter_kernel<StorePair>), dim3(n_tiles), dim3(kThreads), 0, st, kin, vin, n, 8u * p, n_tiles, table, digit_total,
                               StorePair{(p & 1u) ? keys : keys2, (p & 1u) ? vals : vals2});
    }
    return hipGetLastError();
}

// ---- scan of two u32 arrays: out = exclusive prefix sums, totals[0], totals[1] = the sums (64 bits)
__global__ __launch_bounds__(256) void scan2_tile_sums_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n,
                                                              uint32_t* __restrict__ sums_a, uint32_t* __restrict__ sums_b) {
    __shared__ uint32_t wave_sums[4];
    const uint32_t i0 = blockIdx.x * kTile + threadIdx.x * kItems;
    uint32_t sa = 0u, sb = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kItems; k++)
        if (i0 + k < n) { sa += a[i0 + k]; sb += b[i0 + k]; }
    uint32_t ta, tb;
    (void)block_exclusive_scan(sa, wave_sums, &ta);
    (void)block_exclusive_scan(sb, wave_sums, &tb);
    if (threadIdx.x == 0) { sums_a[blockIdx.x] = ta; sums_b[blockIdx.x] = tb; }
}
__global__ __launch_bounds__(256) void scan2_sums_kernel(uint32_t* __restrict__ sums_a, uint32_t* __restrict__ sums_b, uint32_t n_tiles,
                                                         unsigned long long* __restrict__ totals) {
    __shared__ uint32_t wave_sums[4];
    unsigned long long run_a = 0ull, run_b = 0ull;   // (the offsets written back are used only once the totals are known to fit 32 bits)
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += kThreads) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t va = t < n_tiles ? sums_a[t] : 0u, vb = t < n_tiles ? sums_b[t] : 0u;
        uint32_t ta, tb;
        const uint32_t ea = block_exclusive_scan(va, wave_sums, &ta);
        const uint32_t eb = block_exclusive_scan(vb, wave_sums, &tb);
        if (t < n_tiles) { sums_a[t] = uint32_t(run_a) + ea; sums_b[t] = uint32_t(run_b) + eb; }
        run_a += ta;
        run_b += tb;
    }
    if (threadIdx.x == 0) { totals[0] = run_a; totals[1] = run_b; }
}
__global__ __launch_bounds__(256) void scan2_apply_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n,
                                                          const uint32_t* __restrict__ sums_a, const uint32_t* __restrict__ sums_b,
                                                          uint32_t* __restrict__ out_a, uint32_t* __restrict__ out_b) {
    __shared__ uint32_t wave_sums[4];
    const uint32_t i0 = blockIdx.x * kTile + threadIdx.x * kItems;
    uint32_t va[kItems], vb[kItems], sa = 0u, sb = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kItems; k++) {
        va[k] = i0 + k < n ? a[i0 + k] : 0u;
        vb[k] = i0 + k < n ? b[i0 + k] : 0u;
        sa += va[k];
        sb += vb[k];
    }
    uint32_t ta, tb;
    uint32_t ea = sums_a[blockIdx.x] + block_exclusive_scan(sa, wave_sums, &ta);
    uint32_t eb = sums_b[blockIdx.x] + block_exclusive_scan(sb, wave_sums, &tb);
#pragma unroll
    for (uint32_t k = 0; k < kItems; k++)
        if (i0 + k < n) {
            out_a[i0 + k] = ea;
            out_b[i0 + k] = eb;
            ea += va[k];
            eb += vb[k];
        }
}
inline size_t scan2_temp_bytes(uint32_t n) { return size_t(2) * rsort_tiles(n) * sizeof(uint32_t); }
inline hipError_t exclusive_scan2(const uint32_t* a, const uint32_t* b, uint32_t n, uint32_t* out_a, uint32_t* out_b, unsigned long long* totals,
                                  void* temp, hipStream_t st) {
    const uint32_t n_tiles = rsort_tiles(n);
    uint32_t* const sums_a = static_cast<uint32_t*>(temp);
    uint32_t* const sums_b = sums_a + n_tiles;
    if (n_tiles) hipLaunchKernelGGL(scan2_tile_sums_kernel, dim3(n_tiles), dim3(kThreads), 0, st, a, b, n, sums_a, sums_b);
    hipLaunchKernelGGL(scan2_sums_kernel, dim3(1), dim3(kThreads), 0, st, sums_a, sums_b, n_tiles, totals);
    if (n_tiles) hipLaunchKernelGGL(scan2_apply_kernel, dim3(n_tiles), dim3(kThreads), 0, st, a, b, n, sums_a, sums_b, out_a, out_b);
    return hipGetLastError();
}

}  // namespace ss
}  // namespace rptg
