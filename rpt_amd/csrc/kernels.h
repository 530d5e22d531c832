// kernels.h — host-callable launchers of the HIP kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include "gpu_layout.h"

namespace rptg {

struct RenderArgs {
    SceneView sc;
    CameraG cam;
    uint32_t width, height;
    float inv_dim;            // 1 / max(width, height)
    uint32_t max_bounces;
    uint32_t iterations;      // paths per pixel in this call
    uint32_t sample_offset;
    uint32_t chunk_spp;       // samples per work item
    uint32_t n_chunks;        // ceil(iterations / chunk_spp)
    uint64_t seed_mixed;      // mix64(seed + GOLDEN)
    const uint32_t* tiles;    // owned 32x32 tiles (tile id = ty * tiles_x + tx)
    uint32_t n_tiles, tiles_x;
    uint32_t n_owned;         // n_tiles * 1024 pixel slots
    uint32_t n_items;         // n_owned * n_chunks
    float* slab;              // [n_chunks][n_owned] float4 partial sums
    float* slab2;             // a second term per (chunk, pixel) that resolve_kernel adds (the split photon camera pass), or null
    unsigned long long* queue;  // 64-bit work counter (zeroed before the launch)
    unsigned long long* counters;  // 8 x u64 or nullptr
    uint32_t lds_stack;       // 1: BVH stack in dynamic LDS
    uint32_t defer_lanes;     // per-mesh-tree kernels: parked tree walks per wave that trigger a walk (1..64)
    uint32_t defer_stop;      // ... and the number of still-walking lanes below which the wave leaves the walk
    uint32_t walk_leaf_quarters;  // ... and the descent of a walk pauses for the leaves when 4 x (lanes at a leaf) >= this x (lanes descending); 0: never
    uint32_t detach;              // per-mesh-tree kernels in a medium: 1 = shadow queries that need a tree walk leave their path (wave queue
                                  // in LDS); 2 = every tree walk leaves its path (ring + parked path contexts in stream_scratch)
    uint32_t pull_batch;          // lanes that must wait for a new work item before the wave runs the item bookkeeping (it also runs when no lane has anything else to do)
    uint32_t detach_trigger;      // detach = 1: a walk session is due as soon as the queue holds this many (1..32)
    uint32_t stream_backlog;      // detach = 2: ... as soon as the ring holds this many queries
    uint32_t stream_contexts;     // detach = 2: parked paths per lane (1..6)
    uint32_t n_twin_lights;       // object lights that can be visible (each may write one shadow query per lane and trip)
    uint32_t* stream_scratch;     // detach = 2: stream_scratch_bytes_per_block() per block of the grid
};

// The device functions read the scene view at kernarg + 0 (kernarg_scene in device_core.h): every kernel that calls them
// takes ONE argument struct that begins with the SceneView.
static_assert(offsetof(RenderArgs, sc) == 0, "RenderArgs must begin with the SceneView");

struct KernelInfo {
    int vgprs, sgprs, lds, max_blocks_per_cu;
};

// Persistent megakernel: grid = n_blocks x 256 threads.
hipError_t launch_render(const RenderArgs& a, int n_blocks, hipStream_t stream);
hipError_t render_occupancy(bool medium, int bvh, int* blocks_per_cu, int detach = 0);  // bvh: bvh_mode(); detach: RenderArgs::detach
size_t stream_scratch_bytes_per_block();
int bvh_mode(const SceneView& sc);  // 0 no tree, 1 per-mesh trees, 2 scene-level tree
// out[pixel] = sum_chunks slab / iterations * scale for owned pixels (others untouched).
hipError_t launch_buffer_add(uint32_t n_pixels, const double* d_batch, double* d_sum, double* d_sumsq, hipStream_t st);
hipError_t launch_buffer_image(uint32_t w, uint32_t h, uint32_t radius, uint32_t n_batches, const double* d_sum, uint8_t* d_out,
                               hipStream_t st);
hipError_t launch_buffer_variance(uint32_t n_pixels, uint32_t n_batches, const double* d_sum, const double* d_sumsq, double* d_out,
                                  hipStream_t st);
hipError_t launch_resolve(const RenderArgs& a, double scale, double* d_out, hipStream_t stream);
// Frame exchange: owned tiles <-> packed blocks of 32 x 32 x 3 f64 (tile list on the device).
hipError_t launch_frame_pack(const double* d_frame, double* d_packed, const uint32_t* d_tiles, uint32_t n_tiles, uint32_t tiles_x,
                             uint32_t width, uint32_t height, hipStream_t st);
hipError_t launch_frame_unpack(const double* d_packed, double* d_frame, const uint32_t* d_tiles, uint32_t n_tiles, uint32_t tiles_x,
                               uint32_t width, uint32_t height, hipStream_t st);
hipError_t launch_intersect(const SceneView& sc, uint64_t n, const float* d_o, const float* d_d, float* d_t,
                            int32_t* d_obj, float* d_n, bool bvh, hipStream_t stream);
hipError_t launch_debug_rng(uint64_t seed_mixed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* d_out,
                            hipStream_t stream);
hipError_t launch_debug_sample_f(const Material& m, uint64_t n, const float* d_n, const float* d_wo,
                                 uint64_t seed_mixed, float* d_wi, float* d_pdf, int32_t* d_some, hipStream_t s);
hipError_t launch_debug_bsdf(const Material& m, uint64_t n, const float* d_n, const float* d_wo, const float* d_wi,
                             float* d_out, hipStream_t s);
hipError_t launch_debug_camera(const CameraG& cam, uint32_t w, uint32_t h, uint64_t seed_mixed, uint32_t sample,
                               float* d_o, float* d_d, hipStream_t s);

}  // namespace rptg
namespace rpt64 { struct Args; struct ShootArgs64; struct SurfArgs64; }
namespace rptg {
// Reference-epsilon mode (kernels_f64.hip): persistent grid over (pixel, chunk) items with an fp64 slab, then its resolve.
hipError_t launch_render_f64(const rpt64::Args& a, int n_blocks, hipStream_t stream);
hipError_t launch_resolve_f64(const rpt64::Args& a, double scale, double* d_out, hipStream_t stream);
hipError_t render_f64_occupancy(bool medium, int* blocks_per_cu);
// Photon mapping in that mode: the shooting pass (count pass when a.surf and a.vol are null) and the camera pass's surface estimate.
hipError_t launch_photon_shoot_f64(const rpt64::ShootArgs64& a, int n_blocks, hipStream_t stream);
hipError_t launch_photon_surface_f64(const rpt64::SurfArgs64& a, int n_blocks, hipStream_t stream);
hipError_t launch_resolve_photon_f64(const rpt64::Args& a, const void* slab32, uint32_t n_chunks32, double scale_over_total, bool accumulate,
                                     double* d_out, hipStream_t stream);
}  // namespace rptg
