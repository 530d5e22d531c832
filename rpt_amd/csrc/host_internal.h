// host_internal.h — internals of rpt_capi.cpp shared with the photon-mapping translation unit.
#pragma once
#include <hip/hip_runtime_api.h>

#include <functional>
#include <string>

#include "../../include/rpt_hip.h"
#include "kernels.h"

namespace rpt64 { struct Args; }

namespace rpti {
int fail(int code, const std::string& msg);
uint64_t seed_mix(uint64_t seed);
struct SceneDev {
    bool committed;
    int device, n_cus;
    rptg::SceneView view;
    int first_object_light;  // index into scene.lights of the first Light::Object, or -1
    bool epsilon64;          // committed in the reference-epsilon mode (epsilon_policy = 1)
};
SceneDev scene_dev(rpt_scene* s);
void*& photon_slot(rpt_scene* s);  // owned by photon.hip (PhotonMapDev*), released through photon_release
void photon_release(void* p);      // defined in photon.hip
// Fills camera, tiles, slab, queue, chunking exactly as for the path tracer; `st` is the stream the launch will run
// on (it selects the launch set: slab + work counter).
// min_chunk: lower bound of the automatic samples-per-work-item choice (an explicit "chunk_spp" option wins).
// fixed_chunk != 0: the caller's kernel has its own work decomposition with exactly that many samples per chunk
// (the photon camera pass: 64); option and automatic rule are ignored, so the slab [n_chunks][n_owned] this
// function sizes is the one that kernel and resolve_kernel index.
int prepare_render(rpt_scene* s, hipStream_t st, const rpt_camera* cam, const rpt_render_params* prm, uint32_t iterations, uint64_t seed,
                   uint32_t sample_offset, rptg::RenderArgs& a, uint32_t min_chunk = 0, uint32_t fixed_chunk = 0,
                   uint32_t slab_item_bytes = 16);   // 32: the reference-epsilon mode's partial sums are fp64
// Zeroes the queue / sharded frame, calls `launch(args, n_blocks, stream)` with a persistent grid of
// blocks_per_cu blocks per CU, then resolves the slab into d_out.
int run_persistent(rpt_scene* s, const rpt_render_params* prm, const rptg::RenderArgs& a, double* d_out, hipStream_t st,
                   int blocks_per_cu, const std::function<hipError_t(const rptg::RenderArgs&, int, hipStream_t)>& launch,
                   bool indexed_start = false, bool wave_items = false,   // wave_items: n_items counts one item per wave, not per lane
                   const std::function<hipError_t(double, double*, hipStream_t)>& resolve = nullptr);   // (scale, d_out, stream): instead of resolve_kernel
int serialize_with_other_streams(rpt_scene* s, hipStream_t st);  // for launches with per-scene scratch outside the launch set
int fetch_counters(rpt_scene* s, const rptg::RenderArgs& a);  // after the stream has been synchronised
// Arguments the fp64 kernels share (rpt_capi.cpp): scene; camera, frame and `a`'s tiles / chunking / work counter / slab when given.
void fill_args64(rpt_scene* s, const rpt_camera* cam, const rpt_render_params* prm, const rptg::RenderArgs* a, rpt64::Args& q);
double* scratch_out(rpt_scene* s, size_t bytes);  // cached device frame for the host-buffer entry points
int64_t option_photon_skip(rpt_scene* s);  // option "photon_skip" of the scene: diagnostic bit mask for the camera pass
int64_t option_f64_photon_slice(rpt_scene* s);    // option "f64_photon_slice": samples per slice of the reference-epsilon photon camera pass (0: automatic)
int64_t option_photon_parts(rpt_scene* s);        // option "photon_parts": strips per 8x8 pixel block of the camera pass (1, 2, 4, 8)
int64_t option_photon_coop_gather(rpt_scene* s);  // option "photon_coop_gather": wave-level surface gather on (default) / off
int64_t option_photon_split(rpt_scene* s);        // option "photon_split": volume and surface estimate of the beam kinds in two launches (default off: measured slower)
int64_t option_photon_block_lists(rpt_scene* s);  // option "photon_block_lists": per-block candidate lists on (default) / off
}  // namespace rpti

#define RPTI_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return rpti::fail(RPT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));  \
    } while (0)
