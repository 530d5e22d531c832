/* rpt_hip.h — C ABI of the MI355X-native path-tracing core (drop-in for rpt's hot path).
 *
 * The reference (neevparikh/rpt, Rust, #![forbid(unsafe_code)], src/lib.rs:3) has no FFI
 * seam.  The seam this library replaces is the private method
 *     Renderer::sample(&self, iterations: u32, buffer: &mut Buffer)     src/renderer.rs:158-171
 * i.e. (immutable Scene, Camera, width, height, exposure_value, max_bounces, iterations)
 *      -> width*height linear-RGB f64 means, row-major, y = 0 at the top,
 * which `render()` (src/renderer.rs:137-141) and `iterative_render()` (:144-156) call and
 * feed to `Buffer::add_samples` (src/buffer.rs:32-40).  The entry points below are what a
 * Rust `extern "C"` block (shown in INTEGRATION.md), or the C++ mirror in include/rpt.hpp,
 * binds: plain pointers and sizes, opaque handles owned by the caller, caller-allocated
 * output buffers, no caller pointer retained past a call (mesh data is copied at `add`).
 *
 * Every function returns 0 on success and a negative code on error (never aborts);
 * rpt_last_error() returns the thread-local message of the last failure.
 */
#ifndef RPT_HIP_H
#define RPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPT_OK 0
#define RPT_ERR_INVALID (-1)   /* bad argument / would panic in the reference          */
#define RPT_ERR_STATE (-2)     /* call order (e.g. add after commit, render before)    */
#define RPT_ERR_DEVICE (-3)    /* HIP runtime failure (message carries hipGetErrorString) */
#define RPT_ERR_UNSUPPORTED (-4)

/* Opaque scene handle: mirrors `Scene` (src/scene.rs:12-24). */
typedef struct rpt_scene rpt_scene;

/* Shape kinds: the closed set of `impl Shape` on the hot path
 * (src/shape/sphere.rs, cube.rs, plane.rs, mesh.rs; `Mesh = KdTree<Triangle>`). */
enum { RPT_SHAPE_SPHERE = 0, RPT_SHAPE_CUBE = 1, RPT_SHAPE_PLANE = 2, RPT_SHAPE_MESH = 3,
       RPT_SHAPE_GROUP = 4 /* KdTree<Box<dyn Bounded>> of other shapes (src/kdtree.rs:103-146,
                              examples/fractal_spheres.rs:45); children must be Bounded (no planes) */ };

/* One `Box<dyn Shape>`: a unit primitive or mesh, optionally wrapped in `Transformed<T>`
 * (src/shape.rs:102-152).  `transform` is the composed homogeneous matrix M, row-major;
 * chained builder calls left-multiply (src/shape.rs:237-284).  The library derives
 * M^-1, the linear part, M^-T and det exactly as `Transformed::new` does (:112-125). */
typedef struct rpt_shape_desc {
    int32_t kind;            /* RPT_SHAPE_*                                              */
    int32_t has_transform;   /* 0: bare shape, 1: Transformed<shape>                     */
    double transform[16];    /* row-major 4x4, used iff has_transform                    */
    double plane_normal[3];  /* Plane { normal, value } (src/shape/plane.rs:7-13)        */
    double plane_value;
    const double* tris;      /* n_tris * 18 doubles: v1 v2 v3 n1 n2 n3 (src/shape/mesh.rs:9-23) */
    uint64_t n_tris;
    const struct rpt_shape_desc* children; /* RPT_SHAPE_GROUP: the kd-tree's objects, each a     */
    uint64_t n_children;                   /* full shape (own transform, may itself be a group)  */
} rpt_shape_desc;

/* `enum Material` (src/material.rs:8-23). */
enum { RPT_MAT_LAMBERTIAN = 0, RPT_MAT_PHONG = 1, RPT_MAT_MIRROR = 2, RPT_MAT_TRANSMISSIVE = 3 };
typedef struct rpt_material {
    int32_t kind;
    int32_t _pad;
    double albedo[3];
    double emittance;  /* Lambertian / Phong */
    double shininess;  /* Phong */
    double ior;        /* Transmissive */
} rpt_material;

/* `Medium` constructors (src/medium.rs:80-122): the fields are private in the reference,
 * so these two are the closed set a user can create. */
enum { RPT_MEDIUM_HOMOGENEOUS_ISOTROPIC = 0, RPT_MEDIUM_COLORED_GLOWING_FOG = 1 };

/* `Camera` (src/camera.rs:9-27). */
typedef struct rpt_camera {
    double eye[3], direction[3], up[3];
    double fov, aperture, focal_distance;
} rpt_camera;

/* The `Renderer` fields the sampling path reads (src/renderer.rs:23-56) plus tile sharding. */
typedef struct rpt_render_params {
    uint32_t width, height;
    double exposure_value;
    uint32_t max_bounces;
    /* Multi-GPU tile sharding (no counterpart in the reference, which forks rayon tasks per
     * row, src/renderer.rs:159-162): 32x32 pixel tiles, tile (tx,ty) is rendered iff
     * (tx + ty) % shard_count == shard_rank; other pixels are written as 0 so a sum-reduce
     * over ranks assembles the frame bit-exactly.  shard_count = 0 or 1: whole frame. */
    uint32_t shard_rank, shard_count;
} rpt_render_params;

int rpt_device_count(void);
/* Tile ownership used by the renderer (pure host function): writes the ids (ty * tiles_x + tx,
 * tiles_x = ceil(width/32)) of the 32x32 tiles owned by shard_rank, in render order, and returns
 * their number (or a negative error).  tiles_out may be NULL to query the count. */
int64_t rpt_shard_tiles(uint32_t width, uint32_t height, uint32_t shard_rank, uint32_t shard_count,
                        uint32_t* tiles_out, uint64_t capacity);
const char* rpt_last_error(void);

rpt_scene* rpt_scene_create(void);                 /* Scene::new()            src/scene.rs:26-31 */
void rpt_scene_destroy(rpt_scene*);
/* SceneAdd<Object>  src/scene.rs:40-44; returns the object index (>= 0) or an error. */
int rpt_scene_add_object(rpt_scene*, const rpt_shape_desc*, const rpt_material*);
/* SceneAdd<Light>   src/scene.rs:46-50, one entry point per `enum Light` arm (src/light.rs:7-19). */
int rpt_scene_add_light_point(rpt_scene*, const double color[3], const double location[3]);
int rpt_scene_add_light_ambient(rpt_scene*, const double color[3]);
int rpt_scene_add_light_directional(rpt_scene*, const double color[3], const double direction[3]);
/* Light::Object.  A plane is rejected (Plane::sample is unimplemented!(), src/shape/plane.rs:34). */
int rpt_scene_add_light_object(rpt_scene*, const rpt_shape_desc*, const rpt_material*);
/* SceneAdd<Medium>  src/scene.rs:77-81.  Only media[0] is used (src/renderer.rs:190). */
int rpt_scene_add_medium(rpt_scene*, int32_t kind, double absorption, double scattering);
/* Environment::Color (src/environment.rs:56-77). */
int rpt_scene_set_environment_color(rpt_scene*, const double rgb[3]);
/* Environment::Hdri(Hdri::new(width, height, buf)) (src/environment.rs:3-52): equirectangular image of
 * width*height linear-RGB triples, row-major, looked up bilinearly by direction. */
int rpt_scene_set_environment_hdri(rpt_scene*, uint32_t width, uint32_t height, const double* rgb);
/* Flatten to the device layout and upload; the scene is immutable afterwards
 * (the reference shares `&Scene` immutably across rayon workers, src/renderer.rs:25). */
int rpt_scene_commit(rpt_scene*, int device);

/* Renderer::sample (src/renderer.rs:158-171) for `iterations` paths per pixel.
 * out_rgb: width*height*3 doubles, row-major, y = 0 top; each pixel =
 * mean(trace_ray) * 2^exposure_value, exactly what get_color returns (:173-184).
 * seed / sample_offset key the counter-based RNG stream per (pixel, sample_offset + s):
 * an additive deviation (the reference seeds from entropy, :163). */
int rpt_render_sample(rpt_scene*, const rpt_camera*, const rpt_render_params*, uint32_t iterations,
                      uint64_t seed, uint32_t sample_offset, double* out_rgb);
/* Same, asynchronous: d_out_rgb is a DEVICE pointer (width*height*3 doubles) on the scene's
 * device and hip_stream a hipStream_t (NULL = default stream).  Nothing is copied to the host.  Calls on one
 * stream run in order; calls that alternate between two streams overlap (the scene keeps the per-launch scratch
 * twice), which hides the tail of each launch behind the start of the next; a third stream waits for the scratch
 * it takes over (photon-mapped renders of one scene never overlap: their photon scratch exists once).  Calls on
 * the same scene must still come from one host thread at a time. */
int rpt_render_sample_device(rpt_scene*, const rpt_camera*, const rpt_render_params*, uint32_t iterations,
                             uint64_t seed, uint32_t sample_offset, void* d_out_rgb, void* hip_stream);

/* Renderer::get_closest_hit (src/renderer.rs:416-425) over n rays (host pointers, fp32).
 * t = +inf, object = -1 on a miss.  normal may be NULL. */
int rpt_intersect_batch(rpt_scene*, uint64_t n, const float* origins, const float* dirs, float* t,
                        int32_t* object, float* normal);

/* Flattened-layout statistics of a committed scene: [0] spheres, [1] general (rotated) cubes,
 * [2] planes, [3] linearly scanned triangles, [4] axis-aligned boxes, [5] axis-aligned
 * rectangles (pairs of wall triangles), [6] BVH triangles, [7] BVH nodes, [8] bytes of scan
 * records every closest-hit query walks, [9] bytes of scene data resident in HBM, [10] 1 if one scene-level
 * tree replaces the scan (2: ... and the meshes with trees of their own stay outside it, their walks parked), [11] primitives in it, [12] mesh instances, [13] meshes stored once and instanced,
 * [14] wall rectangles folded into a box shell, [15] levels of the deepest walk (scene tree + mesh tree). */
int rpt_scene_stats(rpt_scene*, uint64_t out[16]);
/* Counters of the last rpt_render_sample* call on this scene (device-side, exact):
 * [0] camera samples, [1] closest-hit queries (rays), [2] path vertices, [3] kernel loop trips
 * (wave-iterations summed over waves), [4] primitive tests, [5] BVH nodes visited,
 * [6] BVH triangle tests, [7] tree walks that found their stack full (must be 0: scenes whose trees could overflow it
 * are refused at commit).  Filled only when the library is built with RPT_COUNTERS or
 * rpt_set_option("counters", 1) was called before the render; otherwise zeros (also for a scene with a
 * group as a Light::Object: that kernel flavour has no counters build). */
int rpt_get_counters(rpt_scene*, uint64_t out[8]);
/* Diagnostic (counters on): for section k of the megakernel's loop body (kernels.hip, SECT(k)),
 * out[2k] = wave-level executions and out[2k+1] = lanes active in them during the last path-traced
 * render: the lane utilisation of each divergent piece of code. */
int rpt_debug_section_counters(rpt_scene*, uint64_t out[56]);
/* HIP-event timing of the last render on this scene (needs rpt_set_option("timing", 1)):
 * milliseconds of the megakernel and of the resolve kernel on the stream they ran on, and the
 * persistent grid size.  Synchronises on the last recorded event. */
int rpt_get_timing(rpt_scene*, double* render_ms, double* resolve_ms, int32_t* grid_blocks);
/* The same over every timed render since the previous call of this function (at most the latest 1024): mean
 * milliseconds per launch.  Renders do not wait for their events, so a loop of rpt_render_sample_device calls
 * stays asynchronous and is measured afterwards.  Starts a new measurement. */
int rpt_get_timing_mean(rpt_scene*, double* render_ms, double* resolve_ms, int32_t* launches);
/* The work decomposition rpt_render_sample* will use for `iterations` samples per pixel (pure host function of
 * `iterations` and the "chunk_spp" option): samples per work item and work items (= partial-sum slab entries of
 * 16 bytes) per pixel. */
int rpt_render_chunking(uint32_t iterations, uint32_t* chunk_spp, uint32_t* n_chunks);
/* The same for one scene: reads that scene's "chunk_spp" option (rpt_scene_set_option), i.e. exactly what its
 * renders use; rpt_render_chunking reads the process defaults. */
int rpt_scene_render_chunking(rpt_scene*, uint32_t iterations, uint32_t* chunk_spp, uint32_t* n_chunks);
/* Options.  Every scene has its own set: a copy of the process defaults taken by rpt_scene_create, changed with
 * rpt_scene_set_option (before rpt_scene_commit for the options the commit reads, at any time for the others; nothing
 * a commit or render reads is process-global, so scenes with different options may be driven from different host
 * threads).  rpt_set_option changes the defaults, i.e. the scenes created afterwards (and what rpt_render_chunking
 * reports).  Names (all optional): "counters" 0/1, "chunk_spp" (samples per work item, 0 = auto),
 * "blocks_per_cu" (persistent grid size), "timing" 0/1, "scene_bvh_min" (read by rpt_scene_commit:
 * number of bounded primitives + BVH meshes from which one scene-level BVH replaces the linear
 * object scan, default 64), "instancing" 0/1 (read by rpt_scene_commit: store a mesh that several
 * shapes share once and instance it, default 1), "room_shell" 0/1 (read by rpt_scene_commit: answer the
 * rectangles that are the faces of one axis-aligned box with a single slab test, default 1),
 * "photon_block_lists" 0/1 (camera pass of the beam x point kind: collect the photon spheres of each strip of an
 * 8x8 pixel block once per work item and test them with one photon per lane, default 1; 0 walks the tree per
 * sample), "photon_parts" (work items per 8x8 pixel block and sample chunk of the photon camera pass: the block's
 * rows in 1, 2, 4 or 8 strips, default 4; changes the fp32 order of a pixel's beam sum, nothing else),
 * "photon_coop_gather" 0/1 (surface estimate of the photon camera pass: the wave collects the candidates of a
 * pixel's samples together and every lane picks its K nearest from that list, default 1; 0 searches per lane),
 * "photon_skip" (diagnostic bit mask that switches parts of the photon camera pass off), "defer_lanes" / "defer_stop" (scenes whose meshes
 * have their own trees: a wave starts its parked tree walks when this many lanes wait, default 32, and leaves
 * them when fewer than this many are still walking, default 16; the image does not depend on either),
 * "detach_shadows" 0/1/2 (scenes whose meshes have their own trees, in a medium: 1, the default: a shadow query that needs a tree walk
 * leaves its path and is answered from a queue of its wave, "detach_lanes" (default 44) waiting + queued queries or "detach_trigger"
 * (default 28) queued ones start a walk session; 0: shadow queries park like primary ones; 2: primary queries leave as well, their
 * paths wait in memory, "stream_backlog" (default 48) queued queries start a session and a lane may have "stream_contexts" (1..6,
 * default 1) paths waiting -- slower, kept for measurement; the image depends on none of the thresholds),
 * "scene_tree_meshes" 0/1 (read by rpt_scene_commit: in a scene that has a scene-level tree, 1 makes the meshes with trees of their own
 * leaves of it; default 0: their walks are parked beside it), "bvh_sweep_below" (read by rpt_scene_commit: ranges of at most this many
 * triangles are split by an exact SAH sweep instead of 16 bins, default 4096, 0 = bins only),
 * "photon_split" 0/1 (camera pass of the beam kinds in a medium: volume and surface estimate as two launches, default 0 -- slower),
 * "pull_batch" (path tracer: a wave hands out new work items when this many of its lanes wait for one, or when none of its lanes
 * has anything else to do; 1..64, default 2; no effect on the image),
 * "walk_leaf_quarters" (same scenes: the descent of such a walk pauses for the triangle tests as soon as 4 x the lanes
 * waiting at a leaf >= this x the lanes still descending, default 6, 0 = when every lane is at a leaf; no effect on the image),
 * "bvh_leaf_max" (read by rpt_scene_commit: triangles per leaf of a mesh tree, default 4 -- C5: 49.8 / 43.1 / 41.1 /
 * 41.1 / 41.7 ms for 1 / 2 / 4 / 6 / 8), "bvh_max_depth" (read by rpt_scene_commit: a mesh tree that the SAH builder makes deeper than this is rebuilt
 * with object-median splits, default and maximum 20 -- the traversal stack holds 21 entries per mesh tree, 32 for scene tree + mesh tree;
 * a scene that still does not fit is refused with RPT_ERR_UNSUPPORTED);
 * returns RPT_ERR_INVALID for unknown names. */
int rpt_set_option(const char* name, int64_t value);
int rpt_scene_set_option(rpt_scene*, const char* name, int64_t value);

/* ---- Buffer on the device (src/buffer.rs:5-97): the samples of each pixel are kept as running
 * sums, so image() = box filter (Filter::Box(radius), :76-97) + color_bytes (src/color.rs:18-24)
 * and variance() (:60-74) run where the frame is and only width*height*3 bytes come back. */
typedef struct rpt_buffer rpt_buffer;
rpt_buffer* rpt_buffer_create(int device, uint32_t width, uint32_t height, uint32_t filter_radius); /* Buffer::new */
void rpt_buffer_destroy(rpt_buffer*);
int rpt_buffer_add_samples(rpt_buffer*, const double* rgb /* host, width*height*3 */);           /* Buffer::add_samples */
int rpt_buffer_add_samples_device(rpt_buffer*, const void* d_rgb, void* hip_stream);
int rpt_buffer_image(rpt_buffer*, uint8_t* out_rgb8 /* host, width*height*3 */);                 /* Buffer::image */
int rpt_buffer_variance(rpt_buffer*, double* out);                                               /* Buffer::variance */
int rpt_buffer_batches(rpt_buffer*, uint32_t* n);
/* Renderer::sample(&self, iterations, &mut Buffer) itself (src/renderer.rs:158-171): render one
 * batch and push its means into the buffer without leaving the device. */
int rpt_render_into_buffer(rpt_scene*, const rpt_camera*, const rpt_render_params*, uint32_t iterations,
                           uint64_t seed, uint32_t sample_offset, rpt_buffer*);

/* ---- photon mapping (next tier: src/photon.rs; config C4 = photon_point_query_beam_render) ----
 * `enum PhotonRenderKind` (src/photon.rs:631-639): point-point (photon_map_render), beam-point
 * (photon_point_query_beam_render, config C4) and beam-beam (photon_beam_query_beam_render). */
enum { RPT_PHOTON_MAP = 0, RPT_PHOTON_POINT_BEAM = 1, RPT_PHOTON_BEAM_BEAM = 2 };
/* Renderer::photon_render, first half (src/photon.rs:655-704): shoot `photon_count` photons of
 * power watts/photon_count from the first Light::Object (shoot_photon / trace_photon, :724-946),
 * build the surface and volume point maps and the per-photon gather radii (:204-247).  The map is
 * stored in the scene handle and replaced by the next build.  Photon i draws from the RNG stream
 * (seed, i, 0x80000000 + (i >> 32)). */
int rpt_photon_map_build(rpt_scene*, uint64_t photon_count, int32_t kind, double watts, uint64_t seed);
/* The same map built by several GPUs (one process each): the "Shooting photons" loop
 * (src/photon.rs:656-690) is sharded by photon index, the map is needed whole on every GPU.
 *   1. rpt_photon_shoot: shoot photons [rank*N/count, (rank+1)*N/count) of the N-photon map and keep
 *      their records on the device; n_out = {surface, volume} record counts of this shard.
 *   2. rpt_photon_records: device pointer + count of those records (RPT_PHOTON_RECORD_BYTES each,
 *      shooting order); the caller all-gathers them in rank order (RCCL), which reproduces the
 *      single-GPU arrays exactly because the blocks are contiguous.
 *   3. rpt_photon_map_from_records: build the maps of rpt_photon_map_build from device arrays
 *      (the gathered ones; they are read, not kept).  Invalidates the pointers of step 2. */
#define RPT_PHOTON_RECORD_BYTES 48
int rpt_photon_shoot(rpt_scene*, uint64_t photon_count, int32_t kind, double watts, uint64_t seed,
                     uint32_t shard_rank, uint32_t shard_count, uint64_t n_out[2]);
int rpt_photon_records(rpt_scene*, int32_t which /* 0 surface, 1 volume */, void** d_records, uint64_t* n);
int rpt_photon_map_from_records(rpt_scene*, uint64_t photon_count, int32_t kind, const void* d_surface,
                                uint64_t n_surface, const void* d_volume, uint64_t n_volume);
/* [0] surface photons, [1] volume photons, [2] photons shot, [3] shooting us, [4] map build us. */
int rpt_photon_map_stats(rpt_scene*, uint64_t out[8]);
/* Test hook: which = 0 surface / 1 volume; out = n * 10 floats in shooting order:
 * position, direction (toward the previous vertex), power, gather radius (volume photons). */
int rpt_photon_map_download(rpt_scene*, int32_t which, float* out, uint64_t capacity_photons);
/* Renderer::photon_render, second half = get_color_with_photon_map over the frame
 * (src/photon.rs:706-716, 950-985; estimate_indirect :316-628) with `num_samples` camera
 * samples per pixel; same output convention, seed keying and sharding as rpt_render_sample. */
int rpt_photon_render_sample(rpt_scene*, const rpt_camera*, const rpt_render_params*, uint64_t gather_size,
                             uint64_t gather_size_volume, uint32_t num_samples, uint64_t seed,
                             uint32_t sample_offset, double* out_rgb);
int rpt_photon_render_sample_device(rpt_scene*, const rpt_camera*, const rpt_render_params*, uint64_t gather_size,
                                    uint64_t gather_size_volume, uint32_t num_samples, uint64_t seed,
                                    uint32_t sample_offset, void* d_out_rgb, void* hip_stream);

/* ---- frame exchange between the GPUs of one node (the loop being sharded is src/renderer.rs:158-171) ----
 * One process per GPU renders the 32x32 tiles it owns (rpt_render_params.shard_rank / shard_count) into a full-size
 * device frame; rank 0 assembles the image.  The exchange is a gather of the OWNED tiles over RCCL (xGMI), not a reduce
 * of whole frames: rank r packs its tiles (rpt_shard_tiles order; 32 x 32 pixels x 3 f64 each, pixels of a clipped tile
 * that lie outside the image are zero) and sends them to rank 0, which receives every rank's block in one ncclGroup and
 * scatters it into its frame.  The payload stays f64 -- the frame's own type -- so the assembled frame is bit-identical
 * to the one a single GPU renders (C5: 12.6 MB per rank instead of a 100 MB zero-padded frame per rank).
 * RCCL is loaded at run time (dlopen of librccl.so.1, the copy already in the process if there is one): the library
 * has no link-time dependency on it and single-GPU users never touch it. */
typedef struct rpt_comm rpt_comm;
#define RPT_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks by any means (the launcher's store). */
int rpt_comm_unique_id(void* id_out);
/* ncclCommInitRank on `device` (collective over the n_ranks processes). */
int rpt_comm_create(const void* id, int rank, int n_ranks, int device, rpt_comm** out);
void rpt_comm_destroy(rpt_comm*);
int rpt_comm_rank(const rpt_comm*, int* rank, int* n_ranks);
/* Collective.  d_shard: this rank's frame (width*height*3 f64 on its device, as written by rpt_render_sample_device
 * with shard_rank = the communicator's rank); d_frame: on rank 0 the assembled frame (may be d_shard itself: the
 * received tiles are then written in place), ignored on the other ranks.  Everything is enqueued on hip_stream.
 * flags: RPT_GATHER_LOOPBACK makes rank 0 send its own tiles to itself through RCCL as well (self-test of the
 * transport with one rank; needs d_frame != d_shard to be observable). */
#define RPT_GATHER_LOOPBACK 1u
int rpt_gather_frame_device(rpt_comm*, uint32_t width, uint32_t height, const void* d_shard, void* d_frame,
                            uint32_t flags, void* hip_stream);
/* The photon maps' exchange step: the shooting loop (src/photon.rs:656-690) is sharded by photon index (rpt_photon_shoot) and
 * every rank builds the whole map, so the shot records (rpt_photon_records: RPT_PHOTON_RECORD_BYTES each) are all-gathered in
 * rank order -- contiguous blocks in rank order ARE the single-GPU arrays.  Collective.  d_local / n_local: this rank's records
 * (device); d_out: room for `capacity` records (device); n_per_rank (optional, n_ranks values) and n_total are filled on the host;
 * the call synchronises hip_stream once (the second all-gather is sized by the counts of the first).  If `capacity` is too
 * small the call fails with RPT_ERR_INVALID after filling n_total -- on every rank alike -- and can be repeated with room.
 * d_out = NULL with capacity = 0 exchanges the counts only (RPT_OK): how a caller learns the size to bring (a photon stores
 * one record per scattering event, so no multiple of the photon count bounds it). */
int rpt_allgather_records_device(rpt_comm*, const void* d_local, uint64_t n_local, void* d_out, uint64_t capacity,
                                 uint64_t* n_per_rank, uint64_t* n_total, void* hip_stream);
/* The packed layout (pure host function): tile_offsets[r] = first tile of rank r's block in the gathered buffer,
 * r = 0..n_ranks (tile_offsets[n_ranks] = tiles of the whole frame); a tile is 32*32*3 f64. */
int rpt_frame_pack_layout(uint32_t width, uint32_t height, uint32_t n_ranks, uint64_t* tile_offsets);
/* Pack / unpack on one device without any transport (what the exchange does on either side; test hooks):
 * d_packed holds rpt_shard_tiles(rank).count * 3072 f64. */
int rpt_frame_pack_device(uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks, const void* d_frame,
                          void* d_packed, void* hip_stream);
int rpt_frame_unpack_device(uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks, const void* d_packed,
                            void* d_frame, void* hip_stream);

/* ---- reference-epsilon mode: rpt_scene_set_option(scene, "epsilon_policy", 1) before rpt_scene_commit ----
 * The fp32 path replaces the reference's 1e-12 epsilons (src/renderer.rs:17, 348, 396, 420), which fp32 cannot resolve, by
 * scaled tolerances and a geometric twin test; its images are brighter than rpt's by the energy rpt loses to
 * self-intersections and near-miss shadow rejections.  A scene committed with epsilon_policy = 1
 * is rendered by a second, fp64 kernel whose arithmetic follows the reference literally instead: every object the generic
 * shape under its own Transformed matrices, in scene order, t_min = 1e-12, light visible iff |hit - dist| < 1e-12, f64
 * colours, no fused multiply-adds (only the objects a ray's padded fp32 box test keeps are evaluated -- the result is that of
 * the full scan bit for bit; option "f64_cull" = 0 runs the full scan).  Same entry points (rpt_render_sample*,
 * rpt_render_into_buffer), same RNG streams, same sharding; 4-5 times slower than the fp32 path (C3: 2.9 Gsamples/s).
 * Supported: spheres, cubes, planes, meshes (scanned triangle by triangle), KdTree groups of them as objects and as
 * Light::Objects (nested at most three deep), all materials, lights and media, Environment::Color and Environment::Hdri.
 * Photon mapping (rpt_photon_map_build, rpt_photon_render_sample*): the shooting pass and the surface estimate's visibility rays run in
 * fp64 with the reference's tests (t_min = 1e-12; a gathered photon counts unless len > hit.time, src/photon.rs:357-361); the maps, the
 * k-nearest selection and the volume estimates are those of the fp32 records.
 * Refused with RPT_ERR_UNSUPPORTED: groups nested deeper; rpt_photon_shoot / rpt_photon_map_from_records (the 48-byte records do not
 * carry the photons' fp64 positions: every rank builds the whole map with rpt_photon_map_build).
 * Options of the mode: "f64_cull" (1; 0 = full scan, 2 = the counters build keeps the search limits), "f64_surf_batch"
 * (8: lanes of a wave that wait at a surface event in a medium before the wave runs the surface code).
 * "f64_photon_slice" (0 = automatic: as many whole chunks of 256 samples as keep the per-sample selections, (gather_size + 2) dwords
 * each, within 32 GB): samples per slice of the photon camera pass.  "photon_skip" bits 4096 / 16384: no visibility rays / visibility rays without
 * the search limit at the query point (diagnostics).
 * rpt_debug_epsilon_counters (option "counters" = 1): [0] closest-hit queries, [1] accepted hits, [2] accepted hits with
 * t < 1e-9 (1 + |origin|) -- a ray hitting the surface it starts on --, [3] shadow tests, [4] passed, [5] failed although
 * |hit - dist| < 1e-6 dist -- the light's own surface missed by rounding --, [6] camera samples, [7] path vertices; the
 * schedule: [8] objects evaluated in fp64 (all lanes), [9] wave-level evaluation rounds, [10] wave-level loop trips, [11] lanes
 * holding a path summed over the trips. */
int rpt_debug_epsilon_counters(rpt_scene*, uint64_t out[12]);

/* ---- device self-test hooks (each runs the device function in a one-block kernel) ---- */
int rpt_debug_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out);
int rpt_debug_material_sample_f(const rpt_material*, uint64_t n, const float* normals, const float* wos,
                                uint64_t seed, float* wi, float* pdf, int32_t* some);
int rpt_debug_material_bsdf(const rpt_material*, uint64_t n, const float* normals, const float* wos,
                            const float* wis, float* out_rgb);
/* sort_scan.h, the photon-map build's own device-wide primitives (host arrays in and out): the stable radix sort of 64-bit keys --
 * keys_out = the keys in ascending order, order_out[i] = input position of the i-th (equal keys keep their input order) -- and the
 * exclusive prefix sums of two u32 arrays with their 64-bit totals. */
int rpt_debug_radix_sort(uint64_t n, const uint64_t* keys, uint64_t* keys_out, uint32_t* order_out);
int rpt_debug_exclusive_scan2(uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out_a, uint32_t* out_b, uint64_t totals[2]);
/* Reference-epsilon mode: the surface photons' positions as the shooting pass holds them (fp64, 3 per photon, shooting order -- the
 * order of rpt_photon_map_download(which = 0)); capacity in photons. */
int rpt_debug_photon_positions64(rpt_scene*, double* out, uint64_t capacity);
/* ... and what the last camera pass (its last slice) handed from the k-nearest selection to the fp64 surface estimate:
 * dims = {owned pixel slots, gather_size + 2, samples}; out[slot][row][sample], rows: photon indices (sorted order), their number,
 * the squared distance of the farthest (float bits).  out may be null (dims only). */
int rpt_debug_photon_selections(rpt_scene*, uint32_t* out, uint64_t capacity_words, uint64_t dims[3]);
int rpt_debug_camera_rays(const rpt_camera*, const rpt_render_params*, uint64_t seed, uint32_t sample,
                          float* origins, float* dirs); /* one ray per pixel, width*height*3 each */

#ifdef __cplusplus
}
#endif
#endif /* RPT_HIP_H */
