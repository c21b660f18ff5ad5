/* include/skr.h — C ABI of libskr.so, the MI355X-native replacement for the
 * reference's per-pixel hot path (lilinitsy/skele-raytracer).
 *
 * The reference has no plugin/FFI interface: its seam is the in-process call
 *     glm::vec3 shade(Ray, Scene, int depth, bool monte_carlo, short num_path_traces)
 *         src/raytrace.h:139, called once per pixel-sample from src/main.cpp:64,83,162,181
 * fed by   Scene parseScene(std::string)          src/scene.h:31, src/scene.cpp:12
 * and      struct Options                         src/utils.h:26-34
 * and drained by the inline P6 writer             src/main.cpp:199-211 (== :88-100).
 * This header lifts that seam to frame / row-tile granularity: plain pointers
 * and sizes, no C++ or torch types.  Every entry point names the reference
 * interface it replaces.  Return value 0 = ok; otherwise an skr_status (the
 * reference itself has no error returns: it prints and exits with status 0).
 *
 * Threading: a renderer is bound to one HIP device; calls on one renderer must
 * not overlap; different renderers are independent.  All device work is
 * enqueued on the caller's stream and is asynchronous unless stated.
 */
#ifndef SKR_H
#define SKR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKR_ABI_VERSION 3 /* 3: skr_options grew by shade_triangles, progressive_passes and legacy_reflect (56 bytes); 2: multi-GPU entry points, skr_scene_info.n_directional_lights */

typedef enum {
	SKR_OK = 0,
	SKR_ERR_IO = 1,            /* cannot open / write a file */
	SKR_ERR_ARG = 2,           /* bad argument (null, zero size, bad row range ...) */
	SKR_ERR_HIP = 3,           /* a HIP runtime call failed; see skr_last_error() */
	SKR_ERR_UNSUPPORTED = 4,   /* configuration outside what the kernels cover (see skr_last_error()) */
	SKR_ERR_NO_DEVICE = 5      /* no gfx950 device / HIP runtime unusable: there is NO CPU fallback */
} skr_status;

/* Host-side scene in the SoA layout that is uploaded to HBM.  Opaque. */
typedef struct skr_scene skr_scene;
/* Device context: one HIP device, the scene resident in HBM.  Opaque. */
typedef struct skr_renderer skr_renderer;

/* Mirrors struct Options (utils.h:26-34) plus the locals main() folds into the
 * scene before rendering (width/height/use_shadows, main.cpp:236-244,393-396).
 * skr_options_default() fills the reference's defaults. */
typedef struct {
	int32_t width;            /* --width,  default 1920 (scene.h:15) */
	int32_t height;           /* --height, default 1080 */
	float fov;                /* --fov degrees, default 60 (utils.h:30) */
	int32_t monte_carlo;      /* --gillum given (utils.h:28) */
	int32_t num_path_traces;  /* --gillum N, default 1 (utils.h:31, a short) */
	int32_t grid_size;        /* --jsample g, default 0 = pixel centres (utils.h:32) */
	int32_t max_depth;        /* --depth d > 0, default 3 (utils.h:33) */
	int32_t use_shadows;      /* --shadow (main.cpp:375-378); absent == 0 */
	uint64_t seed;            /* new: key of the counter RNG that replaces srand(time(0)) (main.cpp:400) */
	int32_t shade_triangles;  /* new, default 0 (`raytracer --shade-triangles`).  0 = HEAD: a ray whose closest hit is a
	                           * triangle returns black (raytrace.h:221-224).  1 = triangles are surfaces: among the triangles
	                           * utils.h:181-213 accepts with t > 0 (the one the ray starts on excepted) the smallest t wins if it is
	                           * strictly below the closest sphere's (equal t: lower index in the file); the hit is shaded as a
	                           * sphere is (blinn_phong.h, raytrace.h:107-136,208-218) with the material in force on its
	                           * `triangle` line and the geometric normal normalize(cross(v1-v0, v2-v0)) turned against the ray;
	                           * shadow rays still test spheres only (utils.h:42-76); the child rays of a triangle hit start at
	                           * P + 1e-5 like a sphere's (raytrace.h:128).  No counterpart in the reference, so no reference output
	                           * pins it (tests/test_shade_triangles.py).  Any --depth, with or without --gillum (the general level pipeline). */
	int32_t progressive_passes; /* new, default 1 (`raytracer --progressive K`, SURVEY.md 8f-4: what the SDL viewer of main.cpp:183-197 is
	                           * for, headless).  K > 1: every render entry point traces K whole frames under the seeds seed, seed+1, ...,
	                           * seed+K-1, sums them in binary32 in that order, divides by (float) K once and quantises the mean like
	                           * a single frame (main.cpp:205).  K <= 1 is the single frame, bit for bit. */
	int32_t legacy_reflect;   /* new, default 0 (`raytracer --legacy-reflect`).  1 = the code behind the early
	                           * `return total_colour;` of raytrace.h:44 runs: the Fresnel term fr (blinn_phong.h:156-184) and, where the
	                           * material's specular colour is not (0,0,0), for every light (point lights first) a refraction ray
	                           * (:143-153; `refraction_colour = fr * shade(...)`: the last light's stays) and a reflection ray (:137-140:
	                           * the LIGHT direction mirrored at the normal; `+= (1 - fr) * specular * shade(...)`), both from the hit point
	                           * itself with depth - 1, added to the direct term as raytrace.h:102 does.  The index of refraction is the
	                           * 14th number of the `material` line (skr_scene_set_sphere_ior for scenes from arrays; default 1).
	                           * Children of the counter RNG's tree: arity N + 2 L (L lights); child N + 2 l = refraction of light l,
	                           * N + 2 l + 1 = its reflection.  Unreachable at HEAD, so no output of the reference's program pins it: the
	                           * reference's own fresnel / refraction / reflect_direction functions pin the three formulas, a restated
	                           * composition the rest, and the README pictures (made when it ran) are the visual check
	                           * (tests/test_legacy_reflect.py).  Any --depth, with --gillum and with shade_triangles (the general level pipeline). */
} skr_options;

typedef struct {
	int32_t n_spheres, n_triangles, n_point_lights, n_vertices;
	int32_t n_directional_dropped; /* parsed and never pushed, scene.cpp:139-163 (0 under SKR_SCN_STRICT) */
	int32_t n_fog_skipped;         /* spherical_fog lines (UB in the reference, scene.cpp:207-212): warned + skipped */
	int32_t n_unknown;             /* "WARNING. Do not know command" lines, scene.cpp:214-217 */
	int32_t n_bad_triangles;       /* triangle lines whose indices fall outside the vertex pool (skipped) */
	int32_t film_width, film_height; /* film_resolution: parsed, overridden by the CLI (main.cpp:393-395) */
	int32_t max_depth_parsed;      /* max_depth: parsed, never read (scene.cpp:192-198) */
	float camera[13];              /* position, direction, up, right (camera.h:30), half_height_angle */
	float background[3];
	float ambient[3];
	int32_t n_directional_lights;  /* SKR_SCN_STRICT: directional lights kept (after the point lights in shading order) */
} skr_scene_info;

/* ---- scene: replaces Scene parseScene(std::string) (scene.cpp:12-227) ---- */
/* echo != 0 prints the reference's per-line echo to stdout (scene.cpp:50,...). */
int skr_scene_create_from_scn(const char *path, int echo, skr_scene **out);
/* flags: SKR_SCN_STRICT = the loader as its author evidently meant it (`raytracer --strict-scn`):
 * directional lights are pushed (scene.cpp:139-163 builds each one, clamps its colour to <= 1 and forgets the push_back)
 * and shaded by the reference's own loops (blinn_phong.h:77-85,122-131; shadow test utils.h:60-76).  film_resolution and
 * max_depth are reported in skr_scene_info either way; --strict-scn makes the CLI honour them. */
#define SKR_SCN_STRICT 1u
int skr_scene_create_from_scn_ex(const char *path, int echo, uint32_t flags, skr_scene **out);
/* Build a scene from arrays (synthetic tests): spheres[n][14] = centre(3) radius
 * ambient(3) diffuse(3) specular(3) power; triangles[n][9] = v0 v1 v2;
 * point_lights[n][6] = position colour; camera[9] = position direction up. */
int skr_scene_create_from_arrays(const float *spheres, int32_t n_spheres, const float *triangles, int32_t n_triangles,
								 const float *point_lights, int32_t n_point_lights, const float camera[9],
								 const float background[3], const float ambient[3], skr_scene **out);
void skr_scene_destroy(skr_scene *scene);
int skr_scene_get_info(const skr_scene *scene, skr_scene_info *info);
/* Copy the parsed arrays back out in the skr_scene_create_from_arrays layouts
 * (any pointer may be NULL).  Used by the loader parity tests. */
int skr_scene_get_arrays(const skr_scene *scene, float *spheres, float *triangles, float *point_lights);
/* The culling data the triangle walk of raytrace.h:171-186 runs on, as uploaded:
 * device_tris[n_triangles][12] = {v0, 0, v1-v0, file index (int32 bits), v2-v0, 0} in device (Morton) order; chunk_spheres[n_chunks][8]
 * = {centre, R^2, axis / kappa, R_tight^2} of every *chunk_size consecutive device triangles (R_tight applies to
 * rays with (d . axis / kappa)^2 >= d . d; axis = 0 and R_tight = R where there is none); above them a tree in
 * depth-first order, node_spheres[n_nodes][8] likewise and node_links[n_nodes][4] = {skip (index of the next node
 * that is not below this one), first chunk, chunk count (height-1 nodes only, else 0), height}.  level 0..2
 * selects the R built for ray directions up to 4 / 32 / 256 long (the launcher picks by camera and --fov).  Any
 * pointer may be NULL; the counts are returned first so the caller can size the arrays.  Used by the host-logic
 * tests. */
int skr_scene_get_culling(const skr_scene *scene, int32_t level, int32_t *chunk_size, int32_t *n_nodes, int32_t *n_chunks,
						  float *device_tris, float *node_spheres, int32_t *node_links, float *chunk_spheres);

/* Materials of the triangles of a scene built from arrays, materials[n_triangles][10] = ambient(3) diffuse(3) specular(3)
 * phong power — what the `material` line in force gives a `triangle` line in a .scn file (scene.cpp:110-137; the reference
 * keeps no material for a triangle, shapes.h:26).  Read by skr_options.shade_triangles only; default: material.h:9-17's. */
int skr_scene_set_triangle_materials(skr_scene *scene, const float *materials);

/* Index of refraction of every sphere of a scene built from arrays, ior[n_spheres] (material.h:16; the 14th number of a
 * `material` line, scene.cpp:110-137).  Read by skr_options.legacy_reflect only; default 1. */
int skr_scene_set_sphere_ior(skr_scene *scene, const float *ior);

/* ---- options: replaces Options' in-class defaults (utils.h:28-33) ---- */
void skr_options_default(skr_options *opt);
/* R = W*H*max(1,g*g)*sum_{k<depth} N^k: number of shade() calls with depth > 0
 * (SURVEY.md §8d); the metric's numerator. */
uint64_t skr_radiance_ray_count(const skr_options *opt);

/* ---- device ---- */
int skr_device_count(void);
/* Uploads the SoA scene to HBM of `device`. */
int skr_renderer_create(const skr_scene *scene, int device, skr_renderer **out);
/* A second renderer for the same scene on the same device with its own scratch tables: what a second frame in flight needs (the
 * pipelined frame steps make one themselves).  It shares the uploaded scene and the work counters with `src`, which must outlive it. */
int skr_renderer_clone(const skr_renderer *src, skr_renderer **out);
void skr_renderer_destroy(skr_renderer *r);

/* The hot path: replaces the loop nest main.cpp:125-197 (== :36-85) and every
 * shade() call under it, for image rows owned by this caller.
 *
 * Rows are grouped in row tiles of `tile_rows` rows; this call renders tiles
 * first_tile, first_tile + tile_stride, ... (the interleaved partition used to
 * shard a frame over GPUs: rank r of G passes first_tile = r, tile_stride = G).
 * Output is compact and tile-major: the k-th rendered tile occupies rows
 * [k*tile_rows, (k+1)*tile_rows) of d_rgb (W*3 bytes per row, quantised exactly
 * like main.cpp:205: (unsigned char)(std::min(1.0f, c) * 255)); rows of a final
 * partial tile beyond the image are left untouched.  d_rgbf, if not NULL,
 * receives the unquantised float3 image in the same layout (tests).
 * Both are DEVICE pointers.  stream is a hipStream_t (NULL = default stream).
 * Random numbers are keyed by the global pixel index, so the image does not
 * depend on the partition. */
int skr_render_tiles(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint32_t first_tile,
					 uint32_t tile_stride, uint8_t *d_rgb, float *d_rgbf, void *stream);
/* The same for an explicit list of tiles: slot k of the compact output (rows k*tile_rows ..) holds tile d_tiles[k] — a DEVICE array
 * of n_slots tile indices, 0xFFFFFFFF = an empty slot (its rows are left untouched).  What the multi-GPU frame steps render with
 * once the tiles are dealt by cost instead of by `t mod G`. */
int skr_render_tile_list(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, const uint32_t *d_tiles, uint32_t n_slots,
						 uint8_t *d_rgb, float *d_rgbf, void *stream);
/* Per tile of tile_rows image rows, the work it costs, counted: the tile is rendered on its own and its rays, shaded hits and
 * ray-sphere tests (skr_renderer_read_work) are priced in flops as bench.py prices a frame.  The numbers behind the multi-GPU tile
 * map; integer counts of a bit-reproducible render, so every rank of a job gets the same ones.  Synchronous (one small render per
 * tile); the renderer's work counters are preserved.  h_cost has ceil(height / tile_rows) entries. */
int skr_tile_costs(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint64_t *h_cost);
/* Number of tiles / output rows skr_render_tiles will produce for this partition. */
uint32_t skr_tile_count(const skr_options *opt, uint32_t tile_rows, uint32_t first_tile, uint32_t tile_stride);
/* Contiguous rows [y0, y1) (one tile of y1-y0 rows starting at y0). */
int skr_render_rows(skr_renderer *r, const skr_options *opt, uint32_t y0, uint32_t y1, uint8_t *d_rgb, float *d_rgbf,
					void *stream);
/* The two steps of the progressive mean, for callers that want to look at it while it forms (`raytracer --progressive K
 * --progressive-every M`): d_acc = d_frame (first != 0) or d_acc + d_frame over n_floats binary32 values; then
 * d_rgbf = d_acc / (float) passes and d_rgb = its quantised bytes for a whole width x height frame.  DEVICE pointers. */
int skr_accumulate(float *d_acc, const float *d_frame, uint64_t n_floats, int first, void *stream);
int skr_resolve_accumulated(const float *d_acc, uint32_t passes, uint32_t width, uint32_t height, uint8_t *d_rgb, float *d_rgbf, void *stream);
/* Work counters accumulated by the kernels since the last reset (synchronous):
 * out[0] radiance rays = shade() calls with depth > 0, out[1] sphere hits shaded,
 * out[2] shadow rays (one per light per hit; the reference casts each twice). */
int skr_renderer_read_counters(skr_renderer *r, uint64_t out[3], int reset);
/* The same plus out[3] = ray-sphere tests as the reference runs them: every sphere for a radiance ray
 * (raytrace.h:152-165), up to and including the first occluder for a shadow ray (utils.h:52-55).  The
 * numerator of bench.py's FP32-VALU roofline; asserted equal to the oracle's count. */
int skr_renderer_read_work(skr_renderer *r, uint64_t out[4], int reset);
/* The work (as skr_renderer_read_work counts it) of the ONE kernel skr_renderer_kernel_ms times, in the last launch made with kernel
 * timing on: the counters are copied on the launch stream in front of and behind that kernel, outside the timed window.  Do not
 * reset the counters between that launch and this call.  Synchronous. */
int skr_renderer_kernel_work(skr_renderer *r, uint64_t out[4]);
/* The triangle walks count what they execute only while this is on (off at creation): counting costs the dragon walk ~19 %, so a
 * measurement takes its counts from frames rendered for that purpose and its times from frames rendered without. */
int skr_renderer_count_triangle_work(skr_renderer *r, int enable);
/* What the triangle walks did while counting since the last reset (synchronous; read it BEFORE resetting the counters above): out[0] culling-sphere
 * tests and out[1] ray-triangle tests (utils.h:181-213) the kernels executed — lanes that needed the test, counted by the walks
 * themselves —, out[2] the ray-triangle tests the reference's loop runs for the same rays (raytrace.h:171-186: every triangle for
 * every radiance ray).  bench.py prices mesh scenes with out[0] and out[1]; out[2] / out[1] is what the exact culling saves. */
int skr_renderer_read_triangle_work(skr_renderer *r, uint64_t out[3], int reset);
/* The SKR_* development switches (kernel variant, budgets: DESIGN.md) are read from the environment
 * once, at skr_renderer_create; this reads them again (tests and A/B tools change them between frames). */
int skr_renderer_reload_switches(skr_renderer *r);
/* Time the dominant kernel of each launch with HIP events recorded on the launch stream (off by
 * default).  skr_renderer_kernel_ms() waits for the launches made since the last call and returns
 * their mean duration in ms and their number: what bench.py's roofline is computed from. */
int skr_renderer_kernel_timing(skr_renderer *r, int enable);
int skr_renderer_kernel_ms(skr_renderer *r, float *mean_ms, int32_t *launches);
/* Number of primary sphere hits the last launch (its last band) queued as level-0 nodes / parents for the
 * --gillum kernels (0 if that launch had no --gillum tree); synchronous. */
int skr_renderer_last_parent_count(skr_renderer *r, uint32_t *n);
/* Number of level-1 sphere hit records the last launch (its last band) queued (node pipeline;
 * 0 for the other kernel variants and at depth 2); synchronous. */
int skr_renderer_last_level1_count(skr_renderer *r, uint32_t *n);
/* Whole frame into HOST memory (W*H*3 bytes), synchronous; what the CLI uses. */
int skr_render_frame_host(skr_renderer *r, const skr_options *opt, uint8_t *h_rgb, float *kernel_ms);
/* The same with the float frame as a second output (h_rgb or h_rgbf may be NULL, not both) and — under
 * opt->progressive_passes = K > 1 — a look at the mean while it forms: after every `every` passes and after the last one the
 * mean so far is resolved into the host buffers and `progress(user, passes_done, K, h_rgb, h_rgbf)` is called (the place where
 * the SDL viewer of main.cpp:183-197 blits; a non-zero return stops the render with that mean in the buffers).  every == 0 or
 * progress == NULL: one call at the end / none.  The final buffers never depend on `every`.  kernel_ms: device time, summed. */
typedef int (*skr_progress_fn)(void *user, uint32_t passes_done, uint32_t passes, const uint8_t *h_rgb, const float *h_rgbf);
int skr_render_progressive_host(skr_renderer *r, const skr_options *opt, uint32_t every, uint8_t *h_rgb, float *h_rgbf,
								skr_progress_fn progress, void *user, float *kernel_ms);

/* ---- multi-GPU: the frame sharded over the GPUs of one node ----
 * Replaces the reference's only parallel entry, `generate_rays_parallel` (main.cpp:19-104: `#pragma omp parallel for`
 * over the rows at :33, dispatched at main.cpp:402-410) — the reference has no distributed path.
 * The framebuffer is cut into tiles of tile_rows rows and the tiles are dealt to the ranks: tile t to rank t mod G, unless the counted
 * work of the tiles (skr_tile_costs) says that leaves the heaviest rank above 1.10 of the mean — then longest-processing-time-first
 * over those costs (skr_shard_plan; SKR_SHARD=interleave / lpt in the environment forces one or the other).  Every rank renders its
 * tiles (skr_render_tile_list) straight
 * into its slot of a gather buffer, ONE RCCL all-gather over xGMI brings the slots together and rank 0 de-interleaves on the device.
 * The image does not depend on G or on the map (the RNG is keyed by the global pixel).  RCCL is bound at run time;
 * skr_rccl_available() says whether it could be. */
int skr_rccl_available(void);
/* (a) ONE process, N devices (ncclCommInitAll; a renderer, a stream and a worker thread per device).
 * devices == NULL: devices 0 .. n_devices-1.  What `raytracer --gpus N` uses. */
typedef struct skr_multi skr_multi;
int skr_multi_create(const skr_scene *scene, int n_devices, const int *devices, skr_multi **out);
void skr_multi_destroy(skr_multi *m);
int skr_multi_device_count(const skr_multi *m);
skr_renderer *skr_multi_renderer(skr_multi *m, int i); /* device i's renderer (counters, timing); owned by m */
/* Synchronous.  *d_frame: the W*H*3 frame in device 0's memory (owned by m, valid until the next call);
 * frame_ms: first launch to de-interleaved frame, on the root's stream. */
int skr_multi_render_frame(skr_multi *m, const skr_options *opt, uint32_t tile_rows, uint8_t **d_frame, float *frame_ms);
int skr_multi_render_frame_host(skr_multi *m, const skr_options *opt, uint32_t tile_rows, uint8_t *h_rgb, float *frame_ms);
/* The pipelined form (throughput of a run of frames; what skr_comm_render_frame_async is to shape (b)): frame f's all-gather and
 * de-interleave go to a second stream per device while the render streams take frame f + 1 into the other of two buffer sets.
 * Returns once frame f is enqueued; *d_prev_frame = the PREVIOUS call's frame, complete (NULL on the first call; pass NULL not to
 * wait for it).  skr_multi_flush waits for everything in flight; *d_frame = the last frame. */
int skr_multi_render_frame_async(skr_multi *m, const skr_options *opt, uint32_t tile_rows, uint8_t **d_prev_frame);
int skr_multi_flush(skr_multi *m, uint8_t **d_frame);
/* (b) one process PER device (torchrun, mpirun): rank 0 makes an id, the caller broadcasts it by whatever transport
 * it has, every rank creates its communicator on its renderer's device.  id == NULL with world == 1: no RCCL at all. */
#define SKR_COMM_ID_BYTES 128
typedef struct skr_comm skr_comm;
int skr_comm_unique_id(uint8_t id[SKR_COMM_ID_BYTES]);
int skr_comm_create(skr_renderer *r, int device, const uint8_t id[SKR_COMM_ID_BYTES], int rank, int world, skr_comm **out);
void skr_comm_destroy(skr_comm *c);
/* Asynchronous on `stream`: this rank's tiles, the all-gather, rank 0's de-interleave.  *d_frame: rank 0's finished
 * frame (device memory owned by c; NULL on the other ranks) once the stream has drained. */
int skr_comm_render_frame(skr_comm *c, const skr_options *opt, uint32_t tile_rows, uint8_t **d_frame, void *stream);
/* The frame step with its collective off the render stream (throughput of a run of frames): frame f's all-gather and
 * de-interleave run on a stream the communicator owns while `stream` renders frame f + 1 into the other of two buffer sets.
 * *d_prev_frame (rank 0; NULL elsewhere and on the first call): the PREVIOUS call's frame, complete in `stream` order after
 * this call (pass d_prev_frame = NULL not to look at it: one stream wait less per frame).  skr_comm_flush ends the run: `stream` waits for the last collective, *d_frame = the last frame.  Every frame is
 * the one skr_comm_render_frame would have produced. */
int skr_comm_render_frame_async(skr_comm *c, const skr_options *opt, uint32_t tile_rows, uint8_t **d_prev_frame, void *stream);
int skr_comm_flush(skr_comm *c, uint8_t **d_frame, void *stream);
/* Rank 0: waits for `stream` and copies that frame to host memory (W*H*3 bytes). */
int skr_comm_frame_to_host(skr_comm *c, uint8_t *h_rgb, void *stream);
/* The partition itself (host logic, no GPU): padded tiles per rank, and the de-interleave of a rank-major gathered
 * buffer [world][tiles_per_rank * tile_rows][W*3] into frame[H][W*3]. */
uint32_t skr_shard_tiles_per_rank(int32_t height, uint32_t tile_rows, uint32_t world);
int skr_shard_deinterleave_host(const uint8_t *gathered, uint8_t *frame, int32_t width, int32_t height, uint32_t tile_rows, uint32_t world);
/* Maps: slot_of_tile[t] = rank * k_max + slot (k_max = skr_shard_tiles_per_rank) for n_tiles tiles of cost[t] each.
 * skr_shard_lpt: most expensive tile first, each to the least loaded rank with a free slot; a rank's tiles sit in its slots in image
 * order; deterministic.  skr_shard_by_cost: the rule of the frame steps — tile t to rank t mod world unless that leaves the heaviest
 * rank above 1.10 of the mean cost, then skr_shard_lpt.  skr_shard_plan: the map a frame step of `world` ranks uses for this renderer
 * and these options (skr_tile_costs + the rule).  skr_shard_deinterleave_map_host: the de-interleave under such a map. */
int skr_shard_lpt(const uint64_t *cost, uint32_t n_tiles, uint32_t world, uint32_t *slot_of_tile);
int skr_shard_by_cost(const uint64_t *cost, uint32_t n_tiles, uint32_t world, uint32_t *slot_of_tile);
int skr_shard_plan(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint32_t world, uint32_t *slot_of_tile);
int skr_shard_deinterleave_map_host(const uint8_t *gathered, uint8_t *frame, int32_t width, int32_t height, uint32_t tile_rows, const uint32_t *slot_of_tile);

/* ---- image file: replaces the inline writer main.cpp:199-211 ---- */
int skr_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb);
/* Other outputs (new; `raytracer --format ppm|png|pfm`).  PNG: the same bytes as the PPM, 8-bit RGB (stored
 * deflate blocks: no compressor is linked).  PFM: the unquantised float frame, "PF\nW H\n-1.0\n" + binary32 RGB, bottom row
 * first; rgbf is top row first like every buffer of this interface. */
int skr_write_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb);
int skr_write_pfm(const char *path, uint32_t width, uint32_t height, const float *rgbf);

/* ---- diagnostics ---- */
const char *skr_last_error(void);      /* thread-local text of the last failure */
const char *skr_kernel_variant(void);  /* name of the kernel the last render launched */
/* Device-side evaluation of the arithmetic spec for unit tests: op selects
 * 0 philox(ctr4,key2 -> out4 u32), 1 sincos(phi -> s,c), 2 powf(x,p),
 * 3 smallest_root(a,b,c), 4 triangle test (o,d,v0,v1,v2 -> hit,t),
 * 5 quantise(c -> u8 as u32), 6 basis(n -> nt,nb), 7 (a,b -> sqrtf(a), a/b), 8 philox at 10 rounds (op 0: the 7 the draws use),
 * 9 (hi16 -> mismatch counts of the short exact sqrt, 1/x, x/pi, x/pdf forms against the correctly rounded expansions over the
 * 65536 binary32 values with those high 16 bits).  in/out are DEVICE pointers
 * to n records of the op's input/output width in 32-bit words. */
int skr_debug_eval(int op, const void *d_in, void *d_out, uint32_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SKR_H */
