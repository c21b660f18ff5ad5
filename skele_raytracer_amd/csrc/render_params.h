// Kernel argument block shared by the launcher (api.cpp) and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct f3 {
	float x, y, z;
};

#define SKR_COUNTER_SHARDS 4096u
// Counters that thousands of waves hit with atomics sit SKR_PULL_STRIDE uint32 apart (one word sustains only ~88 atomics/us);
// SKR_PULL_QUEUES words of that kind are reserved behind the work counters (api.cpp).
#ifndef SKR_PULL_QUEUES
#define SKR_PULL_QUEUES 16u
#endif
#define SKR_PULL_STRIDE 256u
// the level pipelines append the hit records of a level to SKR_P1_REGIONS regions, one counter each
#define SKR_P1_REGIONS 64u

// The SKR_* development switches (A/B runs, tests), read from the environment ONCE per renderer (skr_renderer_create,
// skr_renderer_reload_switches) — the launch path never calls getenv.
enum SkrPipeline { SKR_PIPE_AUTO = 0, SKR_PIPE_NODES, SKR_PIPE_GENERIC, SKR_PIPE_OTHER };
struct SkrSwitches {
	int32_t pipeline = SKR_PIPE_AUTO; // SKR_PIPELINE = nodes | generic: which level pipeline takes a --gillum tree (tests, A/B runs)
	int32_t no_cones = 0, no_cull = 0; // SKR_NO_CONES, SKR_NO_CULL: triangle-walk culling off
	int32_t budget_mb = 0;            // SKR_LEVELS_BUDGET_MB: scratch budget of the level pipelines (0 = default)
	int32_t flat = 0;                 // SKR_FLAT = 1 | 0: the node pipeline's flat schedule forced on (+1) / off (-1); unset: by launch size
};

struct RenderParams {
	SkrSwitches sw; // (host side only)
	// image and partition (include/skr.h skr_render_tiles)
	int32_t width, height;
	uint32_t tile_rows, first_tile, tile_stride, out_rows;
	const uint32_t *tile_table; // device, or null: slot k of the compact output holds tile tile_table[k] (0xFFFFFFFF: an empty padding slot) instead of first_tile + k * tile_stride
	uint32_t band_row0, band_rows; // the band of output rows [band_row0, band_row0 + band_rows) a launch of the general level pipeline works on (the whole launch elsewhere)
	// per-frame invariants of main.cpp:134-137, computed once on the host
	float inv_width, inv_height, aspect, angle;
	// camera.h:8-32 (direction/up/right keep the file's magnitudes) and scene.h:24
	f3 cam_pos, cam_dir, cam_up, cam_right, background;
	// SoA scene in HBM (scene_host.h)
	int32_t n_spheres, n_tris, n_lights;
	const float4 *sph_geom, *sph_amb, *sph_kd, *sph_ks, *lights, *tris;
	const float4 *cam_ec;     // per sphere {cam_pos - centre, |cam_pos - centre|^2 - r^2}: e and c of utils.h:115-118 for every ray that starts at the camera (skr_camec_kernel)
	const float4 *tri_chunks; // the chunk tree of the triangle walk (scene_host.h): 3 float4 per node, depth-first, skip links, then 2 float4 per chunk
	int32_t tri_chunk_size;
	int32_t tri_cones;        // some entry has a tight radius for non-grazing rays (else the cone test is compiled out of the walk)
	int32_t n_tri_chunks;     // its node count; 0 = culling off (ray directions longer than the bounds were built for)
	// utils.h:26-34 Options + scene.use_shadows
	int32_t monte_carlo, num_path_traces, grid_size, max_depth, use_shadows;
	uint32_t seed_lo, seed_hi;
	int32_t pow_steps;        // bit length of the scene's largest integer phong exponent in [1, 1024] (device_math.h powf_spec): 1 .. 11
	// outputs (device)
	uint8_t *rgb;
	float *rgbf;
	unsigned long long *counters; // SKR_COUNTER_SHARDS x {radiance rays, sphere hits shaded, shadow rays, pad}
	unsigned long long *tri_work; // 256 x {culling-sphere tests, triangle tests} the triangle walks executed (shade_common.h tri_work_add); null: not counted
	uint32_t *qctr;     // [0] the number of level-0 nodes skr_primary_kernel appended
	float *acc;         // float3 per output pixel: the running `image[y][x] += shade(...)` of main.cpp:162 (AA under --gillum: one pass per sample)
	uint32_t aa_index;  // which AA sample this launch traces
	// node pipeline (render_nodes.hip): the --gillum tree cut at every level.  A node is a shaded sphere hit; level 0 = the
	// primary hits.  A node is two rows of two float4 in two arrays, so that every kernel reads only the half it needs:
	//   geometry (tracing its children): [co.xyz N.x] [N.yz pixel node-id]                       (node id 0 at level 0)
	//   shading  (summing them):         [direct.xyz sphere] [r1 record|output-pixel pixel node-id]   (level 0: output pixel; deeper: its own record)
	const float4 *nd_src;     // geometry rows of the nodes whose children are traced (trace, activate, leaf)
	const float4 *ns_src;     // shading rows of the nodes whose children are summed (finalize, the depth-2 leaf kernel)
	float4 *nd_dst, *ns_dst;  // nodes being written (primary hits; activated records)
	uint32_t nd_src_level0;   // nd_src / ns_src hold the primary hits
	const uint32_t *nd_count; // number of nodes in nd_src
	float4 *rc;               // hit records of the level being produced (trace) or consumed (activate, leaf): [parent, sphere | child << 16, r1, r2]
	uint32_t rc_cap;          // records per region (SKR_P1_REGIONS regions)
	uint32_t *rc_ctr;         // that level's counters: [STRIDE r] records in region r, [STRIDE (64 + r)] units handed out, [STRIDE 128] exhausted mask, [STRIDE 129 ..] prefix sums
	uint4 *ixh;               // per trace wave (64 sibling pairs of the nd_src nodes, pair = node * PP + j) three uint4: {first record of its hits, how many of them are even children's, -, -}, the ballots of the even / odd children that hit (64 bits each), the ballots of the children a triangle took (triangle scenes)
	uint32_t band_blk0, band_nblk, blocks_x; // node_layout: skr_primary_kernel covers the 16x16 pixel blocks [band_blk0, band_blk0 + band_nblk) of the launch (row-major, blocks_x per row)
	void *node_scratch;       // (host) the pipeline's one allocation
	const float *res_in;      // (colour r1)/pdf of every child record (finalize)
	float *res_out;           // the same for this level's records (leaf kernel; finalize of a level >= 1)
	// --shade-triangles (general level pipeline): triangles are surfaces, not black holes
	int32_t shade_triangles;
	int32_t legacy_reflect;   // --legacy-reflect (general level pipeline): raytrace.h:45-103 runs; sph_ks[i].w = the sphere's index of refraction
	const float4 *tri_mats;   // 3 float4 per triangle, in tris[] order: [La*ka, power] [kd] [ks] (the rows sph_amb / sph_kd / sph_ks hold for a sphere)
	// general level pipeline (render_generic.hip): one lane per ray, every mode, any depth
	uint32_t g_level;         // the level a launch works on (trace / activate: the rays' level, 1 = primary; finalize: the nodes' level, 0 = the camera)
	uint32_t g_arity;         // children per node of the level whose children are traced / summed (1 at the camera level); activate: the tree's arity (node ids)
	uint32_t g_last;          // activate: the hits of the last level (their children are shade(depth 0) == 0) are finished at once
	const float4 *g_nodes_src; // nodes of the level above (trace, activate) / of the level being summed (finalize): 5 float4 each
	float4 *g_nodes_dst;      // nodes being written (activate)
};

// Optional timing of the dominant kernel of a launch (skr_renderer_kernel_ms): the launcher records the
// two events right around that kernel on the launch stream — and, where `snap` is set, copies the work counters in front of the
// first event and behind the second (stream-ordered device-to-device copies outside the timed window), so that the work of that
// one kernel can be told from the frame's (skr_renderer_kernel_work: the numerator of bench.py's kernel-level roofline).
struct SkrTimingHook {
	hipEvent_t start = nullptr, stop = nullptr;
	unsigned long long *snap = nullptr;             // device: 2 x SKR_COUNTER_SHARDS x 4 words, or null
	const unsigned long long *counters = nullptr;   // device: RenderParams::counters
};
static inline void skr_hook_start(const SkrTimingHook *h, hipStream_t stream)
{
	if(!h) return;
	if(h->snap) (void) hipMemcpyAsync(h->snap, h->counters, (size_t) SKR_COUNTER_SHARDS * 4 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream);
	if(h->start) (void) hipEventRecord(h->start, stream);
}
static inline void skr_hook_stop(const SkrTimingHook *h, hipStream_t stream)
{
	if(!h) return;
	if(h->stop) (void) hipEventRecord(h->stop, stream);
	if(h->snap) (void) hipMemcpyAsync(h->snap + (size_t) SKR_COUNTER_SHARDS * 4, h->counters, (size_t) SKR_COUNTER_SHARDS * 4 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream);
}
