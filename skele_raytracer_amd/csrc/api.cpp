// C ABI glue: device context, scene upload, launches (include/skr.h).
// There is deliberately no CPU fallback: without a usable HIP device every
// device entry point fails with SKR_ERR_NO_DEVICE / SKR_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

#include "render_params.h"
#include "scene_host.h"

hipError_t skr_launch_render(const RenderParams &p, hipStream_t stream, const char **variant, const SkrTimingHook *hook);
hipError_t skr_launch_debug(int op, const void *d_in, void *d_out, uint32_t n, hipStream_t stream);
// accumulate.hip
hipError_t skr_launch_accumulate(float *acc, const float *frame, size_t n, int first, hipStream_t stream);
hipError_t skr_launch_resolve_accumulated(const float *acc, uint32_t passes, uint32_t width, uint32_t out_rows, uint32_t height, uint32_t tile_rows,
										  uint32_t first_tile, uint32_t tile_stride, const uint32_t *tile_table, uint8_t *rgb, float *rgbf, hipStream_t stream);
size_t skr_render_lds_bytes(const RenderParams &p);
bool skr_nodes_selected(const RenderParams &p);
size_t skr_nodes_scratch_bytes(const RenderParams &p);
bool skr_generic_selected(const RenderParams &p);   // render_kernel.hip
bool skr_generic_supported(const RenderParams &p);  // render_generic.hip
size_t skr_generic_scratch_bytes(const RenderParams &p);
bool skr_nodes_counter_layout(const RenderParams &p, size_t *off_ctr, size_t *level_words, int *levels);

static thread_local const char *g_variant = "none";

#define SKR_HIP(call)                                                                                      \
	do                                                                                                     \
	{                                                                                                      \
		hipError_t e_ = (call);                                                                            \
		if(e_ != hipSuccess)                                                                               \
		{                                                                                                  \
			skr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);     \
			return SKR_ERR_HIP;                                                                            \
		}                                                                                                  \
	} while(0)

struct skr_renderer {
	int device = 0;
	skr_scene_info info{};
	float4 *d_blob = nullptr; // one allocation: geom | amb | kd | ks | lights | tris | chunk trees | triangle materials
	size_t blob_bytes = 0;
	float4 *d_camec = nullptr; // per sphere {cam_pos - centre, |.|^2 - r^2} (render_wave.hip skr_camec_kernel), + 16 rows of padding
	bool is_clone = false;    // skr_renderer_clone: the scene blob and the work counters belong to the renderer it was cloned from
	size_t off_amb = 0, off_kd = 0, off_ks = 0, off_lights = 0, off_tris = 0, off_chunks = 0, off_tri_mats = 0;
	int n_chunks = 0, chunk_size = 0, cones = 0;
	size_t chunk_stride = 0;
	unsigned long long *d_counters = nullptr;
	unsigned long long *d_snap = nullptr; // skr_renderer_kernel_work: the work counters in front of and behind the dominant kernel of the last timed launch
	unsigned long long *d_tri_work = nullptr; // 256 x {culling-sphere tests, triangle tests} executed by the triangle walks (skr_renderer_read_triangle_work)
	int lds_limit = 0;
	int pow_steps = 11; // bit length of the largest integer phong exponent in [1, 1024] among the scene's materials (device_math.h powf_spec)
	// scratch, grown on demand and kept
	void *d_nodes = nullptr;  // node pipeline: every table of one band (render_nodes.hip NodePlan)
	size_t nodes_cap = 0;
	float *d_acc = nullptr;
	size_t acc_cap = 0;
	float *d_prog = nullptr; // progressive accumulation: this pass's float frame | the running sum
	size_t prog_cap = 0;
	uint8_t *d_frame = nullptr; // skr_render_frame_host
	size_t frame_cap = 0;
	hipEvent_t frame_e0 = nullptr, frame_e1 = nullptr;
	// skr_renderer_kernel_ms: event pairs around the dominant kernel of recent launches
	SkrSwitches sw; // the SKR_* development switches, read once (load_switches)
	RenderParams last_p{}; // the launch last enqueued (skr_renderer_last_*_count)
	bool last_nodes = false;
	bool timing = false;
	bool count_tri = false; // skr_renderer_count_triangle_work
	std::vector<SkrTimingHook> timed;
	std::vector<SkrTimingHook> free_pairs;
};

static void load_switches(SkrSwitches &sw)
{
	sw = SkrSwitches();
	if(const char *e = getenv("SKR_PIPELINE"))
		sw.pipeline = !strcmp(e, "nodes") ? SKR_PIPE_NODES : !strcmp(e, "generic") ? SKR_PIPE_GENERIC : SKR_PIPE_OTHER;
	sw.no_cones = getenv("SKR_NO_CONES") != nullptr;
	sw.no_cull = getenv("SKR_NO_CULL") != nullptr;
	if(const char *e = getenv("SKR_LEVELS_BUDGET_MB")) sw.budget_mb = atoi(e) > 0 ? atoi(e) : 1;
	if(const char *e = getenv("SKR_FLAT")) sw.flat = atoi(e) > 0 ? 1 : -1;
}

hipError_t skr_launch_camec(const float4 *geom, int ns, f3 cam_pos, float4 *out, hipStream_t stream); // render_wave.hip

// (multi_gpu.cpp) a clone follows its source's development switches: tests change them between frames
void skr_copy_switches(skr_renderer *dst, const skr_renderer *src) { dst->sw = src->sw; }

extern "C" {

int skr_device_count(void)
{
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int skr_renderer_create(const skr_scene *scene, int device, skr_renderer **out)
{
	if(!scene || !out)
	{
		skr_set_error("skr_renderer_create: null argument");
		return SKR_ERR_ARG;
	}
	*out = nullptr;
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
	{
		skr_set_error("no usable HIP device (count=%d, asked for %d); libskr has no CPU fallback", n, device);
		return SKR_ERR_NO_DEVICE;
	}
	SKR_HIP(hipSetDevice(device));
	hipDeviceProp_t prop;
	SKR_HIP(hipGetDeviceProperties(&prop, device));
	if(strncmp(prop.gcnArchName, "gfx950", 6) != 0)
	{
		skr_set_error("device %d is %s; libskr is built for gfx950 (MI355X) only", device, prop.gcnArchName);
		return SKR_ERR_NO_DEVICE;
	}
	skr_renderer *r = new skr_renderer();
	load_switches(r->sw);
	r->device = device;
	r->info = scene->info;
	r->lds_limit = (int) prop.sharedMemPerBlock;
	const size_t ns = scene->sph_geom.size(), nl2 = scene->lights.size(), nt3 = scene->tris.size();
	r->off_amb = ns;
	r->off_kd = 2 * ns;
	r->off_ks = 3 * ns;
	r->off_lights = 4 * ns;
	r->off_tris = 4 * ns + nl2;
	const size_t nch = scene->tri_chunks.size();
	r->off_chunks = 4 * ns + nl2 + nt3;
	r->chunk_size = scene->tri_chunk_size;
	r->chunk_stride = scene->tri_chunk_stride;
	r->cones = scene->tri_any_cone ? 1 : 0;
	r->n_chunks = scene->info.n_triangles ? scene->tri_node_count : 0; // nodes of the chunk tree (scene_host.h)
	const size_t ntm = scene->tri_mats.size();
	r->off_tri_mats = 4 * ns + nl2 + nt3 + nch;
	const size_t total = 4 * ns + nl2 + nt3 + nch + ntm;
	std::vector<skr_f4> blob(total + 16); // (+ 16 rows: the sphere loops ask for the rows of a trip ahead without a bounds test, shade_common.h sphere_rows)
	if(ns)
	{
		memcpy(&blob[0], scene->sph_geom.data(), ns * 16);
		memcpy(&blob[r->off_amb], scene->sph_amb.data(), ns * 16);
		memcpy(&blob[r->off_kd], scene->sph_kd.data(), ns * 16);
		memcpy(&blob[r->off_ks], scene->sph_ks.data(), ns * 16);
	}
	{ // the straight-line pow runs as many squarings as the scene's largest integer exponent has bits
		float top = 1.0f;
		auto look = [&](float pw) { if(pw >= 1.0f && pw <= 1024.0f && pw == rintf(pw) && pw > top) top = pw; };
		for(size_t i = 0; i < ns; i++) look(scene->sph_amb[i].w);
		for(size_t i = 0; i + 2 < scene->tri_mats.size(); i += 3) look(scene->tri_mats[i].w);
		int bits = 0;
		for(unsigned v = (unsigned) top; v; v >>= 1) bits++;
		r->pow_steps = bits < 1 ? 1 : bits;
	}
	if(nl2) memcpy(&blob[r->off_lights], scene->lights.data(), nl2 * 16);
	if(nt3) memcpy(&blob[r->off_tris], scene->tris.data(), nt3 * 16);
	if(nch) memcpy(&blob[r->off_chunks], scene->tri_chunks.data(), nch * 16);
	if(ntm) memcpy(&blob[r->off_tri_mats], scene->tri_mats.data(), ntm * 16);
	r->blob_bytes = blob.size() * 16;
	hipError_t e = hipMalloc((void **) &r->d_blob, blob.size() * 16);
	if(e == hipSuccess) e = hipMemcpy(r->d_blob, blob.data(), blob.size() * 16, hipMemcpyHostToDevice);
	if(e == hipSuccess) e = hipMalloc((void **) &r->d_counters, (SKR_COUNTER_SHARDS * 4 + 16) * sizeof(unsigned long long) + (SKR_PULL_QUEUES + 2 + 2 * SKR_P1_REGIONS) * SKR_PULL_STRIDE * sizeof(uint32_t));
	if(e == hipSuccess) e = hipMemset(r->d_counters, 0, (SKR_COUNTER_SHARDS * 4 + 16) * sizeof(unsigned long long) + (SKR_PULL_QUEUES + 2 + 2 * SKR_P1_REGIONS) * SKR_PULL_STRIDE * sizeof(uint32_t));
	if(e == hipSuccess) e = hipMalloc((void **) &r->d_tri_work, 256 * 2 * sizeof(unsigned long long));
	if(e == hipSuccess) e = hipMemset(r->d_tri_work, 0, 256 * 2 * sizeof(unsigned long long));
	if(e == hipSuccess) e = hipMalloc((void **) &r->d_camec, (ns + 16) * 16);
	if(e == hipSuccess) e = hipMemset(r->d_camec, 0, (ns + 16) * 16);
	if(e == hipSuccess)
	{ // the camera belongs to the scene: its (e, c) rows are formed once, on the device, by the operations every ray would perform
		const float *c = scene->info.camera;
		e = skr_launch_camec(r->d_blob, (int) ns, f3{c[0], c[1], c[2]}, r->d_camec, nullptr);
		if(e == hipSuccess) e = hipDeviceSynchronize();
	}
	if(e != hipSuccess)
	{
		skr_set_error("scene upload failed: %s", hipGetErrorString(e));
		if(r->d_camec) (void) hipFree(r->d_camec);
		if(r->d_blob) (void) hipFree(r->d_blob);
		if(r->d_counters) (void) hipFree(r->d_counters);
		if(r->d_tri_work) (void) hipFree(r->d_tri_work);
		delete r;
		return SKR_ERR_HIP;
	}
	*out = r;
	return SKR_OK;
}

// A second renderer for the same scene on the same device with its own scratch (tables, accumulation buffers, timing events) — what a
// second frame in flight needs (multi_gpu.cpp: consecutive frames of a run alternate between a renderer and its clone on two streams).
// The scene blob (read-only) and the work counters (atomics) are SHARED with `src`, which must outlive the clone: rays counted by
// either are read through either.
int skr_renderer_clone(const skr_renderer *src, skr_renderer **out)
{
	if(!src || !out)
	{
		skr_set_error("skr_renderer_clone: null argument");
		return SKR_ERR_ARG;
	}
	*out = nullptr;
	SKR_HIP(hipSetDevice(src->device));
	skr_renderer *r = new skr_renderer();
	r->device = src->device;
	r->info = src->info;
	r->d_blob = src->d_blob;
	r->blob_bytes = src->blob_bytes;
	r->d_camec = src->d_camec;
	r->is_clone = true;
	r->off_amb = src->off_amb; r->off_kd = src->off_kd; r->off_ks = src->off_ks; r->off_lights = src->off_lights;
	r->off_tris = src->off_tris; r->off_chunks = src->off_chunks; r->off_tri_mats = src->off_tri_mats;
	r->n_chunks = src->n_chunks; r->chunk_size = src->chunk_size; r->cones = src->cones; r->chunk_stride = src->chunk_stride;
	r->d_counters = src->d_counters;
	r->d_tri_work = src->d_tri_work;
	r->lds_limit = src->lds_limit;
	r->pow_steps = src->pow_steps;
	r->sw = src->sw;
	*out = r;
	return SKR_OK;
}

void skr_renderer_destroy(skr_renderer *r)
{
	if(!r) return;
	(void) hipSetDevice(r->device);
	if(r->d_blob && !r->is_clone) (void) hipFree(r->d_blob);
	if(r->d_camec && !r->is_clone) (void) hipFree(r->d_camec);
	if(r->d_counters && !r->is_clone) (void) hipFree(r->d_counters);
	if(r->d_tri_work && !r->is_clone) (void) hipFree(r->d_tri_work);
	if(r->d_snap) (void) hipFree(r->d_snap);
	if(r->d_nodes) (void) hipFree(r->d_nodes);
	if(r->d_acc) (void) hipFree(r->d_acc);
	if(r->d_prog) (void) hipFree(r->d_prog);
	if(r->d_frame) (void) hipFree(r->d_frame);
	if(r->frame_e0) (void) hipEventDestroy(r->frame_e0);
	if(r->frame_e1) (void) hipEventDestroy(r->frame_e1);
	for(SkrTimingHook &h : r->timed) { (void) hipEventDestroy(h.start); (void) hipEventDestroy(h.stop); }
	for(SkrTimingHook &h : r->free_pairs) { (void) hipEventDestroy(h.start); (void) hipEventDestroy(h.stop); }
	delete r;
}

uint32_t skr_tile_count(const skr_options *opt, uint32_t tile_rows, uint32_t first_tile, uint32_t tile_stride)
{
	if(!opt || opt->height <= 0 || tile_rows == 0 || tile_stride == 0) return 0;
	const uint32_t total = ((uint32_t) opt->height + tile_rows - 1) / tile_rows;
	if(first_tile >= total) return 0;
	return (total - first_tile + tile_stride - 1) / tile_stride;
}

static int check_options(const skr_options *opt)
{
	if(opt->width <= 0 || opt->height <= 0 || opt->width > 65536 || opt->height > 65536)
	{
		skr_set_error("bad image size %dx%d", opt->width, opt->height);
		return SKR_ERR_ARG;
	}
	if(opt->max_depth <= 0)
	{ // main.cpp:318-329: "depth takes a positive int"
		skr_set_error("depth takes a positive int after flag for the max depth");
		return SKR_ERR_ARG;
	}
	if(opt->grid_size < 0 || opt->grid_size > 1024 || opt->num_path_traces < 0 || opt->num_path_traces > 32767)
	{
		skr_set_error("jsample/gillum out of range (%d, %d)", opt->grid_size, opt->num_path_traces);
		return SKR_ERR_ARG;
	}
	return SKR_OK;
}

// one frame (one pass of a progressive render) of the tiles first_tile, first_tile + tile_stride, ...
// the tiles of a launch: first, first + stride, ... (table == nullptr) or the n_slots entries of a device table (skr_render_tile_list)
struct TileSel {
	uint32_t first = 0, stride = 1, max_tiles = 0xffffffffu;
	const uint32_t *d_table = nullptr;
	uint32_t n_slots = 0;
};
static uint32_t sel_tiles(const skr_options *opt, uint32_t tile_rows, const TileSel &ts)
{
	if(ts.d_table) return ts.n_slots;
	const uint32_t n = skr_tile_count(opt, tile_rows, ts.first, ts.stride);
	return n > ts.max_tiles ? ts.max_tiles : n;
}

static int render_pass(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, const TileSel &ts, uint8_t *d_rgb, float *d_rgbf, void *stream)
{
	const uint32_t first_tile = ts.first, tile_stride = ts.stride;
	if(!r || !opt || (!d_rgb && !d_rgbf) || tile_rows == 0 || tile_stride == 0)
	{
		skr_set_error("skr_render_tiles: bad argument");
		return SKR_ERR_ARG;
	}
	int rc = check_options(opt);
	if(rc != SKR_OK) return rc;
	const uint32_t n_tiles = sel_tiles(opt, tile_rows, ts);
	if(n_tiles == 0) return SKR_OK;
	SKR_HIP(hipSetDevice(r->device));

	RenderParams p{};
	p.sw = r->sw;
	p.width = opt->width;
	p.height = opt->height;
	p.tile_rows = tile_rows;
	p.first_tile = first_tile;
	p.tile_table = ts.d_table;
	p.tile_stride = tile_stride;
	p.out_rows = n_tiles * tile_rows;
	p.band_row0 = 0;
	p.band_rows = p.out_rows;
	// main.cpp:134-137, hoisted: identical float/double expressions evaluated once
	p.inv_width = 1 / float(opt->width);
	p.inv_height = 1 / float(opt->height);
	p.aspect = opt->width / float(opt->height);
	p.angle = (float) tan(M_PI * 0.5 * opt->fov / 180.);
	const float *c = r->info.camera;
	p.cam_pos = f3{c[0], c[1], c[2]};
	p.cam_dir = f3{c[3], c[4], c[5]};
	p.cam_up = f3{c[6], c[7], c[8]};
	p.cam_right = f3{c[9], c[10], c[11]};
	p.background = f3{r->info.background[0], r->info.background[1], r->info.background[2]};
	p.n_spheres = r->info.n_spheres;
	p.n_tris = r->info.n_triangles;
	p.n_lights = r->info.n_point_lights + r->info.n_directional_lights; // (directional ones only under --strict-scn)
	p.sph_geom = r->d_blob;
	p.cam_ec = r->d_camec;
	p.sph_amb = r->d_blob + r->off_amb;
	p.sph_kd = r->d_blob + r->off_kd;
	p.sph_ks = r->d_blob + r->off_ks;
	p.lights = r->d_blob + r->off_lights;
	p.tris = r->d_blob + r->off_tris;
	p.tri_chunks = r->d_blob + r->off_chunks;
	p.tri_chunk_size = r->chunk_size;
	p.tri_cones = (r->cones && !r->sw.no_cones) ? 1 : 0;
	{ // pick the tightest set of chunk spheres whose |d| bound covers this frame's camera rays (GI children stay below 4,
	  // the smallest bound): primary directions are dir + u right + v up (main.cpp:154-155)
		auto len3 = [](const float *v) { return std::sqrt((double) v[0] * v[0] + (double) v[1] * v[1] + (double) v[2] * v[2]); };
		const double umax = std::fabs((double) p.angle * p.aspect) * 1.001, vmax = std::fabs((double) p.angle) * 1.001;
		const double dmax = len3(c + 3) + umax * len3(c + 9) + vmax * len3(c + 6);
		const double bound[SKR_CULL_LEVELS] = SKR_CULL_DMAX_LIST;
		int level = 0;
		while(level < SKR_CULL_LEVELS && !(dmax < bound[level])) level++;
		p.n_tri_chunks = (level < SKR_CULL_LEVELS && !r->sw.no_cull) ? r->n_chunks : 0;
		if(p.n_tri_chunks) p.tri_chunks += (size_t) level * r->chunk_stride;
	}
	p.monte_carlo = opt->monte_carlo ? 1 : 0;
	p.num_path_traces = opt->num_path_traces;
	p.grid_size = opt->grid_size;
	p.max_depth = opt->max_depth;
	p.use_shadows = opt->use_shadows ? 1 : 0;
	p.pow_steps = r->pow_steps;
	// shade() only recurses under --gillum and only below a sphere hit (raytrace.h:208-218), and with N = 0 there is no
	// child to recurse into: every --depth is then the depth-1 image
	// (--shade-triangles: a triangle hit recurses too)
	p.shade_triangles = (opt->shade_triangles && p.n_tris > 0) ? 1 : 0;
	p.tri_mats = r->d_blob + r->off_tri_mats;
	// Only a sphere hit has the terms of raytrace.h:45-103, so a scene without spheres has nothing to add — but the flag also sets the
	// arity of the counter RNG's node ids (N + 2 L, include/skr.h), and a tree over triangle surfaces (--shade-triangles --gillum) has
	// nodes whatever the scene holds: there the flag stays, the legacy children simply never exist (found by tests/fuzz_parity.py:
	// with the flag folded away the node ids from the fourth level down were numbered with arity N)
	p.legacy_reflect = (opt->legacy_reflect && (p.n_spheres > 0 || p.shade_triangles)) ? 1 : 0;
	if(!p.legacy_reflect && (!p.monte_carlo || (p.n_spheres == 0 && !p.shade_triangles) || p.num_path_traces == 0)) p.max_depth = 1;
	if(p.max_depth > 1)
	{ // tree node ids are 32-bit RNG counter words: need N^(depth-1) < 2^32
		double nodes = 1;
		const double arity = (double) (p.monte_carlo ? p.num_path_traces : 0) + (p.legacy_reflect ? 2.0 * p.n_lights : 0.0); // children per node
		for(int k = 1; k < p.max_depth && nodes < 4294967296.0; k++) nodes = nodes * arity + 1;
		if(nodes >= 4294967296.0)
		{
			skr_set_error("gillum %d at depth %d needs more than 2^32 tree nodes per sample", p.num_path_traces, p.max_depth);
			return SKR_ERR_UNSUPPORTED;
		}
	}
	p.seed_lo = (uint32_t) opt->seed;
	p.seed_hi = (uint32_t) (opt->seed >> 32);
	p.rgb = d_rgb;
	p.rgbf = d_rgbf;
	p.counters = r->d_counters;
	p.tri_work = r->count_tri ? r->d_tri_work : nullptr;
	p.qctr = reinterpret_cast<uint32_t *>(r->d_counters + (size_t) SKR_COUNTER_SHARDS * 4 + 8);
	const bool generic_path = skr_generic_selected(p);
	const bool nodes_path = !generic_path && skr_nodes_selected(p);
	if(generic_path && !skr_generic_supported(p))
	{
		skr_set_error("--depth %d with %d children per node: the tables of one output row exceed the scratch budget (SKR_LEVELS_BUDGET_MB)", p.max_depth, p.num_path_traces + (p.legacy_reflect ? 2 * p.n_lights : 0));
		return SKR_ERR_UNSUPPORTED;
	}
	if(nodes_path || generic_path)
	{
		const size_t need = nodes_path ? skr_nodes_scratch_bytes(p) : skr_generic_scratch_bytes(p);
		if(need > r->nodes_cap)
		{
			if(r->d_nodes) SKR_HIP(hipFree(r->d_nodes));
			r->d_nodes = nullptr;
			r->nodes_cap = 0;
			SKR_HIP(hipMalloc(&r->d_nodes, need));
			r->nodes_cap = need;
		}
		p.node_scratch = r->d_nodes;
		if(p.grid_size > 0)
		{
			const size_t need_acc = (size_t) p.width * p.out_rows * 12;
			if(need_acc > r->acc_cap)
			{
				if(r->d_acc) SKR_HIP(hipFree(r->d_acc));
				r->d_acc = nullptr;
				r->acc_cap = 0;
				SKR_HIP(hipMalloc((void **) &r->d_acc, need_acc));
				r->acc_cap = need_acc;
			}
			p.acc = r->d_acc;
		}
	}
	if(skr_render_lds_bytes(p) > (size_t) r->lds_limit)
	{
		skr_set_error("scene needs %zu bytes of LDS (%d spheres, %d lights); the device allows %d per workgroup", skr_render_lds_bytes(p),
					  p.n_spheres, p.n_lights, r->lds_limit);
		return SKR_ERR_UNSUPPORTED;
	}
	SkrTimingHook hook;
	if(r->timing)
	{
		if(!r->free_pairs.empty())
		{
			hook = r->free_pairs.back();
			r->free_pairs.pop_back();
		}
		else
		{
			SKR_HIP(hipEventCreate(&hook.start));
			SKR_HIP(hipEventCreate(&hook.stop));
		}
		if(!r->d_snap)
		{
			SKR_HIP(hipMalloc((void **) &r->d_snap, (size_t) 2 * SKR_COUNTER_SHARDS * 4 * sizeof(unsigned long long)));
			SKR_HIP(hipMemset(r->d_snap, 0, (size_t) 2 * SKR_COUNTER_SHARDS * 4 * sizeof(unsigned long long)));
		}
		hook.snap = r->d_snap;
		hook.counters = r->d_counters;
	}
	r->last_p = p;
	r->last_nodes = nodes_path;
	SKR_HIP(skr_launch_render(p, (hipStream_t) stream, &g_variant, r->timing ? &hook : nullptr));
	if(r->timing) r->timed.push_back(hook);
	return SKR_OK;
}

// grows the renderer's progressive scratch to 2 x n floats: this pass's frame | the running sum
static int ensure_progressive_scratch(skr_renderer *r, size_t n)
{
	if(2 * n * sizeof(float) > r->prog_cap)
	{
		if(r->d_prog) SKR_HIP(hipFree(r->d_prog));
		r->d_prog = nullptr;
		r->prog_cap = 0;
		SKR_HIP(hipMalloc((void **) &r->d_prog, 2 * n * sizeof(float)));
		r->prog_cap = 2 * n * sizeof(float);
	}
	return SKR_OK;
}

// skr_options.progressive_passes: K frames under the seeds s, s+1, ..., s+K-1, summed in binary32 in pass
// order, divided by K once and quantised like a single frame (accumulate.hip).  K <= 1 is the single frame itself.
static int render_impl(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, const TileSel &ts, uint8_t *d_rgb, float *d_rgbf, void *stream)
{
	if(!opt || opt->progressive_passes <= 1) return render_pass(r, opt, tile_rows, ts, d_rgb, d_rgbf, stream);
	if(!r || (!d_rgb && !d_rgbf) || tile_rows == 0 || ts.stride == 0)
	{
		skr_set_error("skr_render_tiles: bad argument");
		return SKR_ERR_ARG;
	}
	int rc = check_options(opt);
	if(rc != SKR_OK) return rc;
	const uint32_t n_tiles = sel_tiles(opt, tile_rows, ts);
	if(n_tiles == 0) return SKR_OK;
	SKR_HIP(hipSetDevice(r->device));
	const uint32_t out_rows = n_tiles * tile_rows;
	const size_t n = (size_t) opt->width * out_rows * 3;
	rc = ensure_progressive_scratch(r, n);
	if(rc != SKR_OK) return rc;
	float *frame = r->d_prog, *acc = r->d_prog + n;
	skr_options pass = *opt;
	pass.progressive_passes = 1;
	for(int32_t k = 0; k < opt->progressive_passes; k++)
	{
		pass.seed = opt->seed + (uint64_t) k;
		rc = render_pass(r, &pass, tile_rows, ts, nullptr, frame, stream);
		if(rc != SKR_OK) return rc;
		SKR_HIP(skr_launch_accumulate(acc, frame, n, k == 0, (hipStream_t) stream));
	}
	SKR_HIP(skr_launch_resolve_accumulated(acc, (uint32_t) opt->progressive_passes, (uint32_t) opt->width, out_rows, (uint32_t) opt->height, tile_rows, ts.first,
										   ts.stride, ts.d_table, d_rgb, d_rgbf, (hipStream_t) stream));
	return SKR_OK;
}

int skr_render_tiles(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint32_t first_tile, uint32_t tile_stride,
					 uint8_t *d_rgb, float *d_rgbf, void *stream)
{
	TileSel ts;
	ts.first = first_tile;
	ts.stride = tile_stride;
	return render_impl(r, opt, tile_rows, ts, d_rgb, d_rgbf, stream);
}

int skr_render_tile_list(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, const uint32_t *d_tiles, uint32_t n_slots, uint8_t *d_rgb, float *d_rgbf, void *stream)
{
	if(!d_tiles)
	{
		skr_set_error("skr_render_tile_list: no tile table");
		return SKR_ERR_ARG;
	}
	TileSel ts;
	ts.d_table = d_tiles;
	ts.n_slots = n_slots;
	return render_impl(r, opt, tile_rows, ts, d_rgb, d_rgbf, stream);
}

// Per tile of `tile_rows` image rows: the work its pixels cost, counted — the tile is rendered on its own and the work counters read
// (radiance rays, shaded hits, ray-sphere tests as the reference's loops run them), priced in flops like bench.py prices a frame
// (34 per ray-sphere test, 150 per shaded hit; a ray through a triangle mesh walks ~6 chunk spheres and ~5 triangles).
// Integer counts of a bit-reproducible render: every rank of a job computes the same numbers.  Synchronous, one small render per tile
// (tens of milliseconds for a 1080p frame: the frame steps do it once per frame geometry); the caller's work counters are preserved.
int skr_tile_costs(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint64_t *h_cost)
{
	if(!r || !opt || !h_cost || tile_rows == 0) return SKR_ERR_ARG;
	int rc = check_options(opt);
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipSetDevice(r->device));
	const uint32_t T = ((uint32_t) opt->height + tile_rows - 1) / tile_rows;
	SKR_HIP(hipDeviceSynchronize()); // (frames still in flight on other streams add to the counters this probe borrows)
	std::vector<unsigned long long> saved((size_t) SKR_COUNTER_SHARDS * 4 + 8);
	SKR_HIP(hipMemcpy(saved.data(), r->d_counters, saved.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	SKR_HIP(hipMemset(r->d_counters, 0, saved.size() * sizeof(unsigned long long)));
	std::vector<uint32_t> tiles(T);
	for(uint32_t t = 0; t < T; t++) tiles[t] = t;
	uint32_t *d_tiles = nullptr;
	uint8_t *d_rgb = nullptr;
	hipError_t e = hipMalloc((void **) &d_tiles, (size_t) T * sizeof(uint32_t));
	if(e == hipSuccess) e = hipMalloc((void **) &d_rgb, (size_t) tile_rows * (size_t) opt->width * 3);
	if(e == hipSuccess) e = hipMemcpy(d_tiles, tiles.data(), (size_t) T * sizeof(uint32_t), hipMemcpyHostToDevice);
	const bool was_timing = r->count_tri;
	r->count_tri = false;
	for(uint32_t t = 0; t < T && e == hipSuccess && rc == SKR_OK; t++)
	{
		rc = skr_render_tile_list(r, opt, tile_rows, d_tiles + t, 1, d_rgb, nullptr, nullptr);
		uint64_t w[4] = {0, 0, 0, 0};
		if(rc == SKR_OK) rc = skr_renderer_read_work(r, w, 1);
		const uint64_t mesh = r->info.n_triangles > 64 ? (uint64_t) (6 * 19 + 5 * 46) : (uint64_t) r->info.n_triangles * 46;
		h_cost[t] = 34 * w[3] + 150 * w[1] + (20 + mesh) * (w[0] + w[2]);
	}
	r->count_tri = was_timing;
	if(d_tiles) (void) hipFree(d_tiles);
	if(d_rgb) (void) hipFree(d_rgb);
	if(e == hipSuccess) e = hipMemcpy(r->d_counters, saved.data(), saved.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
	if(e != hipSuccess)
	{
		skr_set_error("skr_tile_costs: %s", hipGetErrorString(e));
		return SKR_ERR_HIP;
	}
	return rc;
}

int skr_accumulate(float *d_acc, const float *d_frame, uint64_t n_floats, int first, void *stream)
{
	if(!d_acc || !d_frame)
	{
		skr_set_error("skr_accumulate: null argument");
		return SKR_ERR_ARG;
	}
	SKR_HIP(skr_launch_accumulate(d_acc, d_frame, (size_t) n_floats, first, (hipStream_t) stream));
	return SKR_OK;
}

int skr_resolve_accumulated(const float *d_acc, uint32_t passes, uint32_t width, uint32_t height, uint8_t *d_rgb, float *d_rgbf, void *stream)
{
	if(!d_acc || passes == 0 || (!d_rgb && !d_rgbf))
	{
		skr_set_error("skr_resolve_accumulated: bad argument");
		return SKR_ERR_ARG;
	}
	SKR_HIP(skr_launch_resolve_accumulated(d_acc, passes, width, height, height, height ? height : 1, 0, 1, nullptr, d_rgb, d_rgbf, (hipStream_t) stream));
	return SKR_OK;
}

int skr_render_rows(skr_renderer *r, const skr_options *opt, uint32_t y0, uint32_t y1, uint8_t *d_rgb, float *d_rgbf, void *stream)
{
	if(!opt || y1 <= y0 || y1 > (uint32_t) opt->height)
	{
		skr_set_error("skr_render_rows: bad row range [%u,%u)", y0, y1);
		return SKR_ERR_ARG;
	}
	// rows [y0,y1) = consecutive tiles of g = gcd(y0, y1-y0) rows starting at tile y0/g
	uint32_t a = y0, b = y1 - y0;
	while(b)
	{
		const uint32_t t = a % b;
		a = b;
		b = t;
	}
	TileSel ts;
	ts.first = y0 / a;
	ts.stride = 1;
	ts.max_tiles = (y1 - y0) / a;
	return render_impl(r, opt, a, ts, d_rgb, d_rgbf, stream);
}

int skr_renderer_reload_switches(skr_renderer *r)
{
	if(!r) return SKR_ERR_ARG;
	load_switches(r->sw);
	return SKR_OK;
}

int skr_renderer_kernel_timing(skr_renderer *r, int enable)
{
	if(!r) return SKR_ERR_ARG;
	r->timing = enable != 0;
	return SKR_OK;
}

int skr_renderer_kernel_ms(skr_renderer *r, float *mean_ms, int32_t *launches)
{
	if(!r || !mean_ms) return SKR_ERR_ARG;
	SKR_HIP(hipSetDevice(r->device));
	double sum = 0;
	int n = 0;
	for(SkrTimingHook &h : r->timed)
	{
		SKR_HIP(hipEventSynchronize(h.stop));
		float ms = 0;
		SKR_HIP(hipEventElapsedTime(&ms, h.start, h.stop));
		sum += ms;
		n++;
		r->free_pairs.push_back(h);
	}
	r->timed.clear();
	*mean_ms = n ? (float) (sum / n) : 0.0f;
	if(launches) *launches = n;
	return SKR_OK;
}

// node pipeline: records of level `level` in the band last rendered (level 0: its level-0 nodes)
static int nodes_level_count(skr_renderer *r, int level, uint32_t *n)
{
	size_t off = 0, words = 0;
	int levels = 0;
	*n = 0;
	if(!skr_nodes_counter_layout(r->last_p, &off, &words, &levels) || level >= levels) return SKR_OK;
	const uint32_t *ctr = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(r->d_nodes) + off);
	if(level == 0)
	{
		SKR_HIP(hipMemcpy(n, ctr, sizeof(uint32_t), hipMemcpyDeviceToHost));
		return SKR_OK;
	}
	std::vector<uint32_t> h((size_t) SKR_P1_REGIONS * SKR_PULL_STRIDE);
	SKR_HIP(hipMemcpy(h.data(), ctr + SKR_PULL_STRIDE + words * (size_t) level, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
	uint64_t total = 0;
	for(uint32_t k = 0; k < SKR_P1_REGIONS; k++) total += h[(size_t) SKR_PULL_STRIDE * k];
	*n = (uint32_t) total;
	return SKR_OK;
}

int skr_renderer_last_parent_count(skr_renderer *r, uint32_t *n)
{
	if(!r || !n) return SKR_ERR_ARG;
	SKR_HIP(hipSetDevice(r->device));
	*n = 0;
	if(r->last_nodes) return nodes_level_count(r, 0, n); // (only the node pipeline has level tables of this layout)
	return SKR_OK;
}

int skr_renderer_last_level1_count(skr_renderer *r, uint32_t *n)
{
	if(!r || !n) return SKR_ERR_ARG;
	SKR_HIP(hipSetDevice(r->device));
	*n = 0;
	if(r->last_nodes) return nodes_level_count(r, 1, n);
	return SKR_OK;
}

static int read_work(skr_renderer *r, uint64_t *out, int n_out, int reset)
{
	if(!r || !out) return SKR_ERR_ARG;
	SKR_HIP(hipSetDevice(r->device));
	// the work counters and (diagnostic builds) the 8 phase stamps behind them; the queue counters that follow are not touched
	std::vector<unsigned long long> h((size_t) SKR_COUNTER_SHARDS * 4 + 8);
	SKR_HIP(hipMemcpy(h.data(), r->d_counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost)); // synchronises with prior launches
	for(int k = 0; k < n_out; k++) out[k] = 0;
	for(size_t s = 0; s < SKR_COUNTER_SHARDS; s++)
		for(int k = 0; k < n_out; k++) out[k] += h[4 * s + k];
	if(getenv("SKR_PRINT_STAMPS"))
	{ // diagnostic builds (-DSKR_STAMPS=1) only: per-phase cycle sums
		for(int k = 0; k < 8; k++) fprintf(stderr, "stamp[%d] = %llu\n", k, h[(size_t) SKR_COUNTER_SHARDS * 4 + k]);
	}
	// (null stream: callers read the counters between frames, after synchronising their render stream)
	if(reset) SKR_HIP(hipMemset(r->d_counters, 0, h.size() * sizeof(unsigned long long)));
	return SKR_OK;
}

int skr_renderer_read_counters(skr_renderer *r, uint64_t out[3], int reset) { return read_work(r, out, 3, reset); }

int skr_renderer_read_work(skr_renderer *r, uint64_t out[4], int reset)
{
	const int rc = read_work(r, out, 4, reset);
	if(rc != SKR_OK) return rc;
	// every radiance ray tests every sphere (raytrace.h:152-165); a shadow ray stops at its first occluder (utils.h:52-55)
	out[3] += out[0] * (uint64_t) r->info.n_spheres;
	return SKR_OK;
}

int skr_renderer_count_triangle_work(skr_renderer *r, int enable)
{
	if(!r) return SKR_ERR_ARG;
	r->count_tri = enable != 0;
	return SKR_OK;
}

int skr_renderer_kernel_work(skr_renderer *r, uint64_t out[4])
{
	if(!r || !out) return SKR_ERR_ARG;
	for(int k = 0; k < 4; k++) out[k] = 0;
	if(!r->d_snap) return SKR_OK;
	SKR_HIP(hipSetDevice(r->device));
	std::vector<unsigned long long> h((size_t) 2 * SKR_COUNTER_SHARDS * 4);
	SKR_HIP(hipMemcpy(h.data(), r->d_snap, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost)); // synchronises with prior launches
	const size_t half = (size_t) SKR_COUNTER_SHARDS * 4;
	for(size_t s = 0; s < SKR_COUNTER_SHARDS; s++)
		for(int k = 0; k < 4; k++) out[k] += h[half + 4 * s + k] - h[4 * s + k];
	out[3] += out[0] * (uint64_t) r->info.n_spheres; // (as skr_renderer_read_work)
	return SKR_OK;
}

int skr_renderer_read_triangle_work(skr_renderer *r, uint64_t out[3], int reset)
{
	if(!r || !out) return SKR_ERR_ARG;
	SKR_HIP(hipSetDevice(r->device));
	unsigned long long h[256 * 2];
	SKR_HIP(hipMemcpy(h, r->d_tri_work, sizeof(h), hipMemcpyDeviceToHost)); // synchronises with prior launches
	out[0] = out[1] = 0;
	for(int s = 0; s < 256; s++)
	{
		out[0] += h[2 * s];
		out[1] += h[2 * s + 1];
	}
	uint64_t w[4];
	const int rc = read_work(r, w, 4, 0);
	if(rc != SKR_OK) return rc;
	out[2] = w[0] * (uint64_t) r->info.n_triangles; // raytrace.h:171-186: every radiance ray tests every triangle
	if(reset) SKR_HIP(hipMemset(r->d_tri_work, 0, sizeof(h)));
	return SKR_OK;
}

int skr_render_progressive_host(skr_renderer *r, const skr_options *opt, uint32_t every, uint8_t *h_rgb, float *h_rgbf, skr_progress_fn progress, void *user,
								float *kernel_ms)
{
	if(!r || !opt || (!h_rgb && !h_rgbf))
	{
		skr_set_error("skr_render_progressive_host: bad argument");
		return SKR_ERR_ARG;
	}
	int rc = check_options(opt); // before anything is sized from width x height
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipSetDevice(r->device));
	const size_t pixels = (size_t) opt->width * opt->height, bytes = pixels * 3, fbytes = h_rgbf ? pixels * 12 : 0;
	if(bytes + fbytes > r->frame_cap)
	{ // the device frame (u8, then float) and the two events live in the renderer: nothing to leak on an early return
		if(r->d_frame) SKR_HIP(hipFree(r->d_frame));
		r->d_frame = nullptr;
		r->frame_cap = 0;
		SKR_HIP(hipMalloc((void **) &r->d_frame, ((bytes + 15) & ~(size_t) 15) + pixels * 12));
		r->frame_cap = bytes + pixels * 12;
	}
	uint8_t *d_rgb = r->d_frame;
	float *d_rgbf = h_rgbf ? reinterpret_cast<float *>(r->d_frame + ((bytes + 15) & ~(size_t) 15)) : nullptr;
	if(!r->frame_e0) SKR_HIP(hipEventCreate(&r->frame_e0));
	if(!r->frame_e1) SKR_HIP(hipEventCreate(&r->frame_e1));
	const uint32_t passes = opt->progressive_passes > 1 ? (uint32_t) opt->progressive_passes : 1u;
	float total_ms = 0;
	if(!progress || every == 0 || every >= passes)
	{ // nobody looks before the end: the frame (or the K-pass mean) in one launch sequence
		SKR_HIP(hipEventRecord(r->frame_e0, nullptr));
		rc = skr_render_tiles(r, opt, (uint32_t) opt->height, 0, 1, d_rgb, d_rgbf, nullptr);
		if(rc != SKR_OK) return rc;
		SKR_HIP(hipEventRecord(r->frame_e1, nullptr));
		if(h_rgb) SKR_HIP(hipMemcpy(h_rgb, d_rgb, bytes, hipMemcpyDeviceToHost));
		if(h_rgbf) SKR_HIP(hipMemcpy(h_rgbf, d_rgbf, fbytes, hipMemcpyDeviceToHost));
		SKR_HIP(hipEventElapsedTime(&total_ms, r->frame_e0, r->frame_e1));
		if(progress) (void) progress(user, passes, passes, h_rgb, h_rgbf);
		if(kernel_ms) *kernel_ms = total_ms;
		return SKR_OK;
	}
	// the mean is shown while it forms: after every `every` passes (and after the last) it is resolved, copied out and handed to
	// the callback — where the SDL viewer of main.cpp:183-197 would blit.  The final mean is the one-launch result, bit for bit.
	const size_t n = pixels * 3;
	rc = ensure_progressive_scratch(r, n);
	if(rc != SKR_OK) return rc;
	float *frame = r->d_prog, *acc = r->d_prog + n;
	skr_options pass = *opt;
	pass.progressive_passes = 1;
	for(uint32_t k = 0; k < passes; k++)
	{
		pass.seed = opt->seed + (uint64_t) k;
		SKR_HIP(hipEventRecord(r->frame_e0, nullptr));
		rc = render_pass(r, &pass, (uint32_t) opt->height, TileSel(), nullptr, frame, nullptr);
		if(rc != SKR_OK) return rc;
		SKR_HIP(skr_launch_accumulate(acc, frame, n, k == 0, nullptr));
		const bool show = (k + 1) % every == 0 || k + 1 == passes;
		if(show) SKR_HIP(skr_launch_resolve_accumulated(acc, k + 1, (uint32_t) opt->width, (uint32_t) opt->height, (uint32_t) opt->height, (uint32_t) opt->height, 0, 1, nullptr, d_rgb, d_rgbf, nullptr));
		SKR_HIP(hipEventRecord(r->frame_e1, nullptr));
		if(show)
		{
			if(h_rgb) SKR_HIP(hipMemcpy(h_rgb, d_rgb, bytes, hipMemcpyDeviceToHost));
			if(h_rgbf) SKR_HIP(hipMemcpy(h_rgbf, d_rgbf, fbytes, hipMemcpyDeviceToHost));
		}
		else SKR_HIP(hipEventSynchronize(r->frame_e1));
		float ms = 0;
		SKR_HIP(hipEventElapsedTime(&ms, r->frame_e0, r->frame_e1));
		total_ms += ms;
		if(show && progress(user, k + 1, passes, h_rgb, h_rgbf) != 0) break; // the viewer was closed: what is in the buffers is the mean so far
	}
	if(kernel_ms) *kernel_ms = total_ms;
	return SKR_OK;
}

int skr_render_frame_host(skr_renderer *r, const skr_options *opt, uint8_t *h_rgb, float *kernel_ms)
{
	if(!h_rgb) return SKR_ERR_ARG;
	return skr_render_progressive_host(r, opt, 0, h_rgb, nullptr, nullptr, nullptr, kernel_ms);
}

const char *skr_kernel_variant(void) { return g_variant; }

int skr_debug_eval(int op, const void *d_in, void *d_out, uint32_t n, void *stream)
{
	if(!d_in || !d_out || op < 0 || op > 15) return SKR_ERR_ARG;
	if(n == 0) return SKR_OK;
	SKR_HIP(skr_launch_debug(op, d_in, d_out, n, (hipStream_t) stream));
	return SKR_OK;
}

} // extern "C"
