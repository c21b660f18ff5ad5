// The general level pipeline: shade() (src/raytrace.h:139-227) with every kind of child ray it can spawn, cut at every level of the
// tree, for any --depth (main.cpp:318-329 accepts any positive depth).  One lane per RAY — not per sibling pair, no state carried
// between kernels but the tables — so that the modes the pair kernels of render_nodes.hip do not know fit in:
//
//   * triangle meshes under --gillum (HEAD semantics: an accepted triangle nearer than the closest sphere blackens the sample,
//     raytrace.h:171-186, :221-224) — one wave-level walk of the culling tree per 64 rays;
//   * --shade-triangles: a triangle is a surface; its hit is a node like a sphere's, its --gillum children do not
//     test the triangle they start on;
//   * --legacy-reflect: every shaded sphere hit has, besides its N --gillum children, two children per light —
//     the refraction and the reflection ray of raytrace.h:54-99 — from the hit point itself;
//   * --gillum beyond 256 children per node.
//
// Levels.  Level 0 is the camera: one root per pixel of the band with ONE child, the primary ray (main.cpp:140-182).  A hit of level k
// (k >= 1) is a node of level k; it is shaded when its record is activated and, while k < --depth, its A = N + 2 L children are
// the rays of level k + 1 (child c < N: the c-th --gillum ray; child N + 2 l: the refraction ray of light l; N + 2 l + 1: its
// reflection ray — the numbering of the counter RNG's node ids, DESIGN.md "Counter RNG").  Kernels, in launch order, per level:
//
//   skr_gtrace_kernel     one lane per child ray of the level above: the ray, the closest sphere, the triangles; per wave (64 rays) a
//                         32-byte header — first record, ballot of the rays that hit a surface, ballot of the rays a triangle
//                         blackened — and per hit a 32-byte record {parent, surface, child, t, d} appended to one of 64 regions
//   skr_gactivate_kernel  one lane per record: the hit is shaded (raytrace.h:194-207) and becomes a node (80 bytes); a hit of the
//                         last level, whose children are shade(depth 0) == 0, is finished here
//   skr_gfinalize_kernel  deepest level first, one lane per node: its children's values in the reference's order —
//                         refraction_colour = fr * shade() (an assignment: the last light's stays), reflection_colour += (1 - fr) *
//                         specular * shade() (:70-76), total += r1 * shade() / pdf (:130), (direct / pi + 2 indirect) * kd (:213) —;
//                         level 0: the pixel
//
// Every float operation and every order of summation is the reference's; the image is bit-identical to the oracle's
// (tests/test_gpu_parity.py, test_shade_triangles.py, test_legacy_reflect.py run every mode through it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wave_common.h"

namespace {

constexpr uint32_t SURF_TRI = 0x80000000u; // surface code: a sphere index, or SURF_TRI | the triangle's slot in tris[]
constexpr int GNODE_ROWS = 5;              // float4 per node: [P.xyz N.x] [N.yz d.x d.y] [d.z surface pixel node-id] [direct.xyz fr] [own record, file index of a triangle (-1), -, -]
constexpr int GREC_ROWS = 2;               // float4 per record: [parent surface child t] [d.xyz -]
constexpr int GHDR_ROWS = 2;               // uint4 per trace wave: [first record, -, -, -] [hit ballot lo hi, black ballot lo hi]

struct GChild { // one child ray of a node, as the trace kernel forms it and the finalize kernel weighs it
	bool exists;
	f3 o, d;
	int from_tri; // --shade-triangles: the file index of the triangle the ray starts on (-1: none)
};

struct GNode {
	f3 P, N, d, direct;
	float fr;
	uint32_t surf, pixel, node_id, rec;
	int file;
};

SKR_DEV GNode load_node(const float4 *rows)
{
	const float4 a = rows[0], b = rows[1], c = rows[2], e = rows[3], f = rows[4];
	GNode n;
	n.P = mk3(a.x, a.y, a.z);
	n.N = mk3(a.w, b.x, b.y);
	n.d = mk3(b.z, b.w, c.x);
	n.surf = __float_as_uint(c.y);
	n.pixel = __float_as_uint(c.z);
	n.node_id = __float_as_uint(c.w);
	n.direct = mk3(e.x, e.y, e.z);
	n.fr = e.w;
	n.rec = __float_as_uint(f.x);
	n.file = (int) __float_as_uint(f.y);
	return n;
}

// the material rows of a surface: {La * ka, power}, kd, {ks, ior}
SKR_DEV void surface_material(const SceneView &sv, const RenderParams &p, uint32_t surf, float4 &ambp, f3 &kd, float4 &ks4)
{
	if(surf & SURF_TRI)
	{
		const uint32_t slot = surf & ~SURF_TRI;
		ambp = p.tri_mats[3 * slot];
		kd = ld3(p.tri_mats[3 * slot + 1]);
		ks4 = p.tri_mats[3 * slot + 2];
	}
	else
	{
		ambp = sv.amb[surf];
		kd = ld3(sv.kd[surf]);
		ks4 = sv.ks[surf];
	}
}

// does the node spawn the legacy children (raytrace.h:52: a specular colour other than (0,0,0); only a sphere hit has these terms)
SKR_DEV bool legacy_children(const RenderParams &p, const SceneView &sv, uint32_t surf)
{
	if(!p.legacy_reflect || (surf & SURF_TRI)) return false;
	const float4 ks = sv.ks[surf];
	return ks.x != 0.0f || ks.y != 0.0f || ks.z != 0.0f;
}

// child c of node n (raytrace.h:117-131 for c < N; :54-99 for the two rays of light (c - N) / 2)
SKR_DEV GChild child_of(const SceneView &sv, const RenderParams &p, const GNode &n, uint32_t c)
{
	GChild ch;
	ch.from_tri = -1;
	ch.o = n.P;
	ch.d = mk3(0, 0, 1);
	const uint32_t N = (uint32_t) (p.monte_carlo ? p.num_path_traces : 0);
	if(c < N)
	{
		ch.exists = true;
		f3 nt, nb;
		tangent_basis(n.N, nt, nb);
		uint32_t rnd[4];
		philox4x32(n.pixel, p.aa_index, n.node_id, c >> 1, p.seed_lo, p.seed_hi, rnd);
		const float r1 = u31_to_unit(rnd[2 * (c & 1u)]), r2 = u31_to_unit(rnd[2 * (c & 1u) + 1]);
		ch.d = gi_direction(r1, r2, n.N, nt, nb);
		ch.o = add_scalar(n.P, 0.00001f);
		ch.from_tri = n.file;
		return ch;
	}
	const uint32_t l = (c - N) >> 1;
	const bool reflection = (c - N) & 1u;
	ch.exists = legacy_children(p, sv, n.surf) && (reflection || n.fr < 1);
	if(ch.exists)
	{
		if(reflection) ch.d = legacy_reflect_dir(light_term(sv, (int) l, n.P).L, n.N);
		else ch.d = legacy_refraction_dir(n.d, n.N, sv.ks[n.surf].w);
	}
	return ch;
}

} // namespace

// =====================================================================================================================
// trace: one lane per child ray of the level above (level 1: the primary rays)
// =====================================================================================================================
__global__ __launch_bounds__(256) void skr_gtrace_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const uint32_t A = p.g_arity; // children per node of the level above (1 at the camera level)
	const uint64_t n_rays = (uint64_t) *p.nd_count * A;
	if((uint64_t) blockIdx.x * 256u >= n_rays) return; // (uniform per workgroup)
	const SceneView sv = stage_scene(p, lds4, true);
	const int tid = threadIdx.x, lane = tid & 63;
	Counters cn{0, 0, 0};
	for(uint32_t blk = blockIdx.x; (uint64_t) blk * 256u < n_rays; blk += gridDim.x)
	{
		const uint32_t chunk = blk * 4u + (uint32_t) (tid >> 6);
		const uint64_t ri = (uint64_t) chunk * 64u + (uint32_t) lane;
		const bool valid = ri < n_rays;
		const uint32_t node = valid ? (uint32_t) (ri / A) : 0u, c = valid ? (uint32_t) (ri - (uint64_t) node * A) : 0u;
		GChild ch;
		ch.exists = false;
		ch.o = ch.d = mk3(0, 0, 1);
		ch.from_tri = -1;
		if(valid)
		{
			if(p.g_level == 1)
			{ // the camera: root `node` is pixel (x, row) of the band, its one child the primary ray (main.cpp:140-182)
				const uint32_t bw = (uint32_t) p.width, row = p.band_row0 + node / bw, x = node - (node / bw) * bw;
				const uint32_t y = image_row(p, row);
				ch.exists = y < (uint32_t) p.height;
				if(ch.exists)
				{
					ch.o = p.cam_pos;
					primary_ray(p, (int) x, y, y * bw + x, p.aa_index, ch.d);
				}
			}
			else ch = child_of(sv, p, load_node(p.g_nodes_src + (size_t) node * GNODE_ROWS), c);
		}
		bool hit = false, black = false;
		uint32_t surf = 0;
		float t = 0.0f;
		if(ch.exists)
		{
			cn.rays++;
			const RayConst r = make_ray(ch.o, ch.d);
			float tmin;
			const int sph = closest_sphere(sv, r, tmin); // raytrace.h:152-165
			if(p.shade_triangles)
			{
				TriBest b{tmin, -1, -1};
				if(sv.nt > 0) closest_triangle(sv, r, ch.from_tri, b);
				hit = sph >= 0 || b.slot >= 0;
				surf = b.slot >= 0 ? (SURF_TRI | (uint32_t) b.slot) : (uint32_t) sph;
				t = b.t;
			}
			else
			{
				black = sv.nt > 0 && any_triangle_closer(sv, r, tmin); // raytrace.h:171-186, :221-224
				hit = !black && sph >= 0;
				surf = (uint32_t) sph;
				t = tmin;
			}
		}
		// append the wave's hits to its region: rank by ballot, one atomic per wave
		const unsigned long long mh = __ballot(hit), mb = __ballot(black);
		const uint32_t region = chunk & (SKR_P1_REGIONS - 1u);
		const uint32_t nh = (uint32_t) __popcll(mh);
		uint32_t base = 0;
		if(nh != 0u)
		{
			if(lane == 0) base = atomicAdd(lc_count(p.rc_ctr, region), nh);
			base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
		}
		const uint32_t rec_base = region * p.rc_cap + base;
		if(hit)
		{
			float4 *rec = p.rc + (size_t) (rec_base + (uint32_t) lanes_below(mh)) * GREC_ROWS;
			rec[0] = make_float4(__uint_as_float(node), __uint_as_float(surf), __uint_as_float(c), t);
			rec[1] = make_float4(ch.d.x, ch.d.y, ch.d.z, 0.0f);
		}
		if(lane == 0)
		{
			p.ixh[GHDR_ROWS * (size_t) chunk] = make_uint4(rec_base, 0u, 0u, 0u);
			p.ixh[GHDR_ROWS * (size_t) chunk + 1] = make_uint4((uint32_t) mh, (uint32_t) (mh >> 32), (uint32_t) mb, (uint32_t) (mb >> 32));
		}
	}
	add_counters(p, cn, (uint32_t) blockIdx.x * 4u + (uint32_t) (tid >> 6), lane);
}

// =====================================================================================================================
// activate: record -> node (raytrace.h:194-207); the last level's hits are finished here
// =====================================================================================================================
namespace {

// the value of a node whose children's values are known — or all (0,0,0): shade(depth 0), raytrace.h:142-145.  `value_of(c)`: child c.
template <typename F>
SKR_DEV f3 node_value(const SceneView &sv, const RenderParams &p, const GNode &n, F value_of)
{
	float4 ambp, ks4;
	f3 kd;
	surface_material(sv, p, n.surf, ambp, kd, ks4);
	const uint32_t N = (uint32_t) (p.monte_carlo ? p.num_path_traces : 0);
	f3 direct = n.direct;
	if(p.legacy_reflect && !(n.surf & SURF_TRI))
	{ // raytrace.h:45-102
		f3 refraction_colour = mk3(0, 0, 0), reflection_colour = mk3(0, 0, 0);
		if(legacy_children(p, sv, n.surf))
		{
			const f3 ks = ld3(ks4);
			for(int l = 0; l < sv.nl; l++)
			{
				if(n.fr < 1) refraction_colour = value_of(N + 2u * (uint32_t) l) * n.fr;                                // :70 (=, not +=)
				reflection_colour = reflection_colour + (ks * (1 - n.fr)) * value_of(N + 2u * (uint32_t) l + 1u);       // :76
			}
		}
		direct = (direct + refraction_colour) + reflection_colour; // :102
	}
	if(!p.monte_carlo) return direct; // :218
	f3 total = mk3(0, 0, 0);
	uint32_t rnd[4] = {0, 0, 0, 0};
	for(uint32_t c = 0; c < N; c++)
	{ // :117-131: total += r1 * shade() / pdf, in child order
		if((c & 1u) == 0u) philox4x32(n.pixel, p.aa_index, n.node_id, c >> 1, p.seed_lo, p.seed_hi, rnd);
		const float r1 = u31_to_unit(rnd[2 * (c & 1u)]);
		total = total + div3_const(value_of(c) * r1, SKR_DIV_PDF);
	}
	total = total / (float) p.num_path_traces;                              // :133 (N == 0: 0/0, as in the reference)
	return (div3_const(direct, SKR_DIV_PI) + total * 2.0f) * kd; // :213
}

} // namespace

__global__ __launch_bounds__(256) void skr_gactivate_kernel(const RenderParams p)
{ // a workgroup covers 256 consecutive positions of one region; positions past the region's count exit
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const uint32_t per_region = (p.rc_cap + 255u) / 256u;
	const uint32_t region = (uint32_t) blockIdx.x / per_region, pos0 = ((uint32_t) blockIdx.x % per_region) * 256u;
	const uint32_t cnt = *lc_count(p.rc_ctr, region);
	if(pos0 >= cnt && blockIdx.x != 0) return;
	__shared__ uint32_t s_pre[65];
	region_prefix(p, s_pre, blockIdx.x == 0); // (workgroup 0 leaves the level's record count for the kernels that follow)
	if(pos0 >= cnt) return;
	const SceneView sv = stage_scene(p, lds4, true);
	const uint32_t pos = pos0 + threadIdx.x;
	const bool act = pos < cnt;
	const uint32_t rec = region * p.rc_cap + pos;
	Counters cn{0, 0, 0};
	if(act)
	{
		const float4 r0 = p.rc[(size_t) rec * GREC_ROWS], r1 = p.rc[(size_t) rec * GREC_ROWS + 1];
		const uint32_t parent = __float_as_uint(r0.x), surf = __float_as_uint(r0.y), c = __float_as_uint(r0.z);
		const float t = r0.w;
		GNode n;
		n.d = mk3(r1.x, r1.y, r1.z);
		n.surf = surf;
		n.rec = rec;
		n.file = -1;
		f3 o;
		if(p.g_level == 1)
		{
			const uint32_t bw = (uint32_t) p.width, row = p.band_row0 + parent / bw, x = parent - (parent / bw) * bw;
			const uint32_t y = image_row(p, row);
			n.pixel = y * bw + x;
			n.node_id = 0u;
			o = p.cam_pos;
		}
		else
		{
			const GNode pn = load_node(p.g_nodes_src + (size_t) parent * GNODE_ROWS);
			n.pixel = pn.pixel;
			n.node_id = pn.node_id * p.g_arity + c + 1u; // DESIGN.md "Counter RNG": child c of node n
			const uint32_t N = (uint32_t) (p.monte_carlo ? p.num_path_traces : 0);
			o = c < N ? add_scalar(pn.P, 0.00001f) : pn.P; // raytrace.h:128 / :66, :73
		}
		n.P = o + n.d * t; // raytrace.h:204 (t of the winner == the loop's minimum)
		float4 ambp, ks4;
		f3 kd;
		surface_material(sv, p, surf, ambp, kd, ks4);
		if(surf & SURF_TRI)
		{ // include/skr.h skr_options.shade_triangles: the geometric normal, turned against the ray
			const uint32_t slot = surf & ~SURF_TRI;
			n.N = normalize3(cross3(ld3(sv.tris[3 * slot + 1]), ld3(sv.tris[3 * slot + 2])));
			if(dot3(n.N, n.d) > 0.0f) n.N = mk3(-n.N.x, -n.N.y, -n.N.z);
			n.file = __float_as_int(sv.tris[3 * slot + 1].w);
		}
		else n.N = normalize3(n.P - ld3(sv.geom[surf])); // :205
		cn.hits++;
		n.direct = direct_light_of<false>(sv, p, kd, ld3(ks4), ambp, n.P, n.N, cn);
		n.fr = (p.legacy_reflect && !(surf & SURF_TRI)) ? legacy_fresnel(n.d, n.N, ks4.w) : 0.0f; // :46
		if(p.g_last)
		{ // its children are shade(depth 0) == (0,0,0): the node is finished
			store3(p.res_out + (size_t) rec * 3, node_value(sv, p, n, [](uint32_t) { return mk3(0, 0, 0); }));
		}
		else
		{
			float4 *rows = p.g_nodes_dst + (size_t) (s_pre[region] + pos) * GNODE_ROWS;
			rows[0] = make_float4(n.P.x, n.P.y, n.P.z, n.N.x);
			rows[1] = make_float4(n.N.y, n.N.z, n.d.x, n.d.y);
			rows[2] = make_float4(n.d.z, __uint_as_float(n.surf), __uint_as_float(n.pixel), __uint_as_float(n.node_id));
			rows[3] = make_float4(n.direct.x, n.direct.y, n.direct.z, n.fr);
			rows[4] = make_float4(__uint_as_float(n.rec), __uint_as_float((uint32_t) n.file), 0.0f, 0.0f);
		}
	}
	add_counters(p, cn, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63);
}

// =====================================================================================================================
// finalize: one lane per node of a level (level 0: per pixel of the band)
// =====================================================================================================================
__global__ __launch_bounds__(256) void skr_gfinalize_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const uint32_t n_nodes = *p.nd_count;
	if((uint32_t) blockIdx.x * 256u >= n_nodes) return;
	const SceneView sv = stage_scene(p, lds4, true);
	const uint32_t node = (uint32_t) blockIdx.x * 256u + threadIdx.x;
	if(node >= n_nodes) return;
	const uint32_t A = p.g_arity;
	// the value of child c: its record's (a hit), (0,0,0) (a triangle took it: raytrace.h:221-224) or the background (:189-192)
	auto value_of = [&](uint32_t c) -> f3 {
		const uint64_t ri = (uint64_t) node * A + c;
		const uint4 *h = p.ixh + GHDR_ROWS * (size_t) (ri >> 6);
		const uint4 h0 = h[0], h1 = h[1];
		const unsigned long long mh = (unsigned long long) h1.y << 32 | h1.x, mb = (unsigned long long) h1.w << 32 | h1.z;
		const uint32_t bit = (uint32_t) ri & 63u;
		if((mh >> bit) & 1ull)
		{
			const float *r = p.res_in + (size_t) (h0.x + (uint32_t) __popcll(mh & ((1ull << bit) - 1ull))) * 3;
			return mk3(r[0], r[1], r[2]);
		}
		return ((mb >> bit) & 1ull) ? mk3(0, 0, 0) : p.background;
	};
	if(p.g_level == 0)
	{ // the camera level: the pixel is its primary ray's value (rows outside the image have no ray and no pixel)
		const uint32_t bw = (uint32_t) p.width, row = p.band_row0 + node / bw, x = node - (node / bw) * bw;
		if(image_row(p, row) < (uint32_t) p.height) emit_sample(p, row * bw + x, value_of(0u));
		return;
	}
	const GNode n = load_node(p.g_nodes_src + (size_t) node * GNODE_ROWS);
	GNode m = n;
	// a child that does not exist (no legacy terms for this surface; fr >= 1) is never asked for by node_value()
	store3(p.res_out + (size_t) n.rec * 3, node_value(sv, p, m, value_of));
}

// =====================================================================================================================
// host side
// =====================================================================================================================
hipError_t skr_launch_resolve(const RenderParams &p, hipStream_t stream); // render_wave.hip

constexpr int SKR_GLEVELS_MAX = 64;
struct GPlan {
	int levels = 0;            // traced levels 1 .. levels (= --depth)
	uint32_t band_rows = 0;    // output rows per band
	uint64_t nodes_max[SKR_GLEVELS_MAX + 1] = {}; // worst case: roots of the band; then every ray of the level above a hit
	uint32_t cap[SKR_GLEVELS_MAX + 1] = {};       // records per region
	size_t off_nodes[SKR_GLEVELS_MAX + 1] = {}, off_recs[SKR_GLEVELS_MAX + 1] = {}, off_res[SKR_GLEVELS_MAX + 1] = {}, off_hdr[SKR_GLEVELS_MAX + 1] = {};
	size_t off_ctr = 0, ctr_bytes = 0, total = 0;
};
static const size_t G_LVL_CTR_WORDS = (size_t) SKR_PULL_STRIDE * (2u * SKR_P1_REGIONS + 2u);
static uint32_t *g_prefix_host(uint32_t *ctr) { return ctr + SKR_PULL_STRIDE * (2u * SKR_P1_REGIONS + 1u) + 64; } // the level's record count (region_prefix)

static uint32_t g_arity(const RenderParams &p) { return (uint32_t) (p.monte_carlo ? p.num_path_traces : 0) + (p.legacy_reflect ? 2u * (uint32_t) p.n_lights : 0u); }

static bool gplan_for(const RenderParams &p, uint32_t rows, GPlan &pl)
{
	const uint64_t A = g_arity(p);
	pl.levels = (A == 0) ? 1 : p.max_depth;
	if(pl.levels < 1 || pl.levels > SKR_GLEVELS_MAX) return false;
	pl.band_rows = rows;
	pl.nodes_max[0] = (uint64_t) rows * (uint64_t) p.width;
	size_t off = 0;
	auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
	pl.ctr_bytes = (SKR_PULL_STRIDE + G_LVL_CTR_WORDS * (size_t) (pl.levels + 1)) * sizeof(uint32_t);
	pl.off_ctr = take(pl.ctr_bytes);
	for(int L = 1; L <= pl.levels; L++)
	{ // a region receives at most 64 hits from each of its trace waves (64 rays): `cap` record slots per region; the level cannot hold
	  // more hits than it has rays, which bounds the next level's rays (a band of one row would otherwise be sized for 64 x 64 nodes)
		const uint64_t rays = pl.nodes_max[L - 1] * (L == 1 ? 1 : A);
		const uint64_t chunks = (rays + 63) / 64;
		const uint64_t cap = (chunks + SKR_P1_REGIONS - 1) / SKR_P1_REGIONS * 64;
		if(cap * SKR_P1_REGIONS >= (1ull << 31)) return false;
		const uint64_t slots = cap * SKR_P1_REGIONS;
		pl.cap[L] = (uint32_t) cap;
		pl.nodes_max[L] = slots < rays ? slots : rays;
		pl.off_hdr[L] = take((chunks + 4) * GHDR_ROWS * 16);
		pl.off_recs[L] = take((size_t) slots * GREC_ROWS * 16);
		pl.off_res[L] = take((size_t) slots * 12 + 16);
		if(L < pl.levels) pl.off_nodes[L] = take((size_t) pl.nodes_max[L] * GNODE_ROWS * 16);
		if(off > ((size_t) 1 << 40)) return false;
	}
	pl.total = off;
	return true;
}

static uint64_t g_budget(const RenderParams &p) { return p.sw.budget_mb ? (uint64_t) p.sw.budget_mb << 20 : 6ull << 30; }

// the largest band (whole tile rows of 8 output rows) whose worst-case tables fit the budget
static bool gplan(const RenderParams &p, GPlan &pl)
{
	const uint32_t rows_all = p.out_rows;
	if(gplan_for(p, rows_all, pl) && pl.total <= g_budget(p)) return true;
	uint32_t lo = 1, hi = rows_all;
	if(!gplan_for(p, lo, pl) || pl.total > g_budget(p)) return false;
	while(hi - lo > 1)
	{
		const uint32_t mid = lo + (hi - lo) / 2;
		if(gplan_for(p, mid, pl) && pl.total <= g_budget(p)) lo = mid;
		else hi = mid;
	}
	return gplan_for(p, lo, pl);
}

bool skr_generic_supported(const RenderParams &p)
{
	if(p.n_spheres >= 65536 || p.n_tris >= (1 << 30)) return false;
	GPlan pl;
	return gplan(p, pl);
}

size_t skr_generic_scratch_bytes(const RenderParams &p)
{
	GPlan pl;
	return gplan(p, pl) ? pl.total : 0;
}

size_t skr_generic_lds_bytes(const RenderParams &p) { return ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 32; }

hipError_t skr_launch_generic(const RenderParams &p_in, hipStream_t stream, const SkrTimingHook *hook)
{
	RenderParams p = p_in;
	GPlan pl;
	if(!p.node_scratch || !gplan(p, pl)) return hipErrorInvalidValue;
	char *base = reinterpret_cast<char *>(p.node_scratch);
	uint32_t *ctr0 = reinterpret_cast<uint32_t *>(base + pl.off_ctr);
	auto lvl_ctr = [&](int L) { return ctr0 + SKR_PULL_STRIDE + G_LVL_CTR_WORDS * (size_t) L; };
	const int nsamp = p.grid_size > 0 ? p.grid_size * p.grid_size : 1;
	const size_t lds = skr_generic_lds_bytes(p);
	const uint32_t A = g_arity(p);
	const int D = pl.levels;
	hipError_t e = hipSuccess;
	for(int s = 0; s < nsamp; s++)
	{
		p.aa_index = (uint32_t) s;
		for(uint32_t row0 = 0; row0 < p.out_rows; row0 += pl.band_rows)
		{ // every band is a complete pass
			const uint32_t rows = p.out_rows - row0 < pl.band_rows ? p.out_rows - row0 : pl.band_rows;
			const bool timed = hook && s == nsamp - 1 && row0 == 0; // (the first band of the last sample: a full-size band)
			p.band_row0 = row0;
			p.band_rows = rows;
			e = hipMemsetAsync(ctr0, 0, pl.ctr_bytes, stream);
			if(e != hipSuccess) return e;
			e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctr0), (int) (rows * (uint32_t) p.width), 1, stream); // [0]: the band's roots
			if(e != hipSuccess) return e;
			if(timed) skr_hook_start(hook, stream);
			for(int L = 1; L <= D; L++)
			{
				p.g_level = (uint32_t) L;
				p.g_arity = L == 1 ? 1u : A;
				p.g_last = L == D ? 1u : 0u;
				p.nd_count = L == 1 ? ctr0 : g_prefix_host(lvl_ctr(L - 1));
				p.g_nodes_src = L == 1 ? nullptr : reinterpret_cast<const float4 *>(base + pl.off_nodes[L - 1]);
				p.rc = reinterpret_cast<float4 *>(base + pl.off_recs[L]);
				p.rc_cap = pl.cap[L];
				p.rc_ctr = lvl_ctr(L);
				p.ixh = reinterpret_cast<uint4 *>(base + pl.off_hdr[L]);
				const uint64_t wg_t = (pl.nodes_max[L - 1] * (uint64_t) p.g_arity + 255) / 256;
				const unsigned grid_t = (unsigned) (wg_t < 49152u ? wg_t : 49152u);
				hipLaunchKernelGGL(skr_gtrace_kernel, dim3(grid_t), dim3(256), lds, stream, p);
				p.g_arity = A; // (node ids of this level's hits: parent id * A + child + 1)
				p.g_nodes_dst = L < D ? reinterpret_cast<float4 *>(base + pl.off_nodes[L]) : nullptr;
				p.res_out = reinterpret_cast<float *>(base + pl.off_res[L]);
				const unsigned grid_a = SKR_P1_REGIONS * ((pl.cap[L] + 255u) / 256u);
				hipLaunchKernelGGL(skr_gactivate_kernel, dim3(grid_a), dim3(256), lds, stream, p);
			}
			for(int L = D - 1; L >= 0; L--)
			{ // sums, deepest level first; level 0 writes the pixels
				p.g_level = (uint32_t) L;
				p.g_arity = L == 0 ? 1u : A;
				p.nd_count = L == 0 ? ctr0 : g_prefix_host(lvl_ctr(L));
				p.g_nodes_src = L == 0 ? nullptr : reinterpret_cast<const float4 *>(base + pl.off_nodes[L]);
				p.ixh = reinterpret_cast<uint4 *>(base + pl.off_hdr[L + 1]);
				p.res_in = reinterpret_cast<const float *>(base + pl.off_res[L + 1]);
				p.res_out = L == 0 ? nullptr : reinterpret_cast<float *>(base + pl.off_res[L]);
				hipLaunchKernelGGL(skr_gfinalize_kernel, dim3((unsigned) ((pl.nodes_max[L] + 255) / 256)), dim3(256), lds, stream, p);
			}
			if(timed) skr_hook_stop(hook, stream);
			e = hipGetLastError();
			if(e != hipSuccess) return e;
		}
	}
	if(p.grid_size > 0) return skr_launch_resolve(p, stream);
	return hipSuccess;
}
