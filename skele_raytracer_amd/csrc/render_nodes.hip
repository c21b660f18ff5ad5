// Node pipeline: the --gillum recursion of the reference (src/raytrace.h:107-136 called from :208-216) cut at EVERY
// level of the tree, for any --depth >= 2 (main.cpp:318-329 accepts any positive depth).
//
// A *node* is a shaded sphere hit whose N children are to be traced; level 0 = the primary hits.  Rays of the last
// level (their own children are shade(depth 0) == 0, raytrace.h:142-145) are leaves.  Kernels, in launch order:
//
//   skr_primary_kernel   (render_wave.hip) primary rays + direct light; every sphere hit becomes a level-0 node (a 32-byte geometry
//                        row for the kernels that trace its children, a 32-byte shading row for the one that sums them)
//   skr_trace_kernel     one lane per sibling pair of a node: traces the two child rays (one Philox call, one (e, c)
//                        per sphere for both); per wave (64 pairs) one 48-byte header — the first record of its hits and the
//                        ballots of the children that hit: finalize finds a child's record from the population count of the
//                        ballot below its pair, and draws r1 of a miss again from the same Philox counter — and per sphere
//                        hit a 16-byte record (parent, sphere, child, the two draws) appended to one of 64 regions (ballot
//                        ranks + one atomic per wave).  The grid is capped; a workgroup strides over the blocks of 256 pairs.
//   skr_activate_kernel  (depth >= 4, and the flat schedule) one lane per record: shades the hit (:194-207) and writes it as a
//                        node of the next level, which the trace kernel then expands; forms the level's prefix sums itself
//   skr_leaf_kernel2     persistent waves, one unit = 64 records of the last-but-one level, ONE LANE PER RECORD: the 64
//                        hits are shaded full-width and stay in their lanes (origin, normal, basis, RNG key), then for
//                        every sibling pair j the lanes trace children 2j, 2j+1 of their own node; leaf hits go through
//                        an LDS ring and are shaded 64 at a time; contributions wait in a ring of four LDS slot
//                        windows (one round each) until they are added, strictly in child order (:130), to the
//                        lane's running sum.  Depth 2: the units are the level-0 nodes themselves and the pixels are
//                        written here.
//   skr_shade_leaf_kernel (flat schedule: small launches, where the persistent kernel has one or two units per wave and a long
//                        tail) the last level too is traced into records by skr_trace_kernel; this kernel shades them one per lane
//   skr_finalize_kernel2 one lane per node, deepest level first: the N terms in child order from the wave headers,
//                        (direct/pi + 2 indirect) * kd (:213), times r1/pdf into the parent's level — or the pixel
//
// Every float operation and every order of summation is the reference's (DESIGN.md "Arithmetic spec"); only the
// schedule differs, so the image is bit-identical to the other kernel variants and to the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wave_common.h"

namespace {

// One header per trace wave (64 sibling pairs) tells finalize what became of their children: [0] = {first record of the wave's
// hits, -, -, -}, [1] = the ballots of the even and the odd children that hit a sphere (2 x 64 bits: a child's record is the
// base + the number of set bits below its pair, the odd children's records behind the even ones'), [2] (triangle scenes only) =
// the ballots of the children a triangle took.  r1 of a miss comes from the same Philox call the trace made.
constexpr uint32_t PC_HIT0 = 1u, PC_HIT1 = 2u, PC_BLACK0 = 4u, PC_BLACK1 = 8u; // (finalize's per-pair flags, formed from the ballots)
constexpr int IXH_ROWS = 3; // uint4 per header
constexpr int NQ_CAP = 176;                 // leaf-hit ring: <= NQ_PRE left over when a round starts + the 128 hits it can add
constexpr int NQ_PRE = NQ_CAP - 128;        // more than that waiting at the start of a round: one (>= 3/4 full) batch is shaded first
constexpr int NQ_F = 5;                     // dwords per entry: d.xyz, ids, r1 (b and D of utils.h:116-118 are formed again from d when the hit is shaded)
constexpr int NWIN = 4;                     // slot windows (rounds whose contributions may still be waiting for their hits' shading)
constexpr int WIN_FLOATS = 2 * 3 * 64;      // [child 0|1][component][lane]
constexpr int LEAF2_WAVE_FLOATS = NQ_CAP * NQ_F + NWIN * WIN_FLOATS;

// raytrace.h:171-186 + :189-192 / :221-224 for one traced child whose closest sphere is s: true = the child is a sphere
// hit to be shaded; otherwise `black` says whether a triangle took it (else it left the scene)
SKR_DEV bool classify_child(const SceneView &sv, f3 co, f3 d, float two_a, float four_a, const BestState &s, bool &black)
{
	black = false;
	if(sv.nt > 0)
	{ // the triangle walk needs the sphere's exact t to compare against
		const float tmin = (s.best >= 0) ? near_root_exact(two_a, s.b, s.D) : __builtin_inff();
		black = any_triangle_closer(sv, RayConst{co, d, two_a, four_a}, tmin);
	}
	return !black && s.best >= 0;
}

} // namespace

// =====================================================================================================================
// trace: one lane per sibling pair of a node
// =====================================================================================================================
#ifndef SKR_TRACE_LOOP
#define SKR_TRACE_LOOP 1 // A/B builds: 0 = one block of 256 pairs per workgroup, the grid sized for the worst case
#endif
#if SKR_TRACE_LOOP
#define SKR_TRACE_STRIDE gridDim.x
#else
#define SKR_TRACE_STRIDE 0x40000000u
#undef SKR_TRACE_GRID_MAX
#define SKR_TRACE_GRID_MAX 0x7fffffffu
#endif
#ifndef SKR_TRACE_WAVES
#define SKR_TRACE_WAVES 0 // waves per SIMD the trace kernel is held to (0: whatever its registers allow) — A/B builds
#endif
#if SKR_TRACE_WAVES
#define SKR_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(SKR_TRACE_WAVES, SKR_TRACE_WAVES)))
#else
#define SKR_TRACE_ATTR
#endif
template <bool TRIS>
__global__ __launch_bounds__(256) SKR_TRACE_ATTR void skr_trace_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const uint32_t N = (uint32_t) p.num_path_traces, PP = (N + 1u) >> 1; // children, sibling pairs per node
	const uint64_t n_pairs = (uint64_t) *p.nd_count * PP;
	if((uint64_t) blockIdx.x * 256u >= n_pairs) return; // (uniform per workgroup)
	const SceneView sv = stage_scene(p, lds4, TRIS);
	const int tid = threadIdx.x, lane = tid & 63;
	Counters cn{0, 0, 0};
	// (the grid is sized for the worst case — every ray of the level above a hit — and capped: a workgroup takes every
	// gridDim.x-th block of 256 pairs, so a launch far below the worst case does not pay for its empty workgroups)
	for(uint32_t blk = blockIdx.x; (uint64_t) blk * 256u < n_pairs; blk += SKR_TRACE_STRIDE)
	{
	const uint32_t chunk = blk * 4u + (uint32_t) (tid >> 6);
	const uint64_t tp = (uint64_t) chunk * 64u + (uint32_t) lane;
	const bool valid = tp < n_pairs;
	const uint32_t node = valid ? (uint32_t) (tp / PP) : 0u, j = valid ? (uint32_t) (tp - (uint64_t) node * PP) : 0u;
	const bool second = valid && 2u * j + 1u < N;
	bool hit0 = false, hit1 = false;
	float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0;
	bool black0 = false, black1 = false;
	if(valid)
	{
		const float4 *row = p.nd_src + (size_t) node * 2;
		const float4 a0 = row[0], a1 = row[1];
		const f3 co = mk3(a0.x, a0.y, a0.z), Nn = mk3(a0.w, a1.x, a1.y);
		const uint32_t pixel = __float_as_uint(a1.z), node_id = __float_as_uint(a1.w);
		f3 nt, nb;
		tangent_basis(Nn, nt, nb);
		uint32_t rnd[4];
		philox4x32(pixel, p.aa_index, node_id, j, p.seed_lo, p.seed_hi, rnd);
		const float q1a = u31_to_unit(rnd[0]), q2a = u31_to_unit(rnd[1]), q1b = u31_to_unit(rnd[2]), q2b = u31_to_unit(rnd[3]);
		const DirPair dp = gi_direction_pair(q1a, q2a, q1b, q2b, Nn, nt, nb);
		const f3 d0 = dp.d0, d1 = dp.d1;
		cn.rays += second ? 2u : 1u;
		const RayPair rp = make_pair(d0, d1);
		BestState s0, s1;
		closest_pair(sv, co, d0, d1, second, rp, s0, s1);
		hit0 = classify_child(sv, co, d0, rp.two_a.x, rp.four_a.x, s0, black0);
		rec0 = make_float4(__uint_as_float(node), __uint_as_float((uint32_t) (s0.best & 0xffff) | ((2u * j) << 16)), q1a, q2a);
		if(second)
		{
			hit1 = classify_child(sv, co, d1, rp.two_a.y, rp.four_a.y, s1, black1);
			rec1 = make_float4(__uint_as_float(node), __uint_as_float((uint32_t) (s1.best & 0xffff) | ((2u * j + 1u) << 16)), q1b, q2b);
		}
	}
	// append the wave's hits to its region: rank by ballot, one atomic per wave
	const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
	const uint32_t region = chunk & (SKR_P1_REGIONS - 1u);
	const uint32_t n0h = (uint32_t) __popcll(m0), n1h = (uint32_t) __popcll(m1);
	uint32_t base = 0;
	if(n0h + n1h != 0u)
	{
		if(lane == 0) base = atomicAdd(lc_count(p.rc_ctr, region), n0h + n1h);
		base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
	}
	const uint32_t rec_base = region * p.rc_cap + base; // the even children's hits first, then the odd ones
	const uint32_t rank0 = (uint32_t) lanes_below(m0), rank1 = (uint32_t) lanes_below(m1);
	if(hit0) p.rc[rec_base + rank0] = rec0;
	if(hit1) p.rc[rec_base + n0h + rank1] = rec1;
	if(TRIS)
	{
		const unsigned long long k0 = __ballot(black0), k1 = __ballot(black1);
		if(lane == 0) p.ixh[IXH_ROWS * (size_t) chunk + 2] = make_uint4((uint32_t) k0, (uint32_t) (k0 >> 32), (uint32_t) k1, (uint32_t) (k1 >> 32));
	}
	if(lane == 0)
	{
		p.ixh[IXH_ROWS * (size_t) chunk] = make_uint4(rec_base, n0h, 0u, 0u);
		p.ixh[IXH_ROWS * (size_t) chunk + 1] = make_uint4((uint32_t) m0, (uint32_t) (m0 >> 32), (uint32_t) m1, (uint32_t) (m1 >> 32));
	}
	}
	add_counters(p, cn, (uint32_t) blockIdx.x * 4u + (uint32_t) (tid >> 6), lane);
}

// =====================================================================================================================
// activate (depth >= 4): record -> node of the next level
// =====================================================================================================================
// the hit a record describes, shaded: raytrace.h:194-207 (+ the child origin of :128)
struct Activated {
	f3 co, N, direct;
	uint32_t pixel, node_id, sph, child;
	float r1;
};

SKR_DEV Activated activate_record(const SceneView &sv, const RenderParams &p, bool act, uint32_t rec, Counters &cn)
{
	Activated a;
	a.co = a.N = mk3(0, 0, 1);
	a.direct = mk3(0, 0, 0);
	a.pixel = a.node_id = a.sph = a.child = 0;
	a.r1 = 0.0f;
	if(act)
	{
		typedef float v4f __attribute__((ext_vector_type(4)));
		const v4f n0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p.rc + rec)); // streamed once
		const uint32_t parent = __float_as_uint(n0.x), sc = __float_as_uint(n0.y);
		a.r1 = n0.z;
		a.sph = sc & 0xffffu;
		a.child = sc >> 16;
		const float4 *row = p.nd_src + (size_t) parent * 2; // (one aligned 32-byte sector per parent)
		const float4 p0 = row[0], p1 = row[1];
		const f3 co0 = mk3(p0.x, p0.y, p0.z), N0 = mk3(p0.w, p1.x, p1.y);
		a.pixel = __float_as_uint(p1.z);
		const uint32_t pnode = __float_as_uint(p1.w); // (0 at level 0)
		a.node_id = pnode * (uint32_t) p.num_path_traces + a.child + 1u; // DESIGN.md "RNG": child c of node n
		// The ray that found this hit, formed again from its two draws exactly as the trace kernel formed it (raytrace.h:119-125:
		// same operations on the same values, so the same direction), and utils.h:115-118 for the sphere it hit: 16 bytes per
		// record instead of 32
		f3 nt0, nb0;
		tangent_basis(N0, nt0, nb0);
		const f3 d = gi_direction(a.r1, n0.w, N0, nt0, nb0);
		const float4 g = sv.geom[a.sph];
		const f3 ec = co0 - ld3(g);
		const float aa = dot3(d, d);
		const float b = 2 * dot3(d, ec);
		const float D = b * b - (4 * aa) * (dot3(ec, ec) - g.w);
		const float two_a = 2 * aa;
		const float t = near_root_exact(two_a, b, D);
		const f3 P = co0 + d * t;
		a.N = normalize3(P - ld3(sv.geom[a.sph]));
		cn.hits++;
		a.direct = direct_light<false>(sv, p, (int) a.sph, P, a.N, cn);
		a.co = add_scalar(P, 0.00001f);
	}
	return a;
}

#ifndef SKR_SHADE_WAVES
#define SKR_SHADE_WAVES 5 // waves per SIMD the record-shading kernels are held to (0: whatever their registers allow: 102 VGPRs, 4 waves; at 5 a 1/8 headline share takes 0.290 instead of 0.298 ms, at 6 0.296)
#endif
#if SKR_SHADE_WAVES
#define SKR_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(SKR_SHADE_WAVES, SKR_SHADE_WAVES)))
#else
#define SKR_SHADE_ATTR
#endif
template <bool TRIS>
__global__ __launch_bounds__(256) SKR_SHADE_ATTR void skr_activate_kernel(const RenderParams p)
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
	const SceneView sv = stage_scene(p, lds4, TRIS);
	const uint32_t pos = pos0 + threadIdx.x;
	const bool act = pos < cnt;
	const uint32_t rec = region * p.rc_cap + pos;
	Counters cn{0, 0, 0};
	const Activated a = activate_record(sv, p, act, rec, cn);
	if(act)
	{
		const size_t n = (size_t) (s_pre[region] + pos);
		float4 *grow = p.nd_dst + n * 2, *srow = p.ns_dst + n * 2;
		grow[0] = make_float4(a.co.x, a.co.y, a.co.z, a.N.x);
		grow[1] = make_float4(a.N.y, a.N.z, __uint_as_float(a.pixel), __uint_as_float(a.node_id));
		srow[0] = make_float4(a.direct.x, a.direct.y, a.direct.z, __uint_as_float(a.sph));
		srow[1] = make_float4(a.r1, __uint_as_float(rec), __uint_as_float(a.pixel), __uint_as_float(a.node_id));
	}
	add_counters(p, cn, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Flat schedule (small launches): the hit records of the LAST level are shaded one per lane, no children (they are shade(depth 0)
// == 0, raytrace.h:142-145): raytrace.h:194-213 with indirect = (0,0,0)/N, then the parent's accumulation term (:130) — the
// arithmetic of leaf_batch() below, on records instead of ring entries.  Dense numbering through the level's prefix sums
// (region_prefix): position i belongs to the region whose prefix range holds it.
template <bool TRIS>
__global__ __launch_bounds__(256) SKR_SHADE_ATTR void skr_shade_leaf_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	__shared__ uint32_t prefix[65];
	region_prefix(p, prefix, false);
	const uint32_t total = prefix[64];
	if((uint32_t) blockIdx.x * 256u >= total) return; // (uniform per workgroup)
	const SceneView sv = stage_scene(p, lds4, TRIS);
	Counters cn{0, 0, 0};
	for(uint32_t base = (uint32_t) blockIdx.x * 256u; base < total; base += gridDim.x * 256u)
	{
		const uint32_t i = base + threadIdx.x;
		const bool act = i < total;
		uint32_t lo = 0, hi = SKR_P1_REGIONS; // the region r with prefix[r] <= i < prefix[r + 1]
		while(hi - lo > 1u)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if(prefix[mid] <= (act ? i : 0u)) lo = mid;
			else hi = mid;
		}
		const uint32_t rec = lo * p.rc_cap + ((act ? i : 0u) - prefix[lo]);
		const Activated a = activate_record(sv, p, act, rec, cn);
		if(act)
		{
			const f3 tot = mk3(0, 0, 0); // (0,0,0) / N with N >= 1 (skr_nodes_supported): +0 in every component, no division needed
			const f3 colour = (div3_const(a.direct, SKR_DIV_PI) + tot * 2.0f) * ld3(sv.kd[a.sph]);
			store3(p.res_out + (size_t) rec * 3, div3_const(colour * a.r1, SKR_DIV_PDF));
		}
	}
	add_counters(p, cn, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63);
}

// =====================================================================================================================
// leaf: one lane per record (FIRST: per level-0 node), its N leaf rays traced two per round
// =====================================================================================================================
namespace {

struct Ring {
	float *base; // SoA by field, NQ_CAP entries
	int head, count; // wave-uniform
};

SKR_DEV void ring_push(Ring &q, bool pred, f3 d, uint32_t ids, float r1)
{
	const unsigned long long m = __ballot(pred);
	if(pred)
	{
		int e = q.head + q.count + lanes_below(m); // head < cap, count + rank < cap
		e -= (e >= NQ_CAP) ? NQ_CAP : 0;
		float *r = q.base + e;
		r[0 * NQ_CAP] = d.x;
		r[1 * NQ_CAP] = d.y;
		r[2 * NQ_CAP] = d.z;
		r[3 * NQ_CAP] = __uint_as_float(ids);
		r[4 * NQ_CAP] = r1;
	}
	q.count = uni(q.count + (int) __popcll(m));
}

// Shade the m <= 64 oldest queued leaf hits, one per lane: raytrace.h:194-213 with indirect = (0,0,0)/N (the leaf's own
// children are shade(depth 0)), then its parent's accumulation term (:130) into the slot window of its round.
// `co` = the child-ray origin of the node THIS lane holds (the hits' parents are lanes of this wave).
SKR_DEV void leaf_batch(const SceneView &sv, const RenderParams &p, Ring &q, float *slots, f3 co, int lane, int m, Counters &cn)
{
	wave_lds_fence();
	const bool act = lane < m;
	DIAG_WAVE(9, 1);
	DIAG_WAVE(10, m);
	int e = q.head + (act ? lane : 0);
	e -= (e >= NQ_CAP) ? NQ_CAP : 0;
	const float *r = q.base + e;
	const f3 d = mk3(r[0 * NQ_CAP], r[1 * NQ_CAP], r[2 * NQ_CAP]);
	const float r1 = r[4 * NQ_CAP];
	const uint32_t ids = __float_as_uint(r[3 * NQ_CAP]);
	const int kl = (int) ((ids >> 16) & 63u);
	const f3 co_k = shfl3(co, act ? kl : 0);
	if(act)
	{
		const int sph = (int) (ids & 0xffffu);
		// utils.h:115-118 for the winning sphere, as the trace formed them (same operations on the same values): b, D
		const float4 g = sv.geom[sph];
		const f3 ec = co_k - ld3(g);
		const float a = dot3(d, d);
		const float b = 2 * dot3(d, ec);
		const float D = b * b - (4 * a) * (dot3(ec, ec) - g.w);
		const float two_a = 2 * a;
		const float t = near_root_exact(two_a, b, D);
		const f3 P = co_k + d * t;
		const f3 Nn = normalize3(P - ld3(sv.geom[sph]));
		cn.hits++;
		const f3 direct = direct_light<false>(sv, p, sph, P, Nn, cn);
		const f3 total = mk3(0, 0, 0); // (0,0,0) / N with N >= 1 (skr_nodes_supported): +0 in every component, no division needed
		const f3 colour = (div3_const(direct, SKR_DIV_PI) + total * 2.0f) * ld3(sv.kd[sph]);
		const f3 c = div3_const(colour * r1, SKR_DIV_PDF);
		float *s = slots + (int) ((ids >> 23) & (NWIN - 1)) * WIN_FLOATS + (int) ((ids >> 22) & 1u) * 192 + kl; // [round][child][component][lane]
		s[0] = c.x;
		s[64] = c.y;
		s[128] = c.z;
	}
	int nh = q.head + m;
	nh -= (nh >= NQ_CAP) ? NQ_CAP : 0;
	q.head = uni(nh);
	q.count = uni(q.count - m);
	wave_lds_fence();
}

} // namespace

#if defined(SKR_DIAG) && SKR_DIAG
extern "C" void skr_diag_read_nodes(unsigned long long *out, int reset)
{ // this translation unit's copy of the 32 event counters (device_math.h), summed over their 64 shards
	unsigned long long h[32 * 64];
	(void) hipDeviceSynchronize();
	(void) hipMemcpyFromSymbol(h, HIP_SYMBOL(skr_diag), sizeof(h));
	for(int i = 0; i < 32; i++)
	{
		out[i] = 0;
		for(int k = 0; k < 64; k++) out[i] += h[i * 64 + k];
	}
	if(reset)
	{
		memset(h, 0, sizeof(h));
		(void) hipMemcpyToSymbol(HIP_SYMBOL(skr_diag), h, sizeof(h));
	}
}
#endif
#if defined(SKR_TIMELINE) && SKR_TIMELINE
// per-wave timeline of the leaf kernel (tools/leaf_timeline.py): entry, first unit in hand, last unit done, exit on the
// device-wide 100 MHz clock, units done, and the wave's shader-clock cycles between entry and exit (the clock it ran at)
static __device__ unsigned long long skr_leaf2_times[6 * 4096];
extern "C" void skr_leaf2_times_read(unsigned long long *out)
{
	(void) hipDeviceSynchronize();
	(void) hipMemcpyFromSymbol(out, HIP_SYMBOL(skr_leaf2_times), sizeof(unsigned long long) * 6 * 4096);
}
#endif
#ifndef SKR_LEAF2_OCC
#define SKR_LEAF2_OCC 4 // waves per SIMD the register allocation aims at (A/B builds)
#endif
template <bool TRIS, bool FIRST>
__global__ __launch_bounds__(256, SKR_LEAF2_OCC) void skr_leaf_kernel2(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const SceneView sv = stage_scene(p, lds4, TRIS); // the only workgroup barrier
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const uint32_t g = (uint32_t) blockIdx.x * 4u + (uint32_t) wave;
	float *wbase = reinterpret_cast<float *>(lds4 + 4 * p.n_spheres + 1 + 2 * p.n_lights) + wave * LEAF2_WAVE_FLOATS;
	Ring q{wbase, 0, 0};
	float *slots = wbase + NQ_CAP * NQ_F;
	const int N = p.num_path_traces, PP = (N + 1) >> 1;
	uint32_t region = g & (SKR_P1_REGIONS - 1u);
	unsigned long long dead = 0; // regions this wave has seen exhausted
	// FIRST: the level-0 nodes form virtual regions — unit u of the node array belongs to region u mod 64
	const uint32_t n_first = FIRST ? *p.nd_count : 0u;
	const uint32_t US = 64u; // records per unit, one lane each
	const int pl = lane, cs = 64, R = PP;
	const uint32_t units_first = (n_first + US - 1u) / US;
	Counters cn{0, 0, 0};
	STAMP_DECL; // (diagnostic builds: 0 pull, 1 activation, 2 trace, 3 leaf shading, 4 pushes + window sums, 5 unit end)
#if defined(SKR_TIMELINE) && SKR_TIMELINE
	const unsigned long long wt_start = wall_clock64(), wc_start = __builtin_readcyclecounter();
	unsigned long long wt_first = 0, wt_last = 0;
	uint32_t wt_units = 0;
#endif
	for(;;)
	{
		uint32_t first = 0;
		int m = 0;
		bool got = false;
		for(;;)
		{ // pull the next unit (wave-uniform): this region's, or another region's once this one is exhausted
			uint32_t k = 0;
			if(lane == 0) k = atomicAdd(lc_taken(p.rc_ctr, region), 1u);
			k = (uint32_t) __builtin_amdgcn_readfirstlane((int) k);
			if(FIRST)
			{
				const uint32_t u = k * SKR_P1_REGIONS + region;
				if(u < units_first)
				{
					first = u * US;
					m = (int) (n_first - first < US ? n_first - first : US);
					got = true;
				}
			}
			else
			{
				const uint32_t cnt = *lc_count(p.rc_ctr, region);
				const uint32_t units = (cnt + US - 1u) / US;
				if(k < units)
				{
					first = k * US;
					m = (int) (cnt - first < US ? cnt - first : US);
					got = true;
				}
			}
			if(got) break;
			// This region is exhausted.  The exhausted regions are published in one 64-bit mask: the atomic OR that adds this
			// region returns everybody else's findings, and the wave goes to the first region after its own that nobody has
			// seen dry — or leaves when there is none.
			unsigned long long seen = 0;
			if(lane == 0) seen = atomicOr(lc_dead(p.rc_ctr), 1ull << region);
			const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) seen), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (seen >> 32));
			dead |= ((unsigned long long) hi << 32 | lo) | (1ull << region);
			const unsigned long long live = ~dead;
			if(live == 0ull) break;
			const unsigned long long after = region == 63u ? 0ull : (live >> (region + 1u)) << (region + 1u); // regions above this one
			region = (uint32_t) __builtin_ctzll(after ? after : live);
		}
		if(!got) break;
		STAMP(0);
#if defined(SKR_TIMELINE) && SKR_TIMELINE
		if(wt_units == 0) wt_first = wall_clock64();
#endif

		// ---- the unit's records become the nodes their lanes hold
		const bool act0 = lane < m; // the lane that shades record `lane` of the unit and writes its result
		f3 co = mk3(0, 0, 1), Nn = mk3(0, 0, 1), direct1 = mk3(0, 0, 0);
		uint32_t pixel = 0, node_id = 0, out_idx = 0; // out_idx: output pixel (FIRST) or this record's index
		if(FIRST)
		{
			if(act0)
			{
				const size_t nn = (size_t) (first + (uint32_t) lane);
				const float4 a0 = p.nd_src[nn * 2], a1 = p.nd_src[nn * 2 + 1], b0 = p.ns_src[nn * 2], b1 = p.ns_src[nn * 2 + 1];
				co = mk3(a0.x, a0.y, a0.z);
				Nn = mk3(a0.w, a1.x, a1.y);
				direct1 = mk3(b0.x, b0.y, b0.z);
				pixel = __float_as_uint(a1.z);
				out_idx = __float_as_uint(b1.y);
			}
		}
		else
		{
			out_idx = region * p.rc_cap + first + (uint32_t) lane;
			const Activated a = activate_record(sv, p, act0, out_idx, cn);
			co = a.co;
			Nn = a.N;
			direct1 = a.direct;
			pixel = a.pixel;
			node_id = a.node_id;
		}
		const bool act = act0;
		f3 nt, nb;
		tangent_basis(Nn, nt, nb);
		// (r1 and the sphere are only needed when the unit is finished: they are read again from the record / node row then)
		STAMP(1);
		// ---- rounds: children 2j, 2j+1 of every lane's node
		f3 acc = mk3(0, 0, 0);
		for(int jj = 0; jj < R; jj++)
		{
			const int j = jj;                         // this lane's sibling pair: children 2j, 2j + 1
			const bool second = 2 * j + 1 < N;        // (wave-uniform)
			float *wrow = slots + (jj & (NWIN - 1)) * WIN_FLOATS + pl; // the round's window: [child][component][64 lanes]
			if(jj >= NWIN)
			{ // the window is reused: add round jj - NWIN (its hits were shaded at the end of the previous round), child order
				wave_lds_fence();
				for(int c = 0; c < 2; c++) acc = acc + mk3(wrow[(3 * c) * cs], wrow[(3 * c + 1) * cs], wrow[(3 * c + 2) * cs]); // (rounds before the last have both children)
				wave_lds_fence();
			}
			float *win = wrow; // (the ring has room for the 128 hits this round can add: the previous round left at most NQ_PRE waiting)
			bool hit0 = false, hit1 = false;
			f3 d0 = mk3(0, 0, 1), d1 = mk3(0, 0, 1);
			float q1a = 0, q1b = 0;
			int s0b = 0, s1b = 0;
			if(act)
			{
				uint32_t rnd[4];
				philox4x32(pixel, p.aa_index, node_id, (uint32_t) j, p.seed_lo, p.seed_hi, rnd);
				q1a = u31_to_unit(rnd[0]);
				q1b = u31_to_unit(rnd[2]);
				const float q2a = u31_to_unit(rnd[1]), q2b = u31_to_unit(rnd[3]);
				f3 nq = Nn;
				asm volatile("" : "+v"(nq.x), "+v"(nq.y), "+v"(nq.z)); // (keeps the next line inside the loop)
				const f3 nbr = cross3(nq, nt); // (utils.h:164, formed again per round: three registers less to carry)
				const DirPair dp = gi_direction_pair(q1a, q2a, q1b, q2b, Nn, nt, nbr);
				d0 = dp.d0;
				d1 = dp.d1;
				cn.rays += second ? 2u : 1u;
				const RayPair rp = make_pair(d0, d1);
				BestState s0, s1;
				closest_pair(sv, co, d0, d1, second, rp, s0, s1);
				bool black;
				hit0 = classify_child(sv, co, d0, rp.two_a.x, rp.four_a.x, s0, black);
				if(!hit0)
				{ // raytrace.h:189-192 / :221-224, then :130: total += (r1 * colour) / pdf
					const f3 c = div3_const((black ? mk3(0, 0, 0) : p.background) * q1a, SKR_DIV_PDF);
					win[0] = c.x;
					win[cs] = c.y;
					win[2 * cs] = c.z;
				}
				s0b = s0.best;
				if(second)
				{
					hit1 = classify_child(sv, co, d1, rp.two_a.y, rp.four_a.y, s1, black);
					if(!hit1)
					{
						const f3 c = div3_const((black ? mk3(0, 0, 0) : p.background) * q1b, SKR_DIV_PDF);
						win[3 * cs] = c.x;
						win[4 * cs] = c.y;
						win[5 * cs] = c.z;
					}
					s1b = s1.best;
				}
			}
			STAMP(2);
			const bool last = jj + 1 == R;
			const uint32_t idj = ((uint32_t) lane << 16) | ((uint32_t) jj << 23);
			ring_push(q, hit0, d0, (uint32_t) (s0b & 0xffff) | idj, q1a);
			ring_push(q, hit1, d1, (uint32_t) (s1b & 0xffff) | idj | (1u << 22), q1b);
			for(;;)
			{
				bool go = q.count >= 64 || (!last && q.count > NQ_PRE); // a full batch; or room for the next round's 128 hits (one >= 3/4 full batch)
				if(!go && q.count > 0)
				{ // after the round: everything, if it was the last one; otherwise the hits of the round whose window the NEXT round reuses
					go = last;
					if(!go && jj + 1 >= NWIN)
					{
						wave_lds_fence();
						const uint32_t head_ids = (uint32_t) uni((int) __float_as_uint(q.base[3 * NQ_CAP + q.head]));
						go = (int) (head_ids >> 23) <= jj + 1 - NWIN;
					}
				}
				if(!go) break;
				STAMP(4);
				leaf_batch(sv, p, q, slots, co, lane, q.count < 64 ? q.count : 64, cn);
				STAMP(3);
			}
			STAMP(4);
		}
		// ---- the rounds still in their windows, in order
		wave_lds_fence();
		for(int jr = (R > NWIN ? R - NWIN : 0); jr < R; jr++)
		{
			const float *wrow = slots + (jr & (NWIN - 1)) * WIN_FLOATS + pl;
			for(int c = 0; c < 2; c++)
				if(2 * jr + c < N) acc = acc + mk3(wrow[(3 * c) * cs], wrow[(3 * c + 1) * cs], wrow[(3 * c + 2) * cs]);
		}
		wave_lds_fence();
		if(act0)
		{ // raytrace.h:133 + :213
			f3 direct;
			uint32_t sph;
			float r1 = 0.0f;
			if(FIRST)
			{
				direct = direct1;
				sph = __float_as_uint(p.ns_src[(size_t) (first + (uint32_t) lane) * 2].w);
			}
			else
			{
				direct = direct1;
				const float4 r1row = p.rc[out_idx]; // (asked for here, behind the window sums: asking earlier was measured and changes nothing, 1.570 / 1.569 ms)
				r1 = r1row.z;
				sph = __float_as_uint(r1row.y) & 0xffffu;
			}
			const f3 total = acc / (float) N;
			const f3 colour = (div3_const(direct, SKR_DIV_PI) + total * 2.0f) * ld3(sv.kd[sph]);
			if(FIRST) emit_sample(p, out_idx, colour);
			else store3(p.res_out + (size_t) out_idx * 3, div3_const(colour * r1, SKR_DIV_PDF)); // :130, into the parent's sum
		}
		STAMP(5);
#if defined(SKR_TIMELINE) && SKR_TIMELINE
		wt_units++;
		wt_last = wall_clock64();
#endif
	}
#if defined(SKR_TIMELINE) && SKR_TIMELINE
	if(lane == 0 && g < 4096u)
	{
		skr_leaf2_times[6 * g] = wt_start;
		skr_leaf2_times[6 * g + 1] = wt_first;
		skr_leaf2_times[6 * g + 2] = wt_last;
		skr_leaf2_times[6 * g + 3] = wall_clock64();
		skr_leaf2_times[6 * g + 4] = wt_units;
		skr_leaf2_times[6 * g + 5] = __builtin_readcyclecounter() - wc_start;
	}
#endif
#if defined(SKR_STAMPS) && SKR_STAMPS
	if(p.counters && lane == 0)
		for(int k = 0; k < 8; k++) atomicAdd(&p.counters[4u * SKR_COUNTER_SHARDS + k], st_acc[k]);
#endif
	add_counters(p, cn, g, lane);
}

// =====================================================================================================================
// finalize: one lane per node
// =====================================================================================================================
__global__ __launch_bounds__(256) void skr_finalize_kernel2(const RenderParams p)
{
	const uint32_t n = *p.nd_count;
	const uint32_t node = (uint32_t) blockIdx.x * 256u + threadIdx.x;
	if((uint32_t) blockIdx.x * 256u >= n) return;
	if(node >= n) return;
	const int N = p.num_path_traces, PP = (N + 1) >> 1;
	const float4 *row = p.ns_src + (size_t) node * 2; // the shading row is all this kernel reads of a node
	const float4 b0 = row[0], b1 = row[1];
	const uint32_t pixel = __float_as_uint(b1.z), node_id = __float_as_uint(b1.w);
	f3 total = mk3(0, 0, 0);
	for(int j0 = 0; j0 < PP; j0 += 4)
	{ // 4 sibling pairs (8 children) per trip: codes, headers and every gather issued before the first add
		uint32_t code[4], rec[8];
		f3 v[8];
#pragma unroll
		for(int k = 0; k < 4; k++)
		{
			const uint64_t tp = (uint64_t) node * (uint32_t) PP + (uint32_t) (j0 + k);
			code[k] = 0;
			rec[2 * k] = rec[2 * k + 1] = 0;
			if(j0 + k < PP)
			{
				const uint4 *h = p.ixh + IXH_ROWS * (size_t) (tp >> 6);
				const uint4 h0 = h[0], h1 = h[1];
				const unsigned long long m0 = (unsigned long long) h1.y << 32 | h1.x, m1 = (unsigned long long) h1.w << 32 | h1.z;
				const uint32_t bit = (uint32_t) tp & 63u;
				const unsigned long long below = (1ull << bit) - 1ull;
				code[k] = (uint32_t) ((m0 >> bit) & 1ull) * PC_HIT0 | (uint32_t) ((m1 >> bit) & 1ull) * PC_HIT1;
				if(p.n_tris > 0)
				{
					const uint4 h2 = h[2];
					const unsigned long long k0 = (unsigned long long) h2.y << 32 | h2.x, k1 = (unsigned long long) h2.w << 32 | h2.z;
					code[k] |= (uint32_t) ((k0 >> bit) & 1ull) * PC_BLACK0 | (uint32_t) ((k1 >> bit) & 1ull) * PC_BLACK1;
				}
				rec[2 * k] = h0.x + (uint32_t) __popcll(m0 & below);
				rec[2 * k + 1] = h0.x + h0.y + (uint32_t) __popcll(m1 & below);
			}
		}
#pragma unroll
		for(int k = 0; k < 8; k++)
		{
			v[k] = mk3(0, 0, 0);
			if(code[k >> 1] & ((k & 1) ? PC_HIT1 : PC_HIT0))
			{
				const float *r = p.res_in + (size_t) rec[k] * 3;
				v[k] = mk3(r[0], r[1], r[2]);
			}
		}
#pragma unroll
		for(int k = 0; k < 4; k++)
		{
			if(j0 + k < PP)
			{
				uint32_t rnd[4]; // the draws the trace kernel made for this pair: r1 of children 2j, 2j+1 (DESIGN.md "RNG")
				philox4x32(pixel, p.aa_index, node_id, (uint32_t) (j0 + k), p.seed_lo, p.seed_hi, rnd);
#pragma unroll
				for(int c = 0; c < 2; c++)
				{
					if(2 * (j0 + k) + c < N)
					{
						f3 term = v[2 * k + c];
						if(!(code[k] & (c ? PC_HIT1 : PC_HIT0)))
						{ // raytrace.h:189-192 / :221-224, then :130: (r1 * colour) / pdf; a triangle's (0 * r1) / pdf is +0
							const float r1 = u31_to_unit(rnd[2 * c]);
							term = (code[k] & (c ? PC_BLACK1 : PC_BLACK0)) ? mk3(0, 0, 0) : div3_const(p.background * r1, SKR_DIV_PDF);
						}
						total = total + term;
					}
				}
			}
		}
	}
	const f3 direct = mk3(b0.x, b0.y, b0.z);
	const uint32_t sph = __float_as_uint(b0.w);
	total = total / (float) N;
	const f3 colour = (div3_const(direct, SKR_DIV_PI) + total * 2.0f) * ld3(p.sph_kd[sph]); // raytrace.h:213
	if(p.nd_src_level0) emit_sample(p, __float_as_uint(b1.y), colour);
	else store3(p.res_out + (size_t) __float_as_uint(b1.y) * 3, div3_const(colour * b1.x, SKR_DIV_PDF));
}

// =====================================================================================================================
// host side: the plan of one band (table sizes for the worst case: every pixel a node, every child a hit), selection, launch
// =====================================================================================================================
hipError_t skr_launch_primary(const RenderParams &p, dim3 grid, size_t lds, hipStream_t stream); // render_wave.hip
hipError_t skr_launch_resolve(const RenderParams &p, hipStream_t stream);

constexpr int SKR_NODE_LEVELS_MAX = 33;
#ifndef SKR_TRACE_GRID_MAX
#define SKR_TRACE_GRID_MAX 49152u // workgroups of skr_trace_kernel at most (it strides over the blocks of 256 pairs)
#endif
#ifndef SKR_FLAT_BELOW
#define SKR_FLAT_BELOW 1.2e7 // worst-case records of the last-but-one level (W x rows x N^(depth-2): what the persistent kernel cuts into units of 64 for its 4096 waves) below which a launch takes the flat schedule — a quarter of the headline frame: 0.83e7, a half: 1.66e7: tools/ab_nodes.py, tools/ab_flat.py, DESIGN.md 5.0n
#endif
struct NodePlan {
	bool flat = false;       // the flat schedule (below): the leaves' hits are a record level of their own
	int levels = 0;          // node / record levels 0 .. max_depth - 2 (flat: .. max_depth - 1)
	uint32_t band_nblk = 0;  // 16x16 pixel blocks per band
	uint64_t nodes_max[SKR_NODE_LEVELS_MAX] = {};
	uint32_t cap[SKR_NODE_LEVELS_MAX] = {};
	size_t off_nodes[SKR_NODE_LEVELS_MAX] = {}, off_shade[SKR_NODE_LEVELS_MAX] = {}, off_recs[SKR_NODE_LEVELS_MAX] = {}, off_res[SKR_NODE_LEVELS_MAX] = {}, off_ixh[SKR_NODE_LEVELS_MAX] = {};
	size_t off_ctr = 0, ctr_bytes = 0, total = 0, banded = 0;
};
static uint32_t *lc_prefix_host(uint32_t *ctr) { return ctr + SKR_PULL_STRIDE * (2u * SKR_P1_REGIONS + 1u) + 64; } // the level's record count (region_prefix, published by skr_activate_kernel's first workgroup)
static const uint32_t LEAF2_GRID = 256u * SKR_LEAF2_OCC;                   // every workgroup resident: 256 CUs x 4 workgroups of 4 waves
static const size_t LVL_CTR_WORDS = (size_t) SKR_PULL_STRIDE * (2u * SKR_P1_REGIONS + 2u); // counts, taken, mask, prefix

// Small launches (a rank's share of a frame cut over 4 or 8 GPUs; a full frame with a small tree) take the FLAT schedule: the last
// level too is traced by skr_trace_kernel into records, which skr_shade_leaf_kernel shades — every kernel a plain grid, nothing
// persistent.  The persistent leaf kernel has a few units per wave on such a launch and ends a whole unit after its queues run
// dry (a 1/8 headline frame: 66 % busy); on a large launch its ring and windows save the records' 1.2 GB of HBM traffic instead.
// The measure is the persistent kernel's own: how many records it would cut into units (worst case).  SKR_FLAT=1 / 0 forces one
// or the other (A/B runs; tools/ab_flat.py shows the rule picking the faster schedule on a spread of configurations).
static bool nodes_flat_wanted(const RenderParams &p)
{
	if(p.sw.flat) return p.sw.flat > 0;
	double recs = (double) p.width * p.out_rows; // records the persistent kernel would cut into its units, if every ray hit
	for(int k = 2; k < p.max_depth; k++) recs *= (double) p.num_path_traces;
	return recs < SKR_FLAT_BELOW;
}

static bool plan_for(const RenderParams &p, uint32_t nblk, bool flat, NodePlan &pl)
{
	const uint64_t N = (uint64_t) p.num_path_traces, PP = (N + 1) >> 1;
	pl.flat = flat;
	pl.levels = p.max_depth - 1 + (flat ? 1 : 0);
	if(pl.levels < 1 || pl.levels > SKR_NODE_LEVELS_MAX) return false;
	pl.band_nblk = nblk;
	pl.nodes_max[0] = (uint64_t) nblk * 256u;
	pl.cap[0] = 0;
	for(int L = 1; L < pl.levels; L++)
	{ // a region receives at most 128 hits from each of its trace waves (64 sibling pairs)
		const uint64_t chunks = (pl.nodes_max[L - 1] * PP + 63) / 64;
		const uint64_t cap = (chunks + SKR_P1_REGIONS - 1) / SKR_P1_REGIONS * 128;
		if(cap * SKR_P1_REGIONS >= (1ull << 31)) return false; // record indices carry a flag bit
		pl.cap[L] = (uint32_t) cap;
		pl.nodes_max[L] = cap * SKR_P1_REGIONS;
	}
	size_t off = 0;
	auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
	pl.ctr_bytes = (SKR_PULL_STRIDE + LVL_CTR_WORDS * (size_t) pl.levels) * sizeof(uint32_t); // [0] = level-0 node count, then one block per level
	pl.off_ctr = take(pl.ctr_bytes);
	for(int L = 0; L < pl.levels; L++)
	{
		const size_t n = (size_t) pl.nodes_max[L];
		if(L > 0)
		{
			pl.off_recs[L] = take(n * 16);
			pl.off_res[L] = take(n * 12 + 16);
		}
		if(L < pl.levels - 1 || pl.levels == 1)
		{ // levels whose nodes exist (the last record level is only shaded): geometry rows and shading rows, 32 bytes each per node
			pl.off_nodes[L] = take(n * 32);
			pl.off_shade[L] = take(n * 32);
		}
		if(L < pl.levels - 1)
		{ // the children of level L: a 48-byte header per trace wave (64 sibling pairs)
			pl.off_ixh[L] = take(((n * PP + 63) / 64 + 4) * IXH_ROWS * 16);
		}
	}
	pl.total = off;
	pl.banded = off - (pl.off_nodes[0]); // what grows with the band; the counters are fixed
	return true;
}

static uint64_t nodes_budget(const RenderParams &p, bool flat)
{ // (tests force several bands with a small budget.)  The flat schedule runs in one piece or not at all, and its tables are its price:
  // 4.6 GB for a quarter of the headline frame, of 288
	return p.sw.budget_mb ? (uint64_t) p.sw.budget_mb << 20 : (flat ? 8ull : 4ull) << 30;
}

// the largest band (in 16x16 pixel blocks) whose worst-case tables fit the budget; false: not even one block does
static bool plan_bands(const RenderParams &p, bool flat, NodePlan &pl)
{
	const uint32_t bx = (uint32_t) (p.width + 15) / 16, by = (p.out_rows + 15) / 16;
	const uint64_t budget = nodes_budget(p, flat);
	uint32_t lo = 1, hi = bx * by;
	if(!plan_for(p, lo, flat, pl) || pl.banded > budget) return false;
	if(plan_for(p, hi, flat, pl) && pl.banded <= budget) return true;
	while(hi - lo > 1)
	{ // plan size grows with the block count
		const uint32_t mid = lo + (hi - lo) / 2;
		if(plan_for(p, mid, flat, pl) && pl.banded <= budget) lo = mid;
		else hi = mid;
	}
	if(lo > bx) lo = lo / bx * bx; // whole block rows where possible
	return plan_for(p, lo, flat, pl);
}

static bool skr_nodes_plan(const RenderParams &p, NodePlan &pl)
{
	if(nodes_flat_wanted(p))
	{ // flat only in one piece (SKR_FLAT=1: wherever it fits at all): in bands the persistent kernel is the better schedule
		const uint32_t all = (uint32_t) ((p.width + 15) / 16) * ((p.out_rows + 15) / 16);
		if(plan_bands(p, true, pl) && (p.sw.flat > 0 || pl.band_nblk >= all)) return true;
	}
	return plan_bands(p, false, pl);
}

// The node pipeline covers --gillum trees of any depth >= 2 on sphere scenes (<= 256 children per node, < 65536 spheres).
// SKR_PIPELINE=nodes forces it wherever it applies; other values of SKR_PIPELINE exclude it.
bool skr_nodes_supported(const RenderParams &p)
{
	if(p.shade_triangles || p.legacy_reflect) return false; // (render_generic.hip)
	if(!(p.monte_carlo && p.n_spheres > 0 && p.n_spheres < 65536 && p.max_depth >= 2 && p.num_path_traces > 0 && p.num_path_traces <= 256)) return false;
	NodePlan pl;
	return skr_nodes_plan(p, pl);
}

bool skr_nodes_selected(const RenderParams &p)
{
	const bool forced = p.sw.pipeline == SKR_PIPE_NODES;
	if(p.sw.pipeline != SKR_PIPE_AUTO && !forced) return false;
	if(!skr_nodes_supported(p)) return false;
	if(forced) return true;
	// triangle meshes go to the general level pipeline (render_generic.hip: one lane per ray, one walk of the culling tree per 64
	// rays; test.scn 640x360 --gillum 4: 1.18 ms there against 2.15 ms here, where a lane walks the tree once per sibling);
	// a handful of triangles (spheres1.scn has two) are tested in line by the pair kernels
	return p.n_tris <= 64;
}

// the schedule this launch takes: the flat one (small launches) or the persistent leaf kernel
bool skr_nodes_flat(const RenderParams &p)
{
	NodePlan pl;
	return skr_nodes_plan(p, pl) && pl.flat;
}

size_t skr_nodes_scratch_bytes(const RenderParams &p)
{
	NodePlan pl;
	return skr_nodes_plan(p, pl) ? pl.total : 0;
}

// where the counters of the band last rendered sit in the scratch: [0] = level-0 nodes, then per level 64 region counts
bool skr_nodes_counter_layout(const RenderParams &p, size_t *off_ctr, size_t *level_words, int *levels)
{
	NodePlan pl;
	if(!skr_nodes_plan(p, pl)) return false;
	*off_ctr = pl.off_ctr;
	*level_words = LVL_CTR_WORDS;
	*levels = pl.levels;
	return true;
}

size_t skr_nodes_lds_bytes(const RenderParams &p) { return ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 32 + (size_t) 4 * LEAF2_WAVE_FLOATS * sizeof(float); }

template <bool FIRST>
static hipError_t launch_leaf2(const RenderParams &p, size_t lds, hipStream_t stream)
{
	const void *fn = p.n_tris > 0 ? reinterpret_cast<const void *>(skr_leaf_kernel2<true, FIRST>) : reinterpret_cast<const void *>(skr_leaf_kernel2<false, FIRST>);
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	if(e != hipSuccess) return e;
	if(p.n_tris > 0) hipLaunchKernelGGL((skr_leaf_kernel2<true, FIRST>), dim3(LEAF2_GRID), dim3(256), lds, stream, p);
	else hipLaunchKernelGGL((skr_leaf_kernel2<false, FIRST>), dim3(LEAF2_GRID), dim3(256), lds, stream, p);
	return hipGetLastError();
}

hipError_t skr_launch_nodes(const RenderParams &p_in, hipStream_t stream, const SkrTimingHook *hook)
{
	RenderParams p = p_in;
	NodePlan pl;
	if(!p.node_scratch || !skr_nodes_plan(p, pl)) return hipErrorInvalidValue;
	char *base = reinterpret_cast<char *>(p.node_scratch);
	uint32_t *ctr0 = reinterpret_cast<uint32_t *>(base + pl.off_ctr);
	auto lvl_ctr = [&](int L) { return ctr0 + SKR_PULL_STRIDE + LVL_CTR_WORDS * (size_t) L; };
	auto nodes = [&](int L) { return reinterpret_cast<float4 *>(base + pl.off_nodes[L]); };   // geometry rows
	auto shade = [&](int L) { return reinterpret_cast<float4 *>(base + pl.off_shade[L]); };   // shading rows
	const int nsamp = p.grid_size > 0 ? p.grid_size * p.grid_size : 1;
	const size_t lds_scene = ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 32;
	const size_t lds_leaf = skr_nodes_lds_bytes(p);
	const bool tris = p.n_tris > 0;
	const bool flat = pl.flat;
	const int D = p.max_depth, last = flat ? D - 1 : D - 2; // record levels 1 .. last; the leaf kernel (flat: skr_shade_leaf_kernel) works on level `last`
	const uint32_t blocks_x = (uint32_t) (p.width + 15) / 16, blocks = blocks_x * ((p.out_rows + 15) / 16);
	p.blocks_x = blocks_x;
	p.qctr = ctr0; // [0]: the primary kernel counts its level-0 nodes here
	hipError_t e = hipSuccess;
	for(int s = 0; s < nsamp; s++)
	{
		p.aa_index = (uint32_t) s;
		for(uint32_t blk0 = 0; blk0 < blocks; blk0 += pl.band_nblk)
		{ // every band is a complete pass: its level-0 nodes, their trees, their pixels
			const bool timed = hook && s == nsamp - 1 && blk0 == 0; // (the first band of the last sample: a full-size band)
			p.band_blk0 = blk0;
			p.band_nblk = blocks - blk0 < pl.band_nblk ? blocks - blk0 : pl.band_nblk;
			e = hipMemsetAsync(ctr0, 0, pl.ctr_bytes, stream);
			if(e != hipSuccess) return e;
			p.nd_dst = nodes(0);
			p.ns_dst = shade(0);
			e = skr_launch_primary(p, dim3(p.band_nblk), lds_scene, stream);
			if(e != hipSuccess) return e;
			if(D == 2 && !flat)
			{ // the level-0 nodes' children are the leaves
				p.nd_src = nodes(0);
				p.ns_src = shade(0);
				p.nd_src_level0 = 1;
				p.nd_count = ctr0;
				p.rc_ctr = lvl_ctr(0);
				if(timed) skr_hook_start(hook, stream);
				e = launch_leaf2<true>(p, lds_leaf, stream);
				if(timed) skr_hook_stop(hook, stream);
				if(e != hipSuccess) return e;
				continue;
			}
			for(int L = 1; L <= last; L++)
			{ // the children of level L - 1: hit records of level L, wave headers of level L - 1
				p.nd_src = nodes(L - 1);
				p.ns_src = shade(L - 1);
				p.nd_src_level0 = L == 1;
				p.nd_count = L == 1 ? ctr0 : lc_prefix_host(lvl_ctr(L - 1));
				p.rc = reinterpret_cast<float4 *>(base + pl.off_recs[L]);
				p.rc_cap = pl.cap[L];
				p.rc_ctr = lvl_ctr(L);
				p.ixh = reinterpret_cast<uint4 *>(base + pl.off_ixh[L - 1]);
				const uint64_t wg_t = (pl.nodes_max[L - 1] * (uint64_t) ((p.num_path_traces + 1) >> 1) + 255) / 256;
				const unsigned grid_t = (unsigned) (wg_t < SKR_TRACE_GRID_MAX ? wg_t : SKR_TRACE_GRID_MAX);
				if(flat && L == last && timed) skr_hook_start(hook, stream); // (flat: the last level's trace + shading are the dominant pair)
				if(tris) hipLaunchKernelGGL(skr_trace_kernel<true>, dim3(grid_t), dim3(256), lds_scene, stream, p);
				else hipLaunchKernelGGL(skr_trace_kernel<false>, dim3(grid_t), dim3(256), lds_scene, stream, p);
				if(L < last)
				{ // its records become the nodes of level L
					p.nd_dst = nodes(L);
					p.ns_dst = shade(L);
					const unsigned grid_a = SKR_P1_REGIONS * ((pl.cap[L] + 255u) / 256u);
					if(tris) hipLaunchKernelGGL(skr_activate_kernel<true>, dim3(grid_a), dim3(256), lds_scene, stream, p);
					else hipLaunchKernelGGL(skr_activate_kernel<false>, dim3(grid_a), dim3(256), lds_scene, stream, p);
				}
			}
			p.res_out = reinterpret_cast<float *>(base + pl.off_res[last]);
			if(flat)
			{ // the records of the last level are the leaves' hits: shaded one per lane, results into res[last]
				const uint64_t wg = (pl.nodes_max[last] + 255) / 256;
				const unsigned grid_s = (unsigned) (wg < 16384 ? wg : 16384);
				if(tris) hipLaunchKernelGGL(skr_shade_leaf_kernel<true>, dim3(grid_s), dim3(256), lds_scene, stream, p);
				else hipLaunchKernelGGL(skr_shade_leaf_kernel<false>, dim3(grid_s), dim3(256), lds_scene, stream, p);
				if(timed) skr_hook_stop(hook, stream);
			}
			else
			{ // leaf kernel: the records of the last level (their parents: level last - 1), results into res[last]
				if(timed) skr_hook_start(hook, stream);
				e = launch_leaf2<false>(p, lds_leaf, stream);
				if(timed) skr_hook_stop(hook, stream);
				if(e != hipSuccess) return e;
			}
			for(int L = last - 1; L >= 0; L--)
			{ // sums, deepest level first
				p.nd_src = nodes(L);
				p.ns_src = shade(L);
				p.nd_src_level0 = L == 0;
				p.nd_count = L == 0 ? ctr0 : lc_prefix_host(lvl_ctr(L));
				p.ixh = reinterpret_cast<uint4 *>(base + pl.off_ixh[L]);
				p.res_in = reinterpret_cast<const float *>(base + pl.off_res[L + 1]);
				p.res_out = L == 0 ? nullptr : reinterpret_cast<float *>(base + pl.off_res[L]);
				hipLaunchKernelGGL(skr_finalize_kernel2, dim3((unsigned) ((pl.nodes_max[L] + 255) / 256)), dim3(256), 0, stream, p);
			}
			e = hipGetLastError();
			if(e != hipSuccess) return e;
		}
	}
	if(p.grid_size > 0) return skr_launch_resolve(p, stream);
	return hipSuccess;
}
