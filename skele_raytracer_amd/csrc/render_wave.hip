// "wave-streaming" megakernel: the whole per-pixel loop of the reference
// (src/main.cpp:129-182) and the shade() tree under it (src/raytrace.h:139-227)
// for --depth <= 3, organised for 64-lane waves instead of one recursion per pixel.
//
// The --gillum recursion is an N-ary tree whose size varies from 1 ray (sky) to
// 1 + N + N^2 rays per sample, and only ~1/4 of the rays hit anything that needs
// shading.  Walking it one pixel per lane leaves ~18 % of the lanes active
// (profiles/r01_v1_*).  Here every wave owns an 8x8 pixel tile and streams the
// tree level by level through small LDS queues:
//
//   primary rays (1 lane = 1 pixel)
//     -> parents are compacted into lanes [0,G) a group at a time
//        -> child rays are dealt round-robin, 64 per round: lane = (parent, child)
//           misses deposit their contribution in the parent's slot array at once,
//           sphere hits are pushed (ballot + prefix count) into a ring queue
//        -> whenever a queue holds a full batch it is drained with one lane per
//           hit: level-1 hits become the next set of parents (held in registers of
//           lanes [0,A)), leaf hits are shaded 64 at a time
//     -> slot arrays are summed per parent strictly in child order, because the
//        reference accumulates `total += r1*shade()/pdf` sequentially in float
//        (raytrace.h:117-131) and float addition does not commute with reordering.
//
// Arithmetic is the same spec as everywhere else (device_math.h); only the
// schedule differs, so results are bit-identical to the per-pixel kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wave_common.h"

namespace {

// Two LDS/VGPR budgets are compiled: OCC = waves per SIMD the workgroup's footprint allows.
//   Cfg<3>: 12.5 KB of LDS per wave, <= 168 VGPRs  — best for gillum <= 32 (more waves hide latency)
//   Cfg<2>: 17.5 KB of LDS per wave, <= 256 VGPRs  — bigger slot windows, best for large gillum
constexpr int QF = 8;        // dwords per queue record: d.xyz, b, D, packed ids, slot, r1
// GLOBAL0: the level-1 contribution slots live in a per-wave HBM scratch (L2-resident) instead of LDS — the
// GI kernel of the parent-queue pipeline: groups of up to 32 parents keep the activation batches full.
template <int OCC, bool GLOBAL0 = false>
struct Cfg {
	static constexpr int S0_MAX = (OCC >= 3) ? 128 : 256; // level-1 contribution slots of one parent group (G * N <= S0_MAX)
	static constexpr int S1_MAX = (OCC >= 3) ? 256 : 512; // leaf contribution slots of one window          (AW * N <= S1_MAX)
	// leaf hits waiting to be shaded: <= 63 left over + the 128 a pair round can add (or 64 at a time
	// with a drain in between when the ring is small)
	static constexpr int Q2_CAP = (OCC >= 3) ? 128 : 192;
	// level-1 hits are turned into parents ACT_MAX at a time; the ring holds <= ACT_MAX-1 left over + 64 new
	static constexpr int ACT_MAX = (OCC >= 3) ? 56 : 64;
	static constexpr int Q1_CAP = ACT_MAX + 64;
	static constexpr int PAR0_MAX = GLOBAL0 ? 24 : ((OCC >= 3) ? 8 : 16); // level-1 parents per group (G)
	static constexpr int REGION0_FLOATS = GLOBAL0 ? 0 : S0_MAX * 3 + PAR0_MAX; // + one pad dword per parent (bank spread)
	static constexpr int AW_MAX = (OCC >= 3) ? 16 : 32;          // parents per leaf-slot window
	static constexpr int REGION1_FLOATS = S1_MAX * 3 + AW_MAX;
	static constexpr int SLOT_FLOATS = REGION0_FLOATS + REGION1_FLOATS;
	// the parent-lane table, the group results and the 8x8 u8 tile alias the (then idle) leaf slot region
	// level-1 parents of the current group live in LDS (8 dwords each), not in registers: they are only
	// touched once per round, and holding them in VGPRs through the leaf phases cost occupancy
	static constexpr int WAVE_LDS_FLOATS = SLOT_FLOATS + (Q1_CAP + Q2_CAP) * QF + PAR0_MAX * 8;
	// the depth-1 instance (no --gillum tree) only ever touches the u8 tile staging area: 2 KB per wave instead of
	// 13 KB, so that LDS no longer caps it at 3 waves per SIMD
	static constexpr int DEPTH1_WAVE_FLOATS = REGION0_FLOATS + 64 + PAR0_MAX * 3 + 64;
	static_assert(REGION1_FLOATS >= 64 + 16 * 3 + 48, "aliases must fit");
};
constexpr int GILLUM_MAX = 256; // child index is 8 bits in HitRec.ids; Cfg<2>::S0_MAX

// Slot offsets with this bit address the wave's HBM scratch instead of its LDS area.  The scratch is
// written and read by lanes of ONE wave only: workgroup-scope accesses (plain global loads/stores; the
// CU's L1 is coherent for its own waves) ordered by the wave's own vmcnt(0).  Plain stores stay in the
// XCD's write-back L2, and a wave reuses its few KB for every group, so almost none of it reaches HBM
// (sc1 write-through stores measured 0.85 GB per frame here).
constexpr int SLOT_GLOBAL = 0x40000000;

SKR_DEV void g_store(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SKR_DEV float g_load(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct Queue { // ring of sphere-hit records in LDS, SoA by field; head/count are wave-uniform
	float *base;
	int cap;
	int head, count;
};

struct HitRec {
	f3 d;          // ray direction
	float b, D;    // the spec's float quadratic coefficients of the winning sphere
	uint32_t ids;  // sphere | parent lane << 16 | child index << 24
	int slot;      // dword offset of the contribution slot in the wave's slot area
	float r1;
};

SKR_DEV void q_push(Queue &q, bool pred, const HitRec &h)
{
	const unsigned long long m = __ballot(pred);
	if(pred)
	{
		int e = q.head + q.count + lanes_below(m); // head < cap, count + 64 <= cap + 63
		e -= (e >= q.cap) ? q.cap : 0;
		float *r = q.base + e;
		r[0 * q.cap] = h.d.x;
		r[1 * q.cap] = h.d.y;
		r[2 * q.cap] = h.d.z;
		r[3 * q.cap] = h.b;
		r[4 * q.cap] = h.D;
		r[5 * q.cap] = __uint_as_float(h.ids);
		r[6 * q.cap] = __int_as_float(h.slot);
		r[7 * q.cap] = h.r1;
	}
	q.count = uni(q.count + (int) __popcll(m));
}

SKR_DEV HitRec q_read(const Queue &q, int j)
{
	int e = q.head + j;
	e -= (e >= q.cap) ? q.cap : 0;
	const float *r = q.base + e;
	HitRec h;
	h.d = mk3(r[0 * q.cap], r[1 * q.cap], r[2 * q.cap]);
	h.b = r[3 * q.cap];
	h.D = r[4 * q.cap];
	h.ids = __float_as_uint(r[5 * q.cap]);
	h.slot = __float_as_int(r[6 * q.cap]);
	h.r1 = r[7 * q.cap];
	return h;
}

SKR_DEV void q_drop(Queue &q, int m)
{
	int nh = q.head + m;
	nh -= (nh >= q.cap) ? q.cap : 0;
	q.head = uni(nh);
	q.count = uni(q.count - m);
}

// A tree node whose children are being traced, held in the registers of one lane.
struct Parent {
	f3 co;          // child ray origin: P + 0.00001f (raytrace.h:128)
	f3 N;           // normal; the tangent basis of utils.h:148-165 is re-formed from it per round
	                // (same float operations, so the same basis) instead of occupying 6 more VGPRs
	uint32_t pixel; // RNG key: global pixel index
	uint32_t node;  // RNG key: this node's id (root 0, child c of n = n*N + c + 1)
};

struct Wave {
	SceneView sv;
	const RenderParams *p;
	float *slots;
	int lane;
	int N;            // num_path_traces
	uint32_t magicN;  // ceil(2^24 / N): t / N == (t * magicN) >> 24 for t < 65536, N <= 256
	uint32_t magicPP; // the same for pairs per parent, (N+1)/2
	uint32_t aa;
	float pdf;
	int s0_max, s1_max, sbase1, par0_max, aw_max, act_max; // Cfg<OCC> of this kernel instance
	const float *par0_tbl;                // level-1 parents of the current group: co.xyz, N.xyz, pixel, -
	float *slot0_g;                       // != nullptr: level-1 slots of parent k, child i at slot0_g[k*3N + 3i] (HBM scratch)
	bool q2_two_step;           // the leaf ring cannot take both halves of a pair round at once
	bool slot_plain = false;    // SLOT_GLOBAL slots are only read by a later kernel (level-queue pipeline)
};

SKR_DEV void slot_store(const Wave &w, int slot, f3 v)
{
	if(slot & SLOT_GLOBAL)
	{
		float *g = w.slot0_g + (slot & ~SLOT_GLOBAL);
		if(w.slot_plain)
		{ // read by a later kernel only: one 12-byte store
			struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
			*reinterpret_cast<F3 *>(g) = F3{v.x, v.y, v.z};
		}
		else
		{
			g_store(g, v.x);
			g_store(g + 1, v.y);
			g_store(g + 2, v.z);
		}
	}
	else
	{
		float *s = w.slots + slot;
		s[0] = v.x;
		s[1] = v.y;
		s[2] = v.z;
	}
}

// Where a round's parents live: registers of lanes (shuffle) or the wave's LDS table.
struct ParSrc {
	const Parent *regs; // nullptr => LDS table
	const float *tbl;
};

SKR_DEV void fetch_parent(const ParSrc &s, int kl, f3 &co, f3 &N, uint32_t &pixel, uint32_t &node)
{
	if(s.regs)
	{
		co = shfl3(s.regs->co, kl);
		N = shfl3(s.regs->N, kl);
		pixel = (uint32_t) __shfl((int) s.regs->pixel, kl, 64);
		node = (uint32_t) __shfl((int) s.regs->node, kl, 64);
	}
	else
	{
		const float *r = s.tbl + 8 * kl;
		co = mk3(r[0], r[1], r[2]);
		N = mk3(r[3], r[4], r[5]);
		pixel = __float_as_uint(r[6]);
		node = 0; // level-1 parents are the primary hits: tree root
	}
}

SKR_DEV f3 fetch_parent_origin(const ParSrc &s, int kl)
{
	if(s.regs) return shfl3(s.regs->co, kl);
	const float *r = s.tbl + 8 * kl;
	return mk3(r[0], r[1], r[2]);
}

// closest accepted sphere without forming the winner's exact t2 (done later, in the
// compacted shading pass): returns the sphere and its float coefficients b, D.
SKR_DEV int closest_sphere_deferred(const SceneView &sv, f3 o, f3 d, const RayFilt &f, float &b_out, float &D_out)
{
	int best = -1;
	float best_lo = __builtin_inff(), best_hi = __builtin_inff(), others_lo = __builtin_inff();
	b_out = 0.0f;
	D_out = 0.0f;
	float4 g_next = sv.geom[0];
	for(int i = 0; i < sv.ns; i++)
	{
		const float4 g = g_next;
		g_next = sv.geom[i + 1]; // software prefetch; geom[] carries one pad entry
		float lo, hi, b, D;
		if(sphere_bracket(o, d, f, g, lo, hi, b, D))
		{
			if(hi < best_hi)
			{
				others_lo = __builtin_fminf(others_lo, best_lo);
				best_lo = lo;
				best_hi = hi;
				best = i;
				b_out = b;
				D_out = D;
			}
			else others_lo = __builtin_fminf(others_lo, lo);
		}
	}
	if(best >= 0 && !(others_lo > best_hi))
	{ // brackets overlap: the exact loop names the winner; recompute its coefficients
		float tmin;
		const RayConst r = make_ray(o, d);
		best = closest_sphere_exact(sv, r, tmin);
		const f3 e = o - ld3(sv.geom[best]);
		b_out = 2 * dot3(d, e);
		const float c = dot3(e, e) - sv.geom[best].w;
		D_out = b_out * b_out - f.four_a * c;
	}
	return best;
}

// One round of child rays: lane = task task0 + lane = (parent k, child i) of the parents
// held in lanes [kbase, kbase+np).  The contribution slot of (k, i) is slots[sbase + k*(3N+1) + 3i].
SKR_DEV void child_round(const Wave &w, const ParSrc &par, int kbase, int np, int task0, int sbase, Queue &q, Counters &cn)
{
	const int t = task0 + w.lane;
	const bool valid = t < np * w.N;
	const int k = valid ? (int) (((uint32_t) t * w.magicN) >> 24) : 0; // parent within the window [kbase, kbase+np)
	const int i = t - k * w.N;
	const int kl = kbase + k;                                           // lane that holds the parent
	f3 co, N, nt, nb;
	uint32_t pixel, node;
	fetch_parent(par, kl, co, N, pixel, node);
	tangent_basis(N, nt, nb);
	HitRec h;
	h.d = mk3(0, 0, 0);
	h.b = h.D = h.r1 = 0.0f;
	h.ids = 0;
	// level-1 rounds (sbase 0) of the GI kernel deposit into the HBM scratch
	h.slot = (sbase == 0 && w.slot0_g) ? (SLOT_GLOBAL | (k * 3 * w.N + 3 * i)) : sbase + k * (3 * w.N + 1) + 3 * i;
	bool hit = false;
	if(valid)
	{
		uint32_t rnd[4];
		philox4x32(pixel, w.aa, node, (uint32_t) i >> 1, w.p->seed_lo, w.p->seed_hi, rnd);
		const float r1 = (i & 1) ? u31_to_unit(rnd[2]) : u31_to_unit(rnd[0]);
		const float r2 = (i & 1) ? u31_to_unit(rnd[3]) : u31_to_unit(rnd[1]);
		const f3 d = gi_direction(r1, r2, N, nt, nb);
		cn.rays++;
		const RayFilt f = make_filt(d);
		float b, D;
		const int sph = closest_sphere_deferred(w.sv, co, d, f, b, D);
		bool tri = false;
		if(w.sv.nt > 0)
		{ // raytrace.h:171-186 needs the sphere's exact t to compare against
			const float tmin = (sph >= 0) ? near_root_exact(f.two_a, b, D) : __builtin_inff();
			const RayConst r = RayConst{co, d, f.two_a, f.four_a};
			tri = any_triangle_closer(w.sv, r, tmin);
		}
		if(tri || sph < 0)
		{ // raytrace.h:189-192 / :221-224, then :130: total += (r1 * colour) / pdf
			const f3 colour = tri ? mk3(0, 0, 0) : w.p->background;
			slot_store(w, h.slot, (colour * r1) / w.pdf);
		}
		else
		{
			hit = true;
			h.d = d;
			h.b = b;
			h.D = D;
			h.ids = (uint32_t) sph | ((uint32_t) kl << 16) | ((uint32_t) i << 24);
			h.r1 = r1;
		}
	}
	q_push(q, hit, h);
}

struct PairOut {
	HitRec h0, h1;
	bool hit0, hit1;
};

// Finish one traced child: a miss / triangle deposits its term now, a sphere hit fills a record.
SKR_DEV bool finish_child(const Wave &w, f3 co, f3 d, float two_a, float four_a, const BestState &s, int kl, int i, float r1, int slot, HitRec &h)
{
	bool tri = false;
	if(w.sv.nt > 0)
	{ // raytrace.h:171-186 needs the sphere's exact t to compare against
		const float tmin = (s.best >= 0) ? near_root_exact(two_a, s.b, s.D) : __builtin_inff();
		const RayConst r = RayConst{co, d, two_a, four_a};
		tri = any_triangle_closer(w.sv, r, tmin);
	}
	h.d = d;
	h.b = s.b;
	h.D = s.D;
	h.ids = (uint32_t) (s.best & 0xffff) | ((uint32_t) kl << 16) | ((uint32_t) i << 24);
	h.slot = slot;
	h.r1 = r1;
	if(tri || s.best < 0)
	{ // raytrace.h:189-192 / :221-224, then :130: total += (r1 * colour) / pdf
		const f3 colour = tri ? mk3(0, 0, 0) : w.p->background;
		const f3 contrib = (colour * r1) / w.pdf;
		float *sl = w.slots + slot;
		sl[0] = contrib.x;
		sl[1] = contrib.y;
		sl[2] = contrib.z;
		return false;
	}
	return true;
}

// One round of 64 sibling PAIRS (up to 128 child rays) of the parents in lanes [kbase, kbase+np).
SKR_DEV PairOut child_round_pairs(const Wave &w, const ParSrc &par, int kbase, int np, int pair0, int sbase, Counters &cn)
{
	const int PP = (w.N + 1) >> 1; // pairs per parent
	const int t = pair0 + w.lane;
	const bool valid = t < np * PP;
	const int k = valid ? (int) (((uint32_t) t * w.magicPP) >> 24) : 0;
	const int j = t - k * PP;
	const int i0 = 2 * j, i1 = 2 * j + 1;
	const bool second = valid && i1 < w.N;
	const int kl = kbase + k;
	f3 co, N, nt, nb;
	uint32_t pixel, node;
	fetch_parent(par, kl, co, N, pixel, node);
	tangent_basis(N, nt, nb);
	PairOut po;
	po.hit0 = po.hit1 = false;
	po.h0.d = po.h1.d = mk3(0, 0, 0);
	po.h0.b = po.h0.D = po.h0.r1 = po.h1.b = po.h1.D = po.h1.r1 = 0.0f;
	po.h0.ids = po.h1.ids = 0;
	po.h0.slot = po.h1.slot = 0;
	if(valid)
	{
		uint32_t rnd[4];
		philox4x32(pixel, w.aa, node, (uint32_t) j, w.p->seed_lo, w.p->seed_hi, rnd);
		const float r1a = u31_to_unit(rnd[0]), r2a = u31_to_unit(rnd[1]);
		const float r1b = u31_to_unit(rnd[2]), r2b = u31_to_unit(rnd[3]);
		const DirPair dp = gi_direction_pair(r1a, r2a, r1b, r2b, N, nt, nb);
		const f3 d0 = dp.d0, d1 = dp.d1;
		cn.rays += second ? 2u : 1u;
		const RayPair rp = make_pair(d0, d1);
		BestState s0, s1;
		closest_pair(w.sv, co, d0, d1, second, rp, s0, s1);
		const int slot0 = sbase + k * (3 * w.N + 1) + 3 * i0;
		po.hit0 = finish_child(w, co, d0, rp.two_a.x, rp.four_a.x, s0, kl, i0, r1a, slot0, po.h0);
		if(second) po.hit1 = finish_child(w, co, d1, rp.two_a.y, rp.four_a.y, s1, kl, i1, r1b, slot0 + 3, po.h1);
	}
	return po;
}

// Shade m <= 64 queued hits whose node has depth 1 (its own children are shade(depth 0) == 0):
// raytrace.h:194-213 with indirect = (0,0,0)/N, then the parent's accumulation term (:130).
SKR_DEV void shade_leaf_batch(const Wave &w, Queue &q, const ParSrc &par, int m, Counters &cn)
{
	wave_lds_fence();
	const bool act = w.lane < m;
	DIAG_WAVE(9, 1);
	DIAG_WAVE(10, m);
	HitRec h = q_read(q, act ? w.lane : 0);
	const int k = (int) ((h.ids >> 16) & 0xffu);
	const f3 co = fetch_parent_origin(par, act ? k : 0);
	if(act)
	{
		const int sph = (int) (h.ids & 0xffffu);
		const float two_a = 2 * dot3(h.d, h.d);
		const float t = near_root_exact(two_a, h.b, h.D);
		const f3 P = co + h.d * t;
		const f3 N = normalize3(P - ld3(w.sv.geom[sph]));
		cn.hits++;
		const f3 direct = direct_light(w.sv, *w.p, sph, P, N, cn);
		const f3 total = mk3(0, 0, 0) / (float) w.N;
		const f3 colour = (direct / (float) 3.14159265358979323846 + total * 2.0f) * ld3(w.sv.kd[sph]);
		slot_store(w, h.slot, (colour * h.r1) / w.pdf);
	}
	q_drop(q, m);
	wave_lds_fence();
}

// Sum the N child contributions of the parent held by this lane, strictly in child order.
SKR_DEV f3 sum_slots(const Wave &w, int sbase, int k)
{
	const float *s = w.slots + sbase + k * (3 * w.N + 1);
	f3 total = mk3(0, 0, 0);
	for(int i = 0; i < w.N; i++) total = total + mk3(s[3 * i], s[3 * i + 1], s[3 * i + 2]);
	return total;
}

// DEPTH == 3: m <= 64 queued level-1 hits become the active parents (lanes [0,m)); their
// N leaf rays each are traced in rounds, leaf hits are shaded in batches of 64, and each
// parent's result is deposited in ITS parent's slot (raytrace.h:130).
// The m level-1 hits held by lanes [0, m) — record h, origin co0 of the ray that found it, pixel — become parents.
SKR_DEV void expand_level1_hits(const Wave &w, int m, const HitRec &h, f3 co0, uint32_t pixel, Queue &q2, Counters &cn STAMP_ARG)
{
	const bool act = w.lane < m;
	DIAG_WAVE(11, 1);
	DIAG_WAVE(12, m);
	Parent par1;
	par1.co = par1.N = mk3(0, 0, 1);
	par1.pixel = pixel;
	par1.node = ((h.ids >> 24) & 0xffu) + 1u; // child i of the root (node 0): 0*N + i + 1
	f3 direct1 = mk3(0, 0, 0);
	int sph1 = 0;
	if(act)
	{
		sph1 = (int) (h.ids & 0xffffu);
		const float two_a = 2 * dot3(h.d, h.d);
		const float t = near_root_exact(two_a, h.b, h.D);
		const f3 P = co0 + h.d * t;
		par1.N = normalize3(P - ld3(w.sv.geom[sph1]));
		cn.hits++;
		direct1 = direct_light(w.sv, *w.p, sph1, P, par1.N, cn);
		par1.co = add_scalar(P, 0.00001f);
	}
	const int sbase1 = w.sbase1;
	const ParSrc src1{&par1, nullptr};
	STAMP(4);
	// the m parents were shaded together (full-width); their leaf rays go through the slot area a
	// window of AW parents at a time
	const int AW = uni(w.s1_max / w.N < w.aw_max ? w.s1_max / w.N : w.aw_max);
	for(int w0 = 0; w0 < m; w0 += AW)
	{
		const int mw = uni(m - w0 < AW ? m - w0 : AW);
		const int npairs = mw * ((w.N + 1) >> 1);
		for(int pair0 = 0; pair0 < npairs; pair0 += 64)
		{
			const PairOut po = child_round_pairs(w, src1, w0, mw, pair0, sbase1, cn);
			STAMP(2);
			const bool last = pair0 + 64 >= npairs;
			if(w.q2_two_step)
			{ // small ring: push the even children's hits, drain, then the odd ones (never more than 63 + 64 queued)
#pragma nounroll
				for(int sub = 0; sub < 2; sub++)
				{
					q_push(q2, sub ? po.hit1 : po.hit0, sub ? po.h1 : po.h0);
					while(q2.count >= 64 || (last && sub == 1 && q2.count > 0))
					{
						shade_leaf_batch(w, q2, src1, q2.count < 64 ? q2.count : 64, cn);
						STAMP(3);
					}
				}
			}
			else
			{
				q_push(q2, po.hit0, po.h0);
				q_push(q2, po.hit1, po.h1);
				// one call site (code size): full batches as they form, the remainder after the window's last round
				while(q2.count >= 64 || (last && q2.count > 0))
				{
					shade_leaf_batch(w, q2, src1, q2.count < 64 ? q2.count : 64, cn);
					STAMP(3);
				}
			}
		}
		wave_lds_fence();
		if(act && w.lane >= w0 && w.lane < w0 + mw)
		{
			f3 total = sum_slots(w, sbase1, w.lane - w0);
			total = total / (float) w.N;
			const f3 colour = (direct1 / (float) 3.14159265358979323846 + total * 2.0f) * ld3(w.sv.kd[sph1]);
			slot_store(w, h.slot, (colour * h.r1) / w.pdf);
		}
		wave_lds_fence();
		STAMP(5);
	}
}

SKR_DEV void expand_level1_batch(const Wave &w, Queue &q1, Queue &q2, int m, Counters &cn STAMP_ARG)
{
	STAMP(1);
	wave_lds_fence();
	const bool act = w.lane < m;
	const HitRec h = q_read(q1, act ? w.lane : 0);
	const int k0 = (int) ((h.ids >> 16) & 0xffu);
	const float *p0 = w.par0_tbl + 8 * (act ? k0 : 0);
	const f3 co0 = mk3(p0[0], p0[1], p0[2]);
	const uint32_t pixel = __float_as_uint(p0[6]);
	q_drop(q1, m);
	expand_level1_hits(w, m, h, co0, pixel, q2, cn STAMP_PASS);
}

// All child rays (and, at depth 3, grandchild rays) of the gp parents in the wave's LDS parent table:
// afterwards slot region 0 holds every parent's N accumulation terms (raytrace.h:130).
template <int DEPTH>
SKR_DEV void run_group(const Wave &w, int gp, Queue &q1, Queue &q2, Counters &cn STAMP_ARG)
{
	const ParSrc src0{nullptr, w.par0_tbl};
	const int ntasks = gp * w.N;
	for(int task0 = 0; task0 < ntasks; task0 += 64)
	{
		const bool last = task0 + 64 >= ntasks;
		if constexpr(DEPTH == 2)
		{
			child_round(w, src0, 0, gp, task0, 0, q2, cn);
			while(q2.count >= 64 || (last && q2.count > 0)) shade_leaf_batch(w, q2, src0, q2.count < 64 ? q2.count : 64, cn);
		}
		else
		{
			child_round(w, src0, 0, gp, task0, 0, q1, cn);
			STAMP(1);
			while(q1.count >= w.act_max || (last && q1.count > 0)) expand_level1_batch(w, q1, q2, q1.count < w.act_max ? q1.count : w.act_max, cn STAMP_PASS);
		}
	}
	wave_lds_fence();
}

// One sample of every pixel of the wave's tile: raytrace.h:139-227 at depth DEPTH.
template <int DEPTH>
SKR_DEV f3 shade_tile_sample(const Wave &w, bool valid, f3 o, f3 d, uint32_t pixel, Queue &q1, Queue &q2, int *lane_tbl, float *gres, Counters &cn STAMP_ARG)
{
	const RenderParams &p = *w.p;
	// ---- primary rays: one lane per pixel
	f3 result = mk3(0, 0, 0);
	bool hit = false;
	int sph0 = 0;
	Parent mine;
	mine.co = mine.N = mk3(0, 0, 1);
	mine.pixel = pixel;
	mine.node = 0;
	if(valid)
	{
		cn.rays++;
		const RayConst r = make_ray(o, d);
		float tmin;
		const int sph = closest_sphere(w.sv, r, tmin);
		if(w.sv.nt > 0 && any_triangle_closer(w.sv, r, tmin)) result = mk3(0, 0, 0);
		else if(sph < 0) result = p.background;
		else
		{
			hit = true;
			sph0 = sph;
			cn.hits++;
			const f3 P = o + d * tmin;
			mine.N = normalize3(P - ld3(w.sv.geom[sph]));
			result = direct_light(w.sv, p, sph, P, mine.N, cn); // direct colour; combined with the indirect term below
			if(DEPTH > 1 && p.monte_carlo) mine.co = add_scalar(P, 0.00001f);
		}
	}
	STAMP(0);
	if(!p.monte_carlo) return result;

	f3 indirect = mk3(0, 0, 0); // sum of the children's terms; stays 0 when they are all shade(depth 0)
	if constexpr(DEPTH > 1)
	{
		const unsigned long long M0 = __ballot(hit);
		const int n0 = (int) __popcll(M0);
		const int rank = lanes_below(M0);
		const int G = uni(w.s0_max / (w.N > 0 ? w.N : 1) < w.par0_max ? w.s0_max / (w.N > 0 ? w.N : 1) : w.par0_max);
		for(int g0 = 0; g0 < n0 && w.N > 0; g0 += G)
		{
			const int gp = uni(n0 - g0 < G ? n0 - g0 : G);
			const bool in_group = hit && rank >= g0 && rank < g0 + gp;
			// compact this group's parents into lanes [0, gp)
			if(in_group) lane_tbl[rank - g0] = w.lane;
			wave_lds_fence();
			const int src = (w.lane < gp) ? lane_tbl[w.lane] : 0;
			{ // the group's parents go to the LDS table: co.xyz, N.xyz, pixel
				const f3 pco = shfl3(mine.co, src), pN = shfl3(mine.N, src);
				const uint32_t ppix = (uint32_t) __shfl((int) mine.pixel, src, 64);
				if(w.lane < gp)
				{
					float *r = const_cast<float *>(w.par0_tbl) + 8 * w.lane;
					r[0] = pco.x; r[1] = pco.y; r[2] = pco.z;
					r[3] = pN.x;  r[4] = pN.y;  r[5] = pN.z;
					r[6] = __uint_as_float(ppix);
				}
				wave_lds_fence();
			}
			run_group<DEPTH>(w, gp, q1, q2, cn STAMP_PASS);
			if(w.lane < gp)
			{
				const f3 total = sum_slots(w, 0, w.lane);
				gres[3 * w.lane] = total.x;
				gres[3 * w.lane + 1] = total.y;
				gres[3 * w.lane + 2] = total.z;
			}
			wave_lds_fence();
			if(in_group) indirect = mk3(gres[3 * (rank - g0)], gres[3 * (rank - g0) + 1], gres[3 * (rank - g0) + 2]);
			wave_lds_fence();
		}
	}
	if(hit)
	{ // raytrace.h:133 + :213
		const f3 total = indirect / (float) w.N;
		result = (result / (float) 3.14159265358979323846 + total * 2.0f) * ld3(w.sv.kd[sph0]);
	}
	return result;
}

} // namespace

// One workgroup = 4 independent waves; wave w of block (bx, by) owns the 8x8 pixel tile
// (2*bx + (w&1), 2*by + (w>>1)).  Dynamic LDS: scene SoA (shared, staged once) | 4 wave areas.
template <int DEPTH, int OCC, bool TRIS = true> // TRIS = false: no triangles in the scene, the walk is compiled out
__global__ __launch_bounds__(256, OCC) void skr_wave_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const int ns = p.n_spheres, nl = p.n_lights;
	float4 *s_geom = lds4, *s_amb = lds4 + ns + 1, *s_kd = s_amb + ns, *s_ks = s_kd + ns, *s_lights = s_ks + ns;
	const int tid = threadIdx.x;
	for(int i = tid; i < ns; i += 256)
	{
		s_geom[i] = p.sph_geom[i];
		s_amb[i] = p.sph_amb[i];
		s_kd[i] = p.sph_kd[i];
		s_ks[i] = p.sph_ks[i];
	}
	for(int i = tid; i < 2 * nl; i += 256) s_lights[i] = p.lights[i];
	if(tid == 0) s_geom[ns] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	__syncthreads(); // the only workgroup barrier: from here on the four waves never meet again

	const int wave = tid >> 6, lane = tid & 63;
	using C = Cfg<OCC>;
	float *wbase = reinterpret_cast<float *>(lds4 + 4 * ns + 1 + 2 * nl) + wave * (DEPTH == 1 ? C::DEPTH1_WAVE_FLOATS : C::WAVE_LDS_FLOATS);
	Wave w;
	w.sv = SceneView{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, TRIS ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work};
	w.p = &p;
	w.slots = wbase;
	w.lane = lane;
	w.N = p.num_path_traces;
	w.magicN = (uint32_t) (((1u << 24) + (uint32_t) (w.N > 0 ? w.N : 1) - 1u) / (uint32_t) (w.N > 0 ? w.N : 1));
	{
		const uint32_t pp = (uint32_t) ((w.N > 0 ? w.N : 1) + 1) >> 1;
		w.magicPP = ((1u << 24) + pp - 1u) / pp;
	}
	w.aa = 0;
	w.pdf = (float) (1 / 3.14159265358979323846);
	w.s0_max = C::S0_MAX;
	w.s1_max = C::S1_MAX;
	w.sbase1 = C::REGION0_FLOATS;
	w.q2_two_step = C::Q2_CAP < 63 + 128;
	w.par0_max = C::PAR0_MAX;
	w.aw_max = C::AW_MAX;
	w.act_max = C::ACT_MAX;
	w.par0_tbl = wbase + C::SLOT_FLOATS + (C::Q1_CAP + C::Q2_CAP) * QF;
	w.slot0_g = nullptr;
	Queue q1{wbase + C::SLOT_FLOATS, C::Q1_CAP, 0, 0}, q2{wbase + C::SLOT_FLOATS + C::Q1_CAP * QF, C::Q2_CAP, 0, 0};
	int *lane_tbl = reinterpret_cast<int *>(wbase + C::REGION0_FLOATS); // aliases of the leaf slot region, see Cfg
	float *gres = wbase + C::REGION0_FLOATS + 64;
	unsigned char *s_tile = reinterpret_cast<unsigned char *>(gres + C::PAR0_MAX * 3);

	// A wave's tile is 8x8, 8x4 or 4x4 pixels (p.tile_w_log2/p.tile_h_log2): small launches (one
	// GPU's share of a frame sharded 8 ways) use smaller tiles so that there are several tiles per
	// wave slot and one deep tile cannot dominate the frame time.  Lanes beyond the tile idle in the
	// primary pass only; the --gillum rounds are packed by (parent, child) regardless.
	const int tw = 1 << p.tile_w_log2, th = 1 << p.tile_h_log2;
	const int lx = lane & (tw - 1), ly = lane >> p.tile_w_log2;
	const int x0 = (blockIdx.x * 2 + (wave & 1)) * tw;
	const uint32_t orow0 = (blockIdx.y * 2 + (wave >> 1)) * (uint32_t) th;
	const int x = x0 + lx;
	const uint32_t orow = orow0 + ly;
	const uint32_t k = orow / p.tile_rows;
	const uint32_t y = (p.first_tile + k * p.tile_stride) * p.tile_rows + (orow - k * p.tile_rows);
	const bool valid = ly < th && x < p.width && orow < p.out_rows && y < (uint32_t) p.height;
	const uint32_t pixel = y * (uint32_t) p.width + (uint32_t) x;

	Counters cn{0, 0, 0};
	STAMP_DECL;
	f3 px = mk3(0, 0, 0);
	// main.cpp:140-182: g*g jittered samples (one draw r for both axes, all-float) or one centre
	// sample (u, v formed in double).  One call site for both (code size).
	const int nsamp = p.grid_size > 0 ? p.grid_size * p.grid_size : 1;
	for(int s = 0; s < nsamp; s++)
	{
		w.aa = (uint32_t) s;
		float u, v;
		if(p.grid_size > 0)
		{
			uint32_t rnd[4];
			philox4x32(pixel, (uint32_t) s, 0u, 0xFFFFFFFFu, p.seed_lo, p.seed_hi, rnd);
			const float r = u31_to_unit(rnd[0]);
			u = ((2 * (((float) x + r) * p.inv_width) - 1) * p.angle) * p.aspect;
			v = (1 - 2 * (((float) (int) y + r) * p.inv_height)) * p.angle;
		}
		else
		{
			u = (float) (((2 * (((double) x + 0.5) * (double) p.inv_width) - 1) * (double) p.angle) * (double) p.aspect);
			v = (float) ((1 - 2 * (((double) (int) y + 0.5) * (double) p.inv_height)) * (double) p.angle);
		}
		const f3 dir = (p.cam_dir + p.cam_right * u) + p.cam_up * v;
		const f3 smp = shade_tile_sample<DEPTH>(w, valid, p.cam_pos, dir, pixel, q1, q2, lane_tbl, gres, cn STAMP_PASS);
		px = (p.grid_size > 0) ? px + smp : smp; // image[y][x] += shade(...) from zero, or = shade(...)
	}
	if(p.grid_size > 0) px = px / (float) nsamp;

	if(valid && p.rgbf)
	{
		float *o = p.rgbf + ((size_t) orow * p.width + x) * 3;
		o[0] = px.x;
		o[1] = px.y;
		o[2] = px.z;
	}
	if(p.rgb)
	{ // pack to u8 in LDS, then store the tile's 8 rows x 24 bytes as 48 dwords
		unsigned char *t = s_tile + ((ly & 7) * tw + lx) * 3;
		if(ly < th)
		{
			t[0] = (unsigned char) quantise(px.x);
			t[1] = (unsigned char) quantise(px.y);
			t[2] = (unsigned char) quantise(px.z);
		}
		wave_lds_fence();
		const bool full = (x0 + tw <= p.width) && ((p.width & 3) == 0);
		if(full)
		{ // th rows of tw*3 bytes (24 or 12) = dw dwords each
			const int dw = (tw * 3) >> 2;
			if(lane < th * dw)
			{
				const int row = lane / dw, j = lane - row * dw;
				const uint32_t orow2 = orow0 + row;
				const uint32_t k2 = orow2 / p.tile_rows;
				const uint32_t y2 = (p.first_tile + k2 * p.tile_stride) * p.tile_rows + (orow2 - k2 * p.tile_rows);
				if(orow2 < p.out_rows && y2 < (uint32_t) p.height)
				{
					uint32_t *dst = reinterpret_cast<uint32_t *>(p.rgb + ((size_t) orow2 * p.width + x0) * 3);
					dst[j] = reinterpret_cast<const uint32_t *>(s_tile + row * tw * 3)[j];
				}
			}
		}
		else if(valid)
		{
			unsigned char *dst = p.rgb + ((size_t) orow * p.width + x) * 3;
			dst[0] = t[0];
			dst[1] = t[1];
			dst[2] = t[2];
		}
	}
#if defined(SKR_STAMPS) && SKR_STAMPS
	STAMP(6);
	if(p.counters && lane == 0)
		for(int k = 0; k < 8; k++) atomicAdd(&p.counters[4u * SKR_COUNTER_SHARDS + k], st_acc[k]);
#endif
	if(p.counters)
	{
		const uint32_t a = wave_sum(cn.rays), b = wave_sum(cn.hits), c = wave_sum(cn.shadow_rays), d4 = wave_sum(cn.shadow_tests);
		if(lane == 0)
		{ // sharded: thousands of waves adding to ONE word serialise at ~88 atomics/us (1.1 ms per 1080p frame)
			const uint32_t shard = ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (uint32_t) wave) & (SKR_COUNTER_SHARDS - 1u);
			unsigned long long *c4 = p.counters + 4u * shard;
			atomicAdd(&c4[0], (unsigned long long) a);
			atomicAdd(&c4[1], (unsigned long long) b);
			atomicAdd(&c4[2], (unsigned long long) c);
			atomicAdd(&c4[3], (unsigned long long) d4);
		}
	}
}

// gillum <= 32: the 3-waves-per-SIMD budget wins (3.6 vs 4.1 ms at N = 16); above, the larger slot
// windows of the 2-wave budget do (21.9 vs 35 ms at N = 64, 960x540).  Measured: DESIGN.md §6.
// =====================================================================================
// Parent-queue pipeline (the product path for --gillum at depth 2..3).
//
// Inside one megakernel launch a wave's latency is set by its deepest pixel tile (one 8x8 tile of
// ground pixels takes ~1.7 ms of the 3.1 ms frame), which caps strong scaling and leaves waves with
// sparse tiles half empty.  Here the tree is cut once, under the primary hit:
//   skr_primary_kernel   primary rays + direct light for every pixel; each sphere hit appends a
//                        64-byte parent record (ballot + one atomic per workgroup) to a device queue
//   skr_gi_kernel        persistent waves pull groups of G parents from the queue and run the same
//                        level-synchronous streaming as skr_wave_kernel (run_group), then finish
//                        `(direct/pi + 2*indirect) * kd` (raytrace.h:213) and write the pixel
//   skr_resolve_kernel   (AA only) `image /= g*g` (main.cpp:165) and the u8 quantiser
// Work items are 8..16 parents (a few hundred rays) instead of 32..64 pixels with their whole trees.
// Values are the same spec: the image is bit-identical to the single-kernel path.
// =====================================================================================

namespace {

} // namespace

// One workgroup = a 16x16 pixel block, one lane per pixel.  TRIS = false: no triangles in the scene, the walk is compiled out.
#ifndef SKR_PRIMARY_WAVES
#define SKR_PRIMARY_WAVES 0 // waves per SIMD the primary kernel is held to (0: whatever its registers allow) — A/B builds
#endif
#if SKR_PRIMARY_WAVES
#define SKR_PRIMARY_ATTR __attribute__((amdgpu_waves_per_eu(SKR_PRIMARY_WAVES, SKR_PRIMARY_WAVES)))
#else
#define SKR_PRIMARY_ATTR
#endif
template <bool TRIS>
__global__ __launch_bounds__(256) SKR_PRIMARY_ATTR void skr_primary_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const int ns = p.n_spheres, nl = p.n_lights;
	float4 *s_geom = lds4, *s_amb = lds4 + ns + 1, *s_kd = s_amb + ns, *s_ks = s_kd + ns, *s_lights = s_ks + ns;
	uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds4 + 4 * ns + 1 + 2 * nl); // 4 wave counts + block base
	const int tid = threadIdx.x;
	for(int i = tid; i < ns; i += 256)
	{
		s_geom[i] = p.sph_geom[i];
		s_amb[i] = p.sph_amb[i];
		s_kd[i] = p.sph_kd[i];
		s_ks[i] = p.sph_ks[i];
	}
	for(int i = tid; i < 2 * nl; i += 256) s_lights[i] = p.lights[i];
	if(tid == 0) s_geom[ns] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	__syncthreads();
	const SceneView sv{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, TRIS ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work};

	const int wave = tid >> 6, lane = tid & 63;
	const int lx = ((wave & 1) << 3) | (lane & 7), ly = ((wave >> 1) << 3) | (lane >> 3);
	// node pipeline: a band is a run of 16x16 blocks in row-major order (1-D grid); otherwise a band of rows (2-D grid)
	const uint32_t blk = p.node_layout ? p.band_blk0 + blockIdx.x : 0u;
	const uint32_t bx = p.node_layout ? blk % p.blocks_x : blockIdx.x, by = p.node_layout ? blk / p.blocks_x : blockIdx.y;
	const int x = (int) bx * 16 + lx;
	const uint32_t brow = by * 16 + ly, orow = p.band_row0 + brow;
	const uint32_t k = orow / p.tile_rows;
	const uint32_t y = (p.first_tile + k * p.tile_stride) * p.tile_rows + (orow - k * p.tile_rows);
	const bool valid = x < p.width && (p.node_layout || brow < p.band_rows) && orow < p.out_rows && y < (uint32_t) p.height;
	const uint32_t pixel = y * (uint32_t) p.width + (uint32_t) x;
	const uint32_t out_pix = orow * (uint32_t) p.width + (uint32_t) x;

	Counters cn{0, 0, 0};
	f3 colour = mk3(0, 0, 0), co = mk3(0, 0, 0), N = mk3(0, 0, 1), kd = mk3(0, 0, 0);
	bool hit = false;
	int sph_hit = 0;
	if(valid)
	{
		f3 dir;
		primary_ray(p, x, y, pixel, p.aa_index, dir);
		cn.rays++;
		const RayConst r = make_ray(p.cam_pos, dir);
		float tmin;
		const int sph = closest_sphere(sv, r, tmin);
		if(sv.nt > 0 && any_triangle_closer(sv, r, tmin)) colour = mk3(0, 0, 0);
		else if(sph < 0) colour = p.background;
		else
		{
			hit = true;
			sph_hit = sph;
			cn.hits++;
			const f3 P = p.cam_pos + dir * tmin;
			N = normalize3(P - ld3(sv.geom[sph]));
			colour = direct_light(sv, p, sph, P, N, cn);
			co = add_scalar(P, 0.00001f);
			kd = ld3(sv.kd[sph]);
		}
	}
	// append the hits: wave-level ranks, one atomic per workgroup
	const unsigned long long M = __ballot(hit);
	if(lane == 0) s_cnt[wave] = (uint32_t) __popcll(M);
	__syncthreads();
	if(tid == 0)
	{
		const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
		s_cnt[4] = total ? atomicAdd(&p.qctr[0], total) : 0u;
	}
	__syncthreads();
	if(hit)
	{
		uint32_t idx = s_cnt[4] + (uint32_t) lanes_below(M);
		for(int wv = 0; wv < wave; wv++) idx += s_cnt[wv];
		if(p.node_layout)
		{ // a level-0 node of the node pipeline (render_params.h)
			// (render_params.h: a node is a 32-byte geometry row — what tracing its children needs — and a 32-byte shading row — what summing them needs)
			float4 *grow = p.nd_dst + (size_t) idx * 2, *srow = p.ns_dst + (size_t) idx * 2;
			grow[0] = make_float4(co.x, co.y, co.z, N.x);
			grow[1] = make_float4(N.y, N.z, __uint_as_float(pixel), __uint_as_float(0u));
			srow[0] = make_float4(colour.x, colour.y, colour.z, __uint_as_float((uint32_t) sph_hit));
			srow[1] = make_float4(0.0f, __uint_as_float(out_pix), __uint_as_float(pixel), __uint_as_float(0u));
		}
		else
		{
			float4 *rec = p.parents + (size_t) idx * 4;
			rec[0] = make_float4(co.x, co.y, co.z, N.x);
			rec[1] = make_float4(N.y, N.z, colour.x, colour.y);
			rec[2] = make_float4(colour.z, kd.x, kd.y, kd.z);
			rec[3] = make_float4(__uint_as_float(pixel), __uint_as_float(out_pix), 0.0f, 0.0f);
		}
	}
	else if(valid) emit_sample(p, out_pix, colour); // this sample of this pixel is final
	if(p.counters)
	{
		const uint32_t a = wave_sum(cn.rays), b = wave_sum(cn.hits), c = wave_sum(cn.shadow_rays), d4 = wave_sum(cn.shadow_tests);
		if(lane == 0)
		{
			const uint32_t shard = ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (uint32_t) wave) & (SKR_COUNTER_SHARDS - 1u);
			unsigned long long *c4 = p.counters + 4u * shard;
			atomicAdd(&c4[0], (unsigned long long) a);
			atomicAdd(&c4[1], (unsigned long long) b);
			atomicAdd(&c4[2], (unsigned long long) c);
			atomicAdd(&c4[3], (unsigned long long) d4);
		}
	}
}

// Persistent waves: pull groups of parents, stream their trees, finish their pixels.
template <int DEPTH, int OCC, bool TRIS> // TRIS = false: no triangles in the scene, the walk is compiled out
__global__ __launch_bounds__(256, OCC) void skr_gi_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const int ns = p.n_spheres, nl = p.n_lights;
	float4 *s_geom = lds4, *s_amb = lds4 + ns + 1, *s_kd = s_amb + ns, *s_ks = s_kd + ns, *s_lights = s_ks + ns;
	const int tid = threadIdx.x;
	for(int i = tid; i < ns; i += 256)
	{
		s_geom[i] = p.sph_geom[i];
		s_amb[i] = p.sph_amb[i];
		s_kd[i] = p.sph_kd[i];
		s_ks[i] = p.sph_ks[i];
	}
	for(int i = tid; i < 2 * nl; i += 256) s_lights[i] = p.lights[i];
	if(tid == 0) s_geom[ns] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	__syncthreads(); // the only workgroup barrier

	const int wave = tid >> 6, lane = tid & 63;
	using C = Cfg<OCC, true>;
	float *wbase = reinterpret_cast<float *>(lds4 + 4 * ns + 1 + 2 * nl) + wave * C::WAVE_LDS_FLOATS;
	Wave w;
	w.sv = SceneView{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, TRIS ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work};
	w.p = &p;
	w.slots = wbase;
	w.lane = lane;
	w.N = p.num_path_traces;
	w.magicN = (uint32_t) (((1u << 24) + (uint32_t) (w.N > 0 ? w.N : 1) - 1u) / (uint32_t) (w.N > 0 ? w.N : 1));
	{
		const uint32_t pp = (uint32_t) ((w.N > 0 ? w.N : 1) + 1) >> 1;
		w.magicPP = ((1u << 24) + pp - 1u) / pp;
	}
	w.aa = p.aa_index;
	w.pdf = (float) (1 / 3.14159265358979323846);
	w.s0_max = C::S0_MAX;
	w.s1_max = C::S1_MAX;
	w.sbase1 = C::REGION0_FLOATS;
	w.q2_two_step = C::Q2_CAP < 63 + 128;
	w.par0_max = C::PAR0_MAX;
	w.aw_max = C::AW_MAX;
	w.act_max = C::ACT_MAX;
	w.par0_tbl = wbase + C::SLOT_FLOATS + (C::Q1_CAP + C::Q2_CAP) * QF;
	// this wave's private level-1 slot scratch: PAR0_MAX parents x N children x float3
	w.slot0_g = p.slot0_scratch + (size_t) (blockIdx.x * 4u + (uint32_t) wave) * (size_t) (C::PAR0_MAX * 3) * (size_t) (w.N > 0 ? w.N : 1);
	Queue q1{wbase + C::SLOT_FLOATS, C::Q1_CAP, 0, 0}, q2{wbase + C::SLOT_FLOATS + C::Q1_CAP * QF, C::Q2_CAP, 0, 0};

	const uint32_t n_parents = p.qctr[0];
	// parents per group: as many as keep ~8 groups per wave slot, between 8 and the table's 32 (big groups
	// fill the 56-wide activation batches; small queues need small groups to balance)
	const uint32_t slots = gridDim.x * 4u;
	uint32_t G = (n_parents / (slots * p.gi_groups_per_slot)) & ~(p.gi_group_round - 1u);
	G = G < 8u ? 8u : (G > (uint32_t) w.par0_max ? (uint32_t) w.par0_max : G);
	const uint32_t n_groups = (n_parents + G - 1) / G;
	// One group per atomic, and SKR_PULL_QUEUES counters instead of one: a single word sustains ~88 atomics/us,
	// which 3072 waves pulling 8-parent groups exceed (a 1/8 frame needs 22 000 pulls: 0.25 ms of a 0.47 ms
	// kernel was spent queueing for that word).  Queue k owns the group indices congruent to k mod K (every queue
	// sweeps the frame front to back, like the single counter did); a wave starts at
	// queue (its index mod K) and moves on to the next one when its queue runs dry, until all K did.
	const uint32_t K = SKR_PULL_QUEUES;
	uint32_t qk = (blockIdx.x * 4u + (uint32_t) wave) % K, dry = 0;
	Counters cn{0, 0, 0};
	STAMP_DECL;
	for(;;)
	{
		uint32_t g;
		for(;;)
		{ // wave-uniform
			uint32_t g0 = 0;
			if(lane == 0) g0 = atomicAdd(&p.qctr[SKR_PULL_STRIDE * (1u + qk)], 1u);
			g = (uint32_t) __builtin_amdgcn_readfirstlane((int) g0) * K + qk;
			if(g < n_groups) break;
			qk = qk + 1u == K ? 0u : qk + 1u;
			if(++dry == K) break;
		}
		if(dry == K) break;
		// A group is G consecutive queue entries (screen neighbours: coherent rays, 6 % faster on a full
		// frame) — unless there are fewer than ~12 groups per wave slot: then one run of deep ground pixels
		// (8 x 273 rays) decides the frame time, and group g takes parents g, g + n_groups, g + 2 n_groups, ...
		// instead, which mixes deep and shallow trees (1/8 frame: 0.525 -> 0.486 ms).
		const bool strided = n_groups < 12u * slots;
		const uint32_t first = strided ? g : g * G, step = strided ? n_groups : 1u;
		const uint32_t avail = strided ? (n_parents - 1u - g) / n_groups + 1u : n_parents - first;
		const int gp = (int) (avail < G ? avail : G);
		f3 direct0 = mk3(0, 0, 0), kd0 = mk3(0, 0, 0);
		uint32_t out_pix = 0;
		if(lane < gp)
		{ // record -> LDS parent table (co, N, pixel); direct colour, kd and the output index stay in this lane
			const float4 *rec = p.parents + (size_t) (first + (uint32_t) lane * step) * 4;
			// streamed once: non-temporal, so the records do not push the waves' slot scratch out of L2
			typedef float v4f __attribute__((ext_vector_type(4)));
			const v4f *rv = reinterpret_cast<const v4f *>(rec);
			const v4f a0 = __builtin_nontemporal_load(&rv[0]), a1 = __builtin_nontemporal_load(&rv[1]),
					  a2 = __builtin_nontemporal_load(&rv[2]), a3 = __builtin_nontemporal_load(&rv[3]);
			const float4 r0 = make_float4(a0.x, a0.y, a0.z, a0.w), r1 = make_float4(a1.x, a1.y, a1.z, a1.w),
						 r2 = make_float4(a2.x, a2.y, a2.z, a2.w), r3 = make_float4(a3.x, a3.y, a3.z, a3.w);
			float *tb = const_cast<float *>(w.par0_tbl) + 8 * lane;
			tb[0] = r0.x; tb[1] = r0.y; tb[2] = r0.z;
			tb[3] = r0.w; tb[4] = r1.x; tb[5] = r1.y;
			tb[6] = r3.x;
			direct0 = mk3(r1.z, r1.w, r2.x);
			kd0 = mk3(r2.y, r2.z, r2.w);
			out_pix = __float_as_uint(r3.y);
		}
		wave_lds_fence();
		STAMP(0);
		run_group<DEPTH>(w, gp, q1, q2, cn STAMP_PASS);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's slot stores are complete
		__builtin_amdgcn_wave_barrier();
		STAMP(6); // (diagnostic builds: 6 = store drain, 7 = slot sums + emit)
		if(lane < gp)
		{ // raytrace.h:133 + :213: the N terms strictly in child order
			const float *s = w.slot0_g + lane * 3 * w.N;
			f3 total = mk3(0, 0, 0);
			// 16 children per trip to L2: all their loads are issued before the first add (the adds stay in child
			// order).  One trip per child made this the critical path of small queues: 16 x ~1 us per group.
			for(int i0 = 0; i0 < w.N; i0 += 16)
			{
				float v[16][3];
#pragma unroll
				for(int k = 0; k < 16; k++)
				{
					const int i = (i0 + k < w.N) ? i0 + k : w.N - 1;
					v[k][0] = g_load(s + 3 * i);
					v[k][1] = g_load(s + 3 * i + 1);
					v[k][2] = g_load(s + 3 * i + 2);
				}
#pragma unroll
				for(int k = 0; k < 16; k++)
				{
					const f3 sum = total + mk3(v[k][0], v[k][1], v[k][2]);
					if(i0 + k < w.N) total = sum;
				}
			}
			total = total / (float) w.N;
			emit_sample(p, out_pix, (direct0 / (float) 3.14159265358979323846 + total * 2.0f) * kd0);
		}
		wave_lds_fence();
		STAMP(7);
	}
#if defined(SKR_STAMPS) && SKR_STAMPS
	if(p.counters && lane == 0)
		for(int k = 0; k < 8; k++) atomicAdd(&p.counters[4u * SKR_COUNTER_SHARDS + k], st_acc[k]);
#endif
	if(p.counters)
	{
		const uint32_t a = wave_sum(cn.rays), b = wave_sum(cn.hits), c = wave_sum(cn.shadow_rays), d4 = wave_sum(cn.shadow_tests);
		if(lane == 0)
		{
			const uint32_t shard = (blockIdx.x * 4u + (uint32_t) wave) & (SKR_COUNTER_SHARDS - 1u);
			unsigned long long *c4 = p.counters + 4u * shard;
			atomicAdd(&c4[0], (unsigned long long) a);
			atomicAdd(&c4[1], (unsigned long long) b);
			atomicAdd(&c4[2], (unsigned long long) c);
			atomicAdd(&c4[3], (unsigned long long) d4);
		}
	}
}

// =====================================================================================
// Level-queue pipeline (opt-in: SKR_PIPELINE=levels; depth 3).  The parent-queue pipeline balances parents, whose
// trees differ by a factor of 16 (17 .. 273 rays); here the tree is cut a second time, under the level-1 hits, whose
// subtrees are all alike (N leaf rays + their shading):
//   skr_primary_kernel   as above: 64-byte parent records
//   skr_level1_kernel    one lane per (parent, child): traces the level-1 ray; a miss deposits its term in
//                        slot1[parent * N + child], a sphere hit appends a 64-byte record to one of SKR_P1_REGIONS
//                        regions (ballot + one atomic per wave; region = wave index mod 64, so a region can never
//                        overflow: it only receives hits of its own waves)
//   skr_leaf_kernel      one wave per 64 records of a region: shades the 64 level-1 hits full-width, traces their
//                        64 * N leaf rays in sibling pairs, shades the leaf hits in batches of 64, deposits each
//                        record's result in its slot1 entry (expand_level1_hits: the code the other pipelines run)
//   skr_finalize_kernel  one lane per parent: the N slots strictly in child order, (direct/pi + 2 indirect) * kd
// Same values, same order of every float sum: the image is bit-identical to the other paths.
// =====================================================================================
#if defined(SKR_DIAG) && SKR_DIAG
extern "C" void skr_diag_read(unsigned long long *out, int reset)
{ // 32 counters, summed over their 64 shards
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
#if defined(SKR_STAMPS) && SKR_STAMPS
static __device__ unsigned long long skr_leaf_times[5 * 4096];
extern "C" void skr_leaf_times_read(unsigned long long *out)
{
	(void) hipDeviceSynchronize();
	(void) hipMemcpyFromSymbol(out, HIP_SYMBOL(skr_leaf_times), sizeof(unsigned long long) * 5 * 4096);
}
#endif

namespace {
constexpr int LEAF_S1 = 384, LEAF_AW = 24, LEAF_Q2 = 128;                       // leaf slots of a window, parents per window, leaf-hit ring
constexpr int LEAF_WAVE_FLOATS = LEAF_S1 * 3 + LEAF_AW + LEAF_Q2 * QF;
static_assert(SKR_P1_REGIONS == 64u, "the leaf kernel looks at one region per lane when its own runs dry");
SKR_DEV uint32_t *p1_counter(const RenderParams &p, uint32_t region) { return p.qctr + SKR_PULL_STRIDE * (1u + SKR_PULL_QUEUES + region); }
SKR_DEV uint32_t *p1_taken(const RenderParams &p, uint32_t region) { return p.qctr + SKR_PULL_STRIDE * (1u + SKR_PULL_QUEUES + SKR_P1_REGIONS + region); } // units handed out
SKR_DEV unsigned long long *p1_dead_mask(const RenderParams &p) { return reinterpret_cast<unsigned long long *>(p.qctr + SKR_PULL_STRIDE * (1u + SKR_PULL_QUEUES + 2u * SKR_P1_REGIONS)); } // regions seen exhausted
} // namespace

// TRIS = false: the scene has no triangles (the launcher's default case) and the walk is compiled out.
template <bool TRIS>
__global__ __launch_bounds__(256) void skr_level1_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const int ns = p.n_spheres, nl = p.n_lights;
	float4 *s_geom = lds4, *s_amb = lds4 + ns + 1, *s_kd = s_amb + ns, *s_ks = s_kd + ns, *s_lights = s_ks + ns;
	const int tid = threadIdx.x;
	const uint32_t N = (uint32_t) p.num_path_traces, PP = (N + 1u) >> 1; // children, sibling pairs per parent
	const uint32_t n_pairs = p.qctr[0] * PP;
	if((uint32_t) blockIdx.x * 256u >= n_pairs) return; // (uniform per workgroup)
	for(int i = tid; i < ns; i += 256)
	{
		s_geom[i] = p.sph_geom[i];
		s_amb[i] = p.sph_amb[i];
		s_kd[i] = p.sph_kd[i];
		s_ks[i] = p.sph_ks[i];
	}
	for(int i = tid; i < 2 * nl; i += 256) s_lights[i] = p.lights[i];
	if(tid == 0) s_geom[ns] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	__syncthreads();
	const SceneView sv{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, TRIS ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work};
	const int lane = tid & 63;
	const uint32_t wave1 = (uint32_t) blockIdx.x * 4u + (uint32_t) (tid >> 6);
	// one lane per sibling pair (children 2j, 2j+1 of a parent): one Philox call and one (e, c) per sphere for both
	const uint32_t tp = wave1 * 64u + (uint32_t) lane;
	const bool valid = tp < n_pairs;
	const uint32_t parent = valid ? tp / PP : 0u, j = valid ? tp - parent * PP : 0u;
	const uint32_t i0 = 2u * j, i1 = 2u * j + 1u;
	const bool second = valid && i1 < N;
	Counters cn{0, 0, 0};
	bool hit0 = false, hit1 = false;
	float4 rec0[3], rec1[3];
	rec0[0] = rec0[1] = rec0[2] = rec1[0] = rec1[1] = rec1[2] = make_float4(0, 0, 0, 0);
	if(valid)
	{
		const float4 *rec = p.parents + (size_t) parent * 4;
		const float4 a0 = rec[0], a1 = rec[1], a3 = rec[3];
		const f3 co = mk3(a0.x, a0.y, a0.z), Nn = mk3(a0.w, a1.x, a1.y);
		const uint32_t pixel = __float_as_uint(a3.x);
		f3 nt, nb;
		tangent_basis(Nn, nt, nb);
		uint32_t rnd[4];
		philox4x32(pixel, p.aa_index, 0u, j, p.seed_lo, p.seed_hi, rnd); // node 0: the children of the primary hit
		const float r1a = u31_to_unit(rnd[0]), r2a = u31_to_unit(rnd[1]), r1b = u31_to_unit(rnd[2]), r2b = u31_to_unit(rnd[3]);
		const DirPair dp = gi_direction_pair(r1a, r2a, r1b, r2b, Nn, nt, nb);
		const f3 d0 = dp.d0, d1 = dp.d1;
		cn.rays += second ? 2u : 1u;
		const RayPair rp = make_pair(d0, d1);
		BestState s0, s1;
		closest_pair(sv, co, d0, d1, second, rp, s0, s1);
		const float pdf = (float) (1 / 3.14159265358979323846);
#pragma nounroll
		for(int c = 0; c < 2; c++)
		{
			if(c == 1 && !second) break;
			const f3 d = c ? d1 : d0;
			const BestState &s = c ? s1 : s0;
			const float two_a = c ? rp.two_a.y : rp.two_a.x, four_a = c ? rp.four_a.y : rp.four_a.x, r1 = c ? r1b : r1a;
			const uint32_t i = c ? i1 : i0, t = parent * N + i;
			bool tri = false;
			if(sv.nt > 0)
			{ // raytrace.h:171-186 needs the sphere's exact t to compare against
				const float tmin = (s.best >= 0) ? near_root_exact(two_a, s.b, s.D) : __builtin_inff();
				tri = any_triangle_closer(sv, RayConst{co, d, two_a, four_a}, tmin);
			}
			if(tri || s.best < 0)
			{ // raytrace.h:189-192 / :221-224, then :130: total += (r1 * colour) / pdf
				const f3 colour = tri ? mk3(0, 0, 0) : p.background;
				const f3 cc = (colour * r1) / pdf;
				struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
				*reinterpret_cast<F3 *>(p.slot1 + (size_t) t * 3) = F3{cc.x, cc.y, cc.z};
			}
			else
			{
				float4 *o = c ? rec1 : rec0;
				o[0] = make_float4(co.x, co.y, co.z, d.x);
				o[1] = make_float4(d.y, d.z, s.b, s.D);
				o[2] = make_float4(__uint_as_float((uint32_t) s.best), __uint_as_float(i), r1, __uint_as_float(pixel));
				if(c) hit1 = true; else hit0 = true;
			}
		}
	}
	// append the wave's hits to its region: rank by ballot, one atomic per wave
	const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
	const uint32_t region = wave1 & (SKR_P1_REGIONS - 1u);
	const uint32_t n0h = (uint32_t) __popcll(m0), n1h = (uint32_t) __popcll(m1);
	uint32_t base = 0;
	if(n0h + n1h != 0u)
	{
		if(lane == 0) base = atomicAdd(p1_counter(p, region), n0h + n1h);
		base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
	}
	float4 *reg = p.p1 + (size_t) region * p.p1_region_cap * 4;
	if(hit0)
	{
		float4 *dst = reg + (size_t) (base + (uint32_t) lanes_below(m0)) * 4;
		dst[0] = rec0[0]; dst[1] = rec0[1]; dst[2] = rec0[2];
		dst[3] = make_float4(__uint_as_float(parent * N + i0), 0.0f, 0.0f, 0.0f);
	}
	if(hit1)
	{
		float4 *dst = reg + (size_t) (base + n0h + (uint32_t) lanes_below(m1)) * 4;
		dst[0] = rec1[0]; dst[1] = rec1[1]; dst[2] = rec1[2];
		dst[3] = make_float4(__uint_as_float(parent * N + i1), 0.0f, 0.0f, 0.0f);
	}
	if(p.counters)
	{
		const uint32_t a = wave_sum(cn.rays);
		if(lane == 0 && a) atomicAdd(&p.counters[4u * (wave1 & (SKR_COUNTER_SHARDS - 1u))], (unsigned long long) a);
	}
}

template <bool TRIS>
__global__ __launch_bounds__(256, 4) void skr_leaf_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const int ns = p.n_spheres, nl = p.n_lights;
	float4 *s_geom = lds4, *s_amb = lds4 + ns + 1, *s_kd = s_amb + ns, *s_ks = s_kd + ns, *s_lights = s_ks + ns;
	const int tid = threadIdx.x;
	// unit = 64 records of a region.  Persistent waves: wave g starts at region g mod R and pulls that region's next
	// unit with one atomic; when the region is exhausted it moves on to the next one, until all R are.  The units are
	// all alike (64 level-1 hits, 64 N leaf rays), and a small launch has only a few per wave: pulling keeps the last
	// wave from being a whole unit late.
	for(int i = tid; i < ns; i += 256)
	{
		s_geom[i] = p.sph_geom[i];
		s_amb[i] = p.sph_amb[i];
		s_kd[i] = p.sph_kd[i];
		s_ks[i] = p.sph_ks[i];
	}
	for(int i = tid; i < 2 * nl; i += 256) s_lights[i] = p.lights[i];
	if(tid == 0) s_geom[ns] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	__syncthreads(); // the only workgroup barrier
	const int wave = tid >> 6, lane = tid & 63;
	const uint32_t g = (uint32_t) blockIdx.x * 4u + (uint32_t) wave;
	uint32_t region = g & (SKR_P1_REGIONS - 1u);
	unsigned long long dead = 0; // regions this wave has seen exhausted
#if defined(SKR_STAMPS) && SKR_STAMPS
	const unsigned long long wt_start = wall_clock64(); // 100 MHz, one clock for the whole device (the cycle counter is per XCD)
	unsigned long long wt_first = 0, wt_last = 0;
	uint32_t wt_units = 0;
#endif
	float *wbase = reinterpret_cast<float *>(lds4 + 4 * ns + 1 + 2 * nl) + wave * LEAF_WAVE_FLOATS;
	Wave w;
	w.sv = SceneView{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, TRIS ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work};
	w.p = &p;
	w.slots = wbase;
	w.lane = lane;
	w.N = p.num_path_traces;
	w.magicN = (uint32_t) (((1u << 24) + (uint32_t) w.N - 1u) / (uint32_t) w.N);
	{
		const uint32_t pp = (uint32_t) (w.N + 1) >> 1;
		w.magicPP = ((1u << 24) + pp - 1u) / pp;
	}
	w.aa = p.aa_index;
	w.pdf = (float) (1 / 3.14159265358979323846);
	w.s0_max = 0;
	w.s1_max = LEAF_S1;
	w.sbase1 = 0;
	w.q2_two_step = LEAF_Q2 < 63 + 128;
	w.par0_max = 0;
	w.aw_max = LEAF_AW;
	w.act_max = 64;
	w.par0_tbl = nullptr;
	w.slot0_g = p.slot1; // a record's result goes to slot1[3 t] (SLOT_GLOBAL offsets)
	w.slot_plain = true;
	Queue q2{wbase + LEAF_S1 * 3 + LEAF_AW, LEAF_Q2, 0, 0};
	Counters cn{0, 0, 0};
	STAMP_DECL;
	for(;;)
	{
		uint32_t first = 0, cnt = 0;
		bool got = false;
		for(;;)
		{ // pull the next unit (wave-uniform): this region's, or another region's once this one is exhausted
			cnt = *p1_counter(p, region);
			// (16-record pulls for a region's last quarter — units differ 2-3x in cost, and a wave's last whole unit sets
			// the kernel's tail — were measured slower, 1.58 -> 1.67 ms: a small unit costs far more than its share)
			uint32_t k = 0;
			if(lane == 0) k = atomicAdd(p1_taken(p, region), 1u);
			first = (uint32_t) __builtin_amdgcn_readfirstlane((int) k) * 64u;
			if(first < cnt)
			{
				got = true;
				break;
			}
			// This region is exhausted.  Walking the other 63 one atomic at a time cost every wave up to 64 round trips at
			// the end (per-wave timeline: 90 .. 270 us between the last unit and the exit; plain loads of the other
			// regions' counters are too stale to help).  Instead the exhausted regions are published in one 64-bit mask:
			// the atomic OR that adds this region returns everybody else's findings, and the wave goes to the first
			// region after its own that nobody has seen dry — or leaves when there is none.
			unsigned long long seen = 0;
			if(lane == 0) seen = atomicOr(p1_dead_mask(p), 1ull << region);
			const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) seen), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (seen >> 32));
			dead |= ((unsigned long long) hi << 32 | lo) | (1ull << region);
			const unsigned long long live = ~dead;
			if(live == 0ull) break;
			const unsigned long long after = region == 63u ? 0ull : (live >> (region + 1u)) << (region + 1u); // regions above this one
			region = (uint32_t) __builtin_ctzll(after ? after : live);
		}
		if(!got) break;
		const float4 *reg = p.p1 + (size_t) region * p.p1_region_cap * 4;
		const int m = (int) (cnt - first < 64u ? cnt - first : 64u);
		HitRec h;
		h.d = mk3(0, 0, 1);
		h.b = h.D = h.r1 = 0.0f;
		h.ids = 0;
		h.slot = SLOT_GLOBAL;
		f3 co0 = mk3(0, 0, 0);
		uint32_t pixel = 0;
		if(lane < m)
		{ // (fetching the next unit's records one unit ahead was measured: no gain, the other waves of the SIMD cover the wait)
			typedef float v4f __attribute__((ext_vector_type(4)));
			const v4f *rv = reinterpret_cast<const v4f *>(reg + (size_t) (first + (uint32_t) lane) * 4);
			const v4f n0 = __builtin_nontemporal_load(&rv[0]), n1 = __builtin_nontemporal_load(&rv[1]), n2 = __builtin_nontemporal_load(&rv[2]),
					  n3 = __builtin_nontemporal_load(&rv[3]);
			co0 = mk3(n0.x, n0.y, n0.z);
			h.d = mk3(n0.w, n1.x, n1.y);
			h.b = n1.z;
			h.D = n1.w;
			h.ids = (__float_as_uint(n2.x) & 0xffffu) | (__float_as_uint(n2.y) << 24); // sphere | child index << 24
			h.r1 = n2.z;
			pixel = __float_as_uint(n2.w);
			h.slot = SLOT_GLOBAL | (int) (__float_as_uint(n3.x) * 3u);
		}
		STAMP(0);
#if defined(SKR_STAMPS) && SKR_STAMPS
		if(wt_units == 0) wt_first = wall_clock64();
#endif
		expand_level1_hits(w, m, h, co0, pixel, q2, cn STAMP_PASS);
#if defined(SKR_STAMPS) && SKR_STAMPS
		wt_units++;
		wt_last = wall_clock64();
#endif
	}
#if defined(SKR_STAMPS) && SKR_STAMPS
	if(lane == 0 && g < 4096u)
	{ // per-wave timeline (tools/leaf_timeline.py): kernel entry, first unit in hand, last unit done, exit, units
		skr_leaf_times[5 * g] = wt_start;
		skr_leaf_times[5 * g + 1] = wt_first;
		skr_leaf_times[5 * g + 2] = wt_last;
		skr_leaf_times[5 * g + 3] = wall_clock64();
		skr_leaf_times[5 * g + 4] = wt_units;
	}
#endif
#if defined(SKR_STAMPS) && SKR_STAMPS
	if(p.counters && lane == 0)
		for(int k = 0; k < 8; k++) atomicAdd(&p.counters[4u * SKR_COUNTER_SHARDS + k], st_acc[k]);
#endif
	if(p.counters)
	{
		const uint32_t a = wave_sum(cn.rays), b = wave_sum(cn.hits), c = wave_sum(cn.shadow_rays), d4 = wave_sum(cn.shadow_tests);
		if(lane == 0)
		{
			unsigned long long *c4 = p.counters + 4u * (g & (SKR_COUNTER_SHARDS - 1u));
			atomicAdd(&c4[0], (unsigned long long) a);
			atomicAdd(&c4[1], (unsigned long long) b);
			atomicAdd(&c4[2], (unsigned long long) c);
			atomicAdd(&c4[3], (unsigned long long) d4);
		}
	}
}

__global__ __launch_bounds__(256) void skr_finalize_kernel(const RenderParams p)
{ // a wave = 64 parents; their slots are read as one contiguous run (coalesced), 16 children at a time, through LDS
	__shared__ float s_t[4][64 * 49];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t n0 = p.qctr[0];
	const uint32_t parent0 = ((uint32_t) blockIdx.x * 4u + (uint32_t) wave) * 64u;
	if(parent0 >= n0) return;
	const uint32_t parent = parent0 + (uint32_t) lane;
	const bool valid = parent < n0;
	const int N = p.num_path_traces;
	float *mine = s_t[wave];
	f3 total = mk3(0, 0, 0);
	for(int c0 = 0; c0 < N; c0 += 16)
	{
		const int nc = (N - c0 < 16) ? N - c0 : 16, run = 3 * nc; // floats of one parent in this chunk
		const size_t chunk0 = ((size_t) parent0 * N + c0) * 3; // first float of the wave's run in this chunk
		if((run & 3) == 0 && ((3 * N) & 3) == 0)
		{ // 16-byte loads: run / 4 float4 per parent
			const int run4 = run >> 2;
			for(int idx = lane; idx < 64 * run4; idx += 64)
			{
				const int pl = idx / run4, q = idx - pl * run4;
				if(parent0 + (uint32_t) pl < n0)
				{
					const float4 v = *reinterpret_cast<const float4 *>(p.slot1 + chunk0 + (size_t) pl * 3 * N + 4 * q);
					float *d = mine + pl * 49 + 4 * q;
					d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
				}
			}
		}
		else
			for(int idx = lane; idx < 64 * run; idx += 64)
			{
				const int pl = idx / run, off = idx - pl * run;
				if(parent0 + (uint32_t) pl < n0) mine[pl * 49 + off] = p.slot1[chunk0 + (size_t) pl * 3 * N + off];
			}
		wave_lds_fence();
		if(valid)
			for(int i = 0; i < nc; i++) total = total + mk3(mine[lane * 49 + 3 * i], mine[lane * 49 + 3 * i + 1], mine[lane * 49 + 3 * i + 2]); // raytrace.h:130, child order
		wave_lds_fence();
	}
	if(!valid) return;
	const float4 *rec = p.parents + (size_t) parent * 4;
	const float4 a1 = rec[1], a2 = rec[2], a3 = rec[3];
	const f3 direct0 = mk3(a1.z, a1.w, a2.x), kd0 = mk3(a2.y, a2.z, a2.w);
	total = total / (float) N;
	emit_sample(p, __float_as_uint(a3.y), (direct0 / (float) 3.14159265358979323846 + total * 2.0f) * kd0); // raytrace.h:213
}

// AA only: image[y][x] /= g*g (main.cpp:165), then the quantiser (main.cpp:205).
__global__ __launch_bounds__(256) void skr_resolve_kernel(const RenderParams p)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	const size_t n = (size_t) p.width * p.out_rows;
	if(i >= n) return;
	const uint32_t orow = (uint32_t) (i / (size_t) p.width);
	const uint32_t k = orow / p.tile_rows;
	const uint32_t y = (p.first_tile + k * p.tile_stride) * p.tile_rows + (orow - k * p.tile_rows);
	if(y >= (uint32_t) p.height) return;
	const float ns2 = (float) (p.grid_size * p.grid_size);
	const f3 px = mk3(p.acc[3 * i], p.acc[3 * i + 1], p.acc[3 * i + 2]) / ns2;
	if(p.rgbf)
	{
		p.rgbf[3 * i] = px.x;
		p.rgbf[3 * i + 1] = px.y;
		p.rgbf[3 * i + 2] = px.z;
	}
	if(p.rgb)
	{
		p.rgb[3 * i] = (unsigned char) quantise(px.x);
		p.rgb[3 * i + 1] = (unsigned char) quantise(px.y);
		p.rgb[3 * i + 2] = (unsigned char) quantise(px.z);
	}
}

static size_t wave_block_lds(const RenderParams &p, int occ, bool global0 = false)
{
	const size_t per_wave = (global0 ? (occ == 3 ? Cfg<3, true>::WAVE_LDS_FLOATS : Cfg<2, true>::WAVE_LDS_FLOATS)
									  : (occ == 3 ? Cfg<3>::WAVE_LDS_FLOATS : Cfg<2>::WAVE_LDS_FLOATS)) * sizeof(float);
	return ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 4 * per_wave;
}

// Which LDS/VGPR budget to launch.  gillum <= 32: three waves per SIMD win (3.1 vs 3.6 ms at N = 16);
// above, the larger slot windows of the two-wave budget do (19 vs 35 ms at N = 64, 960x540) — provided
// three workgroups of the small budget really fit the CU's 160 KiB (a scene with many spheres can push the third workgroup out, and the small budget at two
// waves per SIMD is the worst of both).  SKR_OCC=2|3 forces one (A/B runs).
static int wave_occ_for(const RenderParams &p)
{
	if(p.sw.occ) return p.sw.occ;
	// (and for gillum < 8 the small budget's 8-parent groups cannot fill a 64-lane round)
	if(p.monte_carlo && p.n_spheres > 0 && p.max_depth > 1 && (p.num_path_traces > 32 || p.num_path_traces < 8)) return 2;
	// measured on MI355X: LDS is granted in 1280-byte granules (160 KiB / 128): 3 x 53,264 B and
	// 2 x 81,680 B are resident together, 3 x 53,904 B and 2 x 81,936 B are not
	const size_t granule = 1280, cu_lds = 160 * 1024;
	const size_t blk3 = (wave_block_lds(p, 3) + granule - 1) / granule * granule;
	return 3 * blk3 <= cu_lds ? 3 : 2;
}

size_t skr_wave_lds_bytes(const RenderParams &p) { return wave_block_lds(p, wave_occ_for(p)); }

// The streaming kernel covers --depth 1..3, gillum <= 256, <= 65535 spheres.
bool skr_wave_supported(const RenderParams &p)
{
	return p.max_depth >= 1 && p.max_depth <= 3 && p.num_path_traces <= GILLUM_MAX && p.n_spheres < 65536;
}

template <int D, int OCC, bool TRIS = true>
static hipError_t launch_wave_depth(const RenderParams &p, dim3 grid, size_t lds, hipStream_t stream)
{
	// > 64 KiB of dynamic LDS per workgroup has to be opted into
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(skr_wave_kernel<D, OCC, TRIS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	if(e != hipSuccess) return e;
	hipLaunchKernelGGL((skr_wave_kernel<D, OCC, TRIS>), grid, dim3(256), lds, stream, p);
	return hipGetLastError();
}

hipError_t skr_launch_wave(const RenderParams &p_in, hipStream_t stream)
{
	RenderParams p = p_in;
	// tile shape.  Measured on the headline frame (tools/time_shard.py, slowest rank's kernel, ms):
	//   share of the frame   8x8     8x4     4x4
	//   1/1                  3.12    3.03    3.30
	//   1/2                  2.11    1.55    1.64
	//   1/4                  1.78    1.04    0.84
	//   1/8                  1.76    0.98    0.55
	// one deep 8x8 tile alone takes ~1.7 ms, so 8x8 stops scaling at two GPUs.  Default 8x4; 4x4 once
	// a launch has fewer than ~8 8x4-tiles per resident wave slot.  Without --gillum trees there is
	// nothing to balance and 8x8 keeps every primary-pass lane busy.  SKR_TILE=64|32|16 forces a shape.
	int tile_px = 32;
	{
		const uint64_t slots = 256ull * 4 * 3;
		const uint64_t pixels = (uint64_t) p.width * p.out_rows;
		if(pixels / 32 < 8 * slots) tile_px = 16;
		if(!p.monte_carlo || p.n_spheres == 0 || p.max_depth < 2) tile_px = 64;
		if(p.sw.tile) tile_px = p.sw.tile;
	}
	p.tile_w_log2 = tile_px == 16 ? 2 : 3;
	p.tile_h_log2 = tile_px == 64 ? 3 : 2;
	const int tw = 1 << p.tile_w_log2, th = 1 << p.tile_h_log2;
	const dim3 grid((p.width + 2 * tw - 1) / (2 * tw), (p.out_rows + 2 * th - 1) / (2 * th));
	const size_t lds = skr_wave_lds_bytes(p);
	const bool occ3 = wave_occ_for(p) == 3;
	// without --gillum shade() never recurses (raytrace.h:208-218), and without spheres nothing is ever
	// hit that would: every depth is then the depth-1 instance
	switch((p.monte_carlo && p.n_spheres > 0) ? p.max_depth : 1)
	{
		case 1:
		{
			const size_t lds1 = ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + (size_t) 4 * Cfg<3>::DEPTH1_WAVE_FLOATS * sizeof(float);
			// measured: config 2 (spheres2 --jsample 5) 1.83 -> 1.63 ms with the small allocation (4+ waves per SIMD), but the
			// triangle walk of dragon 1.28 -> 1.36 ms (1 / 2 / 3 / 4+ waves per SIMD: 2.05 / 1.31 / 1.28 / 1.36 ms): meshes keep the large one
			return p.n_tris > 0 ? launch_wave_depth<1, 3, true>(p, grid, lds, stream) : launch_wave_depth<1, 3, false>(p, grid, lds1, stream);
		}
		case 2: return occ3 ? launch_wave_depth<2, 3>(p, grid, lds, stream) : launch_wave_depth<2, 2>(p, grid, lds, stream);
		case 3: return occ3 ? launch_wave_depth<3, 3>(p, grid, lds, stream) : launch_wave_depth<3, 2>(p, grid, lds, stream);
		default: return hipErrorInvalidValue;
	}
}

// ---- parent-queue pipeline: host side ----
// Used for --gillum trees (depth 2..3) unless SKR_PIPELINE=mega asks for the single megakernel.
bool skr_queue_selected(const RenderParams &p)
{
	if(p.sw.pipeline == SKR_PIPE_MEGA) return false;
	if(p.sw.pipeline == SKR_PIPE_QUEUE) return skr_wave_supported(p) && p.monte_carlo && p.max_depth >= 2 && p.num_path_traces > 0;
	return skr_wave_supported(p) && p.monte_carlo && p.n_spheres > 0 && p.max_depth >= 2 && p.num_path_traces > 0;
}

// scratch the caller must provide (api.cpp allocates it once per renderer and keeps it)
void skr_queue_scratch_bytes(const RenderParams &p, size_t *parent_bytes, size_t *acc_bytes)
{
	const size_t pixels = (size_t) p.width * p.out_rows;
	// parent records, then the GI kernel's per-wave level-1 slot scratch (768 workgroups x 4 waves x 32 parents x N x float3)
	*parent_bytes = pixels * 64 + (size_t) 256 * 3 * 4 * 32 * 3 * sizeof(float) * (size_t) (p.num_path_traces > 0 ? p.num_path_traces : 1);
	*acc_bytes = p.grid_size > 0 ? pixels * 12 : 0;
}

// ---- level-queue pipeline: selection, scratch, launch
static uint32_t levels_band_rows(const RenderParams &p)
{ // rows per band: the level-1 records of a band (64 B x 2 x width x N / 2 per row, every pixel a parent, every child a hit)
  // stay within ~3 GiB; a 1080p --gillum 16 frame is one band, 4K --gillum 64 works in bands of ~200 rows
	uint64_t budget = 3ull << 30;
	if(p.sw.budget_mb) budget = (uint64_t) p.sw.budget_mb << 20; // tests: force several bands
	const uint64_t per_row = (uint64_t) p.width * (uint64_t) (((p.num_path_traces + 1) >> 1) * 2) * 64;
	uint64_t rows = budget / (per_row ? per_row : 1);
	rows = rows / 16 * 16;
	if(rows < 16) rows = 16;
	return (uint32_t) (rows < p.out_rows ? rows : p.out_rows);
}
static uint64_t levels_pairs_max(const RenderParams &p, uint32_t rows)
{ // every pixel of the band could be a parent
	return (uint64_t) p.width * rows * (uint64_t) (((p.num_path_traces > 0 ? p.num_path_traces : 1) + 1) >> 1);
}
static uint64_t levels_tasks_max(const RenderParams &p, uint32_t rows)
{ // record capacity: a region receives at most 128 hits from each of its skr_level1_kernel waves (64 sibling pairs)
	const uint64_t waves = (levels_pairs_max(p, rows) + 63) / 64;
	return (waves + SKR_P1_REGIONS - 1) / SKR_P1_REGIONS * 128 * SKR_P1_REGIONS;
}

// Default for --gillum at depth 3 on sphere scenes (measured against the parent-queue pipeline, 1080p: headline 2.54 ->
// 2.42 ms, no shadows 1.98 -> 1.85, bear 1.02 -> 0.63, gillum 4 / 8 / 64 / 255: -37 / -16 / -17 / -46 %, one rank's 1/8 frame
// 0.46 -> 0.41); scenes with triangle meshes (more than 64 triangles) stay on the parent-queue pipeline (test.scn: 1.5 vs 2.8 ms —
// their rounds are long and few).  SKR_PIPELINE=levels | queue | mega forces one.
bool skr_levels_selected(const RenderParams &p)
{
	const bool forced = p.sw.pipeline == SKR_PIPE_LEVELS;
	if(!forced) return false; // round 1's pipeline: A/B runs only (the node pipeline of render_nodes.hip replaced it)
	if(!(skr_wave_supported(p) && p.monte_carlo && p.n_spheres > 0 && p.max_depth == 3 && p.num_path_traces > 0 && p.num_path_traces <= 255)) return false;
	if(p.n_tris > 64 && !forced) return false; // a handful of triangles costs nothing (spheres1: 1.31 -> 1.18 ms); meshes stay on the parent queue
	return levels_tasks_max(p, levels_band_rows(p)) * 3 < (1ull << 30);
}

bool skr_levels_scratch_bytes(RenderParams &p, size_t *p1_bytes, size_t *slot1_bytes)
{
	if(!skr_levels_selected(p)) return false;
	const uint32_t rows = levels_band_rows(p);
	const uint64_t t = levels_tasks_max(p, rows);
	p.p1_region_cap = (uint32_t) (t / SKR_P1_REGIONS);
	*p1_bytes = (size_t) t * 64;
	*slot1_bytes = (size_t) p.width * rows * (size_t) p.num_path_traces * 12;
	return true;
}

hipError_t skr_launch_levels(const RenderParams &p_in, hipStream_t stream, const SkrTimingHook *hook)
{
	RenderParams p = p_in;
	const int nsamp = p.grid_size > 0 ? p.grid_size * p.grid_size : 1;
	const size_t lds_scene = ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 32;
	const size_t lds_leaf = lds_scene + (size_t) 4 * LEAF_WAVE_FLOATS * sizeof(float);
	const bool tris = p.n_tris > 0;
	hipError_t e = hipFuncSetAttribute(tris ? reinterpret_cast<const void *>(skr_leaf_kernel<true>) : reinterpret_cast<const void *>(skr_leaf_kernel<false>),
									   hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_leaf);
	if(e != hipSuccess) return e;
	const uint32_t band = levels_band_rows(p);
	for(int s = 0; s < nsamp; s++)
	{
		p.aa_index = (uint32_t) s;
		for(uint32_t row0 = 0; row0 < p.out_rows; row0 += band)
		{ // every band is a complete pass: its parents, their level-1 hits, their pixels
			p.band_row0 = row0;
			p.band_rows = p.out_rows - row0 < band ? p.out_rows - row0 : band;
			const uint64_t pixels = (uint64_t) p.width * p.band_rows;
			const bool last = s == nsamp - 1 && row0 + band >= p.out_rows;
			e = hipMemsetAsync(p.qctr, 0, (SKR_PULL_QUEUES + 2 + 2 * SKR_P1_REGIONS) * SKR_PULL_STRIDE * sizeof(uint32_t), stream);
			if(e != hipSuccess) return e;
			if(tris) hipLaunchKernelGGL(skr_primary_kernel<true>, dim3((p.width + 15) / 16, (p.band_rows + 15) / 16), dim3(256), lds_scene, stream, p);
			else hipLaunchKernelGGL(skr_primary_kernel<false>, dim3((p.width + 15) / 16, (p.band_rows + 15) / 16), dim3(256), lds_scene, stream, p);
			const dim3 grid1((unsigned) ((levels_pairs_max(p, p.band_rows) + 255) / 256));
			if(tris) hipLaunchKernelGGL(skr_level1_kernel<true>, grid1, dim3(256), lds_scene, stream, p);
			else hipLaunchKernelGGL(skr_level1_kernel<false>, grid1, dim3(256), lds_scene, stream, p);
			// the leaf kernel is the dominant one: time it alone (the last band's launch when there are several)
			if(last) skr_hook_start(hook, stream);
			if(tris) hipLaunchKernelGGL(skr_leaf_kernel<true>, dim3(256u * 4u), dim3(256), lds_leaf, stream, p); // every workgroup resident
			else hipLaunchKernelGGL(skr_leaf_kernel<false>, dim3(256u * 4u), dim3(256), lds_leaf, stream, p);
			if(last) skr_hook_stop(hook, stream);
			hipLaunchKernelGGL(skr_finalize_kernel, dim3((unsigned) ((pixels + 255) / 256)), dim3(256), 0, stream, p);
			e = hipGetLastError();
			if(e != hipSuccess) return e;
		}
	}
	if(p.grid_size > 0)
	{
		const size_t n = (size_t) p.width * p.out_rows;
		hipLaunchKernelGGL(skr_resolve_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, p);
		return hipGetLastError();
	}
	return hipSuccess;
}

template <int D, int OCC, bool TRIS>
static hipError_t launch_gi_t(const RenderParams &p, size_t lds, hipStream_t stream)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(skr_gi_kernel<D, OCC, TRIS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	if(e != hipSuccess) return e;
	const dim3 grid(256u * (uint32_t) OCC); // every workgroup resident: 256 CUs x OCC workgroups of 4 waves
	hipLaunchKernelGGL((skr_gi_kernel<D, OCC, TRIS>), grid, dim3(256), lds, stream, p);
	return hipGetLastError();
}
template <int D, int OCC>
static hipError_t launch_gi(const RenderParams &p, size_t lds, hipStream_t stream)
{
	return p.n_tris > 0 ? launch_gi_t<D, OCC, true>(p, lds, stream) : launch_gi_t<D, OCC, false>(p, lds, stream);
}

hipError_t skr_launch_queue(const RenderParams &p_in, hipStream_t stream, const SkrTimingHook *hook)
{
	RenderParams p = p_in;
	p.gi_groups_per_slot = 8u; // 2..8 groups per wave slot and multiples of 4 or 8 measured flat on 1/8..1/32 frames
	p.gi_group_round = 8u;
	const int nsamp = p.grid_size > 0 ? p.grid_size * p.grid_size : 1;
	const size_t lds1 = ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 32;
	const bool occ3 = wave_occ_for(p) == 3;
	const size_t lds2 = wave_block_lds(p, occ3 ? 3 : 2, true);
	const dim3 grid1((p.width + 15) / 16, (p.out_rows + 15) / 16);
	for(int s = 0; s < nsamp; s++)
	{
		p.aa_index = (uint32_t) s;
		hipError_t e = hipMemsetAsync(p.qctr, 0, (SKR_PULL_QUEUES + 1) * SKR_PULL_STRIDE * sizeof(uint32_t), stream);
		if(e != hipSuccess) return e;
		if(p.n_tris > 0) hipLaunchKernelGGL(skr_primary_kernel<true>, grid1, dim3(256), lds1, stream, p);
		else hipLaunchKernelGGL(skr_primary_kernel<false>, grid1, dim3(256), lds1, stream, p);
		e = hipGetLastError();
		if(e != hipSuccess) return e;
		// the GI kernel is the dominant one: time it alone (last sample's launch when there are several)
		if(s == nsamp - 1) skr_hook_start(hook, stream);
		if(p.max_depth == 2) e = occ3 ? launch_gi<2, 3>(p, lds2, stream) : launch_gi<2, 2>(p, lds2, stream);
		else e = occ3 ? launch_gi<3, 3>(p, lds2, stream) : launch_gi<3, 2>(p, lds2, stream);
		if(s == nsamp - 1) skr_hook_stop(hook, stream);
		if(e != hipSuccess) return e;
	}
	if(p.grid_size > 0)
	{
		const size_t n = (size_t) p.width * p.out_rows;
		hipLaunchKernelGGL(skr_resolve_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, p);
		return hipGetLastError();
	}
	return hipSuccess;
}

// ---- launch wrappers for the node pipeline (render_nodes.hip)
hipError_t skr_launch_primary(const RenderParams &p, dim3 grid, size_t lds, hipStream_t stream)
{
	if(p.n_tris > 0) hipLaunchKernelGGL(skr_primary_kernel<true>, grid, dim3(256), lds, stream, p);
	else hipLaunchKernelGGL(skr_primary_kernel<false>, grid, dim3(256), lds, stream, p);
	return hipGetLastError();
}
hipError_t skr_launch_resolve(const RenderParams &p, hipStream_t stream)
{
	const size_t n = (size_t) p.width * p.out_rows;
	hipLaunchKernelGGL(skr_resolve_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, p);
	return hipGetLastError();
}
