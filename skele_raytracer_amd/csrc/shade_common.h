// Device functions shared by the render kernels: scene view, traversal,
// shading (reference src/raytrace.h, blinn_phong.h, utils.h restated for gfx950).
#pragma once

#include "device_math.h"
#include "render_params.h"
#include "tri_chunks.h"

namespace {


// Scene as the kernel sees it: pointers into LDS (spheres, materials, lights)
// and HBM (triangles, read with wave-uniform addresses).
struct SceneView {
	const float4 *geom; // LDS  centre.xyz, r*r; ns + 1 entries (the last is a pad for the loops' prefetch)
	const float4 *amb;  // LDS  La*ka, .w = phong power
	const float4 *kd;   // LDS
	const float4 *ks;   // LDS
	const float4 *lights; // LDS [2i] position [2i+1] colour
	const float4 *tris; // HBM  [3i] v0 [3i+1] e1 [3i+2] e2
	int ns, nt, nl;
	const float4 *chunks; // HBM  culling data of the triangle walk (scene_host.h): the skip-linked tree, 2 float4 per node,
	                      //      + a pad node, then one conservative sphere (centre, radius^2) per chunk of triangles
	int nchunks;          // nodes in the tree; 0 = walk every triangle
	int chunk;            // triangles per chunk sphere (tri_chunks.h)
	int cones;            // some entry carries a tight radius for non-grazing rays
	unsigned long long *tri_work; // HBM, or null (not counting): SKR_TRI_WORK_SHARDS x {culling-sphere tests, triangle tests} the walks executed (lanes that needed them)
	const float4 *geom_u; // HBM: the same rows as `geom`, for the loops that walk the spheres in order with a wave-uniform index (sphere_rows)
};
typedef float skr_v4f __attribute__((ext_vector_type(4)));
// One aligned 16-byte row of a table that no kernel writes, at a wave-uniform index, through the constant address space: that is what
// makes the compiler take the scalar path (s_load_dwordx4 into SGPRs) — through a plain pointer it cannot prove that no store of the
// kernel aliases the row and issues a vector load with a uniform address.
SKR_DEV float4 load_const4(const float4 *base, int i)
{
	const skr_v4f __attribute__((address_space(4))) *q = (const skr_v4f __attribute__((address_space(4))) *) (unsigned long long) base;
	const skr_v4f v = q[i];
	return make_float4(v.x, v.y, v.z, v.w);
}

// The sphere loops of the level pipelines (closest_pair_deferred, occluded_pair<false>): `test(row, index)` for every sphere in order,
// SKR_SPHERE_TRIP spheres per trip, the rows of the next trip asked for a trip ahead with ONE scalar load (s_load_dwordx8 / x16: the rows
// are consecutive).  The rows come through the scalar cache into SGPRs (geom_u), so the spheres in flight cost no vector register for
// their data — and several independent tests per trip are what a SIMD with four waves wants.  Headline leaf kernel, same box, spheres
// per trip 1 / 2 / 3 / 4 / 5 / 6 / 8: 1.240 / 1.205 / 1.207 / 1.19 / 1.240 / 1.224 / 1.282 ms (LDS rows, one per trip: 1.25; LDS rows,
// two per trip: 1.35 — the second sphere's registers spill; scalar rows fetched one by one with a bounds test each: 1.25).  The lanes of
// these kernels are incoherent; the direct kernel's coherent loops keep LDS rows and their wave-wide early exit.
#ifndef SKR_SPHERE_TRIP
#define SKR_SPHERE_TRIP 4
#endif
template <int K, typename F, typename G>
SKR_DEV void table_rows(const float4 *base, int n, F test, G go_on)
{
	auto row = [&](int i) { return load_const4(base, i); }; // (up to 2 K - 1 rows behind the table are asked for and never used: the tables are padded for them, api.cpp)
	float4 nx[K];
#pragma unroll
	for(int k = 0; k < K; k++) nx[k] = row(k);
	int i = 0;
	for(; i + K <= n; i += K)
	{
		float4 g[K];
#pragma unroll
		for(int k = 0; k < K; k++)
		{
			g[k] = nx[k];
			nx[k] = row(i + K + k);
		}
		__builtin_amdgcn_sched_barrier(0); // the next trip's rows are asked for here, a whole trip ahead of their use
#pragma unroll
		for(int k = 0; k < K; k++) test(g[k], i + k);
		if(!go_on()) return;
	}
#pragma unroll
	for(int k = 0; k < K - 1; k++)
		if(i + k < n) test(nx[k], i + k);
}
template <typename F>
SKR_DEV void sphere_rows(const SceneView &sv, F test)
{
	table_rows<SKR_SPHERE_TRIP>(sv.geom_u, sv.ns, test, [] { return true; });
}

// Row i of the mesh tables (triangles, culling data: HBM, never written by a kernel), i wave-uniform: one s_load_dwordx4 into SGPRs.
// Through the plain pointer the compiler issues a VECTOR load with a uniform address (it cannot prove that no store of the kernel
// aliases the table): 26 M vector-memory instructions per dragon frame, found in the round-3 counters (SQ_INSTS_VMEM against
// SQ_INSTS_SMEM = 0.5 M) under a comment that said "scalar loads".  Through the constant address space: dragon.scn 1080p 1.306 ->
// 1.18 ms, with --shade-triangles 2.99 -> 2.10 ms; test.scn 640x360 --gillum 4 (a dozen triangles, incoherent lanes) 1.10 -> 1.16 ms.
#ifndef SKR_MESH_SMEM
#define SKR_MESH_SMEM 1
#endif
SKR_DEV float4 mesh_row(const float4 *base, int i)
{
#if SKR_MESH_SMEM
	return load_const4(base, i);
#else
	return base[i];
#endif
}

// What a triangle walk executed, counted per wave on the scalar unit (population counts of lane masks the walk forms anyway) and added
// to one of SKR_TRI_WORK_SHARDS words by one lane when the walk ends: bench.py's FP32-VALU figure for mesh scenes is built from
// these counts, not from the 10 002 tests per ray the reference's loop runs (raytrace.h:171-186).
#define SKR_TRI_WORK_SHARDS 256u
SKR_DEV void tri_work_add(const SceneView &sv, uint32_t n_cull, uint32_t n_tri)
{
	if(sv.tri_work && (n_cull | n_tri))
	{
		const unsigned long long m = __ballot(true);
		if(__builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)) == 0)
		{
			unsigned long long *w = sv.tri_work + 2u * ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (SKR_TRI_WORK_SHARDS - 1u));
			atomicAdd(&w[0], (unsigned long long) n_cull);
			atomicAdd(&w[1], (unsigned long long) n_tri);
		}
	}
}

struct Counters {
	uint32_t rays, hits, shadow_rays;
	uint32_t shadow_tests; // ray-sphere tests of utils.h:42-58 as the reference runs them: up to and including the first occluder
};

struct RayConst { // per-ray invariants of utils.h:113-121
	f3 o, d;
	float two_a, four_a;
};

SKR_DEV RayConst make_ray(f3 o, f3 d)
{
	const float a = dot3(d, d);
	return RayConst{o, d, 2 * a, 4 * a};
}

// raytrace.h:152-165: closest accepted sphere (strict <, first index wins ties),
// evaluated exactly as written: the binary64 root for every sphere with D >= 0.
SKR_DEV int closest_sphere_exact(const SceneView &sv, const RayConst &r, float &tmin)
{
	int best = -1;
	tmin = __builtin_inff();
	float4 g_next = sv.geom[0];
	for(int i = 0; i < sv.ns; i++)
	{
		const float4 g = g_next;
		g_next = sv.geom[i + 1]; // software prefetch; geom[] carries one pad entry
		const float t = sphere_distance(r.o, r.d, r.two_a, r.four_a, g);
		if(accept_distance(t) && t < tmin)
		{
			tmin = t;
			best = i;
		}
	}
	return best;
}

// Same result through the binary32 brackets of device_math.h: the winner is
// known as soon as its bracket lies strictly below every other accepted
// sphere's; its exact t2 is then formed once.  Overlapping brackets (two
// surfaces within ~1e-6 relative of each other along the ray) fall back to the
// exact loop for that lane.
SKR_DEV int closest_sphere(const SceneView &sv, const RayConst &r, float &tmin)
{
	const RayFilt f = make_filt(r.d);
	int best = -1;
	float best_lo = __builtin_inff(), best_hi = __builtin_inff(), others_lo = __builtin_inff();
	float best_b = 0.0f, best_D = 0.0f;
	sphere_rows(sv, [&](const float4 g, int i)
	{
		float lo, hi, b, D;
		if(sphere_bracket(r.o, r.d, f, g, lo, hi, b, D))
		{
			if(hi < best_hi)
			{
				others_lo = __builtin_fminf(others_lo, best_lo);
				best_lo = lo;
				best_hi = hi;
				best = i;
				best_b = b;
				best_D = D;
			}
			else others_lo = __builtin_fminf(others_lo, lo);
		}
	});
	tmin = __builtin_inff();
	if(best >= 0)
	{
		if(others_lo > best_hi) tmin = (best_lo == best_hi) ? best_lo : near_root_exact(f.two_a, best_b, best_D);
		else best = closest_sphere_exact(sv, r, tmin);
	}
	return best;
}

// closest_sphere() for rays that all start at ONE point (the camera: main.cpp:140-182), with e = o - C and c = e.e - r^2 of utils.h:115-118
// formed once per sphere and renderer (ec[i] = {e.xyz, c}: the same subtractions, products and sums in the same order, on the device, so the
// same floats: skr_camec_kernel, render_wave.hip) instead of once per ray: 9 of the ~17 instructions a sphere costs a ray.  The rows are
// read like the sphere rows of the level pipelines: scalar loads, several per trip (table_rows).
#ifndef SKR_CAMERA_TRIP
#define SKR_CAMERA_TRIP 4 // spheres per trip of closest_sphere_from: config 2 (15 spheres) 1.312 / 1.284 / 1.19 ms at 1 / 2 / 4 (with the shadow loop at the same count), bear.scn (31) 0.518 / 0.500 / 0.451
#endif
SKR_DEV int closest_sphere_from(const SceneView &sv, const float4 *ec, const RayConst &r, float &tmin)
{
	const RayFilt f = make_filt(r.d);
	int best = -1;
	float best_lo = __builtin_inff(), best_hi = __builtin_inff(), others_lo = __builtin_inff();
	float best_b = 0.0f, best_D = 0.0f;
	table_rows<SKR_CAMERA_TRIP>(ec, sv.ns, [&](const float4 q, int i)
	{
		float lo, hi, b, D;
		if(bracket_from_ec(ld3(q), q.w, r.d, f, lo, hi, b, D))
		{
			if(hi < best_hi)
			{
				others_lo = __builtin_fminf(others_lo, best_lo);
				best_lo = lo;
				best_hi = hi;
				best = i;
				best_b = b;
				best_D = D;
			}
			else others_lo = __builtin_fminf(others_lo, lo);
		}
	}, [] { return true; });
	tmin = __builtin_inff();
	if(best >= 0)
	{
		if(others_lo > best_hi) tmin = (best_lo == best_hi) ? best_lo : near_root_exact(f.two_a, best_b, best_D);
		else best = closest_sphere_exact(sv, r, tmin);
	}
	return best;
}

// The conservative line-sphere test of the culling data (scene_host.cpp build_triangle_chunks): false only where
// no triangle below the entry can accept this lane's line.  A = {centre, R^2}; B = {axis / kappa, R_tight^2}: a ray
// that is not grazing for the entry's (nearly coplanar) triangles, (d . axis / kappa)^2 >= d . d, is held to the
// tight radius.  NaN anywhere => true.
template <bool CONES>
SKR_DEV bool line_touches(const RayConst &r, float dd, float4 A, float4 B)
{
	const f3 e = ld3(A) - r.o;
	const f3 cr = cross3(e, r.d);
	float R2 = A.w;
	if constexpr(CONES)
	{
		const float gb = dot3(r.d, ld3(B));
		R2 = (gb * gb >= dd) ? B.w : A.w;
	}
	return !(dot3(cr, cr) > R2 * dd); // |e x d|^2 <= R^2 |d|^2
}

// A line that misses a conservative sphere cannot pass the test for any triangle below it, so a node or chunk that
// no lane's line touches is skipped whole.  The levels above the chunks are stored depth-first with skip links:
// one wave-uniform index, no stack; both possible successors are fetched (scalar loads: mesh_row) while the sphere is tested.
template <bool CONES, bool COUNT>
SKR_DEV bool tree_walk(const SceneView &sv, const RayConst &r, float tmin)
{
	bool hit = false;
	const float dd = r.two_a * 0.5f; // dot(d, d)
	int i = 0;
	uint32_t n_cull = 0, n_tri = 0; // (wave-uniform: scalar registers)
	const float4 *chunk_ent = sv.chunks + 3 * (sv.nchunks + 1); // behind the nodes and their pad
	float4 A = mesh_row(sv.chunks, 0), B = mesh_row(sv.chunks, 1), lk = mesh_row(sv.chunks, 2);
	while(i < sv.nchunks)
	{
		const int i_out = __float_as_int(lk.x);
		// first child (or the next node after a height-1 node) and next sibling (padded past the end)
		const float4 A_in = mesh_row(sv.chunks, 3 * i + 3), B_in = mesh_row(sv.chunks, 3 * i + 4), lk_in = mesh_row(sv.chunks, 3 * i + 5);
		const float4 A_out = mesh_row(sv.chunks, 3 * i_out), B_out = mesh_row(sv.chunks, 3 * i_out + 1), lk_out = mesh_row(sv.chunks, 3 * i_out + 2);
		if(COUNT) n_cull += (uint32_t) __popcll(__ballot(!hit));
		const bool enter = __any(!hit && line_touches<CONES>(r, dd, A, B));
		const int count = __float_as_int(lk.z);
		if(enter && count > 0)
		{ // height 1: its chunk entries are contiguous — tight loop, next entry prefetched
			const int c0 = __float_as_int(lk.y), c1 = c0 + count;
			float4 cA_next = mesh_row(chunk_ent, 2 * c0), cB_next = mesh_row(chunk_ent, 2 * c0 + 1);
			for(int c = c0; c < c1; c++)
			{
				const float4 cA = cA_next, cB = cB_next;
				cA_next = mesh_row(chunk_ent, 2 * c + 2);
				cB_next = mesh_row(chunk_ent, 2 * c + 3);
				if(COUNT) n_cull += (uint32_t) __popcll(__ballot(!hit));
				const bool mine = !hit && line_touches<CONES>(r, dd, cA, cB);
				if(__any(mine))
				{
					const int i0 = c * sv.chunk, i1 = (i0 + sv.chunk < sv.nt) ? i0 + sv.chunk : sv.nt;
					// (one triangle per scalar-cache round trip, the next one asked for meanwhile; a whole 4-triangle chunk asked for at once was measured
					// slower: dragon 1.19 -> 1.27 ms, 48 more SGPRs live)
					float4 n0 = mesh_row(sv.tris, 3 * i0), n1 = mesh_row(sv.tris, 3 * i0 + 1), n2 = mesh_row(sv.tris, 3 * i0 + 2);
					for(int k = i0; k < i1; k++)
					{
						const f3 v0 = ld3(n0), e1 = ld3(n1), e2 = ld3(n2);
						n0 = mesh_row(sv.tris, 3 * k + 3);
						n1 = mesh_row(sv.tris, 3 * k + 4);
						n2 = mesh_row(sv.tris, 3 * k + 5);
						float t;
						if(COUNT) n_tri += (uint32_t) __popcll(__ballot(mine && !hit));
						if(mine && !hit && triangle_hit(r.o, r.d, v0, e1, e2, t) && t < tmin) hit = true;
					}
				}
			}
			if(__all(hit)) break;
		}
		i = enter ? i + 1 : i_out;
		A = enter ? A_in : A_out;
		B = enter ? B_in : B_out;
		lk = enter ? lk_in : lk_out;
	}
	if(COUNT) tri_work_add(sv, n_cull, n_tri);
	return hit;
}

// raytrace.h:171-186.  The outcome is binary: once a triangle passes with
// t < min_distance the sample is black (:221-224) whatever comes later, so a
// lane stops testing at its first accepted triangle and the wave leaves the
// loop when every active lane has.
SKR_DEV bool any_triangle_closer(const SceneView &sv, const RayConst &r, float tmin)
{
	if(sv.nchunks > 0)
	{ // (counting costs the dragon walk 19 %: the counting instantiation runs only while sv.tri_work is set — skr_renderer_count_triangle_work)
		if(__builtin_expect(sv.tri_work != nullptr, 0)) return sv.cones ? tree_walk<true, true>(sv, r, tmin) : tree_walk<false, true>(sv, r, tmin);
		return sv.cones ? tree_walk<true, false>(sv, r, tmin) : tree_walk<false, false>(sv, r, tmin);
	}
	bool hit = false;
	uint32_t n_tri = 0;
	// wave-uniform addresses => scalar loads; triangle i+1 is fetched while i is tested
	// (tris[] carries one pad triangle so the prefetch needs no bounds test)
	float4 n0 = mesh_row(sv.tris, 0), n1 = mesh_row(sv.tris, 1), n2 = mesh_row(sv.tris, 2);
	for(int i = 0; i < sv.nt; i++)
	{
		const f3 v0 = ld3(n0), e1 = ld3(n1), e2 = ld3(n2);
		n0 = mesh_row(sv.tris, 3 * i + 3);
		n1 = mesh_row(sv.tris, 3 * i + 4);
		n2 = mesh_row(sv.tris, 3 * i + 5);
		float t;
		if(sv.tri_work) n_tri += (uint32_t) __popcll(__ballot(!hit));
		if(!hit && triangle_hit(r.o, r.d, v0, e1, e2, t) && t < tmin) hit = true;
		if((i & 7) == 7 && __all(hit)) break;
	}
	tri_work_add(sv, 0u, n_tri);
	return hit;
}

// utils.h:42-58: any sphere with 1 < t < inf along the (unbounded) shadow ray; two lights at a
// time, because both shadow rays start at the same point and share e and c per sphere.
// COHERENT: the lanes of the wave are neighbouring pixels (the direct kernel's 8x8 tiles), where a whole wave in one shadow is common
// and the loop is left once every lane's rays are occluded.  The level pipelines' lanes are hits from all over the scene: the wave-wide
// test never fires there and costs a branch and half a dozen instructions per sphere (headline leaf kernel 1.282 -> 1.247 ms without
// it; two spheres per trip written out by hand, on top: 1.287 ms — eight more live registers, 47 spilled instead of 24).
#ifndef SKR_COHERENT_TRIP
#define SKR_COHERENT_TRIP 4 // spheres per trip of the coherent shadow loop (the wave-wide exit is looked at once per trip)
#endif
template <bool COHERENT>
SKR_DEV void occluded_pair(const SceneView &sv, f3 P, f3 L0, f3 L1, bool second, bool &occ0, bool &occ1, uint32_t &tests)
{
	const f3 o = add_scalar(P, 0.000001f);
	const RayPair rp = make_pair(L0, L1);
	const PairAny pa{rp.two_a, rp.two_a * 0.25f, rp.sane0, rp.sane1};
	occ0 = false;
	occ1 = !second;
	auto test = [&](const float4 g, int i)
	{
		const f3 e = o - ld3(g);
		const float c = dot3(e, e) - g.w;
		f2 b, D;
		pair_bD(rp, e, c, b, D);
		// b >= 0 or D < 0 (or NaN): certain miss for that ray
		const bool cand0 = !occ0 && (D.x >= 0.0f) && (b.x < 0.0f);
		const bool cand1 = !occ1 && (D.y >= 0.0f) && (b.y < 0.0f);
		DIAG_WAVE(4, 1);
		if(cand0 || cand1)
		{
			DIAG_WAVE(5, 1);
			DIAG_LANES(6);
			f2 m, al, rl;
			pair_any_m(pa, b, m, al, rl);
			if(cand0)
			{
				occ0 = any_decide(pa.sane0, pa.two_a.x, pa.quarter.x, b.x, D.x, m.x, al.x, rl.x);
				if(occ0) tests += (uint32_t) i + 1u; // the reference's loop returns here (utils.h:52-55)
			}
			if(cand1)
			{
				occ1 = any_decide(pa.sane1, pa.two_a.y, pa.quarter.y, b.y, D.y, m.y, al.y, rl.y);
				if(occ1) tests += (uint32_t) i + 1u;
			}
		}
	};
	if(COHERENT) table_rows<SKR_COHERENT_TRIP>(sv.geom_u, sv.ns, test, [&] { return !__all(occ0 && occ1); }); // (the wave-wide exit, once per trip)
	else sphere_rows(sv, test);
	if(!occ0) tests += (uint32_t) sv.ns;
	if(second && !occ1) tests += (uint32_t) sv.ns;
	if(!second) occ1 = false;
}

struct LightTerm { // the per-light quantities of blinn_phong.h:67-72 / :100-117
	f3 L, lc;
	float intensity; // 1 / powf(|Lp - P|, 2) (== 1 / (d * d): SURVEY.md 8c); exactly 1 for a directional light
};

SKR_DEV LightTerm light_term(const SceneView &sv, int i, f3 P)
{
	LightTerm t;
	const float4 lp4 = sv.lights[2 * i];
	t.lc = ld3(sv.lights[2 * i + 1]);
	if(lp4.w != 0.0f)
	{ // a directional light (--strict-scn only; blinn_phong.h:81-82,126-128): L = normalize(direction) and no 1/d^2 — the intensity
	  // factor of the point-light expression is exactly 1, and x * 1 == x
		t.L = normalize3(ld3(lp4));
		t.intensity = 1.0f;
		return t;
	}
	const f3 to_l = ld3(lp4) - P;
	const LenTerms lt = len_terms<true>(sqr3(to_l));
	t.L = to_l * lt.inv;
	t.intensity = lt.inv2;
	return t;
}

// raytrace.h:36-44 = bp::ambient (blinn_phong.h:13) + diffuse (:47) + specular (:90).
// The reference casts the same shadow ray in diffuse and again in specular; one cast serves both.
// (kd, ks, ambp = {La * ka, power}: the material rows of the surface hit)
template <bool COHERENT>
SKR_DEV f3 direct_light_of(const SceneView &sv, const RenderParams &p, f3 kd, f3 ks, float4 ambp, f3 P, f3 N, Counters &cn)
{
	f3 diffuse = mk3(0, 0, 0), specular = mk3(0, 0, 0);
	const f3 view = normalize3(p.cam_pos - P); // always the camera (blinn_phong.h:93)
	for(int i = 0; i < sv.nl; i += 2)
	{
		const bool second = i + 1 < sv.nl;
		const LightTerm t0 = light_term(sv, i, P), t1 = light_term(sv, second ? i + 1 : i, P);
		bool occ0 = false, occ1 = false;
		if(p.use_shadows)
		{
			cn.shadow_rays += second ? 2u : 1u;
			occluded_pair<COHERENT>(sv, P, t0.L, t1.L, second, occ0, occ1, cn.shadow_tests);
		}
		auto add_light = [&](const LightTerm &t, bool lit)
		{
			if(lit)
			{
				diffuse = diffuse + ((kd * t.lc) * t.intensity) * max0(dot3(N, t.L));
				const f3 vl = view + t.L;
				const f3 H = vl / length3(vl);
				specular = specular + ((ks * t.lc) * t.intensity) * powf_spec(max0(dot3(N, H)), ambp.w, p.pow_steps);
			}
		};
		add_light(t0, !occ0);
		add_light(t1, second && !occ1);
	}
	f3 total = mk3(0, 0, 0);
	total = total + ld3(ambp);
	total = total + diffuse;
	total = total + specular;
	return total;
}

template <bool COHERENT>
SKR_DEV f3 direct_light(const SceneView &sv, const RenderParams &p, int sph, f3 P, f3 N, Counters &cn)
{
	return direct_light_of<COHERENT>(sv, p, ld3(sv.kd[sph]), ld3(sv.ks[sph]), sv.amb[sph], P, N, cn);
}

// raytrace.h:22-30 + :117-125: hemisphere sample and the reference's basis mix
// (perp_to_both.y/.z where perp_to_normal.y/.z belongs — kept), for the two sibling rays of a pair at once in packed binary32
// (every component is the one-ray expression: v_pk_* round each half like the scalar instruction).
#ifndef SKR_GI_INLINE
#define SKR_GI_INLINE 0 // 1: inline gi_direction_pair into its callers (A/B builds)
#endif
struct DirPair { f3 d0, d1; }; // (returned by value: in registers, where reference parameters of an out-of-line function go through scratch)
#if SKR_GI_INLINE
SKR_DEV
#else
static __device__ __attribute__((noinline))
#endif
DirPair gi_direction_pair(float r1a, float r2a, float r1b, float r2b, f3 N, f3 nt, f3 nb)
{
	const f2 r1 = f2{r1a, r1b};
	const f2 om = 1.0f - r1 * r1;
	const f2 s_theta = f2{sk_sqrtf(om.x), sk_sqrtf(om.y)};
	// (2.0f*M_PI)*r2 in double, narrowed (raytrace.h:25: `float phi = 2 * M_PI * r2`)
	const f2 phi = f2{(float) ((2.0 * 3.14159265358979323846) * (double) r2a), (float) ((2.0 * 3.14159265358979323846) * (double) r2b)};
	f2 sn, cs;
	sincos_spec2(phi, sn, cs);
	const f2 sx = s_theta * cs, sy = r1, sz = s_theta * sn;
	const f2 x = (sx * nb.x + sy * N.x) + sz * nt.x;
	const f2 y = (sx * nb.y + sy * N.y) + sz * nb.y;
	const f2 z = (sx * nb.z + sy * N.z) + sz * nb.z;
	return DirPair{mk3(x.x, y.x, z.x), mk3(x.y, y.y, z.y)};
}

SKR_DEV f3 gi_direction(float r1, float r2, f3 N, f3 nt, f3 nb)
{
	return gi_direction_pair(r1, r2, r1, r2, N, nt, nb).d0;
}

// ---- --shade-triangles (the rules: include/skr.h skr_options.shade_triangles): the closest accepted triangle ----
struct TriBest {
	float t;  // smallest accepted distance so far (starts at the closest sphere's)
	int file; // index of that triangle in the scene file, -1 = the sphere still wins
	int slot; // its position in tris[]
};

SKR_DEV void tri_consider(const RayConst &r, bool mine, f3 v0, float4 n1, float4 n2, int slot, int from_tri, TriBest &b)
{
	float t;
	if(mine && triangle_hit(r.o, r.d, v0, ld3(n1), ld3(n2), t) && t > 0.0f)
	{
		const int file = __float_as_int(n1.w);
		if(file != from_tri && (t < b.t || (t == b.t && b.file >= 0 && file < b.file)))
		{
			b.t = t;
			b.file = file;
			b.slot = slot;
		}
	}
}

// line_touches() for the closest-hit walk: false also where every hit under the entry would lie BEHIND the running best.  A hit point
// o + t d of an accepted triangle lies inside the entry's sphere (that is what the sphere bounds), so t |d| >= d^ . (C - o) - R:
// with lhs = d . (C - o) - t_best (d . d) the entry cannot hold a nearer hit once lhs > R |d|.  The radii carry 16x the rounding slack
// of the test's own u, v; the float t of a near-degenerate triangle can be off by as much again, so the entry is only skipped
// at lhs > 1.125 R |d| (two slacks to spare).  An equal t must still be visited (the lower file index wins a tie): strict test.
template <bool CONES>
SKR_DEV bool entry_may_hold_nearer(const RayConst &r, float dd, float4 A, float4 B, float t_best)
{
	const f3 e = ld3(A) - r.o;
	const f3 cr = cross3(e, r.d);
	float R2 = A.w;
	if constexpr(CONES)
	{
		const float gb = dot3(r.d, ld3(B));
		R2 = (gb * gb >= dd) ? B.w : A.w;
	}
	const float lim = R2 * dd;
	if(dot3(cr, cr) > lim) return false; // the line misses the sphere (NaN: falls through, "enter")
	const float lhs = dot3(r.d, e) - t_best * dd;
	return !(lhs > 0.0f && lhs * lhs > lim * 1.27f);
}

// The walk of tree_walk() as a closest-hit walk: every chunk whose conservative sphere this lane's line touches in front of its
// running best is tested to the end (the spheres bound the accept test itself, whatever t comes out).
template <bool CONES>
SKR_DEV void tree_walk_closest(const SceneView &sv, const RayConst &r, int from_tri, TriBest &b)
{
	const float dd = r.two_a * 0.5f; // dot(d, d)
	int i = 0;
	uint32_t n_cull = 0, n_tri = 0;
	const float4 *chunk_ent = sv.chunks + 3 * (sv.nchunks + 1);
	float4 A = mesh_row(sv.chunks, 0), B = mesh_row(sv.chunks, 1), lk = mesh_row(sv.chunks, 2);
	while(i < sv.nchunks)
	{
		const int i_out = __float_as_int(lk.x);
		const float4 A_in = mesh_row(sv.chunks, 3 * i + 3), B_in = mesh_row(sv.chunks, 3 * i + 4), lk_in = mesh_row(sv.chunks, 3 * i + 5);
		const float4 A_out = mesh_row(sv.chunks, 3 * i_out), B_out = mesh_row(sv.chunks, 3 * i_out + 1), lk_out = mesh_row(sv.chunks, 3 * i_out + 2);
		if(sv.tri_work) n_cull += (uint32_t) __popcll(__ballot(true));
		const bool enter = __any(entry_may_hold_nearer<CONES>(r, dd, A, B, b.t));
		const int count = __float_as_int(lk.z);
		if(enter && count > 0)
		{
			const int c0 = __float_as_int(lk.y), c1 = c0 + count;
			for(int c = c0; c < c1; c++)
			{
				if(sv.tri_work) n_cull += (uint32_t) __popcll(__ballot(true));
				const bool mine = entry_may_hold_nearer<CONES>(r, dd, mesh_row(chunk_ent, 2 * c), mesh_row(chunk_ent, 2 * c + 1), b.t);
				if(__any(mine))
				{
					const int i0 = c * sv.chunk, i1 = (i0 + sv.chunk < sv.nt) ? i0 + sv.chunk : sv.nt;
					if(sv.tri_work) n_tri += (uint32_t) __popcll(__ballot(mine)) * (uint32_t) (i1 - i0);
					for(int k = i0; k < i1; k++) tri_consider(r, mine, ld3(mesh_row(sv.tris, 3 * k)), mesh_row(sv.tris, 3 * k + 1), mesh_row(sv.tris, 3 * k + 2), k, from_tri, b);
				}
			}
		}
		i = enter ? i + 1 : i_out;
		A = enter ? A_in : A_out;
		B = enter ? B_in : B_out;
		lk = enter ? lk_in : lk_out;
	}
	tri_work_add(sv, n_cull, n_tri);
}

SKR_DEV void closest_triangle(const SceneView &sv, const RayConst &r, int from_tri, TriBest &b)
{
	if(sv.nchunks > 0)
	{
		if(sv.cones) tree_walk_closest<true>(sv, r, from_tri, b);
		else tree_walk_closest<false>(sv, r, from_tri, b);
		return;
	}
	for(int k = 0; k < sv.nt; k++) tri_consider(r, true, ld3(mesh_row(sv.tris, 3 * k)), mesh_row(sv.tris, 3 * k + 1), mesh_row(sv.tris, 3 * k + 2), k, from_tri, b);
}

// ---- --legacy-reflect: the leaf functions of raytrace.h:45-103 ----
// blinn_phong.h:156-184 (its unqualified sqrt is ::sqrt(double); powf(x, 2.0f) == x * x; utils.h:132-146 clamp)
SKR_DEV float legacy_fresnel(f3 dir, f3 N, float mat_ior)
{
	float cos_internal = dot3(dir, N);
	cos_internal = cos_internal < -1.0f ? -1.0f : (cos_internal > 1.0f ? 1.0f : cos_internal);
	float et = 1.0f, ior = mat_ior;
	if(cos_internal > 0)
	{
		const float t = et;
		et = ior;
		ior = t;
	}
	const float sint = (float) ((double) sk_divf(et, ior) * sqrt((double) max0(1.0f - cos_internal * cos_internal)));
	if(sint >= 1.0f) return 1.0f;
	const float cos_theta = (float) sqrt((double) max0(1 - sint * sint));
	cos_internal = __builtin_fabsf(cos_internal);
	const float Rs = sk_divf((ior * cos_internal) - (et * cos_theta), (ior * cos_internal) + (et * cos_theta));
	const float Rp = sk_divf((et * cos_internal) - (ior * cos_theta), (ior * cos_internal) + (et * cos_theta));
	return sk_divf(Rs * Rs + Rp * Rp, 2.0f);
}

// blinn_phong.h:143-153 refraction(): (0,0,0) on total internal reflection
SKR_DEV f3 legacy_refraction_dir(f3 d, f3 N, float mat_ior)
{
	const float dn = dot3(d, N);
	const float k = 1.0f - (mat_ior * mat_ior) * (1.0f - dn * dn);
	return (k < 0.0f) ? mk3(0, 0, 0) : (d * mat_ior - N * (mat_ior * dn + sk_sqrtf(k)));
}
// blinn_phong.h:137-140 reflect_direction(): the LIGHT direction mirrored at the normal
SKR_DEV f3 legacy_reflect_dir(f3 L, f3 N) { return normalize3(L - N * (2.0f * dot3(L, N))); }

SKR_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
	for(int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}


} // namespace
