// Wave-level helpers shared by the streaming kernels (render_wave.hip) and the node pipeline (render_nodes.hip).
#pragma once

#include "shade_common.h"

namespace {

// Diagnostic build only (-DSKR_STAMPS=1): per-phase cycle shares, summed over waves into
// counters[4*SKR_COUNTER_SHARDS + phase].  Compiles to nothing otherwise.
#if defined(SKR_STAMPS) && SKR_STAMPS
#define STAMP_DECL unsigned long long st_t0 = __builtin_readcyclecounter(); unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP_ARG , unsigned long long &st_t0, unsigned long long (&st_acc)[8]
#define STAMP_PASS , st_t0, st_acc
#define STAMP(phase) do { const unsigned long long st_now = __builtin_readcyclecounter(); st_acc[phase] += st_now - st_t0; st_t0 = st_now; } while(0)
#else
#define STAMP_DECL
#define STAMP_ARG
#define STAMP_PASS
#define STAMP(phase)
#endif

SKR_DEV void wave_lds_fence()
{ // producer and consumer lanes are in the same wave: ordering only, no instruction
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}


SKR_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
SKR_DEV f3 shfl3(f3 v, int src) { return mk3(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64)); }
SKR_DEV int lanes_below(unsigned long long m)
{
	return (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
}


// ---- two sibling rays per lane ------------------------------------------------
// Children 2j and 2j+1 of a node share their origin (raytrace.h:128), one Philox call
// (DESIGN.md "RNG") and, per sphere, e = o - C and c = e.e - r^2; the per-ray part runs in
// packed binary32 (device_math.h RayPair).  Same values as child_round(), half the issue slots.
struct BestState {
	int best;
	float lo, hi, others_lo, b, D;
};

SKR_DEV void best_update(BestState &s, bool acc, int i, float lo, float hi, float b, float D)
{
	if(acc)
	{
		if(hi < s.hi)
		{
			s.others_lo = __builtin_fminf(s.others_lo, s.lo);
			s.lo = lo;
			s.hi = hi;
			s.best = i;
			s.b = b;
			s.D = D;
		}
		else s.others_lo = __builtin_fminf(s.others_lo, lo);
	}
}

SKR_DEV void best_resolve(const SceneView &sv, f3 o, f3 d, float four_a, BestState &s)
{
	if(s.best >= 0 && !(s.others_lo > s.hi))
	{ // brackets overlap: the exact loop names the winner; recompute its coefficients
		DIAG_WAVE(8, 1);
		float tmin;
		const RayConst r = make_ray(o, d);
		s.best = closest_sphere_exact(sv, r, tmin);
		const f3 e = o - ld3(sv.geom[s.best]);
		s.b = 2 * dot3(d, e);
		const float c = dot3(e, e) - sv.geom[s.best].w;
		s.D = s.b * s.b - four_a * c;
	}
}

SKR_DEV void closest_pair_deferred(const SceneView &sv, f3 o, f3 d0, f3 d1, bool second, const RayPair &rp, BestState &s0, BestState &s1)
{
	s0 = BestState{-1, __builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f, 0.0f};
	s1 = s0;
	auto test = [&](const float4 g, int i)
	{
		const f3 e = o - ld3(g);
		const float c = dot3(e, e) - g.w;
		f2 b, D;
		pair_bD(rp, e, c, b, D);
		const bool cand0 = (D.x >= 0.0f) && (b.x < 0.0f);
		const bool cand1 = second && (D.y >= 0.0f) && (b.y < 0.0f);
		DIAG_WAVE(0, 1);
		if(cand0 || cand1)
		{
			DIAG_WAVE(1, 1);
			DIAG_LANES(2);
			f2 lo, hi;
			pair_bracket(rp, b, D, lo, hi);
			float l0 = lo.x, h0 = hi.x, l1 = lo.y, h1 = hi.y;
			const bool acc0 = cand0 && bracket_decide(rp.sane0, rp.two_a.x, b.x, D.x, l0, h0);
			const bool acc1 = cand1 && bracket_decide(rp.sane1, rp.two_a.y, b.y, D.y, l1, h1);
			best_update(s0, acc0, i, l0, h0, b.x, D.x);
			best_update(s1, acc1, i, l1, h1, b.y, D.y);
		}
	};
	sphere_rows(sv, test);
	best_resolve(sv, o, d0, rp.four_a.x, s0);
	if(second) best_resolve(sv, o, d1, rp.four_a.y, s1);
}


SKR_DEV void closest_pair(const SceneView &sv, f3 o, f3 d0, f3 d1, bool second, const RayPair &rp, BestState &s0, BestState &s1)
{
	closest_pair_deferred(sv, o, d0, d1, second, rp, s0, s1);
}

// the image row of row `orow` of the compact output (include/skr.h skr_render_tiles / skr_render_tile_list); >= height: no such row
SKR_DEV uint32_t image_row(const RenderParams &p, uint32_t orow)
{
	const uint32_t k = orow / p.tile_rows;
	const uint32_t t = p.tile_table ? p.tile_table[k] : p.first_tile + k * p.tile_stride;
	if(t == 0xFFFFFFFFu) return 0xFFFFFFFFu;
	return t * p.tile_rows + (orow - k * p.tile_rows);
}

SKR_DEV void primary_ray(const RenderParams &p, int x, uint32_t y, uint32_t pixel, uint32_t aa, f3 &dir)
{ // main.cpp:140-182
	float u, v;
	if(p.grid_size > 0)
	{
		uint32_t rnd[4];
		philox4x32(pixel, aa, 0u, 0xFFFFFFFFu, p.seed_lo, p.seed_hi, rnd);
		const float r = u31_to_unit(rnd[0]);
		u = ((2 * (((float) x + r) * p.inv_width) - 1) * p.angle) * p.aspect;
		v = (1 - 2 * (((float) (int) y + r) * p.inv_height)) * p.angle;
	}
	else
	{
		u = (float) (((2 * (((double) x + 0.5) * (double) p.inv_width) - 1) * (double) p.angle) * (double) p.aspect);
		v = (float) ((1 - 2 * (((double) (int) y + 0.5) * (double) p.inv_height)) * (double) p.angle);
	}
	dir = (p.cam_dir + p.cam_right * u) + p.cam_up * v;
}

SKR_DEV void emit_sample(const RenderParams &p, uint32_t out_pix, f3 c)
{ // one sample of one pixel is final: store it (1 spp) or add it to the running sum (AA, sample order = launch order)
	if(p.grid_size > 0)
	{
		float *a = p.acc + (size_t) out_pix * 3;
		if(p.aa_index == 0) { a[0] = c.x; a[1] = c.y; a[2] = c.z; }
		else { a[0] = a[0] + c.x; a[1] = a[1] + c.y; a[2] = a[2] + c.z; }
	}
	else
	{
		if(p.rgbf)
		{
			float *o = p.rgbf + (size_t) out_pix * 3;
			o[0] = c.x; o[1] = c.y; o[2] = c.z;
		}
		if(p.rgb)
		{
			unsigned char *o = p.rgb + (size_t) out_pix * 3;
			o[0] = (unsigned char) quantise(c.x);
			o[1] = (unsigned char) quantise(c.y);
			o[2] = (unsigned char) quantise(c.z);
		}
	}
}

// ---- tables of the level pipelines (render_nodes.hip, render_generic.hip) ----
// Hit records of a level are appended to SKR_P1_REGIONS regions (region = trace wave index mod 64: a region only receives hits of its
// own waves); the level's counter block: [STRIDE r] records in region r, [STRIDE (64 + r)] units handed out, [STRIDE 128] the
// exhausted-regions mask, [STRIDE 129 ..] prefix sums.
SKR_DEV uint32_t *lc_count(uint32_t *ctr, uint32_t region) { return ctr + SKR_PULL_STRIDE * region; }
SKR_DEV uint32_t *lc_taken(uint32_t *ctr, uint32_t region) { return ctr + SKR_PULL_STRIDE * (SKR_P1_REGIONS + region); }
SKR_DEV unsigned long long *lc_dead(uint32_t *ctr) { return reinterpret_cast<unsigned long long *>(ctr + SKR_PULL_STRIDE * (2u * SKR_P1_REGIONS)); }
SKR_DEV uint32_t *lc_prefix(uint32_t *ctr) { return ctr + SKR_PULL_STRIDE * (2u * SKR_P1_REGIONS + 1u); } // 65 words: records before region r; [64] = all

// the scene SoA staged into the workgroup's LDS (one __syncthreads); returns the kernel's view of it
SKR_DEV SceneView stage_scene(const RenderParams &p, float4 *lds4, bool tris)
{
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
	__syncthreads();
	return SceneView{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, tris ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work, p.sph_geom};
}

SKR_DEV void add_counters(const RenderParams &p, const Counters &cn, uint32_t shard, int lane)
{
	if(!p.counters) return;
	const uint32_t a = wave_sum(cn.rays), b = wave_sum(cn.hits), c = wave_sum(cn.shadow_rays), d4 = wave_sum(cn.shadow_tests);
	if(lane == 0)
	{ // sharded: thousands of waves adding to one word serialise
		unsigned long long *c4 = p.counters + 4u * (shard & (SKR_COUNTER_SHARDS - 1u));
		if(a) atomicAdd(&c4[0], (unsigned long long) a);
		if(b) atomicAdd(&c4[1], (unsigned long long) b);
		if(c) atomicAdd(&c4[2], (unsigned long long) c);
		if(d4) atomicAdd(&c4[3], (unsigned long long) d4);
	}
}

struct F3p { float x, y, z; } __attribute__((packed, aligned(4)));
SKR_DEV void store3(float *g, f3 v) { *reinterpret_cast<F3p *>(g) = F3p{v.x, v.y, v.z}; }

// Prefix sums of a level's 64 region counts (the dense numbering of its records), formed by the first wave of a workgroup
// into 65 words of LDS: s_pre[r] = records before region r, s_pre[64] = all.  `publish`: also written behind the level's
// counters, where the kernels launched later read the level's record count (lc_prefix_host).  Ends in a workgroup barrier.
SKR_DEV void region_prefix(const RenderParams &p, uint32_t *s_pre, bool publish)
{
	if(threadIdx.x < 64)
	{
		const int lane = threadIdx.x;
		const uint32_t v = *lc_count(p.rc_ctr, (uint32_t) lane);
		uint32_t incl = v;
#pragma unroll
		for(int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = (uint32_t) __shfl_up((int) incl, off, 64);
			if(lane >= off) incl += o;
		}
		s_pre[lane] = incl - v;
		if(lane == 63) s_pre[64] = incl;
		if(publish)
		{
			uint32_t *pre = lc_prefix(p.rc_ctr);
			pre[lane] = incl - v;
			if(lane == 63) pre[64] = incl;
		}
	}
	__syncthreads();
}

} // namespace
