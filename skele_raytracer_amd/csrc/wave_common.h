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
	float4 g_next = sv.geom[0];
#pragma unroll 2 // two spheres per trip: the prefetched sphere needs no register-to-register copy (3 of the 24 instructions of a trip)
	for(int i = 0; i < sv.ns; i++)
	{
		const float4 g = g_next;
		g_next = sv.geom[i + 1];
		__builtin_amdgcn_sched_barrier(0); // the prefetch is issued here, a whole trip ahead of its use, not behind the arithmetic
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
#if defined(SKR_DIAG) && SKR_DIAG
			{ // how many candidates a t2 <= 1 pre-test (device_math.h any_decide's reject half) would have kept out
				const f2 mm = (-b) - rp.two_a, m2 = mm * mm * 0x1.00004p+0f;
				const bool k0 = cand0 && ((mm.x <= 0.0f) || (m2.x < D.x)), k1 = cand1 && ((mm.y <= 0.0f) || (m2.y < D.y));
				const unsigned long long c0 = __ballot(cand0), c1 = __ballot(cand1), a0 = __ballot(acc0), a1 = __ballot(acc1), r0 = __ballot(k0), r1 = __ballot(k1);
				DIAG_WAVE(13, __popcll(c0) + __popcll(c1));                 // candidate rays
				DIAG_WAVE(14, __popcll(a0) + __popcll(a1));                 // accepted
				DIAG_WAVE(17, __popcll(r0) + __popcll(r1));                 // certainly t2 <= 1
				DIAG_WAVE(18, ((c0 & ~r0) | (c1 & ~r1)) ? 1 : 0);           // paths still taken with the pre-test
			}
#endif
			best_update(s0, acc0, i, l0, h0, b.x, D.x);
			best_update(s1, acc1, i, l1, h1, b.y, D.y);
		}
	}
	best_resolve(sv, o, d0, rp.four_a.x, s0);
	if(second) best_resolve(sv, o, d1, rp.four_a.y, s1);
}


SKR_DEV void closest_pair(const SceneView &sv, f3 o, f3 d0, f3 d1, bool second, const RayPair &rp, BestState &s0, BestState &s1)
{
	closest_pair_deferred(sv, o, d0, d1, second, rp, s0, s1);
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

} // namespace
