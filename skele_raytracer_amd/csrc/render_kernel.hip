// The hot path as one HIP megakernel per launch of row tiles (gfx950 / CDNA4).
//
// Replaces the per-pixel loop body reference src/main.cpp:129-182 (== :36-85)
// and the whole shade() tree under it (src/raytrace.h:139-227, blinn_phong.h,
// utils.h).  Written from the behaviour, not translated: SoA scene staged in
// LDS, 64-lane waves own 8x8 pixel tiles, wave-uniform primitive loops with
// ballot early-outs, counter-based RNG, u8 packing and row-coalesced stores on
// device.  DESIGN.md describes the layout and the arithmetic spec.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "shade_common.h"

namespace {

// shade() (raytrace.h:139-227) with the recursion depth as a template
// parameter: LEVELS == the `depth` argument the reference would carry here.
template <int LEVELS>
SKR_DEV f3 shade(const SceneView &sv, const RenderParams &p, f3 o, f3 d, uint32_t node, uint32_t pixel, uint32_t aa, Counters &cn)
{
	if constexpr(LEVELS <= 0) return mk3(0, 0, 0);
	else
	{
		cn.rays++;
		const RayConst r = make_ray(o, d);
		float tmin;
		const int sph = closest_sphere(sv, r, tmin);
		if(sv.nt > 0 && any_triangle_closer(sv, r, tmin)) return mk3(0, 0, 0);
		if(sph < 0) return p.background;
		cn.hits++;
		// raytrace.h:197-205: t recomputed for the winner == tmin
		const f3 P = o + d * tmin;
		const f3 N = normalize3(P - ld3(sv.geom[sph]));
		const f3 direct = direct_light(sv, p, sph, P, N, cn);
		if(!p.monte_carlo) return direct;

		f3 total = mk3(0, 0, 0);
		if constexpr(LEVELS > 1)
		{
			f3 nt, nb;
			tangent_basis(N, nt, nb);
			const float pdf = (float) (1 / 3.14159265358979323846);
			const f3 co = add_scalar(P, 0.00001f);
			uint32_t rnd[4];
			for(int i = 0; i < p.num_path_traces; i++)
			{
				if((i & 1) == 0) philox4x32(pixel, aa, node, (uint32_t) i >> 1, p.seed_lo, p.seed_hi, rnd);
				const float r1 = u31_to_unit(rnd[2 * (i & 1)]), r2 = u31_to_unit(rnd[2 * (i & 1) + 1]);
				const f3 w = gi_direction(r1, r2, N, nt, nb);
				const f3 child = shade<LEVELS - 1>(sv, p, co, w, node * (uint32_t) p.num_path_traces + (uint32_t) i + 1u, pixel, aa, cn);
				total = total + (child * r1) / pdf;
			}
		}
		// LEVELS == 1: every child is shade(depth 0) == (0,0,0); the sum stays (0,0,0)
		total = total / (float) p.num_path_traces;
		return (direct / (float) 3.14159265358979323846 + total * 2.0f) * ld3(sv.kd[sph]);
	}
}

// ---- --shade-triangles (the rules: include/skr.h skr_options.shade_triangles) ----
// HEAD turns every accepted triangle into a black sample (raytrace.h:221-224).  In this mode a triangle is a surface:
// among the triangles utils.h:181-213 accepts with 0 < t (the triangle the ray starts on excepted) the one with the
// smallest t wins if that t is strictly below the closest sphere's; equal t: the lower index in the file.  It is then
// shaded exactly as a sphere is (blinn_phong.h, raytrace.h:107-136,208-218) with the material in force on its
// `triangle` line and the geometric normal normalize(cross(v1 - v0, v2 - v0)), turned against the ray.

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

// The walk of shade_common.h tree_walk() without its early-outs: every chunk whose conservative sphere this lane's
// line touches is tested to the end (the spheres bound the accept test itself, whatever t comes out).
template <bool CONES>
SKR_DEV void tree_walk_closest(const SceneView &sv, const RayConst &r, int from_tri, TriBest &b)
{
	const float dd = r.two_a * 0.5f; // dot(d, d)
	int i = 0;
	const float4 *chunk_ent = sv.chunks + 3 * (sv.nchunks + 1);
	float4 A = sv.chunks[0], B = sv.chunks[1], lk = sv.chunks[2];
	while(i < sv.nchunks)
	{
		const int i_out = __float_as_int(lk.x);
		const float4 A_in = sv.chunks[3 * i + 3], B_in = sv.chunks[3 * i + 4], lk_in = sv.chunks[3 * i + 5];
		const float4 A_out = sv.chunks[3 * i_out], B_out = sv.chunks[3 * i_out + 1], lk_out = sv.chunks[3 * i_out + 2];
		const bool enter = __any(line_touches<CONES>(r, dd, A, B));
		const int count = __float_as_int(lk.z);
		if(enter && count > 0)
		{
			const int c0 = __float_as_int(lk.y), c1 = c0 + count;
			for(int c = c0; c < c1; c++)
			{
				const bool mine = line_touches<CONES>(r, dd, chunk_ent[2 * c], chunk_ent[2 * c + 1]);
				if(__any(mine))
				{
					const int i0 = c * sv.chunk, i1 = (i0 + sv.chunk < sv.nt) ? i0 + sv.chunk : sv.nt;
					for(int k = i0; k < i1; k++) tri_consider(r, mine, ld3(sv.tris[3 * k]), sv.tris[3 * k + 1], sv.tris[3 * k + 2], k, from_tri, b);
				}
			}
		}
		i = enter ? i + 1 : i_out;
		A = enter ? A_in : A_out;
		B = enter ? B_in : B_out;
		lk = enter ? lk_in : lk_out;
	}
}

SKR_DEV void closest_triangle(const SceneView &sv, const RayConst &r, int from_tri, TriBest &b)
{
	if(sv.nchunks > 0)
	{
		if(sv.cones) tree_walk_closest<true>(sv, r, from_tri, b);
		else tree_walk_closest<false>(sv, r, from_tri, b);
		return;
	}
	for(int k = 0; k < sv.nt; k++) tri_consider(r, true, ld3(sv.tris[3 * k]), sv.tris[3 * k + 1], sv.tris[3 * k + 2], k, from_tri, b);
}

// shade() with shaded triangles; from_tri = file index of the triangle this ray starts on (-1: none)
template <int LEVELS>
SKR_DEV f3 shade_surfaces(const SceneView &sv, const RenderParams &p, f3 o, f3 d, uint32_t node, uint32_t pixel, uint32_t aa, int from_tri, Counters &cn)
{
	if constexpr(LEVELS <= 0) return mk3(0, 0, 0);
	else
	{
		cn.rays++;
		const RayConst r = make_ray(o, d);
		float tmin;
		const int sph = closest_sphere(sv, r, tmin);
		TriBest b{tmin, -1, -1};
		closest_triangle(sv, r, from_tri, b);
		if(sph < 0 && b.slot < 0) return p.background;
		cn.hits++;
		const f3 P = o + d * b.t; // (== tmin for a sphere)
		f3 N, kd, ks;
		float4 ambp;
		if(b.slot >= 0)
		{
			N = normalize3(cross3(ld3(sv.tris[3 * b.slot + 1]), ld3(sv.tris[3 * b.slot + 2])));
			if(dot3(N, d) > 0.0f) N = mk3(-N.x, -N.y, -N.z);
			ambp = p.tri_mats[3 * b.slot];
			kd = ld3(p.tri_mats[3 * b.slot + 1]);
			ks = ld3(p.tri_mats[3 * b.slot + 2]);
		}
		else
		{
			N = normalize3(P - ld3(sv.geom[sph]));
			ambp = sv.amb[sph];
			kd = ld3(sv.kd[sph]);
			ks = ld3(sv.ks[sph]);
		}
		const f3 direct = direct_light_of(sv, p, kd, ks, ambp, P, N, cn);
		if(!p.monte_carlo) return direct;

		f3 total = mk3(0, 0, 0);
		if constexpr(LEVELS > 1)
		{
			f3 nt, nb;
			tangent_basis(N, nt, nb);
			const float pdf = (float) (1 / 3.14159265358979323846);
			const f3 co = add_scalar(P, 0.00001f);
			uint32_t rnd[4];
			for(int i = 0; i < p.num_path_traces; i++)
			{
				if((i & 1) == 0) philox4x32(pixel, aa, node, (uint32_t) i >> 1, p.seed_lo, p.seed_hi, rnd);
				const float r1 = u31_to_unit(rnd[2 * (i & 1)]), r2 = u31_to_unit(rnd[2 * (i & 1) + 1]);
				const f3 w = gi_direction(r1, r2, N, nt, nb);
				const f3 child = shade_surfaces<LEVELS - 1>(sv, p, co, w, node * (uint32_t) p.num_path_traces + (uint32_t) i + 1u, pixel, aa, b.file, cn);
				total = total + (child * r1) / pdf;
			}
		}
		total = total / (float) p.num_path_traces;
		return (direct / (float) 3.14159265358979323846 + total * 2.0f) * kd;
	}
}

// ---- --legacy-reflect (the rules: include/skr.h skr_options.legacy_reflect) ----
// The code behind the early return of raytrace.h:44: Fresnel term, one refraction and one reflection ray per light from the hit
// point itself, each shade(depth - 1), added to the direct term.  Unreachable at HEAD.  Out of line on purpose: three call
// sites per level.

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

template <int LEVELS>
static __device__ __attribute__((noinline)) f3 shade_legacy(const SceneView &sv, const RenderParams &p, f3 o, f3 d, uint32_t node, uint32_t pixel, uint32_t aa, Counters &cn)
{
	if constexpr(LEVELS <= 0) return mk3(0, 0, 0);
	else
	{
		cn.rays++;
		const RayConst r = make_ray(o, d);
		float tmin;
		const int sph = closest_sphere(sv, r, tmin);
		if(sv.nt > 0 && any_triangle_closer(sv, r, tmin)) return mk3(0, 0, 0);
		if(sph < 0) return p.background;
		cn.hits++;
		const f3 P = o + d * tmin;
		const f3 N = normalize3(P - ld3(sv.geom[sph]));
		f3 direct = direct_light(sv, p, sph, P, N, cn);
		const uint32_t A = (uint32_t) p.num_path_traces + 2u * (uint32_t) sv.nl; // children per node: the --gillum rays, then two per light
		{ // raytrace.h:45-103
			const float4 ks4 = sv.ks[sph];
			const f3 ks = ld3(ks4);
			const float mat_ior = ks4.w;
			const float fr = legacy_fresnel(d, N, mat_ior);
			f3 refraction_colour = mk3(0, 0, 0), reflection_colour = mk3(0, 0, 0);
			if(ks.x != 0.0f || ks.y != 0.0f || ks.z != 0.0f)
			{ // (depth > 0 holds here)
				const uint32_t base = node * A + (uint32_t) p.num_path_traces + 1u;
				for(int i = 0; i < sv.nl; i++)
				{
					const f3 L = light_term(sv, i, P).L; // glm::normalize(position - P) / normalize(direction)
					if(fr < 1)
					{ // blinn_phong.h:143-153
						const float dn = dot3(d, N);
						const float k = 1.0f - (mat_ior * mat_ior) * (1.0f - dn * dn);
						const f3 rd = (k < 0.0f) ? mk3(0, 0, 0) : (d * mat_ior - N * (mat_ior * dn + sk_sqrtf(k)));
						refraction_colour = shade_legacy<LEVELS - 1>(sv, p, P, rd, base + 2u * (uint32_t) i, pixel, aa, cn) * fr; // (=, not +=)
					}
					const f3 md = normalize3(L - N * (2.0f * dot3(L, N))); // blinn_phong.h:137-140: the LIGHT direction mirrored
					const f3 c = shade_legacy<LEVELS - 1>(sv, p, P, md, base + 2u * (uint32_t) i + 1u, pixel, aa, cn);
					reflection_colour = reflection_colour + (ks * (1 - fr)) * c;
				}
			}
			direct = (direct + refraction_colour) + reflection_colour; // :102
		}
		if(!p.monte_carlo) return direct;

		f3 total = mk3(0, 0, 0);
		if constexpr(LEVELS > 1)
		{
			f3 nt, nb;
			tangent_basis(N, nt, nb);
			const float pdf = (float) (1 / 3.14159265358979323846);
			const f3 co = add_scalar(P, 0.00001f);
			uint32_t rnd[4];
			for(int i = 0; i < p.num_path_traces; i++)
			{
				if((i & 1) == 0) philox4x32(pixel, aa, node, (uint32_t) i >> 1, p.seed_lo, p.seed_hi, rnd);
				const float r1 = u31_to_unit(rnd[2 * (i & 1)]), r2 = u31_to_unit(rnd[2 * (i & 1) + 1]);
				const f3 w = gi_direction(r1, r2, N, nt, nb);
				const f3 child = shade_legacy<LEVELS - 1>(sv, p, co, w, node * A + (uint32_t) i + 1u, pixel, aa, cn);
				total = total + (child * r1) / pdf;
			}
		}
		total = total / (float) p.num_path_traces;
		return (direct / (float) 3.14159265358979323846 + total * 2.0f) * ld3(sv.kd[sph]);
	}
}

} // namespace

// One workgroup = 4 waves = a 16x16 pixel tile; each wave owns an 8x8 sub-tile.
// Dynamic LDS: scene SoA | 16 rows x 48 bytes of packed RGB for the tile.
template <int DEPTH, int MODE> // MODE 0: HEAD; 1: --shade-triangles; 2: --legacy-reflect
__global__ __launch_bounds__(256) void skr_render_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const int ns = p.n_spheres, nl = p.n_lights;
	float4 *s_geom = lds4, *s_amb = lds4 + ns + 1, *s_kd = s_amb + ns, *s_ks = s_kd + ns, *s_lights = s_ks + ns;
	unsigned char *s_tile = reinterpret_cast<unsigned char *>(lds4 + 4 * ns + 1 + 2 * nl);

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

	const SceneView sv{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, p.n_tris, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work};

	const int wave = tid >> 6, lane = tid & 63;
	const int lx = ((wave & 1) << 3) | (lane & 7), ly = ((wave >> 1) << 3) | (lane >> 3);
	const int x = blockIdx.x * 16 + lx;
	const uint32_t orow = blockIdx.y * 16 + ly;          // row in the compact output
	const uint32_t k = orow / p.tile_rows;               // which of this launch's tiles
	const uint32_t y = (p.first_tile + k * p.tile_stride) * p.tile_rows + (orow - k * p.tile_rows);
	const bool valid = x < p.width && orow < p.out_rows && y < (uint32_t) p.height;

	Counters cn{0, 0, 0};
	f3 px = mk3(0, 0, 0);
	if(valid)
	{
		const uint32_t pixel = y * (uint32_t) p.width + (uint32_t) x;
		if(p.grid_size > 0)
		{ // main.cpp:140-166: g*g samples, one draw r for both axes, all-float
			const int ns2 = p.grid_size * p.grid_size;
			for(int s = 0; s < ns2; s++)
			{
				uint32_t rnd[4];
				philox4x32(pixel, (uint32_t) s, 0u, 0xFFFFFFFFu, p.seed_lo, p.seed_hi, rnd);
				const float r = u31_to_unit(rnd[0]);
				const float u = ((2 * (((float) x + r) * p.inv_width) - 1) * p.angle) * p.aspect;
				const float v = (1 - 2 * (((float) (int) y + r) * p.inv_height)) * p.angle;
				const f3 dir = (p.cam_dir + p.cam_right * u) + p.cam_up * v;
				if constexpr(MODE == 1) px = px + shade_surfaces<DEPTH>(sv, p, p.cam_pos, dir, 0u, pixel, (uint32_t) s, -1, cn);
				else if constexpr(MODE == 2) px = px + shade_legacy<DEPTH>(sv, p, p.cam_pos, dir, 0u, pixel, (uint32_t) s, cn);
				else px = px + shade<DEPTH>(sv, p, p.cam_pos, dir, 0u, pixel, (uint32_t) s, cn);
			}
			px = px / (float) ns2;
		}
		else
		{ // main.cpp:168-182: pixel centre, u and v formed in double then narrowed
			const float u = (float) (((2 * (((double) x + 0.5) * (double) p.inv_width) - 1) * (double) p.angle) * (double) p.aspect);
			const float v = (float) ((1 - 2 * (((double) (int) y + 0.5) * (double) p.inv_height)) * (double) p.angle);
			const f3 dir = (p.cam_dir + p.cam_right * u) + p.cam_up * v;
			if constexpr(MODE == 1) px = shade_surfaces<DEPTH>(sv, p, p.cam_pos, dir, 0u, pixel, 0u, -1, cn);
			else if constexpr(MODE == 2) px = shade_legacy<DEPTH>(sv, p, p.cam_pos, dir, 0u, pixel, 0u, cn);
			else px = shade<DEPTH>(sv, p, p.cam_pos, dir, 0u, pixel, 0u, cn);
		}
		if(p.rgbf)
		{
			float *o = p.rgbf + ((size_t) orow * p.width + x) * 3;
			o[0] = px.x;
			o[1] = px.y;
			o[2] = px.z;
		}
	}

	// pack to u8 in LDS, then store whole 48-byte row segments as dwords
	unsigned char *t = s_tile + (ly * 16 + lx) * 3;
	t[0] = (unsigned char) quantise(px.x);
	t[1] = (unsigned char) quantise(px.y);
	t[2] = (unsigned char) quantise(px.z);
	__syncthreads();
	if(p.rgb)
	{
		const int x0 = blockIdx.x * 16;
		const bool full = (x0 + 16 <= p.width) && ((p.width & 3) == 0);
		if(full)
		{
			if(tid < 192)
			{
				const int row = tid / 12, j = tid - row * 12;
				const uint32_t orow2 = blockIdx.y * 16 + row;
				const uint32_t k2 = orow2 / p.tile_rows;
				const uint32_t y2 = (p.first_tile + k2 * p.tile_stride) * p.tile_rows + (orow2 - k2 * p.tile_rows);
				if(orow2 < p.out_rows && y2 < (uint32_t) p.height)
				{
					uint32_t *dst = reinterpret_cast<uint32_t *>(p.rgb + ((size_t) orow2 * p.width + x0) * 3);
					dst[j] = reinterpret_cast<const uint32_t *>(s_tile + row * 48)[j];
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

// ---------------------------------------------------------------- launch ----

// render_wave.hip
size_t skr_wave_lds_bytes(const RenderParams &p);
bool skr_wave_supported(const RenderParams &p);
hipError_t skr_launch_wave(const RenderParams &p, hipStream_t stream);
bool skr_queue_selected(const RenderParams &p);
bool skr_levels_selected(const RenderParams &p);
hipError_t skr_launch_levels(const RenderParams &p, hipStream_t stream, const SkrTimingHook *hook);
hipError_t skr_launch_queue(const RenderParams &p, hipStream_t stream, const SkrTimingHook *hook);
// render_nodes.hip
bool skr_nodes_selected(const RenderParams &p);
bool skr_nodes_flat(const RenderParams &p);
size_t skr_nodes_lds_bytes(const RenderParams &p);
hipError_t skr_launch_nodes(const RenderParams &p, hipStream_t stream, const SkrTimingHook *hook);

// The wave-streaming kernel is the product path wherever it applies (depth <= 3, gillum <= 256);
// the per-pixel kernel covers the rest (depth 4..6).  SKR_KERNEL=v1 forces the latter (A/B runs).
static bool use_wave_kernel(const RenderParams &p)
{
	if(p.sw.kernel_v1) return false;
	return skr_wave_supported(p);
}

size_t skr_render_lds_bytes(const RenderParams &p)
{
	if(skr_nodes_selected(p)) return skr_nodes_lds_bytes(p);
	if(use_wave_kernel(p)) return skr_wave_lds_bytes(p);
	return ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 16 * 48;
}

template <int D>
static hipError_t launch_depth(const RenderParams &p, dim3 grid, size_t lds, hipStream_t stream)
{
	if(p.legacy_reflect) hipLaunchKernelGGL((skr_render_kernel<D, 2>), grid, dim3(256), lds, stream, p);
	else if(p.shade_triangles) hipLaunchKernelGGL((skr_render_kernel<D, 1>), grid, dim3(256), lds, stream, p);
	else hipLaunchKernelGGL((skr_render_kernel<D, 0>), grid, dim3(256), lds, stream, p);
	return hipGetLastError();
}

hipError_t skr_launch_render(const RenderParams &p, hipStream_t stream, const char **variant, const SkrTimingHook *hook)
{
	if(skr_nodes_selected(p) && p.node_scratch)
	{
		*variant = skr_nodes_flat(p) ? "node_levels_v5_flat" : "node_levels_v5";
		return skr_launch_nodes(p, stream, hook);
	}
	if(use_wave_kernel(p))
	{
		if(skr_levels_selected(p) && p.parents && p.qctr && p.p1 && p.slot1)
		{
			*variant = "level_queues_v4";
			return skr_launch_levels(p, stream, hook);
		}
		if(skr_queue_selected(p) && p.parents && p.qctr)
		{
			*variant = "parent_queue_v3";
			return skr_launch_queue(p, stream, hook);
		}
		*variant = "wave_streaming_v2";
		skr_hook_start(hook, stream);
		const hipError_t e = skr_launch_wave(p, stream);
		skr_hook_stop(hook, stream);
		return e;
	}
	const dim3 grid((p.width + 15) / 16, (p.out_rows + 15) / 16);
	const size_t lds = skr_render_lds_bytes(p);
	*variant = p.legacy_reflect ? "lane_per_pixel_legacy_v1r" : p.shade_triangles ? "lane_per_pixel_surfaces_v1s" : "lane_per_pixel_dfs_v1f";
	skr_hook_start(hook, stream);
	hipError_t e = hipErrorInvalidValue;
	switch(p.max_depth)
	{
		case 1: e = launch_depth<1>(p, grid, lds, stream); break;
		case 2: e = launch_depth<2>(p, grid, lds, stream); break;
		case 3: e = launch_depth<3>(p, grid, lds, stream); break;
		case 4: e = launch_depth<4>(p, grid, lds, stream); break;
		case 5: e = launch_depth<5>(p, grid, lds, stream); break;
		case 6: e = launch_depth<6>(p, grid, lds, stream); break;
		default: break;
	}
	skr_hook_stop(hook, stream);
	return e;
}

// ------------------------------------------------------------ debug eval ----
// Device-side evaluation of the arithmetic spec, one record per thread
// (skr_debug_eval in include/skr.h).
__global__ void skr_debug_kernel(int op, const uint32_t *in, uint32_t *out, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	auto F = [](uint32_t u) { return __uint_as_float(u); };
	auto U = [](float f) { return __float_as_uint(f); };
	switch(op)
	{
		case 0: {
			uint32_t o[4];
			const uint32_t *c = in + 6 * i;
			philox4x32(c[0], c[1], c[2], c[3], c[4], c[5], o);
			for(int k = 0; k < 4; k++) out[4 * i + k] = o[k];
			break;
		}
		case 9: { // EXHAUSTIVE check of the short exact forms (device_math.h) against the compiler's correctly rounded expansions: record i
		          // covers the 65536 bit patterns in[i] << 16 ...; out = mismatches of {sk_sqrtf, sk_rcpf, div_const pi, div_const pdf}
			uint32_t bad[4] = {0, 0, 0, 0};
			auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); };
			for(uint32_t k = 0; k < 65536u; k++)
			{
				const float x = F((in[i] << 16) | k);
				bad[0] += !same(sk_sqrtf(x), __builtin_sqrtf(x));
				bad[1] += !same(sk_rcpf(x), 1.0f / x);
				bad[2] += !same(div_const(x, SKR_DIV_PI), x / (float) 3.14159265358979323846);
				bad[3] += !same(div_const(x, SKR_DIV_PDF), x / (float) (1 / 3.14159265358979323846));
			}
			for(int k = 0; k < 4; k++) out[4 * i + k] = bad[k];
			break;
		}
		case 8: { // the round function at Random123's default count (known-answer vectors exist for 7 and for 10 rounds)
			uint32_t o[4];
			const uint32_t *c = in + 6 * i;
			philox4x32_r<10>(c[0], c[1], c[2], c[3], c[4], c[5], o);
			for(int k = 0; k < 4; k++) out[4 * i + k] = o[k];
			break;
		}
		case 1: {
			float s, c;
			sincos_spec(F(in[i]), s, c);
			out[2 * i] = U(s);
			out[2 * i + 1] = U(c);
			break;
		}
		case 2: out[i] = U(powf_spec(F(in[2 * i]), F(in[2 * i + 1]), 11)); break;
		case 3: {
			const float a = F(in[3 * i]), b = F(in[3 * i + 1]), c = F(in[3 * i + 2]);
			const float D = b * b - (4 * a) * c;
			out[i] = U((D < 0) ? __builtin_inff() : near_root_exact(2 * a, b, D));
			break;
		}
		case 4: {
			const uint32_t *r = in + 15 * i;
			const f3 o = mk3(F(r[0]), F(r[1]), F(r[2])), d = mk3(F(r[3]), F(r[4]), F(r[5]));
			const f3 v0 = mk3(F(r[6]), F(r[7]), F(r[8])), v1 = mk3(F(r[9]), F(r[10]), F(r[11])), v2 = mk3(F(r[12]), F(r[13]), F(r[14]));
			float t = 0.0f;
			const bool h = triangle_hit(o, d, v0, v1 - v0, v2 - v0, t);
			out[2 * i] = h ? 1u : 0u;
			out[2 * i + 1] = h ? U(t) : 0u;
			break;
		}
		case 5: out[i] = quantise(F(in[i])); break;
		case 6: {
			f3 nt, nb;
			tangent_basis(mk3(F(in[3 * i]), F(in[3 * i + 1]), F(in[3 * i + 2])), nt, nb);
			out[6 * i] = U(nt.x); out[6 * i + 1] = U(nt.y); out[6 * i + 2] = U(nt.z);
			out[6 * i + 3] = U(nb.x); out[6 * i + 4] = U(nb.y); out[6 * i + 5] = U(nb.z);
			break;
		}
		case 7: { // binary32 sqrt and divide must be the correctly rounded forms
			out[2 * i] = U(sk_sqrtf(F(in[2 * i])));
			out[2 * i + 1] = U(sk_divf(F(in[2 * i]), F(in[2 * i + 1])));
			break;
		}
		default: break;
	}
}

hipError_t skr_launch_debug(int op, const void *d_in, void *d_out, uint32_t n, hipStream_t stream)
{
	hipLaunchKernelGGL(skr_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, (const uint32_t *) d_in, (uint32_t *) d_out, n);
	return hipGetLastError();
}
