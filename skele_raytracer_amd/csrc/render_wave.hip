// Frames without a tree, and the first kernel of the node pipeline.
//
//   skr_direct_kernel    the whole per-pixel loop of the reference (src/main.cpp:129-182) where shade() does not recurse
//                        (src/raytrace.h:208-218: no --gillum, no spheres, or --depth 1): every wave owns an 8x8 pixel tile, all the
//                        --jsample samples of a pixel are traced by its lane (image[y][x] += shade(...) in sample order,
//                        main.cpp:162-165), the tile is packed to u8 in LDS and leaves as whole 24-byte row segments
//   skr_primary_kernel   the primary rays of a --gillum frame: direct light for every pixel; a sphere hit becomes a level-0 node of
//                        the node pipeline (render_nodes.hip), anything else is final
//   skr_resolve_kernel   (AA under --gillum: the samples are separate passes) image /= g*g and the quantiser
//
// Rounds 1-2 kept three more schedules of the --gillum tree in this file (a single megakernel, a parent-queue and a level-queue
// pipeline: DESIGN.md "History"); the node pipeline and the general level pipeline (render_generic.hip) replaced them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wave_common.h"

// One workgroup = 4 independent waves; wave w of block (bx, by) owns the 8x8 pixel tile (2*bx + (w&1), 2*by + (w>>1)).
// Dynamic LDS: scene SoA (shared, staged once) | per wave 192 bytes of packed RGB.  TRIS = false: no triangles in the scene,
// the walk is compiled out; SPH = false: no spheres, the sphere test and all the shading are (a frame that is all mesh is all walk, and
// the walk's scalar registers are what its speed hangs on: with the sphere code compiled in beside it dragon.scn takes 1.27 instead of 1.2 ms).
template <bool TRIS, bool SPH>
__global__ __launch_bounds__(256) void skr_direct_kernel(const RenderParams p)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float4 *lds4 = reinterpret_cast<float4 *>(lds_raw);
	const SceneView sv = stage_scene(p, lds4, TRIS); // the only workgroup barrier
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	unsigned char *s_tile = reinterpret_cast<unsigned char *>(lds4 + 4 * p.n_spheres + 1 + 2 * p.n_lights) + wave * 192;
	const int lx = lane & 7, ly = lane >> 3;
	const int x0 = (blockIdx.x * 2 + (wave & 1)) * 8;
	const uint32_t orow0 = (blockIdx.y * 2 + (wave >> 1)) * 8u;
	const int x = x0 + lx;
	const uint32_t orow = orow0 + ly;
	const uint32_t y = orow < p.out_rows ? image_row(p, orow) : 0xFFFFFFFFu;
	const bool valid = x < p.width && orow < p.out_rows && y < (uint32_t) p.height;
	const uint32_t pixel = y * (uint32_t) p.width + (uint32_t) x;

	Counters cn{0, 0, 0};
	f3 px = mk3(0, 0, 0);
	// main.cpp:140-182: g*g jittered samples (one draw r for both axes, all-float) or one centre sample (u, v formed in double)
	const int nsamp = p.grid_size > 0 ? p.grid_size * p.grid_size : 1;
	for(int s = 0; s < nsamp; s++)
	{
		f3 smp = mk3(0, 0, 0);
		if(valid)
		{
			f3 dir;
			primary_ray(p, x, y, pixel, (uint32_t) s, dir);
			cn.rays++;
			const RayConst r = make_ray(p.cam_pos, dir);
			float tmin = __builtin_inff();
			int sph = -1;
			if(SPH) sph = closest_sphere_from(sv, p.cam_ec, r, tmin);                 // raytrace.h:152-165 (every primary ray starts at the camera)
			if(sv.nt > 0 && any_triangle_closer(sv, r, tmin)) smp = mk3(0, 0, 0);     // :171-186, :221-224
			else if(!SPH || sph < 0) smp = p.background;                              // :189-192
			else
			{
				cn.hits++;
				const f3 P = p.cam_pos + dir * tmin;
				const f3 N = normalize3(P - ld3(sv.geom[sph]));
				smp = direct_light<true>(sv, p, sph, P, N, cn);
				if(p.monte_carlo)
				{ // --gillum at --depth 1: the N children are shade(depth 0) == 0 (:142-145), the combination of :133, :213 stays
					const f3 total = mk3(0, 0, 0) / (float) p.num_path_traces;
					smp = (div3_const(smp, SKR_DIV_PI) + total * 2.0f) * ld3(sv.kd[sph]);
				}
			}
		}
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
		unsigned char *t = s_tile + (ly * 8 + lx) * 3;
		t[0] = (unsigned char) quantise(px.x);
		t[1] = (unsigned char) quantise(px.y);
		t[2] = (unsigned char) quantise(px.z);
		wave_lds_fence();
		const bool full = (x0 + 8 <= p.width) && ((p.width & 3) == 0);
		if(full)
		{
			if(lane < 48)
			{
				const int row = lane / 6, j = lane - row * 6;
				const uint32_t orow2 = orow0 + row;
				if(orow2 < p.out_rows && image_row(p, orow2) < (uint32_t) p.height)
				{
					uint32_t *dst = reinterpret_cast<uint32_t *>(p.rgb + ((size_t) orow2 * p.width + x0) * 3);
					dst[j] = reinterpret_cast<const uint32_t *>(s_tile + row * 24)[j];
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
	add_counters(p, cn, ((uint32_t) blockIdx.y * gridDim.x + blockIdx.x) * 4u + (uint32_t) wave, lane);
}

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
	const SceneView sv{s_geom, s_amb, s_kd, s_ks, s_lights, p.tris, ns, TRIS ? p.n_tris : 0, nl, p.tri_chunks, p.n_tri_chunks, p.tri_chunk_size, p.tri_cones, p.tri_work, p.sph_geom};

	const int wave = tid >> 6, lane = tid & 63;
	const int lx = ((wave & 1) << 3) | (lane & 7), ly = ((wave >> 1) << 3) | (lane >> 3);
	// a band of the node pipeline is a run of 16x16 blocks in row-major order (1-D grid)
	const uint32_t blk = p.band_blk0 + blockIdx.x;
	const uint32_t bx = blk % p.blocks_x, by = blk / p.blocks_x;
	const int x = (int) bx * 16 + lx;
	const uint32_t brow = by * 16 + ly, orow = p.band_row0 + brow;
	const uint32_t y = orow < p.out_rows ? image_row(p, orow) : 0xFFFFFFFFu;
	const bool valid = x < p.width && orow < p.out_rows && y < (uint32_t) p.height;
	const uint32_t pixel = y * (uint32_t) p.width + (uint32_t) x;
	const uint32_t out_pix = orow * (uint32_t) p.width + (uint32_t) x;

	Counters cn{0, 0, 0};
	f3 colour = mk3(0, 0, 0), co = mk3(0, 0, 0), N = mk3(0, 0, 1);
	bool hit = false;
	int sph_hit = 0;
	if(valid)
	{
		f3 dir;
		primary_ray(p, x, y, pixel, p.aa_index, dir);
		cn.rays++;
		const RayConst r = make_ray(p.cam_pos, dir);
		float tmin;
		const int sph = closest_sphere_from(sv, p.cam_ec, r, tmin);
		if(sv.nt > 0 && any_triangle_closer(sv, r, tmin)) colour = mk3(0, 0, 0);
		else if(sph < 0) colour = p.background;
		else
		{
			hit = true;
			sph_hit = sph;
			cn.hits++;
			const f3 P = p.cam_pos + dir * tmin;
			N = normalize3(P - ld3(sv.geom[sph]));
			colour = direct_light<true>(sv, p, sph, P, N, cn);
			co = add_scalar(P, 0.00001f);
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
		// a level-0 node of the node pipeline (render_params.h): a 32-byte geometry row — what tracing its children needs — and a
		// 32-byte shading row — what summing them needs
		float4 *grow = p.nd_dst + (size_t) idx * 2, *srow = p.ns_dst + (size_t) idx * 2;
		grow[0] = make_float4(co.x, co.y, co.z, N.x);
		grow[1] = make_float4(N.y, N.z, __uint_as_float(pixel), __uint_as_float(0u));
		srow[0] = make_float4(colour.x, colour.y, colour.z, __uint_as_float((uint32_t) sph_hit));
		srow[1] = make_float4(0.0f, __uint_as_float(out_pix), __uint_as_float(pixel), __uint_as_float(0u));
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

// Once per renderer: e = cam_pos - C and c = e.e - r^2 of utils.h:115-118 per sphere, for every ray that starts at the camera (closest_sphere_from).
__global__ void skr_camec_kernel(const float4 *geom, int ns, f3 cam_pos, float4 *out)
{
	const int i = (int) (blockIdx.x * blockDim.x + threadIdx.x);
	if(i >= ns) return;
	const float4 g = geom[i];
	const f3 e = cam_pos - ld3(g);
	out[i] = make_float4(e.x, e.y, e.z, dot3(e, e) - g.w);
}
hipError_t skr_launch_camec(const float4 *geom, int ns, f3 cam_pos, float4 *out, hipStream_t stream)
{
	if(ns > 0) hipLaunchKernelGGL(skr_camec_kernel, dim3((unsigned) (ns + 255) / 256), dim3(256), 0, stream, geom, ns, cam_pos, out);
	return hipGetLastError();
}

// AA only: image[y][x] /= g*g (main.cpp:165), then the quantiser (main.cpp:205).
__global__ __launch_bounds__(256) void skr_resolve_kernel(const RenderParams p)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	const size_t n = (size_t) p.width * p.out_rows;
	if(i >= n) return;
	const uint32_t orow = (uint32_t) (i / (size_t) p.width);
	if(image_row(p, orow) >= (uint32_t) p.height) return;
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

// the direct kernel's workgroup: the scene + 4 x 192 bytes of tile — and, for a scene that is all mesh (every lane's time is the triangle
// walk, whose scalar loads go through a 16 KB cache that more waves only thrash), padding up to a third of the CU's LDS: dragon.scn
// runs 1.18 / 1.24 / 1.26 ms at 3 / 4 / 5 waves per SIMD.  A mesh among spheres wants the fourth wave to hide the shading's latencies:
// test.scn (1800 triangles, 4 spheres) 0.556 / 0.485 ms at 3 / 4.
size_t skr_wave_lds_bytes(const RenderParams &p)
{
	const size_t need = ((size_t) 4 * p.n_spheres + 1 + 2 * p.n_lights) * 16 + 4 * 192; // scene | tile bytes
#ifndef SKR_MESH_LDS_PAD
#define SKR_MESH_LDS_PAD 53248 // 3 x 53 248 B are co-resident on a CU (1280-byte granules), 4 are not
#endif
	const size_t third = SKR_MESH_LDS_PAD;
	return (p.n_tris > 0 && p.n_spheres == 0 && need < third) ? third : need;
}

// one launch, no tree: any scene the LDS holds
bool skr_wave_supported(const RenderParams &p) { return p.n_spheres < 65536; }

hipError_t skr_launch_wave(const RenderParams &p, hipStream_t stream)
{
	const dim3 grid((p.width + 15) / 16, (p.out_rows + 15) / 16);
	const size_t lds = skr_wave_lds_bytes(p);
	const bool tris = p.n_tris > 0, sph = p.n_spheres > 0;
	const void *fn = tris ? (sph ? reinterpret_cast<const void *>(skr_direct_kernel<true, true>) : reinterpret_cast<const void *>(skr_direct_kernel<true, false>))
	                      : (sph ? reinterpret_cast<const void *>(skr_direct_kernel<false, true>) : reinterpret_cast<const void *>(skr_direct_kernel<false, false>));
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); // > 64 KiB of dynamic LDS per workgroup has to be opted into
	if(e != hipSuccess) return e;
	if(tris && sph) hipLaunchKernelGGL((skr_direct_kernel<true, true>), grid, dim3(256), lds, stream, p);
	else if(tris) hipLaunchKernelGGL((skr_direct_kernel<true, false>), grid, dim3(256), lds, stream, p);
	else if(sph) hipLaunchKernelGGL((skr_direct_kernel<false, true>), grid, dim3(256), lds, stream, p);
	else hipLaunchKernelGGL((skr_direct_kernel<false, false>), grid, dim3(256), lds, stream, p);
	return hipGetLastError();
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
