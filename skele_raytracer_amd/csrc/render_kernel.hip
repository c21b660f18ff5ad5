// Which kernels render a launch (gfx950 / CDNA4), and the device-side evaluation of the arithmetic spec for the unit tests.
//
// The per-pixel loop of the reference (src/main.cpp:129-182) and the shade() tree under it (src/raytrace.h:139-227) run as
//   * skr_direct_kernel (render_wave.hip) where shade() does not recurse: no --gillum, no spheres, --depth 1;
//   * the node pipeline (render_nodes.hip: sibling-pair kernels, persistent leaf kernel) for --gillum trees over sphere scenes
//     (and scenes with a handful of triangles) — the headline path;
//   * the general level pipeline (render_generic.hip: one lane per ray) for meshes under --gillum, --shade-triangles,
//     --legacy-reflect and more than 256 children per node.
// Rounds 1-2 also had a lane-per-pixel kernel with the recursion depth as a template parameter (--depth <= 6) here; the level
// pipelines take any depth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "shade_common.h"

// render_wave.hip
size_t skr_wave_lds_bytes(const RenderParams &p);
bool skr_wave_supported(const RenderParams &p);
hipError_t skr_launch_wave(const RenderParams &p, hipStream_t stream);
// render_nodes.hip
bool skr_nodes_selected(const RenderParams &p);
bool skr_nodes_flat(const RenderParams &p);
size_t skr_nodes_lds_bytes(const RenderParams &p);
hipError_t skr_launch_nodes(const RenderParams &p, hipStream_t stream, const SkrTimingHook *hook);
// render_generic.hip
size_t skr_generic_lds_bytes(const RenderParams &p);
hipError_t skr_launch_generic(const RenderParams &p, hipStream_t stream, const SkrTimingHook *hook);

// does shade() recurse at all in this launch (api.cpp folds --depth to 1 where it cannot: raytrace.h:208-218)
static bool has_tree(const RenderParams &p) { return p.max_depth > 1; }

// The general level pipeline takes every tree the pair kernels of the node pipeline do not, and the two modes only it knows.
// SKR_PIPELINE=generic forces it for every launch, SKR_PIPELINE=nodes keeps triangle meshes on the node pipeline (tests, A/B runs).
bool skr_generic_selected(const RenderParams &p)
{
	if(p.sw.pipeline == SKR_PIPE_GENERIC) return true;
	if(p.shade_triangles || p.legacy_reflect) return true;
	return has_tree(p) && !skr_nodes_selected(p);
}

size_t skr_render_lds_bytes(const RenderParams &p)
{
	if(skr_generic_selected(p)) return skr_generic_lds_bytes(p);
	if(skr_nodes_selected(p)) return skr_nodes_lds_bytes(p);
	return skr_wave_lds_bytes(p);
}

hipError_t skr_launch_render(const RenderParams &p, hipStream_t stream, const char **variant, const SkrTimingHook *hook)
{
	if(skr_generic_selected(p))
	{
		if(!p.node_scratch) return hipErrorInvalidValue;
		*variant = "level_pipeline_g1";
		return skr_launch_generic(p, stream, hook);
	}
	if(skr_nodes_selected(p))
	{
		if(!p.node_scratch) return hipErrorInvalidValue;
		*variant = skr_nodes_flat(p) ? "node_levels_v5_flat" : "node_levels_v5";
		return skr_launch_nodes(p, stream, hook);
	}
	if(!skr_wave_supported(p)) return hipErrorInvalidValue;
	*variant = "direct_v3";
	skr_hook_start(hook, stream);
	const hipError_t e = skr_launch_wave(p, stream);
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
