// Dev aid (not product): which short instruction sequences reproduce IEEE binary32 1/b, sqrt(x) and a/b BIT FOR BIT on gfx950,
// and on which inputs they do not.  The reference's arithmetic needs the correctly rounded results (DESIGN.md "Arithmetic spec");
// the compiler's expansions of `/` and sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt cost 10-16 VALU instructions each.
// Candidates are checked against those expansions on the device itself:
//   1/b and sqrt(x): EXHAUSTIVELY, all 2^32 bit patterns, mismatches histogrammed by the input's biased exponent;
//   a/b: 2^36 pseudo-random pairs (xorshift over raw bit patterns: every exponent combination) + 2^32 pairs with both operands in
//   the product's range — a theorem covers the rest (Markstein: with y = RN(1/b) and q within one ulp of a/b, RN(q + (a - b q) y)
//   is RN(a/b) barring over/underflow); the exhaustive run on y settles its premise.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o tools/ubench/exact_ops tools/ubench/exact_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define DEV static __device__ __forceinline__
DEV float rcp_hw(float b) { return __builtin_amdgcn_rcpf(b); }
DEV float rsq_hw(float x) { return __builtin_amdgcn_rsqf(x); }
DEV float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }

// ---- candidates ----
DEV float rcp_c1(float b)
{ // one Newton step from the 1-ulp hardware reciprocal
	const float y0 = rcp_hw(b);
	const float e = __builtin_fmaf(-b, y0, 1.0f);
	return __builtin_fmaf(e, y0, y0);
}
DEV float rcp_c2(float b)
{ // two steps
	const float y1 = rcp_c1(b);
	const float e = __builtin_fmaf(-b, y1, 1.0f);
	return __builtin_fmaf(e, y1, y1);
}
DEV float sqrt_c1(float x)
{ // rsq, one coupled step
	const float y = rsq_hw(x);
	const float s0 = x * y, h = 0.5f * y;
	const float r = __builtin_fmaf(-s0, s0, x);
	return __builtin_fmaf(r, h, s0);
}
DEV float sqrt_c2(float x)
{ // + a second residual step with the same h
	const float y = rsq_hw(x);
	const float s0 = x * y, h = 0.5f * y;
	const float r = __builtin_fmaf(-s0, s0, x);
	const float s1 = __builtin_fmaf(r, h, s0);
	const float r1 = __builtin_fmaf(-s1, s1, x);
	return __builtin_fmaf(r1, h, s1);
}
DEV float sqrt_c3(float x)
{ // hardware sqrt (1 ulp) + one residual step with h = 0.5 * rsq
	const float s0 = sqrt_hw(x);
	const float h = 0.5f * rsq_hw(x);
	const float r = __builtin_fmaf(-s0, s0, x);
	return __builtin_fmaf(r, h, s0);
}
DEV float div_c1(float a, float b)
{ // y = RN(1/b) (rcp_c1), q0, one correction
	const float y = rcp_c1(b);
	const float q0 = a * y;
	const float r0 = __builtin_fmaf(-b, q0, a);
	return __builtin_fmaf(r0, y, q0);
}
DEV float div_c2(float a, float b)
{ // two corrections: the second is Markstein's final step on a quotient already within one ulp
	const float y = rcp_c1(b);
	const float q0 = a * y;
	const float r0 = __builtin_fmaf(-b, q0, a);
	const float q1 = __builtin_fmaf(r0, y, q0);
	const float r1 = __builtin_fmaf(-b, q1, a);
	return __builtin_fmaf(r1, y, q1);
}

DEV bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

// hist[c * 256 + biased exponent of the input] += mismatches of candidate c
__global__ void unary_kernel(unsigned long long *hist, unsigned long long *example)
{
	const uint64_t n_threads = (uint64_t) gridDim.x * blockDim.x, tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	for(uint64_t u = tid; u < (1ull << 32); u += n_threads)
	{
		const float x = __uint_as_float((uint32_t) u);
		const int e = (int) ((u >> 23) & 0xff);
		const float want_r = 1.0f / x, want_s = __builtin_sqrtf(x);
		const float got[5] = {rcp_c1(x), rcp_c2(x), sqrt_c1(x), sqrt_c2(x), sqrt_c3(x)};
#pragma unroll
		for(int c = 0; c < 5; c++)
		{
			const bool neg_sqrt = c >= 2 && (u >> 31) && (u << 1); // sqrt of a negative number: NaN either way, not of interest
			if(!neg_sqrt && !same(got[c], c < 2 ? want_r : want_s))
			{
				atomicAdd(&hist[c * 256 + e], 1ull);
				example[c * 256 + e] = u;
			}
		}
	}
}

DEV uint64_t xorshift(uint64_t &s)
{
	s ^= s << 13;
	s ^= s >> 7;
	s ^= s << 17;
	return s;
}

// counts[0..1]: mismatches of div_c1 / div_c2 on raw random pairs with |a|, |b| in [2^-60, 2^60] (and a/b then in [2^-120, 2^120]);
// counts[2..3]: the same on pairs from the whole encoding space (the filter of the product excludes what fails here);
// counts[4..5]: pairs tested
__global__ void div_kernel(unsigned long long *counts, unsigned long long *example, int per_thread)
{
	uint64_t s = 0x9E3779B97F4A7C15ull * ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x + 1);
	unsigned long long bad[4] = {0, 0, 0, 0};
	for(int i = 0; i < per_thread; i++)
	{
		const uint64_t r = xorshift(s);
		const uint32_t ua = (uint32_t) r, ub = (uint32_t) (r >> 32);
		{ // in range: force both exponents into [127 - 60, 127 + 60]
			const uint32_t ea = 67u + ((ua >> 23) & 0xffu) % 121u, eb = 67u + ((ub >> 23) & 0xffu) % 121u;
			const float a = __uint_as_float((ua & 0x807fffffu) | (ea << 23)), b = __uint_as_float((ub & 0x807fffffu) | (eb << 23));
			const float want = a / b;
			if(!same(div_c1(a, b), want)) { bad[0]++; example[0] = ((uint64_t) __float_as_uint(a) << 32) | __float_as_uint(b); }
			if(!same(div_c2(a, b), want)) { bad[1]++; example[1] = ((uint64_t) __float_as_uint(a) << 32) | __float_as_uint(b); }
		}
		{
			const float a = __uint_as_float(ua), b = __uint_as_float(ub);
			const float want = a / b;
			if(!same(div_c1(a, b), want)) bad[2]++;
			if(!same(div_c2(a, b), want)) bad[3]++;
		}
	}
	for(int k = 0; k < 4; k++)
		if(bad[k]) atomicAdd(&counts[k], bad[k]);
	if(threadIdx.x == 0) atomicAdd(&counts[4], (unsigned long long) per_thread * blockDim.x);
}

// a/b with the mantissas of both swept over a structured lattice at fixed exponents: all-ones / all-zeros neighbourhoods, where
// the rounding of 1/b and of the quotient is most delicate
__global__ void div_edge_kernel(unsigned long long *counts, unsigned long long *example)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // 2^24 threads: 4096 x 4096 mantissa patterns
	const uint32_t ia = i & 4095u, ib = i >> 12;
	auto mant = [](uint32_t k) -> uint32_t { // 4096 patterns: 1024 lowest, 1024 highest, 2048 spread
		if(k < 1024u) return k;
		if(k < 2048u) return 0x7fffffu - (k - 1024u);
		return ((k - 2048u) * 4099u * 2039u) & 0x7fffffu;
	};
	unsigned long long bad1 = 0, bad2 = 0;
	for(int ea = 120; ea <= 134; ea += 7)
		for(int sb = 0; sb < 2; sb++)
		{
			const float a = __uint_as_float(mant(ia) | ((uint32_t) ea << 23)), b = __uint_as_float(mant(ib) | (127u << 23) | ((uint32_t) sb << 31));
			const float want = a / b;
			if(!same(div_c1(a, b), want)) bad1++;
			if(!same(div_c2(a, b), want)) { bad2++; example[2] = ((uint64_t) __float_as_uint(a) << 32) | __float_as_uint(b); }
		}
	if(bad1) atomicAdd(&counts[6], bad1);
	if(bad2) atomicAdd(&counts[7], bad2);
}

int main()
{
	unsigned long long *d_hist, *d_ex, *d_cnt, *d_cex;
	(void) hipMalloc(&d_hist, 5 * 256 * 8);
	(void) hipMalloc(&d_ex, 5 * 256 * 8);
	(void) hipMalloc(&d_cnt, 8 * 8);
	(void) hipMalloc(&d_cex, 8 * 8);
	(void) hipMemset(d_hist, 0, 5 * 256 * 8);
	(void) hipMemset(d_ex, 0, 5 * 256 * 8);
	(void) hipMemset(d_cnt, 0, 8 * 8);
	(void) hipMemset(d_cex, 0, 8 * 8);
	hipLaunchKernelGGL(unary_kernel, dim3(4096), dim3(256), 0, 0, d_hist, d_ex);
	(void) hipDeviceSynchronize();
	std::vector<unsigned long long> h(5 * 256), ex(5 * 256);
	(void) hipMemcpy(h.data(), d_hist, h.size() * 8, hipMemcpyDeviceToHost);
	(void) hipMemcpy(ex.data(), d_ex, ex.size() * 8, hipMemcpyDeviceToHost);
	const char *names[5] = {"1/b: rcp + 1 Newton step", "1/b: rcp + 2 Newton steps", "sqrt: rsq, 1 coupled step", "sqrt: rsq, 2 steps", "sqrt: v_sqrt + 1 step (h = rsq/2)"};
	for(int c = 0; c < 5; c++)
	{
		unsigned long long tot = 0;
		for(int e = 0; e < 256; e++) tot += h[c * 256 + e];
		printf("%-34s mismatches over all 2^32 inputs: %llu; by biased exponent of the input:", names[c], tot);
		for(int e = 0; e < 256; e++)
			if(h[c * 256 + e]) printf(" %d:%llu(e.g. %08llx)", e, h[c * 256 + e], ex[c * 256 + e]);
		printf("\n");
	}
	const int per_thread = 1 << 14; // 4096 x 256 threads x 2^14 = 2^34 pairs of each kind per launch; 4 launches
	for(int k = 0; k < 4; k++) hipLaunchKernelGGL(div_kernel, dim3(4096 + k), dim3(256), 0, 0, d_cnt, d_cex, per_thread);
	hipLaunchKernelGGL(div_edge_kernel, dim3(65536), dim3(256), 0, 0, d_cnt, d_cex);
	(void) hipDeviceSynchronize();
	unsigned long long cnt[8], cex[8];
	(void) hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost);
	(void) hipMemcpy(cex, d_cex, sizeof(cex), hipMemcpyDeviceToHost);
	printf("a/b, %llu random pairs with exponents in [-60, 60]: 1 correction %llu mismatches (e.g. %016llx), 2 corrections %llu (e.g. %016llx)\n", cnt[4], cnt[0], cex[0], cnt[1], cex[1]);
	printf("a/b, %llu random pairs over all encodings:          1 correction %llu mismatches, 2 corrections %llu\n", cnt[4], cnt[2], cnt[3]);
	printf("a/b, 4096 x 4096 edge mantissas x 3 exponents x 2 signs: 1 correction %llu mismatches, 2 corrections %llu (e.g. %016llx)\n", cnt[6], cnt[7], cex[2]);
	return 0;
}
