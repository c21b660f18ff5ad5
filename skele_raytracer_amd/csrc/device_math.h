// Device-side arithmetic of the hot path (gfx950 only).
//
// Every function here has to produce the same BITS as the reference's host
// arithmetic (IEEE binary32/64, round-to-nearest-even, glm 0.9.5.4 operation
// order), so the kernels are compiled with -ffp-contract=off and use fma() only
// where the arithmetic spec (DESIGN.md "Arithmetic spec") writes one.  Float
// divide and sqrt are the correctly rounded forms (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt; asserted by tests/test_gpu_units.py).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "render_params.h" // struct f3

#define SKR_DEV static __device__ __forceinline__

// Diagnostic build only (-DSKR_DIAG=1): event counts, 64 shards per counter (tools/diag_counts.py).  Compiles to nothing otherwise.
// (Not together with timing: every event is a global atomic.)
#if defined(SKR_DIAG) && SKR_DIAG
static __device__ unsigned long long skr_diag[32 * 64];
#define DIAG_WAVE(i, v) do { const unsigned long long dv_ = (unsigned long long) (v); const unsigned long long dm_ = __ballot(true); if(__builtin_amdgcn_mbcnt_hi((uint32_t) (dm_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) dm_, 0u)) == 0) atomicAdd(&skr_diag[(i) * 64 + (blockIdx.x & 63)], dv_); } while(0)
#define DIAG_LANES(i) do { const unsigned long long dl_ = __ballot(true); DIAG_WAVE(i, __popcll(dl_)); } while(0)
#else
#define DIAG_WAVE(i, v)
#define DIAG_LANES(i)
#endif


// Correctly rounded binary32 sqrt and divide.  NOT __fsqrt_rn/__fdiv_rn: in this ROCm's
// __clang_hip_math.h __fsqrt_rn is __ocml_native_sqrt_f32 (approximate).  Plain sqrtf and `/`
// are IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt (tests/test_gpu_units.py).
SKR_DEV float sk_divf(float a, float b) { return a / b; }
// The same two results from short sequences wherever those provably return them.  The compiler's expansions above cost 16 (sqrt) and
// 11 (1/b) VALU instructions: denormal scaling, v_div_scale / v_div_fmas / v_div_fixup.  For an operand in [2^-100, 2^101) one coupled
// Newton step on the hardware's 1-ulp estimate is already the correctly rounded value:
//   sqrt(x):  y = v_rsq(x); s = x*y; h = y/2;  s + fma(-s, s, x) * h          1/b:  y = v_rcp(b);  y + fma(-b, y, 1) * y
// checked against the expansions EXHAUSTIVELY — every one of the 2^32 binary32 inputs, on the device (tools/ubench/exact_ops.hip,
// profiles/r03_exact_ops.txt; tests/test_gpu_units.py repeats it on every run): the sequences differ only below 2^-102 (sqrt) and
// for denormal operands / results (1/b: biased exponent 0, 253, 254), so anything outside [2^-100, 2^101) — zero, denormals, huge
// values, infinities, NaN, negative x — takes the expansion.  SKR_FAST_EXACT=0 restores the expansions everywhere (A/B builds).
#ifndef SKR_FAST_EXACT
#define SKR_FAST_EXACT 1
#endif

SKR_DEV bool mid_range_pos(float x) { return __float_as_uint(x) - (27u << 23) < (201u << 23); }                // 2^-100 <= x < 2^101
SKR_DEV bool mid_range_abs(float x) { return (__float_as_uint(x) << 1) - (27u << 24) < (201u << 24); }         // 2^-100 <= |x| < 2^101
SKR_DEV float sk_sqrtf(float x)
{
#if SKR_FAST_EXACT
	if(__builtin_expect(mid_range_pos(x), 1))
	{
		const float y = __builtin_amdgcn_rsqf(x);
		const float s = x * y, h = 0.5f * y;
		return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
	}
#endif
	return __builtin_sqrtf(x);
}
SKR_DEV float sk_rcpf(float b)
{ // 1.0f / b
#if SKR_FAST_EXACT
	if(__builtin_expect(mid_range_abs(b), 1))
	{
		const float y = __builtin_amdgcn_rcpf(b);
		return __builtin_fmaf(__builtin_fmaf(-b, y, 1.0f), y, y);
	}
#endif
	return 1.0f / b;
}

SKR_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
SKR_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
SKR_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
SKR_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
SKR_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
SKR_DEV f3 operator/(f3 a, float s) { return mk3(sk_divf(a.x, s), sk_divf(a.y, s), sk_divf(a.z, s)); } // glm: per-component divide
SKR_DEV f3 add_scalar(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
// glm::dot for vec3: (x + y) + z of the component products (func_geometric.inl:66-73)
SKR_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross operand order (func_geometric.inl:216-226)
SKR_DEV f3 cross3(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
SKR_DEV float sqr3(f3 v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }
SKR_DEV float length3(f3 v) { return sk_sqrtf(sqr3(v)); }
// sqrt(ss), 1.0f / sqrt(ss) and 1.0f / (sqrt(ss) * sqrt(ss)) — a length, glm::normalize's factor and the 1 / d^2 of
// blinn_phong.h:69-70 — behind ONE range test: for ss in [2^-98, 2^99) the length lies in [2^-49, 2^49.5] and its square in
// [2^-98, 2^99.1], all inside the range the short exact forms are verified on.
struct LenTerms { float len, inv, inv2; };
template <bool WANT_INV2>
SKR_DEV LenTerms len_terms(float ss)
{
	LenTerms t;
#if SKR_FAST_EXACT
	if(__builtin_expect(__float_as_uint(ss) - (29u << 23) < (197u << 23), 1))
	{
		const float y = __builtin_amdgcn_rsqf(ss);
		const float s0 = ss * y, h = 0.5f * y;
		t.len = __builtin_fmaf(__builtin_fmaf(-s0, s0, ss), h, s0);
		const float r = __builtin_amdgcn_rcpf(t.len);
		t.inv = __builtin_fmaf(__builtin_fmaf(-t.len, r, 1.0f), r, r);
		t.inv2 = 0.0f;
		if(WANT_INV2)
		{
			const float l2 = t.len * t.len, r2 = __builtin_amdgcn_rcpf(l2);
			t.inv2 = __builtin_fmaf(__builtin_fmaf(-l2, r2, 1.0f), r2, r2);
		}
		return t;
	}
#endif
	t.len = __builtin_sqrtf(ss);
	t.inv = 1.0f / t.len;
	t.inv2 = WANT_INV2 ? 1.0f / (t.len * t.len) : 0.0f;
	return t;
}
// glm::normalize: v * (1.0f / sqrt(sum)) (func_geometric.inl:253-261, func_exponential.inl:226-229)
SKR_DEV f3 normalize3(f3 v) { return v * len_terms<false>(sqr3(v)).inv; }
// std::max(0.0f, x): NaN -> 0
SKR_DEV float max0(float x) { return (0.0f < x) ? x : 0.0f; }
SKR_DEV f3 ld3(const float4 v) { return mk3(v.x, v.y, v.z); }

// ---------------------------------------------- division by a constant ----
// v / d for the two divisors the integrator divides by again and again: pi (raytrace.h:213 `direct_diffuse / float(M_PI)`) and
// pdf = float(1 / pi) (:130 `... / pdf`).  The compiler's correctly rounded `/` costs 10-12 VALU instructions per component; with
// zh = RN(1/d), zl = RN(1/d - zh) the two-instruction form  fma(x, zh, x * zl)  returns the SAME BITS as x / d for every binary32
// x that is zero, infinite, NaN or at least 2^-100 in magnitude — checked EXHAUSTIVELY over all 2^32 x for both divisors, on the
// CPU (tools/constdiv_exhaustive.c, profiles/r03_constdiv_exhaustive.txt) and on the device (tests/test_gpu_units.py); below
// 2^-100 the product x * zl starts to lose bits to underflow, so lanes holding such a component take the division itself.
struct ConstDiv { float d, zh, zl; };
#define SKR_DIV_PI  ConstDiv{0x1.921fb6p+1f, 0x1.45f306p-2f, 0x1.11be6cp-28f}   // d = float(M_PI)
#define SKR_DIV_PDF ConstDiv{0x1.45f306p-2f, 0x1.921fb6p+1f, 0x1.51b7ccp-25f}   // d = float(1 / M_PI)
#ifndef SKR_FAST_CONSTDIV
#define SKR_FAST_CONSTDIV 1 // 0: the plain divisions (A/B builds)
#endif
SKR_DEV uint32_t tiny_key(float x) { return (__float_as_uint(x) << 1) - 1u; } // >= (27 << 24) - 1 iff x == 0 or |x| >= 2^-100 (or inf / NaN)
SKR_DEV f3 div3_const(f3 v, const ConstDiv k)
{
#if SKR_FAST_CONSTDIV
	const uint32_t a = tiny_key(v.x), b = tiny_key(v.y), c = tiny_key(v.z);
	const uint32_t m = (a < b ? a : b) < c ? (a < b ? a : b) : c;
	if(__builtin_expect(m >= (27u << 24) - 1u, 1))
		return mk3(__builtin_fmaf(v.x, k.zh, v.x * k.zl), __builtin_fmaf(v.y, k.zh, v.y * k.zl), __builtin_fmaf(v.z, k.zh, v.z * k.zl));
#endif
	return mk3(sk_divf(v.x, k.d), sk_divf(v.y, k.d), sk_divf(v.z, k.d));
}
SKR_DEV float div_const(float x, const ConstDiv k)
{
#if SKR_FAST_CONSTDIV
	if(__builtin_expect(tiny_key(x) >= (27u << 24) - 1u, 1)) return __builtin_fmaf(x, k.zh, x * k.zl);
#endif
	return sk_divf(x, k.d);
}

// ---------------------------------------------------------------- RNG ----
// Philox4x32-R (Salmon et al., SC'11).  One call yields the (r1, r2) pairs of two sibling GI rays.  The product draws from
// SKR_PHILOX_ROUNDS = 7 rounds, the smallest count that passes BigCrush in the paper (10, this build's choice through round 2, is
// the paper's default with a safety margin): 30 of the 100 VALU instructions of a call, once per leaf round.  The CPU checker of the
// test suite carries the same constant; tests assert Random123's known-answer vectors for 7 and for 10 rounds on both sides.
#ifndef SKR_PHILOX_ROUNDS
#define SKR_PHILOX_ROUNDS 7
#endif
template <int ROUNDS>
SKR_DEV void philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
	for(int r = 0; r < ROUNDS; r++)
	{
		// one widening multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: integer multiplies
		// issue at quarter rate on CDNA, and this function is ~10 % of a leaf round
		const uint64_t p0 = (uint64_t) 0xD2511F53u * c0, p1 = (uint64_t) 0xCD9E8D57u * c2;
		const uint32_t hi0 = (uint32_t) (p0 >> 32), lo0 = (uint32_t) p0;
		const uint32_t hi1 = (uint32_t) (p1 >> 32), lo1 = (uint32_t) p1;
		const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
		c0 = n0;
		c1 = lo1;
		c2 = n2;
		c3 = lo0;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	out[0] = c0;
	out[1] = c1;
	out[2] = c2;
	out[3] = c3;
}
// the counter RNG of the product (DESIGN.md "Counter RNG")
SKR_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
	philox4x32_r<SKR_PHILOX_ROUNDS>(c0, c1, c2, c3, k0, k1, out);
}

// float(k) / float(RAND_MAX) with k = 31 random bits: the map the reference
// applies to rand() (raytrace.h:119-120, main.cpp:146); float(RAND_MAX) == 2^31.
SKR_DEV float u31_to_unit(uint32_t w) { return (float) (w >> 1) * 4.656612873077392578125e-10f; } // * 2^-31 is exact

// ------------------------------------------------------- shared math ----
typedef float f2 __attribute__((ext_vector_type(2)));
SKR_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); } // v_pk_fma_f32: two IEEE fmaf
SKR_DEV f2 splat2(float v) { return f2{v, v}; }

// sin/cos of a binary32 angle in binary32 arithmetic: operation for operation what the CPU checker of the test suite does (round 3; through round 2
// both were evaluated in binary64: 21 binary64 instructions per ray).  k = rint(phi * 2/pi); three-term Cody-Waite reduction (one
// fma each); minimax polynomials in z = y*y (Horner, fma); quadrant by a select and a sign flip each.  Every step is one IEEE
// binary32 operation, so host and device agree bit for bit; max error 1.43 ulp over every float of [0, 2 pi] (exhaustive,
// tools/sincos_exhaustive.c).  Written for TWO angles at once — the sibling rays of a pair — in packed arithmetic; each component
// is exactly what the one-angle form computes.
SKR_DEV void sincos_spec2(f2 phi, f2 &s, f2 &c)
{
	f2 kf = phi * 0x1.45f306p-1f;
	kf = f2{__builtin_rintf(kf.x), __builtin_rintf(kf.y)};
	const f2 nk = -kf;
	f2 y = fma2(nk, splat2(0x1.921fb6p+0f), phi);
	y = fma2(nk, splat2(-0x1.777a5cp-25f), y);
	y = fma2(nk, splat2(-0x1.ee59dap-50f), y);
	const f2 z = y * y, yz = y * z;
	f2 ps = splat2(0x1.66997ap-19f);
	ps = fma2(ps, z, splat2(-0x1.9ff9bcp-13f));
	ps = fma2(ps, z, splat2(0x1.1110f8p-7f));
	ps = fma2(ps, z, splat2(-0x1.555556p-3f));
	const f2 sy = fma2(yz, ps, y);
	f2 pc = splat2(0x1.9a52ccp-16f);
	pc = fma2(pc, z, splat2(-0x1.6c0db0p-10f));
	pc = fma2(pc, z, splat2(0x1.55554cp-5f));
	pc = fma2(pc, z, splat2(-0.5f));
	const f2 cy = fma2(z, pc, splat2(1.0f));
	// quadrant k mod 4: (s, c) = (sy, cy), (cy, -sy), (-sy, -cy), (-cy, sy)
	const uint32_t q0 = (uint32_t) (int) kf.x, q1 = (uint32_t) (int) kf.y;
	const bool sw0 = q0 & 1u, sw1 = q1 & 1u;
	s = f2{__uint_as_float(__float_as_uint(sw0 ? cy.x : sy.x) ^ ((q0 & 2u) << 30)), __uint_as_float(__float_as_uint(sw1 ? cy.y : sy.y) ^ ((q1 & 2u) << 30))};
	c = f2{__uint_as_float(__float_as_uint(sw0 ? sy.x : cy.x) ^ (((q0 + 1u) & 2u) << 30)), __uint_as_float(__float_as_uint(sw1 ? sy.y : cy.y) ^ (((q1 + 1u) & 2u) << 30))};
}

SKR_DEV void sincos_spec(float phi, float &s, float &c)
{
	f2 s2, c2;
	sincos_spec2(splat2(phi), s2, c2);
	s = s2.x;
	c = c2.x;
}

// General (non-integer exponent) branch of powf_spec: 2^(p log2 x) in binary64.  Cold for every
// shipped scene, so it is kept out of line: inlined, its two dozen binary64 constants and
// temporaries raise the register pressure of every shading path.
static __device__ __attribute__((noinline)) float powf_general(float x, float p)
{
	const uint64_t b = (uint64_t) __double_as_longlong((double) x);
	int e = (int) ((b >> 52) & 0x7ff) - 1023;
	double m = __longlong_as_double((long long) ((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
	if(m > 0x1.6A09E667F3BCDp+0)
	{
		m *= 0.5;
		e += 1;
	}
	const double s = (m - 1.0) / (m + 1.0);
	const double s2 = s * s;
	double q = 1.0 / 21.0;
	q = fma(q, s2, 1.0 / 19.0);
	q = fma(q, s2, 1.0 / 17.0);
	q = fma(q, s2, 1.0 / 15.0);
	q = fma(q, s2, 1.0 / 13.0);
	q = fma(q, s2, 1.0 / 11.0);
	q = fma(q, s2, 1.0 / 9.0);
	q = fma(q, s2, 1.0 / 7.0);
	q = fma(q, s2, 1.0 / 5.0);
	q = fma(q, s2, 1.0 / 3.0);
	q = fma(q, s2, 1.0);
	const double ln_m = 2.0 * s * q;
	const double log2x = fma(ln_m, 0x1.71547652B82FEp+0, (double) e);
	const double y = (double) p * log2x;
	if(y >= 128.0) return __builtin_inff();
	if(y < -150.0) return 0.0f;
	const double n = rint(y);
	const double t = (y - n) * 0x1.62E42FEFA39EFp-1;
	double r = 1.0 / 6227020800.0;
	r = fma(r, t, 1.0 / 479001600.0);
	r = fma(r, t, 1.0 / 39916800.0);
	r = fma(r, t, 1.0 / 3628800.0);
	r = fma(r, t, 1.0 / 362880.0);
	r = fma(r, t, 1.0 / 40320.0);
	r = fma(r, t, 1.0 / 5040.0);
	r = fma(r, t, 1.0 / 720.0);
	r = fma(r, t, 1.0 / 120.0);
	r = fma(r, t, 1.0 / 24.0);
	r = fma(r, t, 1.0 / 6.0);
	r = fma(r, t, 0.5);
	r = fma(r, t, 1.0);
	r = fma(r, t, 1.0);
	const double scale = __longlong_as_double((long long) ((uint64_t) ((int) n + 1023) << 52));
	return (float) (r * scale);
}

// powf(x, p), x >= 0, in binary64, rounded once: square-and-multiply for integer p in [1,1024],
// 2^(p log2 x) otherwise.
static __device__ __attribute__((noinline)) float powf_spec_cold(float x, float p)
{ // every case the straight-line form below does not take: p == 0, NaN operands, negative x, a non-integer or huge exponent
	if(p == 0.0f) return 1.0f;
	if(x != x || p != p) return x + p;
	if(x == 0.0f) return (p > 0.0f) ? 0.0f : __builtin_inff();
	if(x == 1.0f) return 1.0f;
	if(x == __builtin_inff()) return (p > 0.0f) ? __builtin_inff() : 0.0f;
	if(x < 0.0f) return __builtin_nanf("");
	if(p >= 1.0f && p <= 1024.0f && p == __builtin_rintf(p))
	{ // integer phong exponents (every shipped scene): square-and-multiply in binary64, low bit first
		unsigned n = (unsigned) p;
		double r = 1.0, base = (double) x;
		while(n)
		{
			if(n & 1u) r *= base;
			n >>= 1;
			if(n) base *= base;
		}
		return (float) r;
	}
	return powf_general(x, p);
}

// The common case — an integer exponent in [1, 1024] and 0 <= x <= inf — without a branch: the loop of the spec above runs as
// many trips as the lane's exponent has bits, an s_cbranch each (~100 cycles of a wave's time apiece on this part, twice per
// shaded hit); here 7 or 11 trips (by the bit length of the scene's largest integer exponent, RenderParams::pow_steps: wave-uniform)
// are unrolled with a select where the spec has `if(n & 1)`.  The same products in the same order: squarings past a lane's top
// bit are formed and never used; x = 0, 1, inf come out of the multiplications as the spec's early returns state them.
#ifndef SKR_FLAT_POW
#define SKR_FLAT_POW 1 // 0: the loop (A/B builds)
#endif
template <int STEPS>
SKR_DEV float pow_int_flat(float x, unsigned n)
{
	double r = 1.0, base = (double) x;
#pragma unroll
	for(int k = 0; k < STEPS; k++)
	{
		const double rb = r * base;
		r = ((n >> k) & 1u) ? rb : r;
		if(k + 1 < STEPS) base *= base;
	}
	return (float) r;
}
SKR_DEV float powf_spec(float x, float p, int steps)
{
#if SKR_FLAT_POW
	const bool plain = (p >= 1.0f) && (p <= 1024.0f) && (p == __builtin_rintf(p)) && (x >= 0.0f);
	if(__builtin_expect(!plain, 0)) return powf_spec_cold(x, p);
	// (`steps` is wave-uniform: one scalar branch picks the unrolled length — exponents below 128 in every shipped scene)
	return steps <= 7 ? pow_int_flat<7>(x, (unsigned) p) : pow_int_flat<11>(x, (unsigned) p);
#else
	return powf_spec_cold(x, p);
#endif
}

// ---------------------------------------------------------- geometry ----
// utils.h:87-110 smallest_root given the float discriminant D >= 0.
// -b and 2a are floats promoted to double; sqrt and the divide are binary64
// (the unqualified sqrt() binds to ::sqrt(double)); one
// rounding to float.  Only the near root can be returned.
SKR_DEV float near_root_exact(float two_a, float b, float D)
{
	const double q = ((double) (-b) - sqrt((double) D)) / (double) two_a;
	const float t2 = (float) q;
	return (t2 >= 0.0f) ? t2 : __builtin_inff();
}

// utils.h:87 + :113 for one sphere.  two_a = 2*a, four_a = 4*a are per-ray
// invariants (exact doublings of a = dot(d,d)).
SKR_DEV float sphere_distance(f3 o, f3 d, float two_a, float four_a, float4 sph)
{
	const f3 e = o - ld3(sph);
	const float b = 2 * dot3(d, e);
	const float c = dot3(e, e) - sph.w; // sph.w = radius*radius
	const float D = b * b - four_a * c;
	if(D < 0) return __builtin_inff();
	return near_root_exact(two_a, b, D);
}

// utils.h:169-179
SKR_DEV bool accept_distance(float t) { return !(t <= 1.0f || t == __builtin_inff()); }

// ------------------------------------------------ filtered predicates ----
// The reference decides "1 < t < inf" and orders candidates by
//   t2 = fl32( fl64( fl64(-b - fl64(sqrt(D))) / (2a) ) )          (utils.h:87-110)
// which costs a binary64 sqrt and divide per candidate sphere.  The kernels first
// bracket t2 with binary32 arithmetic: with s = v_sqrt_f32(D), num = fl32(-b - s),
// ta = fl32(num * rcp(2a)) (v_sqrt/v_rcp are 1-ulp instructions) every rounding is a
// relative perturbation <= 2^-22 of s, num or the quotient, so
//   |ta - t2| <= E = (s + |num|) * rcp(2a) * 2^-20 + |ta| * 2^-22
// with a factor >= 4 to spare (DESIGN.md "Filtered predicates" has the derivation).
// A decision is taken from [ta-E, ta+E] only when the whole interval is on one side;
// otherwise (and whenever a NaN/inf/denormal shows up: every comparison below is
// written so that NaN falls through) the exact binary64 form is evaluated.  Results
// are therefore identical to evaluating the exact form everywhere.
struct RayFilt {
	float two_a, four_a; // exact per-ray invariants (2a, 4a)
	float inv2a;         // ~ 1/(2a)
	float k_err;         // inv2a * 2^-20
	bool sane;           // a in a range where the relative bounds hold
};

SKR_DEV RayFilt make_filt(f3 d)
{
	const float a = dot3(d, d);
	RayFilt f;
	f.two_a = 2 * a;
	f.four_a = 4 * a;
	f.inv2a = __builtin_amdgcn_rcpf(f.two_a);
	f.k_err = f.inv2a * 0x1p-20f;
	f.sane = (a > 1e-18f) && (a < 1e18f);
	return f;
}

// One sphere against one ray.  Returns false for a certain miss.  Otherwise
// [lo, hi] brackets t2 (lo == hi when it had to be resolved exactly) and b, D
// are the spec's float coefficients (kept for the exact evaluation of the winner).
// Rays that share an origin (the N children of one node, raytrace.h:128; the shadow rays of one
// hit point, utils.h:45) share e = o - C and c = e.e - r^2 of utils.h:115-118: formed once per sphere.
SKR_DEV bool bracket_from_ec(f3 e, float c, f3 d, const RayFilt &f, float &lo, float &hi, float &b, float &D)
{
	b = 2 * dot3(d, e);
	D = b * b - f.four_a * c;
	// b >= 0 => -b - sqrt(D) <= 0 => t2 <= 0 (or NaN): never accepted.  D < 0 / NaN: miss.
	if(!(D >= 0.0f) || !(b < 0.0f)) return false;
	const float s = __builtin_amdgcn_sqrtf(D);
	const float num = (-b) - s;
	const float ta = num * f.inv2a;
	const float E = __builtin_fmaf(__builtin_fabsf(ta), 0x1p-22f, (s + __builtin_fabsf(num)) * f.k_err);
	lo = ta - E;
	hi = ta + E;
	const bool normal = f.sane && (D > 1e-30f) && (b < -1e-15f);
	const bool certain_accept = normal && (lo > 1.0f) && (hi < 3.0e38f);
	const bool certain_reject = normal && (hi < 1.0f);
	if(certain_reject) return false;
	if(!certain_accept)
	{ // too close to call in binary32: the exact form decides
		const float t = near_root_exact(f.two_a, b, D);
		if(!accept_distance(t)) return false;
		lo = hi = t;
	}
	return true;
}

// ---- two rays with a common origin, evaluated with packed binary32 arithmetic ----
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 do two IEEE binary32 operations per lane per
// instruction with the same rounding as the scalar forms, so every component below is the
// same value sphere_bracket() would produce for that ray alone.
struct RayPair {
	f2 dx, dy, dz;              // directions of ray 0 / ray 1, component-wise
	f2 two_a, four_a, inv2a, k_err;
	bool sane0, sane1;
};

SKR_DEV RayPair make_pair(f3 d0, f3 d1)
{
	RayPair r;
	r.dx = f2{d0.x, d1.x};
	r.dy = f2{d0.y, d1.y};
	r.dz = f2{d0.z, d1.z};
	const f2 a = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz; // dot3(d, d)
	r.two_a = 2.0f * a;
	r.four_a = 4.0f * a;
	r.inv2a = f2{__builtin_amdgcn_rcpf(r.two_a.x), __builtin_amdgcn_rcpf(r.two_a.y)};
	r.k_err = r.inv2a * 0x1p-20f;
	r.sane0 = (a.x > 1e-18f) && (a.x < 1e18f);
	r.sane1 = (a.y > 1e-18f) && (a.y < 1e18f);
	return r;
}

// b and D of utils.h:116-118 / :89 for both rays against one sphere given the shared e, c.
SKR_DEV void pair_bD(const RayPair &r, f3 e, float c, f2 &b, f2 &D)
{
	b = 2.0f * ((r.dx * e.x + r.dy * e.y) + r.dz * e.z);
	D = b * b - r.four_a * c;
}

// Brackets [lo, hi] of t2 for both rays (valid where D >= 0 and b < 0).
SKR_DEV void pair_bracket(const RayPair &r, f2 b, f2 D, f2 &lo, f2 &hi)
{
	const f2 s = f2{__builtin_amdgcn_sqrtf(D.x), __builtin_amdgcn_sqrtf(D.y)};
	const f2 num = (-b) - s;
	const f2 ta = num * r.inv2a;
	const f2 abs_num = __builtin_elementwise_max(num, -num);
	const f2 abs_ta = __builtin_elementwise_max(ta, -ta);
	const f2 E = abs_ta * 0x1p-22f + (s + abs_num) * r.k_err; // bound only: its own rounding is covered by the 4x slack
	lo = ta - E;
	hi = ta + E;
}

// Classify one component: returns accept; *resolved_t is set (and lo = hi = t) when the exact form had to decide.
SKR_DEV bool bracket_decide(bool sane, float two_a, float b, float D, float &lo, float &hi)
{
	const bool normal = sane && (D > 1e-30f) && (b < -1e-15f);
	const bool certain_accept = normal && (lo > 1.0f) && (hi < 3.0e38f);
	const bool certain_reject = normal && (hi < 1.0f);
	if(certain_reject) return false;
	if(!certain_accept)
	{
		DIAG_WAVE(7, 1);
		DIAG_LANES(3);
		const float t = near_root_exact(two_a, b, D);
		if(!accept_distance(t)) return false;
		lo = hi = t;
	}
	return true;
}

// The any-hit form (shadow rays, utils.h:42-58) needs no bracket, only "is t2 > 1": with m = fl(-b - 2a) — one rounding
// of an exact difference, so m = (-b - 2a)(1 + th), |th| <= 2^-24, and its sign is exact —
//   t2 = 1 + (-b - 2a - sqrt(D)) / (2a)   (in reals, on the float values b, D, 2a)
// is > 1 + 2^-23 (so the spec's rounded t2 is > 1) when m > a/2 and m^2 (1 - 2^-18) > D, and is <= 1 (so the spec's
// t2 is too: rounding is monotone) when m <= 0 or m^2 (1 + 2^-18) < D; the margins leave 2^-20 relative between m
// and sqrt(D), far more than the three binary32 roundings of the test and the binary64 roundings of the spec take.
// Anything else — and every non-finite or tiny operand — is decided by the exact form.  No sqrt, no reciprocal.
struct PairAny {
	f2 two_a, quarter; // 2a and a/2 of the two rays
	bool sane0, sane1;
};
SKR_DEV void pair_any_m(const PairAny &pa, f2 b, f2 &m, f2 &acc_lhs, f2 &rej_lhs)
{
	m = (-b) - pa.two_a;
	const f2 m2 = m * m;
	acc_lhs = m2 * 0x1.ffff8p-1f; // 1 - 2^-18
	rej_lhs = m2 * 0x1.00004p+0f; // 1 + 2^-18
}
SKR_DEV bool any_decide(bool sane, float two_a, float quarter, float b, float D, float m, float acc_lhs, float rej_lhs)
{
	const bool normal = sane && (D > 1e-30f) && (b < -1e-15f) && (rej_lhs < 3.0e38f);
	if(normal && ((m <= 0.0f) || (rej_lhs < D))) return false;
	if(normal && (m > quarter) && (acc_lhs > D)) return true;
	DIAG_WAVE(7, 1);
	return accept_distance(near_root_exact(two_a, b, D));
}

SKR_DEV bool sphere_bracket(f3 o, f3 d, const RayFilt &f, float4 sph, float &lo, float &hi, float &b, float &D)
{
	const f3 e = o - ld3(sph);
	const float c = dot3(e, e) - sph.w;
	return bracket_from_ec(e, c, d, f, lo, hi, b, D);
}

// utils.h:181-213 with the edges e1 = v1-v0, e2 = v2-v0 precomputed on the host
// (same subtractions).  u carries the reference's flipped sign; no t>0 test.
SKR_DEV bool triangle_hit(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float &t)
{
	const f3 p = cross3(d, e2);
	const float det = dot3(e1, p);
	if(fabsf(det) < 0.00001f) return false;
	const float inv = sk_divf(1.0f, det); // (the compiler's expansion: in this divergent inner loop the short form's range branch cost the dragon walk 20 %)
	const f3 tv = o - v0;
	const float u = inv * dot3(mk3(-tv.x, -tv.y, -tv.z), p);
	if(u < 0 || u > 1) return false;
	const f3 q = cross3(tv, e1);
	const float v = dot3(d, q) * inv;
	if(v < 0 || u + v > 1) return false;
	t = dot3(e2, q) * inv;
	return true;
}

// utils.h:148-165 transform_coordinate_space
SKR_DEV void tangent_basis(f3 n, f3 &nt, f3 &nb)
{
	if(fabsf(n.x) > fabsf(n.y)) nt = mk3(n.z, 0.0f, -n.x) / sk_sqrtf(n.x * n.x + n.z * n.z);
	else nt = mk3(0.0f, -n.z, n.y) / sk_sqrtf(n.y * n.y + n.z * n.z);
	nb = cross3(n, nt);
}

// main.cpp:205: (unsigned char)(std::min(float(1), c) * 255); NaN -> 255
SKR_DEV uint32_t quantise(float c)
{
	const float m = (c < 1.0f) ? c : 1.0f;
	const float s = m * 255;
	if(!(s > -2147483648.0f)) return 0;
	return (uint32_t) (int32_t) s & 0xffu;
}
