/* oracle/skr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11 + OpenMP) of the reference's hot path
 *   per-pixel loop  src/main.cpp:129-182 (== :36-85)
 *   shade() tree    src/raytrace.h:139-227 and everything it calls
 *   .scn loader     src/scene.cpp:12-227
 *   PPM quantiser   src/main.cpp:199-211
 * written from the reference's behaviour, each function citing the lines it
 * follows.  All citations are relative to /root/reference/.
 *
 * PINNING.  In mode (SKO_RNG_GLIBC_REPLAY, SKO_MATH_LIBM) this file must give
 * byte-identical PPMs (and bit-identical float images) to oracle/_ref/ref_render,
 * which is the reference's own shade()/parseScene() compiled unmodified; the
 * committed fixtures tests/golden/ref_*.ppm were produced by that binary
 * (tools/make_golden.py) and tests/test_oracle_golden.py checks them on every
 * run, together with the reference's own pixel-exact fixture renders/testcpu.ppm.
 *
 * The GPU cannot replay glibc rand() in DFS order, so the product defines a
 * second, order-independent semantic that differs from the above ONLY in where
 * random numbers and three libm calls come from:
 *   SKO_RNG_COUNTER  r = float(k)/2^31 with k = 31 bits of Philox4x32-7 keyed by
 *                    (seed; pixel, aa sample, parent node id, child pair)
 *                    instead of float(rand())/float(RAND_MAX)
 *   SKO_MATH_SHARED  sinf/cosf(2*pi*r2) evaluated by the binary32 recipe and powf(x, phong)
 *                    by the binary64 recipe below (identical operation
 *                    sequences exist in the HIP kernel); powf(x,2) == x*x.
 * That mode is what the HIP kernel is compared with, bit for bit.
 *
 * Arithmetic rules (must hold for the HIP twin as well): IEEE binary32/64,
 * round-to-nearest-even, no FMA contraction (-ffp-contract=off), explicit fma()
 * only where written, glm 0.9.5.4 operation order:
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z      glm/detail/func_geometric.inl:66-73
 *   cross(x,y)    = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)   :216-226
 *   length(v)     = sqrtf((v.x*v.x + v.y*v.y) + v.z*v.z)                         :108-114
 *   normalize(v)  = v * (1.0f / sqrtf(dot-like sum))                              :253-261, func_exponential.inl:226-229
 *   v / s         = per-component division                 glm/detail/type_vec3.inl:580-590
 */
#define _GNU_SOURCE
#include "skr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vec3 */

typedef sko_vec3 v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vadds(v3 a, float s) { return V(a.x + s, a.y + s, a.z + s); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 x, v3 y) { return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
static inline float vsqr(v3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
static inline float vlength(v3 v) { return sqrtf(vsqr(v)); }
static inline v3 vnormalize(v3 v) { return vscale(v, 1.0f / sqrtf(vsqr(v))); }
/* std::max(0.0f, x) == (0.0f < x) ? x : 0.0f ; NaN -> 0 */
static inline float max0(float x) { return (0.0f < x) ? x : 0.0f; }

/* ------------------------------------------------------------ Philox4x32-R */
/* Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3"
 * (SC'11).  The counter RNG of the product draws from Philox4x32-7 (SKO_PHILOX_ROUNDS): seven rounds is the
 * smallest count that passes BigCrush in the paper ("Philox4x32-7 ... Crush-resistant"; ten, the paper's default and
 * this build's choice through round 2, adds a safety margin); Random123 ships it as philox4x32_R(7, ...).
 * This is the product's own choice (the reference draws rand(), raytrace.h:119-120): the statistical pin against the
 * reference's frames is tests/test_statistics.py.  Known-answer vectors of Random123 for 7 AND 10 rounds are asserted
 * in tests/test_spec_units.py and, on the device, tests/test_gpu_units.py. */
void sko_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
	uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
	uint32_t k0 = key[0], k1 = key[1];
	for(int r = 0; r < rounds; r++)
	{
		uint64_t p0 = (uint64_t) 0xD2511F53u * c0;
		uint64_t p1 = (uint64_t) 0xCD9E8D57u * c2;
		uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0;
		uint32_t n1 = (uint32_t) p1;
		uint32_t n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
		uint32_t n3 = (uint32_t) p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void sko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { sko_philox4x32_r(ctr, key, 10, out); }
void sko_philox4x32_spec(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { sko_philox4x32_r(ctr, key, SKO_PHILOX_ROUNDS, out); }

/* Counter layout (DESIGN.md "RNG"):
 *   key = (seed lo, seed hi)
 *   GI draws of child c of tree node `parent`:  ctr = (pixel, aa, parent, c>>1);
 *        r1 = u31(out[2*(c&1)]), r2 = u31(out[2*(c&1)+1])
 *   jitter draw of AA sample `aa`:              ctr = (pixel, aa, 0, 0xFFFFFFFF); r = u31(out[0])
 * u31(w) = float(w >> 1) / 2147483648.0f, the same map the reference applies to
 * rand() (float(rand())/float(RAND_MAX), raytrace.h:119-120, main.cpp:146).
 * Node ids: primary sample = 0, child c of node n = n*N + c + 1. */
static inline float u31(uint32_t w) { return (float) (w >> 1) / 2147483648.0f; }

void sko_counter_draws(uint64_t seed, uint32_t pixel, uint32_t aa, uint32_t parent_node, uint32_t child, float *r1, float *r2)
{
	uint32_t ctr[4] = {pixel, aa, parent_node, child >> 1}, key[2] = {(uint32_t) seed, (uint32_t) (seed >> 32)}, o[4];
	sko_philox4x32_spec(ctr, key, o);
	*r1 = u31(o[2 * (child & 1)]);
	*r2 = u31(o[2 * (child & 1) + 1]);
}

float sko_counter_jitter(uint64_t seed, uint32_t pixel, uint32_t aa)
{
	uint32_t ctr[4] = {pixel, aa, 0, 0xFFFFFFFFu}, key[2] = {(uint32_t) seed, (uint32_t) (seed >> 32)}, o[4];
	sko_philox4x32_spec(ctr, key, o);
	return u31(o[0]);
}

/* ------------------------------------------------------ shared-math spec */

static inline double as_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t as_u64(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* sin/cos of a binary32 angle in binary32 arithmetic (round 3; rounds 1-2 evaluated them in binary64 and rounded once:
 * 21 binary64 instructions per ray on the device for digits nothing downstream can see).  The reference calls libm's
 * cosf/sinf(phi) (raytrace.h:26-27); this is the product's replacement, identical operation for operation in
 * csrc/device_math.h sincos_spec (packed there for two sibling rays):
 *   k = rint(phi * 2/pi);  y = ((phi - k*C1) - k*C2) - k*C3  (three-term Cody-Waite, one fma each; the first is exact
 *   for |phi| <= 8);  odd / even minimax polynomials in z = y*y (Horner, fma);  quadrant select on k & 3.
 * Every step is a single IEEE binary32 operation (fmaf = one rounding), so x86 and gfx950 produce the same bits.
 * Accuracy, measured EXHAUSTIVELY over the 1 086 918 620 floats of [0, 2 pi] against correctly rounded values
 * (tools/sincos_exhaustive.c, profiles/r03_sincos_exhaustive.txt): max error 1.43 ulp (sin), 1.43 ulp (cos); 99.996 % of
 * the results within 1 ulp.  phi = 2 pi r2 lies in [0, 2 pi]; beyond |phi| ~ 1e4 the reduction loses accuracy (never
 * reached: r2 is a draw from [0, 1]). */
void sko_sincos_shared(float phi, float *s, float *c)
{
	const float TWO_OVER_PI = 0x1.45f306p-1f;
	const float PIO2_1 = 0x1.921fb6p+0f;    /* pi/2 rounded to binary32 */
	const float PIO2_2 = -0x1.777a5cp-25f;  /* pi/2 - PIO2_1, rounded */
	const float PIO2_3 = -0x1.ee59dap-50f;  /* pi/2 - PIO2_1 - PIO2_2, rounded */
	const float kf = rintf(phi * TWO_OVER_PI);
	const int k = (int) kf;
	float y = fmaf(-kf, PIO2_1, phi);
	y = fmaf(-kf, PIO2_2, y);
	y = fmaf(-kf, PIO2_3, y);
	const float z = y * y, yz = y * z;
	/* sin(y) = y + y z (S1 + z (S2 + z (S3 + z S4))) on |y| <= pi/4 */
	float ps = 0x1.66997ap-19f;
	ps = fmaf(ps, z, -0x1.9ff9bcp-13f);
	ps = fmaf(ps, z, 0x1.1110f8p-7f);
	ps = fmaf(ps, z, -0x1.555556p-3f);
	const float sy = fmaf(yz, ps, y);
	/* cos(y) = 1 + z (-1/2 + z (C2 + z (C3 + z C4))) */
	float pc = 0x1.9a52ccp-16f;
	pc = fmaf(pc, z, -0x1.6c0db0p-10f);
	pc = fmaf(pc, z, 0x1.55554cp-5f);
	pc = fmaf(pc, z, -0.5f);
	const float cy = fmaf(z, pc, 1.0f);
	float sv, cv;
	switch(k & 3)
	{
		case 0: sv = sy; cv = cy; break;
		case 1: sv = cy; cv = -sy; break;
		case 2: sv = -sy; cv = -cy; break;
		default: sv = -cy; cv = sy; break;
	}
	*s = sv;
	*c = cv;
}

/* powf(x, p) for x >= 0 (x is max(0, N.H), blinn_phong.h:117), evaluated in
 * binary64 and rounded once: by square-and-multiply for integer p in [1,1024],
 * as 2^(p*log2 x) otherwise. */
float sko_powf_shared(float x, float p)
{
	if(p == 0.0f) return 1.0f;
	if(x != x || p != p) return x + p;
	if(x == 0.0f) return (p > 0.0f) ? 0.0f : INFINITY;
	if(x == 1.0f) return 1.0f;
	if(x == INFINITY) return (p > 0.0f) ? INFINITY : 0.0f;
	if(x < 0.0f) return NAN;
	if(p >= 1.0f && p <= 1024.0f && p == rintf(p))
	{ /* integer phong exponents (every shipped scene): square-and-multiply in binary64, low bit first */
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
	uint64_t b = as_u64((double) x);
	int e = (int) ((b >> 52) & 0x7ff) - 1023;
	double m = as_double((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
	if(m > 0x1.6A09E667F3BCDp+0) /* sqrt(2) */
	{
		m *= 0.5;
		e += 1;
	}
	/* ln m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716 */
	double s = (m - 1.0) / (m + 1.0);
	double s2 = s * s;
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
	double ln_m = 2.0 * s * q;
	double log2x = fma(ln_m, 0x1.71547652B82FEp+0 /* 1/ln 2 */, (double) e);
	double y = (double) p * log2x;
	if(y >= 128.0) return INFINITY;
	if(y < -150.0) return 0.0f;
	double n = rint(y);
	double t = (y - n) * 0x1.62E42FEFA39EFp-1; /* ln 2 */
	double r = 1.0 / 6227020800.0;             /* 1/13! */
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
	double scale = as_double((uint64_t) ((int) n + 1023) << 52);
	return (float) (r * scale);
}

/* ------------------------------------------------------------ geometry */

/* utils.h:87-110 smallest_root.  The unqualified sqrt() there binds to
 * ::sqrt(double): -b and 2*a are float, promoted; one
 * rounding to float at the assignment.  With a >= 0 the first branch
 * (t1 < t2 && t1 >= 0) can never be taken (sqrt >= 0 => t1 >= t2, and NaN
 * compares false), so only t2 decides — both roots are still formed here
 * exactly as written, the kernel forms t2 only. */
float sko_smallest_root(float a, float b, float c)
{
	float discriminant = b * b - 4 * a * c;
	if(discriminant < 0) return INFINITY;
	float t1 = (float) (((double) (-b) + sqrt((double) discriminant)) / (double) (2 * a));
	float t2 = (float) (((double) (-b) - sqrt((double) discriminant)) / (double) (2 * a));
	if(t1 < t2 && t1 >= 0) return t1;
	else if(t2 >= 0) return t2;
	return INFINITY;
}

/* utils.h:113-121 collision_distance */
static inline float collision_distance(v3 o, v3 d, const sko_sphere *sp)
{
	v3 e_c = vsub(o, sp->center);
	float a = vdot(d, d);
	float b = 2 * vdot(d, e_c);
	float c = vdot(e_c, e_c) - sp->radius * sp->radius;
	return sko_smallest_root(a, b, c);
}

/* utils.h:169-179 intersection_occurs: accept iff 1 < t < inf */
static inline int intersection_occurs(float distance) { return !(distance <= 1.0f || distance == INFINITY); }

/* utils.h:181-213 triangle_intersection_occurs (sign of u flipped vs the
 * textbook, no t>0 test; fabs() on a float is exact in either overload) */
static inline int triangle_test(v3 o, v3 d, const sko_triangle *tr, float *t)
{
	v3 v0v1 = vsub(tr->v1, tr->v0);
	v3 v0v2 = vsub(tr->v2, tr->v0);
	v3 p = vcross(d, v0v2);
	float det = vdot(v0v1, p);
	if(fabsf(det) < 0.00001f) return 0;
	float inverse = 1.0f / det;
	v3 tv = vsub(o, tr->v0);
	float u = inverse * vdot(V(-tv.x, -tv.y, -tv.z), p);
	if(u < 0 || u > 1) return 0;
	v3 q = vcross(tv, v0v1);
	float v = vdot(d, q) * inverse;
	if(v < 0 || u + v > 1) return 0;
	*t = vdot(v0v2, q) * inverse;
	return 1;
}

int sko_triangle_test(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3], float *t)
{
	sko_triangle tr = {V(v0[0], v0[1], v0[2]), V(v1[0], v1[1], v1[2]), V(v2[0], v2[1], v2[2])};
	return triangle_test(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), &tr, t);
}

/* utils.h:148-165 transform_coordinate_space */
static inline void basis(v3 n, v3 *nt, v3 *nb)
{
	if(fabsf(n.x) > fabsf(n.y)) *nt = vdivs(V(n.z, 0, -n.x), sqrtf(n.x * n.x + n.z * n.z));
	else *nt = vdivs(V(0, -n.z, n.y), sqrtf(n.y * n.y + n.z * n.z));
	*nb = vcross(n, *nt);
}

void sko_basis(const float n[3], float nt[3], float nb[3])
{
	v3 a, b;
	basis(V(n[0], n[1], n[2]), &a, &b);
	nt[0] = a.x; nt[1] = a.y; nt[2] = a.z;
	nb[0] = b.x; nb[1] = b.y; nb[2] = b.z;
}

/* ------------------------------------------------------------ integrator */

typedef struct {
	const sko_scene *sc;
	const sko_options *op;
	uint32_t pixel, aa;
	uint64_t n_rays, n_hits, n_shadow, n_sph_tests, n_tri_tests;
} ctx_t;

/* utils.h:42-58 shadow(Scene, P, PointLight): origin P + 1e-6 (scalar added to
 * every component), dir normalize(Lp - P); true if ANY sphere has 1 < t < inf
 * (spheres beyond the light occlude too). */
static int shadowed(ctx_t *cx, v3 P, v3 L)
{
	v3 o = vadds(P, 0.000001f);
	const sko_scene *sc = cx->sc;
	cx->n_shadow++;
	for(int i = 0; i < sc->n_spheres; i++)
	{
		cx->n_sph_tests++;
		if(intersection_occurs(collision_distance(o, L, &sc->spheres[i]))) return 1;
	}
	return 0;
}

static inline float powf_mode(const ctx_t *cx, float x, float p) { return cx->op->math_mode == SKO_MATH_LIBM ? powf(x, p) : sko_powf_shared(x, p); }
/* powf(x, 2.0f) == x*x bit for bit on the domains that occur (SURVEY.md §8c,
 * exhaustive); libm mode still calls powf so that mode stays the literal reference. */
static inline float sq_mode(const ctx_t *cx, float x) { return cx->op->math_mode == SKO_MATH_LIBM ? powf(x, 2.0f) : x * x; }

/* raytrace.h:36-44 direct_illumination (live part) =
 * bp::ambient_shading blinn_phong.h:13-17 + diffuse_shading :47-87 +
 * specular_shading :90-134.  The reference casts the same shadow ray once in
 * diffuse and once in specular; the answer is the same, so it is cast once. */
static v3 direct_illumination(ctx_t *cx, const sko_sphere *sp, v3 P, v3 N)
{
	const sko_scene *sc = cx->sc;
	v3 ambient = vmul(sc->ambient, sp->ambient);
	v3 diffuse = V(0, 0, 0), specular = V(0, 0, 0);
	v3 view = vnormalize(vsub(sc->cam_pos, P)); /* always the camera, blinn_phong.h:93 */
	for(int i = 0; i < sc->n_point_lights; i++)
	{
		const sko_point_light *pl = &sc->point_lights[i];
		v3 to_l = vsub(pl->position, P);
		v3 L = vnormalize(to_l);
		if(cx->op->use_shadows && shadowed(cx, P, L)) continue;
		float distance = vlength(to_l);
		float intensity = 1.0f / sq_mode(cx, fabsf(distance));
		/* blinn_phong.h:72 */
		diffuse = vadd(diffuse, vscale(vscale(vmul(sp->diffuse, pl->colour), intensity), max0(vdot(N, L))));
		/* blinn_phong.h:100-117 */
		v3 vl = vadd(view, L);
		v3 H = vdivs(vl, vlength(vl));
		specular = vadd(specular, vscale(vscale(vmul(sp->specular, pl->colour), intensity), powf_mode(cx, max0(vdot(N, H)), sp->power)));
	}
	/* blinn_phong.h:77-85 and :122-131 (empty at HEAD: scene.cpp never pushes a directional light; --strict-scn does).
	 * shadow(Scene, P, DirectionalLight), utils.h:60-76, is the point-light test along normalize(direction). */
	for(int i = 0; i < sc->n_directional_lights; i++)
	{
		const sko_directional_light *dl = &sc->directional_lights[i];
		v3 L = vnormalize(dl->direction);
		if(cx->op->use_shadows && shadowed(cx, P, L)) continue;
		diffuse = vadd(diffuse, vscale(vmul(sp->diffuse, dl->colour), max0(vdot(N, L))));
		v3 vl = vadd(view, L);
		v3 H = vdivs(vl, vlength(vl));
		specular = vadd(specular, vscale(vmul(sp->specular, dl->colour), powf_mode(cx, max0(vdot(N, H)), sp->power)));
	}
	v3 total = V(0, 0, 0);
	total = vadd(total, ambient);
	total = vadd(total, diffuse);
	total = vadd(total, specular);
	return total;
}

static v3 shade_from(ctx_t *cx, v3 o, v3 d, int depth, uint32_t node, int from_triangle);
static inline v3 shade(ctx_t *cx, v3 o, v3 d, int depth, uint32_t node) { return shade_from(cx, o, d, depth, node, -1); }

/* children per node of the counter RNG's tree: the --gillum rays, then (--legacy-reflect) a refraction and a reflection ray per light */
static inline uint32_t node_arity(const ctx_t *cx)
{
	uint32_t a = (uint32_t) cx->op->num_path_traces;
	if(cx->op->legacy_reflect) a += 2u * (uint32_t) (cx->sc->n_point_lights + cx->sc->n_directional_lights);
	return a;
}

/* utils.h:132-146 clamp(a, b, input) */
static inline float clampf(float a, float b, float x) { return x < a ? a : (x > b ? b : x); }

/* blinn_phong.h:156-184 fresnel().  `sqrt` is unqualified there: ::sqrt(double) on a float argument, its double result meeting floats
 * (sint: float * double -> double, narrowed by the assignment; cos_theta: narrowed).  powf(x, 2.0f) == x * x.  Everything else binary32. */
static float legacy_fresnel(v3 dir, v3 N, float mat_ior)
{
	float cos_internal = clampf(-1.0f, 1.0f, vdot(dir, N));
	float et = 1.0f, ior = mat_ior;
	if(cos_internal > 0) { float t = et; et = ior; ior = t; }
	float sint = (float) ((double) (et / ior) * sqrt((double) max0(1.0f - cos_internal * cos_internal)));
	if(sint >= 1.0f) return 1.0f;
	float cos_theta = (float) sqrt((double) max0(1 - sint * sint));
	cos_internal = fabsf(cos_internal);
	float Rs = ((ior * cos_internal) - (et * cos_theta)) / ((ior * cos_internal) + (et * cos_theta));
	float Rp = ((et * cos_internal) - (ior * cos_theta)) / ((ior * cos_internal) + (et * cos_theta));
	return (Rs * Rs + Rp * Rp) / 2.0f;
}

/* blinn_phong.h:143-153 refraction() */
static v3 legacy_refraction(v3 dir, v3 N, float ior)
{
	float dn = vdot(dir, N);
	float k = 1.0f - (ior * ior) * (1.0f - dn * dn);
	if(k < 0.0f) return V(0, 0, 0);
	return vsub(vscale(dir, ior), vscale(N, ior * dn + sqrtf(k)));
}

/* blinn_phong.h:137-140 reflect_direction(): glm::normalize(L - 2.0f * dot(L, N) * N) */
static v3 legacy_reflect_direction(v3 L, v3 N)
{
	return vnormalize(vsub(L, vscale(N, 2.0f * vdot(L, N))));
}

/* The three functions above on one (direction, normal, ior) triple, for the function-level pin against the reference's own
 * (tests/golden/ref_legacy_eval.npy.gz, written by oracle/_ref/ref_render --eval-legacy): out = {fresnel, refraction.xyz,
 * reflect_direction(normalize(dir), n).xyz}. */
void sko_legacy_eval(const float d[3], const float n[3], float ior, float out[7])
{
	v3 dir = V(d[0], d[1], d[2]), N = V(n[0], n[1], n[2]);
	out[0] = legacy_fresnel(dir, N, ior);
	v3 rf = legacy_refraction(dir, N, ior), rl = legacy_reflect_direction(vnormalize(dir), N);
	out[1] = rf.x; out[2] = rf.y; out[3] = rf.z;
	out[4] = rl.x; out[5] = rl.y; out[6] = rl.z;
}

/* raytrace.h:45-103, the part of direct_illumination() behind its early return */
static v3 legacy_terms(ctx_t *cx, v3 total_colour, v3 ray_dir, const sko_sphere *sp, v3 P, v3 N, int depth, uint32_t node)
{
	const sko_scene *sc = cx->sc;
	float fr = legacy_fresnel(ray_dir, N, sp->ior);
	v3 refraction_colour = V(0, 0, 0), reflection_colour = V(0, 0, 0);
	if((sp->specular.x != 0.0f || sp->specular.y != 0.0f || sp->specular.z != 0.0f) && depth > 0)
	{
		const uint32_t A = node_arity(cx), base = node * A + (uint32_t) cx->op->num_path_traces + 1u;
		const int nl = sc->n_point_lights + sc->n_directional_lights;
		for(int i = 0; i < nl; i++)
		{ /* :54-77 the point lights, :80-99 the directional ones: the same statements */
			v3 L = i < sc->n_point_lights ? vnormalize(vsub(sc->point_lights[i].position, P)) : vnormalize(sc->directional_lights[i - sc->n_point_lights].direction);
			if(fr < 1)
			{
				v3 rd = legacy_refraction(ray_dir, N, sp->ior);
				refraction_colour = vscale(shade_from(cx, P, rd, depth - 1, base + 2u * (uint32_t) i, -1), fr); /* (=, not +=) */
			}
			v3 md = legacy_reflect_direction(L, N);
			v3 c = shade_from(cx, P, md, depth - 1, base + 2u * (uint32_t) i + 1u, -1);
			reflection_colour = vadd(reflection_colour, vmul(vscale(sp->specular, 1 - fr), c)); /* (1 - fr) * specular * shade(...) */
		}
	}
	return vadd(vadd(total_colour, refraction_colour), reflection_colour); /* :102, in that order */
}

/* raytrace.h:22-30 uniform_sample_hemi */
static inline v3 sample_hemi(const ctx_t *cx, float r1, float r2)
{
	float s_theta = sqrtf(1 - sq_mode(cx, r1));
	float phi = (float) (2.0f * M_PI * r2); /* (2.0*pi)*double(r2), narrowed */
	float sn, cs;
	if(cx->op->math_mode == SKO_MATH_LIBM) { cs = cosf(phi); sn = sinf(phi); }
	else sko_sincos_shared(phi, &sn, &cs);
	return V(s_theta * cs, r1, s_theta * sn);
}

/* raytrace.h:107-136 montecarlo_global_illumination, including the basis mix
 * of :123-125 (perp_to_both.y/.z where perp_to_normal.y/.z belongs). */
static v3 global_illumination(ctx_t *cx, v3 P, v3 N, int depth, uint32_t node, int from_triangle)
{ /* from_triangle: --shade-triangles only, the triangle these child rays start on (they do not test it); -1 otherwise */
	const int n_rays = cx->op->num_path_traces;
	v3 total = V(0, 0, 0);
	v3 nt, nb;
	basis(N, &nt, &nb);
	float pdf = (float) (1 / M_PI);
	for(int i = 0; i < n_rays; i++)
	{
		float r1 = 0.0f, r2 = 0.0f;
		v3 child = V(0, 0, 0);
		if(cx->op->rng_mode == SKO_RNG_GLIBC_REPLAY)
		{
			r1 = (float) rand() / (float) RAND_MAX;
			r2 = (float) rand() / (float) RAND_MAX;
		}
		else if(depth - 1 > 0) sko_counter_draws(cx->op->seed, cx->pixel, cx->aa, node, (uint32_t) i, &r1, &r2);
		if(depth - 1 > 0)
		{
			v3 s = sample_hemi(cx, r1, r2);
			v3 w = V(s.x * nb.x + s.y * N.x + s.z * nt.x,
					 s.x * nb.y + s.y * N.y + s.z * nb.y,
					 s.x * nb.z + s.y * N.z + s.z * nb.z);
			child = shade_from(cx, vadds(P, 0.00001f), w, depth - 1, node * node_arity(cx) + (uint32_t) i + 1u, from_triangle);
		}
		/* depth-1 <= 0: shade() returns (0,0,0) at once (raytrace.h:142-145); r1*0/pdf == 0 for any finite r1 */
		total = vadd(total, vdivs(vscale(child, r1), pdf));
	}
	total = vdivs(total, (float) n_rays);
	return total;
}

/* raytrace.h:139-227 shade */
static v3 shade_from(ctx_t *cx, v3 o, v3 d, int depth, uint32_t node, int from_triangle)
{
	const sko_scene *sc = cx->sc;
	if(depth <= 0) return V(0, 0, 0);
	cx->n_rays++;

	float min_distance = INFINITY;
	int hit_sphere = -1;
	int hit_a_sphere = 0, hit_a_triangle = 0;
	for(int i = 0; i < sc->n_spheres; i++)
	{
		cx->n_sph_tests++;
		float distance = collision_distance(o, d, &sc->spheres[i]);
		if(intersection_occurs(distance))
		{
			hit_a_sphere = 1;
			if(distance < min_distance)
			{
				min_distance = distance;
				hit_sphere = i;
			}
		}
	}
	int hit_triangle = -1;
	for(int i = 0; i < sc->n_triangles; i++)
	{
		float t;
		cx->n_tri_tests++;
		if(triangle_test(o, d, &sc->triangles[i], &t))
		{
			if(cx->op->shade_triangles && (!(t > 0.0f) || i == from_triangle)) continue; /* skr_oracle.h: the mode's own rules */
			if(t < min_distance)
			{
				min_distance = t;
				hit_a_sphere = 0;
				hit_a_triangle = 1;
				hit_triangle = i;
			}
		}
	}
	if(!hit_a_sphere && !hit_a_triangle) return sc->background;
	if(hit_a_triangle && cx->op->shade_triangles)
	{ /* skr_oracle.h sko_options.shade_triangles: no counterpart in the reference */
		const sko_triangle *tr = &sc->triangles[hit_triangle];
		const sko_sphere *mt = &sc->triangle_materials[hit_triangle];
		v3 P = vadd(o, vscale(d, min_distance));
		v3 N = vnormalize(vcross(vsub(tr->v1, tr->v0), vsub(tr->v2, tr->v0)));
		if(vdot(N, d) > 0.0f) N = V(-N.x, -N.y, -N.z);
		cx->n_hits++;
		v3 direct = direct_illumination(cx, mt, P, N);
		if(cx->op->monte_carlo)
		{
			v3 indirect = global_illumination(cx, P, N, depth, node, hit_triangle);
			return vmul(vadd(vdivs(direct, (float) M_PI), vscale(indirect, 2.0f)), mt->diffuse);
		}
		return direct;
	}
	if(hit_a_sphere)
	{
		const sko_sphere *sp = &sc->spheres[hit_sphere];
		/* raytrace.h:197-201 recomputes t for the winner: same value as min_distance */
		float t = collision_distance(o, d, sp);
		v3 P = vadd(o, vscale(d, t));
		v3 N = vnormalize(vsub(P, sp->center));
		cx->n_hits++;
		v3 direct = direct_illumination(cx, sp, P, N);
		if(cx->op->legacy_reflect) direct = legacy_terms(cx, direct, d, sp, P, N, depth, node);
		if(cx->op->monte_carlo)
		{
			v3 indirect = global_illumination(cx, P, N, depth, node, -1);
			/* raytrace.h:213 */
			return vmul(vadd(vdivs(direct, (float) M_PI), vscale(indirect, 2.0f)), sp->diffuse);
		}
		return direct;
	}
	return V(0, 0, 0); /* triangle hit: raytrace.h:221-224 */
}

/* main.cpp:199-211: (unsigned char)(std::min(float(1), c) * 255).
 * std::min(1, c) = (c < 1) ? c : 1, so NaN -> 1 -> 255.  The float->uchar
 * conversion of a negative product is UB in C++; x86 compilers go through a
 * truncating int conversion and keep the low byte, restated explicitly. */
uint8_t sko_quantise(float c)
{
	float m = (c < 1.0f) ? c : 1.0f;
	float s = m * 255;
	if(!(s > -2147483648.0f)) return 0;
	return (uint8_t) (int32_t) s;
}

/* The primary direction of one sample: main.cpp:146-155 (jitter: one draw r for both axes, all-float) or :170-174 (pixel centre:
 * u and v formed in double).  ONE definition: the render loop below calls it, and it is exported so that a test can compare it
 * with the expressions of main.cpp evaluated independently (tests/test_statistics.py) — the jitter branch of
 * oracle/ref_driver.cpp is a hand restatement that no reference-held fixture covers. */
static inline v3 primary_direction(const sko_scene *scene, int x, int y, int jitter, float r, float inv_width, float inv_height, float aspect_ratio, float angle)
{
	float u, v;
	if(jitter)
	{
		u = (2 * ((x + r) * inv_width) - 1) * angle * aspect_ratio;
		v = (1 - 2 * ((y + r) * inv_height)) * angle;
	}
	else
	{
		u = (float) ((2 * ((x + 0.5) * inv_width) - 1) * angle * aspect_ratio);
		v = (float) ((1 - 2 * ((y + 0.5) * inv_height)) * angle);
	}
	return vadd(vadd(scene->cam_dir, vscale(scene->cam_right, u)), vscale(scene->cam_up, v));
}

void sko_primary_direction(const sko_scene *scene, int width, int height, float fov, int x, int y, int jitter, float r, float out[3])
{
	const v3 d = primary_direction(scene, x, y, jitter, r, 1 / (float) width, 1 / (float) height, width / (float) height, (float) tan(M_PI * 0.5 * fov / 180.));
	out[0] = d.x;
	out[1] = d.y;
	out[2] = d.z;
}

int sko_render(const sko_scene *scene, const sko_options *opt, uint8_t *rgb, float *rgbf, uint64_t *stats)
{
	const int W = opt->width, H = opt->height;
	if(W <= 0 || H <= 0 || opt->y0 < 0 || opt->y1 > H || opt->y0 > opt->y1) return 1;
	const int uses_rand = opt->grid_size > 0 || opt->monte_carlo;
	int threads = opt->threads > 0 ? opt->threads : 1;
	if(opt->rng_mode == SKO_RNG_GLIBC_REPLAY)
	{
		if(uses_rand) threads = 1;
		srand((unsigned) opt->seed); /* main.cpp:400 with time(0) pinned */
	}
	/* main.cpp:134-137, loop invariants */
	const float inv_width = 1 / (float) W;
	const float inv_height = 1 / (float) H;
	const float aspect_ratio = W / (float) H;
	const float angle = (float) tan(M_PI * 0.5 * opt->fov / 180.);
	uint64_t tot[5] = {0, 0, 0, 0, 0};

	/* replay mode must visit every row from 0 so the rand() stream lines up */
	const int ystart = (opt->rng_mode == SKO_RNG_GLIBC_REPLAY && uses_rand) ? 0 : opt->y0;
	/* work items = (row, 32-pixel span), handed out dynamically: rows differ a lot in cost
	 * (sky vs ground) and a band may have fewer rows than there are threads */
	const int span = 32;
	const int spans_per_row = (W + span - 1) / span;
	const long n_items = (long) (opt->y1 - ystart) * spans_per_row;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : tot[:5])
#endif
	for(long item = 0; item < n_items; item++)
	{
		const int y = ystart + (int) (item / spans_per_row);
		const int xs = (int) (item % spans_per_row) * span;
		const int xe = (xs + span < W) ? xs + span : W;
		ctx_t cx = {scene, opt, 0, 0, 0, 0, 0, 0, 0};
		for(int x = xs; x < xe; x++)
		{
			v3 px = V(0, 0, 0);
			cx.pixel = (uint32_t) y * (uint32_t) W + (uint32_t) x;
			if(opt->grid_size > 0)
			{ /* main.cpp:140-166: g*g samples, ONE draw r used for both axes */
				const int g = opt->grid_size;
				for(int s = 0; s < g * g; s++)
				{
					cx.aa = (uint32_t) s;
					float r = (opt->rng_mode == SKO_RNG_GLIBC_REPLAY) ? (float) rand() / (float) RAND_MAX : sko_counter_jitter(opt->seed, cx.pixel, cx.aa);
					v3 dir = primary_direction(scene, x, y, 1, r, inv_width, inv_height, aspect_ratio, angle);
					px = vadd(px, shade(&cx, scene->cam_pos, dir, opt->max_depth, 0));
				}
				px = vdivs(px, (float) (g * g));
			}
			else
			{ /* main.cpp:168-182: pixel centre, u and v formed in double */
				cx.aa = 0;
				v3 dir = primary_direction(scene, x, y, 0, 0.0f, inv_width, inv_height, aspect_ratio, angle);
				px = shade(&cx, scene->cam_pos, dir, opt->max_depth, 0);
			}
			if(y >= opt->y0)
			{
				size_t o = ((size_t) (y - opt->y0) * W + x) * 3;
				if(rgb) { rgb[o] = sko_quantise(px.x); rgb[o + 1] = sko_quantise(px.y); rgb[o + 2] = sko_quantise(px.z); }
				if(rgbf) { rgbf[o] = px.x; rgbf[o + 1] = px.y; rgbf[o + 2] = px.z; }
			}
		}
		tot[0] += cx.n_rays; tot[1] += cx.n_hits; tot[2] += cx.n_shadow; tot[3] += cx.n_sph_tests; tot[4] += cx.n_tri_tests;
	}
	if(stats) memcpy(stats, tot, sizeof tot);
	return 0;
}

int sko_write_ppm(const char *path, int w, int h, const uint8_t *rgb)
{
	FILE *f = fopen(path, "wb");
	if(!f) return 1;
	fprintf(f, "P6\n%d %d\n255\n", w, h);
	fwrite(rgb, 1, (size_t) w * h * 3, f);
	fclose(f);
	return 0;
}

/* ------------------------------------------------------------ .scn loader */
/* scene.cpp:12-227.  Line-oriented; first byte '#' = comment; first token
 * selects the command; a stateful current material applies to later spheres
 * and triangles; ambient_light accumulates; directional lights are parsed and
 * dropped; spherical_fog is skipped (UB in the reference); triangle indices are
 * read as floats. */
int sko_scene_load(const char *path, sko_scene *out) { return sko_scene_load_ex(path, 0, out); }

int sko_scene_load_ex(const char *path, int strict, sko_scene *out)
{
	memset(out, 0, sizeof *out);
	int cap_d = 0;
	FILE *fp = fopen(path, "r");
	if(!fp) return 1;
	out->film_w = 1920; out->film_h = 1080; out->max_depth_parsed = 1; /* scene.h:15,26 */
	/* camera.h:16-22 default camera: all zero */
	sko_sphere mat;
	memset(&mat, 0, sizeof mat);
	mat.power = 1.0f; mat.ior = 1.0f; /* material.h:16-17 */
	int cap_s = 0, cap_t = 0, cap_l = 0, cap_v = 0;
	sko_vec3 *verts = NULL;
	char line[1024];
	while(fgets(line, 1024, fp))
	{
		if(line[0] == '#') continue;
		char command[1024];
		if(sscanf(line, "%s ", command) < 1) continue;
		if(!strcmp(command, "sphere"))
		{
			float x = 0, y = 0, z = 0, r = 0;
			sscanf(line, "sphere %f %f %f %f", &x, &y, &z, &r);
			if(out->n_spheres == cap_s) { cap_s = cap_s ? cap_s * 2 : 16; out->spheres = realloc(out->spheres, sizeof(sko_sphere) * cap_s); }
			sko_sphere sp = mat;
			sp.center = V(x, y, z);
			sp.radius = r;
			out->spheres[out->n_spheres++] = sp;
		}
		else if(!strcmp(command, "vertex"))
		{
			float x = 0, y = 0, z = 0;
			sscanf(line, "vertex %f %f %f", &x, &y, &z);
			if(out->n_vertices == cap_v) { cap_v = cap_v ? cap_v * 2 : 1024; verts = realloc(verts, sizeof(sko_vec3) * cap_v); }
			verts[out->n_vertices++] = V(x, y, z);
		}
		else if(!strcmp(command, "triangle"))
		{
			float a = 0, b = 0, c = 0; /* scene.cpp:69-70: indices read as floats */
			sscanf(line, "triangle %f %f %f", &a, &b, &c);
			long i0 = (long) a, i1 = (long) b, i2 = (long) c;
			if(i0 < 0 || i1 < 0 || i2 < 0 || i0 >= out->n_vertices || i1 >= out->n_vertices || i2 >= out->n_vertices)
			{ /* the reference indexes out of bounds here; no shipped scene does */
				out->n_bad_triangles++;
				continue;
			}
			if(out->n_triangles == cap_t)
			{
				cap_t = cap_t ? cap_t * 2 : 1024;
				out->triangles = realloc(out->triangles, sizeof(sko_triangle) * cap_t);
				out->triangle_materials = realloc(out->triangle_materials, sizeof(sko_sphere) * cap_t);
			}
			sko_triangle tr = {verts[i0], verts[i1], verts[i2]};
			out->triangle_materials[out->n_triangles] = mat; /* shapes.h:26: the Triangle keeps the current material */
			out->triangles[out->n_triangles++] = tr;
		}
		else if(!strcmp(command, "camera"))
		{
			float p[3] = {0, 0, 0}, d[3] = {0, 0, 0}, u[3] = {0, 0, 0}, ha = 0;
			sscanf(line, "camera %f %f %f %f %f %f %f %f %f %f\n", &p[0], &p[1], &p[2], &d[0], &d[1], &d[2], &u[0], &u[1], &u[2], &ha);
			out->cam_pos = V(p[0], p[1], p[2]);
			out->cam_dir = V(d[0], d[1], d[2]); /* scene.cpp:92-93 discard normalize(): magnitudes kept */
			out->cam_up = V(u[0], u[1], u[2]);
			out->cam_half_angle = ha;
			out->cam_right = vcross(vscale(out->cam_dir, -1.0f), out->cam_up); /* camera.h:30 */
		}
		else if(!strcmp(command, "film_resolution")) sscanf(line, "film_resolution %d %d", &out->film_w, &out->film_h);
		else if(!strcmp(command, "background"))
		{
			float r = 0, g = 0, b = 0;
			sscanf(line, "background %f %f %f", &r, &g, &b);
			out->background = V(r, g, b);
		}
		else if(!strcmp(command, "material"))
		{
			float m[14] = {0};
			sscanf(line, "material %f %f %f %f %f %f %f %f %f %f %f %f %f %f", &m[0], &m[1], &m[2], &m[3], &m[4], &m[5], &m[6], &m[7], &m[8], &m[9], &m[10], &m[11], &m[12], &m[13]);
			mat.ambient = V(m[0], m[1], m[2]);
			mat.diffuse = V(m[3], m[4], m[5]);
			mat.specular = V(m[6], m[7], m[8]);
			mat.power = m[9];
			mat.transmissive = V(m[10], m[11], m[12]);
			mat.ior = m[13];
		}
		else if(!strcmp(command, "directional_light"))
		{
			if(!strict) out->n_directional_dropped++; /* scene.cpp:139-163: built, echoed, never pushed */
			else
			{
				float r = 0, g = 0, b = 0, x = 0, y = 0, z = 0;
				sscanf(line, "directional_light %f %f %f %f %f %f", &r, &g, &b, &x, &y, &z);
				if(r > 1) r = 1; /* scene.cpp:143-154 */
				if(g > 1) g = 1;
				if(b > 1) b = 1;
				if(out->n_directional_lights == cap_d) { cap_d = cap_d ? cap_d * 2 : 8; out->directional_lights = realloc(out->directional_lights, sizeof(sko_directional_light) * cap_d); }
				sko_directional_light dl = {V(x, y, z), V(r, g, b)};
				out->directional_lights[out->n_directional_lights++] = dl;
			}
		}
		else if(!strcmp(command, "point_light"))
		{
			float r = 0, g = 0, b = 0, x = 0, y = 0, z = 0;
			sscanf(line, "point_light %f %f %f %f %f %f", &r, &g, &b, &x, &y, &z);
			if(out->n_point_lights == cap_l) { cap_l = cap_l ? cap_l * 2 : 8; out->point_lights = realloc(out->point_lights, sizeof(sko_point_light) * cap_l); }
			sko_point_light pl = {V(x, y, z), V(r, g, b)};
			out->point_lights[out->n_point_lights++] = pl;
		}
		else if(!strcmp(command, "ambient_light"))
		{
			float r = 0, g = 0, b = 0;
			sscanf(line, "ambient_light %f %f %f", &r, &g, &b);
			out->ambient = vadd(out->ambient, V(r, g, b)); /* scene.cpp:187-189: += */
		}
		else if(!strcmp(command, "max_depth"))
		{
			float n = 0;
			sscanf(line, "max_depth %f", &n);
			out->max_depth_parsed = (int) n;
		}
		else if(!strcmp(command, "output_image")) { /* echo only */ }
		else if(!strcmp(command, "spherical_fog")) out->n_fog_skipped++;
		else out->n_unknown++;
	}
	fclose(fp);
	free(verts);
	return 0;
}

void sko_scene_free(sko_scene *s)
{
	free(s->spheres);
	free(s->triangles);
	free(s->triangle_materials);
	free(s->point_lights);
	free(s->directional_lights);
	memset(s, 0, sizeof *s);
}
