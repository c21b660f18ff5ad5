/* Dev aid (not product): accuracy of the shared binary32 sincos (oracle/skr_oracle.c sko_sincos_shared — the device's
 * sincos_spec is compared with it bit for bit by tests/test_gpu_units.py) over EVERY binary32 value of [0, 2 pi], against the
 * correctly rounded sine and cosine (binary64 libm, rounded once: exact to well under 1e-7 ulp of a binary32 result).
 * Build and run:  gcc -O2 -fopenmp -o /tmp/sincos_exhaustive tools/sincos_exhaustive.c -Loracle -loracle -lm && LD_LIBRARY_PATH=oracle /tmp/sincos_exhaustive */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
void sko_sincos_shared(float phi, float *s, float *c);
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static double ulp_err(float got, double want)
{
	const float w = (float) want;
	int e;
	frexp((double) w, &e);
	const double ulp = w == 0 ? ldexp(1.0, -149) : ldexp(1.0, e - 24);
	return fabs((double) got - want) / ulp;
}
int main(void)
{
	const uint32_t top = asu(6.2831855f); /* float(2 pi): the largest phi = float(2 pi r2), r2 in [0, 1], can be */
	double ms = 0, mc = 0;
	float as = 0, ac = 0;
	uint64_t over_s = 0, over_c = 0, inexact_s = 0, inexact_c = 0;
#pragma omp parallel
	{
		double lms = 0, lmc = 0;
		float las = 0, lac = 0;
		uint64_t los = 0, loc = 0, lis = 0, lic = 0;
#pragma omp for schedule(static)
		for(uint32_t u = 0; u <= top; u++)
		{
			const float phi = asf(u);
			float s, c;
			sko_sincos_shared(phi, &s, &c);
			const double ws = sin((double) phi), wc = cos((double) phi);
			const double es = ulp_err(s, ws), ec = ulp_err(c, wc);
			if(es > lms) { lms = es; las = phi; }
			if(ec > lmc) { lmc = ec; lac = phi; }
			los += es > 1.0;
			loc += ec > 1.0;
			lis += s != (float) ws;
			lic += c != (float) wc;
		}
#pragma omp critical
		{
			if(lms > ms) { ms = lms; as = las; }
			if(lmc > mc) { mc = lmc; ac = lac; }
			over_s += los; over_c += loc; inexact_s += lis; inexact_c += lic;
		}
	}
	printf("binary32 inputs in [0, 2 pi]: %u\n", top + 1);
	printf("sin: max error %.4f ulp at phi = %a; more than 1 ulp off: %llu (%.5f %%); not the correctly rounded value: %llu (%.3f %%)\n", ms, as,
		   (unsigned long long) over_s, 100.0 * over_s / (top + 1.0), (unsigned long long) inexact_s, 100.0 * inexact_s / (top + 1.0));
	printf("cos: max error %.4f ulp at phi = %a; more than 1 ulp off: %llu (%.5f %%); not the correctly rounded value: %llu (%.3f %%)\n", mc, ac,
		   (unsigned long long) over_c, 100.0 * over_c / (top + 1.0), (unsigned long long) inexact_c, 100.0 * inexact_c / (top + 1.0));
	return 0;
}
