/* Dev aid (not product): for d = float(M_PI) and d = float(1 / M_PI), on which binary32 x does  fma(x, zh, x * zl)  with
 * zh = RN(1/d), zl = RN(1/d - zh) differ from the correctly rounded x / d?  All 2^32 x; mismatches by biased exponent of x.
 * (csrc/device_math.h div3_const takes the short form only for x == 0 or |x| >= 2^-100: biased exponent >= 27.)
 * Build and run: gcc -O2 -fopenmp -mfma -ffp-contract=off -o /tmp/constdiv tools/constdiv_exhaustive.c -lm && /tmp/constdiv */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void)
{
	const float ds[2] = {(float) 3.14159265358979323846, (float) (1 / 3.14159265358979323846)};
	for(int di = 0; di < 2; di++)
	{
		const float d = ds[di];
		const double inv = 1.0 / (double) d;
		const float zh = (float) inv, zl = (float) (inv - (double) zh);
		uint64_t hist[256] = {0};
#pragma omp parallel
		{
			uint64_t lh[256] = {0};
#pragma omp for schedule(static)
			for(uint64_t u = 0; u < (1ull << 32); u++)
			{
				const float x = asf((uint32_t) u), want = x / d, got = fmaf(x, zh, x * zl);
				if(!(want != want && got != got) && asu(got) != asu(want)) lh[(u >> 23) & 0xff]++;
			}
#pragma omp critical
			for(int i = 0; i < 256; i++) hist[i] += lh[i];
		}
		uint64_t below = 0, above = 0, zero_bad = (asu(fmaf(0.0f, zh, 0.0f * zl)) != asu(0.0f / d)) + (asu(fmaf(-0.0f, zh, -0.0f * zl)) != asu(-0.0f / d));
		int top = -1;
		for(int i = 0; i < 256; i++) { if(hist[i]) top = i; if(i < 27) below += hist[i]; else above += hist[i]; }
		printf("d = %a (zh = %a, zl = %a): mismatches with |x| < 2^-100: %llu (highest biased exponent with one: %d); with |x| >= 2^-100, inf, NaN: %llu; at +-0: %llu\n",
			   d, zh, zl, (unsigned long long) below, top, (unsigned long long) above, (unsigned long long) zero_bad);
	}
	return 0;
}
