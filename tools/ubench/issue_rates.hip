// Dev aid (not product): per-instruction issue cost on gfx950, stated in SHADER CYCLES and in wall time, with the clock the chip
// held while it ran (round 3: the round-2 version printed a wall figure "@2.4 GHz" although the all-CU runs sat at 1.3-2.1 GHz).
//
// Every wave runs ITER x 16 independent instances of one instruction between two reads of BOTH clocks:
//   s_memtime      the shader clock (one tick per shader cycle: /opt/skills/guides/MI355X_MICROARCH.md "s_memtime tick")
//   s_memrealtime  the constant 100 MHz clock
// so  cycles per wave-instruction = d(memtime) / (16 ITER),  clock = d(memtime) / d(memrealtime) x 100 MHz,  and the SIMD's
// throughput figure is cycles per wave-instruction / waves per SIMD.  Launch shapes: every CU (256 workgroups x waves per
// SIMD) and one XCD only (the same grid, workgroups whose index is not a multiple of 8 leave at once: workgroup i lands on
// XCD i mod 8), which shows what the chip-wide load does to the clock.  Each shape is launched back to back for >= 0.25 s
// before the launch that is read (DVFS settles in that time), and runs ~1 ms or more per launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed, int iters, int one_xcd)
{
	if(one_xcd && (blockIdx.x & 7)) return;
	__shared__ float lds[1024];
	lds[threadIdx.x] = seed + threadIdx.x;
	__syncthreads();
	float r[16];
	double dd[16];
	unsigned long long uu[16];
#pragma unroll
	for(int i = 0; i < 16; i++) { r[i] = seed + i + threadIdx.x; dd[i] = r[i]; uu[i] = (unsigned long long) (i + threadIdx.x); }
	float2 p[16];
#pragma unroll
	for(int i = 0; i < 16; i++) p[i] = make_float2(r[i], r[i] + 1);
	float4 q[4];
	int addr = (threadIdx.x & 63) * 4, zero = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
	for(int it = 0; it < iters; it++)
	{
		if(OP == 0) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
			REP16(X)
#undef X
		} else if(OP == 1) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
			REP16(X)
#undef X
		} else if(OP == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(dd[i]) : "v"(dd[(i + 1) & 15]));
			REP16(X)
#undef X
		} else if(OP == 3) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(uu[i]) : "v"((unsigned) i), "v"((unsigned) threadIdx.x) : "vcc");
			REP16(X)
#undef X
		} else if(OP == 4) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
			REP16(X)
#undef X
		} else if(OP == 5) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
			REP16(X)
#undef X
		} else if(OP == 6) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(seed));
			REP16(X)
#undef X
		} else if(OP == 7) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(r[i]) : "v"(addr));
			REP16(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if(OP == 8) {
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(zero));
			REP16(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if(OP == 9) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(seed));
			REP16(X)
#undef X
		} else if(OP == 10) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
			REP16(X)
#undef X
		} else if(OP == 11) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(seed) : "vcc");
			REP16(X)
#undef X
		} else if(OP == 12) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) & 15]));
			REP16(X)
#undef X
		} else if(OP == 13) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(seed) : "vcc");
			REP16(X)
#undef X
		} else if(OP == 14) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
			REP16(X)
#undef X
		} else if(OP == 15) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
			REP16(X)
#undef X
		} else if(OP == 16) {
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(dd[i]));
			REP16(X)
#undef X
		} else if(OP == 17) {
#define X(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(r[i]) : "s20");
			REP16(X)
#undef X
		} else if(OP == 18) {
#define X(i) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(r[i]));
			REP16(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if(OP == 19) { // wave-uniform 16-byte loads through the scalar cache: how the triangle walk reads its triangles (shade_common.h)
			typedef unsigned v4u __attribute__((ext_vector_type(4)));
			v4u sq0, sq1, sq2, sq3;
#define X(i) asm volatile("s_load_dwordx4 %0, %1, 0x" #i "0" : "=s"(i & 2 ? (i & 1 ? sq3 : sq2) : (i & 1 ? sq1 : sq0)) : "s"(out));
			X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)
#undef X
#define X(i) asm volatile("s_load_dwordx4 %0, %1, 0x" #i "0" : "=s"(sq0) : "s"(out));
			X(a) X(b) X(c) X(d) X(e) X(f)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
			asm volatile("" : : "s"(sq0), "s"(sq1), "s"(sq2), "s"(sq3));
		} else if(OP == 20) { // a DEPENDENT chain of v_fma_f32: what one wave issues when every instruction waits for the one before
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[0]) : "v"(seed));
			REP16(X)
#undef X
		} else if(OP == 22) { // 128 per trip: the loop's s_cbranch (and the instruction refetch behind it) per 128 instead of per 16
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
			REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
		} else if(OP == 23) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(seed));
			REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
		} else if(OP == 24) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
			REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
		} else if(OP == 25) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(dd[i]) : "v"(dd[(i + 1) & 15]));
			REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
		} else if(OP == 26) { // 128 v_mul per trip with a TAKEN forward branch (to the very next instruction) after every 16
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
#define BR(n) asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 .Lt" #n "_%=\n\ts_nop 0\n.Lt" #n "_%=:" : : : "scc");
			REP16(X) BR(0) REP16(X) BR(1) REP16(X) BR(2) REP16(X) BR(3) REP16(X) BR(4) REP16(X) BR(5) REP16(X) BR(6) REP16(X) BR(7)
#undef BR
#undef X
		} else if(OP == 27) { // the same with the branches NOT taken
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
#define BR(n) asm volatile("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 .Ln" #n "_%=\n\ts_nop 0\n.Ln" #n "_%=:" : : : "scc");
			REP16(X) BR(0) REP16(X) BR(1) REP16(X) BR(2) REP16(X) BR(3) REP16(X) BR(4) REP16(X) BR(5) REP16(X) BR(6) REP16(X) BR(7)
#undef BR
#undef X
		} else if(OP == 28) { // 128 v_mul per trip, 64 of them issued with EXEC = 0 (what an un-skipped divergent block costs when no lane is in it)
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
#define OFF asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0" : : : "s20", "s21");
#define ON asm volatile("s_mov_b64 exec, s[20:21]" : : : "s20", "s21");
			REP16(X) OFF REP16(X) ON REP16(X) OFF REP16(X) ON REP16(X) OFF REP16(X) ON REP16(X) OFF REP16(X) ON
#undef OFF
#undef ON
#undef X
		} else if(OP == 21) { // the closest-hit sphere loop's mix (render_nodes.hip closest_pair_deferred, main path of one trip): 9 plain + 9 packed + 4 compares
			asm volatile("v_sub_f32 %0, %0, %1\n\tv_pk_mul_f32 %2, %2, %3\n\tv_mul_f32 %4, %4, %1\n\tv_pk_add_f32 %5, %5, %3\n\tv_add_f32 %0, %0, %4\n\tv_pk_mul_f32 %2, %2, %5\n\tv_cmp_le_f32 vcc, 0, %0\n\tv_pk_add_f32 %3, %3, %2"
						 : "+v"(r[0]), "+v"(r[1]), "+v"(p[0]), "+v"(p[1]), "+v"(r[2]), "+v"(p[2]) : : "vcc");
			asm volatile("v_sub_f32 %0, %0, %1\n\tv_pk_mul_f32 %2, %2, %3\n\tv_mul_f32 %4, %4, %1\n\tv_pk_add_f32 %5, %5, %3\n\tv_add_f32 %0, %0, %4\n\tv_pk_mul_f32 %2, %2, %5\n\tv_cmp_gt_f32 vcc, 0, %0\n\tv_pk_add_f32 %3, %3, %2"
						 : "+v"(r[3]), "+v"(r[4]), "+v"(p[3]), "+v"(p[4]), "+v"(r[5]), "+v"(p[5]) : : "vcc");
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
	float acc = 0;
	double dacc = 0;
#pragma unroll
	for(int i = 0; i < 16; i++) { acc += r[i] + p[i].x + p[i].y + (float) uu[i]; dacc += dd[i]; }
	acc += q[0].x + q[1].y + q[2].z + q[3].w;
	if(acc == 123.456f || dacc == 1.25) out[0] = 1; // keep results alive
	if((threadIdx.x & 63) == 0)
	{
		const size_t w = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
		out[1 + 4 * w] = t1 - t0;
		out[2 + 4 * w] = w1 - w0;
		out[3 + 4 * w] = w0;
		out[4 + 4 * w] = w1;
	}
}

static double median(std::vector<double> v)
{
	std::sort(v.begin(), v.end());
	return v.empty() ? 0.0 : v[v.size() / 2];
}

static int g_iters = 16384;
static float g_warm_ms = 250.0f;
static int g_plain = 0; // --plain: a fixed number of launches per shape and nothing else (for rocprofv3 --pmc GRBM_GUI_ACTIVE / --kernel-trace passes)

template <int OP>
static void run(const char *name, int per_iter = 16)
{
	for(int shape = 0; shape < 4; shape++)
	{ // 256-thread workgroups = one wave per SIMD each; wps of them per CU.  shape 3: one XCD only, 4 waves per SIMD
		const int wps = shape == 3 ? 4 : (1 << shape), one_xcd = shape == 3;
		const int blocks = 256 * wps;
		const int iters = per_iter > 16 ? g_iters * 16 / per_iter : g_iters;
		unsigned long long *d;
		const size_t words = 1 + (size_t) blocks * 16;
		(void) hipMalloc(&d, words * 8);
		(void) hipMemset(d, 0, words * 8);
		if(g_plain)
		{
			for(int i = 0; i < 12; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, iters, one_xcd);
			(void) hipDeviceSynchronize();
			(void) hipFree(d);
			continue;
		}
		hipEvent_t e0, e1;
		(void) hipEventCreate(&e0);
		(void) hipEventCreate(&e1);
		// settle: back-to-back launches for >= g_warm_ms
		float warm_ms = 0;
		int launches = 0;
		while(warm_ms < g_warm_ms && launches < 4000)
		{
			(void) hipEventRecord(e0);
			for(int i = 0; i < 8; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, iters, one_xcd);
			(void) hipEventRecord(e1);
			(void) hipEventSynchronize(e1);
			float ms = 0;
			(void) hipEventElapsedTime(&ms, e0, e1);
			warm_ms += ms;
			launches += 8;
		}
		(void) hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, iters, one_xcd);
		(void) hipEventRecord(e1);
		(void) hipDeviceSynchronize();
		float ms = 0;
		(void) hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h(words);
		(void) hipMemcpy(h.data(), d, words * 8, hipMemcpyDeviceToHost);
		std::vector<double> cyc, ghz, ns, start, end;
		const double n_instr = (double) iters * per_iter;
		for(int w = 0; w < blocks * 4; w++)
		{
			const double dt = (double) h[1 + 4 * w], dw = (double) h[2 + 4 * w];
			if(dw <= 0) continue; // (one-XCD shape: the workgroups that left)
			cyc.push_back(dt / n_instr);
			ghz.push_back(dt / dw * 0.1);
			ns.push_back(dw * 10.0 / n_instr);
			start.push_back((double) h[3 + 4 * w]);
			end.push_back((double) h[4 + 4 * w]);
		}
		const double c = median(cyc), g = median(ghz), n = median(ns);
		const double s0 = *std::min_element(start.begin(), start.end()), s1 = *std::max_element(start.begin(), start.end()), e1x = *std::max_element(end.begin(), end.end());
		printf("%-16s %-9s waves/SIMD %d: cycles per wave-instr %6.3f  per SIMD %6.3f | s_memtime/s_memrealtime %5.3f GHz | ns per wave-instr per SIMD %6.3f (a wave's own span), %6.3f (first start to last end), %6.3f (hipEvent) | waves %zu, starts spread over %.1f us of a %.1f us launch\n",
			   name, one_xcd ? "one XCD" : "all CUs", wps, c, c / wps, g, n / wps, (e1x - s0) * 10.0 / n_instr / wps, ms * 1e6 / n_instr / wps, cyc.size(), (s1 - s0) * 0.01, (e1x - s0) * 0.01);
		fflush(stdout);
		(void) hipFree(d);
		(void) hipEventDestroy(e0);
		(void) hipEventDestroy(e1);
	}
}

int main(int argc, char **argv)
{
	bool quick = false;
	for(int i = 1; i < argc; i++)
	{
		if(!strcmp(argv[i], "--quick")) quick = true;
		else if(!strcmp(argv[i], "--plain")) g_plain = 1;
		else if(!strcmp(argv[i], "--iters") && i + 1 < argc) g_iters = atoi(argv[++i]);
		else if(!strcmp(argv[i], "--warm-ms") && i + 1 < argc) g_warm_ms = (float) atof(argv[++i]);
	}
	printf("iters %d x 16 instructions per wave, settle %.0f ms%s\n", g_iters, g_warm_ms, g_plain ? ", plain launches" : "");
	run<22>("v_mul_f32 x128", 128);
	run<23>("v_fma_f32 x128", 128);
	run<24>("v_pk_fma_f32 x128", 128);
	run<25>("v_fma_f64 x128", 128);
	run<26>("x128 +8 taken br", 128);
	run<27>("x128 +8 untaken", 128);
	run<28>("x128 half EXEC=0", 128);
	if(quick) return 0;
	run<0>("v_mul_f32");
	run<9>("v_fma_f32");
	run<20>("v_fma_f32 dep.");
	run<1>("v_pk_mul_f32");
	run<10>("v_pk_fma_f32");
	run<21>("sphere-loop mix", 16);
	run<2>("v_fma_f64");
	if(quick) return 0;
	run<12>("v_mul_f64");
	run<16>("v_rsq_f64");
	run<3>("v_mad_u64_u32");
	run<14>("v_mul_lo_u32");
	run<15>("v_xor_b32");
	run<4>("v_sqrt_f32");
	run<5>("v_rcp_f32");
	run<6>("v_mov_b32");
	run<11>("v_cndmask_b32");
	run<13>("v_cmp_lt_f32");
	run<17>("v_readlane_b32");
	run<7>("ds_bpermute_b32");
	run<8>("ds_read_b128 bc");
	run<19>("s_load_dwordx4");
	run<18>("ds_write_b32");
	return 0;
}
