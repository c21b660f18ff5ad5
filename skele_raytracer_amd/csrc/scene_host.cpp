// .scn loader, options defaults, PPM writer: the host side of the drop-in
// boundary.  Behavioural restatement of reference src/scene.cpp:12-227,
// src/utils.h:26-34 and src/main.cpp:199-211 (citations relative to
// /root/reference).  No GPU code in this file.
#include "scene_host.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <fstream>

static thread_local char g_err[512] = "";

void skr_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}

extern "C" const char *skr_last_error(void) { return g_err; }

namespace {

// Reads up to n floats the way sscanf("%f ...") does (strtof underneath);
// fields that are missing stay 0 (the reference leaves them uninitialised).
int read_floats(const char *p, float *dst, int n)
{
	int got = 0;
	for(; got < n; got++)
	{
		char *end = nullptr;
		float v = strtof(p, &end);
		if(end == p) break;
		dst[got] = v;
		p = end;
	}
	return got;
}

struct Material { // material.h:9-17
	float ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
	float power = 1.0f;
	float ior = 1.0f; // (only the dead code of raytrace.h:45-103 reads it: --legacy-reflect)
};

} // namespace

void skr_scene::finalize()
{
	const int ns = info.n_spheres, nt = info.n_triangles, nl = info.n_point_lights;
	sph_geom.resize(ns);
	sph_amb.resize(ns);
	sph_kd.resize(ns);
	sph_ks.resize(ns);
	for(int i = 0; i < ns; i++)
	{
		const float *s = &raw_spheres[(size_t) i * 14];
		sph_geom[i] = {s[0], s[1], s[2], s[3] * s[3]};
		sph_amb[i] = {info.ambient[0] * s[4], info.ambient[1] * s[5], info.ambient[2] * s[6], s[13]};
		sph_kd[i] = {s[7], s[8], s[9], 0.0f};
		sph_ks[i] = {s[10], s[11], s[12], (size_t) i < raw_sphere_ior.size() ? raw_sphere_ior[i] : 1.0f};
	}
	// point lights first, then (--strict-scn) the directional ones: the order blinn_phong.h:50-85 / :95-131 adds them in
	const int nd = (int) (raw_directional_lights.size() / 6);
	lights.resize((size_t) (nl + nd) * 2);
	for(int i = 0; i < nl; i++)
	{
		const float *l = &raw_point_lights[(size_t) i * 6];
		lights[2 * i] = {l[0], l[1], l[2], 0.0f};
		lights[2 * i + 1] = {l[3], l[4], l[5], 0.0f};
	}
	for(int i = 0; i < nd; i++)
	{
		const float *l = &raw_directional_lights[(size_t) i * 6];
		lights[2 * (nl + i)] = {l[0], l[1], l[2], 1.0f}; // .w = 1: a direction, not a position (shade_common.h light_term)
		lights[2 * (nl + i) + 1] = {l[3], l[4], l[5], 0.0f};
	}
	// The triangle walk only answers "does any triangle accept this ray before tmin" (raytrace.h:168-176 turns
	// any such hit black), so the order of tris[] is free: store the triangles along a Morton curve through the
	// centres of their accept regions, which makes every run of tri_chunk_size triangles spatially tight.
	std::vector<int> order(nt);
	{
		std::vector<double> ctr((size_t) nt * 3);
		double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
		for(int i = 0; i < nt; i++)
		{
			const float *t = &raw_triangles[(size_t) i * 9];
			for(int k = 0; k < 3; k++)
			{
				// accept region (v0, v0 - e1, v0 + e2), see build_triangle_chunks()
				const double c = (double) t[k] + (((double) t[6 + k] - t[k]) - ((double) t[3 + k] - t[k])) / 3.0;
				ctr[(size_t) i * 3 + k] = c;
				if(std::isfinite(c)) { lo[k] = std::min(lo[k], c); hi[k] = std::max(hi[k], c); }
			}
		}
		auto spread = [](uint64_t v) { // 21 bits -> every third bit
			v &= 0x1fffff;
			v = (v | v << 32) & 0x1f00000000ffffull;
			v = (v | v << 16) & 0x1f0000ff0000ffull;
			v = (v | v << 8) & 0x100f00f00f00f00full;
			v = (v | v << 4) & 0x10c30c30c30c30c3ull;
			v = (v | v << 2) & 0x1249249249249249ull;
			return v;
		};
		std::vector<uint64_t> key(nt);
		for(int i = 0; i < nt; i++)
		{
			uint64_t code = 0;
			bool ok = true;
			for(int k = 0; k < 3; k++)
			{
				const double c = ctr[(size_t) i * 3 + k], ext = hi[k] - lo[k];
				if(!std::isfinite(c)) { ok = false; break; }
				const double u = ext > 0 ? (c - lo[k]) / ext : 0.0;
				code |= spread((uint64_t) std::min(2097151.0, std::max(0.0, u * 2097151.0))) << k;
			}
			key[i] = ok ? code : ~0ull;
			order[i] = i;
		}
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
	}
	tris.assign((size_t) nt * 3 + 3, skr_f4{0.0f, 0.0f, 0.0f, 0.0f}); // + one pad triangle (kernel prefetch)
	for(int i = 0; i < nt; i++)
	{
		const float *t = &raw_triangles[(size_t) order[i] * 9];
		tris[3 * i] = {t[0], t[1], t[2], 0.0f};
		float file_index;
		const int32_t fi = order[i];
		memcpy(&file_index, &fi, 4);
		tris[3 * i + 1] = {t[3] - t[0], t[4] - t[1], t[5] - t[2], file_index};
		tris[3 * i + 2] = {t[6] - t[0], t[7] - t[1], t[8] - t[2], 0.0f};
	}
	tri_order = order;
	build_triangle_materials();
	build_triangle_chunks();
}

void skr_scene::build_triangle_materials()
{
	const int nt = info.n_triangles;
	tri_mats.assign((size_t) nt * 3, skr_f4{0.0f, 0.0f, 0.0f, 0.0f});
	for(int i = 0; i < nt; i++)
	{
		const Material dflt;
		const size_t at = (size_t) tri_order[i] * 10;
		const bool have = at + 10 <= raw_triangle_materials.size();
		const float *m = have ? &raw_triangle_materials[at] : dflt.ambient;
		const float power = have ? m[9] : dflt.power;
		const float *kd = have ? m + 3 : dflt.diffuse, *ks = have ? m + 6 : dflt.specular;
		tri_mats[3 * i] = {info.ambient[0] * m[0], info.ambient[1] * m[1], info.ambient[2] * m[2], power};
		tri_mats[3 * i + 1] = {kd[0], kd[1], kd[2], 0.0f};
		tri_mats[3 * i + 2] = {ks[0], ks[1], ks[2], 0.0f};
	}
}

// Exact-preserving culling data for the triangle walk (DESIGN.md "Triangle chunks").
//
// In exact arithmetic utils.h:181-213 accepts the line o + t d iff it meets the plane of the triangle in
// X = v0 + a e1 + b e2 with a in [-1, 0] (the reference's u carries a flipped sign), b >= 0, b - a <= 1,
// i.e. inside the triangle (v0, v0 - e1, v0 + e2), and |det| >= 1e-5.  In binary32 the computed u, v differ
// from the exact ones by at most
//     eta_u <= 7 eps |d| |e2| (|T| + 1.01 |e1|) / 0.99e-5,   eta_v <= 7 eps |d| |e1| (|T| + 1.01 |e2|) / 0.99e-5
// (eps = 2^-24; three-term dot products and 2x2 cross terms of rounded inputs, divided by a determinant that
// the test itself bounds away from zero), so an accepted line passes within eta_u |e1| + eta_v |e2| of that
// triangle.  Every chunk gets a sphere around its triangles' accept regions, inflated by 16x that slack
// (evaluated for |d| <= d_max, one set of spheres per entry of SKR_CULL_DMAX_LIST, and the farthest possible ray origin: camera or any sphere surface) plus
// an absolute term for the rounding of the device's own line-sphere test.  Where the slack is not small the
// radius becomes infinite and the chunk is simply never culled.
void skr_scene::build_triangle_chunks()
{
	tri_chunks.clear();
	tri_any_cone = false;
	if(info.n_triangles == 0) return;
	const double dmax[SKR_CULL_LEVELS] = SKR_CULL_DMAX_LIST;
	for(int level = 0; level < SKR_CULL_LEVELS; level++)
	{
		std::vector<skr_f4> one;
		build_triangle_chunk_level(dmax[level], one);
		tri_chunk_stride = one.size();
		tri_chunks.insert(tri_chunks.end(), one.begin(), one.end());
	}
}

// Grazing rays are what makes the slack above large: |det| = |d . (e1 x e2)| may be as small as 1e-5.  Where a
// chunk's triangles are (nearly) coplanar — unit normals within an angle alpha of an axis a — every ray with
// |d . a| >= kappa |d| meets all of them at |d^ . n^_i| >= s = kappa cos(alpha) - sin(alpha) > 0, so that
// |det_i| >= |d| |e1_i x e2_i| s and
//     eta_u <= 7 eps |e2| (|T| + 1.01 |e1|) / (0.99 |e1 x e2| s)      (no |d|, no 1e-5),
// which gives such rays a second, "tight" radius that is finite for triangles of any size (test.scn's wall: eta
// ~ 2e-2 with kappa = 1e-3 against ~ 7 for the general bound).  The device picks per lane: tight where
// (d . a / kappa)^2 >= d . d, the general radius otherwise.  kappa = max(1e-3, 4 sin(alpha)); cones wider than
// kappa = 0.5 and chunks with sliver triangles get no tight radius.  Nodes merge their children's cones:
// kappa_p >= (kappa_c + sin(beta_c)) / cos(beta_c), beta_c the angle between the axes.
namespace {
struct Cone { // unit axis, kappa; valid = a tight radius exists
	double ax, ay, az, kappa;
	bool valid;
};
struct Ball {
	double x, y, z, r_loose, r_tight;
	bool unbounded; // loose radius infinite
	Cone cone;
};
const double KAPPA_MIN = 1e-3, KAPPA_MAX = 0.5;
const double KAPPA_DEVICE_ROOM = 1e-3; // the device's binary32 (d . a / kappa)^2 >= d . d may call a ray 1e-3 (relative) short of kappa non-grazing

float round_up_square(double r, bool infinite)
{
	const double r2 = r * r;
	float f = (float) r2;
	if((double) f < r2) f = std::nextafterf(f, INFINITY);
	if(infinite || !(r2 == r2)) f = INFINITY;
	return f;
}
void ball_entries(const Ball &b, skr_f4 &A, skr_f4 &B)
{
	A = {(float) b.x, (float) b.y, (float) b.z, round_up_square(b.r_loose, b.unbounded)};
	if(b.cone.valid)
	{
		const double inv = 1.0 / b.cone.kappa;
		B = {(float) (b.cone.ax * inv), (float) (b.cone.ay * inv), (float) (b.cone.az * inv), round_up_square(b.r_tight, false)};
		if(!(B.w < A.w)) B = {0.0f, 0.0f, 0.0f, A.w}; // the tight radius must be a gain
	}
	else B = {0.0f, 0.0f, 0.0f, A.w};
}
} // namespace

void skr_scene::build_triangle_chunk_level(double d_max, std::vector<skr_f4> &out)
{
	const int nt = info.n_triangles;
	const double eps = 5.9604644775390625e-08; // 2^-24
	auto norm = [](double x, double y, double z) { return std::sqrt(x * x + y * y + z * z); };
	// where rays can start: the camera, or on a sphere (GI children, raytrace.h:128)
	struct Org { double x, y, z, r; };
	std::vector<Org> orgs;
	orgs.push_back({info.camera[0], info.camera[1], info.camera[2], 1e-3});
	for(int i = 0; i < info.n_spheres; i++)
	{
		const float *s = &raw_spheres[(size_t) i * 14];
		orgs.push_back({s[0], s[1], s[2], std::fabs((double) s[3]) + 1e-3});
	}
	tri_chunk_size = info.n_spheres == 0 ? SKR_TRI_CHUNK_COHERENT : SKR_TRI_CHUNK_MIXED; // see tri_chunks.h
	if(const char *e = getenv("SKR_TRI_CHUNK")) // tuning runs only
		if(atoi(e) >= 1 && atoi(e) <= 64) tri_chunk_size = atoi(e);
	const int nc = (nt + tri_chunk_size - 1) / tri_chunk_size;
	std::vector<std::vector<Ball>> levels(1);
	for(int c = 0; c < nc; c++)
	{
		const int i0 = c * tri_chunk_size, i1 = std::min(nt, i0 + tri_chunk_size);
		Ball b{0, 0, 0, 0, 0, false, {0, 0, 0, 0, false}};
		int np = 0;
		std::vector<double> pts;
		double slack = 0, mag = 0;
		// pass 1: geometry, the general slack, the normal cone
		struct Tri { double l1, l2, area2, tmax, nx, ny, nz; };
		std::vector<Tri> tr;
		bool cone_ok = true;
		double sx = 0, sy = 0, sz = 0;
		for(int i = i0; i < i1; i++)
		{
			const skr_f4 v0 = tris[3 * i], e1 = tris[3 * i + 1], e2 = tris[3 * i + 2];
			const double P[3][3] = {{v0.x, v0.y, v0.z}, {(double) v0.x - e1.x, (double) v0.y - e1.y, (double) v0.z - e1.z},
									{(double) v0.x + e2.x, (double) v0.y + e2.y, (double) v0.z + e2.z}};
			for(auto &q : P)
			{
				pts.insert(pts.end(), q, q + 3);
				b.x += q[0]; b.y += q[1]; b.z += q[2];
				np++;
				mag = std::max(mag, std::max(std::fabs(q[0]), std::max(std::fabs(q[1]), std::fabs(q[2]))));
			}
			double tmax = 0;
			for(const Org &o : orgs) tmax = std::max(tmax, norm(o.x - v0.x, o.y - v0.y, o.z - v0.z) + o.r);
			const double l1 = norm(e1.x, e1.y, e1.z), l2 = norm(e2.x, e2.y, e2.z);
			const double eta_u = 7 * eps * d_max * l2 * (tmax + 1.01 * l1) / 0.99e-5;
			const double eta_v = 7 * eps * d_max * l1 * (tmax + 1.01 * l2) / 0.99e-5;
			const double rho_det = 7 * eps * d_max * l1 * l2 / 1e-5; // relative error of the computed determinant at the 1e-5 threshold
			if(!(eta_u < 0.25) || !(eta_v < 0.25) || !(rho_det < 0.01)) b.unbounded = true;
			slack = std::max(slack, 16 * (eta_u * l1 + eta_v * l2));
			mag = std::max(mag, tmax);
			// e1 x e2
			const double mx = (double) e1.y * e2.z - (double) e1.z * e2.y, my = (double) e1.z * e2.x - (double) e1.x * e2.z,
						 mz = (double) e1.x * e2.y - (double) e1.y * e2.x;
			const double area2 = norm(mx, my, mz);
			Tri t{l1, l2, area2, tmax, 0, 0, 0};
			if(!(area2 > 1e-3 * l1 * l2) || !(l1 * l2 > 0)) cone_ok = false; // sliver or degenerate: its determinant is noise
			else
			{
				t.nx = mx / area2; t.ny = my / area2; t.nz = mz / area2;
				if(!tr.empty() && t.nx * tr[0].nx + t.ny * tr[0].ny + t.nz * tr[0].nz < 0) { t.nx = -t.nx; t.ny = -t.ny; t.nz = -t.nz; } // |d . n| has no orientation
				sx += t.nx; sy += t.ny; sz += t.nz;
			}
			tr.push_back(t);
		}
		b.x /= np; b.y /= np; b.z /= np;
		double rad = 0;
		for(int k = 0; k < np; k++) rad = std::max(rad, norm(pts[3 * k] - b.x, pts[3 * k + 1] - b.y, pts[3 * k + 2] - b.z));
		b.r_loose = (rad + slack) * (1 + 1e-4) + 1e-5 * (1 + mag); // + relative and absolute room for the device-side test's own rounding
		// pass 2: the tight radius of non-grazing rays
		const double sl = norm(sx, sy, sz);
		if(cone_ok && sl > 0)
		{
			Cone cn{sx / sl, sy / sl, sz / sl, 0, false};
			double cosa = 1;
			for(const Tri &t : tr) cosa = std::min(cosa, std::fabs(t.nx * cn.ax + t.ny * cn.ay + t.nz * cn.az));
			cosa = std::max(0.0, cosa - 1e-12);
			const double sina = std::sqrt(std::max(0.0, 1 - cosa * cosa));
			cn.kappa = std::max(KAPPA_MIN, 4 * sina) * 1.001;
			const double s = (cn.kappa * (1 - KAPPA_DEVICE_ROOM) * cosa - sina) * 0.99;
			if(cn.kappa <= KAPPA_MAX && s > 0)
			{
				double tight = 0;
				bool ok = true;
				for(const Tri &t : tr)
				{
					const double det_min = 0.99 * t.area2 * s; // per unit |d|
					const double eta_u = 7 * eps * t.l2 * (t.tmax + 1.01 * t.l1) / det_min, eta_v = 7 * eps * t.l1 * (t.tmax + 1.01 * t.l2) / det_min;
					const double rho_det = 7 * eps * t.l1 * t.l2 / (t.area2 * s);
					if(!(eta_u < 0.25) || !(eta_v < 0.25) || !(rho_det < 0.01)) ok = false;
					tight = std::max(tight, 16 * (eta_u * t.l1 + eta_v * t.l2));
				}
				if(ok)
				{
					cn.valid = true;
					b.cone = cn;
					b.r_tight = (rad + tight) * (1 + 1e-4) + 1e-5 * (1 + mag);
				}
			}
		}
		levels[0].push_back(b);
	}
	// Upper levels: one ball around every SKR_TRI_SUPER consecutive balls of the level below (a line that touches a
	// child's sphere touches this one — for the general radii, and for the tight ones under the merged cone), until
	// a single root is left.  The levels above the chunks are laid out depth-first with skip links — node =
	// {centre, R^2} {axis / kappa, R_tight^2} {skip, first chunk, chunk count, height} — so that the device walks
	// them with one wave-uniform index and no stack: touched -> next entry, missed -> entry [skip]; a node of
	// height 1 runs over its (contiguous) chunk entries in a tight loop.  The chunk entries ({centre, R^2} {axis /
	// kappa, R_tight^2}) follow the nodes in the same array.
	do
	{
		const std::vector<Ball> &lo = levels.back();
		std::vector<Ball> up;
		for(size_t c0 = 0; c0 < lo.size(); c0 += SKR_TRI_SUPER)
		{
			const size_t c1 = std::min(lo.size(), c0 + SKR_TRI_SUPER);
			Ball b{0, 0, 0, 0, 0, false, {0, 0, 0, 0, false}};
			bool cones = true;
			double sx = 0, sy = 0, sz = 0;
			for(size_t c = c0; c < c1; c++)
			{
				b.x += lo[c].x; b.y += lo[c].y; b.z += lo[c].z;
				b.unbounded = b.unbounded || lo[c].unbounded;
				cones = cones && lo[c].cone.valid;
				if(cones)
				{
					const Cone &k = lo[c].cone, &k0 = lo[c0].cone;
					const double sgn = (k.ax * k0.ax + k.ay * k0.ay + k.az * k0.az) < 0 ? -1.0 : 1.0;
					sx += sgn * k.ax; sy += sgn * k.ay; sz += sgn * k.az;
				}
			}
			b.x /= (double) (c1 - c0); b.y /= (double) (c1 - c0); b.z /= (double) (c1 - c0);
			double mag = 0;
			for(size_t c = c0; c < c1; c++)
			{
				const double off = norm(lo[c].x - b.x, lo[c].y - b.y, lo[c].z - b.z);
				b.r_loose = std::max(b.r_loose, off + lo[c].r_loose * (1 + 1e-6));
				if(cones) b.r_tight = std::max(b.r_tight, off + lo[c].r_tight * (1 + 1e-6));
				mag = std::max(mag, std::max(std::fabs(lo[c].x), std::max(std::fabs(lo[c].y), std::fabs(lo[c].z))) + lo[c].r_loose);
			}
			if(!(mag < 1e300)) mag = 0; // unbounded children: the general radius is infinite anyway
			b.r_loose = b.r_loose * (1 + 1e-4) + 1e-5 * (1 + mag); // room for the float centre and the device-side test's own rounding
			const double sl = norm(sx, sy, sz);
			if(cones && sl > 0)
			{
				Cone cn{sx / sl, sy / sl, sz / sl, 0, true};
				for(size_t c = c0; c < c1 && cn.valid; c++)
				{
					const Cone &k = lo[c].cone;
					const double cosb = std::max(0.0, std::fabs(k.ax * cn.ax + k.ay * cn.ay + k.az * cn.az) - 1e-12);
					const double sinb = std::sqrt(std::max(0.0, 1 - cosb * cosb));
					// a ray the device calls non-grazing here has |d^ . a_p| >= kappa_p (1 - room); it must be non-grazing
					// for the child in the child's own sense, |d^ . a_c| >= kappa_c
					if(!(cosb > 0.5)) cn.valid = false;
					else cn.kappa = std::max(cn.kappa, (k.kappa + sinb) / cosb / (1 - KAPPA_DEVICE_ROOM) * 1.001);
				}
				if(cn.valid && cn.kappa <= KAPPA_MAX)
				{
					b.cone = cn;
					double mt = 0;
					for(size_t c = c0; c < c1; c++) mt = std::max(mt, std::max(std::fabs(lo[c].x), std::max(std::fabs(lo[c].y), std::fabs(lo[c].z))) + lo[c].r_tight);
					b.r_tight = b.r_tight * (1 + 1e-4) + 1e-5 * (1 + mt);
				}
			}
			up.push_back(b);
		}
		levels.push_back(up);
	} while(levels.back().size() > 1);
	std::vector<skr_f4> nodes;
	auto f4i = [](int32_t a, int32_t b, int32_t c, int32_t d) {
		skr_f4 v;
		memcpy(&v.x, &a, 4); memcpy(&v.y, &b, 4); memcpy(&v.z, &c, 4); memcpy(&v.w, &d, 4);
		return v;
	};
	struct Emit {
		const std::vector<std::vector<Ball>> &levels;
		std::vector<skr_f4> &nodes;
		decltype(f4i) &pack;
		void run(int level, size_t idx)
		{
			const size_t me = nodes.size();
			skr_f4 A, B;
			ball_entries(levels[level][idx], A, B);
			nodes.push_back(A);
			nodes.push_back(B);
			nodes.push_back({0, 0, 0, 0});
			int first = 0, count = 0;
			if(level == 1)
			{
				first = (int) idx * SKR_TRI_SUPER;
				count = (int) std::min(levels[0].size(), (size_t) first + SKR_TRI_SUPER) - first;
			}
			else
			{
				const size_t c0 = idx * SKR_TRI_SUPER, c1 = std::min(levels[level - 1].size(), c0 + SKR_TRI_SUPER);
				for(size_t c = c0; c < c1; c++) run(level - 1, c);
			}
			nodes[me + 2] = pack((int32_t) (nodes.size() / 3), first, count, level);
		}
	};
	Emit emit{levels, nodes, f4i};
	emit.run((int) levels.size() - 1, 0);
	// one pad node so that the walk may prefetch past the end, then the chunk entries (+ a pad entry)
	const int32_t n_nodes = (int32_t) (nodes.size() / 3);
	nodes.push_back({0.0f, 0.0f, 0.0f, INFINITY});
	nodes.push_back({0.0f, 0.0f, 0.0f, INFINITY});
	nodes.push_back(f4i(n_nodes + 1, 0, 0, 0));
	size_t with_cone = 0;
	for(const Ball &b : levels[0])
	{
		skr_f4 A, B;
		ball_entries(b, A, B);
		nodes.push_back(A);
		nodes.push_back(B);
		if(B.w < A.w) with_cone++;
	}
	// the cone test costs every sphere test ~8 instructions (dragon: +8 % with one planar chunk in 1251): the walk
	// only compiles it in where at least a quarter of the chunks gain a tighter radius from it
	const bool any_cone = 4 * with_cone >= levels[0].size();
	nodes.push_back({0.0f, 0.0f, 0.0f, INFINITY});
	nodes.push_back({0.0f, 0.0f, 0.0f, INFINITY});
	tri_node_count = n_nodes;
	tri_any_cone = tri_any_cone || any_cone;
	out.swap(nodes);
}

static void set_camera(skr_scene_info &info, const float p[3], const float d[3], const float u[3], float ha)
{
	for(int k = 0; k < 3; k++)
	{
		info.camera[k] = p[k];
		info.camera[3 + k] = d[k]; // scene.cpp:92-93 discard normalize(): file magnitudes are kept
		info.camera[6 + k] = u[k];
	}
	// camera.h:30: right = cross(direction * -1.0f, up), glm::cross operand order
	const float nx = d[0] * -1.0f, ny = d[1] * -1.0f, nz = d[2] * -1.0f;
	info.camera[9] = ny * u[2] - u[1] * nz;
	info.camera[10] = nz * u[0] - u[2] * nx;
	info.camera[11] = nx * u[1] - u[0] * ny;
	info.camera[12] = ha;
}

int skr_parse_scn(const std::string &path, bool echo, bool strict, skr_scene &sc)
{
	FILE *fp = fopen(path.c_str(), "r");
	if(!fp)
	{
		skr_set_error("Can't open file '%s'", path.c_str()); // scene.cpp:24
		return SKR_ERR_IO;
	}
	sc = skr_scene();
	sc.strict = strict;
	skr_scene_info &info = sc.info;
	info.film_width = 1920; // scene.h:15
	info.film_height = 1080;
	info.max_depth_parsed = 1; // scene.h:26
	Material mat;
	std::vector<float> verts;

	char line[1024]; // scene.cpp:19: lines are read in 1024-byte pieces
	while(fgets(line, sizeof line, fp))
	{
		if(line[0] == '#')
		{
			if(echo) printf("Skipping comment: %s\n", line);
			continue;
		}
		// first whitespace-delimited token = command
		const char *p = line;
		while(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f') p++;
		if(!*p) continue;
		const char *q = p;
		while(*q && !(*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n' || *q == '\v' || *q == '\f')) q++;
		const std::string cmd(p, q);
		const char *args = q;

		if(cmd == "sphere")
		{
			float v[4] = {0, 0, 0, 0};
			read_floats(args, v, 4);
			if(echo) printf("Sphere as position (%f, %f, %f) with radius %f\n", v[0], v[1], v[2], v[3]);
			const float rec[14] = {v[0], v[1], v[2], v[3], mat.ambient[0], mat.ambient[1], mat.ambient[2],
								   mat.diffuse[0], mat.diffuse[1], mat.diffuse[2], mat.specular[0], mat.specular[1], mat.specular[2], mat.power};
			sc.raw_spheres.insert(sc.raw_spheres.end(), rec, rec + 14);
			sc.raw_sphere_ior.push_back(mat.ior);
			info.n_spheres++;
		}
		else if(cmd == "vertex")
		{
			float v[3] = {0, 0, 0};
			read_floats(args, v, 3);
			verts.insert(verts.end(), v, v + 3);
			info.n_vertices++;
		}
		else if(cmd == "triangle")
		{
			float v[3] = {0, 0, 0}; // scene.cpp:69-70: indices are read as floats
			read_floats(args, v, 3);
			// the float is range-checked BEFORE the conversion: (long) of nan, inf or 1e39 is undefined behaviour
			long idx[3] = {0, 0, 0};
			bool ok = true;
			for(int k = 0; k < 3; k++)
			{
				ok = ok && std::isfinite(v[k]) && v[k] > -1.0f && v[k] < (float) info.n_vertices; // (long) truncates: -0.5 -> 0
				if(ok) idx[k] = (long) v[k];
			}
			for(long i : idx) ok = ok && i >= 0 && i < info.n_vertices;
			if(!ok)
			{ // the reference reads out of bounds here; no shipped scene does
				fprintf(stderr, "WARNING. triangle references a vertex outside the pool (%d vertices so far): skipped\n", info.n_vertices);
				info.n_bad_triangles++;
				continue;
			}
			for(long i : idx) sc.raw_triangles.insert(sc.raw_triangles.end(), &verts[(size_t) i * 3], &verts[(size_t) i * 3] + 3);
			{
				const float rec[10] = {mat.ambient[0], mat.ambient[1], mat.ambient[2], mat.diffuse[0], mat.diffuse[1], mat.diffuse[2],
									   mat.specular[0], mat.specular[1], mat.specular[2], mat.power};
				sc.raw_triangle_materials.insert(sc.raw_triangle_materials.end(), rec, rec + 10);
			}
			info.n_triangles++;
		}
		else if(cmd == "camera")
		{
			float v[10] = {0};
			read_floats(args, v, 10);
			if(echo)
				printf("Camera with position (%f, %f, %f) with viewing direction (%f, %f, %f), up glm::vec3 (%f, %f, %f), and halfHeightAngle %f\n",
					   v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);
			set_camera(info, v, v + 3, v + 6, v[9]);
			// (the reference also drops a "simplesphere.txt" camera dump into the CWD here,
			//  scene.cpp:96-102 — a debugging leftover that is deliberately not reproduced)
		}
		else if(cmd == "film_resolution")
		{
			int w = info.film_width, h = info.film_height;
			sscanf(args, "%d %d", &w, &h);
			info.film_width = w;
			info.film_height = h;
			if(echo) printf("Film resolution: %d x %d\n", w, h);
		}
		else if(cmd == "background")
		{
			float v[3] = {0, 0, 0};
			read_floats(args, v, 3);
			if(echo) printf("Background color of (%f,%f,%f)\n", v[0], v[1], v[2]);
			memcpy(info.background, v, sizeof v);
		}
		else if(cmd == "material")
		{
			float v[14] = {0};
			read_floats(args, v, 14);
			if(echo)
				printf("material properties with ambient colour (%f, %f, %f), diffuse colour (%f, %f, %f), specular colour (%f, %f, %f), phong Cosine power %f, transmissive colour (%f, %f, %f), index of refraction %f\n",
					   v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13]);
			memcpy(mat.ambient, v, 12);
			memcpy(mat.diffuse, v + 3, 12);
			memcpy(mat.specular, v + 6, 12);
			mat.power = v[9]; // the transmissive colour feeds only dead code (raytrace.h:45-103)
			mat.ior = v[13];  // so does this one: kept for --legacy-reflect
		}
		else if(cmd == "directional_light")
		{
			float v[6] = {0};
			read_floats(args, v, 6);
			if(echo) printf("directional light colour (%f, %f, %f), direction (%f, %f, %f)\n", v[0], v[1], v[2], v[3], v[4], v[5]);
			if(!strict) info.n_directional_dropped++; // scene.cpp:157-163: built, never pushed
			else
			{ // --strict-scn: pushed, with the clamp of scene.cpp:143-154
				for(int k = 0; k < 3; k++)
					if(v[k] > 1) v[k] = 1;
				const float rec[6] = {v[3], v[4], v[5], v[0], v[1], v[2]}; // file order is colour then direction
				sc.raw_directional_lights.insert(sc.raw_directional_lights.end(), rec, rec + 6);
				info.n_directional_lights++;
			}
		}
		else if(cmd == "point_light")
		{
			float v[6] = {0};
			read_floats(args, v, 6);
			if(echo) printf("point light colour (%f, %f, %f), located at (%f, %f, %f)\n", v[0], v[1], v[2], v[3], v[4], v[5]);
			const float rec[6] = {v[3], v[4], v[5], v[0], v[1], v[2]}; // file order is colour then position
			sc.raw_point_lights.insert(sc.raw_point_lights.end(), rec, rec + 6);
			info.n_point_lights++;
		}
		else if(cmd == "ambient_light")
		{
			float v[3] = {0, 0, 0};
			read_floats(args, v, 3);
			if(echo) printf("Ambient light colour (%f, %f, %f)\n", v[0], v[1], v[2]);
			for(int k = 0; k < 3; k++) info.ambient[k] += v[k]; // scene.cpp:187-189 accumulates
		}
		else if(cmd == "max_depth")
		{
			float n = 0;
			read_floats(args, &n, 1);
			if(echo) printf("max_depth %f\n", n);
			info.max_depth_parsed = (std::isfinite(n) && n > -2147483648.0f && n < 2147483648.0f) ? (int) n : 0; // (int) of nan / 1e30 is undefined
		}
		else if(cmd == "output_image")
		{
			if(echo) printf("Render to file named: %s", args + (*args ? 1 : 0));
		}
		else if(cmd == "spherical_fog")
		{
			// scene.cpp:207-212 pushes a fog volume built from uninitialised floats
			// (sscanf "fog ..." never matches): undefined behaviour, pinned as "ignored".
			fprintf(stderr, "WARNING. spherical_fog is not reproducible in the reference (uninitialised data): line skipped\n");
			info.n_fog_skipped++;
		}
		else
		{
			if(echo) printf("WARNING. Do not know command: %s\n", cmd.c_str());
			info.n_unknown++;
		}
	}
	fclose(fp);
	sc.finalize();
	return SKR_OK;
}

extern "C" {

int skr_scene_create_from_scn(const char *path, int echo, skr_scene **out) { return skr_scene_create_from_scn_ex(path, echo, 0, out); }

int skr_scene_create_from_scn_ex(const char *path, int echo, uint32_t flags, skr_scene **out)
{
	if(!path || !out)
	{
		skr_set_error("skr_scene_create_from_scn: null argument");
		return SKR_ERR_ARG;
	}
	skr_scene *sc = new skr_scene();
	int rc = skr_parse_scn(path, echo != 0, (flags & SKR_SCN_STRICT) != 0, *sc);
	if(rc != SKR_OK)
	{
		delete sc;
		*out = nullptr;
		return rc;
	}
	*out = sc;
	return SKR_OK;
}

int skr_scene_create_from_arrays(const float *spheres, int32_t n_spheres, const float *triangles, int32_t n_triangles,
								 const float *point_lights, int32_t n_point_lights, const float camera[9],
								 const float background[3], const float ambient[3], skr_scene **out)
{
	if(!out || !camera || n_spheres < 0 || n_triangles < 0 || n_point_lights < 0 || (n_spheres && !spheres) ||
	   (n_triangles && !triangles) || (n_point_lights && !point_lights))
	{
		skr_set_error("skr_scene_create_from_arrays: bad argument");
		return SKR_ERR_ARG;
	}
	skr_scene *sc = new skr_scene();
	sc->info.film_width = 1920;
	sc->info.film_height = 1080;
	sc->info.max_depth_parsed = 1;
	sc->info.n_spheres = n_spheres;
	sc->info.n_triangles = n_triangles;
	sc->info.n_point_lights = n_point_lights;
	sc->raw_spheres.assign(spheres, spheres + (size_t) n_spheres * 14);
	sc->raw_triangles.assign(triangles, triangles + (size_t) n_triangles * 9);
	sc->raw_point_lights.assign(point_lights, point_lights + (size_t) n_point_lights * 6);
	set_camera(sc->info, camera, camera + 3, camera + 6, 0.0f);
	if(background) memcpy(sc->info.background, background, 12);
	if(ambient) memcpy(sc->info.ambient, ambient, 12);
	sc->finalize();
	*out = sc;
	return SKR_OK;
}

int skr_scene_set_triangle_materials(skr_scene *scene, const float *materials)
{
	if(!scene || (scene->info.n_triangles && !materials))
	{
		skr_set_error("skr_scene_set_triangle_materials: bad argument");
		return SKR_ERR_ARG;
	}
	scene->raw_triangle_materials.assign(materials, materials + (size_t) scene->info.n_triangles * 10);
	scene->build_triangle_materials();
	return SKR_OK;
}

int skr_scene_set_sphere_ior(skr_scene *scene, const float *ior)
{
	if(!scene || (scene->info.n_spheres && !ior))
	{
		skr_set_error("skr_scene_set_sphere_ior: bad argument");
		return SKR_ERR_ARG;
	}
	scene->raw_sphere_ior.assign(ior, ior + scene->info.n_spheres);
	for(int i = 0; i < scene->info.n_spheres; i++) scene->sph_ks[i].w = ior[i];
	return SKR_OK;
}

void skr_scene_destroy(skr_scene *scene) { delete scene; }

int skr_scene_get_info(const skr_scene *scene, skr_scene_info *info)
{
	if(!scene || !info) return SKR_ERR_ARG;
	*info = scene->info;
	return SKR_OK;
}

int skr_scene_get_arrays(const skr_scene *scene, float *spheres, float *triangles, float *point_lights)
{
	if(!scene) return SKR_ERR_ARG;
	if(spheres && !scene->raw_spheres.empty()) memcpy(spheres, scene->raw_spheres.data(), scene->raw_spheres.size() * 4);
	if(triangles && !scene->raw_triangles.empty()) memcpy(triangles, scene->raw_triangles.data(), scene->raw_triangles.size() * 4);
	if(point_lights && !scene->raw_point_lights.empty()) memcpy(point_lights, scene->raw_point_lights.data(), scene->raw_point_lights.size() * 4);
	return SKR_OK;
}

int skr_scene_get_culling(const skr_scene *scene, int32_t level, int32_t *chunk_size, int32_t *n_nodes, int32_t *n_chunks,
						  float *device_tris, float *node_spheres, int32_t *node_links, float *chunk_spheres)
{
	if(!scene || level < 0 || level >= SKR_CULL_LEVELS) return SKR_ERR_ARG;
	const int nt = scene->info.n_triangles, nn = nt ? scene->tri_node_count : 0, cs = scene->tri_chunk_size;
	const int nc = nt ? (nt + cs - 1) / cs : 0;
	if(chunk_size) *chunk_size = cs;
	if(n_nodes) *n_nodes = nn;
	if(n_chunks) *n_chunks = nc;
	if(device_tris && nt) memcpy(device_tris, scene->tris.data(), (size_t) nt * 48);
	const skr_f4 *base = scene->tri_chunks.data() + (size_t) level * scene->tri_chunk_stride;
	for(int i = 0; i < nn; i++)
	{
		if(node_spheres) memcpy(node_spheres + 8 * (size_t) i, base + 3 * (size_t) i, 32);
		if(node_links) memcpy(node_links + 4 * (size_t) i, base + 3 * (size_t) i + 2, 16);
	}
	if(chunk_spheres && nc) memcpy(chunk_spheres, base + 3 * ((size_t) nn + 1), (size_t) nc * 32);
	return SKR_OK;
}

void skr_options_default(skr_options *opt)
{
	if(!opt) return;
	opt->width = 1920; // scene.h:15
	opt->height = 1080;
	opt->fov = 60; // utils.h:28-33
	opt->monte_carlo = 0;
	opt->num_path_traces = 1;
	opt->grid_size = 0;
	opt->max_depth = 3;
	opt->use_shadows = 0;
	opt->seed = 1;
	opt->shade_triangles = 0;
	opt->progressive_passes = 1;
	opt->legacy_reflect = 0;
}

uint64_t skr_radiance_ray_count(const skr_options *opt)
{
	if(!opt || opt->width <= 0 || opt->height <= 0) return 0;
	const uint64_t S = opt->grid_size > 0 ? (uint64_t) opt->grid_size * opt->grid_size : 1;
	uint64_t per = 0, pw = 1;
	for(int k = 0; k < opt->max_depth; k++)
	{
		per += pw;
		if(!opt->monte_carlo) break;
		pw *= (uint64_t) (opt->num_path_traces > 0 ? opt->num_path_traces : 0);
	}
	const uint64_t K = opt->progressive_passes > 1 ? (uint64_t) opt->progressive_passes : 1; // every pass is a whole frame
	return (uint64_t) opt->width * opt->height * S * per * K;
}

// main.cpp:199-211: "P6\n" W " " H "\n255\n" then W*H*3 bytes, top row first.
int skr_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb)
{
	if(!path || !rgb) return SKR_ERR_ARG;
	std::ofstream ofs(path, std::ios::out | std::ios::binary);
	if(!ofs)
	{
		skr_set_error("cannot open '%s' for writing", path);
		return SKR_ERR_IO;
	}
	ofs << "P6\n" << width << " " << height << "\n255\n";
	ofs.write(reinterpret_cast<const char *>(rgb), (std::streamsize) width * height * 3);
	ofs.close();
	return ofs ? SKR_OK : SKR_ERR_IO;
}

// ---- other outputs.  The reference writes P6 only (main.cpp:199-211). ----

// Portable float map: "PF\n<W> <H>\n-1.0\n" (negative scale = little-endian), then W*3 binary32 values per row, BOTTOM row
// first.  Carries the unquantised frame (what main.cpp:205 clamps away): rgbf is top row first, like every buffer here.
int skr_write_pfm(const char *path, uint32_t width, uint32_t height, const float *rgbf)
{
	if(!path || !rgbf) return SKR_ERR_ARG;
	std::ofstream ofs(path, std::ios::out | std::ios::binary);
	if(!ofs)
	{
		skr_set_error("cannot open '%s' for writing", path);
		return SKR_ERR_IO;
	}
	ofs << "PF\n" << width << " " << height << "\n-1.0\n";
	for(uint32_t y = height; y-- > 0;) ofs.write(reinterpret_cast<const char *>(rgbf + (size_t) y * width * 3), (std::streamsize) width * 12);
	ofs.close();
	return ofs ? SKR_OK : SKR_ERR_IO;
}

// PNG, 8-bit RGB, the bytes of the PPM: filter 0 on every row, zlib stream of stored (uncompressed) deflate blocks — no
// compressor is linked; any PNG reader takes it.
int skr_write_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb)
{
	if(!path || !rgb || width == 0 || height == 0) return SKR_ERR_ARG;
	static uint32_t crc_table[256];
	static bool have_table = false;
	if(!have_table)
	{
		for(uint32_t n = 0; n < 256; n++)
		{
			uint32_t c = n;
			for(int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
			crc_table[n] = c;
		}
		have_table = true;
	}
	auto be32 = [](std::vector<uint8_t> &v, uint32_t x) { for(int k = 3; k >= 0; k--) v.push_back((uint8_t) (x >> (8 * k))); };
	std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	auto chunk = [&](const char type[4], const std::vector<uint8_t> &data) {
		be32(out, (uint32_t) data.size());
		const size_t at = out.size();
		out.insert(out.end(), type, type + 4);
		out.insert(out.end(), data.begin(), data.end());
		uint32_t c = 0xffffffffu;
		for(size_t i = at; i < out.size(); i++) c = crc_table[(c ^ out[i]) & 0xffu] ^ (c >> 8);
		be32(out, c ^ 0xffffffffu);
	};
	std::vector<uint8_t> ihdr;
	be32(ihdr, width);
	be32(ihdr, height);
	const uint8_t tail[5] = {8, 2, 0, 0, 0}; // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
	ihdr.insert(ihdr.end(), tail, tail + 5);
	chunk("IHDR", ihdr);
	const size_t row = (size_t) width * 3 + 1, raw_n = row * height;
	if(raw_n > 0x7fffffffull)
	{
		skr_set_error("skr_write_png: image too large for one IDAT chunk");
		return SKR_ERR_ARG;
	}
	std::vector<uint8_t> raw(raw_n);
	for(uint32_t y = 0; y < height; y++)
	{
		raw[row * y] = 0;
		memcpy(&raw[row * y + 1], rgb + (size_t) y * width * 3, (size_t) width * 3);
	}
	std::vector<uint8_t> z = {0x78, 0x01};
	uint32_t a = 1, b = 0; // adler32
	for(size_t at = 0; at < raw_n; at += 65535)
	{
		const size_t len = raw_n - at < 65535 ? raw_n - at : 65535;
		z.push_back(at + len == raw_n ? 1 : 0);
		z.push_back((uint8_t) len);
		z.push_back((uint8_t) (len >> 8));
		z.push_back((uint8_t) ~len);
		z.push_back((uint8_t) (~len >> 8));
		z.insert(z.end(), raw.begin() + (ptrdiff_t) at, raw.begin() + (ptrdiff_t) (at + len));
		for(size_t i = at; i < at + len;)
		{ // 5552 bytes: the longest run before a + b can overflow 32 bits
			const size_t stop = i + 5552 < at + len ? i + 5552 : at + len;
			for(; i < stop; i++) { a += raw[i]; b += a; }
			a %= 65521u;
			b %= 65521u;
		}
	}
	be32(z, (b << 16) | a);
	chunk("IDAT", z);
	chunk("IEND", {});
	std::ofstream ofs(path, std::ios::out | std::ios::binary);
	if(!ofs)
	{
		skr_set_error("cannot open '%s' for writing", path);
		return SKR_ERR_IO;
	}
	ofs.write(reinterpret_cast<const char *>(out.data()), (std::streamsize) out.size());
	ofs.close();
	return ofs ? SKR_OK : SKR_ERR_IO;
}

} // extern "C"
