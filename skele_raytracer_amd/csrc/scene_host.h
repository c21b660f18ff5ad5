// Host-side scene: what the .scn loader produces and what is uploaded to HBM.
// Replaces struct Scene (reference src/scene.h:13-28) and its AoS members
// (shapes.h:12,26  material.h:9  lights.h:19  camera.h:8) with the SoA layout the
// kernels read.  See DESIGN.md "Data layout in HBM".
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/skr.h"
#include "tri_chunks.h"


struct skr_f4 {
	float x, y, z, w;
};

struct skr_scene {
	// raw values as parsed (skr_scene_get_arrays, loader parity tests)
	std::vector<float> raw_spheres;      // [n][14] centre radius ambient diffuse specular power
	std::vector<float> raw_sphere_ior;   // [n] index of refraction (material.h:16); shorter than n = 1.0.  Read by --legacy-reflect only
	std::vector<float> raw_triangles;    // [n][9]  v0 v1 v2
	std::vector<float> raw_triangle_materials; // [n][10] ambient diffuse specular power: the material in force on each `triangle` line
	                                     //         (read by --shade-triangles only; shorter than n = the default material, material.h:9-17)
	std::vector<float> raw_point_lights; // [n][6]  position colour
	std::vector<float> raw_directional_lights; // [n][6] direction colour — --strict-scn only (scene.cpp:139-163 drops them)
	bool strict = false;                 // parsed with SKR_SCN_STRICT
	skr_scene_info info{};

	// SoA arrays as uploaded (built by finalize())
	std::vector<skr_f4> sph_geom; // centre.xyz, radius*radius (utils.h:118 forms r*r per test; same product)
	std::vector<skr_f4> sph_amb;  // ambient_light.colour * material.ambient (blinn_phong.h:15), .w = phong power
	std::vector<skr_f4> sph_kd;   // material.diffuse
	std::vector<skr_f4> sph_ks;   // material.specular, .w = index of refraction
	std::vector<skr_f4> lights;   // [2*i] position (.w = 0) or, behind the point lights, direction (.w = 1: --strict-scn), [2*i+1] colour
	std::vector<skr_f4> tris;     // [3*i] v0, [3*i+1] v1-v0, [3*i+2] v2-v0 (utils.h:183-184 subtractions); [3*i+1].w = the triangle's index in the file (int bits)
	std::vector<skr_f4> tri_mats; // [3*i] La*ka, power  [3*i+1] kd  [3*i+2] ks of the triangle stored at tris[3*i] (--shade-triangles)
	// the culling data of the triangle walk: a tree, depth-first with skip links, three float4 per node — {centre, R^2}
	// {axis / kappa, R_tight^2} {skip, first chunk, chunk count, height} (ints) — + one pad node, then two float4 per
	// chunk of tri_chunk_size consecutive triangles — {centre, R^2} {axis / kappa, R_tight^2} — + one pad entry.
	// Every inner node has SKR_TRI_SUPER children; a sphere is conservative: a line that
	// misses it cannot pass utils.h:181-213 for any triangle of the chunk (see finalize())
	std::vector<skr_f4> tri_chunks; // SKR_CULL_LEVELS sets of them, one per bound on |d| (tri_chunks.h)
	int tri_chunk_size = SKR_TRI_CHUNK_MIXED;

	void finalize();
	size_t tri_chunk_stride = 0; // float4 entries per |d| level
	int tri_node_count = 0;
	bool tri_any_cone = false; // some chunk has a tight radius for non-grazing rays (scene_host.cpp)
	void build_triangle_chunks();
	std::vector<int> tri_order; // tris[3*i] holds triangle tri_order[i] of the file
	void build_triangle_materials();
	void build_triangle_chunk_level(double d_max, std::vector<skr_f4> &out);
};

// scene.cpp:12-227 replacement.  Returns SKR_OK or SKR_ERR_IO.
int skr_parse_scn(const std::string &path, bool echo, bool strict, skr_scene &out);

void skr_set_error(const char *fmt, ...);
