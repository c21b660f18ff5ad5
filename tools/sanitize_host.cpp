// Host-code sanitizer run (CPU only): loader, finalize, culling builder, array getters, PPM writer.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "skr.h"
int main(int argc, char **argv)
{
	for(int a = 1; a < argc; a++)
	{
		skr_scene *sc = nullptr;
		int rc = skr_scene_create_from_scn(argv[a], 0, &sc);
		if(rc != 0) { printf("%s: rc %d (%s)\n", argv[a], rc, skr_last_error()); continue; }
		skr_scene_info info;
		skr_scene_get_info(sc, &info);
		std::vector<float> s((size_t) info.n_spheres * 14 + 1), t((size_t) info.n_triangles * 9 + 1), l((size_t) info.n_point_lights * 6 + 1);
		skr_scene_get_arrays(sc, s.data(), t.data(), l.data());
		for(int level = 0; level < 3; level++)
		{
			int32_t cs = 0, nn = 0, nc = 0;
			skr_scene_get_culling(sc, level, &cs, &nn, &nc, nullptr, nullptr, nullptr, nullptr);
			std::vector<float> dt((size_t) info.n_triangles * 12 + 1), ns((size_t) nn * 8 + 1), ch((size_t) nc * 8 + 1);
			std::vector<int32_t> nl((size_t) nn * 4 + 1);
			skr_scene_get_culling(sc, level, nullptr, nullptr, nullptr, dt.data(), ns.data(), nl.data(), ch.data());
		}
		printf("%s: %d spheres %d triangles %d lights ok\n", argv[a], info.n_spheres, info.n_triangles, info.n_point_lights);
		skr_scene_destroy(sc);
	}
	std::vector<uint8_t> px(7 * 5 * 3, 200);
	skr_write_ppm("build/asan_out.ppm", 7, 5, px.data());
	return 0;
}
