// raytracer — drop-in command line of the reference (src/main.cpp:230-413):
//   ./raytracer --path scene.scn --output image.ppm [--width i] [--height i] [--fov f]
//               [--gillum n] [--jsample g] [--depth d] [--parallel true|false] [--shadow]
// Flags are matched by strcmp anywhere in argv and unknown tokens are ignored,
// messages and the exit-status-0 convention follow the reference.  New,
// non-colliding flags: --seed u64 (counter-RNG key; the reference seeds rand()
// with time(0)), --device i, --quiet (no per-line scene echo), --gpus N, --strict-scn,
// --shade-triangles, --legacy-reflect, --progressive K [--progressive-every M], --format ppm|png|pfm.
// The frame itself is rendered by libskr on the GPU; there is no CPU path here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "skr.h"

int main(int argc, char *argv[])
{
	skr_options option;
	skr_options_default(&option);
	const char *path = nullptr, *output = nullptr;
	bool visual = true; // utils.h:29; kept for compatibility: there is no SDL viewer, both values render on the GPU
	bool quiet = false;
	int device = 0, gpus = 1;
	bool strict_scn = false, width_given = false, height_given = false, depth_given = false; // --strict-scn (new)
	bool sharded = false; // --gpus given (even --gpus 1): the frame goes through the multi-GPU path
	uint32_t tile_rows = 8;
	uint32_t progressive_every = 0; // --progressive-every M: the output file is rewritten after every M passes (the headless "viewer")
	const char *format = "ppm";     // --format ppm | png | pfm (new; the reference writes P6 whatever the name says)

	for(int i = 0; i < argc; i++)
	{
		const bool has_next = i + 1 < argc;
		if(!strcmp(argv[i], "--gillum"))
		{
			if(has_next)
			{
				option.monte_carlo = 1;
				option.num_path_traces = atoi(argv[i + 1]);
			}
			else std::cerr << "gillum takes an int after flag for the number of paths traced" << std::endl; // main.cpp:258: warns, goes on
		}
		if(!strcmp(argv[i], "--fov"))
		{
			if(!has_next)
			{
				std::cerr << "fov takes a float (degrees) after flag for the field of view" << std::endl;
				return 0;
			}
			option.fov = (float) atof(argv[i + 1]);
		}
		if(!strcmp(argv[i], "--jsample"))
		{
			if(!has_next)
			{
				std::cerr << "jsample takes an int after flag for the supersampling grid size" << std::endl;
				return 0;
			}
			option.grid_size = atoi(argv[i + 1]);
		}
		if(!strcmp(argv[i], "--width"))
		{
			if(!has_next)
			{
				std::cerr << "width takes an int after flag for the width" << std::endl;
				return 0;
			}
			option.width = atoi(argv[i + 1]);
			width_given = true;
		}
		if(!strcmp(argv[i], "--height"))
		{
			if(!has_next)
			{
				std::cerr << "height takes an int after flag for the width" << std::endl;
				return 0;
			}
			option.height = atoi(argv[i + 1]);
			height_given = true;
		}
		if(!strcmp(argv[i], "--depth"))
		{
			if(!has_next || atoi(argv[i + 1]) <= 0)
			{
				std::cerr << "depth takes a positive int after flag for the max depth" << std::endl;
				return 0;
			}
			option.max_depth = atoi(argv[i + 1]);
			depth_given = true;
		}
		if(!strcmp(argv[i], "--parallel"))
		{
			if(has_next && !strcmp(argv[i + 1], "true")) visual = false;
			if(has_next && !strcmp(argv[i + 1], "false")) visual = true;
		}
		if(!strcmp(argv[i], "--path"))
		{
			if(!has_next)
			{
				std::cerr << "path must be passed after --path" << std::endl;
				return 0;
			}
			path = argv[i + 1];
		}
		if(!strcmp(argv[i], "--output"))
		{
			if(!has_next)
			{
				std::cerr << "output path must be passed after --output" << std::endl;
				return 0;
			}
			output = argv[i + 1];
		}
		if(!strcmp(argv[i], "--shadow")) option.use_shadows = 1;
		if(!strcmp(argv[i], "--seed") && has_next) option.seed = strtoull(argv[i + 1], nullptr, 10);
		if(!strcmp(argv[i], "--device") && has_next) device = atoi(argv[i + 1]);
		if(!strcmp(argv[i], "--gpus") && has_next) { gpus = atoi(argv[i + 1]) > 0 ? atoi(argv[i + 1]) : 1; sharded = true; }           // new: the frame sharded over the first N devices
		if(!strcmp(argv[i], "--tile-rows") && has_next) tile_rows = (uint32_t) (atoi(argv[i + 1]) > 0 ? atoi(argv[i + 1]) : 8);
		if(!strcmp(argv[i], "--quiet")) quiet = true;
		if(!strcmp(argv[i], "--strict-scn")) strict_scn = true;
		if(!strcmp(argv[i], "--shade-triangles")) option.shade_triangles = 1; // new: triangles as surfaces (include/skr.h skr_options)
		if(!strcmp(argv[i], "--legacy-reflect")) option.legacy_reflect = 1;   // new: the reflection / refraction code behind raytrace.h:44's early return
		if(!strcmp(argv[i], "--progressive") && has_next) option.progressive_passes = atoi(argv[i + 1]) > 1 ? atoi(argv[i + 1]) : 1; // new: mean of K frames, seeds seed..seed+K-1
		if(!strcmp(argv[i], "--progressive-every") && has_next) progressive_every = (uint32_t) (atoi(argv[i + 1]) > 0 ? atoi(argv[i + 1]) : 0);
		if(!strcmp(argv[i], "--format") && has_next) format = argv[i + 1];
	}
	if(!path)
	{
		std::cerr << "no scene file was passed. Pass with --path path_to_scn" << std::endl;
		return 0;
	}
	if(!output)
	{
		std::cerr << "no output destination was passed. Pass with --output destination_path.ppm" << std::endl;
		return 0;
	}

	skr_scene *scene = nullptr;
	if(skr_scene_create_from_scn_ex(path, quiet ? 0 : 1, strict_scn ? SKR_SCN_STRICT : 0u, &scene) != SKR_OK)
	{
		printf("%s\n", skr_last_error()); // scene.cpp:24-25: message, exit(0)
		return 0;
	}
	if(strict_scn)
	{ // the .scn's own film_resolution (scene.cpp:105-109) and max_depth (:192-198) hold unless the command line says otherwise
	  // (the reference parses both and then overrides / never reads them: main.cpp:393-395, scene.h:26)
		skr_scene_info info;
		skr_scene_get_info(scene, &info);
		if(!width_given && info.film_width > 0) option.width = info.film_width;
		if(!height_given && info.film_height > 0) option.height = info.film_height;
		if(!depth_given && info.max_depth_parsed > 0) option.max_depth = info.max_depth_parsed;
	}
	// utils.h:35-38 Options::to_string
	printf("\n\nMonte carlo: %d\nvisual display: %d\nfov: %f\nnum paths traced: %d\nsupersample grid size: %d\nmax depth: %d\n",
		   option.monte_carlo, visual ? 1 : 0, option.fov, option.num_path_traces, option.grid_size, option.max_depth);

	if(option.width <= 0 || option.height <= 0 || option.width > 65536 || option.height > 65536)
	{ // before anything is sized from it (the reference would allocate width x height vec3s here, main.cpp:125-128)
		std::cerr << "raytracer: bad image size " << option.width << "x" << option.height << std::endl;
		return SKR_ERR_ARG;
	}
	const bool want_pfm = !strcmp(format, "pfm"), want_png = !strcmp(format, "png");
	if(!want_pfm && !want_png && strcmp(format, "ppm"))
	{
		std::cerr << "format takes ppm, png or pfm" << std::endl;
		return 0;
	}
	if(sharded && (want_pfm || progressive_every))
	{ // the ranks exchange quantised tiles (one all-gather of bytes): the float frame and the running mean stay on their devices
		std::cerr << "raytracer: --format pfm and --progressive-every need the single-device path (drop --gpus)" << std::endl;
		return SKR_ERR_ARG;
	}
	std::vector<uint8_t> rgb((size_t) option.width * option.height * 3);
	std::vector<float> rgbf(want_pfm ? rgb.size() : 0);
	struct Out {
		const char *path;
		uint32_t w, h;
		bool pfm, png, quiet;
	} out{output, (uint32_t) option.width, (uint32_t) option.height, want_pfm, want_png, quiet};
	auto write_out = [](const Out &o, const uint8_t *b, const float *f) {
		return o.pfm ? skr_write_pfm(o.path, o.w, o.h, f) : o.png ? skr_write_png(o.path, o.w, o.h, b) : skr_write_ppm(o.path, o.w, o.h, b);
	};
	float ms = 0;
	uint64_t counters[3] = {0, 0, 0};
	int rc;
	skr_renderer *renderer = nullptr;
	skr_multi *multi = nullptr;
	if(sharded)
	{ // one process, N devices: interleaved row tiles, one RCCL all-gather, the root de-interleaves (include/skr.h "multi-GPU")
		rc = skr_multi_create(scene, gpus, nullptr, &multi);
		if(rc == SKR_OK) rc = skr_multi_render_frame_host(multi, &option, tile_rows, rgb.data(), &ms);
		if(rc != SKR_OK)
		{
			std::cerr << "raytracer: " << skr_last_error() << std::endl;
			return rc; // new failure class (the reference has no device): non-zero
		}
		for(int i = 0; i < gpus; i++)
		{
			uint64_t c[3] = {0, 0, 0};
			skr_renderer_read_counters(skr_multi_renderer(multi, i), c, 0);
			for(int k = 0; k < 3; k++) counters[k] += c[k];
		}
	}
	else
	{
		// --progressive-every: the file is the window (main.cpp:183-197 redraws its SDL window as rows finish) — rewritten with the
		// mean so far after every M passes
		struct Show {
			const Out *o;
			int (*write)(const Out &, const uint8_t *, const float *);
		} show{&out, write_out};
		skr_progress_fn progress = [](void *user, uint32_t done, uint32_t passes, const uint8_t *b, const float *f) -> int {
			const Show *s = static_cast<const Show *>(user);
			if(done < passes && s->write(*s->o, b, f) != SKR_OK) return 1;
			if(!s->o->quiet) printf("pass %u of %u\n", done, passes);
			return 0;
		};
		rc = skr_renderer_create(scene, device, &renderer);
		if(rc == SKR_OK)
			rc = skr_render_progressive_host(renderer, &option, progressive_every, want_pfm ? nullptr : rgb.data(), want_pfm ? rgbf.data() : nullptr,
											 progressive_every ? progress : nullptr, &show, &ms);
		if(rc != SKR_OK)
		{
			std::cerr << "raytracer: " << skr_last_error() << std::endl;
			return rc;
		}
		skr_renderer_read_counters(renderer, counters, 0);
	}
	rc = write_out(out, rgb.data(), rgbf.data());
	if(rc != SKR_OK)
	{
		std::cerr << "raytracer: " << skr_last_error() << std::endl;
		return rc;
	}
	printf("***\nWROTE TO PPM\n***\n"); // main.cpp:213
	fprintf(stderr, "{\"gpus\": %d, \"kernel\": \"%s\", \"frame_ms\": %.3f, \"radiance_rays\": %llu, \"mrays_per_s\": %.1f, \"shadow_rays\": %llu}\n",
			gpus > 1 ? gpus : 1, skr_kernel_variant(), ms, (unsigned long long) counters[0], ms > 0 ? counters[0] / (ms * 1e3) : 0.0, (unsigned long long) counters[2]);
	skr_multi_destroy(multi);
	skr_renderer_destroy(renderer);
	skr_scene_destroy(scene);
	return 0;
}
