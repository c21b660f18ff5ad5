// oracle/ref_driver.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Harness around the UNMODIFIED reference sources where they lie under
// /root/reference/src (never copied into this repo).  It is compiled by
// oracle/Makefile into oracle/_ref/ref_render together with the reference's
// own scene.cpp; the reference's shade() / bp::* / utils.h / parseScene()
// are therefore the real thing.  Only this container can build it (the GPU
// box has no /root/reference); its outputs are committed as fixtures under
// tests/golden/ by tests/golden/make_golden.py.
//
// Why a driver instead of the reference's main.cpp: main.cpp:8-9 includes
// <SDL.h>/<SDL_opengl.h>, SDL2 is not installed in this image, and writing a
// stand-in SDL header is not allowed.  main.cpp is therefore unbuildable
// here; the part of it that is on the hot path — the per-pixel loop body
// main.cpp:129-182 (== :36-85) and the PPM writer main.cpp:199-211 — is
// restated below and pinned by the reference's one pixel-exact fixture
// (renders/testcpu.ppm, see tests/test_oracle_golden.py).  Everything the
// loop calls is reference code.
//
// Pinned semantics that differ from "run main.cpp as shipped" (SURVEY.md §0):
//  * scene.spherical_fog is cleared after parseScene(): scene.cpp:207-212
//    fills it from uninitialised stack floats (sscanf "fog ..." matches no
//    field), i.e. undefined behaviour.  "fog line ignored" is the only
//    reproducible meaning.
//  * use_shadows is taken from --shadow (main.cpp:244 leaves it
//    uninitialised when the flag is absent; here absent == false).
//  * srand(seed) with an explicit --seed instead of srand(time(0))
//    (main.cpp:400); the loop is serial whenever rand() is consumed so the
//    draw order is the reference's DFS order.
//  * --parallel-entry reproduces generate_rays_parallel's overrides
//    (main.cpp:21-24: 640x480, depth 1, no jitter).
//
// --legacy / --eval-legacy (round 3).  direct_illumination() returns at raytrace.h:44; behind the return sits the
// reflection / refraction / Fresnel code (:45-103), unreachable — so no run of the reference's shade() can exercise it.  What CAN
// be run is every function that code calls: bp::fresnel (blinn_phong.h:156), bp::refraction (:143), bp::reflect_direction (:137)
// are ordinary functions.  --eval-legacy dumps them on 10 000 inputs (the oracle's restatements must give the same bits:
// FUNCTION-LEVEL pin).  --legacy renders with shade_legacy() below: the control flow of raytrace.h:139-227 and :36-103 with the
// early return taken out, restated here, around the reference's own intersection_occurs / collision_distance /
// triangle_intersection_occurs / smallest_root / bp::ambient, diffuse, specular / bp::fresnel, refraction, reflect_direction —
// every number is computed by reference code, only the order of the calls is restated (COMPOSITION restated).  Without --gillum
// (montecarlo_global_illumination calls the reference's own shade(), which would leave the mode).

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <unistd.h>
#include <fcntl.h>

#include "raytrace.h" // the reference's integrator (pulls blinn_phong.h, utils.h, scene.h, glm)

static void write_ppm(const char *path, int w, int h, glm::vec3 *img)
{
	// main.cpp:199-211: "P6\nW H\n255\n", then per channel
	// (unsigned char)(std::min(float(1), c) * 255), row-major, top row first.
	FILE *f = fopen(path, "wb");
	if(!f)
	{
		fprintf(stderr, "ref_render: cannot write %s\n", path);
		exit(2);
	}
	fprintf(f, "P6\n%d %d\n255\n", w, h);
	for(int i = 0; i < w * h; i++)
	{
		unsigned char px[3] = {(unsigned char) (std::min(float(1), img[i].x) * 255),
							   (unsigned char) (std::min(float(1), img[i].y) * 255),
							   (unsigned char) (std::min(float(1), img[i].z) * 255)};
		fwrite(px, 1, 3, f);
	}
	fclose(f);
}

static void hexf(FILE *f, float v)
{
	uint32_t u;
	memcpy(&u, &v, 4);
	fprintf(f, " %08x", u);
}
static void hex3(FILE *f, glm::vec3 v)
{
	hexf(f, v.x);
	hexf(f, v.y);
	hexf(f, v.z);
}

// Dump what parseScene() produced as hex floats, so the product loader and the
// oracle's loader can be compared with the real one field by field.
static void dump_scene(const char *path, const Scene &s)
{
	FILE *f = fopen(path, "w");
	fprintf(f, "camera");
	hex3(f, s.camera.position);
	hex3(f, s.camera.direction);
	hex3(f, s.camera.up);
	hex3(f, s.camera.right);
	fprintf(f, "\nbackground");
	hex3(f, s.background);
	fprintf(f, "\nambient");
	hex3(f, s.ambient_light.colour);
	fprintf(f, "\ncounts %zu %zu %zu %zu\n", s.spheres.size(), s.triangles.size(), s.point_lights.size(), s.directional_lights.size());
	for(size_t i = 0; i < s.spheres.size(); i++)
	{
		fprintf(f, "sphere");
		hex3(f, s.spheres[i].collider.position);
		hexf(f, s.spheres[i].collider.radius);
		hex3(f, s.spheres[i].material.ambient);
		hex3(f, s.spheres[i].material.diffuse);
		hex3(f, s.spheres[i].material.specular);
		hexf(f, s.spheres[i].material.power);
		fprintf(f, "\n");
	}
	for(size_t i = 0; i < s.point_lights.size(); i++)
	{
		fprintf(f, "point_light");
		hex3(f, s.point_lights[i].position);
		hex3(f, s.point_lights[i].colour);
		fprintf(f, "\n");
	}
	for(size_t i = 0; i < s.triangles.size(); i++)
	{
		fprintf(f, "triangle");
		hex3(f, s.triangles[i].v0);
		hex3(f, s.triangles[i].v1);
		hex3(f, s.triangles[i].v2);
		fprintf(f, "\n");
	}
	fclose(f);
}

// raytrace.h:139-227 + :36-103 with the `return total_colour;` of :44 taken out (see the header): control flow restated,
// every value from the reference's own functions.
static glm::vec3 shade_legacy(Ray ray, const Scene &scene, int depth)
{
	if(depth <= 0) return glm::vec3(0.0f, 0.0f, 0.0f); // :141-144
	float min_distance = INFINITY;
	int sphere_index = -1;
	bool hit_a_sphere = false, hit_a_triangle = false;
	for(unsigned int i = 0; i < scene.spheres.size(); i++) // :152-165
		if(intersection_occurs(ray, scene.spheres[i].collider))
		{
			hit_a_sphere = true;
			float distance = collision_distance(ray, scene.spheres[i].collider);
			if(distance < min_distance)
			{
				min_distance = distance;
				sphere_index = (int) i;
			}
		}
	for(unsigned int i = 0; i < scene.triangles.size(); i++) // :171-186
	{
		float t, u, v;
		if(triangle_intersection_occurs(ray, scene.triangles[i], t, u, v) && t < min_distance)
		{
			min_distance = t;
			hit_a_sphere = false;
			hit_a_triangle = true;
		}
	}
	if(!hit_a_sphere && !hit_a_triangle) return scene.background; // :189-192
	if(!hit_a_sphere) return glm::vec3(0.0f, 0.0f, 0.0f);         // :221-224
	const Sphere &sphere = scene.spheres[sphere_index];
	// :197-205
	glm::vec3 e_c = ray.position - sphere.collider.position;
	float a = glm::dot(ray.direction, ray.direction);
	float b = 2 * glm::dot(ray.direction, e_c);
	float c = glm::dot(e_c, e_c) - sphere.collider.radius * sphere.collider.radius;
	float t = smallest_root(a, b, c);
	glm::vec3 P = ray.position + ray.direction * t;
	glm::vec3 N = glm::normalize(P - sphere.collider.position);
	// :38-42
	glm::vec3 total_colour = glm::vec3(0.0f, 0.0f, 0.0f);
	total_colour += bp::ambient_shading(scene, sphere);
	total_colour += bp::diffuse_shading(scene, sphere, P, N);
	total_colour += bp::specular_shading(scene, sphere, P, N);
	// :45-102
	float fr = bp::fresnel(ray.direction, N, sphere);
	glm::vec3 refraction_colour = glm::vec3(0.0f, 0.0f, 0.0f), reflection_colour = glm::vec3(0.0f, 0.0f, 0.0f);
	if(sphere.material.specular != glm::vec3(0.0f, 0.0f, 0.0f) && depth > 0)
	{
		const size_t n_point = scene.point_lights.size(), n_all = n_point + scene.directional_lights.size();
		for(size_t i = 0; i < n_all; i++)
		{ // :54-77 for the point lights, then :80-99 for the directional ones: the same statements around another light direction
			glm::vec3 light_direction = i < n_point ? glm::normalize(scene.point_lights[i].position - P) : glm::normalize(scene.directional_lights[i - n_point].direction);
			if(fr < 1)
			{
				Ray refracted_ray;
				refracted_ray.position = P;
				refracted_ray.direction = bp::refraction(ray.direction, N, sphere);
				refraction_colour = fr * shade_legacy(refracted_ray, scene, depth - 1); // (an assignment at :70 / :92: the last light's stays)
			}
			Ray reflected_ray;
			reflected_ray.position = P;
			reflected_ray.direction = bp::reflect_direction(light_direction, N);
			reflection_colour += (1 - fr) * sphere.material.specular * shade_legacy(reflected_ray, scene, depth - 1);
		}
	}
	return total_colour + refraction_colour + reflection_colour; // :102
}

// --eval-legacy FILE: bp::fresnel / bp::refraction / bp::reflect_direction on 10 000 (direction, normal, ior) triples, inputs and
// outputs as hex words, one triple per line.  The triples come from a fixed LCG: ray directions of length 0.5 .. 2 (primary rays are
// not normalised, main.cpp:155), unit normals, ior in [0.5, 2.5] with every eighth exactly 1.
static int eval_legacy(const char *path)
{
	FILE *f = fopen(path, "w");
	if(!f) return 2;
	uint64_t st = 0x9E3779B97F4A7C15ull;
	auto unit = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float) ((st >> 40) & 0xFFFFFF) / 16777216.0f; };
	for(int k = 0; k < 10000; k++)
	{
		glm::vec3 d(unit() * 2 - 1, unit() * 2 - 1, unit() * 2 - 1), n(unit() * 2 - 1, unit() * 2 - 1, unit() * 2 - 1);
		if(glm::dot(d, d) < 1e-3f || glm::dot(n, n) < 1e-3f) { k--; continue; }
		d = glm::normalize(d) * (0.5f + 1.5f * unit());
		n = glm::normalize(n);
		Sphere sphere;
		sphere.material.ior = (k % 8 == 0) ? 1.0f : 0.5f + 2.0f * unit();
		const float fr = bp::fresnel(d, n, sphere);
		const glm::vec3 rf = bp::refraction(d, n, sphere), rl = bp::reflect_direction(glm::normalize(d), n);
		hex3(f, d);
		hex3(f, n);
		hexf(f, sphere.material.ior);
		hexf(f, fr);
		hex3(f, rf);
		hex3(f, rl);
		fprintf(f, "\n");
	}
	fclose(f);
	return 0;
}

int main(int argc, char **argv)
{
	Options option;
	const char *path = nullptr, *output = nullptr, *float_out = nullptr, *dump = nullptr;
	int width = 1920, height = 1080; // scene.h:15 defaults, overridden by CLI (main.cpp:393-395)
	bool use_shadows = false, parallel_entry = false, strict = false, legacy = false;
	unsigned seed = 1;

	for(int i = 1; i < argc; i++)
	{
		auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[i + 1] : "0"; };
		if(!strcmp(argv[i], "--gillum"))
		{
			option.monte_carlo	   = true; // main.cpp:252-253
			option.num_path_traces = atoi(next());
		}
		else if(!strcmp(argv[i], "--fov")) option.fov = atof(next());
		else if(!strcmp(argv[i], "--jsample")) option.grid_size = atoi(next());
		else if(!strcmp(argv[i], "--width")) width = atoi(next());
		else if(!strcmp(argv[i], "--height")) height = atoi(next());
		else if(!strcmp(argv[i], "--depth")) option.max_depth = atoi(next());
		else if(!strcmp(argv[i], "--shadow")) use_shadows = true;
		else if(!strcmp(argv[i], "--path")) path = next();
		else if(!strcmp(argv[i], "--output")) output = next();
		else if(!strcmp(argv[i], "--seed")) seed = (unsigned) strtoul(next(), nullptr, 10);
		else if(!strcmp(argv[i], "--float-out")) float_out = next();
		else if(!strcmp(argv[i], "--dump-scene")) dump = next();
		else if(!strcmp(argv[i], "--parallel-entry")) parallel_entry = true;
		else if(!strcmp(argv[i], "--strict")) strict = true;
		else if(!strcmp(argv[i], "--legacy")) legacy = true;
		else if(!strcmp(argv[i], "--eval-legacy")) return eval_legacy(next());
	}
	if(legacy && option.monte_carlo)
	{
		fprintf(stderr, "ref_render: --legacy is pinned without --gillum (montecarlo_global_illumination() calls the reference's own shade(), which leaves the mode)\n");
		return 2;
	}
	if(!path || (!output && !dump))
	{
		fprintf(stderr, "usage: ref_render --path X.scn --output Y.ppm [--width --height --fov --gillum --jsample --depth --shadow --seed --float-out F --dump-scene D --parallel-entry]\n");
		return 2;
	}

	// parseScene() echoes every line to stdout and drops simplesphere.txt into
	// the CWD (scene.cpp:96-102): silence the former, send the latter to /tmp.
	char resolved[4096];
	if(!realpath(path, resolved))
	{
		fprintf(stderr, "ref_render: no such scene %s\n", path);
		return 2;
	}
	std::string out_abs, fout_abs, dump_abs;
	char cwd[4096];
	if(!getcwd(cwd, sizeof cwd)) return 2;
	auto absol = [&](const char *p) { return (!p) ? std::string() : (p[0] == '/' ? std::string(p) : std::string(cwd) + "/" + p); };
	out_abs	 = absol(output);
	fout_abs = absol(float_out);
	dump_abs = absol(dump);
	if(chdir("/tmp") != 0) return 2;
	fflush(stdout);
	int saved = dup(1);
	int devnull = open("/dev/null", O_WRONLY);
	dup2(devnull, 1);
	Scene scene = parseScene(resolved);
	fflush(stdout);
	dup2(saved, 1);
	close(devnull);
	close(saved);

	scene.spherical_fog.clear(); // pinned: UB fog line ignored (see header)
	if(strict)
	{ // --strict-scn: scene.cpp:139-163 builds every directional light and forgets to push it.  The lines are
	  // read again here exactly as scene.cpp reads them (same sscanf, same clamp) and pushed; everything that SHADES them —
	  // blinn_phong.h:77-85,122-131, utils.h:60-76 — is the reference's own, unmodified code.
		FILE *fp = fopen(resolved, "r");
		char line[1024], command[100];
		while(fp && fgets(line, 1024, fp))
		{
			if(line[0] == '#') continue;
			if(sscanf(line, "%s ", command) < 1) continue;
			if(strcmp(command, "directional_light") != 0) continue;
			float r, g, b, x, y, z;
			sscanf(line, "directional_light %f %f %f %f %f %f", &r, &g, &b, &x, &y, &z);
			if(r > 1) r = 1;
			if(g > 1) g = 1;
			if(b > 1) b = 1;
			DirectionalLight directional_light;
			directional_light.direction = glm::vec3(x, y, z);
			directional_light.colour	= glm::vec3(r, g, b);
			scene.directional_lights.push_back(directional_light);
		}
		if(fp) fclose(fp);
	}
	scene.width		  = width;
	scene.height	  = height;
	scene.use_shadows = use_shadows;
	if(parallel_entry)
	{ // main.cpp:21-24
		scene.width		 = 640;
		scene.height	 = 480;
		option.max_depth = 1;
		option.grid_size = 0;
	}
	if(dump) dump_scene(dump_abs.c_str(), scene);
	if(!output) return 0;

	srand(seed);
	const int W = scene.width, H = scene.height;
	glm::vec3 *image = new glm::vec3[(size_t) W * H]; // zero-initialised by glm 0.9.5.4's default ctor

	// rand() is consumed iff jitter or GI is on; then the loop must be serial
	// (DFS draw order).  Otherwise rows are independent and may run in parallel.
	const bool uses_rand = option.grid_size > 0 || option.monte_carlo;
#pragma omp parallel for schedule(dynamic, 1) if(!uses_rand)
	for(int y = 0; y < H; y++)
	{
		for(int x = 0; x < W; x++)
		{
			// main.cpp:134-137
			float inv_width	   = 1 / float(scene.width);
			float inv_height   = 1 / float(scene.height);
			float aspect_ratio = scene.width / float(scene.height);
			float angle		   = tan(M_PI * 0.5 * option.fov / 180.);
			glm::vec3 &px	   = image[(size_t) y * W + x];

			if(option.grid_size > 0)
			{ // main.cpp:140-166
				for(int i = 0; i < option.grid_size; i++)
				{
					for(int j = 0; j < option.grid_size; j++)
					{
						float r = static_cast<float>(rand()) / static_cast<float>(RAND_MAX);
						float u = (2 * ((x + r) * inv_width) - 1) * angle * aspect_ratio;
						float v = (1 - 2 * ((y + r) * inv_height)) * angle;
						glm::vec3 ray_dir(scene.camera.direction + u * scene.camera.right + v * scene.camera.up);
						Ray ray;
						ray.position  = scene.camera.position;
						ray.direction = ray_dir; // main.cpp:155 discards the normalize() result
						px += legacy ? shade_legacy(ray, scene, option.max_depth) : shade(ray, scene, option.max_depth, option.monte_carlo, option.num_path_traces);
					}
				}
				px /= (option.grid_size * option.grid_size);
			}
			else
			{ // main.cpp:168-182
				float u = (2 * ((x + 0.5) * inv_width) - 1) * angle * aspect_ratio;
				float v = (1 - 2 * ((y + 0.5) * inv_height)) * angle;
				glm::vec3 ray_dir(scene.camera.direction + u * scene.camera.right + v * scene.camera.up);
				Ray ray;
				ray.position  = scene.camera.position;
				ray.direction = ray_dir;
				px			  = legacy ? shade_legacy(ray, scene, option.max_depth) : shade(ray, scene, option.max_depth, option.monte_carlo, option.num_path_traces);
			}
		}
	}

	write_ppm(out_abs.c_str(), W, H, image);
	if(float_out)
	{
		FILE *f = fopen(fout_abs.c_str(), "wb");
		fwrite(image, sizeof(glm::vec3), (size_t) W * H, f);
		fclose(f);
	}
	delete[] image;
	return 0;
}
