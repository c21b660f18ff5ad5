/* oracle/skr_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C interface of the CPU restatement of the reference's hot path (see
 * skr_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libskr.so, the
 * raytracer CLI) never links, loads or calls it.
 */
#ifndef SKR_ORACLE_H
#define SKR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } sko_vec3;

/* reference: shapes.h:12 Sphere = SphereCollider.h:8 + material.h:9 */
typedef struct {
	sko_vec3 center;
	float radius;
	sko_vec3 ambient, diffuse, specular;
	float power;
	sko_vec3 transmissive; /* parsed, only used by dead code (raytrace.h:45-103) */
	float ior;
} sko_sphere;

/* reference: shapes.h:26 Triangle (material never read: raytrace.h:221-224) */
typedef struct { sko_vec3 v0, v1, v2; } sko_triangle;

/* reference: lights.h:19 */
typedef struct { sko_vec3 position, colour; } sko_point_light;
/* reference: lights.h:13 */
typedef struct { sko_vec3 direction, colour; } sko_directional_light;

/* reference: scene.h:13-28 (+ camera.h:8-32) */
typedef struct {
	sko_vec3 cam_pos, cam_dir, cam_up, cam_right;
	float cam_half_angle;          /* stored, never used (camera.h:14) */
	sko_vec3 background, ambient;
	int n_spheres, n_triangles, n_point_lights, n_vertices;
	sko_sphere *spheres;
	sko_triangle *triangles;
	sko_point_light *point_lights;
	int film_w, film_h;            /* parsed, overridden by the CLI (main.cpp:393-395) */
	int max_depth_parsed;          /* parsed, never read (scene.cpp:192-198) */
	int n_directional_dropped;     /* scene.cpp:139-163 builds the light and never pushes it */
	int n_fog_skipped;             /* scene.cpp:207-212 is UB; pinned as "ignored" */
	int n_unknown, n_bad_triangles;
	/* --strict-scn only (sko_scene_load_ex): the directional lights scene.cpp:139-163 builds and forgets, pushed */
	int n_directional_lights;
	sko_directional_light *directional_lights;
	/* --shade-triangles only: the material in force when each triangle line was read (shapes.h:26 keeps it in the Triangle;
	 * raytrace.h:221-224 never looks at it).  center / radius of these entries are unused. */
	sko_sphere *triangle_materials;
} sko_scene;

enum { SKO_RNG_GLIBC_REPLAY = 0, SKO_RNG_COUNTER = 1 };
enum { SKO_MATH_LIBM = 0, SKO_MATH_SHARED = 1 };

/* reference: utils.h:26-34 Options + main.cpp:236-244 locals */
typedef struct {
	int32_t width, height;
	float fov;
	int32_t monte_carlo;      /* --gillum present */
	int32_t num_path_traces;  /* --gillum N */
	int32_t grid_size;        /* --jsample g */
	int32_t max_depth;        /* --depth d */
	int32_t use_shadows;      /* --shadow */
	int32_t rng_mode;         /* SKO_RNG_* */
	int32_t math_mode;        /* SKO_MATH_* */
	uint64_t seed;            /* srand((unsigned)seed) in replay mode; Philox key in counter mode */
	int32_t y0, y1;           /* rows [y0,y1) to render (whole image: 0,height) */
	int32_t threads;          /* OpenMP threads; replay mode with rand() in use forces 1 */
	/* --shade-triangles (NO reference behaviour exists: raytrace.h:221-224 returns black).  The spec of this mode:
	 * a triangle is hit where utils.h:181-213 accepts it (flipped u and |det| >= 1e-5 kept) with 0 < t < the closest sphere's t
	 * — the t > 0 test utils.h:213 lacks; hits behind the origin are ignored instead of blackening —, except the triangle the ray
	 * starts on (--gillum children of a triangle hit: P + 1e-5 may lie on either side of it); the smallest t wins, the lower file
	 * index on a tie; it is shaded like a sphere hit — the triangle's material, the geometric normal
	 * normalize(cross(v1-v0, v2-v0)) turned against the ray, shadow rays against spheres only (utils.h:42-76), --gillum children
	 * from P + 1e-5 with the basis of utils.h:148-165. */
	int32_t shade_triangles;
	/* --legacy-reflect: the code behind the early `return total_colour;` of raytrace.h:44 runs — the Fresnel term
	 * (blinn_phong.h:156-184), and for every light one refraction ray (:143-153, `=`: the last light's stays) and one reflection ray
	 * (:137-140: the LIGHT direction mirrored at the normal) from the hit point itself, each shade(depth - 1), added to the direct
	 * term (raytrace.h:45-103).  UNREACHABLE at HEAD: no output of the reference covers it.  What the C++ of those lines does is
	 * restated literally (unqualified sqrt = binary64, powf(x, 2) = x*x, abs = fabsf); the children's nodes of the counter RNG:
	 * arity A = N + 2 (lights), child N + 2 l = refraction of light l, N + 2 l + 1 = its reflection (point lights first). */
	int32_t legacy_reflect;
} sko_options;

/* stats[0]=radiance rays (shade() calls with depth>0), [1]=sphere hits shaded,
 * [2]=shadow casts (unique, i.e. one per light per hit), [3]=sphere tests,
 * [4]=triangle tests.  May be NULL. */
int sko_scene_load(const char *path, sko_scene *out);
/* strict != 0: the loader as its author evidently meant it: directional lights are kept (colour
 * clamped to <= 1 as scene.cpp:143-154 does) and shaded by blinn_phong.h:77-85,122-131 with the shadow test of
 * utils.h:60-76.  film_resolution and max_depth are parsed either way; honouring them is the caller's business. */
int sko_scene_load_ex(const char *path, int strict, sko_scene *out);
void sko_scene_free(sko_scene *s);
int sko_render(const sko_scene *scene, const sko_options *opt, uint8_t *rgb, float *rgbf, uint64_t *stats);

/* Spec functions exposed for unit tests (tests compare the device
 * implementations against these on the same inputs). */
#define SKO_PHILOX_ROUNDS 7
void sko_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
void sko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);   /* (known-answer tests of the round function) */
void sko_philox4x32_spec(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]); /* SKO_PHILOX_ROUNDS rounds: what the draws use */
void sko_sincos_shared(float phi, float *s, float *c);
float sko_powf_shared(float x, float p);
float sko_smallest_root(float a, float b, float c);
int sko_triangle_test(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3], float *t);
uint8_t sko_quantise(float c);
void sko_basis(const float n[3], float nt[3], float nb[3]);
void sko_counter_draws(uint64_t seed, uint32_t pixel, uint32_t aa, uint32_t parent_node, uint32_t child, float *r1, float *r2);
float sko_counter_jitter(uint64_t seed, uint32_t pixel, uint32_t aa);
/* The primary ray direction the render loop forms for pixel (x, y): main.cpp:146-155 with the draw r (jitter != 0) or
 * :170-174 at the pixel centre.  The loop calls this very function. */
void sko_primary_direction(const sko_scene *scene, int width, int height, float fov, int x, int y, int jitter, float r, float out[3]);
void sko_legacy_eval(const float d[3], const float n[3], float ior, float out[7]); /* blinn_phong.h:156, :143, :137 restated: {fresnel, refraction, reflect_direction(normalize(d), n)} */
int sko_write_ppm(const char *path, int w, int h, const uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
