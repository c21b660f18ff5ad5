// Shape of the triangle culling hierarchy, shared by the host builder (scene_host.cpp
// build_triangle_chunks) and the device walk (shade_common.h any_triangle_closer).
#pragma once

/* Consecutive (Morton-ordered) triangles per chunk sphere, chosen per scene by skr_scene::finalize(): scenes
 * without spheres only ever trace camera rays, whose 8x8-pixel waves are coherent and gain from small chunks
 * (dragon 1080p under the tree walk: 1.45 / 1.35 / 1.22 / 1.19 ms at 8 / 6 / 4 / 2); scenes with spheres also trace
 * GI children whose waves touch many chunks and pay for every extra sphere test (test.scn: 1.84 / 1.58 / 1.56 /
 * 1.54 / 1.61 ms at 4 / 8 / 12 / 16 / 32).  SKR_TRI_CHUNK in the environment overrides both (tuning runs). */
#ifndef SKR_TRI_CHUNK_COHERENT
#define SKR_TRI_CHUNK_COHERENT 4
#endif
#ifndef SKR_TRI_CHUNK_MIXED
#define SKR_TRI_CHUNK_MIXED 16
#endif
#ifndef SKR_TRI_SUPER
#define SKR_TRI_SUPER 8 /* children per node of the tree above the chunk spheres */
#endif
/* The rounding slack of the triangle test grows with |d| (its |det| >= 1e-5 cut is absolute), so the spheres are
 * built for three bounds on the direction length and the launcher picks the tightest one that covers the frame:
 * GI children are at most 3 long (raytrace.h:123-125 mixes three unit vectors with a unit sample), camera rays
 * |direction + u right + v up| depend on the .scn camera and --fov (main.cpp:154-155). */
#define SKR_CULL_LEVELS 3
#define SKR_CULL_DMAX_LIST {4.0, 32.0, 256.0}
#define SKR_CULL_DMAX 256.0 /* beyond the last bound the walk is brute force */
