// Shape of the triangle culling hierarchy, shared by the host builder (scene_host.cpp
// build_triangle_chunks) and the device walk (shade_common.h any_triangle_closer).
#pragma once

/* Consecutive (Morton-ordered) triangles per first-level sphere, chosen per scene by skr_scene::finalize():
 * scenes without spheres only ever trace camera rays, whose 8x8-pixel waves are coherent and gain from small
 * chunks (dragon 1080p: 2.14 / 1.80 / 1.53 ms at 32 / 16 / 8); scenes with spheres also trace GI children whose
 * waves touch most chunks anyway, and pay for every extra sphere test (test.scn 2.77 / 2.87 / 3.04 ms). */
#define SKR_TRI_CHUNK_COHERENT 8
#define SKR_TRI_CHUNK_MIXED 32
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
