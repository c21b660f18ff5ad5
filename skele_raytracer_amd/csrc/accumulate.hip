// Progressive accumulation: the headless form of what the reference's SDL viewer is for
// (main.cpp:183-197 shows the frame while it forms).  K whole frames under the seeds s, s+1, ..., s+K-1 are summed in
// binary32 in pass order and divided by K once; the mean is quantised exactly like a single frame (main.cpp:205).
//
// Both kernels are plain streams over the float3 image: 36 bytes per pixel for an accumulation step (two reads, one
// write), 12 + 3 (+12) for the final division — HBM-bound, ~10 us per 1080p pass against ~1.8 ms for the frame itself.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"

// acc = frame (first pass) or acc + frame, element by element
__global__ __launch_bounds__(256) void skr_accumulate_kernel(float *__restrict__ acc, const float *__restrict__ frame, size_t n, int first)
{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for(size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc[i] = first ? frame[i] : acc[i] + frame[i];
}

// mean = acc / passes; rows of a final partial tile beyond the image are left untouched, as skr_render_tiles leaves them
__global__ __launch_bounds__(256) void skr_resolve_accumulated_kernel(const float *__restrict__ acc, float passes, uint32_t width, uint32_t out_rows, uint32_t height,
																	  uint32_t tile_rows, uint32_t first_tile, uint32_t tile_stride, const uint32_t *__restrict__ tile_table, uint8_t *__restrict__ rgb, float *__restrict__ rgbf)
{
	const size_t n = (size_t) width * out_rows * 3, stride = (size_t) gridDim.x * blockDim.x;
	for(size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
	{
		const uint32_t orow = (uint32_t) (i / ((size_t) width * 3));
		const uint32_t k = orow / tile_rows;
		const uint32_t t = tile_table ? tile_table[k] : first_tile + k * tile_stride;
		if(t == 0xFFFFFFFFu || t * tile_rows + (orow - k * tile_rows) >= height) continue;
		const float m = sk_divf(acc[i], passes);
		if(rgbf) rgbf[i] = m;
		if(rgb) rgb[i] = (uint8_t) quantise(m);
	}
}

static unsigned blocks_for(size_t n)
{
	const size_t b = (n + 255) / 256;
	return (unsigned) (b < 1 ? 1 : b > 4096 ? 4096 : b); // 16 workgroups per CU: enough loads in flight for HBM
}

hipError_t skr_launch_accumulate(float *acc, const float *frame, size_t n, int first, hipStream_t stream)
{
	if(n == 0) return hipSuccess;
	hipLaunchKernelGGL(skr_accumulate_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, acc, frame, n, first);
	return hipGetLastError();
}

hipError_t skr_launch_resolve_accumulated(const float *acc, uint32_t passes, uint32_t width, uint32_t out_rows, uint32_t height, uint32_t tile_rows,
										  uint32_t first_tile, uint32_t tile_stride, const uint32_t *tile_table, uint8_t *rgb, float *rgbf, hipStream_t stream)
{
	const size_t n = (size_t) width * out_rows * 3;
	if(n == 0) return hipSuccess;
	hipLaunchKernelGGL(skr_resolve_accumulated_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, acc, (float) passes, width, out_rows, height, tile_rows, first_tile,
					   tile_stride, tile_table, rgb, rgbf);
	return hipGetLastError();
}
