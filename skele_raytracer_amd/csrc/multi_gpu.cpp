// Native multi-GPU frame: the framebuffer sharded in interleaved row tiles over the GPUs of one node and gathered with
// ONE RCCL collective over xGMI (include/skr.h "multi-GPU").  The reference's parallel entry is one C++ process that fans
// the rows of a frame out over OpenMP threads (src/main.cpp:19-104, fan-out at :33, dispatch :402-410) and has no
// distributed path at all; this is its MI355X form, in the two shapes a caller needs:
//
//   skr_multi_*   ONE process drives N devices: a renderer, a stream and a worker thread per device, ncclCommInitAll,
//                 per frame one skr_render_tiles launch sequence per device and one grouped ncclAllGather of the u8 tile
//                 buffers (every rank renders straight into its slot of the gather buffer: no staging copy); the root
//                 de-interleaves on the device (a copy kernel) and owns the frame.  What `bin/raytracer --gpus N` uses.
//   skr_comm_*    one process PER device (torchrun, mpirun): the same frame step on a communicator made with
//                 ncclCommInitRank from an id the caller broadcasts by whatever transport it has.  What bench.py uses.
//
// Tile t belongs to rank t mod G (cost is very non-uniform vertically); random numbers are keyed by the global pixel
// index, so the image does not depend on G.  RCCL is bound at run time (dlopen): libskr.so loads, and renders on one
// GPU, on a box without it, and inside a process that already carries an RCCL (PyTorch ships one) it uses that one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/skr.h"

void skr_set_error(const char *fmt, ...);

namespace {

// ---- RCCL, bound lazily ------------------------------------------------------------------------------------------
struct Rccl {
	bool ok = false;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl &rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		// a copy already in the process (PyTorch's) first; then the ROCm installation's
		void *h = nullptr;
		if(dlsym(RTLD_DEFAULT, "ncclAllGather")) h = RTLD_DEFAULT;
		const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for(int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
		if(!h) return;
		r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
		r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
		r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
		r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
		r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
		r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
		r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
		r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
		r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
	});
	return r;
}

#define SKR_HIP(call)                                                                                   \
	do                                                                                                  \
	{                                                                                                   \
		hipError_t e_ = (call);                                                                         \
		if(e_ != hipSuccess)                                                                            \
		{                                                                                               \
			skr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);  \
			return SKR_ERR_HIP;                                                                         \
		}                                                                                               \
	} while(0)
#define SKR_NCCL(call)                                                                                          \
	do                                                                                                          \
	{                                                                                                           \
		ncclResult_t e_ = (call);                                                                               \
		if(e_ != ncclSuccess)                                                                                   \
		{                                                                                                       \
			skr_set_error("%s failed: %s (%s:%d)", #call, rccl().GetErrorString(e_), __FILE__, __LINE__);      \
			return SKR_ERR_HIP;                                                                                 \
		}                                                                                                       \
	} while(0)

// ---- the partition (the one definition both shapes and the tests use) -------------------------------------------
uint32_t tiles_total(int32_t height, uint32_t tile_rows) { return ((uint32_t) height + tile_rows - 1) / tile_rows; }
uint32_t tiles_per_rank(int32_t height, uint32_t tile_rows, uint32_t world) { return (tiles_total(height, tile_rows) + world - 1) / world; }

// gathered: [world][k_max * tile_rows][row_bytes] (rank-major, what the all-gather leaves) -> frame[height][row_bytes].
// Tile t lives at rank t mod world, slot t / world.  One 16-byte word per thread where the rows allow it.
template <typename T>
__global__ __launch_bounds__(256) void skr_deinterleave_kernel(const T *gathered, T *frame, uint32_t height, uint32_t row_words, uint32_t tile_rows, uint32_t world, uint32_t k_max)
{
	const uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x;
	if(i >= (uint64_t) height * row_words) return;
	const uint32_t y = (uint32_t) (i / row_words), x = (uint32_t) (i - (uint64_t) y * row_words);
	const uint32_t t = y / tile_rows, rank = t % world, k = t / world;
	frame[i] = gathered[((uint64_t) rank * k_max * tile_rows + (uint64_t) k * tile_rows + (y - t * tile_rows)) * row_words + x];
}

hipError_t launch_deinterleave(const uint8_t *gathered, uint8_t *frame, int32_t width, int32_t height, uint32_t tile_rows, uint32_t world, uint32_t k_max, hipStream_t stream)
{
	const size_t row_bytes = (size_t) width * 3;
	if(row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(frame)) % 16 == 0)
	{
		const uint32_t rw = (uint32_t) (row_bytes / 16);
		const uint64_t n = (uint64_t) height * rw;
		hipLaunchKernelGGL(skr_deinterleave_kernel<uint4>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(gathered),
						   reinterpret_cast<uint4 *>(frame), (uint32_t) height, rw, tile_rows, world, k_max);
	}
	else
	{
		const uint64_t n = (uint64_t) height * row_bytes;
		hipLaunchKernelGGL(skr_deinterleave_kernel<uint8_t>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, gathered, frame, (uint32_t) height,
						   (uint32_t) row_bytes, tile_rows, world, k_max);
	}
	return hipGetLastError();
}

// one rank's buffers for one frame geometry
struct RankBuffers {
	int32_t width = 0, height = 0;
	uint32_t tile_rows = 0, world = 0, k_max = 0;
	size_t chunk = 0;           // bytes one rank contributes: k_max * tile_rows * width * 3
	uint8_t *d_gather = nullptr; // [world][chunk]; this rank renders into slot `rank`
	uint8_t *d_frame = nullptr;  // root only: the de-interleaved frame
};

int size_buffers(RankBuffers &b, const skr_options *opt, uint32_t tile_rows, uint32_t world, bool root)
{
	if(b.d_gather && b.width == opt->width && b.height == opt->height && b.tile_rows == tile_rows && b.world == world) return SKR_OK;
	if(b.d_gather) SKR_HIP(hipFree(b.d_gather));
	if(b.d_frame) SKR_HIP(hipFree(b.d_frame));
	b.d_gather = b.d_frame = nullptr;
	b.width = opt->width;
	b.height = opt->height;
	b.tile_rows = tile_rows;
	b.world = world;
	b.k_max = tiles_per_rank(opt->height, tile_rows, world);
	b.chunk = (size_t) b.k_max * tile_rows * (size_t) opt->width * 3;
	SKR_HIP(hipMalloc((void **) &b.d_gather, b.chunk * world));
	SKR_HIP(hipMemset(b.d_gather, 0, b.chunk * world)); // the padding rows of a last partial tile travel too
	if(root) SKR_HIP(hipMalloc((void **) &b.d_frame, (size_t) opt->width * opt->height * 3));
	return SKR_OK;
}

void free_buffers(RankBuffers &b)
{
	if(b.d_gather) (void) hipFree(b.d_gather);
	if(b.d_frame) (void) hipFree(b.d_frame);
	b = RankBuffers();
}

int check_frame_args(const skr_options *opt, uint32_t tile_rows)
{
	if(!opt || tile_rows == 0 || opt->width <= 0 || opt->height <= 0 || opt->width > 65536 || opt->height > 65536)
	{
		skr_set_error("multi-GPU frame: bad image size or tile_rows");
		return SKR_ERR_ARG;
	}
	return SKR_OK;
}

} // namespace

// =====================================================================================================================
// one process per device
// =====================================================================================================================
struct skr_comm {
	skr_renderer *r = nullptr; // not owned
	int device = 0, rank = 0, world = 1;
	ncclComm_t comm = nullptr;
	RankBuffers buf;
	// pipelined frames (skr_comm_render_frame_async): two buffer sets, the collective on a stream of its own
	RankBuffers abuf[2];
	hipStream_t cs = nullptr;
	hipEvent_t rendered[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
	bool in_flight[2] = {false, false};
	uint64_t async_frames = 0;
};

extern "C" {

int skr_rccl_available(void) { return rccl().ok ? 1 : 0; }

int skr_comm_unique_id(uint8_t id[SKR_COMM_ID_BYTES])
{
	static_assert(SKR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "include/skr.h carries RCCL's id size");
	if(!id) return SKR_ERR_ARG;
	if(!rccl().ok)
	{
		skr_set_error("RCCL (librccl.so.1) is not loadable in this process");
		return SKR_ERR_UNSUPPORTED;
	}
	ncclUniqueId u;
	SKR_NCCL(rccl().GetUniqueId(&u));
	memcpy(id, u.internal, SKR_COMM_ID_BYTES);
	return SKR_OK;
}

int skr_comm_create(skr_renderer *r, int device, const uint8_t id[SKR_COMM_ID_BYTES], int rank, int world, skr_comm **out)
{
	if(!r || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id))
	{
		skr_set_error("skr_comm_create: bad argument");
		return SKR_ERR_ARG;
	}
	*out = nullptr;
	skr_comm *c = new skr_comm();
	c->r = r;
	c->device = device;
	c->rank = rank;
	c->world = world;
	if(world > 1 || id)
	{ // (a world of one still goes through RCCL when the caller hands an id: how the path is exercised on a one-GPU box)
		if(!rccl().ok)
		{
			delete c;
			skr_set_error("RCCL (librccl.so.1) is not loadable in this process");
			return SKR_ERR_UNSUPPORTED;
		}
		hipError_t e = hipSetDevice(device);
		ncclUniqueId u;
		memcpy(u.internal, id, SKR_COMM_ID_BYTES);
		ncclResult_t ne = e == hipSuccess ? rccl().CommInitRank(&c->comm, world, u, rank) : ncclUnhandledCudaError;
		if(ne != ncclSuccess)
		{
			skr_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, e == hipSuccess ? rccl().GetErrorString(ne) : hipGetErrorString(e));
			delete c;
			return SKR_ERR_HIP;
		}
	}
	*out = c;
	return SKR_OK;
}

void skr_comm_destroy(skr_comm *c)
{
	if(!c) return;
	(void) hipSetDevice(c->device);
	if(c->cs) (void) hipStreamSynchronize(c->cs);
	free_buffers(c->buf);
	for(int k = 0; k < 2; k++)
	{
		free_buffers(c->abuf[k]);
		if(c->rendered[k]) (void) hipEventDestroy(c->rendered[k]);
		if(c->gathered[k]) (void) hipEventDestroy(c->gathered[k]);
	}
	if(c->cs) (void) hipStreamDestroy(c->cs);
	if(c->comm) (void) rccl().CommDestroy(c->comm);
	delete c;
}

// This rank's tiles, the collective, and (rank 0) the de-interleave, all enqueued on `stream`.  *d_frame (rank 0) points
// at the finished W x H x 3 frame in device memory once the stream has drained; other ranks get NULL.
int skr_comm_render_frame(skr_comm *c, const skr_options *opt, uint32_t tile_rows, uint8_t **d_frame, void *stream)
{
	if(!c) return SKR_ERR_ARG;
	int rc = check_frame_args(opt, tile_rows);
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipSetDevice(c->device));
	rc = size_buffers(c->buf, opt, tile_rows, (uint32_t) c->world, c->rank == 0);
	if(rc != SKR_OK) return rc;
	RankBuffers &b = c->buf;
	uint8_t *mine = b.d_gather + (size_t) c->rank * b.chunk;
	rc = skr_render_tiles(c->r, opt, tile_rows, (uint32_t) c->rank, (uint32_t) c->world, mine, nullptr, stream);
	if(rc != SKR_OK) return rc;
	if(c->comm) SKR_NCCL(rccl().AllGather(mine, b.d_gather, b.chunk, ncclUint8, c->comm, (hipStream_t) stream)); // in place: slot `rank` is the send buffer
	if(c->rank == 0) SKR_HIP(launch_deinterleave(b.d_gather, b.d_frame, opt->width, opt->height, tile_rows, (uint32_t) c->world, b.k_max, (hipStream_t) stream));
	if(d_frame) *d_frame = c->rank == 0 ? b.d_frame : nullptr;
	return SKR_OK;
}

// The same frame step with the collective off the render stream: frame f's all-gather and de-interleave run on a stream of the
// communicator's own while `stream` goes on to render frame f + 1 into the other of two buffer sets — on 8 GPUs the collective
// is a third of a 0.3 ms share, and nothing in the next frame depends on it.  *d_prev_frame (rank 0): the frame of the
// PREVIOUS call, complete in `stream` order after this call (NULL on the first call and on the other ranks).
int skr_comm_render_frame_async(skr_comm *c, const skr_options *opt, uint32_t tile_rows, uint8_t **d_prev_frame, void *stream)
{
	if(!c) return SKR_ERR_ARG;
	int rc = check_frame_args(opt, tile_rows);
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipSetDevice(c->device));
	if(!c->cs)
	{
		SKR_HIP(hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking));
		for(int k = 0; k < 2; k++)
		{
			SKR_HIP(hipEventCreateWithFlags(&c->rendered[k], hipEventDisableTiming));
			SKR_HIP(hipEventCreateWithFlags(&c->gathered[k], hipEventDisableTiming));
		}
	}
	const int s = (int) (c->async_frames & 1u), prev = s ^ 1;
	hipStream_t rs = (hipStream_t) stream;
	RankBuffers &b = c->abuf[s];
	if(c->in_flight[s])
	{ // the collective of frame f - 2 read these buffers (a change of geometry frees them: wait on the host then)
		if(b.d_gather && (b.width != opt->width || b.height != opt->height || b.tile_rows != tile_rows)) SKR_HIP(hipEventSynchronize(c->gathered[s]));
		else SKR_HIP(hipStreamWaitEvent(rs, c->gathered[s], 0));
		c->in_flight[s] = false;
	}
	rc = size_buffers(b, opt, tile_rows, (uint32_t) c->world, c->rank == 0);
	if(rc != SKR_OK) return rc;
	uint8_t *mine = b.d_gather + (size_t) c->rank * b.chunk;
	rc = skr_render_tiles(c->r, opt, tile_rows, (uint32_t) c->rank, (uint32_t) c->world, mine, nullptr, stream);
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipEventRecord(c->rendered[s], rs));
	SKR_HIP(hipStreamWaitEvent(c->cs, c->rendered[s], 0));
	if(c->comm) SKR_NCCL(rccl().AllGather(mine, b.d_gather, b.chunk, ncclUint8, c->comm, c->cs));
	if(c->rank == 0) SKR_HIP(launch_deinterleave(b.d_gather, b.d_frame, opt->width, opt->height, tile_rows, (uint32_t) c->world, b.k_max, c->cs));
	SKR_HIP(hipEventRecord(c->gathered[s], c->cs));
	c->in_flight[s] = true;
	c->async_frames++;
	if(d_prev_frame)
	{ // the previous frame: whatever `stream` does from here on sees it whole (a caller that passes NULL does not look, and saves the wait)
		*d_prev_frame = nullptr;
		if(c->in_flight[prev])
		{
			SKR_HIP(hipStreamWaitEvent(rs, c->gathered[prev], 0));
			if(c->rank == 0) *d_prev_frame = c->abuf[prev].d_frame;
		}
	}
	return SKR_OK;
}

// Ends a run of skr_comm_render_frame_async calls: `stream` waits for the last frame's collective; *d_frame (rank 0) = that frame.
int skr_comm_flush(skr_comm *c, uint8_t **d_frame, void *stream)
{
	if(!c) return SKR_ERR_ARG;
	if(d_frame) *d_frame = nullptr;
	if(c->async_frames == 0) return SKR_OK;
	SKR_HIP(hipSetDevice(c->device));
	const int last = (int) ((c->async_frames - 1) & 1u);
	for(int k = 0; k < 2; k++)
		if(c->in_flight[k]) SKR_HIP(hipStreamWaitEvent((hipStream_t) stream, c->gathered[k], 0));
	if(d_frame && c->rank == 0) *d_frame = c->abuf[last].d_frame;
	return SKR_OK;
}

// Rank 0: waits for `stream` and copies the frame of the last skr_comm_render_frame to host memory (W*H*3 bytes).
int skr_comm_frame_to_host(skr_comm *c, uint8_t *h_rgb, void *stream)
{
	if(!c || !h_rgb || c->rank != 0 || !c->buf.d_frame)
	{
		skr_set_error("skr_comm_frame_to_host: rank 0 only, after skr_comm_render_frame");
		return SKR_ERR_ARG;
	}
	SKR_HIP(hipSetDevice(c->device));
	SKR_HIP(hipStreamSynchronize((hipStream_t) stream));
	SKR_HIP(hipMemcpy(h_rgb, c->buf.d_frame, (size_t) c->buf.width * c->buf.height * 3, hipMemcpyDeviceToHost));
	return SKR_OK;
}

} // extern "C"

// =====================================================================================================================
// one process, N devices
// =====================================================================================================================
struct skr_multi {
	int n = 0;
	std::vector<int> devices;
	std::vector<skr_renderer *> renderers;
	std::vector<hipStream_t> streams;
	std::vector<ncclComm_t> comms;
	std::vector<RankBuffers> bufs;
	hipEvent_t e0 = nullptr, e1 = nullptr; // root stream: frame time
	// one worker thread per device (a single thread would enqueue 8 devices' launch sequences one after the other)
	std::vector<std::thread> workers;
	std::mutex mu;
	std::condition_variable cv_go, cv_done;
	uint64_t generation = 0;
	int pending = 0;
	bool quit = false;
	const skr_options *opt = nullptr;
	uint32_t tile_rows = 0;
	std::vector<int> status;
	std::vector<std::string> errors;
};

namespace {

// what one device does for a frame: its tiles into its slot of its gather buffer (the collective follows, grouped, from the caller)
int multi_render_rank(skr_multi *m, int i)
{
	SKR_HIP(hipSetDevice(m->devices[i]));
	int rc = size_buffers(m->bufs[i], m->opt, m->tile_rows, (uint32_t) m->n, i == 0);
	if(rc != SKR_OK) return rc;
	RankBuffers &b = m->bufs[i];
	if(i == 0) SKR_HIP(hipEventRecord(m->e0, m->streams[0]));
	return skr_render_tiles(m->renderers[i], m->opt, m->tile_rows, (uint32_t) i, (uint32_t) m->n, b.d_gather + (size_t) i * b.chunk, nullptr, m->streams[i]);
}

void worker_main(skr_multi *m, int i)
{
	uint64_t seen = 0;
	for(;;)
	{
		{
			std::unique_lock<std::mutex> lk(m->mu);
			m->cv_go.wait(lk, [&] { return m->quit || m->generation != seen; });
			if(m->quit) return;
			seen = m->generation;
		}
		const int rc = multi_render_rank(m, i);
		{
			std::lock_guard<std::mutex> lk(m->mu);
			m->status[i] = rc;
			if(rc != SKR_OK) m->errors[i] = skr_last_error(); // (the error text is thread-local)
			if(--m->pending == 0) m->cv_done.notify_all();
		}
	}
}

} // namespace

extern "C" {

int skr_multi_create(const skr_scene *scene, int n_devices, const int *devices, skr_multi **out)
{
	if(!scene || !out || n_devices < 1)
	{
		skr_set_error("skr_multi_create: bad argument");
		return SKR_ERR_ARG;
	}
	*out = nullptr;
	int have = 0;
	if(hipGetDeviceCount(&have) != hipSuccess || have < n_devices)
	{
		skr_set_error("%d device(s) asked for, %d visible; libskr has no CPU fallback", n_devices, have);
		return SKR_ERR_NO_DEVICE;
	}
	if(n_devices > 1 && !rccl().ok)
	{
		skr_set_error("RCCL (librccl.so.1) is not loadable in this process");
		return SKR_ERR_UNSUPPORTED;
	}
	skr_multi *m = new skr_multi();
	m->n = n_devices;
	m->devices.resize(n_devices);
	for(int i = 0; i < n_devices; i++) m->devices[i] = devices ? devices[i] : i;
	m->renderers.assign(n_devices, nullptr);
	m->streams.assign(n_devices, nullptr);
	m->comms.assign(n_devices, nullptr);
	m->bufs.resize(n_devices);
	m->status.assign(n_devices, SKR_OK);
	m->errors.resize(n_devices);
	int rc = SKR_OK;
	for(int i = 0; i < n_devices && rc == SKR_OK; i++)
	{ // the scene is uploaded to every device from the host: <= 0.5 MB, no collective needed
		rc = skr_renderer_create(scene, m->devices[i], &m->renderers[i]);
		if(rc == SKR_OK && (hipSetDevice(m->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess))
		{
			skr_set_error("stream creation failed on device %d", m->devices[i]);
			rc = SKR_ERR_HIP;
		}
	}
	if(rc == SKR_OK && rccl().ok)
	{ // (one device too, when RCCL is there: the same frame step, and the path that a one-GPU box can test)
		const ncclResult_t ne = rccl().CommInitAll(m->comms.data(), n_devices, m->devices.data());
		if(ne != ncclSuccess)
		{
			skr_set_error("ncclCommInitAll(%d devices) failed: %s", n_devices, rccl().GetErrorString(ne));
			m->comms.assign(n_devices, nullptr);
			rc = SKR_ERR_HIP;
		}
	}
	if(rc == SKR_OK && (hipSetDevice(m->devices[0]) != hipSuccess || hipEventCreate(&m->e0) != hipSuccess || hipEventCreate(&m->e1) != hipSuccess))
	{
		skr_set_error("event creation failed");
		rc = SKR_ERR_HIP;
	}
	if(rc != SKR_OK)
	{
		skr_multi_destroy(m);
		return rc;
	}
	for(int i = 1; i < n_devices; i++) m->workers.emplace_back(worker_main, m, i); // device 0 is driven by the calling thread
	*out = m;
	return SKR_OK;
}

void skr_multi_destroy(skr_multi *m)
{
	if(!m) return;
	{
		std::lock_guard<std::mutex> lk(m->mu);
		m->quit = true;
	}
	m->cv_go.notify_all();
	for(std::thread &t : m->workers) t.join();
	for(int i = 0; i < m->n; i++)
	{
		(void) hipSetDevice(m->devices[i]);
		if(m->streams[i]) (void) hipStreamSynchronize(m->streams[i]);
		free_buffers(m->bufs[i]);
		if(m->comms[i]) (void) rccl().CommDestroy(m->comms[i]);
		if(m->streams[i]) (void) hipStreamDestroy(m->streams[i]);
		if(m->renderers[i]) skr_renderer_destroy(m->renderers[i]);
	}
	if(m->e0) (void) hipEventDestroy(m->e0);
	if(m->e1) (void) hipEventDestroy(m->e1);
	delete m;
}

int skr_multi_device_count(const skr_multi *m) { return m ? m->n : 0; }

skr_renderer *skr_multi_renderer(skr_multi *m, int i) { return (m && i >= 0 && i < m->n) ? m->renderers[i] : nullptr; }

// The whole frame: every device its tiles, one grouped all-gather, the root's de-interleave; synchronous.  *d_frame is the
// W x H x 3 frame in device 0's memory (owned by m, valid until the next call); frame_ms = first launch to de-interleaved
// frame on the root's stream.
int skr_multi_render_frame(skr_multi *m, const skr_options *opt, uint32_t tile_rows, uint8_t **d_frame, float *frame_ms)
{
	if(!m) return SKR_ERR_ARG;
	int rc = check_frame_args(opt, tile_rows);
	if(rc != SKR_OK) return rc;
	{
		std::lock_guard<std::mutex> lk(m->mu);
		m->opt = opt;
		m->tile_rows = tile_rows;
		m->pending = m->n - 1;
		m->generation++;
	}
	m->cv_go.notify_all();
	m->status[0] = multi_render_rank(m, 0);
	if(m->status[0] != SKR_OK) m->errors[0] = skr_last_error();
	{
		std::unique_lock<std::mutex> lk(m->mu);
		m->cv_done.wait(lk, [&] { return m->pending == 0; });
	}
	for(int i = 0; i < m->n; i++)
		if(m->status[i] != SKR_OK)
		{
			skr_set_error("device %d: %s", m->devices[i], m->errors[i].c_str());
			return m->status[i];
		}
	if(m->comms[0])
	{ // one collective: every rank's chunk to every rank (the root is the one that uses it), each on its rank's stream behind its kernels
		SKR_NCCL(rccl().GroupStart());
		for(int i = 0; i < m->n; i++)
		{
			RankBuffers &b = m->bufs[i];
			const ncclResult_t ne = rccl().AllGather(b.d_gather + (size_t) i * b.chunk, b.d_gather, b.chunk, ncclUint8, m->comms[i], m->streams[i]);
			if(ne != ncclSuccess)
			{
				(void) rccl().GroupEnd();
				skr_set_error("ncclAllGather(rank %d) failed: %s", i, rccl().GetErrorString(ne));
				return SKR_ERR_HIP;
			}
		}
		SKR_NCCL(rccl().GroupEnd());
	}
	SKR_HIP(hipSetDevice(m->devices[0]));
	RankBuffers &b0 = m->bufs[0];
	SKR_HIP(launch_deinterleave(b0.d_gather, b0.d_frame, opt->width, opt->height, tile_rows, (uint32_t) m->n, b0.k_max, m->streams[0]));
	SKR_HIP(hipEventRecord(m->e1, m->streams[0]));
	for(int i = m->n - 1; i >= 0; i--)
	{
		SKR_HIP(hipSetDevice(m->devices[i]));
		SKR_HIP(hipStreamSynchronize(m->streams[i]));
	}
	if(frame_ms) SKR_HIP(hipEventElapsedTime(frame_ms, m->e0, m->e1));
	if(d_frame) *d_frame = b0.d_frame;
	return SKR_OK;
}

int skr_multi_render_frame_host(skr_multi *m, const skr_options *opt, uint32_t tile_rows, uint8_t *h_rgb, float *frame_ms)
{
	if(!h_rgb) return SKR_ERR_ARG;
	uint8_t *d = nullptr;
	const int rc = skr_multi_render_frame(m, opt, tile_rows, &d, frame_ms);
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipMemcpy(h_rgb, d, (size_t) opt->width * opt->height * 3, hipMemcpyDeviceToHost));
	return SKR_OK;
}

// The partition, for callers and tests: tiles per rank (padded) and the (rank, slot) of a row's tile.
uint32_t skr_shard_tiles_per_rank(int32_t height, uint32_t tile_rows, uint32_t world)
{
	return (height > 0 && tile_rows && world) ? tiles_per_rank(height, tile_rows, world) : 0;
}

// Host-side de-interleave of a rank-major gathered buffer (what the device kernel does), for tests and for callers that
// gathered by other means.
int skr_shard_deinterleave_host(const uint8_t *gathered, uint8_t *frame, int32_t width, int32_t height, uint32_t tile_rows, uint32_t world)
{
	if(!gathered || !frame || width <= 0 || height <= 0 || !tile_rows || !world) return SKR_ERR_ARG;
	const size_t row = (size_t) width * 3, k_max = tiles_per_rank(height, tile_rows, world);
	for(uint32_t y = 0; y < (uint32_t) height; y++)
	{
		const uint32_t t = y / tile_rows, rank = t % world, k = t / world;
		memcpy(frame + (size_t) y * row, gathered + ((size_t) rank * k_max * tile_rows + (size_t) k * tile_rows + (y - t * tile_rows)) * row, row);
	}
	return SKR_OK;
}

} // extern "C"
