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

#include <algorithm>
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

// The tile -> (rank, slot) map: slot_of_tile[t] = rank * k_max + k; every rank has k_max = ceil(T / G) slots (the all-gather moves
// equal chunks).  Tiles differ in cost by an order of magnitude (sky rows: one ray per pixel; ground rows: a tree of up to
// 1 + N + N^2 rays under every pixel).  Two ways to deal them: blindly, tile t to rank t mod G (the tiles are a few rows high, so every
// rank gets a sample of the whole image), and by cost — every tile's own counted work (skr_tile_costs renders each tile once and reads
// the work counters, once per frame geometry), dealt longest-processing-time-first: most expensive tile first, each to the rank with
// the least work so far that still has a free slot.  Both are deterministic (integer counts of a bit-reproducible render, a stable
// sort, ties to the lower index), so every rank of a job computes the same map without talking to the others, and the image cannot
// change with the map: the RNG is keyed by the global pixel.  shard_rule() below says which one a frame step takes.
void shard_interleaved(uint32_t T, uint32_t world, uint32_t *slot_of_tile)
{
	const uint32_t k_max = (T + world - 1) / world;
	for(uint32_t t = 0; t < T; t++) slot_of_tile[t] = (t % world) * k_max + t / world;
}

void shard_lpt(const uint64_t *cost, uint32_t T, uint32_t world, uint32_t *slot_of_tile)
{
	const uint32_t k_max = (T + world - 1) / world;
	std::vector<uint32_t> order(T);
	for(uint32_t t = 0; t < T; t++) order[t] = t;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
	std::vector<uint64_t> load(world, 0);
	std::vector<uint32_t> used(world, 0);
	for(uint32_t i = 0; i < T; i++)
	{
		const uint32_t t = order[i];
		uint32_t best = world;
		for(uint32_t r = 0; r < world; r++)
			if(used[r] < k_max && (best == world || load[r] < load[best])) best = r;
		slot_of_tile[t] = best; // (the rank for now)
		used[best]++;
		load[best] += cost[t];
	}
	// a rank's tiles sit in its slots in image order, as under the blind map: the kernels cut a launch into regions of consecutive
	// blocks, and dealing the slots in LPT order (all the expensive tiles first) cost 4 % of a share's time
	std::fill(used.begin(), used.end(), 0u);
	for(uint32_t t = 0; t < T; t++)
	{
		const uint32_t r = slot_of_tile[t];
		slot_of_tile[t] = r * k_max + used[r]++;
	}
}

// gathered: [world][k_max * tile_rows][row_bytes] (rank-major, what the all-gather leaves) -> frame[height][row_bytes]; row y of tile
// t = y / tile_rows lives in slot slot_of_tile[t].  One 16-byte word per thread where the rows allow it.
template <typename T>
__global__ __launch_bounds__(256) void skr_deinterleave_kernel(const T *gathered, T *frame, uint32_t height, uint32_t row_words, uint32_t tile_rows, const uint32_t *slot_of_tile)
{
	const uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x;
	if(i >= (uint64_t) height * row_words) return;
	const uint32_t y = (uint32_t) (i / row_words), x = (uint32_t) (i - (uint64_t) y * row_words);
	const uint32_t t = y / tile_rows;
	frame[i] = gathered[((uint64_t) slot_of_tile[t] * tile_rows + (y - t * tile_rows)) * row_words + x];
}

hipError_t launch_deinterleave(const uint8_t *gathered, uint8_t *frame, int32_t width, int32_t height, uint32_t tile_rows, const uint32_t *d_slot_of_tile, hipStream_t stream)
{
	const size_t row_bytes = (size_t) width * 3;
	if(row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(frame)) % 16 == 0)
	{
		const uint32_t rw = (uint32_t) (row_bytes / 16);
		const uint64_t n = (uint64_t) height * rw;
		hipLaunchKernelGGL(skr_deinterleave_kernel<uint4>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(gathered),
						   reinterpret_cast<uint4 *>(frame), (uint32_t) height, rw, tile_rows, d_slot_of_tile);
	}
	else
	{
		const uint64_t n = (uint64_t) height * row_bytes;
		hipLaunchKernelGGL(skr_deinterleave_kernel<uint8_t>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, gathered, frame, (uint32_t) height,
						   (uint32_t) row_bytes, tile_rows, d_slot_of_tile);
	}
	return hipGetLastError();
}

// The map of one frame geometry on one device: host copy, and on the device the rank's own tile list (what skr_render_tile_list
// takes) and slot_of_tile (what the de-interleave takes).  Rebuilt when the geometry, the tree or the scene behind it changes.
struct ShardMap {
	const skr_renderer *r = nullptr;
	int32_t width = 0, height = 0, monte_carlo = 0, gillum = 0, depth = 0;
	float fov = 0;
	uint32_t tile_rows = 0, world = 0, rank = 0, T = 0, k_max = 0;
	std::vector<uint32_t> slot_of_tile;
	uint32_t *d_tiles = nullptr;        // k_max entries: the tiles of `rank` in slot order (0xFFFFFFFF: an empty slot)
	uint32_t *d_slot_of_tile = nullptr; // T entries
};

// Which map a frame step takes.  Measured on the headline frame (tools/time_shard.py, profiles/r03_time_shard.txt): the blind map's
// slowest rank of 8 is 3 % above the mean, and LPT over the counted work does not beat it — ranks whose counted rays, hits and
// ray-sphere tests agree to 0.1 % still differ by 6 % in time when one of them holds the expensive tiles of ONE part of the image
// (what LPT deals first) and the other tiles from all over it.  So interleaving stays the rule, and the counted costs decide only
// whether it is safe: where the blind map's heaviest rank carries more than 1.10 of the mean cost (a frame whose expensive rows repeat
// with a period of G tiles; few tiles per rank), LPT takes over.  SKR_SHARD=interleave / lpt forces one or the other.
enum ShardRule { SHARD_AUTO, SHARD_INTERLEAVE, SHARD_LPT };
ShardRule shard_rule()
{
	const char *e = getenv("SKR_SHARD");
	if(e && !strcmp(e, "interleave")) return SHARD_INTERLEAVE;
	if(e && !strcmp(e, "lpt")) return SHARD_LPT;
	return SHARD_AUTO;
}

// true = the blind map leaves its heaviest rank above 1.10 of the mean cost
bool interleave_unbalanced(const uint64_t *cost, uint32_t T, uint32_t world)
{
	std::vector<uint64_t> load(world, 0);
	uint64_t total = 0;
	for(uint32_t t = 0; t < T; t++)
	{
		load[t % world] += cost[t];
		total += cost[t];
	}
	const uint64_t heaviest = *std::max_element(load.begin(), load.end());
	return (long double) heaviest * world * 100 > (long double) total * 110;
}

void shard_by_cost(const uint64_t *cost, uint32_t T, uint32_t world, ShardRule rule, uint32_t *slot_of_tile)
{
	if(rule == SHARD_LPT || (rule == SHARD_AUTO && interleave_unbalanced(cost, T, world))) shard_lpt(cost, T, world, slot_of_tile);
	else shard_interleaved(T, world, slot_of_tile);
}

int plan_map(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint32_t world, uint32_t *slot_of_tile)
{
	const uint32_t T = tiles_total(opt->height, tile_rows);
	const ShardRule rule = shard_rule();
	if(world < 2 || rule == SHARD_INTERLEAVE)
	{
		shard_interleaved(T, world, slot_of_tile);
		return SKR_OK;
	}
	std::vector<uint64_t> cost(T);
	const int rc = skr_tile_costs(r, opt, tile_rows, cost.data());
	if(rc != SKR_OK) return rc;
	shard_by_cost(cost.data(), T, world, rule, slot_of_tile);
	return SKR_OK;
}

void free_map(ShardMap &m)
{
	if(m.d_tiles) (void) hipFree(m.d_tiles);
	if(m.d_slot_of_tile) (void) hipFree(m.d_slot_of_tile);
	m = ShardMap();
}

// `shared`: a map already computed for this geometry on another device of the same process (skr_multi): only uploaded here
int ensure_map(ShardMap &m, skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint32_t world, uint32_t rank, const std::vector<uint32_t> *shared)
{
	const bool same = m.r == r && m.width == opt->width && m.height == opt->height && m.fov == opt->fov && m.tile_rows == tile_rows && m.world == world && m.rank == rank &&
					  m.monte_carlo == opt->monte_carlo && m.gillum == opt->num_path_traces && m.depth == opt->max_depth && m.d_tiles;
	if(same && !shared) return SKR_OK;
	if(same && shared && *shared == m.slot_of_tile) return SKR_OK;
	free_map(m);
	m.r = r;
	m.width = opt->width;
	m.height = opt->height;
	m.fov = opt->fov;
	m.monte_carlo = opt->monte_carlo;
	m.gillum = opt->num_path_traces;
	m.depth = opt->max_depth;
	m.tile_rows = tile_rows;
	m.world = world;
	m.rank = rank;
	m.T = tiles_total(opt->height, tile_rows);
	m.k_max = tiles_per_rank(opt->height, tile_rows, world);
	if(shared) m.slot_of_tile = *shared;
	else
	{
		m.slot_of_tile.assign(m.T, 0);
		const int rc = plan_map(r, opt, tile_rows, world, m.slot_of_tile.data());
		if(rc != SKR_OK) return rc;
	}
	std::vector<uint32_t> mine(m.k_max, 0xFFFFFFFFu);
	for(uint32_t t = 0; t < m.T; t++)
		if(m.slot_of_tile[t] / m.k_max == rank) mine[m.slot_of_tile[t] % m.k_max] = t;
	SKR_HIP(hipMalloc((void **) &m.d_tiles, (size_t) m.k_max * sizeof(uint32_t)));
	SKR_HIP(hipMalloc((void **) &m.d_slot_of_tile, (size_t) m.T * sizeof(uint32_t)));
	SKR_HIP(hipMemcpy(m.d_tiles, mine.data(), (size_t) m.k_max * sizeof(uint32_t), hipMemcpyHostToDevice));
	SKR_HIP(hipMemcpy(m.d_slot_of_tile, m.slot_of_tile.data(), (size_t) m.T * sizeof(uint32_t), hipMemcpyHostToDevice));
	return SKR_OK;
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
	ShardMap map; // the tile -> rank map of the frame geometry last rendered
	// pipelined frames (skr_comm_render_frame_async): two buffer sets, the collective on a stream of its own
	RankBuffers abuf[2];
	hipStream_t cs = nullptr;
	hipEvent_t rendered[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
	bool in_flight[2] = {false, false};
	uint64_t async_frames = 0;
	// two frames in flight: odd frames of a run are rendered by a clone of the renderer (its own tables) on a stream of the communicator's
	skr_renderer *r2 = nullptr;
	hipStream_t rs2 = nullptr;
	hipEvent_t called = nullptr;
};

void skr_copy_switches(skr_renderer *dst, const skr_renderer *src); // api.cpp

// A rank's share of a frame is eight short dependent kernels: two frames in flight on two streams fill each other's ramps and
// tails — 1/8 of the headline frame 0.258 -> 0.217 ms per frame, 1/4 0.454 -> 0.427, 1/2 0.836 -> 0.800 (tools/inflight.py); a whole frame
// through this step, with the de-interleave behind it, 1.570 -> 1.513 ms.  So a run of frames alternates between the renderer and a clone of
// it (a second set of tables: 1.7 GB for the headline frame); SKR_INFLIGHT=1 keeps it to one.
static bool two_in_flight()
{
	if(const char *e = getenv("SKR_INFLIGHT")) return atoi(e) >= 2;
	return true;
}

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
	free_map(c->map);
	for(int k = 0; k < 2; k++)
	{
		free_buffers(c->abuf[k]);
		if(c->rendered[k]) (void) hipEventDestroy(c->rendered[k]);
		if(c->gathered[k]) (void) hipEventDestroy(c->gathered[k]);
	}
	if(c->cs) (void) hipStreamDestroy(c->cs);
	if(c->rs2) (void) hipStreamSynchronize(c->rs2);
	if(c->r2) skr_renderer_destroy(c->r2);
	if(c->rs2) (void) hipStreamDestroy(c->rs2);
	if(c->called) (void) hipEventDestroy(c->called);
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
	rc = ensure_map(c->map, c->r, opt, tile_rows, (uint32_t) c->world, (uint32_t) c->rank, nullptr);
	if(rc != SKR_OK) return rc;
	RankBuffers &b = c->buf;
	uint8_t *mine = b.d_gather + (size_t) c->rank * b.chunk;
	rc = skr_render_tile_list(c->r, opt, tile_rows, c->map.d_tiles, c->map.k_max, mine, nullptr, stream);
	if(rc != SKR_OK) return rc;
	if(c->comm) SKR_NCCL(rccl().AllGather(mine, b.d_gather, b.chunk, ncclUint8, c->comm, (hipStream_t) stream)); // in place: slot `rank` is the send buffer
	if(c->rank == 0) SKR_HIP(launch_deinterleave(b.d_gather, b.d_frame, opt->width, opt->height, tile_rows, c->map.d_slot_of_tile, (hipStream_t) stream));
	if(d_frame) *d_frame = c->rank == 0 ? b.d_frame : nullptr;
	return SKR_OK;
}

// The same frame step with the collective off the render stream: frame f's all-gather and de-interleave run on a stream of the
// communicator's own while frame f + 1 is rendered into the other of two buffer sets — on 8 GPUs the collective
// is a third of a 0.3 ms share, and nothing in the next frame depends on it.  The frames of a run also
// alternate between two renderers on two streams (two_in_flight above): throughput of a run, not the latency of a frame.  *d_prev_frame (rank 0): the frame of the
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
	hipStream_t rs = (hipStream_t) stream; // the stream this frame is rendered on
	skr_renderer *rr = c->r;
	if(s == 1 && two_in_flight())
	{ // odd frames: the clone, on the communicator's second render stream, behind whatever the caller's stream holds at this call
		if(!c->r2)
		{
			rc = skr_renderer_clone(c->r, &c->r2);
			if(rc != SKR_OK) return rc;
			SKR_HIP(hipStreamCreateWithFlags(&c->rs2, hipStreamNonBlocking));
			SKR_HIP(hipEventCreateWithFlags(&c->called, hipEventDisableTiming));
		}
		skr_copy_switches(c->r2, c->r);
		SKR_HIP(hipEventRecord(c->called, (hipStream_t) stream));
		SKR_HIP(hipStreamWaitEvent(c->rs2, c->called, 0));
		rs = c->rs2;
		rr = c->r2;
	}
	RankBuffers &b = c->abuf[s];
	if(c->in_flight[s])
	{ // the collective of frame f - 2 read these buffers (a change of geometry frees them: wait on the host then)
		if(b.d_gather && (b.width != opt->width || b.height != opt->height || b.tile_rows != tile_rows)) SKR_HIP(hipEventSynchronize(c->gathered[s]));
		else SKR_HIP(hipStreamWaitEvent(rs, c->gathered[s], 0));
		c->in_flight[s] = false;
	}
	rc = size_buffers(b, opt, tile_rows, (uint32_t) c->world, c->rank == 0);
	if(rc != SKR_OK) return rc;
	if(c->map.d_tiles && (c->map.width != opt->width || c->map.height != opt->height || c->map.tile_rows != tile_rows || c->map.fov != opt->fov ||
						  c->map.gillum != opt->num_path_traces || c->map.depth != opt->max_depth || c->map.monte_carlo != opt->monte_carlo))
		SKR_HIP(hipStreamSynchronize(c->cs)); // (a de-interleave in flight still reads the old map)
	rc = ensure_map(c->map, c->r, opt, tile_rows, (uint32_t) c->world, (uint32_t) c->rank, nullptr);
	if(rc != SKR_OK) return rc;
	uint8_t *mine = b.d_gather + (size_t) c->rank * b.chunk;
	rc = skr_render_tile_list(rr, opt, tile_rows, c->map.d_tiles, c->map.k_max, mine, nullptr, rs);
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipEventRecord(c->rendered[s], rs));
	SKR_HIP(hipStreamWaitEvent(c->cs, c->rendered[s], 0));
	if(c->comm) SKR_NCCL(rccl().AllGather(mine, b.d_gather, b.chunk, ncclUint8, c->comm, c->cs));
	if(c->rank == 0) SKR_HIP(launch_deinterleave(b.d_gather, b.d_frame, opt->width, opt->height, tile_rows, c->map.d_slot_of_tile, c->cs));
	SKR_HIP(hipEventRecord(c->gathered[s], c->cs));
	c->in_flight[s] = true;
	c->async_frames++;
	if(d_prev_frame)
	{ // the previous frame: whatever `stream` does from here on sees it whole (a caller that passes NULL does not look, and saves the wait)
		*d_prev_frame = nullptr;
		if(c->in_flight[prev])
		{
			SKR_HIP(hipStreamWaitEvent((hipStream_t) stream, c->gathered[prev], 0));
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
	std::vector<ShardMap> maps; // the tile -> rank map, computed on device 0 and uploaded to every device
	// pipelined frames (skr_multi_render_frame_async): two buffer sets per device, the collective on a stream of its own per device
	std::vector<RankBuffers> abufs[2];
	std::vector<hipStream_t> cstreams;
	std::vector<hipEvent_t> rendered[2], gathered[2];
	bool in_flight[2] = {false, false};
	uint64_t async_frames = 0;
	int async_set = -1; // >= 0: the workers render into abufs[async_set] (and order themselves behind its last collective)
	std::vector<skr_renderer *> renderers2; // two frames in flight (two_in_flight above): the odd frames of a run on a clone per device ...
	std::vector<hipStream_t> streams2;      // ... and a second stream
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
	const int set = m->async_set;
	RankBuffers &b = set >= 0 ? m->abufs[set][i] : m->bufs[i];
	int rc = SKR_OK;
	skr_renderer *rr = m->renderers[i];
	hipStream_t rs = m->streams[i];
	if(set == 1 && two_in_flight())
	{ // buffer set 1 always belongs to the clone and its stream (made by this device's own thread, once)
		if(!m->renderers2[i])
		{
			rc = skr_renderer_clone(m->renderers[i], &m->renderers2[i]);
			if(rc != SKR_OK) return rc;
			SKR_HIP(hipStreamCreateWithFlags(&m->streams2[i], hipStreamNonBlocking));
		}
		skr_copy_switches(m->renderers2[i], m->renderers[i]);
		rr = m->renderers2[i];
		rs = m->streams2[i];
	}
	if(set >= 0 && m->in_flight[set]) SKR_HIP(hipStreamWaitEvent(rs, m->gathered[set][i], 0)); // the collective of frame f - 2 read these buffers
	rc = size_buffers(b, m->opt, m->tile_rows, (uint32_t) m->n, i == 0);
	if(rc != SKR_OK) return rc;
	if(i != 0)
	{ // (device 0's map was made by the caller before the workers were woken: here it is only uploaded)
		rc = ensure_map(m->maps[i], m->renderers[i], m->opt, m->tile_rows, (uint32_t) m->n, (uint32_t) i, &m->maps[0].slot_of_tile);
		if(rc != SKR_OK) return rc;
	}
	if(i == 0 && set < 0) SKR_HIP(hipEventRecord(m->e0, m->streams[0]));
	rc = skr_render_tile_list(rr, m->opt, m->tile_rows, m->maps[i].d_tiles, m->maps[i].k_max, b.d_gather + (size_t) i * b.chunk, nullptr, rs);
	if(rc != SKR_OK) return rc;
	if(set >= 0) SKR_HIP(hipEventRecord(m->rendered[set][i], rs));
	return SKR_OK;
}

// every device renders its tiles of one frame (the calling thread drives device 0, the workers the others); returns when all are enqueued
int multi_render_all(skr_multi *m, const skr_options *opt, uint32_t tile_rows)
{
	SKR_HIP(hipSetDevice(m->devices[0]));
	int rc = ensure_map(m->maps[0], m->renderers[0], opt, tile_rows, (uint32_t) m->n, 0u, nullptr);
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
	return SKR_OK;
}

// one grouped all-gather of every device's chunk, each on the stream given for its device, then the root's de-interleave
int multi_collect(skr_multi *m, std::vector<RankBuffers> &bufs, const std::vector<hipStream_t> &on, const skr_options *opt, uint32_t tile_rows)
{
	if(m->comms[0])
	{
		SKR_NCCL(rccl().GroupStart());
		for(int i = 0; i < m->n; i++)
		{
			RankBuffers &b = bufs[i];
			const ncclResult_t ne = rccl().AllGather(b.d_gather + (size_t) i * b.chunk, b.d_gather, b.chunk, ncclUint8, m->comms[i], on[i]);
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
	SKR_HIP(launch_deinterleave(bufs[0].d_gather, bufs[0].d_frame, opt->width, opt->height, tile_rows, m->maps[0].d_slot_of_tile, on[0]));
	return SKR_OK;
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
	m->renderers2.assign(n_devices, nullptr);
	m->streams2.assign(n_devices, nullptr);
	m->comms.assign(n_devices, nullptr);
	m->bufs.resize(n_devices);
	m->maps.resize(n_devices);
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
		if(i < (int) m->cstreams.size() && m->cstreams[i]) (void) hipStreamSynchronize(m->cstreams[i]);
		free_buffers(m->bufs[i]);
		if(i < (int) m->maps.size()) free_map(m->maps[i]);
		for(int k = 0; k < 2; k++)
		{
			if(i < (int) m->abufs[k].size()) free_buffers(m->abufs[k][i]);
			if(i < (int) m->rendered[k].size() && m->rendered[k][i]) (void) hipEventDestroy(m->rendered[k][i]);
			if(i < (int) m->gathered[k].size() && m->gathered[k][i]) (void) hipEventDestroy(m->gathered[k][i]);
		}
		if(i < (int) m->cstreams.size() && m->cstreams[i]) (void) hipStreamDestroy(m->cstreams[i]);
		if(m->comms[i]) (void) rccl().CommDestroy(m->comms[i]);
		if(i < (int) m->streams2.size() && m->streams2[i]) (void) hipStreamSynchronize(m->streams2[i]);
		if(i < (int) m->renderers2.size() && m->renderers2[i]) skr_renderer_destroy(m->renderers2[i]);
		if(i < (int) m->streams2.size() && m->streams2[i]) (void) hipStreamDestroy(m->streams2[i]);
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
	m->async_set = -1;
	rc = multi_render_all(m, opt, tile_rows);
	if(rc != SKR_OK) return rc;
	rc = multi_collect(m, m->bufs, m->streams, opt, tile_rows); // every rank's chunk to every rank (the root is the one that uses it), each on its rank's stream behind its kernels
	if(rc != SKR_OK) return rc;
	SKR_HIP(hipEventRecord(m->e1, m->streams[0]));
	for(int i = m->n - 1; i >= 0; i--)
	{
		SKR_HIP(hipSetDevice(m->devices[i]));
		SKR_HIP(hipStreamSynchronize(m->streams[i]));
	}
	if(frame_ms) SKR_HIP(hipEventElapsedTime(frame_ms, m->e0, m->e1));
	if(d_frame) *d_frame = m->bufs[0].d_frame;
	return SKR_OK;
}

// The pipelined form (what skr_comm_render_frame_async is to skr_comm_render_frame): frame f's all-gather and de-interleave go to a
// second stream per device, behind an event the render stream records, while the render streams go on to frame f + 1 in the other of
// two buffer sets.  Returns as soon as frame f is enqueued; *d_prev_frame = the frame of the PREVIOUS call, complete (its collective
// is waited for on the host — it ran while this call's kernels were being enqueued), or NULL on the first call.
int skr_multi_render_frame_async(skr_multi *m, const skr_options *opt, uint32_t tile_rows, uint8_t **d_prev_frame)
{
	if(!m) return SKR_ERR_ARG;
	int rc = check_frame_args(opt, tile_rows);
	if(rc != SKR_OK) return rc;
	if(m->cstreams.empty())
	{
		m->cstreams.assign(m->n, nullptr);
		for(int k = 0; k < 2; k++)
		{
			m->abufs[k].resize(m->n);
			m->rendered[k].assign(m->n, nullptr);
			m->gathered[k].assign(m->n, nullptr);
		}
		for(int i = 0; i < m->n; i++)
		{
			SKR_HIP(hipSetDevice(m->devices[i]));
			SKR_HIP(hipStreamCreateWithFlags(&m->cstreams[i], hipStreamNonBlocking));
			for(int k = 0; k < 2; k++)
			{
				SKR_HIP(hipEventCreateWithFlags(&m->rendered[k][i], hipEventDisableTiming));
				SKR_HIP(hipEventCreateWithFlags(&m->gathered[k][i], hipEventDisableTiming));
			}
		}
	}
	const int s = (int) (m->async_frames & 1u), prev = s ^ 1;
	if(m->in_flight[s])
	{ // a change of geometry frees the buffers and the map the collective of frame f - 2 may still read: wait for it on the host then
		const RankBuffers &b = m->abufs[s][0];
		const ShardMap &mp = m->maps[0];
		if(b.width != opt->width || b.height != opt->height || b.tile_rows != tile_rows || mp.fov != opt->fov || mp.gillum != opt->num_path_traces ||
		   mp.depth != opt->max_depth || mp.monte_carlo != opt->monte_carlo)
		{
			for(int k = 0; k < 2; k++)
				for(int i = 0; i < m->n && m->in_flight[k]; i++) SKR_HIP(hipEventSynchronize(m->gathered[k][i]));
			m->in_flight[prev] = false; // (its frame is handed back below only if still in flight: it is gone with the old geometry)
		}
	}
	m->async_set = s;
	rc = multi_render_all(m, opt, tile_rows);
	m->async_set = -1;
	if(rc != SKR_OK) return rc;
	for(int i = 0; i < m->n; i++)
	{
		SKR_HIP(hipSetDevice(m->devices[i]));
		SKR_HIP(hipStreamWaitEvent(m->cstreams[i], m->rendered[s][i], 0));
	}
	rc = multi_collect(m, m->abufs[s], m->cstreams, opt, tile_rows);
	if(rc != SKR_OK) return rc;
	for(int i = 0; i < m->n; i++)
	{
		SKR_HIP(hipSetDevice(m->devices[i]));
		SKR_HIP(hipEventRecord(m->gathered[s][i], m->cstreams[i]));
	}
	m->in_flight[s] = true;
	m->async_frames++;
	if(d_prev_frame)
	{
		*d_prev_frame = nullptr;
		if(m->in_flight[prev])
		{
			SKR_HIP(hipEventSynchronize(m->gathered[prev][0]));
			*d_prev_frame = m->abufs[prev][0].d_frame;
		}
	}
	return SKR_OK;
}

// Ends a run of skr_multi_render_frame_async calls: waits for everything in flight; *d_frame = the last frame.
int skr_multi_flush(skr_multi *m, uint8_t **d_frame)
{
	if(!m) return SKR_ERR_ARG;
	if(d_frame) *d_frame = nullptr;
	if(m->async_frames == 0) return SKR_OK;
	for(int k = 0; k < 2; k++)
		for(int i = 0; i < m->n && m->in_flight[k]; i++) SKR_HIP(hipEventSynchronize(m->gathered[k][i]));
	if(d_frame) *d_frame = m->abufs[(m->async_frames - 1) & 1u][0].d_frame;
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

// The cost-aware map (longest processing time first; see shard_lpt above) for callers and tests: cost[t] per tile, any unit;
// slot_of_tile[t] = rank * k_max + slot.
int skr_shard_lpt(const uint64_t *cost, uint32_t n_tiles, uint32_t world, uint32_t *slot_of_tile)
{
	if(!cost || !slot_of_tile || !n_tiles || !world) return SKR_ERR_ARG;
	shard_lpt(cost, n_tiles, world, slot_of_tile);
	return SKR_OK;
}

// The map a frame step of `world` ranks uses for this renderer's scene and these options (plan_map above).  Synchronous; slot_of_tile has ceil(height / tile_rows) entries.
int skr_shard_plan(skr_renderer *r, const skr_options *opt, uint32_t tile_rows, uint32_t world, uint32_t *slot_of_tile)
{
	if(!r || !slot_of_tile || !world) return SKR_ERR_ARG;
	int rc = check_frame_args(opt, tile_rows);
	if(rc != SKR_OK) return rc;
	return plan_map(r, opt, tile_rows, world, slot_of_tile);
}

// The rule of the frame steps on given costs (tests): the blind map unless it leaves its heaviest rank above 1.10 of the mean cost.
int skr_shard_by_cost(const uint64_t *cost, uint32_t n_tiles, uint32_t world, uint32_t *slot_of_tile)
{
	if(!cost || !slot_of_tile || !n_tiles || !world) return SKR_ERR_ARG;
	shard_by_cost(cost, n_tiles, world, SHARD_AUTO, slot_of_tile);
	return SKR_OK;
}

// Host-side de-interleave of a rank-major gathered buffer under a map (what the device kernel does), for tests.
int skr_shard_deinterleave_map_host(const uint8_t *gathered, uint8_t *frame, int32_t width, int32_t height, uint32_t tile_rows, const uint32_t *slot_of_tile)
{
	if(!gathered || !frame || !slot_of_tile || width <= 0 || height <= 0 || !tile_rows) return SKR_ERR_ARG;
	const size_t row = (size_t) width * 3;
	for(uint32_t y = 0; y < (uint32_t) height; y++)
	{
		const uint32_t t = y / tile_rows;
		memcpy(frame + (size_t) y * row, gathered + ((size_t) slot_of_tile[t] * tile_rows + (y - t * tile_rows)) * row, row);
	}
	return SKR_OK;
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
