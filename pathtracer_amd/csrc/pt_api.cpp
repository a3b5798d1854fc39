// libptcore: context, device memory and the C ABI of include/pt_api.h.
// All computation on the render path happens in the HIP kernels of pt_kernels.hip; this file only
// prepares inputs (pt_host.cpp), moves memory and sequences launches.  There is no CPU fallback:
// without a working HIP device every entry point that renders returns PT_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pt_api.h"
#include "pt_host.h"
#include "pt_launch.h"

namespace {

std::string g_create_error;

// The static staging array of pt_cast_rays_noalloc exists once per GPU and is shared by every context on it: one lock
// per device ordinal serialises the calls that use it.
constexpr int kMaxDevices = 64;
std::mutex g_static_scratch_mu[kMaxDevices];

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	bool fresh = false;                  // set when reserve() had to allocate: the content is undefined until written
	hipError_t reserve(size_t bytes) {
		if(bytes <= cap) return hipSuccess;
		if(p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if(e != hipSuccess) return e; }
		size_t want = bytes < 256 ? 256 : bytes;
		hipError_t e = hipMalloc(&p, want);
		if(e == hipSuccess) { cap = want; fresh = true; }
		return e;
	}
	void release() { if(p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <typename T> T* as() const { return (T*)p; }
};

ptl::Vec vec(pt::f3 v) { return { v.x, v.y, v.z }; }

}  // namespace

struct pt_ctx {
	int device = 0;
	hipStream_t stream = nullptr, side = nullptr;
	std::string err;
	std::mutex mu;                       // guards err and the launch bookkeeping

	// inputs (deep copies)
	pt_world world{};
	pt_settings st{};
	std::vector<pt_object> objs;
	std::atomic<bool> prepared{ false };

	// device-resident scene
	DevBuf geo, cell_start, cell_items, cell_entries, obj_range, plane_ids, partial, fracs, bbox, grid_cursor, grid_sums, tile_cost, tile_order, dirs, disc, samples, offsets, frame_dev, fb, accum, counter, stats_dev, scratch;
	DevBuf thresholds;                   // the screenshot encoder's table
	bool thresholds_ready = false;
	void* dl_host = nullptr;             // pinned staging image of downloads into pageable memory
	size_t dl_host_bytes = 0;
	hipEvent_t dl_event[8] = {};
	void* snap_host = nullptr;           // pinned landing area of snapshots: two halves that take turns (poll thread only)
	hipEvent_t snap_event[2] = { nullptr, nullptr };
	ptl::TraceCaps caps{};               // compute units and resident workgroups per unit, of this context's device
	std::vector<hipStream_t> used_streams;   // caller streams pt_trace launched on since the last wait (the context's own stream is always waited for)
	void* static_scratch = nullptr;      // this device's copy of the library's static staging array (pt_cast_rays_noalloc)
	size_t static_scratch_bytes = 0;
	uint32_t* abort_flag = nullptr;      // host-pinned, read by the trace kernel
	uint32_t* abort_flag_dev = nullptr;  // the same word as the device addresses it
	uint32_t* progress_word = nullptr;   // host-pinned, PT_PROGRESS_WORD words behind the abort word: the trace kernel notes here how many work groups have been handed out
	uint64_t launches_since_readout = 0; // pt_trace calls since the statistics were last folded
	int counter_set = 0;                 // which of the two sets of group counters the next launch uses
	ptl::TraceJob job{};
	// what the polling thread may look at while the worker runs: a copy of the job of the launch in flight (under mu)
	ptl::TraceJob live_job{};
	bool live_valid = false;
	uint64_t live_paths = 0;
	size_t fb_floats = 0;
	bool grid_built_on_device = false;
	// random tables and blue-noise samples kept between renders (they depend on seed, spp_wanted, sample_grid_width)
	bool tables_valid = false;
	uint64_t tables_seed = 0;
	int tables_spp = 0, tables_grid_w = 0;
	std::vector<pt::f2> samples_host;
	std::vector<int32_t> offsets_host;
	uint32_t serial = 0;        // launch counter: marks the range sums each launch writes
	int local_tiles = 0;
	std::vector<uint16_t> tile_cost_host;   // the cost estimate of every owned tile, as the dealing order was made from it

	// launch bookkeeping
	std::vector<std::array<hipEvent_t, 3>> pending;   // events of launches not yet synchronised: start, trace done, resolve done
	std::vector<hipEvent_t> event_pool;
	double kernel_ms_sum = 0;
	int kernel_launches = 0;
	pt_stats stats{};

	// asynchronous render (pt_render_start)
	std::thread worker;
	std::atomic<int> active{ 0 }, done{ 0 }, aborted{ 0 }, status{ 0 };
	std::atomic<uint64_t> done_pixels{ 0 }, done_paths{ 0 };   // of the finished render: pixels and camera samples that count
	std::chrono::steady_clock::time_point t_start, t_end;
	float* user_rgb = nullptr;
	float fraction_shown = 0;            // poll thread: the largest fraction reported for the active render
};


// One frame on several GPUs of one process: a context per partition, the partitions' tile buffers copied device to
// device into the root's gathered buffer, assembled there and downloaded once.
struct pt_group {
	std::vector<pt_ctx*> part;
	std::string err;
	std::mutex mu;
	pt_settings st{};
	float* user_rgb = nullptr;
	DevBuf gathered, image;              // on the root's device (part[0]): partitions side by side, the assembled image
	// resident frames (pt_group_prepare / _trace): a ring of three tile buffers per partition (each on its partition's device)
	// and of three gathered buffers on the root's, so that the copies and the assemble of a frame overlap the next frames' traces
	static constexpr int kRing = 3;
	bool prepared = false;
	size_t stride_floats = 0;
	uint64_t frame = 0;                  // frames queued since pt_group_prepare
	std::vector<std::array<DevBuf, 3>> tiles;
	DevBuf ring[3];
	std::vector<std::array<hipEvent_t, 3>> copied;   // partition p's tiles of ring slot b have landed in the root's gathered buffer
	hipEvent_t assembled[3] = { nullptr, nullptr, nullptr };   // the root has laid out ring slot b as the image
	std::thread worker;
	std::atomic<int> active{ 0 }, done{ 0 }, aborted{ 0 }, status{ 0 }, abort_requested{ 0 };
	std::atomic<uint64_t> done_pixels{ 0 }, done_paths{ 0 };
	uint64_t total_pixels = 0, total_paths = 0;
	std::chrono::steady_clock::time_point t_start, t_end;
};

namespace {

int fail(pt_ctx* c, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if(c) { std::lock_guard<std::mutex> l(c->mu); c->err = buf; }
	else g_create_error = buf;
	return code;
}

#define PT_HIP(c, call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return fail((c), PT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while(0)

hipEvent_t take_event(pt_ctx* c) {
	if(!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if(hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}

int validate(pt_ctx* c, const pt_world* w, const pt_object* objs, int n, const pt_settings* s) {
	if(!w || !s || (n > 0 && !objs) || n < 0) return fail(c, PT_ERR_ARG, "null or negative argument");
	if(s->width < 1 || s->height < 1 || s->width > 16384 || s->height > 16384) return fail(c, PT_ERR_LIMIT, "image size %dx%d outside 1..16384", s->width, s->height);
	if(s->spp_wanted < 1) return fail(c, PT_ERR_ARG, "spp_wanted must be >= 1");
	if(s->sample_grid_width < 1 || s->sample_grid_width > 4) return fail(c, PT_ERR_LIMIT, "sample_grid_width %d outside 1..4", s->sample_grid_width);
	if(s->max_bounces < 1) return fail(c, PT_ERR_ARG, "max_bounces must be >= 1");
	if(s->grid_hint < 1 || s->grid_hint > 512) return fail(c, PT_ERR_LIMIT, "grid_hint %d outside 1..512", s->grid_hint);
	if(s->seed_policy != PT_SEED_PER_SAMPLE) return fail(c, PT_ERR_ARG, "the device path implements seed policy PT_SEED_PER_SAMPLE only");
	if(s->part_count < 1 || s->part_index < 0 || s->part_index >= s->part_count) return fail(c, PT_ERR_ARG, "bad partition %d of %d", s->part_index, s->part_count);
	if(s->frame_index < -1) return fail(c, PT_ERR_ARG, "frame_index must be >= -1");
	if(pt_sample_ranges(s) < 0) return fail(c, PT_ERR_ARG, "sample_ranges must be 0 or a power of two up to 16, not %d", s->sample_ranges);
	for(int i = 0; i < n; i++) {
		if(objs[i].type < PT_GEOM_SPHERE || objs[i].type > PT_GEOM_PLANE)
			return fail(c, PT_ERR_ARG, "object %d: unknown geometry type %d (sphere, box, cylinder, cone, plane)", i, objs[i].type);
	}
	return PT_OK;
}

int upload(pt_ctx* c, DevBuf& b, const void* src, size_t bytes) {
	PT_HIP(c, b.reserve(bytes));
	if(bytes) PT_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
	return PT_OK;
}

// The grid as the device passes see it, with the memory order of its cells chosen (ptl::grid_layout).
ptl::GridDev grid_dev(const pt::GridDesc& gd) {
	ptl::GridDev g{};
	for(int k = 0; k < 3; k++) { g.cmin[k] = gd.coord_min[k]; g.cnt[k] = gd.cell_count[k]; g.cs[k] = gd.cell_size[k]; g.inv_cs[k] = 1.0f / (float)gd.cell_size[k]; }
	ptl::grid_layout(g);
	return g;
}

// Per-cell lists between the reference's cell order (x fastest: what pt::build_grid writes and pt_debug_grid reports) and the
// device's (ptl::grid_layout).  to_device: `start`, `items` come in the reference's order and leave in the device's; else the reverse.
void reorder_lists(const ptl::GridDev& g, bool to_device, std::vector<uint32_t>* start, std::vector<uint32_t>* items) {
	if(ptl::grid_layout_is_reference(g)) return;
	const size_t cells = (size_t)g.cnt[0] * (size_t)g.cnt[1] * (size_t)g.cnt[2];
	std::vector<uint32_t> new_start(cells + 1, 0), new_items(items->size());
	// the same cell's index in the other order
	auto other = [&](size_t cell, bool cell_is_device) {
		uint32_t at[3];
		if(cell_is_device) { ptl::grid_cell_coords(g, cell, at); return ((size_t)at[2] * (size_t)g.cnt[1] + (size_t)at[1]) * (size_t)g.cnt[0] + (size_t)at[0]; }
		at[0] = (uint32_t)(cell % (size_t)g.cnt[0]); at[1] = (uint32_t)((cell / (size_t)g.cnt[0]) % (size_t)g.cnt[1]); at[2] = (uint32_t)(cell / ((size_t)g.cnt[0] * (size_t)g.cnt[1]));
		return ptl::grid_cell(g, (int)at[0], (int)at[1], (int)at[2]);
	};
	for(size_t o = 0; o < cells; o++) {
		const size_t in = other(o, to_device);
		const uint32_t a = (*start)[in], b = (*start)[in + 1];
		new_start[o + 1] = new_start[o] + (b - a);
		for(uint32_t k = a; k < b; k++) new_items[new_start[o] + (k - a)] = (*items)[k];
	}
	start->swap(new_start);
	items->swap(new_items);
}

// Per-cell lists, entity cell ranges and the cooperative cast's list entries built on the device
// (pt_gridbuild.hip) from the uploaded entity records.  *done is false when a cell lists more than
// kLongestDeviceList entities: the ordering pass sorts a list with one thread, so the host builds then.
constexpr uint32_t kLongestDeviceList = 4096;
constexpr size_t kSnapHalfBytes = 8u << 20;      // one half of the pinned landing area of pt_render_snapshot
// PT_CELL_CULL=1 leaves entities out of the cells their own bounding sphere does not reach (ptl::reaches_cell): 7 - 24 % fewer
// list entries and about 1 % less time on the grid frames, but NOT the reference's results - about one pixel in a million
// differs in its last bits.  The intersectors compute a hit point from far away, or at a grazing angle, some way along the ray
// from the entity's true surface (the quadratic's cancellation), the reference accepts it in whatever listed cell holds the
// computed point, and that cell can be one the entity itself stays clear of.  Off unless asked for; read at every prepare.
bool cell_cull() {
	const char* v = getenv("PT_CELL_CULL");
	return v != nullptr && v[0] != '\0' && v[0] != '0';
}

int grid_build_device(pt_ctx* c, const std::vector<float>& rec, int n, const pt::GridDesc& gd, size_t* item_total, bool* done) {
	*done = false;
	const size_t cells = (size_t)gd.cells();
	std::vector<float> box((size_t)n * 8, 0.0f);         // per entity: box extents | filter radius, filter centre | 0
	for(int i = 0; i < n; i++) {
		memcpy(box.data() + (size_t)i * 8, rec.data() + (size_t)i * PT_F_STRIDE + PT_F_BBOX, 12);
		box[(size_t)i * 8 + 3] = pt::filter_sphere(rec.data() + (size_t)i * PT_F_STRIDE, box.data() + (size_t)i * 8 + 4);
	}
	int rc;
	if((rc = upload(c, c->bbox, box.data(), box.size() * 4))) return rc;
	PT_HIP(c, c->cell_start.reserve((cells + 1) * 4));
	PT_HIP(c, c->grid_cursor.reserve(cells * 4));
	PT_HIP(c, c->grid_sums.reserve(ptl::grid_scan_blocks(cells) * 4 + 256));
	PT_HIP(c, c->obj_range.reserve((size_t)n * 8));
	ptl::GridBuildJob job{};
	job.geo = c->geo.as<float4>();
	job.bbox = c->bbox.as<float4>();
	job.n = n;
	job.grid = grid_dev(gd);
	job.obj_range = c->obj_range.as<uint32_t>();
	job.cell_start = c->cell_start.as<uint32_t>();
	job.cursor = c->grid_cursor.as<uint32_t>();
	job.block_sums = c->grid_sums.as<uint32_t>();
	job.longest = c->grid_sums.as<uint32_t>() + ptl::grid_scan_blocks(cells);
	job.cull = cell_cull() ? 1 : 0;
	PT_HIP(c, ptl::launch_grid_count_scan(job, cells, c->stream));
	uint32_t total = 0, longest = 0;
	PT_HIP(c, hipMemcpyAsync(&total, job.cell_start + cells, 4, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipMemcpyAsync(&longest, job.longest, 4, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	if(longest > kLongestDeviceList || total > ptl::CELL_START_MASK) return PT_OK;
	PT_HIP(c, c->cell_items.reserve(((size_t)total + 1) * 4));
	PT_HIP(c, c->cell_entries.reserve(((size_t)total + 1) * 16));
	job.cell_items = c->cell_items.as<uint32_t>();
	job.cell_entries = c->cell_entries.as<float4>();
	PT_HIP(c, ptl::launch_grid_fill(job, cells, c->stream));
	*item_total = total;
	*done = true;
	return PT_OK;
}

// pt_prepare body; the caller holds no lock and has set the device.
int prepare(pt_ctx* c) {
	auto t0 = std::chrono::steady_clock::now();
	const pt_settings& s = c->st;
	const int n = (int)c->objs.size();

	std::vector<float> rec((size_t)(n > 0 ? n : 1) * PT_F_STRIDE, 0.0f);
	pt::flatten_objects(c->objs.data(), n, rec.data());
	pt::WorldConsts wc;
	pt::world_consts(c->world, s.width, s.height, &wc);

	// tables, then blue noise, from PCG stream (0, seed): the order the application draws them in.
	// They depend on three settings only, so a context keeps them (host and device copies) from one
	// render to the next, as the application keeps its tables from start-up (code/app.cpp:669-727).
	const bool tables_kept = c->tables_valid && c->tables_seed == s.seed && c->tables_spp == s.spp_wanted && c->tables_grid_w == s.sample_grid_width;
	std::vector<pt::f2>& samples = c->samples_host;
	std::vector<int32_t>& offsets = c->offsets_host;
	std::vector<pt::f3> dirs;
	std::vector<pt::f2> disc;
	if(!tables_kept) {
		c->tables_valid = false;
		pt::Pcg rng;
		rng.seed(0, s.seed);
		dirs.resize(PT_TABLE_COUNT);
		disc.resize(PT_TABLE_COUNT);
		pt::make_tables(rng, dirs.data(), disc.data());
		pt::make_samples(rng, s.spp_wanted, s.sample_grid_width, &samples, &offsets);
	}
	for(size_t t = 0; t + 1 < offsets.size(); t++)
		if(offsets[t + 1] - offsets[t] > 4096) return fail(c, PT_ERR_LIMIT, "a pixel tile holds %d samples, more than the 4096 the per-sample streams allow", offsets[t + 1] - offsets[t]);

	pt::GridDesc gd{};
	std::vector<uint32_t> cell_start, cell_items;
	bool have_grid = pt::grid_desc(rec.data(), n, s.grid_hint, &gd);
	if(!have_grid) {
		if(n > 0) return fail(c, PT_ERR_LIMIT, "the scene's extent gives an unusable grid (non-finite bounds or more than 2^28 cells)");
		gd = pt::GridDesc{ { 0, 0, 0 }, { 1, 1, 1 }, { 1, 1, 1 } };
		cell_start.assign(2, 0);
	}
	// The per-cell lists are built on the device for large scenes (PT_GRID_BUILD=host|device overrides the
	// choice); the packed cell ranges the device passes use hold 10 bits per axis and 23 bits of entity id.
	const bool ranges_fit = gd.cell_count[0] <= 1023 && gd.cell_count[1] <= 1023 && gd.cell_count[2] <= 1023 && (uint32_t)n < ptl::ENTRY_ID_LIMIT;
	const char* where = getenv("PT_GRID_BUILD");
	bool grid_on_device = have_grid && ranges_fit && (where ? strcmp(where, "device") == 0 : n >= 4096);

	// one 128-byte record of eight float4 per entity (layout documented in pt_device.h)
	const size_t np = (size_t)(n > 0 ? n : 1);
	// ... and behind the records two more float4 per entity, read for cones only, by the trace variants that reproduce the
	// reference cone intersector's phantom hits (pt::cone_slopes)
	std::vector<float> geo(np * (ptl::REC_STRIDE + 2) * 4, 0.0f);
	int features = 0;          // what the entities need of a kernel variant: 1 unit-space kinds, 2 glass, 4 roughness, 8 planes (FEAT_* in pt_device.h), 32 cones
	std::vector<uint32_t> plane_ids;
	for(int i = 0; i < n; i++) {
		const float* r = rec.data() + (size_t)i * PT_F_STRIDE;
		int32_t type, flags;
		memcpy(&type, r + PT_F_TYPE, 4);
		memcpy(&flags, r + PT_F_FLAGS, 4);
		const bool direct = (type == PT_GEOM_SPHERE && (flags & 2)) || (type == PT_GEOM_BOX && !(flags & 1));
		const bool plane = type == PT_GEOM_PLANE;
		if(plane) { features |= 8; plane_ids.push_back((uint32_t)i); }
		else if(!direct) features |= 1;
		if(r[PT_F_IOR] != 1.0f) features |= 2;
		if(r[PT_F_ROUGH] != 0.0f) features |= 4;
		int32_t tf = type | (flags << 8);
		float tf_bits;
		memcpy(&tf_bits, &tf, 4);
		auto slot = [&](int p) { return geo.data() + ((size_t)i * ptl::REC_STRIDE + (size_t)p) * 4; };
		float* g0 = slot(ptl::REC_POS); g0[0] = r[PT_F_POS]; g0[1] = r[PT_F_POS + 1]; g0[2] = r[PT_F_POS + 2]; g0[3] = r[PT_F_BRADIUS];
		memcpy(slot(ptl::REC_ROT), r + PT_F_ROT, 16);
		memcpy(slot(ptl::REC_ROTINV), r + PT_F_ROTINV, 16);
		float* g3 = slot(ptl::REC_DIM); g3[0] = r[PT_F_DIM]; g3[1] = r[PT_F_DIM + 1]; g3[2] = r[PT_F_DIM + 2]; g3[3] = tf_bits;
		float* g4 = slot(ptl::REC_SCALE); g4[0] = r[PT_F_SCALE]; g4[1] = r[PT_F_SCALE + 1]; g4[2] = r[PT_F_SCALE + 2]; g4[3] = r[PT_F_ROUGH];
		float* g5 = slot(ptl::REC_SCALEINV); g5[0] = r[PT_F_SCALEINV]; g5[1] = r[PT_F_SCALEINV + 1]; g5[2] = r[PT_F_SCALEINV + 2]; g5[3] = 0;
		if(type == PT_GEOM_BOX && !(flags & 1)) {
			// an unrotated box reads neither its inverse rotation nor its inverse scale: the two slots hold the corners
			// lineBoxIntersection derives from position and size (code/raycast.cpp:420-426), in the same fp32 operations
			float* lo = slot(ptl::REC_ROTINV), *hi = slot(ptl::REC_SCALEINV);
			for(int k = 0; k < 3; k++) {
				const float half = r[PT_F_DIM + k] / 2;
				lo[k] = r[PT_F_POS + k] - half;
				hi[k] = r[PT_F_POS + k] + half;
			}
			lo[3] = 0; hi[3] = 0;
		}
		if(plane) {
			// a plane reads no inverse rotation: the slot holds its normal, rot * (0, 0, 1), which the reference works out at every test
			float* nrm = slot(ptl::REC_ROTINV);
			pt::plane_normal(r, nrm);
			nrm[3] = 0;
		}
		float* m0 = slot(ptl::REC_COLOR); m0[0] = r[PT_F_COLOR]; m0[1] = r[PT_F_COLOR + 1]; m0[2] = r[PT_F_COLOR + 2]; m0[3] = r[PT_F_REFL];
		float* m1 = slot(ptl::REC_EMIT); m1[0] = r[PT_F_EMIT]; m1[1] = r[PT_F_EMIT + 1]; m1[2] = r[PT_F_EMIT + 2]; m1[3] = r[PT_F_IOR];
		if(type == PT_GEOM_CONE) {
			features |= 32;
			pt::cone_slopes(r, geo.data() + (np * ptl::REC_STRIDE + 2 * (size_t)i) * 4);
		}
	}

	int rc;
	if((rc = upload(c, c->geo, geo.data(), geo.size() * 4))) return rc;
	if(plane_ids.empty()) plane_ids.push_back(0u);       // never read: the count below says so
	if((rc = upload(c, c->plane_ids, plane_ids.data(), plane_ids.size() * 4))) return rc;
	const int n_planes = (features & 8) ? (int)plane_ids.size() : 0;
	bool coop = false;
	size_t item_total = 0;
	c->grid_built_on_device = false;
	if(grid_on_device) {
		bool done = false;
		if((rc = grid_build_device(c, rec, n, gd, &item_total, &done))) return rc;
		grid_on_device = done;           // a very long list sends the build back to the host
		coop = done;
		c->grid_built_on_device = done;
	}
	if(!grid_on_device) {
		if(have_grid) pt::build_grid(rec.data(), n, s.grid_hint, &gd, &cell_start, &cell_items);
		const ptl::GridDev layout = grid_dev(gd);
		reorder_lists(layout, true, &cell_start, &cell_items);
		const bool cull = cell_cull();
		if(cull && have_grid) {
			// the lists without the entities that stay clear of the cell (in place: the write position never passes the read position)
			const int64_t all = gd.cells();
			size_t out = 0;
			for(int64_t cell = 0; cell < all; cell++) {
				uint32_t at[3];
				ptl::grid_cell_coords(layout, (size_t)cell, at);
				const uint32_t k0 = cell_start[(size_t)cell], k1 = cell_start[(size_t)cell + 1];
				cell_start[(size_t)cell] = (uint32_t)out;
				for(uint32_t k = k0; k < k1; k++) {
					float fc[3];
					const float fr = pt::filter_sphere(rec.data() + (size_t)cell_items[k] * PT_F_STRIDE, fc);
					if(ptl::reaches_cell(layout, fc[0], fc[1], fc[2], fr, (int)at[0], (int)at[1], (int)at[2])) cell_items[out++] = cell_items[k];
				}
			}
			cell_start[(size_t)all] = (uint32_t)out;
			cell_items.resize(out);
		}
		item_total = cell_items.size();
		if(item_total > ptl::CELL_START_MASK) return fail(c, PT_ERR_LIMIT, "the grid's cell lists hold %zu entries, more than the %u an index into them may address", item_total, ptl::CELL_START_MASK);
		if((rc = upload(c, c->cell_items, cell_items.data(), cell_items.size() * 4))) return rc;
		std::vector<uint32_t> ranges;
		coop = pt::object_ranges(rec.data(), n, gd, &ranges);
		if((rc = upload(c, c->obj_range, ranges.data(), ranges.size() * 4))) return rc;
		// What the cooperative cast's collection step needs of a list entry, in 16 bytes (ptl::pack_entry)
		std::vector<float> entries(coop ? (cell_items.size() + 1) * 4 : 4, 0.0f);
		const int64_t cells = coop ? gd.cells() : 0;
		for(int64_t cell = 0; cell < cells; cell++) {
			uint32_t at[3];
			ptl::grid_cell_coords(layout, (size_t)cell, at);
			uint32_t all_listed = 63u;       // per face: every entity of this list is listed across it too (ptl::CELL_FLAG_SHIFT)
			const uint32_t k0 = cell_start[(size_t)cell], k1 = cell_start[(size_t)cell + 1];
			for(uint32_t k = k0; k < k1; k++) {
				const uint32_t id = cell_items[k];
				const float* r = rec.data() + (size_t)id * PT_F_STRIDE;
				const uint32_t lo = ranges[(size_t)id * 2], hi = ranges[(size_t)id * 2 + 1];
				uint32_t tag = (lo >> 30) | ((hi >> 30) << 2);
				float fc[3];
				const float radius = pt::filter_sphere(r, fc);
				for(int a = 0; a < 3; a++) {
					// listed in the neighbour across this face: inside the entity's cell range, and not left out of that cell's list
					int nb[3] = { (int)at[0], (int)at[1], (int)at[2] };
					nb[a] = (int)at[a] - 1;
					if(at[a] > ((lo >> (10 * a)) & 1023u) && (!cull || ptl::reaches_cell(layout, fc[0], fc[1], fc[2], radius, nb[0], nb[1], nb[2]))) tag |= 1u << (3 + 2 * a);
					nb[a] = (int)at[a] + 1;
					if(at[a] < ((hi >> (10 * a)) & 1023u) && (!cull || ptl::reaches_cell(layout, fc[0], fc[1], fc[2], radius, nb[0], nb[1], nb[2]))) tag |= 1u << (4 + 2 * a);
				}
				uint32_t w[4];
				ptl::pack_entry(fc[0], fc[1], fc[2], radius, tag, id, w);
				memcpy(entries.data() + (size_t)k * 4, w, 16);
				all_listed &= tag >> 3;
			}
			// (in place: cell k0 is not read again, the next cell's k0 is this cell's k1, taken above before any flag lands on it)
			if(k1 > k0) cell_start[(size_t)cell] = k0 | (all_listed << ptl::CELL_FLAG_SHIFT);
		}
		if((rc = upload(c, c->cell_start, cell_start.data(), cell_start.size() * 4))) return rc;
		if((rc = upload(c, c->cell_entries, entries.data(), entries.size() * 4))) return rc;
	}
	if(!tables_kept) {
		std::vector<float> dirs4((size_t)PT_TABLE_COUNT * 4, 0.0f);
		for(int i = 0; i < PT_TABLE_COUNT; i++) { dirs4[(size_t)i * 4] = dirs[(size_t)i].x; dirs4[(size_t)i * 4 + 1] = dirs[(size_t)i].y; dirs4[(size_t)i * 4 + 2] = dirs[(size_t)i].z; }
		if((rc = upload(c, c->dirs, dirs4.data(), dirs4.size() * 4))) return rc;
		if((rc = upload(c, c->disc, disc.data(), disc.size() * sizeof(pt::f2)))) return rc;
		if((rc = upload(c, c->samples, samples.data(), samples.size() * sizeof(pt::f2)))) return rc;
		if((rc = upload(c, c->offsets, offsets.data(), offsets.size() * 4))) return rc;
		c->tables_seed = s.seed; c->tables_spp = s.spp_wanted; c->tables_grid_w = s.sample_grid_width;
		c->tables_valid = true;
	}

	// image tiling and this context's partition
	const int tiles_x = (s.width + 7) / 8, tiles_y = (s.height + 7) / 8;
	c->local_tiles = pt_partition_tiles(s.width, s.height, s.part_index, s.part_count);
	const bool compact = s.part_count > 1;
	c->fb_floats = compact ? (size_t)c->local_tiles * 64 * 3 : (size_t)s.width * s.height * 3;
	// Newly allocated image, accumulation and range-sum buffers start at zero: an aborted render leaves unfinished
	// pixels as they were, and a stale word of a previous owner of the memory must not pass for a launch serial.
	auto zeroed = [&](DevBuf& b, size_t bytes) -> hipError_t {
		hipError_t e = b.reserve(bytes);
		if(e == hipSuccess && b.fresh) { e = hipMemsetAsync(b.p, 0, b.cap, c->stream); b.fresh = false; }
		return e;
	};
	PT_HIP(c, zeroed(c->fb, c->fb_floats * 4));
	if(s.frame_index >= 0) PT_HIP(c, zeroed(c->accum, c->fb_floats * 4));
	PT_HIP(c, c->counter.reserve(8192));      // two sets of group counters, one cache line each; launches alternate between them
	PT_HIP(c, hipMemsetAsync(c->counter.p, 0, 8192, c->stream));
	c->counter_set = 0;
	PT_HIP(c, c->stats_dev.reserve(ptl::kStatBytes));
	// range sums of every pixel slot, and the quotients x/width, y/height (the divisions of code/raycast.cpp:1050)
	const int sum_ranges = pt_sample_ranges(&s);
	int ranges_log2 = 0;
	while((1 << ranges_log2) < sum_ranges) ranges_log2++;
	PT_HIP(c, zeroed(c->partial, (c->fb_floats / 3) * (size_t)sum_ranges * 16));
	{
		std::vector<float> fracs((size_t)s.width + (size_t)s.height);
		for(int x = 0; x < s.width; x++) fracs[(size_t)x] = (float)x / (float)s.width;
		for(int y = 0; y < s.height; y++) fracs[(size_t)s.width + (size_t)y] = (float)y / (float)s.height;
		if((rc = upload(c, c->fracs, fracs.data(), fracs.size() * 4))) return rc;
	}
	PT_HIP(c, hipMemsetAsync(c->stats_dev.p, 0, ptl::kStatBytes, c->stream));

	ptl::TraceJob& j = c->job;
	j = ptl::TraceJob{};
	j.scene.geo = c->geo.as<float4>();
	j.scene.cell_start = c->cell_start.as<uint32_t>();
	j.scene.cell_items = c->cell_items.as<uint32_t>();
	j.scene.obj_range = c->obj_range.as<uint32_t>();
	j.scene.cell_entries = c->cell_entries.as<float4>();
	j.scene.coop = coop ? 1 : 0;
	j.scene.features = features;
	j.scene.dirs = c->dirs.as<float4>();
	j.scene.disc = c->disc.as<float2>();
	j.scene.samples = c->samples.as<float2>();
	j.scene.offsets = c->offsets.as<int32_t>();
	j.scene.plane_ids = c->plane_ids.as<uint32_t>();
	j.scene.n_planes = n_planes;
	j.scene.n = n;
	j.scene.brute = n < s.brute_force_below ? 1 : 0;
	j.scene.grid = grid_dev(gd);

	// per-job constants of the tile job (code/raycast.cpp:1010-1021), evaluated once on the host
	ptl::FrameDev f{};
	auto mul = [](pt::f3 a, float b) { return pt::f3{ a.x * b, a.y * b, a.z * b }; };
	pt::f3 down = { -wc.cam_up.x, -wc.cam_up.y, -wc.cam_up.z };
	pt::f3 back = { -wc.cam_dir.x, -wc.cam_dir.y, -wc.cam_dir.z };
	f.cam_pos = vec(wc.cam_pos); f.cam_dir = vec(wc.cam_dir); f.right = vec(wc.cam_right); f.down = vec(down); f.back = vec(back);
	f.right_width = vec(mul(wc.cam_right, wc.dim_w));
	f.down_height = vec(mul(down, wc.dim_h));
	f.right_px = vec(mul(mul(wc.cam_right, wc.dim_w), (1.0f / s.width)));
	f.down_px = vec(mul(mul(down, wc.dim_h), (1.0f / s.height)));
	f.top_left = vec(wc.top_left);
	{
		float len = sqrtf(wc.light_dir.x * wc.light_dir.x + wc.light_dir.y * wc.light_dir.y + wc.light_dir.z * wc.light_dir.z);
		f.light_dir = { wc.light_dir.x / len, wc.light_dir.y / len, wc.light_dir.z / len };
	}
	f.bg = vec(wc.bg); f.ambient = vec(wc.ambient); f.light = vec(wc.light);
	f.focal_distance = wc.focal_distance; f.aperture = wc.aperture;
	f.width = s.width; f.height = s.height; f.tile_w = s.sample_grid_width;
	f.max_bounces = s.max_bounces;
	f.seed = s.seed;
	if((rc = upload(c, c->frame_dev, &f, sizeof f))) return rc;
	j.frame = c->frame_dev.as<ptl::FrameDev>();

	j.partial = c->partial.as<float4>();
	j.col_frac = c->fracs.as<float>();
	j.row_frac = c->fracs.as<float>() + s.width;
	j.ranges_log2 = ranges_log2;
	j.slots = c->fb_floats / 3;
	j.fb = c->fb.as<float>();
	j.accum = s.frame_index >= 0 ? c->accum.as<float>() : nullptr;
	j.group_counter = c->counter.as<uint32_t>();
	j.abort_flag = c->abort_flag_dev;
	j.stats = c->stats_dev.as<uint64_t>();
	j.tiles_x = tiles_x; j.tiles_y = tiles_y;
	j.n_groups = c->local_tiles * sum_ranges;
	j.part_index = s.part_index; j.part_count = s.part_count;
	j.frame_index = s.frame_index;
	j.compact = compact ? 1 : 0;

	// The order tiles are dealt in: dearest first by an estimate from sixteen probe paths per tile, so that
	// the frame's last work items are cheap ones (a frame that ends on its dearest tiles idles most of the
	// chip meanwhile).  Equal estimates go bottom-up: the lower rows of an image are usually the busier.
	j.tile_order = nullptr;
	c->tile_cost_host.clear();
	if(c->local_tiles > 1 && getenv("PT_TILE_ORDER_INDEX") == nullptr) {
		PT_HIP(c, c->tile_cost.reserve((size_t)c->local_tiles * 2));
		PT_HIP(c, ptl::launch_tile_cost(j, c->local_tiles, c->tile_cost.as<uint16_t>(), c->stream));
		std::vector<uint16_t>& cost = c->tile_cost_host;
		cost.assign((size_t)c->local_tiles, 0);
		PT_HIP(c, hipMemcpyAsync(cost.data(), c->tile_cost.p, cost.size() * 2, hipMemcpyDeviceToHost, c->stream));
		PT_HIP(c, hipStreamSynchronize(c->stream));
		std::vector<size_t> first(1025, 0);           // counting sort: descending cost, descending index within a cost
		for(uint16_t v : cost) first[1023 - (v > 1023 ? 1023 : v) + 1]++;
		for(int k = 1; k < 1025; k++) first[k] += first[k - 1];
		std::vector<int32_t> order(cost.size());
		for(size_t t = cost.size(); t-- > 0;) order[first[1023 - (cost[t] > 1023 ? 1023 : cost[t])]++] = (int32_t)t;
		if((rc = upload(c, c->tile_order, order.data(), order.size() * 4))) return rc;
		j.tile_order = c->tile_order.as<int32_t>();
	}
	{
		// what a group of this launch costs next to one of the demo scene with one sample per item (ptl::TraceJob::group_work)
		double casts_per_path = 0.5 * s.max_bounces;
		if(!c->tile_cost_host.empty()) {
			double sum = 0;
			for(uint16_t v : c->tile_cost_host) sum += v;
			casts_per_path = sum / (16.0 * (double)c->tile_cost_host.size());
		}
		const double samples_per_item = s.frame_index >= 0 ? 1.0 : std::max(1.0, (double)s.spp_wanted / (double)sum_ranges);
		j.group_work = (float)(samples_per_item * std::max(1.0, casts_per_path / 1.4) * std::max(1.0, 0.5 * n));
	}

	// exact path count of this partition (code/raycast.cpp:1045-1048 summed over its pixels)
	pt_stats& st = c->stats;
	st = pt_stats{};
	const int gw = s.sample_grid_width;
	uint64_t paths = 0, pixels = 0, empty = 0;
	auto tally = [&](int t, uint64_t count) {
		int cnt = offsets[(size_t)t + 1] - offsets[(size_t)t];
		pixels += count;
		if(cnt == 0) empty += count;
		else paths += count * (s.frame_index >= 0 ? 1u : (uint64_t)cnt);
	};
	if(s.part_count == 1) {
		// the whole image: pixels per blue-noise bucket = columns with x % gw == rx times rows with y % gw == ry
		for(int ry = 0; ry < gw; ry++) for(int rx = 0; rx < gw; rx++) {
			uint64_t cols = (uint64_t)((s.width - rx + gw - 1) / gw), rows = (uint64_t)((s.height - ry + gw - 1) / gw);
			if(rx < s.width && ry < s.height) tally(ry * gw + rx, cols * rows);
		}
	} else {
		for(int lt = 0; lt < c->local_tiles; lt++) {
			int tile = lt * s.part_count + s.part_index;
			int x0 = (tile % tiles_x) * 8, y0 = (tile / tiles_x) * 8;
			for(int yy = y0; yy < y0 + 8 && yy < s.height; yy++) for(int xx = x0; xx < x0 + 8 && xx < s.width; xx++)
				tally((yy % gw) * gw + (xx % gw), 1);
		}
	}
	st.paths = paths; st.pixels = pixels; st.zero_sample_pixels = empty;
	st.samples_total = (int)samples.size();
	st.objects = n;
	for(int k = 0; k < 3; k++) { st.cells[k] = gd.cell_count[k]; st.cell_size[k] = gd.cell_size[k]; }
	st.cell_items = (int)item_total;
	PT_HIP(c, hipStreamSynchronize(c->stream));
	st.prepare_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	c->prepared = true;
	return PT_OK;
}

int trace(pt_ctx* c, float* d_rgb, hipStream_t stream, bool sync, bool write_through) {
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_trace before pt_prepare");
	ptl::TraceJob job = c->job;
	if(d_rgb) job.fb = d_rgb;
	hipStream_t s = stream ? stream : c->stream;
	if(s != c->stream && std::find(c->used_streams.begin(), c->used_streams.end(), s) == c->used_streams.end()) c->used_streams.push_back(s);
	if(++c->serial >= 0x80000000u) c->serial = 1;
	job.serial = c->serial | (write_through ? 0x80000000u : 0u);
	// A progressive frame traced synchronously goes straight into the accumulation buffer and the image; an asynchronous render
	// keeps the range sums, which its snapshots and its abort read.  (PT_NO_PIXEL_STORE=1: range sums and the resolve pass, for A/B.)
	static const bool no_pixel_store = getenv("PT_NO_PIXEL_STORE") != nullptr;
	job.pixel_store = (job.frame_index >= 0 && job.accum != nullptr && !write_through && !no_pixel_store) ? 1 : 0;
	job.group_counter = c->counter.as<uint32_t>() + (size_t)c->counter_set * 1024;
	job.group_counter_next = c->counter.as<uint32_t>() + (size_t)(c->counter_set ^ 1) * 1024;
	*c->progress_word = 0;
	c->counter_set ^= 1;
	{ std::lock_guard<std::mutex> l(c->mu); c->live_job = job; c->live_paths = c->stats.paths; c->live_valid = true; }
	// HIP events around the two kernels give the statistics their durations, and cost a few microseconds of a frame
	// that may take 200: the first four launches after every read-out are timed, then one in eight
	// (PT_EVENT_EVERY=n changes the eight, 1 times every launch).
	static const int every = getenv("PT_EVENT_EVERY") ? atoi(getenv("PT_EVENT_EVERY")) : 8;
	bool timed = every <= 1 || (c->serial % (uint32_t)every) == 0;
	{
		// a caller that never reads the statistics out must not pile up events: past 256 unread launches none is timed
		std::lock_guard<std::mutex> l(c->mu);
		if(!timed) timed = c->pending.size() < 4;
		else if(c->pending.size() >= 256) timed = false;
	}
	hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
	if(timed) {
		e0 = take_event(c); e1 = take_event(c); e2 = take_event(c);
		if(!e0 || !e1 || !e2) return fail(c, PT_ERR_DEVICE, "hipEventCreate failed");
		PT_HIP(c, hipEventRecord(e0, s));
	}
	int blocks = 0;
	PT_HIP(c, ptl::launch_trace(job, c->caps, s, &blocks));
	if(timed) PT_HIP(c, hipEventRecord(e1, s));
	if(!job.pixel_store) PT_HIP(c, ptl::launch_resolve(job, c->fb_floats / 3, s));
	if(timed) {
		PT_HIP(c, hipEventRecord(e2, s));
		std::lock_guard<std::mutex> l(c->mu);
		c->pending.push_back({ e0, e1, e2 });
	}
	c->stats.grid_blocks = blocks;
	c->launches_since_readout++;
	if(sync) {
		PT_HIP(c, hipStreamSynchronize(s));
	}
	return PT_OK;
}

// Folds the events of finished launches into the statistics. The caller has synchronised.
int collect(pt_ctx* c) {
	std::vector<std::array<hipEvent_t, 3>> pend;
	{ std::lock_guard<std::mutex> l(c->mu); pend.swap(c->pending); }
	double sum = 0, whole = 0;
	float lo = 3.4e38f, hi = 0;
	for(auto& p : pend) {
		float ms = 0, both = 0;
		PT_HIP(c, hipEventSynchronize(p[2]));
		PT_HIP(c, hipEventElapsedTime(&ms, p[0], p[1]));       // the trace kernel
		PT_HIP(c, hipEventElapsedTime(&both, p[0], p[2]));     // trace + resolve
		sum += ms;
		whole += both;
		lo = ms < lo ? ms : lo;
		hi = ms > hi ? ms : hi;
		for(hipEvent_t e : p) c->event_pool.push_back(e);
	}
	if(!pend.empty()) {
		// on the library's stream: the caller has waited for every stream the launches went to
		uint64_t shards[ptl::kStatBytes / 8];
		PT_HIP(c, hipMemcpyAsync(shards, c->stats_dev.p, sizeof shards, hipMemcpyDeviceToHost, c->stream));
		PT_HIP(c, hipMemsetAsync(c->stats_dev.p, 0, ptl::kStatBytes, c->stream));
		PT_HIP(c, hipStreamSynchronize(c->stream));
		uint64_t rays = 0;
		for(int k = 0; k < ptl::kStatShards; k++) rays += shards[k * ptl::kStatStride];
		std::lock_guard<std::mutex> l(c->mu);             // pt_get_stats may be reading from another thread
		c->kernel_ms_sum = sum;
		c->kernel_launches = (int)pend.size();
		c->stats.trace_ms = (float)(whole / pend.size());
		c->stats.dominant_kernel_ms = (float)(sum / pend.size());
		c->stats.trace_ms_min = lo; c->stats.trace_ms_max = hi; c->stats.trace_launches = (int)pend.size();
		c->stats.rays = rays / (uint64_t)(c->launches_since_readout > 0 ? c->launches_since_readout : 1);   // the device counter adds up every launch, timed or not
		c->launches_since_readout = 0;
	}
	return PT_OK;
}

// Waits for every stream launches went to since the last wait: the context's own and each distinct caller stream (a
// caller that alternates between two streams of its own has work pending on both).
int wait_launches(pt_ctx* c) {
	PT_HIP(c, hipStreamSynchronize(c->stream));
	for(hipStream_t s : c->used_streams) {
		// a caller's stream that no longer exists has nothing left to wait for
		if(hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
	}
	c->used_streams.clear();
	return PT_OK;
}

// Copies pixels of an image in the context's device layout (row-major; tile-major [local tile][64][3] for a partition)
// into the caller's row-major image: all of the context's pixels, or with a mask only those whose byte is set.
// Pixels of other partitions and pixels without their mask byte are left alone.  Counts what it copied.
void place_pixels(const pt_ctx* c, const float* src, const uint8_t* mask, float* rgb, uint64_t* pixels, uint64_t* paths) {
	const pt_settings& s = c->st;
	const int gw = s.sample_grid_width;
	uint64_t n_pix = 0, n_paths = 0;
	auto samples_of = [&](int x, int y) -> uint64_t {
		const size_t t = (size_t)((y % gw) * gw + (x % gw));
		const int cnt = c->offsets_host[t + 1] - c->offsets_host[t];
		return s.frame_index >= 0 ? (cnt > 0 ? 1u : 0u) : (uint64_t)cnt;
	};
	if(s.part_count == 1) {
		for(int y = 0; y < s.height; y++) {
			const size_t row = (size_t)y * s.width;
			for(int x = 0; x < s.width; x++) {
				if(mask && !mask[row + x]) continue;
				memcpy(rgb + (row + x) * 3, src + (row + x) * 3, 12);
				n_pix++; n_paths += samples_of(x, y);
			}
		}
	} else {
		const int tiles_x = (s.width + 7) / 8;
		for(int lt = 0; lt < c->local_tiles; lt++) {
			const int tile = lt * s.part_count + s.part_index;
			const int x0 = (tile % tiles_x) * 8, y0 = (tile / tiles_x) * 8;
			for(int j = 0; j < 64; j++) {
				const int x = x0 + (j & 7), y = y0 + (j >> 3);
				if(x >= s.width || y >= s.height) continue;
				const size_t slot = (size_t)lt * 64 + j;
				if(mask && !mask[slot]) continue;
				memcpy(rgb + ((size_t)y * s.width + x) * 3, src + slot * 3, 12);
				n_pix++; n_paths += samples_of(x, y);
			}
		}
	}
	if(pixels) *pixels = n_pix;
	if(paths) *paths = n_paths;
}

// A whole image on the context's device to the caller.  Into pinned memory (pt_image_alloc, or a buffer the worker managed to pin) the copy
// engine writes directly, at the speed of the bus.  Into pageable memory the runtime's own copy is one thread moving
// bytes out of its staging buffers (17 GB/s measured); here the copy engine fills a pinned staging image of the
// context in segments and a few threads move each segment on as soon as it has landed.
int download_image(pt_ctx* c, const void* d_image, size_t bytes, float* rgb) {
	hipPointerAttribute_t attr;
	bool direct = bytes < (4u << 20);
	if(!direct) {
		direct = hipPointerGetAttributes(&attr, rgb) == hipSuccess && attr.type == hipMemoryTypeHost;
		(void)hipGetLastError();             // an ordinary malloc'ed pointer is reported as an error: that is the answer, not a failure
	}
	static const int threads_wanted = getenv("PT_DOWNLOAD_THREADS") ? atoi(getenv("PT_DOWNLOAD_THREADS")) : 4;
	if(direct || threads_wanted < 1) {
		PT_HIP(c, hipMemcpyAsync(rgb, d_image, bytes, hipMemcpyDeviceToHost, c->stream));
		PT_HIP(c, hipStreamSynchronize(c->stream));
		return PT_OK;
	}
	if(c->dl_host_bytes < bytes) {
		if(c->dl_host) { (void)hipHostFree(c->dl_host); c->dl_host = nullptr; c->dl_host_bytes = 0; }
		PT_HIP(c, hipHostMalloc(&c->dl_host, bytes, hipHostMallocDefault));
		c->dl_host_bytes = bytes;
	}
	constexpr int kSegments = 8;
	for(hipEvent_t& e : c->dl_event) if(!e) PT_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
	const size_t seg = ((bytes / kSegments) + 4095) & ~(size_t)4095;
	auto span = [&](int k, size_t* at, size_t* len) { *at = std::min(bytes, (size_t)k * seg); *len = std::min(seg, bytes - *at); };
	for(int k = 0; k < kSegments; k++) {
		size_t at, len;
		span(k, &at, &len);
		if(len) PT_HIP(c, hipMemcpyAsync((char*)c->dl_host + at, (const char*)d_image + at, len, hipMemcpyDeviceToHost, c->stream));
		PT_HIP(c, hipEventRecord(c->dl_event[k], c->stream));
	}
	const int n_threads = std::min(threads_wanted, kSegments);
	std::atomic<int> failed{ 0 };
	auto mover = [&](int t) {
		for(int k = t; k < kSegments; k += n_threads) {
			size_t at, len;
			span(k, &at, &len);
			if(hipEventSynchronize(c->dl_event[k]) != hipSuccess) { failed.store(1); return; }
			if(len) memcpy((char*)rgb + at, (const char*)c->dl_host + at, len);
		}
	};
	std::vector<std::thread> helpers;
	for(int t = 1; t < n_threads; t++) helpers.emplace_back(mover, t);
	mover(0);
	for(std::thread& h : helpers) h.join();
	if(failed.load()) return fail(c, PT_ERR_DEVICE, "waiting for the image copy failed: %s", hipGetErrorString(hipGetLastError()));
	return PT_OK;
}

int download(pt_ctx* c, float* rgb) {
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_download before pt_prepare");
	if(!rgb) return fail(c, PT_ERR_ARG, "null output");
	const pt_settings& s = c->st;
	if(s.part_count == 1) return download_image(c, c->fb.p, c->fb_floats * 4, rgb);
	// tile-major partition buffer -> the caller's row-major image; other partitions' pixels are left alone
	std::vector<float> tmp(c->fb_floats);
	PT_HIP(c, hipMemcpyAsync(tmp.data(), c->fb.p, c->fb_floats * 4, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	place_pixels(c, tmp.data(), nullptr, rgb, nullptr, nullptr);
	return PT_OK;
}

// After an aborted render: only the pixels that were finished go to the caller's image, the others keep what the
// caller had there (the reference's buffer after Escape, code/app.cpp:1425-1428: its workers stop between pixels).
int download_finished(pt_ctx* c, float* rgb, uint64_t* pixels, uint64_t* paths) {
	const size_t slots = c->fb_floats / 3;
	*pixels = 0; *paths = 0;
	if(slots == 0) return PT_OK;
	PT_HIP(c, c->scratch.reserve(slots));
	ptl::TraceJob job;
	{ std::lock_guard<std::mutex> l(c->mu); job = c->live_job; }
	PT_HIP(c, ptl::launch_finished(job, slots, c->scratch.as<uint8_t>(), c->stream));
	std::vector<float> img(c->fb_floats);
	std::vector<uint8_t> mask(slots);
	PT_HIP(c, hipMemcpyAsync(img.data(), c->fb.p, c->fb_floats * 4, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipMemcpyAsync(mask.data(), c->scratch.p, slots, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	place_pixels(c, img.data(), mask.data(), rgb, pixels, paths);
	return PT_OK;
}

void worker_main(pt_ctx* c) {
	int rc = PT_OK;
	static const bool times = getenv("PT_DEBUG_TIMES") != nullptr;      // phase times of an asynchronous render on stderr
	auto t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if(!times) return;
		auto t1 = std::chrono::steady_clock::now();
		fprintf(stderr, "  %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
		t0 = t1;
	};
	if(times) fprintf(stderr, "render: thread start %.3f ms\n", std::chrono::duration<double, std::milli>(t0 - c->t_start).count());
	if(hipSetDevice(c->device) != hipSuccess) rc = fail(c, PT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
	if(rc == PT_OK) rc = prepare(c);
	lap("prepare");
	if(rc == PT_OK) rc = trace(c, nullptr, nullptr, false, true);
	lap("launch");
	// While the device traces, the caller's image is pinned so that the copy at the end runs at the speed of the
	// bus: a copy into pageable memory goes through the runtime's staging buffers at a fifth of that (2.3 ms of a
	// 4.3 ms 1080p frame).  Worth it from a megabyte up; where pinning fails the pageable copy is used.
	const size_t bytes = (size_t)c->st.width * (size_t)c->st.height * 12;
	bool pinned = false;
	if(rc == PT_OK && c->st.part_count == 1 && bytes >= (1u << 20) && getenv("PT_NO_PIN") == nullptr) {
		hipError_t pe = hipHostRegister(c->user_rgb, bytes, hipHostRegisterDefault);
		pinned = pe == hipSuccess;
		if(times) fprintf(stderr, "  hipHostRegister: %s\n", hipGetErrorString(pe));
		if(!pinned) (void)hipGetLastError();
	}
	lap("pin");
	if(rc == PT_OK) rc = wait_launches(c);
	lap("wait");
	if(rc == PT_OK) rc = collect(c);
	lap("collect");
	const bool aborted = *c->abort_flag != 0;
	uint64_t pixels = c->stats.pixels, paths = c->stats.paths;
	if(rc == PT_OK) rc = aborted ? download_finished(c, c->user_rgb, &pixels, &paths) : download(c, c->user_rgb);
	lap("download");
	if(pinned && hipHostUnregister(c->user_rgb) != hipSuccess) (void)hipGetLastError();
	lap("unpin");
	c->done_pixels.store(rc == PT_OK ? pixels : 0);
	c->done_paths.store(rc == PT_OK ? paths : 0);
	c->aborted.store(aborted ? 1 : 0);
	c->t_end = std::chrono::steady_clock::now();
	c->status.store(rc);
	c->done.store(1);
}

void join_worker(pt_ctx* c) {
	if(c->worker.joinable()) c->worker.join();
	c->active.store(0);
}

}  // namespace

extern "C" {

pt_ctx* pt_create(int device) {
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if(e != hipSuccess || count <= 0) { fail(nullptr, PT_ERR_DEVICE, "no HIP device: %s", hipGetErrorString(e)); return nullptr; }
	if(device < 0 || device >= count) { fail(nullptr, PT_ERR_ARG, "device %d outside 0..%d", device, count - 1); return nullptr; }
	if((e = hipSetDevice(device)) != hipSuccess) { fail(nullptr, PT_ERR_DEVICE, "hipSetDevice: %s", hipGetErrorString(e)); return nullptr; }
	pt_ctx* c = new pt_ctx();
	c->device = device;
	bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
	          hipHostMalloc((void**)&c->abort_flag, 256, hipHostMallocMapped) == hipSuccess;
	if(!ok) { fail(nullptr, PT_ERR_DEVICE, "stream / pinned memory creation failed: %s", hipGetErrorString(hipGetLastError())); pt_destroy(c); return nullptr; }
	c->progress_word = c->abort_flag + PT_PROGRESS_WORD;
	*c->abort_flag = 0;
	*c->progress_word = 0;
	if(hipHostGetDevicePointer((void**)&c->abort_flag_dev, c->abort_flag, 0) != hipSuccess) c->abort_flag_dev = c->abort_flag;
	// with this context's device current: the address of that device's copy of the static staging array
	if((e = ptl::static_scratch(&c->static_scratch, &c->static_scratch_bytes)) != hipSuccess) { fail(nullptr, PT_ERR_DEVICE, "hipGetSymbolAddress: %s", hipGetErrorString(e)); pt_destroy(c); return nullptr; }
	if((e = ptl::query_trace_caps(&c->caps)) != hipSuccess) { fail(nullptr, PT_ERR_DEVICE, "occupancy query failed: %s", hipGetErrorString(e)); pt_destroy(c); return nullptr; }
	return c;
}

void pt_destroy(pt_ctx* c) {
	if(!c) return;
	if(c->active.load()) { *c->abort_flag = 1; join_worker(c); }
	(void)hipSetDevice(c->device);
	if(c->stream) (void)hipStreamSynchronize(c->stream);
	for(DevBuf* b : { &c->geo, &c->cell_start, &c->cell_items, &c->cell_entries, &c->obj_range, &c->plane_ids, &c->partial, &c->fracs, &c->bbox, &c->grid_cursor, &c->grid_sums, &c->tile_cost, &c->tile_order, &c->dirs, &c->disc, &c->samples, &c->offsets, &c->frame_dev, &c->fb, &c->accum, &c->counter, &c->stats_dev, &c->scratch, &c->thresholds }) b->release();
	if(c->snap_host) (void)hipHostFree(c->snap_host);
	for(auto& p : c->pending) for(hipEvent_t e : p) (void)hipEventDestroy(e);
	for(hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
	if(c->abort_flag) (void)hipHostFree(c->abort_flag);
	if(c->stream) (void)hipStreamDestroy(c->stream);
	if(c->side) (void)hipStreamDestroy(c->side);
	for(hipEvent_t e : c->snap_event) if(e) (void)hipEventDestroy(e);
	for(hipEvent_t e : c->dl_event) if(e) (void)hipEventDestroy(e);
	if(c->dl_host) (void)hipHostFree(c->dl_host);
	delete c;
}

const char* pt_last_error(const pt_ctx* c) {
	// a copy per calling thread: the worker of an asynchronous render may set a new message meanwhile
	thread_local std::string text;
	if(!c) { text = g_create_error; return text.c_str(); }
	{ std::lock_guard<std::mutex> l(const_cast<pt_ctx*>(c)->mu); text = c->err; }
	return text.c_str();
}

void pt_default_settings(pt_settings* s) {
	if(!s) return;
	memset(s, 0, sizeof *s);
	s->width = 768; s->height = 432;           // code/app.cpp:639
	s->spp_wanted = 100;                       // code/app.cpp:641
	s->sample_grid_width = 3;                  // code/app.cpp:642
	s->max_bounces = 10;                       // code/app.cpp:643
	s->grid_hint = 10;                         // code/app.cpp:1276
	s->brute_force_below = 10;                 // code/raycast.cpp:1100
	s->seed_policy = PT_SEED_PER_SAMPLE;
	s->seed = 12345;
	s->part_index = 0; s->part_count = 1;
	s->frame_index = -1;
}

int pt_default_scene(pt_world* w, pt_object* o, int capacity, int* count) {
	if(!w || !o || capacity < 2) return PT_ERR_ARG;
	memset(w, 0, sizeof *w);
	memset(o, 0, sizeof(pt_object) * 2);
	w->camera.pos[0] = 0; w->camera.pos[1] = -30; w->camera.pos[2] = 7;
	w->camera.fov = 90; w->camera.plane_width = 10;
	for(int k = 0; k < 3; k++) { w->background_color[k] = 1; w->ambient_color[k] = 1; w->light_color[k] = 1; }
	w->light_rot[0] = 0; w->light_rot[1] = 90;
	w->focal_distance = 30; w->aperture = 0;
	// ground box
	o[0].pos[2] = 0.1f; o[0].rot[0] = 1; o[0].dim[0] = 50; o[0].dim[1] = 50; o[0].dim[2] = 0.01f;
	o[0].color[0] = o[0].color[1] = o[0].color[2] = 0.5f; o[0].type = PT_GEOM_BOX; o[0].refractive_index = 1;
	// mirror sphere
	o[1].pos[2] = 7; o[1].rot[0] = 1; o[1].dim[0] = o[1].dim[1] = o[1].dim[2] = 8;
	o[1].color[0] = o[1].color[1] = o[1].color[2] = 0.7f; o[1].type = PT_GEOM_SPHERE; o[1].reflection_mod = 1; o[1].refractive_index = 1;
	if(count) *count = 2;
	return PT_OK;
}

static int store_inputs(pt_ctx* c, const pt_world* w, const pt_object* objs, int n, const pt_settings* s) {
	int rc = validate(c, w, objs, n, s);
	if(rc) return rc;
	c->world = *w;
	c->st = *s;
	c->objs.assign(objs, objs + n);
	c->prepared = false;
	return PT_OK;
}

int pt_prepare(pt_ctx* c, const pt_world* w, const pt_object* objs, int n, const pt_settings* s) {
	if(!c) return PT_ERR_ARG;
	if(c->active.load() && !c->done.load()) return fail(c, PT_ERR_BUSY, "a render is active");
	if(c->active.load()) join_worker(c);
	int rc = store_inputs(c, w, objs, n, s);
	if(rc) return rc;
	PT_HIP(c, hipSetDevice(c->device));
	*c->abort_flag = 0;
	return prepare(c);
}

int pt_trace(pt_ctx* c, float* d_rgb, void* hip_stream, int sync) {
	if(!c) return PT_ERR_ARG;
	if(c->active.load() && !c->done.load()) return fail(c, PT_ERR_BUSY, "a render is active");
	PT_HIP(c, hipSetDevice(c->device));
	*c->abort_flag = 0;          // an aborted pt_render_start leaves the word set; no asynchronous render is active here
	int rc = trace(c, d_rgb, (hipStream_t)hip_stream, sync != 0, false);
	if(rc == PT_OK && sync) rc = collect(c);
	return rc;
}

int pt_sync(pt_ctx* c) {
	if(!c) return PT_ERR_ARG;
	PT_HIP(c, hipSetDevice(c->device));
	int rc = wait_launches(c);
	return rc ? rc : collect(c);
}

int pt_set_frame(pt_ctx* c, int frame_index) {
	if(!c) return PT_ERR_ARG;
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_set_frame before pt_prepare");
	if(frame_index < 0 || c->st.frame_index < 0) return fail(c, PT_ERR_ARG, "progressive mode needs a scene prepared with frame_index >= 0");
	c->st.frame_index = frame_index;
	c->job.frame_index = frame_index;
	return PT_OK;
}

int pt_download(pt_ctx* c, float* rgb) {
	if(!c) return PT_ERR_ARG;
	PT_HIP(c, hipSetDevice(c->device));
	int rc = wait_launches(c);
	return rc ? rc : download(c, rgb);
}

int pt_get_stats(const pt_ctx* c, pt_stats* out) {
	if(!c || !out) return PT_ERR_ARG;
	// the worker of an asynchronous render folds its launch's events into the statistics under the same lock
	std::lock_guard<std::mutex> l(const_cast<pt_ctx*>(c)->mu);
	*out = c->stats;
	return PT_OK;
}

int pt_render_start(pt_ctx* c, const pt_world* w, const pt_object* objs, int n, const pt_settings* s, float* rgb) {
	if(!c) return PT_ERR_ARG;
	if(!rgb) return fail(c, PT_ERR_ARG, "null output buffer");
	if(c->active.load() && !c->done.load()) return fail(c, PT_ERR_BUSY, "a render is already active");
	if(c->active.load()) join_worker(c);
	int rc = store_inputs(c, w, objs, n, s);
	if(rc) return rc;
	c->user_rgb = rgb;
	*c->abort_flag = 0;
	{ std::lock_guard<std::mutex> l(c->mu); c->live_valid = false; }
	c->done.store(0); c->aborted.store(0); c->status.store(0);
	c->done_pixels.store(0); c->done_paths.store(0);
	c->fraction_shown = 0;
	*c->progress_word = 0;
	c->t_start = std::chrono::steady_clock::now();
	c->active.store(1);
	c->worker = std::thread(worker_main, c);
	return PT_OK;
}

// Work groups handed out so far, as a share: the trace kernel writes the number into a pinned host word every 64th
// group fetch, so a poll is one load.  (Nothing that runs on the device can be used for this: beside the persistent
// trace kernel no kernel starts before the frame ends, and the runtime copies a few bytes with a kernel.)
static void live_fraction(pt_ctx* c, float* fraction, uint64_t* paths_done) {
	int n_groups = 0;
	uint64_t paths = 0;
	{
		std::lock_guard<std::mutex> l(c->mu);
		if(c->live_valid) { n_groups = c->live_job.n_groups; paths = c->live_paths; }
	}
	if(n_groups <= 0) { *fraction = 0; *paths_done = 0; return; }
	const uint32_t got = *(volatile uint32_t*)c->progress_word;
	float f = (float)got / (float)n_groups;
	f = f > 1 ? 1 : f;
	if(f < c->fraction_shown) f = c->fraction_shown;         // wavefronts write in any order: never step back
	c->fraction_shown = f;
	*fraction = f;
	*paths_done = (uint64_t)((double)paths * f);
}

float* pt_image_alloc(pt_ctx* c, int width, int height) {
	if(!c) return nullptr;
	if(width < 1 || height < 1 || width > 16384 || height > 16384) { fail(c, PT_ERR_LIMIT, "image size %dx%d outside 1..16384", width, height); return nullptr; }
	if(hipSetDevice(c->device) != hipSuccess) { fail(c, PT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device); return nullptr; }
	void* p = nullptr;
	hipError_t e = hipHostMalloc(&p, (size_t)width * (size_t)height * 12, hipHostMallocDefault);
	if(e != hipSuccess) { fail(c, PT_ERR_NOMEM, "hipHostMalloc: %s", hipGetErrorString(e)); return nullptr; }
	return (float*)p;
}

void pt_image_free(pt_ctx* c, float* image) {
	(void)c;
	if(image) (void)hipHostFree(image);
}

int pt_render_progress(pt_ctx* c, pt_progress* info) {
	if(!c || !info) return PT_ERR_ARG;
	memset(info, 0, sizeof *info);
	if(!c->active.load()) return fail(c, PT_ERR_STATE, "no render was started");
	auto now = std::chrono::steady_clock::now();
	if(c->done.load()) {
		info->done = 1;
		info->aborted = c->aborted.load();
		const uint64_t pixels = c->stats.pixels;
		info->fraction = info->aborted ? (pixels ? (float)((double)c->done_pixels.load() / (double)pixels) : 0.0f) : 1.0f;
		info->seconds = std::chrono::duration<double>(c->t_end - c->t_start).count();
		info->paths_done = c->done_paths.load();
		return c->status.load();
	}
	info->seconds = std::chrono::duration<double>(now - c->t_start).count();
	live_fraction(c, &info->fraction, &info->paths_done);
	return PT_OK;
}

// The pixels of one context's frame in flight that are finished, copied into the caller's image.
static int snapshot_live(pt_ctx* c, float* rgb, uint64_t* pixels_done) {
	ptl::TraceJob job;
	bool valid;
	{ std::lock_guard<std::mutex> l(c->mu); job = c->live_job; valid = c->live_valid; }
	if(!valid || job.frame_index >= 0 || job.slots == 0) return PT_OK;      // still setting up; progressive frames are whole frames
	PT_HIP(c, hipSetDevice(c->device));
	// No kernel can run beside the persistent trace kernel (it holds every register file of the chip until the frame
	// ends; a compute unit kept free by a stream mask only serves the workgroups dealt to its own XCD).  So the copy
	// engine fetches the range sums, slice by slice into two pinned halves that take turns, and the host folds them with
	// the operations of pt_k_resolve: same order, IEEE division, same clamp - the same floats.
	const int ranges = 1 << job.ranges_log2;
	const size_t slots = job.slots;
	const size_t per_slice = std::max<size_t>(1, std::min(slots, kSnapHalfBytes / (16 * (size_t)ranges)));
	if(!c->snap_host) {
		PT_HIP(c, hipHostMalloc(&c->snap_host, 2 * kSnapHalfBytes, hipHostMallocDefault));
		for(hipEvent_t& e : c->snap_event) PT_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	const size_t slices = (slots + per_slice - 1) / per_slice;
	auto fetch = [&](size_t k) -> hipError_t {
		const size_t first = k * per_slice, count = std::min(per_slice, slots - first);
		char* to = (char*)c->snap_host + (k & 1) * kSnapHalfBytes;
		for(int r = 0; r < ranges; r++) {
			hipError_t e = hipMemcpyAsync(to + (size_t)r * count * 16, job.partial + (size_t)r * slots + first, count * 16, hipMemcpyDeviceToHost, c->side);
			if(e != hipSuccess) return e;
		}
		return hipEventRecord(c->snap_event[k & 1], c->side);
	};
	const pt_settings& s = c->st;
	const int gw = s.sample_grid_width, tiles_x = (s.width + 7) / 8;
	uint64_t pixels = 0;
	PT_HIP(c, fetch(0));
	for(size_t k = 0; k < slices; k++) {
		PT_HIP(c, hipEventSynchronize(c->snap_event[k & 1]));
		if(k + 1 < slices) PT_HIP(c, fetch(k + 1));
		const size_t first = k * per_slice, count = std::min(per_slice, slots - first);
		const float* sums = (const float*)((const char*)c->snap_host + (k & 1) * kSnapHalfBytes);
		for(size_t i = 0; i < count; i++) {
			bool whole = true;
			float sx = 0, sy = 0, sz = 0;
			for(int r = 0; r < ranges; r++) {
				const float* v = sums + ((size_t)r * count + i) * 4;
				uint32_t serial;
				memcpy(&serial, v + 3, 4);
				whole = whole && serial == job.serial;
				if(r == 0) { sx = v[0]; sy = v[1]; sz = v[2]; } else { sx = sx + v[0]; sy = sy + v[1]; sz = sz + v[2]; }
			}
			if(!whole) continue;
			const size_t slot = first + i;
			int x, y;
			if(job.compact) {
				const int tile = (int)(slot / 64) * s.part_count + s.part_index, j = (int)(slot % 64);
				x = (tile % tiles_x) * 8 + (j & 7); y = (tile / tiles_x) * 8 + (j >> 3);
				if(x >= s.width || y >= s.height) continue;
			} else { x = (int)(slot % (size_t)s.width); y = (int)(slot / (size_t)s.width); }
			const size_t t = (size_t)((y % gw) * gw + (x % gw));
			const int n = c->offsets_host[t + 1] - c->offsets_host[t];
			float* o = rgb + ((size_t)y * s.width + x) * 3;
			if(n == 0) { o[0] = 0; o[1] = 0; o[2] = 0; }
			else {
				const float d = (float)n, qx = sx / d, qy = sy / d, qz = sz / d;
				o[0] = qx > 1.0f ? 1.0f : qx; o[1] = qy > 1.0f ? 1.0f : qy; o[2] = qz > 1.0f ? 1.0f : qz;
			}
			pixels++;
		}
	}
	if(pixels_done) *pixels_done = pixels;
	return PT_OK;
}

int pt_render_snapshot(pt_ctx* c, float* rgb, uint64_t* pixels_done) {
	if(!c) return PT_ERR_ARG;
	if(!rgb) return fail(c, PT_ERR_ARG, "null output");
	if(pixels_done) *pixels_done = 0;
	if(!c->active.load()) return fail(c, PT_ERR_STATE, "no render was started");
	if(c->done.load()) {                       // the caller's image already holds every pixel that counts
		if(pixels_done) *pixels_done = c->done_pixels.load();
		return c->status.load();
	}
	return snapshot_live(c, rgb, pixels_done);
}

int pt_render_abort(pt_ctx* c) {
	if(!c) return PT_ERR_ARG;
	if(!c->active.load()) return fail(c, PT_ERR_STATE, "no render was started");
	*c->abort_flag = 1;     // the trace kernel stops fetching tiles; pixels already in flight finish
	return PT_OK;
}

int pt_render_wait(pt_ctx* c) {
	if(!c) return PT_ERR_ARG;
	if(!c->active.load()) return fail(c, PT_ERR_STATE, "no render was started");
	join_worker(c);
	c->active.store(1);     // keep the finished render pollable until the next start
	return c->status.load();
}

}  // extern "C"

/* ---- several GPUs, one process ------------------------------------------------------------------- */

namespace {

int group_fail(pt_group* g, int code, const std::string& text) {
	std::lock_guard<std::mutex> l(g->mu);
	if(g->err.empty() || code != PT_OK) g->err = text;
	return code;
}

// One partition: set-up, trace, then its tile buffer straight into the root's gathered buffer (device to device; over
// xGMI when the two contexts sit on different GPUs).  After an abort request the partition delivers its finished pixels
// through the host instead.  *gathered tells which way it went.
int group_part(pt_group* g, int p, size_t stride_floats, bool* gathered) {
	pt_ctx* c = g->part[(size_t)p];
	pt_ctx* root = g->part[0];
	const int n = (int)g->part.size();
	*gathered = false;
	if(hipSetDevice(c->device) != hipSuccess) return fail(c, PT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
	int rc = prepare(c);
	if(rc == PT_OK) rc = trace(c, nullptr, nullptr, false, true);
	if(rc == PT_OK) rc = wait_launches(c);
	if(rc == PT_OK) rc = collect(c);
	if(rc != PT_OK) return rc;
	uint64_t pixels = c->stats.pixels, paths = c->stats.paths;
	if(g->abort_requested.load()) {
		rc = download_finished(c, g->user_rgb, &pixels, &paths);
	} else if(n == 1) {
		rc = download(c, g->user_rgb);
	} else {
		float* to = g->gathered.as<float>() + (size_t)p * stride_floats;
		if(c->fb_floats > 0) {                 // a partition of a tiny image may own no tile at all
			PT_HIP(c, hipMemcpyPeerAsync(to, root->device, c->fb.p, c->device, c->fb_floats * 4, c->stream));
			PT_HIP(c, hipStreamSynchronize(c->stream));
		}
		*gathered = true;
	}
	if(rc == PT_OK) { g->done_pixels.fetch_add(pixels); g->done_paths.fetch_add(paths); }
	return rc;
}

void group_main(pt_group* g) {
	const int n = (int)g->part.size();
	pt_ctx* root = g->part[0];
	const pt_settings& s = g->st;
	int rc = PT_OK;
	size_t stride = 0;
	for(int p = 0; p < n; p++) stride = std::max(stride, (size_t)pt_partition_tiles(s.width, s.height, p, n) * 192);
	const size_t image_bytes = (size_t)s.width * (size_t)s.height * 12;
	if(n > 1) {
		hipError_t e = hipSetDevice(root->device);
		if(e == hipSuccess) e = g->gathered.reserve((size_t)n * stride * 4);
		if(e == hipSuccess) e = g->image.reserve(image_bytes);
		if(e != hipSuccess) rc = group_fail(g, PT_ERR_DEVICE, std::string("gather buffers: ") + hipGetErrorString(e));
	}
	std::vector<int> rcs((size_t)n, PT_OK);
	std::vector<char> went_by_device((size_t)n, 0);
	if(rc == PT_OK) {
		std::vector<std::thread> threads;
		for(int p = 0; p < n; p++) threads.emplace_back([g, p, stride, &rcs, &went_by_device]() {
			bool by_device = false;
			rcs[(size_t)p] = group_part(g, p, stride, &by_device);
			went_by_device[(size_t)p] = by_device ? 1 : 0;
		});
		for(std::thread& t : threads) t.join();
		for(int p = 0; p < n && rc == PT_OK; p++)
			if(rcs[(size_t)p] != PT_OK) rc = group_fail(g, rcs[(size_t)p], "partition " + std::to_string(p) + ": " + pt_last_error(g->part[(size_t)p]));
	}
	const bool aborted = g->abort_requested.load() != 0;
	if(rc == PT_OK && n > 1) {
		if(!aborted) {
			// every partition is in the gathered buffer: lay them out as the image and copy it out once
			if(hipSetDevice(root->device) != hipSuccess) rc = group_fail(g, PT_ERR_DEVICE, "hipSetDevice failed");
			if(rc == PT_OK && ptl::launch_assemble(g->gathered.as<float>(), stride, n, s.width, s.height, g->image.as<float>(), root->stream) != hipSuccess)
				rc = group_fail(g, PT_ERR_DEVICE, "pt_k_assemble launch failed");
			if(rc == PT_OK && (rc = download_image(root, g->image.p, image_bytes, g->user_rgb)) != PT_OK) group_fail(g, rc, pt_last_error(root));
		} else {
			// the request came while some partitions had already gone to the gathered buffer: those are complete, and
			// reach the caller's image through the host like the unfinished ones did
			for(int p = 0; p < n && rc == PT_OK; p++) {
				if(!went_by_device[(size_t)p]) continue;
				pt_ctx* c = g->part[(size_t)p];
				if(hipSetDevice(c->device) != hipSuccess) rc = group_fail(g, PT_ERR_DEVICE, "hipSetDevice failed");
				else if((rc = download(c, g->user_rgb)) != PT_OK) group_fail(g, rc, pt_last_error(c));
			}
		}
	}
	g->aborted.store(aborted ? 1 : 0);
	g->t_end = std::chrono::steady_clock::now();
	g->status.store(rc);
	g->done.store(1);
}

void group_join(pt_group* g) {
	if(g->worker.joinable()) g->worker.join();
	g->active.store(0);
}

}  // namespace

extern "C" {

namespace {

void group_release_ring(pt_group* g) {
	for(size_t p = 0; p < g->tiles.size(); p++) {
		(void)hipSetDevice(g->part[p]->device);
		for(DevBuf& b : g->tiles[p]) b.release();
		for(hipEvent_t& e : g->copied[p]) { if(e) (void)hipEventDestroy(e); e = nullptr; }
	}
	g->tiles.clear();
	g->copied.clear();
	if(!g->part.empty()) {
		(void)hipSetDevice(g->part[0]->device);
		for(DevBuf& b : g->ring) b.release();
		for(hipEvent_t& e : g->assembled) { if(e) (void)hipEventDestroy(e); e = nullptr; }
	}
	g->prepared = false;
}

// Waits for every partition's stream and the root's side stream, and folds the launches' events into the statistics.
int group_sync(pt_group* g) {
	for(pt_ctx* c : g->part) {
		if(hipSetDevice(c->device) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "hipSetDevice failed");
		int rc = wait_launches(c);
		if(rc == PT_OK) rc = collect(c);
		if(rc != PT_OK) return group_fail(g, rc, pt_last_error(c));
	}
	pt_ctx* root = g->part[0];
	if(hipSetDevice(root->device) != hipSuccess || hipStreamSynchronize(root->side) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "waiting for the root's assemble failed");
	return PT_OK;
}

}  // namespace

extern "C" {

int pt_group_prepare(pt_group* g, const pt_world* w, const pt_object* objs, int n_objects, const pt_settings* s) {
	if(!g) return PT_ERR_ARG;
	if(g->active.load() && !g->done.load()) return group_fail(g, PT_ERR_BUSY, "a render is active");
	if(g->active.load()) group_join(g);
	if(!s) return group_fail(g, PT_ERR_ARG, "null settings");
	const int n = (int)g->part.size();
	if(g->prepared) { int rc = group_sync(g); if(rc) return rc; }
	group_release_ring(g);
	for(int p = 0; p < n; p++) {
		pt_ctx* c = g->part[(size_t)p];
		if(c->active.load()) { if(!c->done.load()) return group_fail(g, PT_ERR_BUSY, "a partition's context is rendering on its own"); join_worker(c); }
		pt_settings sp = *s;
		sp.part_index = p;
		sp.part_count = n;
		int rc = store_inputs(c, w, objs, n_objects, &sp);
		if(rc) return group_fail(g, rc, pt_last_error(c));
		*c->abort_flag = 0;
	}
	{ std::lock_guard<std::mutex> l(g->mu); g->err.clear(); }
	g->st = *s;
	// every partition flattens, builds its grid and probes its tiles' cost on its own device, side by side
	std::vector<int> rcs((size_t)n, PT_OK);
	std::vector<std::thread> threads;
	for(int p = 0; p < n; p++) threads.emplace_back([g, p, &rcs]() {
		pt_ctx* c = g->part[(size_t)p];
		rcs[(size_t)p] = hipSetDevice(c->device) == hipSuccess ? prepare(c) : fail(c, PT_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
	});
	for(std::thread& t : threads) t.join();
	for(int p = 0; p < n; p++)
		if(rcs[(size_t)p] != PT_OK) return group_fail(g, rcs[(size_t)p], "partition " + std::to_string(p) + ": " + pt_last_error(g->part[(size_t)p]));
	size_t stride = 0;
	for(int p = 0; p < n; p++) stride = std::max(stride, (size_t)pt_partition_tiles(s->width, s->height, p, n) * 192);
	g->stride_floats = stride;
	g->tiles.resize((size_t)n);
	g->copied.resize((size_t)n);
	for(int p = 0; p < n; p++) {
		pt_ctx* c = g->part[(size_t)p];
		hipError_t e = hipSetDevice(c->device);
		for(int b = 0; b < pt_group::kRing && e == hipSuccess; b++) {
			g->copied[(size_t)p][(size_t)b] = nullptr;
			e = g->tiles[(size_t)p][(size_t)b].reserve(std::max<size_t>(c->fb_floats, 1) * 4);
			if(e == hipSuccess) e = hipEventCreateWithFlags(&g->copied[(size_t)p][(size_t)b], hipEventDisableTiming);
		}
		if(e != hipSuccess) { group_release_ring(g); return group_fail(g, PT_ERR_DEVICE, std::string("tile buffers: ") + hipGetErrorString(e)); }
	}
	pt_ctx* root = g->part[0];
	hipError_t e = hipSetDevice(root->device);
	for(int b = 0; b < pt_group::kRing && e == hipSuccess; b++) {
		e = g->ring[b].reserve(std::max<size_t>((size_t)n * stride, 1) * 4);
		if(e == hipSuccess) e = hipEventCreateWithFlags(&g->assembled[b], hipEventDisableTiming);
	}
	if(e == hipSuccess) e = g->image.reserve((size_t)s->width * (size_t)s->height * 12);
	if(e != hipSuccess) { group_release_ring(g); return group_fail(g, PT_ERR_DEVICE, std::string("gather buffers: ") + hipGetErrorString(e)); }
	g->frame = 0;
	g->prepared = true;
	return PT_OK;
}

int pt_group_set_frame(pt_group* g, int frame_index) {
	if(!g) return PT_ERR_ARG;
	if(!g->prepared) return group_fail(g, PT_ERR_STATE, "pt_group_set_frame before pt_group_prepare");
	for(pt_ctx* c : g->part) {
		int rc = pt_set_frame(c, frame_index);
		if(rc) return group_fail(g, rc, pt_last_error(c));
	}
	return PT_OK;
}

// One frame of the prepared scene.  Ring slot b = frame % 3: every partition traces into its tile buffer b and copies it, on
// its own stream, straight into the root's gathered buffer b (a peer copy: over xGMI between different GPUs, every partition on
// its own link); the root's side stream waits for the copies' events and lays the image out.  Nothing waits on the host: a
// partition's stream only waits, from the fourth frame on, for the assemble that read gathered buffer b three frames ago.
int pt_group_trace(pt_group* g, int sync) {
	if(!g) return PT_ERR_ARG;
	if(!g->prepared) return group_fail(g, PT_ERR_STATE, "pt_group_trace before pt_group_prepare");
	if(g->active.load() && !g->done.load()) return group_fail(g, PT_ERR_BUSY, "a render is active");
	const int n = (int)g->part.size();
	const int b = (int)(g->frame % pt_group::kRing);
	pt_ctx* root = g->part[0];
	for(int p = 0; p < n; p++) {
		pt_ctx* c = g->part[(size_t)p];
		if(hipSetDevice(c->device) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "hipSetDevice failed");
		if(g->frame >= (uint64_t)pt_group::kRing && hipStreamWaitEvent(c->stream, g->assembled[b], 0) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "hipStreamWaitEvent failed");
		*c->abort_flag = 0;
		int rc = trace(c, g->tiles[(size_t)p][(size_t)b].as<float>(), nullptr, false, false);
		if(rc != PT_OK) return group_fail(g, rc, pt_last_error(c));
		hipError_t e = hipSuccess;
		// (a group of one renders the whole image row by row: its buffer is the image, there is nothing to lay out)
		float* to = n == 1 ? g->image.as<float>() : g->ring[b].as<float>() + (size_t)p * g->stride_floats;
		if(c->fb_floats > 0) e = hipMemcpyPeerAsync(to, root->device, g->tiles[(size_t)p][(size_t)b].p, c->device, c->fb_floats * 4, c->stream);
		if(e == hipSuccess) e = hipEventRecord(g->copied[(size_t)p][(size_t)b], c->stream);
		if(e != hipSuccess) return group_fail(g, PT_ERR_DEVICE, std::string("gathering a partition: ") + hipGetErrorString(e));
	}
	if(hipSetDevice(root->device) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "hipSetDevice failed");
	for(int p = 0; p < n; p++)
		if(hipStreamWaitEvent(root->side, g->copied[(size_t)p][(size_t)b], 0) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "hipStreamWaitEvent failed");
	if(n > 1 && ptl::launch_assemble(g->ring[b].as<float>(), g->stride_floats, n, g->st.width, g->st.height, g->image.as<float>(), root->side) != hipSuccess)
		return group_fail(g, PT_ERR_DEVICE, "pt_k_assemble launch failed");
	if(hipEventRecord(g->assembled[b], root->side) != hipSuccess) return group_fail(g, PT_ERR_DEVICE, "hipEventRecord failed");
	g->frame++;
	return sync ? group_sync(g) : PT_OK;
}

int pt_group_sync(pt_group* g) {
	if(!g) return PT_ERR_ARG;
	if(!g->prepared) return group_fail(g, PT_ERR_STATE, "pt_group_sync before pt_group_prepare");
	return group_sync(g);
}

int pt_group_download(pt_group* g, float* rgb) {
	if(!g) return PT_ERR_ARG;
	if(!rgb) return group_fail(g, PT_ERR_ARG, "null output");
	if(!g->prepared || g->frame == 0) return group_fail(g, PT_ERR_STATE, "pt_group_download before a frame was traced");
	int rc = group_sync(g);
	if(rc) return rc;
	pt_ctx* root = g->part[0];
	if((rc = download_image(root, g->image.p, (size_t)g->st.width * (size_t)g->st.height * 12, rgb)) != PT_OK) return group_fail(g, rc, pt_last_error(root));
	return PT_OK;
}

}  // extern "C"

pt_group* pt_group_create(const int* devices, int n) {
	if(!devices || n < 1 || n > 64) { fail(nullptr, PT_ERR_ARG, "pt_group_create: 1..64 devices"); return nullptr; }
	pt_group* g = new pt_group();
	for(int p = 0; p < n; p++) {
		pt_ctx* c = pt_create(devices[p]);
		if(!c) { pt_group_destroy(g); return nullptr; }
		g->part.push_back(c);
	}
	// direct access between the root and every other device, where the hardware offers it; without it the runtime
	// stages the copies itself
	const int root = devices[0];
	for(int p = 1; p < n; p++) {
		if(devices[p] == root) continue;
		int can = 0;
		if(hipDeviceCanAccessPeer(&can, devices[p], root) == hipSuccess && can && hipSetDevice(devices[p]) == hipSuccess) {
			hipError_t e = hipDeviceEnablePeerAccess(root, 0);
			if(e != hipSuccess) (void)hipGetLastError();      // already enabled counts as enabled
		}
	}
	return g;
}

void pt_group_destroy(pt_group* g) {
	if(!g) return;
	if(g->active.load()) { pt_group_render_abort(g); group_join(g); }
	if(g->prepared) (void)group_sync(g);
	group_release_ring(g);
	if(!g->part.empty()) {
		(void)hipSetDevice(g->part[0]->device);
		g->gathered.release();
		g->image.release();
	}
	for(pt_ctx* c : g->part) pt_destroy(c);
	delete g;
}

int pt_group_size(const pt_group* g) { return g ? (int)g->part.size() : PT_ERR_ARG; }

pt_ctx* pt_group_ctx(pt_group* g, int part) {
	if(!g || part < 0 || part >= (int)g->part.size()) return nullptr;
	return g->part[(size_t)part];
}

const char* pt_group_last_error(const pt_group* g) {
	thread_local std::string text;
	if(!g) return pt_last_error(nullptr);
	{ std::lock_guard<std::mutex> l(const_cast<pt_group*>(g)->mu); text = g->err; }
	return text.c_str();
}

int pt_group_render_start(pt_group* g, const pt_world* w, const pt_object* objs, int n_objects, const pt_settings* s, float* rgb) {
	if(!g) return PT_ERR_ARG;
	if(!rgb) return group_fail(g, PT_ERR_ARG, "null output buffer");
	if(g->active.load() && !g->done.load()) return group_fail(g, PT_ERR_BUSY, "a render is already active");
	if(g->active.load()) group_join(g);
	const int n = (int)g->part.size();
	if(!s) return group_fail(g, PT_ERR_ARG, "null settings");
	for(int p = 0; p < n; p++) {
		pt_ctx* c = g->part[(size_t)p];
		if(c->active.load()) { if(!c->done.load()) return group_fail(g, PT_ERR_BUSY, "a partition's context is rendering on its own"); join_worker(c); }
		pt_settings sp = *s;
		sp.part_index = p;
		sp.part_count = n;
		int rc = store_inputs(c, w, objs, n_objects, &sp);
		if(rc) return group_fail(g, rc, pt_last_error(c));
		*c->abort_flag = 0;
		*c->progress_word = 0;
		c->fraction_shown = 0;
		{ std::lock_guard<std::mutex> l(c->mu); c->live_valid = false; }
	}
	{ std::lock_guard<std::mutex> l(g->mu); g->err.clear(); }
	g->st = *s;
	g->user_rgb = rgb;
	g->total_pixels = (uint64_t)s->width * (uint64_t)s->height;
	g->done.store(0); g->aborted.store(0); g->status.store(0); g->abort_requested.store(0);
	g->done_pixels.store(0); g->done_paths.store(0);
	g->t_start = std::chrono::steady_clock::now();
	g->active.store(1);
	g->worker = std::thread(group_main, g);
	return PT_OK;
}

int pt_group_render_progress(pt_group* g, pt_progress* info) {
	if(!g || !info) return PT_ERR_ARG;
	memset(info, 0, sizeof *info);
	if(!g->active.load()) return group_fail(g, PT_ERR_STATE, "no render was started");
	if(g->done.load()) {
		info->done = 1;
		info->aborted = g->aborted.load();
		info->fraction = info->aborted ? (g->total_pixels ? (float)((double)g->done_pixels.load() / (double)g->total_pixels) : 0.0f) : 1.0f;
		info->seconds = std::chrono::duration<double>(g->t_end - g->t_start).count();
		info->paths_done = g->done_paths.load();
		return g->status.load();
	}
	info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - g->t_start).count();
	double share = 0;
	uint64_t paths = 0;
	for(pt_ctx* c : g->part) {
		float f = 0;
		uint64_t done = 0;
		live_fraction(c, &f, &done);
		share += f;
		paths += done;
	}
	info->fraction = (float)(share / (double)g->part.size());
	info->paths_done = paths;
	return PT_OK;
}

int pt_group_render_snapshot(pt_group* g, float* rgb, uint64_t* pixels_done) {
	if(!g) return PT_ERR_ARG;
	if(!rgb) return group_fail(g, PT_ERR_ARG, "null output");
	if(pixels_done) *pixels_done = 0;
	if(!g->active.load()) return group_fail(g, PT_ERR_STATE, "no render was started");
	if(g->done.load()) {
		if(pixels_done) *pixels_done = g->aborted.load() ? g->done_pixels.load() : g->total_pixels;
		return g->status.load();
	}
	uint64_t total = 0;
	for(pt_ctx* c : g->part) {                 // every partition writes its own pixels of the one image
		uint64_t pixels = 0;
		int rc = snapshot_live(c, rgb, &pixels);
		if(rc != PT_OK) return group_fail(g, rc, pt_last_error(c));
		total += pixels;
	}
	if(pixels_done) *pixels_done = total;
	return PT_OK;
}

int pt_group_render_abort(pt_group* g) {
	if(!g) return PT_ERR_ARG;
	if(!g->active.load()) return group_fail(g, PT_ERR_STATE, "no render was started");
	g->abort_requested.store(1);
	for(pt_ctx* c : g->part) *c->abort_flag = 1;
	return PT_OK;
}

int pt_group_render_wait(pt_group* g) {
	if(!g) return PT_ERR_ARG;
	if(!g->active.load()) return group_fail(g, PT_ERR_STATE, "no render was started");
	group_join(g);
	g->active.store(1);     // keep the finished render pollable until the next start
	return g->status.load();
}

}  // extern "C"

extern "C" {

int pt_debug_primary(pt_ctx* c, int32_t* hit_id, uint64_t* cell_hash) {
	if(!c) return PT_ERR_ARG;
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_debug_primary before pt_prepare");
	PT_HIP(c, hipSetDevice(c->device));
	size_t n = (size_t)c->st.width * c->st.height;
	size_t hash_at = ((n * 4 + 255) / 256) * 256;
	PT_HIP(c, c->scratch.reserve(hash_at + n * 8));
	int32_t* d_id = c->scratch.as<int32_t>();
	uint64_t* d_hash = (uint64_t*)((char*)c->scratch.p + hash_at);
	PT_HIP(c, ptl::launch_primary_aov(c->job.scene, c->job.frame, c->st.width, c->st.height, d_id, d_hash, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	if(hit_id) PT_HIP(c, hipMemcpy(hit_id, d_id, n * 4, hipMemcpyDeviceToHost));
	if(cell_hash) PT_HIP(c, hipMemcpy(cell_hash, d_hash, n * 8, hipMemcpyDeviceToHost));
	return PT_OK;
}

int pt_cast_histogram(pt_ctx* c, uint64_t* histogram, int bins) {
	if(!c) return PT_ERR_ARG;
	if(!histogram || bins < 2 || bins > 65) return fail(c, PT_ERR_ARG, "pt_cast_histogram: 2..65 bins");
	if(c->active.load() && !c->done.load()) return fail(c, PT_ERR_BUSY, "a render is active");
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_cast_histogram before pt_prepare");
	PT_HIP(c, hipSetDevice(c->device));
	int rc = wait_launches(c);
	if(rc) return rc;
	PT_HIP(c, c->scratch.reserve(65 * 8));
	PT_HIP(c, hipMemsetAsync(c->scratch.p, 0, 65 * 8, c->stream));
	PT_HIP(c, ptl::launch_cast_histogram(c->job, c->fb_floats / 3, c->scratch.as<unsigned long long>(), bins, c->stream));
	PT_HIP(c, hipMemcpyAsync(histogram, c->scratch.p, (size_t)bins * 8, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

static int cast_rays(pt_ctx* c, int mode, const float* origins, const float* dirs, int n, int32_t* hit_id, float* hit_pos, float* hit_nrm) {
	if(!c) return PT_ERR_ARG;
	if(c->active.load() && !c->done.load()) return fail(c, PT_ERR_BUSY, "a render is active");
	if(!c->prepared) return fail(c, PT_ERR_STATE, "ray query before pt_prepare");
	if(n < 0 || (n > 0 && (!origins || !dirs || !hit_id))) return fail(c, PT_ERR_ARG, "null argument");
	if(n == 0) return PT_OK;
	PT_HIP(c, hipSetDevice(c->device));
	size_t v = (size_t)n * 12;
	PT_HIP(c, c->scratch.reserve(v * 4 + (size_t)n * 4 + 1024));
	char* base = (char*)c->scratch.p;
	float* d_o = (float*)base; float* d_d = (float*)(base + v); float* d_p = (float*)(base + 2 * v); float* d_n = (float*)(base + 3 * v);
	int32_t* d_id = (int32_t*)(base + 4 * v);
	PT_HIP(c, hipMemcpyAsync(d_o, origins, v, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, hipMemcpyAsync(d_d, dirs, v, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, ptl::launch_cast(c->job.scene, mode, d_o, d_d, n, d_id, d_p, d_n, c->stream));
	PT_HIP(c, hipMemcpyAsync(hit_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
	if(hit_pos) PT_HIP(c, hipMemcpyAsync(hit_pos, d_p, v, hipMemcpyDeviceToHost, c->stream));
	if(hit_nrm) PT_HIP(c, hipMemcpyAsync(hit_nrm, d_n, v, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

int pt_cast_rays(pt_ctx* c, const float* origins, const float* dirs, int n, int32_t* hit_id, float* hit_pos, float* hit_nrm) {
	return cast_rays(c, 0, origins, dirs, n, hit_id, hit_pos, hit_nrm);
}

int pt_cast_rays_noalloc(pt_ctx* c, const float* origins, const float* dirs, int n, int32_t* hit_id, float* hit_pos, float* hit_nrm) {
	if(!c) return PT_ERR_ARG;
	if(c->active.load() && !c->done.load()) return fail(c, PT_ERR_BUSY, "a render is active");
	if(!c->prepared) return fail(c, PT_ERR_STATE, "ray query before pt_prepare");
	if(n < 0 || (n > 0 && (!origins || !dirs || !hit_id))) return fail(c, PT_ERR_ARG, "null argument");
	if(n > PT_NOALLOC_MAX_RAYS) return fail(c, PT_ERR_LIMIT, "pt_cast_rays_noalloc takes at most %d rays a call, not %d (pt_cast_rays has no limit)", PT_NOALLOC_MAX_RAYS, n);
	if(n == 0) return PT_OK;
	static_assert(PT_NOALLOC_MAX_RAYS == ptl::kStaticRays, "the header's limit is the staging array's");
	if(!c->static_scratch || c->static_scratch_bytes < (size_t)PT_NOALLOC_MAX_RAYS * 52) return fail(c, PT_ERR_STATE, "the static staging area is missing");
	PT_HIP(c, hipSetDevice(c->device));
	// every context on this GPU stages in the same array: one call at a time per device
	std::lock_guard<std::mutex> staging(g_static_scratch_mu[c->device % kMaxDevices]);
	// fixed places in the device's static array, whatever n is; every word a call reads back it has written itself
	const size_t v = (size_t)n * 12, span = (size_t)PT_NOALLOC_MAX_RAYS * 12;
	char* base = (char*)c->static_scratch;
	float* d_o = (float*)base; float* d_d = (float*)(base + span); float* d_p = (float*)(base + 2 * span); float* d_n = (float*)(base + 3 * span);
	int32_t* d_id = (int32_t*)(base + 4 * span);
	PT_HIP(c, hipMemcpyAsync(d_o, origins, v, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, hipMemcpyAsync(d_d, dirs, v, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, ptl::launch_cast(c->job.scene, 0, d_o, d_d, n, d_id, d_p, d_n, c->stream));
	PT_HIP(c, hipMemcpyAsync(hit_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
	if(hit_pos) PT_HIP(c, hipMemcpyAsync(hit_pos, d_p, v, hipMemcpyDeviceToHost, c->stream));
	if(hit_nrm) PT_HIP(c, hipMemcpyAsync(hit_nrm, d_n, v, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

int pt_pick_rays(pt_ctx* c, const float* origins, const float* dirs, int n, int32_t* hit_id, float* hit_pos, float* hit_nrm) {
	return cast_rays(c, 1, origins, dirs, n, hit_id, hit_pos, hit_nrm);
}

int pt_debug_device_memory(pt_ctx* c, uint64_t* free_bytes, uint64_t* total_bytes, uint64_t* scratch_capacity) {
	if(!c) return PT_ERR_ARG;
	PT_HIP(c, hipSetDevice(c->device));
	size_t f = 0, t = 0;
	PT_HIP(c, hipMemGetInfo(&f, &t));
	if(free_bytes) *free_bytes = f;
	if(total_bytes) *total_bytes = t;
	if(scratch_capacity) *scratch_capacity = c->scratch.cap;
	return PT_OK;
}

int pt_debug_arith(pt_ctx* c, int op, const float* a, const float* b, int n, float* out) {
	if(!c) return PT_ERR_ARG;
	if(op < 0 || op > 4 || n < 0 || (n > 0 && (!a || !out || (op == 1 && !b)))) return fail(c, PT_ERR_ARG, "bad argument");
	if(n == 0) return PT_OK;
	PT_HIP(c, hipSetDevice(c->device));
	const size_t per = op == 0 ? 1 : 3, va = (size_t)n * per * 4, vb = (size_t)n * 4;
	PT_HIP(c, c->scratch.reserve(2 * va + vb + 1024));
	char* base = (char*)c->scratch.p;
	float* d_a = (float*)base; float* d_o = (float*)(base + va); float* d_b = (float*)(base + 2 * va);
	PT_HIP(c, hipMemcpyAsync(d_a, a, va, hipMemcpyHostToDevice, c->stream));
	if(op == 1) PT_HIP(c, hipMemcpyAsync(d_b, b, vb, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, ptl::launch_arith(op, d_a, d_b, n, d_o, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	PT_HIP(c, hipMemcpy(out, d_o, va, hipMemcpyDeviceToHost));
	return PT_OK;
}

int pt_debug_arith_sweep(pt_ctx* c, int op, uint32_t first_bits, uint64_t count, uint64_t* mismatches) {
	if(!c) return PT_ERR_ARG;
	if(op < 0 || op > 4 || !mismatches || count > (1ull << 32)) return fail(c, PT_ERR_ARG, "bad argument");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, c->scratch.reserve(1024));
	PT_HIP(c, hipMemsetAsync(c->scratch.p, 0, 8, c->stream));
	PT_HIP(c, ptl::launch_arith_sweep(op, first_bits, count, (unsigned long long*)c->scratch.p, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	PT_HIP(c, hipMemcpy(mismatches, c->scratch.p, 8, hipMemcpyDeviceToHost));
	return PT_OK;
}

// The encoder's 255 thresholds, found once per context with the host's C library and kept on the device.
static int srgb8_table(pt_ctx* c) {
	if(c->thresholds_ready) return PT_OK;
	float t[256];
	pt::srgb8_thresholds(t);
	t[255] = 0;
	PT_HIP(c, c->thresholds.reserve(sizeof t));
	PT_HIP(c, hipMemcpyAsync(c->thresholds.p, t, sizeof t, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	c->thresholds_ready = true;
	return PT_OK;
}

int pt_present_srgb8(pt_ctx* c, uint8_t* rgb8) {
	if(!c) return PT_ERR_ARG;
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_present_srgb8 before pt_prepare");
	if(!rgb8) return fail(c, PT_ERR_ARG, "null output");
	if(c->st.part_count != 1) return fail(c, PT_ERR_STATE, "pt_present_srgb8 works on a whole image (part_count == 1)");
	PT_HIP(c, hipSetDevice(c->device));
	int rc = wait_launches(c);
	if(rc == PT_OK) rc = srgb8_table(c);
	if(rc) return rc;
	PT_HIP(c, c->scratch.reserve(c->fb_floats + 16));
	PT_HIP(c, ptl::launch_srgb8(c->fb.as<float>(), c->fb_floats, c->thresholds.as<float>(), c->scratch.as<uint8_t>(), c->stream));
	PT_HIP(c, hipMemcpyAsync(rgb8, c->scratch.p, c->fb_floats, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

int pt_debug_srgb8(pt_ctx* c, const float* linear, size_t n, uint8_t* out) {
	if(!c) return PT_ERR_ARG;
	if(n > 0 && (!linear || !out)) return fail(c, PT_ERR_ARG, "null argument");
	if(n == 0) return PT_OK;
	PT_HIP(c, hipSetDevice(c->device));
	int rc = srgb8_table(c);
	if(rc) return rc;
	const size_t in_bytes = ((n * 4 + 255) / 256) * 256;
	PT_HIP(c, c->scratch.reserve(in_bytes + n + 16));
	float* d_in = c->scratch.as<float>();
	uint8_t* d_out = c->scratch.as<uint8_t>() + in_bytes;
	PT_HIP(c, hipMemcpyAsync(d_in, linear, n * 4, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, ptl::launch_srgb8(d_in, n, c->thresholds.as<float>(), d_out, c->stream));
	PT_HIP(c, hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

int pt_debug_srgb8_sweep(pt_ctx* c, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint64_t* histogram256) {
	if(!c) return PT_ERR_ARG;
	if(!mismatches || !histogram256 || count > (1ull << 32)) return fail(c, PT_ERR_ARG, "bad argument");
	PT_HIP(c, hipSetDevice(c->device));
	int rc = srgb8_table(c);
	if(rc) return rc;
	PT_HIP(c, c->scratch.reserve(257 * 8));
	PT_HIP(c, hipMemsetAsync(c->scratch.p, 0, 257 * 8, c->stream));
	unsigned long long* d = c->scratch.as<unsigned long long>();
	PT_HIP(c, ptl::launch_srgb8_sweep(first_bits, count, c->thresholds.as<float>(), d, d + 1, c->stream));
	uint64_t host[257];
	PT_HIP(c, hipMemcpyAsync(host, d, sizeof host, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	*mismatches = host[0];
	memcpy(histogram256, host + 1, 256 * 8);
	return PT_OK;
}

int pt_assemble(pt_ctx* c, const float* d_parts, size_t stride_floats, int part_count, int width, int height, float* d_rgb, void* hip_stream) {
	if(!c) return PT_ERR_ARG;
	if(!d_parts || !d_rgb || part_count < 1 || width < 1 || height < 1) return fail(c, PT_ERR_ARG, "bad argument");
	PT_HIP(c, hipSetDevice(c->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
	PT_HIP(c, ptl::launch_assemble(d_parts, stride_floats, part_count, width, height, d_rgb, s));
	return PT_OK;
}

/* Diagnostic builds only: event counters of the cooperative cast (0 values otherwise). */
int pt_debug_coop_stats(pt_ctx* c, unsigned long long* out16) {
	if(!c || !out16) return PT_ERR_ARG;
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipDeviceSynchronize());
	return ptl::coop_stats(out16);
}

int pt_debug_tile_cost(pt_ctx* c, uint16_t* cost, int capacity, int* tiles) {
	if(!c) return PT_ERR_ARG;
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_debug_tile_cost before pt_prepare");
	if(tiles) *tiles = (int)c->tile_cost_host.size();
	if(cost) {
		if((size_t)capacity < c->tile_cost_host.size()) return fail(c, PT_ERR_LIMIT, "cost needs %zu entries", c->tile_cost_host.size());
		if(!c->tile_cost_host.empty()) memcpy(cost, c->tile_cost_host.data(), c->tile_cost_host.size() * 2);
	}
	return PT_OK;
}

int pt_debug_grid(pt_ctx* c, uint32_t* cell_start, int cell_capacity, uint32_t* cell_items, int item_capacity, uint32_t* obj_range,
                  int* cell_total, int* item_total, int* built_on_device) {
	if(!c) return PT_ERR_ARG;
	if(!c->prepared) return fail(c, PT_ERR_STATE, "pt_debug_grid before pt_prepare");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	const size_t cells = (size_t)c->stats.cells[0] * (size_t)c->stats.cells[1] * (size_t)c->stats.cells[2];
	const size_t items = (size_t)c->stats.cell_items;
	if(cell_total) *cell_total = (int)cells;
	if(item_total) *item_total = (int)items;
	if(built_on_device) *built_on_device = c->grid_built_on_device ? 1 : 0;
	if(cell_start && (size_t)cell_capacity < cells + 1) return fail(c, PT_ERR_LIMIT, "cell_start needs %zu words", cells + 1);
	if(cell_items && items > 0 && (size_t)item_capacity < items) return fail(c, PT_ERR_LIMIT, "cell_items needs %zu words", items);
	if(cell_start || (cell_items && items > 0)) {
		// the lists are reported in the reference's cell order, whatever order the device keeps them in
		std::vector<uint32_t> start(cells + 1, 0), list(items, 0);
		PT_HIP(c, hipMemcpy(start.data(), c->cell_start.p, (cells + 1) * 4, hipMemcpyDeviceToHost));
		if(items > 0) PT_HIP(c, hipMemcpy(list.data(), c->cell_items.p, items * 4, hipMemcpyDeviceToHost));
		for(uint32_t& v : start) v &= ptl::CELL_START_MASK;      // the flags above the index are the device's own business
		reorder_lists(c->job.scene.grid, false, &start, &list);
		if(cell_start) memcpy(cell_start, start.data(), (cells + 1) * 4);
		if(cell_items && items > 0) memcpy(cell_items, list.data(), items * 4);
	}
	if(obj_range && c->stats.objects > 0) PT_HIP(c, hipMemcpy(obj_range, c->obj_range.p, (size_t)c->stats.objects * 8, hipMemcpyDeviceToHost));
	return PT_OK;
}

int pt_sample_ranges(const pt_settings* s) {
	if(!s) return PT_ERR_ARG;
	const int asked = s->sample_ranges;
	if(asked < 0 || asked > 16 || (asked & (asked - 1)) != 0) return PT_ERR_ARG;
	if(s->frame_index >= 0) return 1;
	if(asked > 0) return asked;
	int ranges = 1;
	while(ranges < 16 && ranges * 2 * 16 <= s->spp_wanted) ranges *= 2;
	return ranges;
}

int pt_partition_tiles(int width, int height, int part_index, int part_count) {
	if(width < 1 || height < 1 || part_count < 1 || part_index < 0 || part_index >= part_count) return PT_ERR_ARG;
	int tiles = ((width + 7) / 8) * ((height + 7) / 8);
	return (tiles - part_index + part_count - 1) / part_count;
}

int pt_partition_floats(const pt_ctx* c, size_t* floats) {
	if(!c || !floats) return PT_ERR_ARG;
	*floats = c->fb_floats;
	return PT_OK;
}

/* ---- host stages --------------------------------------------------------------------------------- */

int pt_host_srgb8_thresholds(float* out255) {
	if(!out255) return PT_ERR_ARG;
	pt::srgb8_thresholds(out255);
	return PT_OK;
}

int pt_host_flatten(const pt_object* objs, int n, float* records) {
	if(n < 0 || (n > 0 && (!objs || !records))) return PT_ERR_ARG;
	pt::flatten_objects(objs, n, records);
	return PT_OK;
}

int pt_host_world(const pt_world* w, int width, int height, float* out) {
	if(!w || !out || width < 1 || height < 1) return PT_ERR_ARG;
	pt::WorldConsts wc;
	pt::world_consts(*w, width, height, &wc);
	memset(out, 0, sizeof(float) * PT_W_COUNT);
	out[PT_W_DIMH] = wc.dim_h; out[PT_W_NEAR] = wc.near_dist;
	auto put = [&](int at, pt::f3 v) { out[at] = v.x; out[at + 1] = v.y; out[at + 2] = v.z; };
	put(PT_W_DIR, wc.cam_dir); put(PT_W_UP, wc.cam_up); put(PT_W_RIGHT, wc.cam_right); put(PT_W_TOPLEFT, wc.top_left);
	put(PT_W_LIGHTDIR, wc.light_dir); put(PT_W_BG, wc.bg); put(PT_W_AMBIENT, wc.ambient); put(PT_W_LIGHT, wc.light);
	return PT_OK;
}

int pt_host_tables(uint64_t seed, int spp_wanted, int grid_width, float* dirs, float* disc, float* samples, int sample_capacity,
                   int32_t* offsets, int* sample_count) {
	if(spp_wanted < 1 || grid_width < 1 || grid_width > 4) return PT_ERR_ARG;
	pt::Pcg rng;
	rng.seed(0, seed);
	std::vector<pt::f3> d(PT_TABLE_COUNT);
	std::vector<pt::f2> c(PT_TABLE_COUNT);
	pt::make_tables(rng, d.data(), c.data());
	std::vector<pt::f2> smp;
	std::vector<int32_t> off;
	pt::make_samples(rng, spp_wanted, grid_width, &smp, &off);
	if(dirs) memcpy(dirs, d.data(), sizeof(pt::f3) * PT_TABLE_COUNT);
	if(disc) memcpy(disc, c.data(), sizeof(pt::f2) * PT_TABLE_COUNT);
	if(sample_count) *sample_count = (int)smp.size();
	if(offsets) memcpy(offsets, off.data(), off.size() * 4);
	if(samples) {
		if((int)smp.size() > sample_capacity) return PT_ERR_LIMIT;
		memcpy(samples, smp.data(), smp.size() * sizeof(pt::f2));
	}
	return PT_OK;
}

int pt_host_grid(const float* records, int n, int hint, int32_t* grid_ints, uint32_t* cell_start, int cell_capacity,
                 uint32_t* cell_items, int item_capacity, int* cell_total, int* item_total) {
	if(!records || n < 1 || hint < 1 || !grid_ints) return PT_ERR_ARG;
	pt::GridDesc g{};
	std::vector<uint32_t> start, items;
	if(!pt::build_grid(records, n, hint, &g, &start, &items)) return PT_ERR_LIMIT;
	for(int k = 0; k < 3; k++) { grid_ints[k] = g.coord_min[k]; grid_ints[3 + k] = g.cell_count[k]; grid_ints[6 + k] = g.cell_size[k]; }
	if(cell_total) *cell_total = (int)g.cells();
	if(item_total) *item_total = (int)items.size();
	if(cell_start) {
		if((int)start.size() > cell_capacity) return PT_ERR_LIMIT;
		memcpy(cell_start, start.data(), start.size() * 4);
	}
	if(cell_items) {
		if((int)items.size() > item_capacity) return PT_ERR_LIMIT;
		if(!items.empty()) memcpy(cell_items, items.data(), items.size() * 4);
	}
	return PT_OK;
}

int pt_host_list_entry(const float centre[3], float radius, uint32_t tag, uint32_t id, uint32_t entry[4], uint32_t* id_back) {
	if(!centre || !entry || id >= ptl::ENTRY_ID_LIMIT || tag > 511u) return PT_ERR_ARG;
	ptl::pack_entry(centre[0], centre[1], centre[2], radius, tag, id, entry);
	if(id_back) *id_back = ptl::entry_id(entry[0], entry[1], entry[2], entry[3]);
	return PT_OK;
}

int pt_host_filter_sphere(const float* record, float centre_radius[4]) {
	if(!record || !centre_radius) return PT_ERR_ARG;
	centre_radius[3] = pt::filter_sphere(record, centre_radius);
	return PT_OK;
}

int pt_host_cone_slopes(const float* record, float out[8]) {
	if(!record || !out) return PT_ERR_ARG;
	int32_t type;
	memcpy(&type, record + PT_F_TYPE, 4);
	if(type != PT_GEOM_CONE) return PT_ERR_ARG;
	pt::cone_slopes(record, out);
	return PT_OK;
}

int pt_host_cell_order(const int32_t cell_count[3], const int32_t cell_size[3], int32_t stride[3]) {
	if(!cell_count || !cell_size || !stride) return PT_ERR_ARG;
	ptl::GridDev g{};
	for(int k = 0; k < 3; k++) { g.cnt[k] = cell_count[k]; g.cs[k] = cell_size[k]; }
	ptl::grid_layout(g);
	for(int k = 0; k < 3; k++) stride[k] = g.stride[k];
	return PT_OK;
}

}  // extern "C"
