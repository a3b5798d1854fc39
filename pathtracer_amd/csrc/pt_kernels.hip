// HIP kernels of the render core (gfx950 / CDNA4, wave64) and their launchers.
//
// pt_k_trace is the whole per-pixel loop nest of the reference's tile job
// (code/raycast.cpp:991-1289) as ONE persistent kernel organised as a wavefront state machine:
// every lane owns one pixel at a time and carries one path; each trip of the wave's loop runs the
// three stages  generate (camera ray of the next sample) -> extend (ray cast) -> shade  for the
// lanes that need them.  A lane whose path ends regenerates the next sample of its pixel in the
// same trip, and a lane whose pixel is finished pulls a new pixel from the wave's tile pool, so the
// live rays stay compacted in the 64 lanes between bounces without ever going through memory.
// The unit of work a lane takes is one RANGE of a pixel's samples (a pixel's samples are cut into
// a few even, contiguous ranges): a lane sums its range in sample order in registers and stores
// the sum; pt_k_resolve then folds a pixel's range sums in range order.  The order of every addition
// is fixed by the inputs, so the image is a pure function of them (independent of wave scheduling,
// grid size and GPU count), while the work units are short enough for the frame to end evenly.
#include <cmath>
#include <cstdlib>

#include "pt_device.h"

using namespace ptd;

namespace {

constexpr int kBlock = 256;            // 4 wavefronts; waves never synchronise with each other
constexpr int kGroup = 64;             // pixels per pool refill = one 8x8 image tile
constexpr uint32_t kAbortPoll = 64;    // group fetches per look at the abort word (power of two)
#ifndef PT_COUNTER_SHARDS
#define PT_COUNTER_SHARDS 8
#endif
constexpr uint32_t kShards = PT_COUNTER_SHARDS;      // group counters
constexpr uint32_t kShardStride = 32;  // words between them (128 bytes)
#ifndef PT_TRACE_WAVES_PER_SIMD
#define PT_TRACE_WAVES_PER_SIMD 4     // register budget of the trace kernel: 128 VGPRs
#endif

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int lanes_below(uint64_t mask) {
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// A range sum and the serial number of the launch that wrote it: one 16-byte store.  An asynchronous render's
// sums go straight through the caches to memory, where the copy engine finds the finished ones when the host asks
// for a snapshot of the frame in progress.  (A vector store with cache-bypass bits; nothing else changes.)
template <bool THROUGH>
__device__ __forceinline__ void store_sum(const ptl::TraceJob& job, float4* at, float x, float y, float z) {
	const v4f v = { x, y, z, __uint_as_float(job.serial) };
	// The compiler cannot see that this is a store of more than eight bytes: an instruction that overwrites one of its four
	// data registers must keep two wait states' distance (on this chip), so they are part of the statement.
	if(THROUGH) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(at), "v"(v) : "memory");
	else *at = make_float4(v.x, v.y, v.z, v.w);
}

// How a finished item leaves the kernel (STORE): 0 as a range sum, 1 as a range sum written through the caches, 2 - the
// progressive frames of a synchronous trace, one sample per pixel and therefore one range - straight into the
// accumulation buffer and the image, with the operations of pt_k_resolve's progressive branch in its order (the item's
// index is the pixel's slot then): no range sum, and no resolve pass after the frame.  `traced` false: a pixel whose bucket
// holds no sample.
enum { STORE_SUM = 0, STORE_SUM_THROUGH = 1, STORE_PIXEL = 2 };
template <int STORE>
__device__ __forceinline__ void finish_item(const ptl::TraceJob& job, size_t item, V3 sum, bool traced) {
	if(STORE == STORE_PIXEL) {
		float* o = job.fb + item * 3;
		if(!traced) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
		float* a = job.accum + item * 3;
		V3 total = sum;
		if(job.frame_index > 0) total = v3(a[0] + sum.x, a[1] + sum.y, a[2] + sum.z);
		a[0] = total.x; a[1] = total.y; a[2] = total.z;
		const V3 c = clamp_hi(total / (float)(job.frame_index + 1), 1.0f);
		o[0] = c.x; o[1] = c.y; o[2] = c.z;
	} else store_sum<STORE == STORE_SUM_THROUGH>(job, job.partial + item, sum.x, sum.y, sum.z);
}

// Pixel centre on the image plane (code/raycast.cpp:1050-1057).  The two quotients x/width and
// y/height come from tables the host filled with the same divisions.
__device__ __forceinline__ V3 pixel_origin_tab(const ptl::TraceJob& job, FrameRef fr, int x, int y) {
	const float fx = job.col_frac[x], fy = job.row_frac[y];
	return (ld3(fr->top_left) + ld3(fr->right_width) * fx) + ld3(fr->down_height) * fy;
}

// Lane states of the trace loop.
enum { NEED_PIXEL = 0, NEED_PATH = 1, IN_FLIGHT = 2, RETIRED = 3 };

// MODE selects the cast at compile time so that each variant carries only its own registers:
// castRayAll for scenes under the brute-force threshold, the per-lane grid walk, or the
// wave-cooperative grid walk (kind-binned intersections through LDS).
enum { CAST_BRUTE = 0, CAST_GRID_LANE = 1, CAST_GRID_COOP = 2 };
// FEAT (brute-force variant only): the scene features the variant is compiled for.  A scene of even spheres and
// unrotated boxes with opaque, smooth materials - the reference's demo scene is one - runs a variant without the
// unit-space intersectors, Fresnel and Oren-Nayar code, which fits five wavefronts per SIMD instead of four; the
// frame is bound by each wavefront's own instruction latencies, so the fifth wavefront is worth 13 %.
// WAVES_SIX (brute-force variants, beside the FEAT bits): the same code compiled for six wavefronts per SIMD (80 registers),
// with the lane state that is only needed once per sample or per item parked in LDS (kCold in the kernel).  On a long frame the
// sixth wavefront is worth 3 %.  (Left to the compiler, twenty values of the lean variant and 36 of the all-features one go
// to scratch memory instead: the lean variant is as fast that way, but writes 1.1 GB of spills per 1080p frame; the
// all-features one is slower than with five wavefronts.)  A short launch - a progressive frame, an eighth of a frame -
// loses with six (more wavefronts to start and to feed from the same counters: +14 % on a one-sample frame), so the launch
// picks by the work there is (launch_trace).
enum { WAVES_SIX = 16 };
// CONE_EXACT (cooperative grid variants, beside the FEAT bits; also the bit of TraceJob's scene features that says the scene has
// cones): the cast also returns the reference cone intersector's phantom hits - for a ray parallel to one of a cone's generators
// to within rounding it reports the crossing of the cone's base plane, wherever that is - which the bounding-sphere filter of the
// plain variants drops before the intersector sees the ray (cast_grid_coop).  A scene with cones runs these variants: +5 % on a
// frame of 49 entities, +15 % on one of 100 000.  PT_FAST_CONES=1 runs the plain ones instead; then about one pixel in a million
// is not the reference's.
enum { CONE_EXACT = 32 };
#ifndef PT_PLAIN_WAVES_PER_SIMD
#define PT_PLAIN_WAVES_PER_SIMD 5
#endif
#ifndef PT_BRUTE_WAVES_PER_SIMD
#define PT_BRUTE_WAVES_PER_SIMD 5     // the all-features brute-force variant: 96 VGPRs with 16 spilled still beats four wavefronts by 7 %
#endif
// THROUGH: the range sums go through the caches to memory (asynchronous renders, whose frames can be polled while they are
// traced).  A compile-time choice: as a run-time one it cost the demo frame 2.4 % through register pressure alone.
}  // namespace

// The trace kernel has a name other units can see: its brute-force variants are compiled as a unit of their own (below).
namespace ptk {
template <int MODE, int VARIANT, int STORE>
__global__ __launch_bounds__(kBlock, MODE != CAST_BRUTE ? PT_TRACE_WAVES_PER_SIMD : (VARIANT & WAVES_SIX) ? 6 : (VARIANT & FEAT_EVERY) == 0 ? PT_PLAIN_WAVES_PER_SIMD : PT_BRUTE_WAVES_PER_SIMD) void pt_k_trace(ptl::TraceJob job) {
	constexpr int FEAT = VARIANT & FEAT_EVERY;
#ifdef PT_COOP_TIMING
	constexpr bool kWaveLds = true;           // the phase timers live in the per-wave LDS block
#else
	constexpr bool kWaveLds = MODE == CAST_GRID_COOP;
#endif
	__shared__ __attribute__((aligned(16))) char lds_raw[kWaveLds ? sizeof(CoopShared) * (kBlock / 64) : 16];
	CoopShared& coop = reinterpret_cast<CoopShared*>(lds_raw)[kWaveLds ? threadIdx.x / 64 : 0];
	const Scene sc = scene_view(job.scene);
	const FrameRef fr = (FrameRef)job.frame;
	const int lane = lane_id();
	const int width = fr->width, height = fr->height, tile_w = fr->tile_w, max_bounces = fr->max_bounces;

	// wave-uniform tile pool: one 8x8 image tile at a time, with what all its pixels share
	int pool_pos = kGroup;                   // next pixel of the tile to hand out; kGroup: tile used up
	int tile_x0 = 0, tile_y0 = 0, tile_slot0 = 0, tile_mx = 0, tile_my = 0, tile_range = 0;
	const int ranges_log2 = job.ranges_log2;
	// the blue-noise bucket offsets (at most 17 words) are read at every hand-out: keep them in LDS
	__shared__ int bucket[20];
	if((int)threadIdx.x <= tile_w * tile_w) bucket[threadIdx.x] = sc.offsets[threadIdx.x];
	// which group counters a wavefront of this workgroup has found dry (they only grow): the others need not ask again.  Every
	// wavefront ends by finding all eight dry, each question an atomic operation that queues at its counter with everybody else's
	__shared__ uint32_t dry_shards;
	if(MODE != CAST_GRID_COOP && threadIdx.x == 0) dry_shards = 0u;
	// the counters of the launch after this one (nobody reads them before this kernel has ended)
	if(blockIdx.x == 0 && threadIdx.x < kShards) job.group_counter_next[threadIdx.x * kShardStride] = 0u;
	__syncthreads();
	const int mod_magic = 65536 / tile_w + 1;   // v % tile_w for v < 16 and tile_w <= 4 without a division
	// Staged hand-out (every variant but the cooperative cast, whose LDS share is full): the wave that fetches a
	// group works out its 64 items with all lanes at once and parks them in LDS, 32 bytes each, so that a lane in
	// need of work only reads its item instead of the whole wave stepping through the per-pixel arithmetic for it.
	//   a: first sample's table index | si + (si_end << 16) | lens-table start | progressive pass
	//   b: pixel centre on the image plane (x y z) | pixel slot
	constexpr bool kStaged = MODE != CAST_GRID_COOP;
	__shared__ uint4 staged[kStaged ? 2 * kBlock : 1];
	uint4* const stage_a = staged + (kStaged ? (threadIdx.x / 64) * 128 : 0);
	uint4* const stage_b = stage_a + (kStaged ? 64 : 0);
	size_t tile_item0 = 0;                     // first range sum of the staged group's range
	// Cold lane state in LDS (the six-wavefront variants): what a lane needs once per sample or once per item - table
	// indices, pixel centre, pixel, the running range sum, where it goes - waits in LDS, a word of all 256 lanes side by
	// side, instead of in fifteen registers that are live across the cast.  (At five wavefronts, where the registers are
	// there, the same parking costs 4 %.)
	constexpr bool kCold = MODE == CAST_BRUTE && (VARIANT & WAVES_SIX) != 0;
	enum { C_BASE = 0, C_END, C_LENS, C_PASS, C_OX, C_OY, C_OZ, C_ITEM_LO, C_ITEM_HI, C_PXY, C_SX, C_SY, C_SZ, C_WORDS };
	__shared__ uint32_t cold_raw[kCold ? C_WORDS * kBlock : 1];
	uint32_t* const cold = cold_raw + (kCold ? threadIdx.x : 0);
	bool dry = false;
	uint32_t shard = blockIdx.x % kShards;     // lane 0's view is the one that counts

	// per-lane pixel state
	int state = NEED_PIXEL;
	int px = 0, py = 0, si = 0, si_end = 0, sample_base = 0, lens = 0;
	size_t item = 0;                         // where the range sum goes: range * pixel slots + pixel slot
	uint32_t pass = 0;
	V3 origin = v3(0, 0, 0), sum = v3(0, 0, 0);
	// per-lane path state
	V3 rp = v3(0, 0, 0), rd = v3(0, 0, 1), att = v3(1, 1, 1), col = v3(0, 0, 0);
	int bounce = 0;
	Rng rng = { 0, 1 };
	uint32_t casts = 0;                          // wave-uniform: rays cast by the whole wavefront
	// cooperative cast only: what a lane keeps of an unfinished cast, and whether its ray is new
	CoopLane walk_state;
	coop_lane_reset(walk_state);
	bool fresh = true;
#ifdef PT_COOP_TIMING
	if(lane < 16) coop.tacc[lane] = 0;
#endif
	PT_TICK(t_kernel);
	PT_TICK(t_loop);
#ifdef PT_COOP_TIMING
	const unsigned long long wall_start = wall_clock64();    // 100 MHz, one time base for the whole chip
#endif

	for(;;) {
		// ---- stage 0: lanes without a pixel draw one from the wave's pool -------------------
		// (Making lanes wait until several ask together was measured: the idle lanes cost more than
		// the hand-outs saved.)
PT_MARK("loop_begin");
		uint64_t want = __ballot(state == NEED_PIXEL);
		while(want != 0) {
			if(PT_RARE2(pool_pos == kGroup)) {     // once in 64 hand-outs
				if(dry) break;
				// Groups are dealt from kShards counters (group = count * kShards + shard), each on a cache
				// line of its own, so that fetches do not queue up behind one address; a wave starts at the
				// shard of its block (blocks go round the XCDs, so mostly that XCD's L2 serves it) and moves on
				// when a shard runs out.  One wave in kAbortPoll fetches looks at the host's abort word (a read
				// over the bus that costs microseconds and does not scale to one per fetch); the wave that finds
				// it set pushes every counter past the end, which stops all other waves at their next fetch.
				uint32_t g = 0x80000000u;
				if(lane == 0) {
					for(uint32_t tried = 0; tried < kShards; tried++) {
						if(!kStaged || ((*(volatile uint32_t*)&dry_shards >> shard) & 1u) == 0u) {      // (the cooperative cast does without: no register to spare, and long launches)
							const uint32_t count = atomicAdd(job.group_counter + shard * kShardStride, 1u);
							const uint32_t cand = count < 0x04000000u ? count * kShards + shard : 0x80000000u;
							if(cand < (uint32_t)job.n_groups) { g = cand; break; }
							if(kStaged) atomicOr(&dry_shards, 1u << shard);
						}
						shard = (shard + 1u) % kShards;
					}
					if(g != 0x80000000u && (g & (kAbortPoll - 1u)) == 0u) {
						if(PT_RARE2(*job.abort_flag != 0u)) {
							for(uint32_t k = 0; k < kShards; k++) atomicOr(job.group_counter + k * kShardStride, 0x80000000u);
							g = 0x80000000u;
						}
						// progress for the host: groups go out in index order, near enough.  The word sits beside the abort word in
						// pinned host memory; a posted write, nobody waits for it
						else ((volatile uint32_t*)job.abort_flag)[PT_PROGRESS_WORD] = g + 1u;       // groups go out in index order, near enough: a posted write, nobody waits for it
					}
				}
				g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
				if(g >= (uint32_t)job.n_groups) { dry = true; break; }
				// tiles are dealt in the order of the host's cost estimate, dearest first, so that the frame ends on cheap ones
				const uint32_t g_tile = g >> ranges_log2, g_range = g & ((1u << ranges_log2) - 1u);
				const int local_tile = job.tile_order ? job.tile_order[g_tile] : (int)g_tile;
				const int tile = local_tile * job.part_count + job.part_index;
				tile_range = (int)g_range;
				tile_x0 = (tile % job.tiles_x) * 8;
				tile_y0 = (tile / job.tiles_x) * 8;
				tile_slot0 = local_tile * kGroup;
				tile_mx = tile_x0 % tile_w;
				tile_my = tile_y0 % tile_w;
				pool_pos = 0;
				if(kStaged) {
					tile_item0 = (size_t)tile_range * job.slots;
					const int j = lane, x = tile_x0 + (j & 7), y = tile_y0 + (j >> 3);
					uint4 a = make_uint4(0u, 0u, 0u, 0u), b = make_uint4(0u, 0u, 0u, 0u);
					if(x < width && y < height) {
						const uint32_t slot = job.compact ? (uint32_t)(tile_slot0 + j) : (uint32_t)y * (uint32_t)width + (uint32_t)x;
						const int vx = tile_mx + (j & 7), vy = tile_my + (j >> 3);
						const int rx = vx - ((vx * mod_magic) >> 16) * tile_w, ry = vy - ((vy * mod_magic) >> 16) * tile_w;
						const int t = ry * tile_w + rx;                // (y % tile_w) * tile_w + (x % tile_w)
						const int first = bucket[t], n_samples = bucket[t + 1] - first;
						int s0 = 0, s1 = 0;
						uint32_t round = 0;
						if(job.frame_index < 0) {
							s0 = (tile_range * n_samples) >> ranges_log2;
							s1 = ((tile_range + 1) * n_samples) >> ranges_log2;
						} else if(n_samples > 0) {
							s0 = job.frame_index % n_samples; s1 = s0 + 1; round = (uint32_t)(job.frame_index / n_samples);
						}
						if(s0 == s1) {
							// nothing to trace (a bucket without samples, code/raycast.cpp:1270, or fewer samples than ranges)
							finish_item<STORE>(job, tile_item0 + slot, v3(0, 0, 0), false);
						} else {
							const V3 o = pixel_origin_tab(job, fr, x, y);
							a = make_uint4((uint32_t)first, (uint32_t)s0 | ((uint32_t)s1 << 16), fr->aperture != 0.0f ? (uint32_t)lens_start(fr, x, y) : 0u, round);
							b = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), slot);
						}
					}
					stage_a[lane] = a;
					stage_b[lane] = b;
					wave_sync();
				}
			}
			int avail = kGroup - pool_pos;
			int rank = lanes_below(want);
			if(kStaged) {
				if(state == NEED_PIXEL && rank < avail) {
					const int j = pool_pos + rank;
					const uint4 a = stage_a[j];
					if(a.y != 0u) {                                  // else: outside the image or nothing to trace - the lane asks again
						const uint4 b = stage_b[j];
						si = (int)(a.y & 0xffffu);
						if(kCold) {
							const size_t it = tile_item0 + b.w;
							cold[C_END * kBlock] = a.y >> 16; cold[C_PASS * kBlock] = a.w;
							cold[C_ITEM_LO * kBlock] = (uint32_t)it; cold[C_ITEM_HI * kBlock] = (uint32_t)((unsigned long long)it >> 32);
							cold[C_PXY * kBlock] = (uint32_t)(tile_x0 + (j & 7)) | ((uint32_t)(tile_y0 + (j >> 3)) << 16);
							cold[C_BASE * kBlock] = a.x; cold[C_LENS * kBlock] = a.z;
							cold[C_OX * kBlock] = b.x; cold[C_OY * kBlock] = b.y; cold[C_OZ * kBlock] = b.z;
							cold[C_SX * kBlock] = 0u; cold[C_SY * kBlock] = 0u; cold[C_SZ * kBlock] = 0u;
						} else {
							px = tile_x0 + (j & 7);
							py = tile_y0 + (j >> 3);
							sample_base = (int)a.x; si_end = (int)(a.y >> 16); lens = (int)a.z; pass = a.w;
							origin = v3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
							item = tile_item0 + b.w;
							sum = v3(0, 0, 0);
						}
						state = NEED_PATH;
					}
				}
			} else if(state == NEED_PIXEL && rank < avail) {
				const int j = pool_pos + rank;
				px = tile_x0 + (j & 7);
				py = tile_y0 + (j >> 3);
				if(px < width && py < height) {
					const size_t slot = job.compact ? (size_t)(tile_slot0 + j) : (size_t)py * (size_t)width + (size_t)px;
					item = (size_t)tile_range * job.slots + slot;         // range-major: the 8 pixels of a tile row fill one 128-byte line
					const int vx = tile_mx + (j & 7), vy = tile_my + (j >> 3);
					const int rx = vx - ((vx * mod_magic) >> 16) * tile_w, ry = vy - ((vy * mod_magic) >> 16) * tile_w;
					const int t = ry * tile_w + rx;                // (py % tile_w) * tile_w + (px % tile_w)
					sample_base = bucket[t];
					const int n_samples = bucket[t + 1] - sample_base;
					if(job.frame_index < 0) {
						si = (tile_range * n_samples) >> ranges_log2;
						si_end = ((tile_range + 1) * n_samples) >> ranges_log2;
						pass = 0;
					} else if(n_samples > 0) {
						si = job.frame_index % n_samples; si_end = si + 1; pass = (uint32_t)(job.frame_index / n_samples);
					} else { si = 0; si_end = 0; }
					if(si == si_end) {
						// nothing to trace (a bucket without samples, code/raycast.cpp:1270, or fewer samples than ranges)
						finish_item<STORE>(job, item, v3(0, 0, 0), false);
					} else {
						lens = fr->aperture != 0.0f ? lens_start(fr, px, py) : 0;
						if(MODE != CAST_GRID_COOP) origin = pixel_origin_tab(job, fr, px, py);
						sum = v3(0, 0, 0);
						state = NEED_PATH;
					}
				}
				// pixels of an edge tile that fall outside the image are skipped: the lane asks again
			}
			int asked = __popcll(want);
			pool_pos += asked < avail ? asked : avail;
			want = __ballot(state == NEED_PIXEL);
		}
PT_MARK("handed_out");
		if(dry && state == NEED_PIXEL) state = RETIRED;
		if(__ballot(state != RETIRED) == 0) break;
		PT_TOCK(coop, MODE == CAST_GRID_COOP ? 15 : 10, t_loop);   // non-cooperative variants: slot 10 = pixel hand-out, 11 = camera ray, 14 = cast, 12 = shade and pixel write

PT_MARK("camera_begin");
		// ---- stage 1: generate the camera ray of the next sample (code/raycast.cpp:1059-1090) ---
		if(state == NEED_PATH) {
			const int table_at = kCold ? (int)cold[C_BASE * kBlock] : sample_base;
			const int lens_at = kCold ? (int)cold[C_LENS * kBlock] : lens;
			v2f s = sc.samples[table_at + si];
			// the cooperative cast is short of registers: it recomputes what the other variants keep per pixel
			const V3 from = MODE == CAST_GRID_COOP ? pixel_origin_tab(job, fr, px, py)
			              : kCold ? v3(__uint_as_float(cold[C_OX * kBlock]), __uint_as_float(cold[C_OY * kBlock]), __uint_as_float(cold[C_OZ * kBlock])) : origin;
			camera_ray(sc, fr, from, s, lens_at + si, rp, rd);
#ifdef PT_EAGER_SEED
			sample_stream(fr, px, py, si, (uint64_t)pass, rng);
#endif
			att = v3(1, 1, 1);
			col = v3(0, 0, 0);
			bounce = 0;
			state = IN_FLIGHT;
		}

PT_MARK("camera_end");
		// ---- stage 2+3: extend one ray and shade the hit (code/raycast.cpp:1092-1261) ---------
		const bool flying = state == IN_FLIGHT;
		V3 hp = v3(0, 0, 0), hn = v3(0, 0, 0);
		Surface surf;
		surf.m0 = (v4f)(0.0f); surf.m1 = (v4f){ 0.0f, 0.0f, 0.0f, 1.0f }; surf.rough = 0;
		int id = 0;
		bool landed = flying;
		PT_TOCK(coop, MODE == CAST_GRID_COOP ? 15 : 11, t_loop);
PT_MARK("cast_begin");
		if(MODE == CAST_GRID_COOP) {
			// the whole wavefront takes part; lanes whose walk is not over yet stay in flight and resume
			id = cast_grid_coop<FEAT, (VARIANT & CONE_EXACT) != 0>(sc, coop, lane, flying, fresh, walk_state, rp, rd, hp, hn, surf, landed);
			fresh = landed || !flying;
			get6(coop.ray[lane], rp, rd);     // the ray waits in LDS during the cast, not in registers
			PT_TOCK(coop, 14, t_loop);
		} else {
			if(flying) {
				uint64_t unused = 0;
				id = MODE == CAST_BRUTE ? cast_all<FEAT>(sc, rp, rd, hp, hn, surf) : cast_grid<false, FEAT>(sc, rp, rd, hp, hn, surf, unused);
			}
			PT_TOCK(coop, 14, t_loop);
		}
PT_MARK("cast_end");
		casts += (uint32_t)__popcll(__ballot(landed));
		if(landed) {
#ifndef PT_EAGER_SEED
			// The sample's random stream is set up when its path first lands on an entity: a path that leaves into the sky
			// draws no number (nearly half the paths of the demo frame), and nothing between the camera and this point draws one.
			if(id != 0 && bounce == 0) {
				if(kCold) {
					const uint32_t pxy = cold[C_PXY * kBlock];
					sample_stream(fr, (int)(pxy & 0xffffu), (int)(pxy >> 16), si, (uint64_t)cold[C_PASS * kBlock], rng);
				} else sample_stream(fr, px, py, si, (uint64_t)pass, rng);
			}
#endif
			bool alive = shade<FEAT>(sc, fr, id, surf, hp, hn, bounce, rp, rd, att, col, rng);
			bounce++;
			if(!alive || bounce >= max_bounces) {
				si++;
				state = NEED_PATH;
				if(kCold) {
					const V3 so_far = v3(__uint_as_float(cold[C_SX * kBlock]), __uint_as_float(cold[C_SY * kBlock]), __uint_as_float(cold[C_SZ * kBlock]));
					const V3 now = so_far + clamp_hi(col, 1.0f);
					if(si == (int)cold[C_END * kBlock]) {
						const size_t it = (size_t)cold[C_ITEM_LO * kBlock] | ((size_t)cold[C_ITEM_HI * kBlock] << 32);
						finish_item<STORE>(job, it, now, true);
						state = NEED_PIXEL;
					} else {
						cold[C_SX * kBlock] = __float_as_uint(now.x); cold[C_SY * kBlock] = __float_as_uint(now.y); cold[C_SZ * kBlock] = __float_as_uint(now.z);
					}
				} else {
					sum = sum + clamp_hi(col, 1.0f);            // per-sample clamp before averaging (code/raycast.cpp:1263)
					if(si == si_end) {
						finish_item<STORE>(job, item, sum, true);
						state = NEED_PIXEL;
					}
				}
			}
		}
PT_MARK("shade_end");
		PT_TOCK(coop, MODE == CAST_GRID_COOP ? 15 : 12, t_loop);
	}

	PT_TOCK(coop, MODE == CAST_GRID_COOP ? 15 : 12, t_loop);
	PT_TOCK(coop, 0, t_kernel);
#ifdef PT_COOP_TIMING
	if(lane == 0) {
		for(int i = 0; i < 16; i++) PT_COUNT_(i, coop.tacc[i]);
		if(MODE != CAST_GRID_COOP) {      // how evenly the wavefronts finish: busy time per wave against the kernel's span
			const unsigned long long wall_end = wall_clock64();
			PT_COUNT_(2, wall_end - wall_start);
			PT_COUNT_(5, 1);
			atomicMin(&g_coop_stats[3], wall_start);
			atomicMax(&g_coop_stats[4], wall_end);
		}
	}
#endif
	// ray-cast count: one atomic per wave
	if(job.stats) {
		if(lane == 0) atomicAdd((unsigned long long*)job.stats + (blockIdx.x % ptl::kStatShards) * ptl::kStatStride, (unsigned long long)casts);
	}
}

// The brute-force variants run 1.3 % faster when the compiler optimises them for size (-Os), the grid variants 1 % slower.
// So the library is built from this file twice: once as it stands with -DPT_SPLIT_BRUTE, which only declares the four
// brute-force variants, and once with -Os -DPT_UNIT_BRUTE, which defines them and nothing else.  Without either macro
// the file is one unit with everything in it (listings, diagnostic builds).
#define PT_BRUTE_VARIANTS(PREFIX) \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, 0, false>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, 0, true>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_ALL, false>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_ALL, true>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_EVERY, false>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_EVERY, true>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, WAVES_SIX, false>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, WAVES_SIX, true>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_ALL | WAVES_SIX, false>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_ALL | WAVES_SIX, true>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, 0, STORE_PIXEL>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_ALL, STORE_PIXEL>(ptl::TraceJob); \
	PREFIX template __global__ void pt_k_trace<CAST_BRUTE, FEAT_EVERY, STORE_PIXEL>(ptl::TraceJob);
#if defined(PT_UNIT_BRUTE)
PT_BRUTE_VARIANTS()
#elif defined(PT_SPLIT_BRUTE)
PT_BRUTE_VARIANTS(extern)
#endif
}  // namespace ptk
using namespace ptk;

#if !defined(PT_UNIT_BRUTE)
namespace {

// Folds the range sums of every pixel, in range order, into the image (code/raycast.cpp:1263-1270:
// average of the per-sample clamped colours, clamped again).  A pixel whose ranges do not all carry
// this launch's serial number was not finished (aborted render) and keeps its previous content.
__global__ __launch_bounds__(kBlock) void pt_k_resolve(ptl::TraceJob job, size_t slots) {
	const size_t slot = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if(slot >= slots) return;
	const Scene sc = scene_view(job.scene);
	const FrameRef fr = (FrameRef)job.frame;
	int px, py;
	if(job.compact) {
		const int tile = (int)(slot / kGroup) * job.part_count + job.part_index, j = (int)(slot % kGroup);
		px = (tile % job.tiles_x) * 8 + (j & 7);
		py = (tile / job.tiles_x) * 8 + (j >> 3);
		if(px >= fr->width || py >= fr->height) return;
	} else {
		px = (int)(slot % (size_t)fr->width);
		py = (int)(slot / (size_t)fr->width);
	}
	const int t = (py % fr->tile_w) * fr->tile_w + (px % fr->tile_w);
	const int n_samples = sc.offsets[t + 1] - sc.offsets[t];
	const int ranges = 1 << job.ranges_log2;
	const float4* part = job.partial + slot;
	float4 first = part[0];
	bool whole = __float_as_uint(first.w) == job.serial;
	V3 sum = v3(first.x, first.y, first.z);
	for(int r = 1; r < ranges; r++) {
		float4 next = part[(size_t)r * slots];
		whole = whole && __float_as_uint(next.w) == job.serial;
		sum = sum + v3(next.x, next.y, next.z);
	}
	if(!whole) return;
	float* o = job.fb + slot * 3;
	if(n_samples == 0) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
	if(job.frame_index < 0) {
		V3 c = clamp_hi(sum / (float)n_samples, 1.0f);
		o[0] = c.x; o[1] = c.y; o[2] = c.z;
	} else {
		float* a = job.accum + slot * 3;
		V3 total = sum;
		if(job.frame_index > 0) total = v3(a[0] + sum.x, a[1] + sum.y, a[2] + sum.z);
		a[0] = total.x; a[1] = total.y; a[2] = total.z;
		V3 c = clamp_hi(total / (float)(job.frame_index + 1), 1.0f);
		o[0] = c.x; o[1] = c.y; o[2] = c.z;
	}
}

// After an aborted render: which pixels were finished (all their range sums carry the launch's serial).  pt_k_resolve has
// written exactly those; the host copies only them to the caller (code/app.cpp:1425-1428: the reference's workers stop
// between pixels and leave the rest of the caller's buffer alone).
__global__ __launch_bounds__(kBlock) void pt_k_finished(ptl::TraceJob job, size_t slots, uint8_t* mask) {
	const size_t slot = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if(slot >= slots) return;
	const FrameRef fr = (FrameRef)job.frame;
	if(job.compact) {
		const int tile = (int)(slot / kGroup) * job.part_count + job.part_index, j = (int)(slot % kGroup);
		const int px = (tile % job.tiles_x) * 8 + (j & 7), py = (tile / job.tiles_x) * 8 + (j >> 3);
		if(px >= fr->width || py >= fr->height) { mask[slot] = 0; return; }
	}
	const int ranges = 1 << job.ranges_log2;
	bool whole = true;
	for(int r = 0; r < ranges; r++) whole = whole && __float_as_uint(job.partial[slot + (size_t)r * slots].w) == job.serial;
	mask[slot] = whole ? 1 : 0;
}

// Cost estimate of every owned 8x8 tile, for the order the trace kernel deals them in: the ray casts of
// sixteen whole paths (camera sample 0 of every second pixel in x and y, traced and shaded exactly as
// the trace kernel does), four tiles per wavefront.  Only scheduling depends on it.
__global__ __launch_bounds__(kBlock) void pt_k_tile_cost(ptl::TraceJob job, int local_tiles, uint16_t* cost) {
	const int lane = lane_id();
	const int lt = (blockIdx.x * (kBlock / 64) + (int)threadIdx.x / 64) * 4 + (lane >> 4);
	const Scene sc = scene_view(job.scene);
	const FrameRef fr = (FrameRef)job.frame;
	const bool owned = lt < local_tiles;
	const int tile = (owned ? lt : 0) * job.part_count + job.part_index;
	const int probe = lane & 15;
	const int x = (tile % job.tiles_x) * 8 + (probe & 3) * 2 + ((probe >> 2) & 1), y = (tile / job.tiles_x) * 8 + (probe >> 2) * 2;
	uint32_t casts = 0;
	if(owned && x < fr->width && y < fr->height) {
		const int t = (y % fr->tile_w) * fr->tile_w + (x % fr->tile_w);
		const int base = sc.offsets[t];
		if(sc.offsets[t + 1] != base) {
			V3 rp, rd;
			const int lens = fr->aperture != 0.0f ? lens_start(fr, x, y) : 0;
			camera_ray(sc, fr, pixel_origin(fr, x, y), sc.samples[base], lens, rp, rd);
			Rng rng;
			sample_stream(fr, x, y, 0, 0, rng);
			V3 att = v3(1, 1, 1), col = v3(0, 0, 0);
			for(int bounce = 0; bounce < fr->max_bounces; bounce++) {
				V3 hp = v3(0, 0, 0), hn = v3(0, 0, 0);
				uint64_t unused = 0;
				Surface surf;
				surf.m0 = (v4f)(0.0f); surf.m1 = (v4f){ 0.0f, 0.0f, 0.0f, 1.0f }; surf.rough = 0;
				const int id = cast<false>(sc, rp, rd, hp, hn, surf, unused);
				casts++;
				if(!shade(sc, fr, id, surf, hp, hn, bounce, rp, rd, att, col, rng)) break;
			}
		}
	}
	for(int off = 8; off > 0; off >>= 1) casts += __shfl_down(casts, off, 64);      // sum over the tile's 16 lanes
	if(owned && probe == 0) cost[lt] = (uint16_t)(casts > 1023u ? 1023u : casts);
}

// Metrics pass: how many rays each camera sample's path casts (the bounce histogram of the harness's report).  One
// thread per pixel slot retraces every sample of its pixel with the trace kernel's own device functions and streams,
// so the paths are the frame's paths; the counts meet in a histogram in LDS and leave it with one atomic per bin and block.
constexpr int kHistBins = 65;
__global__ __launch_bounds__(kBlock) void pt_k_cast_histogram(ptl::TraceJob job, size_t slots, unsigned long long* histogram, int bins) {
	__shared__ unsigned int local[kHistBins];
	if(threadIdx.x < kHistBins) local[threadIdx.x] = 0;
	__syncthreads();
	const size_t slot = (size_t)blockIdx.x * kBlock + threadIdx.x;
	const Scene sc = scene_view(job.scene);
	const FrameRef fr = (FrameRef)job.frame;
	int px = -1, py = -1;
	if(slot < slots) {
		if(job.compact) {
			const int tile = (int)(slot / kGroup) * job.part_count + job.part_index, j = (int)(slot % kGroup);
			px = (tile % job.tiles_x) * 8 + (j & 7);
			py = (tile / job.tiles_x) * 8 + (j >> 3);
		} else {
			px = (int)(slot % (size_t)fr->width);
			py = (int)(slot / (size_t)fr->width);
		}
	}
	if(px >= 0 && px < fr->width && py < fr->height) {
		const int t = (py % fr->tile_w) * fr->tile_w + (px % fr->tile_w);
		const int base = sc.offsets[t], n_samples = sc.offsets[t + 1] - base;
		int s0 = 0, s1 = n_samples;
		uint64_t pass = 0;
		if(job.frame_index >= 0 && n_samples > 0) { s0 = job.frame_index % n_samples; s1 = s0 + 1; pass = (uint64_t)(job.frame_index / n_samples); }
		const int lens = fr->aperture != 0.0f ? lens_start(fr, px, py) : 0;
		const V3 origin = pixel_origin_tab(job, fr, px, py);
		for(int si = s0; si < s1; si++) {
			V3 rp, rd;
			camera_ray(sc, fr, origin, sc.samples[base + si], lens + si, rp, rd);
			Rng rng;
			sample_stream(fr, px, py, si, pass, rng);
			V3 att = v3(1, 1, 1), col = v3(0, 0, 0);
			int casts = 0;
			for(int bounce = 0; bounce < fr->max_bounces; bounce++) {
				V3 hp = v3(0, 0, 0), hn = v3(0, 0, 0);
				uint64_t unused = 0;
				Surface surf;
				surf.m0 = (v4f)(0.0f); surf.m1 = (v4f){ 0.0f, 0.0f, 0.0f, 1.0f }; surf.rough = 0;
				const int id = cast<false>(sc, rp, rd, hp, hn, surf, unused);
				casts++;
				if(!shade(sc, fr, id, surf, hp, hn, bounce, rp, rd, att, col, rng)) break;
			}
			atomicAdd(&local[casts < kHistBins ? casts : kHistBins - 1], 1u);
		}
	}
	__syncthreads();
	if(threadIdx.x < kHistBins && local[threadIdx.x] != 0) {
		const int bin = (int)threadIdx.x < bins ? (int)threadIdx.x : bins - 1;
		atomicAdd(&histogram[bin], (unsigned long long)local[threadIdx.x]);
	}
}

// Debug pass: entity hit by, and grid cells visited by, the primary ray of camera sample 0 of
// every pixel.  Same device functions as the trace kernel.
__global__ __launch_bounds__(kBlock) void pt_k_primary(ptl::SceneDev scene, const ptl::FrameDev* frame, int32_t* hit_id, uint64_t* cell_hash) {
	const Scene sc = scene_view(scene);
	const FrameRef fr = (FrameRef)frame;
	int i = blockIdx.x * kBlock + threadIdx.x;
	if(i >= fr->width * fr->height) return;
	int x = i % fr->width, y = i / fr->width;
	int t = (y % fr->tile_w) * fr->tile_w + (x % fr->tile_w);
	int base = sc.offsets[t];
	int count = sc.offsets[t + 1] - base;
	int id = -1;
	uint64_t h = 0;
	if(count > 0) {
		V3 rp, rd, hp, hn;
		int lens = fr->aperture != 0.0f ? lens_start(fr, x, y) : 0;
		camera_ray(sc, fr, pixel_origin(fr, x, y), sc.samples[base], lens, rp, rd);
		h = PT_HASH_INIT;
		Surface surf;
		id = cast<true>(sc, rp, rd, hp, hn, surf, h);
		if(sc.brute) h = 0;
	}
	if(hit_id) hit_id[i] = id;
	if(cell_hash) cell_hash[i] = h;
}

// One-ray queries (picking, focal-distance probe) and the intersection-level parity tests.
__global__ __launch_bounds__(kBlock) void pt_k_cast(ptl::SceneDev scene, int mode, const float* origins, const float* dirs, int n,
                                                    int32_t* hit_id, float* hit_pos, float* hit_nrm) {
	const Scene sc = scene_view(scene);
	int i = blockIdx.x * kBlock + threadIdx.x;
	if(i >= n) return;
	V3 rp = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
	V3 rd = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
	V3 hp = v3(0, 0, 0), hn = v3(0, 0, 0);
	uint64_t unused = 0;
	Surface surf;
	int id = mode == 1 ? cast_all<FEAT_EVERY, false>(sc, rp, rd, hp, hn, surf) : cast<false>(sc, rp, rd, hp, hn, surf, unused);
	if(id == 0) { hp = v3(0, 0, 0); hn = v3(0, 0, 0); }
	hit_id[i] = id;
	if(hit_pos) { hit_pos[3 * i] = hp.x; hit_pos[3 * i + 1] = hp.y; hit_pos[3 * i + 2] = hp.z; }
	if(hit_nrm) { hit_nrm[3 * i] = hn.x; hit_nrm[3 * i + 1] = hn.y; hit_nrm[3 * i + 2] = hn.z; }
}

// Arithmetic probes: the short forms of division and square root (pt_device.h) on caller-supplied operands,
// and sweeps that compare them on the device with the compiler's own a / b and sqrtf over ranges of bit patterns.
enum { ARITH_ROOT = 0, ARITH_OVER = 1, ARITH_NORMALISE = 2, ARITH_OVER_PI = 3, ARITH_RECIP = 4 };
__global__ __launch_bounds__(kBlock) void pt_k_arith(int op, const float* a, const float* b, int n, float* out) {
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if(i >= n) return;
	if(op == ARITH_ROOT) { out[i] = sq_root(a[i]); return; }
	const V3 v = v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
	V3 q;
	if(op == ARITH_OVER) q = v / b[i];
	else if(op == ARITH_NORMALISE) q = normalise(v);
	else if(op == ARITH_RECIP) q = recip3(v);
	else q = over_pi(v);
	out[3 * i] = q.x; out[3 * i + 1] = q.y; out[3 * i + 2] = q.z;
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {     // a bijective scrambler: related operands from one counter
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
__device__ __forceinline__ bool same_bits(float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p != p && q != q); }
// The long forms, kept from being folded into the short ones: plain C division and sqrtf.
__device__ __noinline__ float long_div(float a, float b) { return a / b; }
__device__ __noinline__ float long_sqrt(float x) { return sqrtf(x); }
__global__ __launch_bounds__(kBlock) void pt_k_arith_sweep(int op, uint32_t first, unsigned long long count, unsigned long long* mismatches) {
	unsigned long long bad = 0;
	for(unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < count; k += (unsigned long long)gridDim.x * kBlock) {
		const uint32_t bits = first + (uint32_t)k;
		const float x = __uint_as_float(bits);
		if(op == ARITH_ROOT) {
			bad += same_bits(sq_root(x), long_sqrt(x)) ? 0 : 1;
		} else {
			// every bit pattern in x; y and z (and the denominator) of related or of unrelated magnitude, by turns
			const uint32_t h = mix32(bits), g = mix32(bits ^ 0x9e3779b9u);
			const bool near = (h >> 31) != 0;
			const float y = near ? x * (0.001f + (float)(h & 0xffffu) * (1.0f / 256.0f)) : __uint_as_float(g);
			const float z = near ? x * -(0.5f + (float)((h >> 16) & 0x7fffu) * (1.0f / 64.0f)) : __uint_as_float(mix32(g));
			const V3 v = v3(x, y, z);
			V3 q, w;
			if(op == ARITH_OVER) { const float d = near ? x * (float)(1 + (g & 255u)) * 0.37f : __uint_as_float(mix32(h)); q = v / d; w = v3(long_div(x, d), long_div(y, d), long_div(z, d)); }
			else if(op == ARITH_RECIP) { q = recip3(v); w = v3(long_div(1.0f, x), long_div(1.0f, y), long_div(1.0f, z)); }
			else if(op == ARITH_NORMALISE) { q = normalise(v); const float l = long_sqrt(x * x + y * y + z * z); w = v3(long_div(x, l), long_div(y, l), long_div(z, l)); }
			else { const float pi = 3.14159274101257324f; q = over_pi(v); w = v3(long_div(x, pi), long_div(y, pi), long_div(z, pi)); }
			bad += (same_bits(q.x, w.x) && same_bits(q.y, w.y) && same_bits(q.z, w.z)) ? 0 : 1;
		}
	}
	for(int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
	if(lane_id() == 0 && bad) atomicAdd(mismatches, bad);
}

// Screenshot encoding (code/app.cpp:1541-1547, code/rt_math.cpp:351-353,3634-3640).  The reference's expression
// (a float pow from the C library, then a rounding to 0..255) is monotone with 256 outputs, so it is
// reproduced exactly by the 255 inputs at which the output steps up: the host finds them with the same
// C library the CPU path uses, the device counts how many of them an input has reached (eight halvings,
// no branch).  thresholds[255] is +infinity; a NaN reaches none and encodes to 0, as on the CPU.
__device__ __forceinline__ uint32_t srgb8_code(const float* thr, float v) {
	uint32_t reached = 0;
	#pragma unroll
	for(uint32_t step = 128; step > 0; step >>= 1) reached += thr[reached + step - 1] <= v ? step : 0u;
	return reached;
}
__global__ __launch_bounds__(kBlock) void pt_k_srgb8(const float* rgb, size_t n, const float* thresholds, uint8_t* out) {
	__shared__ float thr[256];
	thr[threadIdx.x] = threadIdx.x < 255 ? thresholds[threadIdx.x] : __builtin_inff();
	__syncthreads();
	// four values per thread: 16 bytes in, one word out
	const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
	if(i + 4 <= n) {
		const float4 v = *(const float4*)(rgb + i);
		const uint32_t word = srgb8_code(thr, v.x) | (srgb8_code(thr, v.y) << 8) | (srgb8_code(thr, v.z) << 16) | (srgb8_code(thr, v.w) << 24);
		*(uint32_t*)(out + i) = word;
	} else {
		for(size_t k = i; k < n; k++) out[k] = (uint8_t)srgb8_code(thr, rgb[k]);
	}
}
// The encoder over `count` consecutive float bit patterns from `first`: a histogram of the codes, and the number of
// inputs whose code differs from a plain count of the thresholds at or below them.
__global__ __launch_bounds__(kBlock) void pt_k_srgb8_sweep(uint32_t first, unsigned long long count, const float* thresholds,
                                                           unsigned long long* mismatches, unsigned long long* histogram) {
	__shared__ float thr[256];
	__shared__ unsigned int hist[256];
	thr[threadIdx.x] = threadIdx.x < 255 ? thresholds[threadIdx.x] : __builtin_inff();
	hist[threadIdx.x] = 0;
	__syncthreads();
	unsigned long long bad = 0;
	// a block walks whole chunks of 2^16 patterns, so that its LDS histogram cannot overflow a word between flushes
	for(unsigned long long chunk = blockIdx.x; chunk * 65536ull < count; chunk += gridDim.x) {
		const unsigned long long end = (chunk + 1) * 65536ull < count ? (chunk + 1) * 65536ull : count;
		for(unsigned long long k = chunk * 65536ull + threadIdx.x; k < end; k += kBlock) {
			const float v = __uint_as_float(first + (uint32_t)k);
			const uint32_t code = srgb8_code(thr, v);
			uint32_t plain = 0;
			for(int t = 0; t < 255; t++) plain += thr[t] <= v ? 1u : 0u;
			bad += code != plain ? 1 : 0;
			atomicAdd(&hist[code], 1u);
		}
		__syncthreads();
		if(hist[threadIdx.x]) atomicAdd(&histogram[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
		hist[threadIdx.x] = 0;
		__syncthreads();
	}
	for(int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
	if(lane_id() == 0 && bad) atomicAdd(mismatches, bad);
}

// Multi-GPU: tile-major partition buffers -> row-major image (after the framebuffer gather).
__global__ __launch_bounds__(kBlock) void pt_k_assemble(const float* parts, size_t stride, int part_count, int width, int height,
                                                        int tiles_x, float* rgb) {
	int i = blockIdx.x * kBlock + threadIdx.x;
	if(i >= width * height) return;
	int x = i % width, y = i / width;
	int tile = (y >> 3) * tiles_x + (x >> 3);
	int part = tile % part_count, local = tile / part_count;
	const float* src = parts + (size_t)part * stride + ((size_t)local * kGroup + (size_t)((y & 7) * 8 + (x & 7))) * 3;
	float* dst = rgb + (size_t)i * 3;
	dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

}  // namespace

namespace ptl {

const char* trace_kernel_name() { return "pt_k_trace"; }

// Diagnostic build only (-DPT_COOP_STATS): copies and clears the cooperative cast's event counters.
int coop_stats(unsigned long long* out16) {
#if defined(PT_COOP_STATS) || defined(PT_COOP_TIMING)
	unsigned long long zero[16] = { 0 };
	zero[3] = ~0ull;      // slot 3 is a minimum
	if(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_coop_stats), sizeof zero) != hipSuccess) return -1;
	if(hipMemcpyToSymbol(HIP_SYMBOL(g_coop_stats), zero, sizeof zero) != hipSuccess) return -1;
	return 16;
#else
	(void)out16;
	return 0;
#endif
}

namespace {
enum { V_BRUTE_PLAIN = 0, V_BRUTE = 1, V_GRID_LANE = 2, V_GRID_COOP = 3, V_BRUTE_PLANES = 4, V_GRID_COOP_PLANES = 5, V_BRUTE_PLAIN_SIX = 6, V_BRUTE_SIX = 7,
       V_GRID_COOP_CONES = 8, V_GRID_COOP_PLANES_CONES = 9, V_COUNT = 10 };
static_assert(V_COUNT == ptl::kTraceVariants, "TraceCaps holds one entry per variant");
// (the per-lane grid walk is the fallback and the A/B partner of the cooperative cast: one variant with everything in it)
void (*const kTraceKernels[3][V_COUNT])(TraceJob) = {
	{ pt_k_trace<CAST_BRUTE, 0, false>, pt_k_trace<CAST_BRUTE, FEAT_ALL, false>, pt_k_trace<CAST_GRID_LANE, FEAT_EVERY, false>, pt_k_trace<CAST_GRID_COOP, FEAT_ALL, false>,
	  pt_k_trace<CAST_BRUTE, FEAT_EVERY, false>, pt_k_trace<CAST_GRID_COOP, FEAT_EVERY, false>, pt_k_trace<CAST_BRUTE, WAVES_SIX, false>, pt_k_trace<CAST_BRUTE, FEAT_ALL | WAVES_SIX, false>,
	  pt_k_trace<CAST_GRID_COOP, FEAT_ALL | CONE_EXACT, false>, pt_k_trace<CAST_GRID_COOP, FEAT_EVERY | CONE_EXACT, false> },
	{ pt_k_trace<CAST_BRUTE, 0, true>, pt_k_trace<CAST_BRUTE, FEAT_ALL, true>, pt_k_trace<CAST_GRID_LANE, FEAT_EVERY, true>, pt_k_trace<CAST_GRID_COOP, FEAT_ALL, true>,
	  pt_k_trace<CAST_BRUTE, FEAT_EVERY, true>, pt_k_trace<CAST_GRID_COOP, FEAT_EVERY, true>, pt_k_trace<CAST_BRUTE, WAVES_SIX, true>, pt_k_trace<CAST_BRUTE, FEAT_ALL | WAVES_SIX, true>,
	  pt_k_trace<CAST_GRID_COOP, FEAT_ALL | CONE_EXACT, true>, pt_k_trace<CAST_GRID_COOP, FEAT_EVERY | CONE_EXACT, true> },
	// progressive frames written straight to the image (a one-sample frame never runs a six-wavefront variant: their places hold the five-wavefront ones)
	{ pt_k_trace<CAST_BRUTE, 0, STORE_PIXEL>, pt_k_trace<CAST_BRUTE, FEAT_ALL, STORE_PIXEL>, pt_k_trace<CAST_GRID_LANE, FEAT_EVERY, STORE_PIXEL>, pt_k_trace<CAST_GRID_COOP, FEAT_ALL, STORE_PIXEL>,
	  pt_k_trace<CAST_BRUTE, FEAT_EVERY, STORE_PIXEL>, pt_k_trace<CAST_GRID_COOP, FEAT_EVERY, STORE_PIXEL>, pt_k_trace<CAST_BRUTE, 0, STORE_PIXEL>, pt_k_trace<CAST_BRUTE, FEAT_ALL, STORE_PIXEL>,
	  pt_k_trace<CAST_GRID_COOP, FEAT_ALL | CONE_EXACT, STORE_PIXEL>, pt_k_trace<CAST_GRID_COOP, FEAT_EVERY | CONE_EXACT, STORE_PIXEL> } };
}

hipError_t query_trace_caps(TraceCaps* caps) {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if(e != hipSuccess) return e;
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, dev);
	if(e != hipSuccess) return e;
	caps->cus = prop.multiProcessorCount;
	for(int v = 0; v < V_COUNT; v++) {
		int b = 0;
		e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kTraceKernels[0][v], kBlock, 0);      // the write-through twins differ in one store
		if(e != hipSuccess) return e;
		caps->blocks_per_cu[v] = b > 0 ? b : 1;
	}
	return hipSuccess;
}

hipError_t launch_trace(const TraceJob& job, const TraceCaps& caps, hipStream_t stream, int* grid_blocks_out) {
	static const bool lane_walk = getenv("PT_GRID_PER_LANE") != nullptr;     // A/B switches: per-lane grid walk,
	static const bool full_brute = getenv("PT_BRUTE_FULL") != nullptr;       // the all-features brute-force variant for every scene
	const bool planes = (job.scene.features & FEAT_PLANE) != 0;       // a scene with planes runs the variants that know them
	int variant = job.scene.brute ? (planes ? V_BRUTE_PLANES : job.scene.features == 0 && !full_brute ? V_BRUTE_PLAIN : V_BRUTE)
	                              : (job.scene.coop && !lane_walk ? (planes ? V_GRID_COOP_PLANES : V_GRID_COOP) : V_GRID_LANE);
	// a scene with cones: the cooperative variants that also return the reference cone intersector's phantom hits (CONE_EXACT above)
	static const bool fast_cones = getenv("PT_FAST_CONES") != nullptr;
	const bool cones = (job.scene.features & CONE_EXACT) != 0 && !fast_cones;
	if(cones && variant == V_GRID_COOP) variant = V_GRID_COOP_CONES;
	if(cones && variant == V_GRID_COOP_PLANES) variant = V_GRID_COOP_PLANES_CONES;
	// Six wavefronts per SIMD for the brute-force variants when the launch is long enough to gain by them: a frame of every sample
	// with at least four work groups for every wavefront the chip then holds (PT_BRUTE_WAVES=5|6 forces one or the other, for A/B).
	if(variant == V_BRUTE_PLAIN || variant == V_BRUTE) {
		static const char* const forced = getenv("PT_BRUTE_WAVES");
		const int six = variant == V_BRUTE_PLAIN ? V_BRUTE_PLAIN_SIX : V_BRUTE_SIX;
		const long held = (long)caps.cus * caps.blocks_per_cu[six] * (kBlock / 64);
		const bool long_launch = job.frame_index < 0 && (long)job.n_groups >= 4 * held;
		if(forced ? forced[0] == '6' : long_launch) variant = six;
	}
	// persistent launch: every resident wave pulls groups of 64 work items until the partition is drained
	long blocks = ((long)job.n_groups + 3) / 4;
	long resident = (long)caps.cus * caps.blocks_per_cu[variant];
	if(blocks > resident) blocks = resident;
	// A short brute-force launch is faster with fewer workgroups than fit: every workgroup costs its first and its last
	// fetches, atomic operations on counters that all of them share, while the work a group brings is small - a
	// one-sample frame of the demo scene at 1080p takes 0.186 ms with all 1280, 0.133 ms with 512.  With a cost per workgroup
	// and a cost per group the best number grows with the root of the launch's work; four times the root of (groups x
	// group_work) fits the measured optima of one-sample frames of 4 050 ... 32 400 groups (256 ... 768 workgroups) and
	// leaves every launch of sixteen-sample items at what fits.  Beyond four workgroups per compute unit the rule is not
	// applied: a one-sample 4K frame (129 600 groups) measures 0.39 - 0.50 ms for 512 ... 1280 workgroups in no order, and
	// the full grid is as good as any (profiles/r03_ab/grid_size.txt).  PT_GRID_BLOCKS=n forces a size.
	if(job.scene.brute && job.group_work > 0.0f) {
		long best = (long)(4.0 * sqrt((double)job.n_groups * (double)job.group_work)) / 8 * 8;
		if(best < 8) best = 8;
		if(best < blocks && best <= 4L * caps.cus) blocks = best;
	}
	static const char* const grid_env = getenv("PT_GRID_BLOCKS");
	if(grid_env && atol(grid_env) > 0 && atol(grid_env) <= resident && atol(grid_env) <= ((long)job.n_groups + 3) / 4) blocks = atol(grid_env);
	if(blocks < 1) blocks = 1;
	if(grid_blocks_out) *grid_blocks_out = (int)blocks;
	hipLaunchKernelGGL(kTraceKernels[job.pixel_store ? STORE_PIXEL : (job.serial >> 31) & 1u][variant], dim3((unsigned)blocks), dim3(kBlock), 0, stream, job);
	return hipGetLastError();
}

hipError_t launch_tile_cost(const TraceJob& job, int local_tiles, uint16_t* cost, hipStream_t stream) {
	if(local_tiles <= 0) return hipSuccess;
	const int per_block = kBlock / 64 * 4;
	hipLaunchKernelGGL(pt_k_tile_cost, dim3((unsigned)((local_tiles + per_block - 1) / per_block)), dim3(kBlock), 0, stream, job, local_tiles, cost);
	return hipGetLastError();
}

hipError_t launch_resolve(const TraceJob& job, size_t slots, hipStream_t stream) {
	if(slots == 0) return hipSuccess;
	hipLaunchKernelGGL(pt_k_resolve, dim3((unsigned)((slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job, slots);
	return hipGetLastError();
}

hipError_t launch_finished(const TraceJob& job, size_t slots, uint8_t* mask, hipStream_t stream) {
	if(slots == 0) return hipSuccess;
	hipLaunchKernelGGL(pt_k_finished, dim3((unsigned)((slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job, slots, mask);
	return hipGetLastError();
}

hipError_t launch_cast_histogram(const TraceJob& job, size_t slots, unsigned long long* histogram, int bins, hipStream_t stream) {
	if(slots == 0) return hipSuccess;
	hipLaunchKernelGGL(pt_k_cast_histogram, dim3((unsigned)((slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job, slots, histogram, bins);
	return hipGetLastError();
}

hipError_t launch_primary_aov(const SceneDev& sc, const FrameDev* fr, int width, int height, int32_t* hit_id, uint64_t* cell_hash, hipStream_t stream) {
	int n = width * height;
	hipLaunchKernelGGL(pt_k_primary, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sc, fr, hit_id, cell_hash);
	return hipGetLastError();
}

hipError_t launch_cast(const SceneDev& sc, int mode, const float* origins, const float* dirs, int n, int32_t* hit_id, float* hit_pos,
                       float* hit_nrm, hipStream_t stream) {
	if(n <= 0) return hipSuccess;
	hipLaunchKernelGGL(pt_k_cast, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sc, mode, origins, dirs, n, hit_id, hit_pos, hit_nrm);
	return hipGetLastError();
}

// ---- ray queries for callers that cannot allocate ----------------------------------------------------------
// Staging of pt_cast_rays_noalloc: origins, directions, hit positions, hit normals (three floats a ray each) and
// ids of at most kStaticRays rays.  One copy of the array exists per GPU for the life of the process; host code
// reaches it only through the address hipGetSymbolAddress reports for the current device.
static __device__ __align__(256) unsigned char scratch[kStaticRays * 52];

hipError_t static_scratch(void** ptr, size_t* bytes) {
	void* p = nullptr;
	hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
	if(e != hipSuccess) return e;
	*ptr = p;
	*bytes = sizeof scratch;
	return hipSuccess;
}

hipError_t launch_arith(int op, const float* a, const float* b, int n, float* out, hipStream_t stream) {
	if(n <= 0) return hipSuccess;
	hipLaunchKernelGGL(pt_k_arith, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, op, a, b, n, out);
	return hipGetLastError();
}

hipError_t launch_arith_sweep(int op, uint32_t first, unsigned long long count, unsigned long long* mismatches, hipStream_t stream) {
	if(count == 0) return hipSuccess;
	unsigned long long blocks = (count + kBlock - 1) / kBlock;
	if(blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(pt_k_arith_sweep, dim3((unsigned)blocks), dim3(kBlock), 0, stream, op, first, count, mismatches);
	return hipGetLastError();
}

hipError_t launch_srgb8(const float* rgb, size_t n, const float* thresholds, uint8_t* out, hipStream_t stream) {
	if(n == 0) return hipSuccess;
	const size_t threads = (n + 3) / 4;
	hipLaunchKernelGGL(pt_k_srgb8, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, rgb, n, thresholds, out);
	return hipGetLastError();
}

hipError_t launch_srgb8_sweep(uint32_t first_bits, unsigned long long count, const float* thresholds, unsigned long long* mismatches,
                              unsigned long long* histogram, hipStream_t stream) {
	if(count == 0) return hipSuccess;
	unsigned long long blocks = (count + 65535ull) / 65536ull;
	if(blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(pt_k_srgb8_sweep, dim3((unsigned)blocks), dim3(kBlock), 0, stream, first_bits, count, thresholds, mismatches, histogram);
	return hipGetLastError();
}

hipError_t launch_assemble(const float* parts, size_t stride, int part_count, int width, int height, float* rgb, hipStream_t stream) {
	int n = width * height;
	int tiles_x = (width + 7) / 8;
	hipLaunchKernelGGL(pt_k_assemble, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, parts, stride, part_count, width, height, tiles_x, rgb);
	return hipGetLastError();
}

}  // namespace ptl
#endif   // !PT_UNIT_BRUTE
