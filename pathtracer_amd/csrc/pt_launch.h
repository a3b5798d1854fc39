// Host-callable launchers of the device passes (implemented in pt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptl {

#define PT_PROGRESS_WORD 16      // words from the abort word to the progress word (a cache line of its own)

// Plain-data mirrors of the device-side Scene / Frame structs (pt_device.h) so that host code
// does not need to include device code.
struct Vec { float x, y, z; };

// Slots of an entity's 128-byte device record (eight float4).  What every intersection test reads sits in the first
// 64 bytes, what only a hit needs (rotation and scale back to world space, the material) in the second.  (Measured on the
// 100 000-entity scene: the bytes fetched per frame are the same as with the two halves interleaved - a miss in L2 brings
// the whole line in either way.)
//   REC_POS pos.xyz | bounding-sphere radius     REC_ROTINV rotInverse (w,x,y,z); a plane: its normal rot * (0,0,1)     REC_DIM dim.xyz | type + flags<<8 (int bits)
//   REC_SCALEINV dimScaleInverse.xyz | -         (an unrotated box keeps its corners pos - dim/2, pos + dim/2 in REC_ROTINV, REC_SCALEINV)
//   REC_ROT rot (w,x,y,z)     REC_SCALE dimScale.xyz | roughness     REC_COLOR colorLinear.rgb | reflectionMod     REC_EMIT emitColorLinear.rgb | refractiveIndex
enum { REC_POS = 0, REC_ROTINV = 1, REC_DIM = 2, REC_SCALEINV = 3, REC_ROT = 4, REC_SCALE = 5, REC_COLOR = 6, REC_EMIT = 7, REC_SLOTS = 8 };
// float4 slots from one record to the next (a diagnostic build may spread the records to see what their cache lines cost)
#ifndef PT_REC_STRIDE
#define PT_REC_STRIDE 8
#endif
enum { REC_STRIDE = PT_REC_STRIDE };
static_assert(REC_STRIDE >= REC_SLOTS, "a record has eight slots");

struct GridDev { int cmin[3], cnt[3], cs[3]; float inv_cs[3]; int stride[3]; };
// Where cell (x, y, z) sits in cell_start and, with it, where its list sits among the lists: the axis along which the cells are
// thinnest runs fastest, because a ray crosses most cell bounds along it - the cells it visits one after the other, and
// their lists, then lie side by side in memory (a scene of flat cells: a quarter fewer L2 requests than with x fastest).
// Axes of equal cell size keep the order x, y, z, which is the reference's order (code/raycast.cpp:619-625).
inline void grid_layout(GridDev& g) {
	int axes[3] = { 0, 1, 2 };
	for(int i = 1; i < 3; i++)
		for(int j = i; j > 0 && g.cs[axes[j]] < g.cs[axes[j - 1]]; j--) { int t = axes[j]; axes[j] = axes[j - 1]; axes[j - 1] = t; }
	g.stride[axes[0]] = 1;
	g.stride[axes[1]] = g.cnt[axes[0]];
	g.stride[axes[2]] = g.cnt[axes[0]] * g.cnt[axes[1]];
}
// Whether an entity's own bounding sphere (centre p, radius r: pt::filter_sphere) comes near the box of cell (x, y, z).  The
// reference lists an entity in every cell its axis-aligned box covers (code/app.cpp:1297-1312), corners included that the
// entity itself stays clear of.  Lists without such entries (the optional cell cull, and the neighbour bits that go with it)
// are shorter, but not equivalent: the reference accepts a hit in the listed cell that holds the COMPUTED hit point
// (code/raycast.cpp:957-983), and a point computed from far away or at a grazing angle lies some way along the ray from the
// true surface, now and then in a cell the entity does not reach.  Host and device builds decide with these same float
// operations (no contraction in either).
__host__ __device__ inline bool reaches_cell(const GridDev& g, float px, float py, float pz, float r, int x, int y, int z) {
	const float p[3] = { px, py, pz };
	const int c[3] = { x, y, z };
	float dist2 = 0.0f, far = 0.0f;
	for(int a = 0; a < 3; a++) {
		const float lo = (float)((c[a] + g.cmin[a]) * g.cs[a]), hi = (float)((c[a] + g.cmin[a] + 1) * g.cs[a]);
		const float d = p[a] < lo ? lo - p[a] : (p[a] > hi ? p[a] - hi : 0.0f);
		dist2 = dist2 + d * d;
		const float m = p[a] < 0.0f ? -p[a] : p[a];
		far = m > far ? m : far;
	}
	const float reach = (r * 1.001f + 1e-3f) + 1e-5f * (far + r);
	return !(dist2 > reach * reach);           // (an infinite radius - a plane - reaches everything)
}
inline bool grid_layout_is_reference(const GridDev& g) { return g.stride[0] == 1 && g.stride[1] == g.cnt[0] && g.stride[2] == g.cnt[0] * g.cnt[1]; }
__host__ __device__ inline size_t grid_cell(const GridDev& g, int x, int y, int z) {
	return (size_t)x * (size_t)g.stride[0] + (size_t)y * (size_t)g.stride[1] + (size_t)z * (size_t)g.stride[2];
}
__host__ __device__ inline void grid_cell_coords(const GridDev& g, size_t cell, uint32_t at[3]) {
	for(int a = 0; a < 3; a++) at[a] = (uint32_t)((cell / (size_t)g.stride[a]) % (size_t)g.cnt[a]);
}

// cell_start[c] holds, above the 26 bits of the list's first index, six flags: bit 26 + f set = every entity of cell c's list is
// also listed in the neighbour cell across face f (-x +x -y +y -z +z).  A ray that enters c through f from a cell whose list
// the cooperative cast has just collected finds nothing new in c and skips its list (in scenes of flat cells, where entities span
// many layers, most lists a ray meets are such repeats).  Every reader masks the flags off; cell_start[cells] carries none.
constexpr uint32_t CELL_START_MASK = 0x03ffffffu;
constexpr int CELL_FLAG_SHIFT = 26;

// A list entry of the cooperative cast, 16 bytes: the entity's bounding sphere, its intersection kind (tag bits 0..2), in which
// of the six neighbour cells it is listed as well (tag bits 3..8: -x +x -y +y -z +z) - and its id, so that collecting
// candidates reads one array, not two.  The id (23 bits) rides where the sphere can spare bits: bits 0..6 in tag bits 9..15,
// bits 7..12 in the low six mantissa bits of the centre's x, 13..17 and 18..22 in the low five of y and z.  The radius (tag
// bits 16..31, rounded up) grows by what those bits can move the centre, so the sphere still encloses the entity's own.
constexpr uint32_t ENTRY_ID_LIMIT = 1u << 23;
__host__ __device__ inline float entry_ulp(float v) {          // spacing of the floats at |v| (v finite)
	union { float f; uint32_t u; } w;
	w.f = v;
	uint32_t e = (w.u >> 23) & 255u;
	if(e < 24u) e = 24u;                                        // a tiny or zero coordinate: an upper bound is enough
	w.u = (e - 23u) << 23;
	return w.f;
}
__host__ __device__ inline void pack_entry(float x, float y, float z, float radius, uint32_t kind_and_neighbours, uint32_t id, uint32_t out[4]) {
	union { float f; uint32_t u; } w;
	const float pad = (64.0f * entry_ulp(x) + 32.0f * entry_ulp(y)) + 32.0f * entry_ulp(z);
	w.f = (radius + pad) * 1.000001f;
	uint32_t r = w.u;
	if(r & 0xffffu) r = (r & 0xffff0000u) + 0x10000u;
	w.f = x; out[0] = (w.u & ~63u) | ((id >> 7) & 63u);
	w.f = y; out[1] = (w.u & ~31u) | ((id >> 13) & 31u);
	w.f = z; out[2] = (w.u & ~31u) | ((id >> 18) & 31u);
	out[3] = r | ((id & 127u) << 9) | (kind_and_neighbours & 511u);
}
__host__ __device__ inline uint32_t entry_id(uint32_t x, uint32_t y, uint32_t z, uint32_t tag) {
	return ((tag >> 9) & 127u) | ((x & 63u) << 7) | ((y & 31u) << 13) | ((z & 31u) << 18);
}

struct SceneDev {
	const float4* geo;
	const uint32_t* cell_start;
	const uint32_t* cell_items;
	const float4* cell_entries; // cooperative cast: one float4 per cell_items entry (pack_entry)
	const uint32_t* obj_range;  // per entity: inclusive cell range, 10 bits per axis, and its intersection kind (2 words)
	const float4* dirs;
	const float2* disc;
	const float2* samples;
	const int32_t* offsets;     // tile_w^2 + 1 sample offsets
	const uint32_t* plane_ids;  // the entities that are planes, ascending (every ray that misses the grid tests them all)
	int n_planes;
	int n;
	int brute;
	int coop;                   // 1: the wave-cooperative cast may be used (ranges fit their packing)
	int features;               // FEAT_* bits (pt_device.h) of what the entities need: unit-space kinds, glass, roughness, planes; 32: there are cones
	GridDev grid;
};

struct FrameDev {
	Vec cam_pos, cam_dir, right, down, back, right_width, down_height, right_px, down_px, top_left;
	Vec light_dir, bg, ambient, light;
	float focal_distance, aperture;
	int width, height, tile_w;
	int max_bounces;
	uint64_t seed;
};

// Every wavefront adds its ray casts to a device counter when it ends.  One counter for all of them is one address for six
// thousand atomic operations that arrive together at the end of a launch and are served one after the other (a counter serves
// about 88 a microsecond): 35 us of the 1.57 ms demo frame, 22 us of the 0.30 ms an eighth of it takes.  Sixty-four counters,
// a cache line apart, summed by the host.
constexpr int kStatShards = 64, kStatStride = 16, kStatBytes = kStatShards * kStatStride * 8;

// Work is handed to wavefronts in groups of 64 items = one 8x8 image tile x one range of each
// pixel's samples.  A context renders the tiles t with t % part_count == part_index.
struct TraceJob {
	SceneDev scene;
	const FrameDev* frame;      // device copy of the per-frame constants
	float4* partial;            // [range][pixel slot]: sum of the range's clamped sample colours | launch serial
	size_t slots;               // pixel slots of the framebuffer (pixels, or 64 per owned tile when compact)
	const float* col_frac;      // x / width for every column, y / height for every row
	const float* row_frac;
	uint32_t serial;            // marks the range sums of this launch (31 bits); top bit set: launch the kernel twin whose sums bypass the caches on
	                            // their way to memory, so that a snapshot copied out while the frame is being traced finds the finished ones
	int ranges_log2;            // a pixel's samples are summed in 2^ranges_log2 even, contiguous ranges.  (Ranges of falling length -
	                            // halves, or even parts with the last one cut into 8, 4 and 4 samples - dealt so that a launch ends on
	                            // the short ones were built and measured: an eighth of the demo frame took 0.32 - 0.36 ms instead of
	                            // 0.31; a long first range makes the dearest tile's item the critical path, and the extra groups cost
	                            // what the finer end saves.)
	float* fb;                  // row-major image, or tile-major [local tile][64][3] when compact
	float* accum;               // progressive accumulation buffer (same layout as fb), may be null
	const int32_t* tile_order;  // owned tiles in the order they are dealt (dearest first), or null for index order
	uint32_t* group_counter;    // device words: this launch's group counters, zero when it starts
	uint32_t* group_counter_next;   // the next launch's counters, which this launch zeroes (the two sets take turns, so no fill runs between frames)
	const volatile uint32_t* abort_flag;   // host-pinned: word 0 non-zero = stop fetching tiles; word PT_PROGRESS_WORD receives the number of
	                                       // groups handed out so far at every look at word 0
	uint64_t* stats;            // device: ray casts, in kStatShards counters kStatStride words apart (a workgroup adds to shard blockIdx % kStatShards)
	int tiles_x, tiles_y;
	int n_groups;               // tiles owned by this partition x ranges
	float group_work;           // what a group costs, in samples per item of the demo scene's kind (a path of 1.4 casts against two entities):
	                            // samples per item x casts per path / 1.4 x entities / 2, from the tile-cost probe; sizes short brute-force launches
	int part_index, part_count;
	int frame_index;            // -1: every sample; >= 0: one sample, progressive
	int compact;
	int pixel_store;            // a progressive frame (frame_index >= 0, one range) whose lanes add their sample to `accum` and write the pixel
	                            // themselves: no range sums are written and launch_resolve is not to follow (the caller of launch_trace sets it)
};

// What a launch of the persistent trace kernel needs to know of the device: filled once per context, under that
// context's device (compute units, resident workgroups per unit of each kernel variant).
constexpr int kTraceVariants = 10;
struct TraceCaps { int cus; int blocks_per_cu[kTraceVariants]; };
hipError_t query_trace_caps(TraceCaps* caps);
hipError_t launch_trace(const TraceJob& job, const TraceCaps& caps, hipStream_t stream, int* grid_blocks_out);
hipError_t launch_resolve(const TraceJob& job, size_t slots, hipStream_t stream);
// After an aborted launch: mask[slot] = 1 for the pixels whose range sums all carry the launch's serial (the ones
// launch_resolve wrote), 0 for the others.
hipError_t launch_finished(const TraceJob& job, size_t slots, uint8_t* mask, hipStream_t stream);
hipError_t launch_tile_cost(const TraceJob& job, int local_tiles, uint16_t* cost, hipStream_t stream);   // cost: ray casts of 16 probe paths per owned tile, at most 1023
// Metrics: histogram[min(k, bins - 1)] += camera samples of the job's partition whose path casts k rays (k up to 64; bins >= 2, zeroed by the caller)
hipError_t launch_cast_histogram(const TraceJob& job, size_t slots, unsigned long long* histogram, int bins, hipStream_t stream);
hipError_t launch_primary_aov(const SceneDev& sc, const FrameDev* fr, int width, int height, int32_t* hit_id, uint64_t* cell_hash, hipStream_t stream);
// mode 0: the tracer's cast (brute force under the scene's threshold, else the grid walk); 1: the editor's pick, every
// entity tested whatever the scene's size and whether or not the ray meets the grid (code/app.cpp:1637)
hipError_t launch_cast(const SceneDev& sc, int mode, const float* origins, const float* dirs, int n, int32_t* hit_id,
                       float* hit_pos, float* hit_nrm, hipStream_t stream);
// Arithmetic probes of the short division / square-root forms: op 0 root(a[i]), 1 a[3i..] / b[i], 2 normalise(a[3i..]),
// 3 a[3i..] / pi; the sweep counts, over `count` consecutive bit patterns from `first`, results that differ from the
// compiler's own division and sqrtf.
hipError_t launch_arith(int op, const float* a, const float* b, int n, float* out, hipStream_t stream);
hipError_t launch_arith_sweep(int op, uint32_t first, unsigned long long count, unsigned long long* mismatches, hipStream_t stream);
// thresholds: 255 ascending floats, thresholds[k] = the smallest input that encodes to k + 1
hipError_t launch_srgb8(const float* rgb, size_t n, const float* thresholds, uint8_t* out, hipStream_t stream);
hipError_t launch_srgb8_sweep(uint32_t first_bits, unsigned long long count, const float* thresholds, unsigned long long* mismatches, unsigned long long* histogram, hipStream_t stream);
// Tile-major partition buffers (part_count of them, each `stride` floats apart) -> row-major image.
hipError_t launch_assemble(const float* parts, size_t stride, int part_count, int width, int height, float* rgb, hipStream_t stream);
// Device-side grid build (pt_gridbuild.hip).
struct GridBuildJob {
	const float4* geo;          // entity records (position | bounding radius in slot 0, type + flags in slot 3)
	const float4* bbox;         // two per entity: world-space bounding-box extents | radius of the sphere the cooperative cast's filter tests,
	                            // that sphere's centre (the entity's position, except for a cone: pt::filter_sphere) | 0
	int n;
	GridDev grid;
	uint32_t* obj_range;        // out: 2 words per entity
	uint32_t* cell_start;       // out: cells + 1 words
	uint32_t* cursor;           // scratch: cells words
	uint32_t* block_sums;       // scratch: grid_scan_blocks(cells) words
	uint32_t* cell_items;       // out (second pass)
	float4* cell_entries;       // out (second pass), may be null
	uint32_t* longest;          // out: length of the longest list (one word)
	int cull;                   // leave an entity out of the cells its own bounding sphere does not reach (reaches_cell)
};
size_t grid_scan_blocks(size_t cells);
hipError_t launch_grid_count_scan(const GridBuildJob& job, size_t cells, hipStream_t stream);
hipError_t launch_grid_fill(const GridBuildJob& job, size_t cells, hipStream_t stream);
// The statically allocated staging area of the allocation-free ray query, as the CURRENT device addresses it
// (every GPU holds its own copy at its own address), and its size: room for kStaticRays rays.
constexpr int kStaticRays = 4096;
hipError_t static_scratch(void** ptr, size_t* bytes);
const char* trace_kernel_name();
int coop_stats(unsigned long long* out16);

}  // namespace ptl
