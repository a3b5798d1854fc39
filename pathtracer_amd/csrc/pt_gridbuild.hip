// Uniform-grid build on the device: the reference fills its per-cell object lists on the main thread
// at render start (code/app.cpp:1278-1354); here the same lists come out of four passes
//   count  one thread per entity: its inclusive cell range (code/app.cpp:1297-1312), one count per cell
//   scan   inclusive prefix sum over the cells (block scan, scan of the block totals, add back)
//   fill   one thread per entity: appends its id to every cell of its range
//   order  one thread per cell: sorts the cell's list by id (the reference appends entities in id order,
//          and equal-distance hits go to the lowest id), then writes the 16-byte list entries the
//          wave-cooperative cast reads
// The result is word for word the CSR the host build produces (pt_host.cpp: build_grid).
#include "pt_api.h"
#include "pt_launch.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 16;
constexpr int kPerBlock = kBlock * kPerThread;     // cells one scan block covers
constexpr int kTopThreads = 1024;

// clampInt goes through the float clampMax (code/rt_math.cpp:179-181)
__device__ __forceinline__ int clamp_cell(int v, int hi) {
	int a = v < 0 ? 0 : v;
	float f = (float)a, h = (float)hi;
	return (int)(f > h ? h : f);
}

struct CellRange { int a[3], b[3]; };

__device__ __forceinline__ CellRange range_of(const ptl::GridBuildJob& job, int i) {
	const float4 p = job.geo[(size_t)i * ptl::REC_STRIDE];          // position | bounding radius
	const float4 box = job.bbox[2 * (size_t)i];
	const float pos[3] = { p.x, p.y, p.z }, dim[3] = { box.x, box.y, box.z };
	CellRange r;
	for(int k = 0; k < 3; k++) {
		const float half = dim[k] / 2;
		const float mn = pos[k] - half, mx = pos[k] + half;
		int lo = (int)(mn / job.grid.cs[k]) - job.grid.cmin[k];
		if(mn < 0) lo--;
		int hi = (int)(mx / job.grid.cs[k]) - job.grid.cmin[k];
		if(mx < 0) hi--;
		r.a[k] = clamp_cell(lo, job.grid.cnt[k] - 1);
		r.b[k] = clamp_cell(hi, job.grid.cnt[k] - 1);
	}
	return r;
}

// Packed inclusive range and intersection kind of every entity; one count per cell and entity.
__global__ __launch_bounds__(kBlock) void pt_k_grid_count(ptl::GridBuildJob job) {
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if(i >= job.n) return;
	const CellRange r = range_of(job, i);
	const int tf = __float_as_int(job.geo[(size_t)i * ptl::REC_STRIDE + ptl::REC_DIM].w);
	const int type = tf & 0xff, flags = tf >> 8;
	const uint32_t kind = type == PT_GEOM_SPHERE ? ((flags & 2) ? 0u : 3u) : type == PT_GEOM_BOX ? ((flags & 1) ? 2u : 1u) : type == PT_GEOM_CYLINDER ? 4u : type == PT_GEOM_CONE ? 5u : 6u;
	job.obj_range[2 * (size_t)i] = (uint32_t)r.a[0] | ((uint32_t)r.a[1] << 10) | ((uint32_t)r.a[2] << 20) | ((kind & 3u) << 30);
	job.obj_range[2 * (size_t)i + 1] = (uint32_t)r.b[0] | ((uint32_t)r.b[1] << 10) | ((uint32_t)r.b[2] << 20) | ((kind >> 2) << 30);
	const float4 p = job.bbox[2 * (size_t)i + 1];              // the filter sphere's centre
	const float radius = job.bbox[2 * (size_t)i].w;
	for(int z = r.a[2]; z <= r.b[2]; z++) for(int y = r.a[1]; y <= r.b[1]; y++) for(int x = r.a[0]; x <= r.b[0]; x++) {
		if(job.cull && !ptl::reaches_cell(job.grid, p.x, p.y, p.z, radius, x, y, z)) continue;
		atomicAdd(&job.cell_start[ptl::grid_cell(job.grid, x, y, z) + 1], 1u);
	}
}

// Inclusive scan of kPerBlock values per block, in place; the block's total goes to totals[block].
__global__ __launch_bounds__(kBlock) void pt_k_scan_blocks(uint32_t* values, size_t n, uint32_t* totals) {
	__shared__ uint32_t sums[kBlock];
	const size_t base = (size_t)blockIdx.x * kPerBlock + (size_t)threadIdx.x * kPerThread;
	uint32_t v[kPerThread];
	uint32_t run = 0;
	for(int k = 0; k < kPerThread; k++) {
		v[k] = base + k < n ? values[base + k] : 0u;
		run += v[k];
		v[k] = run;
	}
	sums[threadIdx.x] = run;
	__syncthreads();
	for(int off = 1; off < kBlock; off <<= 1) {
		uint32_t add = (int)threadIdx.x >= off ? sums[threadIdx.x - off] : 0u;
		__syncthreads();
		sums[threadIdx.x] += add;
		__syncthreads();
	}
	const uint32_t before = threadIdx.x > 0 ? sums[threadIdx.x - 1] : 0u;
	for(int k = 0; k < kPerThread; k++) if(base + k < n) values[base + k] = v[k] + before;
	if(threadIdx.x == kBlock - 1) totals[blockIdx.x] = sums[kBlock - 1];
}

// Exclusive scan of the block totals, in place, by one block.
__global__ __launch_bounds__(kTopThreads) void pt_k_scan_totals(uint32_t* totals, int n) {
	__shared__ uint32_t sums[kTopThreads];
	const int per = (n + kTopThreads - 1) / kTopThreads;
	const int base = threadIdx.x * per;
	uint32_t run = 0;
	for(int k = 0; k < per; k++) if(base + k < n) run += totals[base + k];
	sums[threadIdx.x] = run;
	__syncthreads();
	for(int off = 1; off < kTopThreads; off <<= 1) {
		uint32_t add = (int)threadIdx.x >= off ? sums[threadIdx.x - off] : 0u;
		__syncthreads();
		sums[threadIdx.x] += add;
		__syncthreads();
	}
	uint32_t before = threadIdx.x > 0 ? sums[threadIdx.x - 1] : 0u;
	for(int k = 0; k < per; k++) if(base + k < n) {
		const uint32_t t = totals[base + k];
		totals[base + k] = before;
		before += t;
	}
}

__global__ __launch_bounds__(kBlock) void pt_k_scan_add(uint32_t* values, size_t n, const uint32_t* offsets) {
	const size_t base = (size_t)blockIdx.x * kPerBlock + (size_t)threadIdx.x * kPerThread;
	const uint32_t add = offsets[blockIdx.x];
	for(int k = 0; k < kPerThread; k++) if(base + k < n) values[base + k] += add;
}

// Length of the longest list: the ordering pass sorts a list with one thread, so the host wants to know.
__global__ __launch_bounds__(kBlock) void pt_k_grid_longest(const uint32_t* cell_start, size_t cells, uint32_t* longest) {
	const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
	uint32_t len = c < cells ? cell_start[c + 1] - cell_start[c] : 0u;
	for(int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_down(len, off, 64); len = o > len ? o : len; }
	if((threadIdx.x & 63) == 0 && len > 0) atomicMax(longest, len);
}

__global__ __launch_bounds__(kBlock) void pt_k_grid_fill(ptl::GridBuildJob job) {
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if(i >= job.n) return;
	const CellRange r = range_of(job, i);
	const float4 p = job.bbox[2 * (size_t)i + 1];
	const float radius = job.bbox[2 * (size_t)i].w;
	for(int z = r.a[2]; z <= r.b[2]; z++) for(int y = r.a[1]; y <= r.b[1]; y++) for(int x = r.a[0]; x <= r.b[0]; x++) {
		if(job.cull && !ptl::reaches_cell(job.grid, p.x, p.y, p.z, radius, x, y, z)) continue;
		const size_t c = ptl::grid_cell(job.grid, x, y, z);
		job.cell_items[job.cell_start[c] + atomicAdd(&job.cursor[c], 1u)] = (uint32_t)i;
	}
}

// Ascending ids inside every cell, then the cooperative cast's list entries (ptl::pack_entry).
__global__ __launch_bounds__(kBlock) void pt_k_grid_order(ptl::GridBuildJob job, size_t cells) {
	const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if(c >= cells) return;
	// (the neighbour thread may already have put its flags into cell_start[c + 1]: they are masked off)
	const uint32_t k0 = job.cell_start[c] & ptl::CELL_START_MASK, k1 = job.cell_start[c + 1] & ptl::CELL_START_MASK;
	uint32_t* list = job.cell_items;
	for(uint32_t k = k0 + 1; k < k1; k++) {              // insertion sort: lists are short
		const uint32_t id = list[k];
		uint32_t j = k;
		while(j > k0 && list[j - 1] > id) { list[j] = list[j - 1]; j--; }
		list[j] = id;
	}
	if(!job.cell_entries) return;
	uint32_t at[3];
	ptl::grid_cell_coords(job.grid, c, at);
	uint32_t all_listed = 63u;            // per face: every entity of this list is listed across it too
	for(uint32_t k = k0; k < k1; k++) {
		const uint32_t id = list[k];
		const uint32_t lo = job.obj_range[2 * (size_t)id], hi = job.obj_range[2 * (size_t)id + 1];
		uint32_t tag = (lo >> 30) | ((hi >> 30) << 2);
		const float4 p = job.bbox[2 * (size_t)id + 1];
		const float radius = job.bbox[2 * (size_t)id].w;
		for(int a = 0; a < 3; a++) {
			// listed in the neighbour across this face: inside the entity's cell range, and not left out of that cell's list
			int n[3] = { (int)at[0], (int)at[1], (int)at[2] };
			n[a] = (int)at[a] - 1;
			if(at[a] > ((lo >> (10 * a)) & 1023u) && (!job.cull || ptl::reaches_cell(job.grid, p.x, p.y, p.z, radius, n[0], n[1], n[2]))) tag |= 1u << (3 + 2 * a);
			n[a] = (int)at[a] + 1;
			if(at[a] < ((hi >> (10 * a)) & 1023u) && (!job.cull || ptl::reaches_cell(job.grid, p.x, p.y, p.z, radius, n[0], n[1], n[2]))) tag |= 1u << (4 + 2 * a);
		}
		uint32_t w[4];
		ptl::pack_entry(p.x, p.y, p.z, radius, tag, id, w);
		job.cell_entries[k] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
		all_listed &= tag >> 3;
	}
	if(k1 > k0 && all_listed != 0) job.cell_start[c] = k0 | (all_listed << ptl::CELL_FLAG_SHIFT);
}

}  // namespace

namespace ptl {

size_t grid_scan_blocks(size_t cells) { return (cells + kPerBlock - 1) / kPerBlock; }

// Counts and scans: afterwards cell_start[] is final and cell_start[cells] is the length of the lists.
hipError_t launch_grid_count_scan(const GridBuildJob& job, size_t cells, hipStream_t stream) {
	hipError_t e = hipMemsetAsync(job.cell_start, 0, (cells + 1) * sizeof(uint32_t), stream);
	if(e != hipSuccess) return e;
	if(job.n > 0) hipLaunchKernelGGL(pt_k_grid_count, dim3((unsigned)((job.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job);
	const size_t blocks = grid_scan_blocks(cells);
	hipLaunchKernelGGL(pt_k_scan_blocks, dim3((unsigned)blocks), dim3(kBlock), 0, stream, job.cell_start + 1, cells, job.block_sums);
	hipLaunchKernelGGL(pt_k_scan_totals, dim3(1), dim3(kTopThreads), 0, stream, job.block_sums, (int)blocks);
	hipLaunchKernelGGL(pt_k_scan_add, dim3((unsigned)blocks), dim3(kBlock), 0, stream, job.cell_start + 1, cells, job.block_sums);
	e = hipMemsetAsync(job.longest, 0, sizeof(uint32_t), stream);
	if(e != hipSuccess) return e;
	hipLaunchKernelGGL(pt_k_grid_longest, dim3((unsigned)((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job.cell_start, cells, job.longest);
	return hipGetLastError();
}

// Fills and orders the lists (cell_items and, if set, cell_entries must hold cell_start[cells] elements).
hipError_t launch_grid_fill(const GridBuildJob& job, size_t cells, hipStream_t stream) {
	hipError_t e = hipMemsetAsync(job.cursor, 0, cells * sizeof(uint32_t), stream);
	if(e != hipSuccess) return e;
	if(job.n > 0) hipLaunchKernelGGL(pt_k_grid_fill, dim3((unsigned)((job.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job);
	hipLaunchKernelGGL(pt_k_grid_order, dim3((unsigned)((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, job, cells);
	return hipGetLastError();
}

}  // namespace ptl
