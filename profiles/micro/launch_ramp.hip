// How fast the chip starts wavefronts: kernels that do next to nothing, timed with HIP events for growing grids, with the
// register and LDS footprint of the trace kernel's variants.  The slope of time against workgroups is the dispatch cost.
// build: hipcc --offload-arch=gfx950 -O3 profiles/micro/launch_ramp.hip -o /tmp/launch_ramp    run: /tmp/launch_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int WAVES_PER_SIMD, int LDS_BYTES>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void touch(unsigned* out, int spin) {
	__shared__ unsigned lds[LDS_BYTES / 4];
	lds[threadIdx.x] = threadIdx.x;
	__syncthreads();
	unsigned v = lds[(threadIdx.x + 1) & 255];
	for(int i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;      // a body of known length, to tell start-up from throughput
	if(v == 0xdeadbeefu) out[0] = v;
}

// the same with 96 vector registers allocated, as the trace kernel has
template <int WAVES_PER_SIMD, int LDS_BYTES>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void touch96(unsigned* out, int spin) {
	__shared__ unsigned lds[LDS_BYTES / 4];
	lds[threadIdx.x] = threadIdx.x;
	__syncthreads();
	unsigned v = lds[(threadIdx.x + 1) & 255];
	asm volatile("v_mov_b32 v95, 0" ::: "v95");
	for(int i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
	if(v == 0xdeadbeefu) out[0] = v;
}

template <int W, int L, bool FAT = false>
static void run(const char* name, int spin) {
	unsigned* out;
	(void)hipMalloc(&out, 4);
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	printf("%-34s", name);
	for(int blocks : { 256, 512, 1024, 1280, 1536, 2560, 5120, 10240 }) {
		std::vector<float> ms;
		for(int rep = 0; rep < 30; rep++) {
			(void)hipEventRecord(a, 0);
			if(FAT) hipLaunchKernelGGL((touch96<W, L>), dim3(blocks), dim3(256), 0, 0, out, spin);
			else hipLaunchKernelGGL((touch<W, L>), dim3(blocks), dim3(256), 0, 0, out, spin);
			(void)hipEventRecord(b, 0);
			(void)hipEventSynchronize(b);
			float t; (void)hipEventElapsedTime(&t, a, b);
			ms.push_back(t);
		}
		std::sort(ms.begin(), ms.end());
		printf(" %6d: %6.1f us", blocks, ms[ms.size() / 2] * 1e3f);
	}
	printf("\n");
	(void)hipFree(out);
}

int main() {
	run<8, 1024>("8 waves/SIMD, 1 KB LDS, empty", 0);
	run<5, 8192>("5 waves/SIMD, 8 KB LDS, empty", 0);
	run<6, 21504>("6 waves/SIMD, 21 KB LDS, empty", 0);
	run<4, 40960>("4 waves/SIMD, 40 KB LDS, empty", 0);
	run<5, 8192, true>("5 waves/SIMD, 8 KB, 96 VGPRs, empty", 0);
	run<6, 21504, true>("6 waves/SIMD, 21 KB, 80+ VGPRs", 0);
	run<5, 8192>("5 waves/SIMD, 8 KB LDS, 2000 steps", 2000);
	run<5, 8192>("5 waves/SIMD, 8 KB LDS, 20000 steps", 20000);
	return 0;
}
