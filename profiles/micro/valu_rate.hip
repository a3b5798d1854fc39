// Issue rate of plain fp32 vector adds and multiplies (no FMA contraction, no packing) against the
// number of wavefronts per SIMD: the ceiling of code that has to round every operation separately.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(64) void spin(float* out, int iters, float seed) {
	float v[CHAINS];
	for(int k = 0; k < CHAINS; k++) v[k] = seed + (float)(threadIdx.x + k);
	for(int i = 0; i < iters; i++) {
		#pragma unroll
		for(int k = 0; k < CHAINS; k++) v[k] = v[k] * 1.0000001f;
		#pragma unroll
		for(int k = 0; k < CHAINS; k++) v[k] = v[k] + 0.5f;
	}
	float s = 0;
	for(int k = 0; k < CHAINS; k++) s += v[k];
	if(s == 12345.678f) out[0] = s;
}

template <int CHAINS>
static void run(int waves_per_simd, float* d) {
	hipDeviceProp_t prop;
	(void)hipGetDeviceProperties(&prop, 0);
	const int simds = prop.multiProcessorCount * 4;
	const int iters = 20000;
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for(int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(spin<CHAINS>, dim3(simds * waves_per_simd), dim3(64), 0, 0, d, iters, 1.0f);
	(void)hipEventRecord(a);
	for(int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(spin<CHAINS>, dim3(simds * waves_per_simd), dim3(64), 0, 0, d, iters, 1.0f);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	const double ops = 5.0 * (double)simds * waves_per_simd * 64.0 * iters * CHAINS * 2.0;
	printf("chains %d  waves/SIMD %d  %.2f T lane-ops/s  (%.2f cycles per wave-instruction and SIMD at 2.4 GHz)\n", CHAINS, waves_per_simd,
	       ops / (ms * 1e-3) / 1e12, 2.4e9 * (ms * 1e-3) / (5.0 * waves_per_simd * iters * CHAINS * 2.0));
}

int main() {
	float* d;
	(void)hipMalloc(&d, 256);
	for(int w : { 1, 2, 4, 8 }) run<1>(w, d);
	for(int w : { 1, 2, 4, 8 }) run<8>(w, d);
	return 0;
}
