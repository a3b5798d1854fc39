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

// The same chains on two-element vectors: v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar instructions, so
// bit-exact code could use them - if they issue faster than two scalar instructions.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CHAINS>
__global__ __launch_bounds__(64) void spin_packed(float* out, int iters, float seed) {
	f2 v[CHAINS];
	for(int k = 0; k < CHAINS; k++) v[k] = (f2){ seed + (float)(threadIdx.x + k), seed - (float)(threadIdx.x + k) };
	const f2 m = { 1.0000001f, 0.9999999f }, a = { 0.5f, 0.25f };
	for(int i = 0; i < iters; i++) {
		#pragma unroll
		for(int k = 0; k < CHAINS; k++) v[k] = v[k] * m;
		#pragma unroll
		for(int k = 0; k < CHAINS; k++) v[k] = v[k] + a;
	}
	float s = 0;
	for(int k = 0; k < CHAINS; k++) s += v[k].x + v[k].y;
	if(s == 12345.678f) out[0] = s;
}

template <int CHAINS, bool PACKED = false>
static void run(int waves_per_simd, float* d) {
	hipDeviceProp_t prop;
	(void)hipGetDeviceProperties(&prop, 0);
	const int simds = prop.multiProcessorCount * 4;
	const int iters = 20000;
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	auto kernel = PACKED ? spin_packed<CHAINS> : spin<CHAINS>;
	for(int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(kernel, dim3(simds * waves_per_simd), dim3(64), 0, 0, d, iters, 1.0f);
	(void)hipEventRecord(a);
	for(int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(kernel, dim3(simds * waves_per_simd), dim3(64), 0, 0, d, iters, 1.0f);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	const double ops = 5.0 * (double)simds * waves_per_simd * 64.0 * iters * CHAINS * 2.0 * (PACKED ? 2.0 : 1.0);
	printf("%s chains %d  waves/SIMD %d  %.2f T lane-ops/s  (%.2f cycles per wave-instruction and SIMD at 2.4 GHz)\n", PACKED ? "packed" : "scalar", CHAINS, waves_per_simd,
	       ops / (ms * 1e-3) / 1e12, 2.4e9 * (ms * 1e-3) / (5.0 * waves_per_simd * iters * CHAINS * 2.0));
}

int main() {
	float* d;
	(void)hipMalloc(&d, 256);
	for(int w : { 1, 2, 4, 8 }) run<1>(w, d);
	for(int w : { 1, 2, 4, 8 }) run<8>(w, d);
	for(int w : { 1, 2, 4, 8 }) run<8, true>(w, d);
	return 0;
}
