// Checks the DPP inclusive prefix sum used by the cooperative cast's walk phase against a serial sum,
// on random and on extreme inputs.  Pure register work: no memory access depends on the scan.
// build: hipcc --offload-arch=gfx950 -O3 -I../../include -I../../pathtracer_amd/csrc wave_scan_test.hip -o wave_scan_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pt_wavescan.h"

__global__ void scan_kernel(const uint32_t* in, uint32_t* out) {
	const int i = blockIdx.x * 64 + threadIdx.x;
	out[i] = ptd::wave_inclusive_sum(in[i]);
}

int main() {
	const int waves = 4096, n = waves * 64;
	std::vector<uint32_t> h(n), want(n), got(n);
	srand(7);
	for(int w = 0; w < waves; w++) {
		uint32_t run = 0;
		for(int l = 0; l < 64; l++) {
			uint32_t v = w == 0 ? 1u : w == 1 ? (l == 63 ? 5u : 0u) : w == 2 ? (l == 0 ? 9u : 0u) : w == 3 ? 0x01000000u : (rand() % 5 == 0 ? 0u : (uint32_t)(rand() % 1000));
			h[w * 64 + l] = v;
			run += v;
			want[w * 64 + l] = run;
		}
	}
	uint32_t *din, *dout;
	if(hipMalloc(&din, n * 4) != hipSuccess || hipMalloc(&dout, n * 4) != hipSuccess) return 2;
	(void)hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(scan_kernel, dim3(waves), dim3(64), 0, 0, din, dout);
	if(hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
	int bad = 0;
	for(int i = 0; i < n; i++) if(got[i] != want[i]) { if(bad < 5) printf("mismatch wave %d lane %d: got %u want %u\n", i / 64, i % 64, got[i], want[i]); bad++; }
	printf("wave_inclusive_sum: %d of %d values wrong\n", bad, n);
	return bad ? 1 : 0;
}
