// Does a kernel on a second stream run beside a grid that fills the chip, when the big grid's stream leaves one
// compute unit out of its mask?  Prints how long the small kernel waited in each arrangement.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

__global__ __launch_bounds__(256) void hog(unsigned long long ticks, int* sink) {
	__shared__ int lds[10 * 1024];               // 40 KB a block: four blocks fill a unit's LDS
	lds[threadIdx.x] = (int)threadIdx.x;
	__syncthreads();
	asm volatile("v_mov_b32 v127, 0" ::: "v127");      // 128 registers a lane: four wavefronts fill a SIMD's file
	const unsigned long long t0 = wall_clock64();
	while(wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
	if(lds[(threadIdx.x * 7) & 255] == -1) *sink = 1;
}
__global__ __launch_bounds__(256) void small(int* out) { out[threadIdx.x] = 1; }

int main() {
	int cus = 0;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	cus = prop.multiProcessorCount;
	int* d = nullptr;
	CK(hipMalloc(&d, 4096));
	for(int masked = 0; masked < 2; masked++) {
		hipStream_t big = nullptr, side = nullptr;
		if(masked) {
			uint32_t mask[32] = { 0 };
			for(int cu = 0; cu + 1 < cus; cu++) mask[cu / 32] |= 1u << (cu % 32);
			CK(hipExtStreamCreateWithCUMask(&big, (uint32_t)((cus + 31) / 32), mask));
		} else CK(hipStreamCreateWithFlags(&big, hipStreamNonBlocking));
		CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
		hipLaunchKernelGGL(small, dim3(1), dim3(256), 0, side, d);      // warm both
		hipLaunchKernelGGL(hog, dim3(8), dim3(256), 0, big, 1000ull, d);
		CK(hipDeviceSynchronize());
		auto t0 = std::chrono::steady_clock::now();
		hipLaunchKernelGGL(hog, dim3(cus * 4), dim3(256), 0, big, 5000000ull, d);      // 50 ms at 100 MHz
		std::this_thread::sleep_for(std::chrono::milliseconds(5));
		auto t1 = std::chrono::steady_clock::now();
		hipLaunchKernelGGL(small, dim3(8), dim3(256), 0, side, d);
		CK(hipStreamSynchronize(side));
		auto t2 = std::chrono::steady_clock::now();
		CK(hipDeviceSynchronize());
		auto t3 = std::chrono::steady_clock::now();
		printf("%s: small kernel done %.2f ms after its launch; big grid took %.2f ms\n", masked ? "one unit masked out" : "no mask",
		       std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t3 - t0).count());
		CK(hipStreamDestroy(big)); CK(hipStreamDestroy(side));
	}
	return 0;
}
