// What a wavefront's instruction stream costs when vector instructions are mixed with scalar ones, no-ops,
// exec-mask regions and LDS reads, against the number of wavefronts per SIMD.  The trace kernels issue one scalar
// instruction per three vector ones; this says whether those are free (another port) or take the wavefront's turn.
// build: hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define V8 \
	"v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n" \
	"v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
#define VS8 \
	"v_mul_f32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 s22, s22, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 s23, s23, 1\n" \
	"v_mul_f32 %4, %4, %8\n s_add_u32 s20, s20, 1\n v_mul_f32 %5, %5, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %6, %6, %8\n s_add_u32 s22, s22, 1\n v_mul_f32 %7, %7, %8\n s_add_u32 s23, s23, 1\n"
#define VN8 \
	"v_mul_f32 %0, %0, %8\n s_nop 0\n v_mul_f32 %1, %1, %8\n s_nop 0\n v_mul_f32 %2, %2, %8\n s_nop 0\n v_mul_f32 %3, %3, %8\n s_nop 0\n" \
	"v_mul_f32 %4, %4, %8\n s_nop 0\n v_mul_f32 %5, %5, %8\n s_nop 0\n v_mul_f32 %6, %6, %8\n s_nop 0\n v_mul_f32 %7, %7, %8\n s_nop 0\n"
// two exec-mask regions per eight vector instructions, as the compiler writes an `if` whose condition holds for every lane
#define VX8 \
	"v_cmp_lt_f32 vcc, %0, %9\n s_and_saveexec_b64 s[24:25], vcc\n s_cbranch_execz 1f\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n1:\n s_or_b64 exec, exec, s[24:25]\n" \
	"v_cmp_lt_f32 vcc, %3, %9\n s_and_saveexec_b64 s[24:25], vcc\n s_cbranch_execz 2f\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n2:\n s_or_b64 exec, exec, s[24:25]\n" \
	"v_mul_f32 %6, %6, %8\n"
// the same work with selects instead of regions
#define VC8 \
	"v_cmp_lt_f32 vcc, %0, %9\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_cndmask_b32 %2, %1, %2, vcc\n" \
	"v_cmp_lt_f32 vcc, %3, %9\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_cndmask_b32 %5, %4, %5, vcc\n" \
	"v_mul_f32 %6, %6, %8\n"
// a dependent chain of eight
#define VD8 \
	"v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n" \
	"v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n"
// two interleaved dependent chains
#define VD4x2 \
	"v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n" \
	"v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n"
// a transcendental and its dependent use (the reciprocal seed of a division)
#define VT8 \
	"v_rcp_f32 %0, %0\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n" \
	"v_rcp_f32 %4, %4\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
// one LDS read and its wait per eight vector instructions
#define VL8 \
	"ds_read_b32 %7, %10\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n" \
	"v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n s_waitcnt lgkmcnt(0)\n v_mul_f32 %7, %7, %8\n"

// integer multiplies: full 32-bit, 24-bit, and the 64-bit multiply-add PCG32's state update is made of
#define VI8 \
	"v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n" \
	"v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
#define VU8 \
	"v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n" \
	"v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
#define VQ8 \
	"v_mad_u64_u32 v[20:21], vcc, %0, %8, v[20:21]\n v_mad_u64_u32 v[22:23], vcc, %1, %8, v[22:23]\n v_mad_u64_u32 v[24:25], vcc, %2, %8, v[24:25]\n v_mad_u64_u32 v[26:27], vcc, %3, %8, v[26:27]\n" \
	"v_mad_u64_u32 v[20:21], vcc, %4, %8, v[20:21]\n v_mad_u64_u32 v[22:23], vcc, %5, %8, v[22:23]\n v_mad_u64_u32 v[24:25], vcc, %6, %8, v[24:25]\n v_mad_u64_u32 v[26:27], vcc, %7, %8, v[26:27]\n"
#define VF8 \
	"v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
	"v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define VS8Q \
	"v_sqrt_f32 %0, %0\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n" \
	"v_sqrt_f32 %4, %4\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"

#define KERNEL(NAME, BODY) \
__global__ __launch_bounds__(64) void NAME(float* out, int iters, float seed) { \
	__shared__ float lds[64]; \
	lds[threadIdx.x] = 1.0f; \
	float a = seed, b = seed + 1, c = seed + 2, d = seed + 3, e = seed + 4, f = seed + 5, g = seed + 6, h = seed + 7; \
	const float m = 1.0000001f, big = 1e30f; \
	const uint32_t at = threadIdx.x * 4; \
	for(int i = 0; i < iters; i++) { \
		asm volatile(BODY BODY BODY BODY : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(m), "v"(big), "v"(at) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory"); \
	} \
	float s = a + b + c + d + e + f + g + h; \
	if(s == 12345.678f) out[0] = s; \
}

KERNEL(k_v, V8)
KERNEL(k_vs, VS8)
KERNEL(k_vn, VN8)
KERNEL(k_vx, VX8)
KERNEL(k_vc, VC8)
KERNEL(k_vd, VD8)
KERNEL(k_vd2, VD4x2)
KERNEL(k_vt, VT8)
KERNEL(k_vl, VL8)
KERNEL(k_vi, VI8)
KERNEL(k_vu, VU8)
KERNEL(k_vq, VQ8)
KERNEL(k_vf, VF8)
KERNEL(k_vsq, VS8Q)

typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t kernel, int valu_per_body, int waves_per_simd, float* d) {
	hipDeviceProp_t prop;
	(void)hipGetDeviceProperties(&prop, 0);
	const int simds = prop.multiProcessorCount * 4;
	const int iters = 10000;
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for(int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(kernel, dim3(simds * waves_per_simd), dim3(64), 0, 0, d, iters, 1.0f);
	(void)hipEventRecord(a);
	for(int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(kernel, dim3(simds * waves_per_simd), dim3(64), 0, 0, d, iters, 1.0f);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	const double cycles_per_body = 2.4e9 * (ms * 1e-3) / (5.0 * iters * 4.0);          // one wavefront's time for one body
	printf("%-34s waves/SIMD %d  %7.1f cycles per body and wave  = %5.2f cycles per SIMD and vector instruction\n", name, waves_per_simd, cycles_per_body,
	       cycles_per_body / valu_per_body / waves_per_simd);
}

int main() {
	float* d;
	(void)hipMalloc(&d, 256);
	for(int w : { 1, 4, 8 }) {
		run("8 vector, independent", k_v, 8, w, d);
		run("8 vector + 8 scalar adds", k_vs, 8, w, d);
		run("8 vector + 8 s_nop 0", k_vn, 8, w, d);
		run("7 vector, 2 cmp, 2 exec regions", k_vx, 9, w, d);
		run("7 vector, 2 cmp, 2 selects", k_vc, 11, w, d);
		run("8 vector, one dependent chain", k_vd, 8, w, d);
		run("8 vector, two dependent chains", k_vd2, 8, w, d);
		run("6 vector + 2 v_rcp", k_vt, 8, w, d);
		run("8 vector + LDS read and wait", k_vl, 8, w, d);
		run("8 v_mul_lo_u32", k_vi, 8, w, d);
		run("8 v_mul_u32_u24", k_vu, 8, w, d);
		run("8 v_mad_u64_u32", k_vq, 8, w, d);
		run("8 v_fma_f32", k_vf, 8, w, d);
		run("6 vector + 2 v_sqrt", k_vsq, 8, w, d);
	}
	return 0;
}
