// Micro-benchmark (development aid, not a test): sustained wave64 VALU issue rate on one CU population
// as a function of resident waves per SIMD and instruction kind.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float a, float b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) {  // independent fma
			x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
			x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
		} else if (KIND == 1) {  // dependent chain
			x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
			x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
		} else if (KIND == 2) {  // fused DPP adds (8 independent)
			asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             "v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
		} else if (KIND == 3) {  // exp (8 independent)
			x0 = __expf(x0); x1 = __expf(x1); x2 = __expf(x2); x3 = __expf(x3);
			x4 = __expf(x4); x5 = __expf(x5); x6 = __expf(x6); x7 = __expf(x7);
		} else if (KIND == 4) {  // IEEE division (8 independent)
			x0 = a / x0; x1 = a / x1; x2 = a / x2; x3 = a / x3; x4 = a / x4; x5 = a / x5; x6 = a / x6; x7 = a / x7;
		} else if (KIND == 5) {  // fast rcp
			x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
			x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
		} else if (KIND == 6) {  // mul + add unfused (contract off)
#pragma clang fp contract(off)
			x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
			x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
		}
	}
	out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND> double run(int waves_per_simd, int iters, float* out) {
	const int blocks = 256 * 4 * waves_per_simd;  // one wave per block; 256 CUs x 4 SIMDs
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	k<KIND><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f);
	hipEventRecord(e0);
	k<KIND><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main() {
	float* out;
	hipMalloc(&out, 256 * 4 * 8 * 64 * 4 * 4);
	const int iters = 20000;
	const char* names[] = {"fma indep x8", "fma dependent", "v_add_f32_dpp x8", "__expf x8", "IEEE div x8", "v_rcp x8", "mul+add x8"};
	for (int w : {1, 2, 3, 4, 8}) {
		double t[7];
		t[0] = run<0>(w, iters, out); t[1] = run<1>(w, iters, out); t[2] = run<2>(w, iters, out); t[3] = run<3>(w, iters, out);
		t[4] = run<4>(w, iters, out); t[5] = run<5>(w, iters, out); t[6] = run<6>(w, iters, out);
		for (int kd = 0; kd < 7; kd++) {
			// ns per source-level op per wave; with w waves per SIMD the SIMD retires w ops in that time
			const double ns_per_op = t[kd] * 1e6 / (iters * 8.0);
			printf("waves/SIMD %d  %-18s %8.3f ms  %.2f ns/op/wave  -> %.2f ns per op per SIMD\n", w, names[kd], t[kd], ns_per_op, ns_per_op / w);
		}
	}
	return 0;
}
