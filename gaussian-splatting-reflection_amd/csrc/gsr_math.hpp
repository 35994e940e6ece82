// Small fp32 vector/matrix helpers and the SH colour evaluation for the per-Gaussian kernels.
// Column-major 3x3 (m[col][row]) with left-to-right accumulation, i.e. the arithmetic order of the
// glm expressions in the reference kernels.  Everything in this header is compiled with FMA contraction
// OFF so that integer results derived from it (radii, tile rects, sort keys) are reproducible bit for
// bit against the CPU oracle, which is built the same way.
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

struct M3 {
	float m[3][3];
};

__device__ __forceinline__ M3 m3_make(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
	M3 r;
	r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
	r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
	r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
	return r;
}
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b) {
#pragma clang fp contract(off)
	M3 r;
#pragma unroll
	for (int j = 0; j < 3; j++)
#pragma unroll
		for (int i = 0; i < 3; i++) r.m[j][i] = a.m[0][i] * b.m[j][0] + a.m[1][i] * b.m[j][1] + a.m[2][i] * b.m[j][2];
	return r;
}
__device__ __forceinline__ M3 m3_T(const M3& a) {
	M3 r;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int n = 0; n < 3; n++) r.m[n][c] = a.m[c][n];
	return r;
}

// SH constants (DSR auxiliary.h:47-64 / DGR auxiliary.h:21-38)
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
#define GSR_SH_C2_0 1.0925484305920792f
#define GSR_SH_C2_1 -1.0925484305920792f
#define GSR_SH_C2_2 0.31539156525252005f
#define GSR_SH_C2_3 -1.0925484305920792f
#define GSR_SH_C2_4 0.5462742152960396f
#define GSR_SH_C3_0 -0.5900435899266435f
#define GSR_SH_C3_1 2.890611442640554f
#define GSR_SH_C3_2 -0.4570457994644658f
#define GSR_SH_C3_3 0.3731763325901154f
#define GSR_SH_C3_4 -0.4570457994644658f
#define GSR_SH_C3_5 1.445305721320277f
#define GSR_SH_C3_6 -0.5900435899266435f

struct F3 {
	float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator*(float f, F3 a) {
#pragma clang fp contract(off)
	return F3{f * a.x, f * a.y, f * a.z};
}
__device__ __forceinline__ F3 operator+(F3 a, F3 b) {
#pragma clang fp contract(off)
	return F3{a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ F3 operator-(F3 a, F3 b) {
#pragma clang fp contract(off)
	return F3{a.x - b.x, a.y - b.y, a.z - b.z};
}

// Loads the (deg+1)^2 SH coefficients of one Gaussian as float4s (rows are 12*M bytes, 16-byte aligned
// whenever M*3 is a multiple of 4, i.e. M = 4, 8, 12, 16, ...; other M fall back to scalar loads).
struct ShRow {
	float v[48];
};
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, int idx, int M, int ncoef, ShRow& s) {
	const float* row = shs + (size_t)idx * M * 3;
	const int nfl = ncoef * 3;
	if (((M * 3) & 3) == 0) {
		const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
		for (int q = 0; q < 12; q++) {
			if (q * 4 < nfl) {
				const float4 t = r4[q];
				s.v[4 * q + 0] = t.x; s.v[4 * q + 1] = t.y; s.v[4 * q + 2] = t.z; s.v[4 * q + 3] = t.w;
			}
		}
	} else {
#pragma unroll
		for (int k = 0; k < 48; k++)
			if (k < nfl) s.v[k] = row[k];
	}
}

// (Measured and dropped: staging the wave's 64 SH rows through LDS with lane-contiguous float4 loads — 12 KB per wave, rows 13
// float4 apart so that the row walk is bank-conflict-free — instead of this one-row-per-lane walk.  surfel_preprocess_kernel
// at C3: 0.105 ms before, 0.109 ms after; the strided walk is absorbed by L1/L2 and the kernel is not bound by it.)

// computeColorFromSH forward (DSR/DGR forward.cu:20-71): returns the unclamped colour + 0.5.
__device__ __forceinline__ F3 sh_eval(int deg, const ShRow& s, float x, float y, float z) {
#pragma clang fp contract(off)
	auto sh = [&](int k) { return F3{s.v[3 * k], s.v[3 * k + 1], s.v[3 * k + 2]}; };
	F3 result = GSR_SH_C0 * sh(0);
	if (deg > 0) {
		result = result - (GSR_SH_C1 * y) * sh(1) + (GSR_SH_C1 * z) * sh(2) - (GSR_SH_C1 * x) * sh(3);
		if (deg > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
			result = result + (GSR_SH_C2_0 * xy) * sh(4) + (GSR_SH_C2_1 * yz) * sh(5) + (GSR_SH_C2_2 * (2.0f * zz - xx - yy)) * sh(6) +
			         (GSR_SH_C2_3 * xz) * sh(7) + (GSR_SH_C2_4 * (xx - yy)) * sh(8);
			if (deg > 2) {
				result = result + (GSR_SH_C3_0 * y * (3.0f * xx - yy)) * sh(9) + (GSR_SH_C3_1 * xy * z) * sh(10) +
				         (GSR_SH_C3_2 * y * (4.0f * zz - xx - yy)) * sh(11) + (GSR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh(12) +
				         (GSR_SH_C3_4 * x * (4.0f * zz - xx - yy)) * sh(13) + (GSR_SH_C3_5 * z * (xx - yy)) * sh(14) +
				         (GSR_SH_C3_6 * x * (xx - 3.0f * yy)) * sh(15);
			}
		}
	}
	result.x += 0.5f;
	result.y += 0.5f;
	result.z += 0.5f;
	return result;
}

// dnormvdv (DSR auxiliary.h:132-142)
__device__ __forceinline__ F3 dnormvdv(F3 v, F3 dv) {
	const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	F3 r;
	r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return r;
}

// computeColorFromSH backward (DSR backward.cu:20-139 / DGR backward.cu:23-142).
// The dL_dsh row of a Gaussian is the outer product w (one weight per coefficient, zero above the active degree) x dL_dRGB
// (zeroed where the forward clamped the channel): sh_backward_weights leaves both and returns the view-direction gradient
// w.r.t. the mean; sh_backward also writes the row dL_dsh[idx, 0:M, :] completely (the reference relies on torch::zeros
// for the part above the active degree).
__device__ __forceinline__ F3 sh_backward_weights(int deg, const ShRow& s, F3 dir_orig, uint8_t clamped_bits, F3& dL_dRGB, float* w) {
	const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
	const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
	auto sh = [&](int k) { return F3{s.v[3 * k], s.v[3 * k + 1], s.v[3 * k + 2]}; };
	dL_dRGB.x *= (clamped_bits & 1) ? 0.f : 1.f;
	dL_dRGB.y *= (clamped_bits & 2) ? 0.f : 1.f;
	dL_dRGB.z *= (clamped_bits & 4) ? 0.f : 1.f;
#pragma unroll
	for (int k = 0; k < 16; k++) w[k] = 0.f;
	F3 dRGBdx{0, 0, 0}, dRGBdy{0, 0, 0}, dRGBdz{0, 0, 0};
	w[0] = GSR_SH_C0;
	if (deg > 0) {
		w[1] = -GSR_SH_C1 * y;
		w[2] = GSR_SH_C1 * z;
		w[3] = -GSR_SH_C1 * x;
		dRGBdx = (-GSR_SH_C1) * sh(3);
		dRGBdy = (-GSR_SH_C1) * sh(1);
		dRGBdz = GSR_SH_C1 * sh(2);
		if (deg > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
			w[4] = GSR_SH_C2_0 * xy;
			w[5] = GSR_SH_C2_1 * yz;
			w[6] = GSR_SH_C2_2 * (2.f * zz - xx - yy);
			w[7] = GSR_SH_C2_3 * xz;
			w[8] = GSR_SH_C2_4 * (xx - yy);
			dRGBdx = dRGBdx + ((GSR_SH_C2_0 * y) * sh(4) + (GSR_SH_C2_2 * 2.f * -x) * sh(6) + (GSR_SH_C2_3 * z) * sh(7) + (GSR_SH_C2_4 * 2.f * x) * sh(8));
			dRGBdy = dRGBdy + ((GSR_SH_C2_0 * x) * sh(4) + (GSR_SH_C2_1 * z) * sh(5) + (GSR_SH_C2_2 * 2.f * -y) * sh(6) + (GSR_SH_C2_4 * 2.f * -y) * sh(8));
			dRGBdz = dRGBdz + ((GSR_SH_C2_1 * y) * sh(5) + (GSR_SH_C2_2 * 2.f * 2.f * z) * sh(6) + (GSR_SH_C2_3 * x) * sh(7));
			if (deg > 2) {
				w[9] = GSR_SH_C3_0 * y * (3.f * xx - yy);
				w[10] = GSR_SH_C3_1 * xy * z;
				w[11] = GSR_SH_C3_2 * y * (4.f * zz - xx - yy);
				w[12] = GSR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
				w[13] = GSR_SH_C3_4 * x * (4.f * zz - xx - yy);
				w[14] = GSR_SH_C3_5 * z * (xx - yy);
				w[15] = GSR_SH_C3_6 * x * (xx - 3.f * yy);
				dRGBdx = dRGBdx + ((GSR_SH_C3_0 * 3.f * 2.f * xy) * sh(9) + (GSR_SH_C3_1 * yz) * sh(10) + (GSR_SH_C3_2 * -2.f * xy) * sh(11) +
				                   (GSR_SH_C3_3 * -3.f * 2.f * xz) * sh(12) + (GSR_SH_C3_4 * (-3.f * xx + 4.f * zz - yy)) * sh(13) +
				                   (GSR_SH_C3_5 * 2.f * xz) * sh(14) + (GSR_SH_C3_6 * 3.f * (xx - yy)) * sh(15));
				dRGBdy = dRGBdy + ((GSR_SH_C3_0 * 3.f * (xx - yy)) * sh(9) + (GSR_SH_C3_1 * xz) * sh(10) + (GSR_SH_C3_2 * (-3.f * yy + 4.f * zz - xx)) * sh(11) +
				                   (GSR_SH_C3_3 * -3.f * 2.f * yz) * sh(12) + (GSR_SH_C3_4 * -2.f * xy) * sh(13) + (GSR_SH_C3_5 * -2.f * yz) * sh(14) +
				                   (GSR_SH_C3_6 * -3.f * 2.f * xy) * sh(15));
				dRGBdz = dRGBdz + ((GSR_SH_C3_1 * xy) * sh(10) + (GSR_SH_C3_2 * 4.f * 2.f * yz) * sh(11) + (GSR_SH_C3_3 * 3.f * (2.f * zz - xx - yy)) * sh(12) +
				                   (GSR_SH_C3_4 * 4.f * 2.f * xz) * sh(13) + (GSR_SH_C3_5 * (xx - yy)) * sh(14));
			}
		}
	}
	const F3 dL_ddir = f3(dRGBdx.x * dL_dRGB.x + dRGBdx.y * dL_dRGB.y + dRGBdx.z * dL_dRGB.z,
	                      dRGBdy.x * dL_dRGB.x + dRGBdy.y * dL_dRGB.y + dRGBdy.z * dL_dRGB.z,
	                      dRGBdz.x * dL_dRGB.x + dRGBdz.y * dL_dRGB.y + dRGBdz.z * dL_dRGB.z);
	return dnormvdv(dir_orig, dL_ddir);
}
// float4 number q (floats 4q .. 4q+3) of the row w x g
__device__ __forceinline__ float4 sh_row_f4(int q, const float* w, F3 g) {
	float e[4];
#pragma unroll
	for (int c = 0; c < 4; c++) {
		const int f = 4 * q + c;  // float index -> coefficient f/3, channel f%3
		const int k = f / 3, ch = f % 3;
		e[c] = w[k] * (ch == 0 ? g.x : (ch == 1 ? g.y : g.z));
	}
	return make_float4(e[0], e[1], e[2], e[3]);
}
template <bool ACC = false>
__device__ __forceinline__ F3 sh_backward(int idx, int deg, int M, const ShRow& s, F3 dir_orig, uint8_t clamped_bits, F3 dL_dRGB,
                                          float* __restrict__ dL_dshs) {
	float w[16];
	const F3 dmean = sh_backward_weights(deg, s, dir_orig, clamped_bits, dL_dRGB, w);
	// dL_dsh row: M*3 floats, contiguous; written as float4 when aligned
	float* out = dL_dshs + (size_t)idx * M * 3;
	if (((M * 3) & 3) == 0 && M <= 16) {
		float4* o4 = reinterpret_cast<float4*>(out);
		const int nq = M * 3 / 4;
#pragma unroll
		for (int q = 0; q < 12; q++) {
			if (q < nq) {
				const float4 e = sh_row_f4(q, w, dL_dRGB);
				if (ACC) {
					const float4 old = o4[q];
					o4[q] = make_float4(old.x + e.x, old.y + e.y, old.z + e.z, old.w + e.w);
				} else {
					o4[q] = e;
				}
			}
		}
	} else {
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if (k < M) {
				out[3 * k + 0] = ACC ? out[3 * k + 0] + w[k] * dL_dRGB.x : w[k] * dL_dRGB.x;
				out[3 * k + 1] = ACC ? out[3 * k + 1] + w[k] * dL_dRGB.y : w[k] * dL_dRGB.y;
				out[3 * k + 2] = ACC ? out[3 * k + 2] + w[k] * dL_dRGB.z : w[k] * dL_dRGB.z;
			}
		}
		for (int k = 16; k < M && !ACC; k++) {
			out[3 * k + 0] = 0.f;
			out[3 * k + 1] = 0.f;
			out[3 * k + 2] = 0.f;
		}
	}
	return dmean;
}

}  // namespace gsr
