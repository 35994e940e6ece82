// Environment cubemap lookup (nearest / bilinear / seamless bilinear) forward + backward, and the fused
// deferred-reflection pixel pass.  Behaviour follows submodules/cubemapencoder/src/cubemapencoder.cu (CME,
// LEFT_TOP_AS_ORIGIN branch) and gaussian_renderer/__init__.py:22-35,148,178-179,197-199 +
// utils/general_utils.py:177-197 of the reference.  One thread per direction / pixel; the cubemap itself
// (<= 4.7 MB at L = 256) lives in L2 / Infinity Cache, the streaming traffic is the per-pixel planes.
#include <cstring>
#include "gsr_internal.hpp"
#include <mutex>
#include <memory>
#include <rocprim/device/device_radix_sort.hpp>
#include "gsr_sort.hpp"
#include <rocprim/iterator/counting_iterator.hpp>

namespace gsr {



// ----------------------------------------------------------------------------------------------
// Face / uv selection (CME cubemapencoder.cu:147-187)
__device__ __forceinline__ void cube_uv(float x, float y, float z, float& u, float& v, int& index) {
	int max_dim = 0;
	const float x_ = fabsf(x), y_ = fabsf(y), z_ = fabsf(z);
	float max_v = x_;
	if (y_ > max_v) { max_v = y_; max_dim = 1; }
	if (z_ > max_v) { max_v = z_; max_dim = 2; }
	if (max_dim == 0) {
		u = z / x; v = y / x;
		if (x >= 0.f) { index = 0; u = -u; v = -v; }
		else { index = 1; u = -u; }
	} else if (max_dim == 1) {
		u = x / y; v = z / y;
		if (y >= 0.f) { index = 2; }
		else { index = 3; u = -u; v = -v; }
	} else {
		u = x / z; v = y / z;
		if (z >= 0.f) { index = 4; v = -v; }
		else { index = 5; }
	}
}

// Neighbour-face texel across a cube edge (CME cubemapencoder.cu:66-106), as a small table:
// for (face, flag in {1,2,4,8}) -> new face and how (x', y') derive from (L-1, 0, x, y, L-1-x, L-1-y).
// Source selectors: 0 -> 0, 1 -> L-1, 2 -> x, 3 -> y, 4 -> L-1-x, 5 -> L-1-y.
__device__ __forceinline__ void edge_table(int L, int flag, int& face, int& x, int& y) {
	// packed as face | sx<<4 | sy<<8, index = face*4 + {flag 1:0, 2:1, 4:2, 8:3}
	const unsigned short tbl[24] = {
	    4 | (1 << 4) | (3 << 8), 5 | (0 << 4) | (3 << 8), 3 | (1 << 4) | (2 << 8), 2 | (1 << 4) | (2 << 8),   // face 0
	    5 | (1 << 4) | (3 << 8), 4 | (0 << 4) | (3 << 8), 3 | (0 << 4) | (4 << 8), 2 | (0 << 4) | (4 << 8),   // face 1
	    1 | (5 << 4) | (1 << 8), 0 | (3 << 4) | (1 << 8), 4 | (2 << 4) | (1 << 8), 5 | (4 << 4) | (1 << 8),   // face 2
	    1 | (5 << 4) | (0 << 8), 0 | (3 << 4) | (0 << 8), 4 | (2 << 4) | (0 << 8), 5 | (4 << 4) | (0 << 8),   // face 3
	    1 | (1 << 4) | (3 << 8), 0 | (0 << 4) | (3 << 8), 3 | (2 << 4) | (0 << 8), 2 | (2 << 4) | (0 << 8),   // face 4
	    0 | (1 << 4) | (3 << 8), 1 | (0 << 4) | (3 << 8), 3 | (4 << 4) | (1 << 8), 2 | (4 << 4) | (1 << 8)};  // face 5
	const int fi = flag == 1 ? 0 : (flag == 2 ? 1 : (flag == 4 ? 2 : 3));
	const unsigned e = tbl[face * 4 + fi];
	const int ix = x, iy = y;
	auto sel = [&](unsigned s) -> int {
		switch (s) {
			case 0: return 0;
			case 1: return L - 1;
			case 2: return ix;
			case 3: return iy;
			case 4: return L - 1 - ix;
			default: return L - 1 - iy;
		}
	};
	face = e & 15;
	x = sel((e >> 4) & 15);
	y = sel((e >> 8) & 15);
}

struct Seamless {
	int f[4], x[4], y[4];  // texel 0: v00, 1: v01 (u neighbour), 2: v10 (v neighbour), 3: v11
	float kx, ky;
	int flag;
	bool is_vertex;
};
// Compute_Seamless_Index (CME cubemapencoder.cu:189-263)
__device__ __forceinline__ void seamless_index(int index, int L, float u, float v, Seamless& s) {
	float lu = u, lv = -v;
	lu = (lu * 0.5f + 0.5f) * (float)L;
	lv = (lv * 0.5f + 0.5f) * (float)L;
	int ux_0 = (int)floorf(lu - 0.5f), uy_0 = (int)floorf(lv - 0.5f);
	int ux_1 = ux_0 + 1, uy_1 = uy_0 + 1;
	float kx = lu - (float)ux_0 - 0.5f;
	float ky = lv - (float)uy_0 - 0.5f;
	ux_0 = min(max(ux_0, 0), L - 1); ux_1 = min(max(ux_1, 0), L - 1);
	uy_0 = min(max(uy_0, 0), L - 1); uy_1 = min(max(uy_1, 0), L - 1);
	int flag = 0;
	if (lu < 0.5f) { flag |= 1; kx = 0.5f - lu; }
	else if (lu >= (float)L - 0.5f) flag |= 2;
	if (lv < 0.5f) { flag |= 4; ky = 0.5f - lv; }
	else if (lv >= (float)L - 0.5f) flag |= 8;
	s.is_vertex = false;
	for (int i = 0; i < 4; i++) { s.f[i] = index; s.x[i] = ux_0; s.y[i] = uy_0; }
	if ((flag & 3) && (flag & 12)) {
		s.is_vertex = true;
		edge_table(L, flag & 3, s.f[1], s.x[1], s.y[1]);
		edge_table(L, flag & 12, s.f[2], s.x[2], s.y[2]);
	} else if (flag & 3) {
		edge_table(L, flag, s.f[1], s.x[1], s.y[1]);
		s.y[2] = uy_1;
		s.y[3] = uy_1;
		edge_table(L, flag, s.f[3], s.x[3], s.y[3]);
	} else if (flag & 12) {
		s.x[1] = ux_1;
		edge_table(L, flag, s.f[2], s.x[2], s.y[2]);
		s.x[3] = ux_1;
		edge_table(L, flag, s.f[3], s.x[3], s.y[3]);
	} else {
		s.x[1] = ux_1;
		s.y[2] = uy_1;
		s.x[3] = ux_1; s.y[3] = uy_1;
	}
	s.kx = kx; s.ky = ky; s.flag = flag;
}

// Compute_Cubemap_UV_Backward (CME cubemapencoder.cu:265-292); (gu, gv) are modified as there.
__device__ __forceinline__ void cube_uv_backward(int index, float x, float y, float z, float gu, float gv, float& gx, float& gy, float& gz) {
	const int face = index / 2;
	if (face == 0) {
		if (index == 0) { gu = -gu; gv = -gv; }
		else { gu = -gu; }
		gx = -(z * gu + y * gv) / (x * x);
		gy = 1.f / x * gv;
		gz = 1.f / x * gu;
	} else if (face == 1) {
		if (index != 2) { gu = -gu; gv = -gv; }
		gx = 1.f / y * gu;
		gy = -(x * gu + z * gv) / (y * y);
		gz = 1.f / y * gv;
	} else {
		if (index == 4) { gv = -gv; }
		gx = 1.f / z * gu;
		gy = 1.f / z * gv;
		gz = -(x * gu + y * gv) / (z * z);
	}
}

__device__ __forceinline__ size_t texel(int f, int c, int y, int x, int C, int L) { return (((size_t)f * C + c) * L + y) * L + x; }

// Plain (non-seamless) footprint shared by the bilinear and nearest modes (CME cubemapencoder.cu:356-378, 409-422)
struct Plain {
	int f, ux0, ux1, uy0, uy1;
	float kx, ky;
};
__device__ __forceinline__ void plain_index(float vx, float vy, float vz, int L, Plain& p, bool nearest) {
	float u, v;
	cube_uv(vx, vy, vz, u, v, p.f);
	v = -v;
	u = (u * 0.5f + 0.5f) * (float)L;
	v = (v * 0.5f + 0.5f) * (float)L;
	if (nearest) {
		p.ux0 = min(max((int)u, 0), L - 1);
		p.uy0 = min(max((int)v, 0), L - 1);
		p.ux1 = p.ux0; p.uy1 = p.uy0; p.kx = 0; p.ky = 0;
		return;
	}
	const int ux_0 = (int)floorf(u - 0.5f), uy_0 = (int)floorf(v - 0.5f);
	p.kx = u - (float)ux_0 - 0.5f;
	p.ky = v - (float)uy_0 - 0.5f;
	p.ux0 = min(max(ux_0, 0), L - 1); p.ux1 = min(max(ux_0 + 1, 0), L - 1);
	p.uy0 = min(max(uy_0, 0), L - 1); p.uy1 = min(max(uy_0 + 1, 0), L - 1);
}

// ----------------------------------------------------------------------------------------------
// cubemap_encode_forward (CME cubemapencoder.cu:297-488): outputs [C,B]
__global__ void __launch_bounds__(256)
cubemap_fwd_kernel(const float* __restrict__ inputs, const float* __restrict__ cubemap, const float* __restrict__ fail_value,
                   float* __restrict__ outputs, int interp, int seamless, uint32_t B, int C, int L) {
	const uint32_t n = blockIdx.x * 256 + threadIdx.x;
	if (n >= B) return;
	const float vx = inputs[(size_t)n * 3], vy = inputs[(size_t)n * 3 + 1], vz = inputs[(size_t)n * 3 + 2];
	if (vx == 0.f && vy == 0.f && vz == 0.f) {
		for (int c = 0; c < C; c++) outputs[(size_t)c * B + n] = fail_value[c];
		return;
	}
	if (interp == 0 || seamless == 0) {
		Plain p;
		plain_index(vx, vy, vz, L, p, interp == 0);
		for (int c = 0; c < C; c++) {
			if (interp == 0) {
				outputs[(size_t)c * B + n] = cubemap[texel(p.f, c, p.uy0, p.ux0, C, L)];
			} else {
				const float v00 = cubemap[texel(p.f, c, p.uy0, p.ux0, C, L)], v01 = cubemap[texel(p.f, c, p.uy0, p.ux1, C, L)];
				const float v10 = cubemap[texel(p.f, c, p.uy1, p.ux0, C, L)], v11 = cubemap[texel(p.f, c, p.uy1, p.ux1, C, L)];
				outputs[(size_t)c * B + n] = (1 - p.ky) * ((1 - p.kx) * v00 + p.kx * v01) + p.ky * ((1 - p.kx) * v10 + p.kx * v11);
			}
		}
		return;
	}
	float u, v;
	int face;
	cube_uv(vx, vy, vz, u, v, face);
	Seamless s;
	seamless_index(face, L, u, v, s);
	for (int c = 0; c < C; c++) {
		const float v00 = cubemap[texel(s.f[0], c, s.y[0], s.x[0], C, L)];
		const float v01 = cubemap[texel(s.f[1], c, s.y[1], s.x[1], C, L)];
		const float v10 = cubemap[texel(s.f[2], c, s.y[2], s.x[2], C, L)];
		const float v11 = s.is_vertex ? (v00 + v01 + v10) / 3.f : cubemap[texel(s.f[3], c, s.y[3], s.x[3], C, L)];
		outputs[(size_t)c * B + n] = (1 - s.ky) * ((1 - s.kx) * v00 + s.kx * v01) + s.ky * ((1 - s.kx) * v10 + s.kx * v11);
	}
}

// Seamless-bilinear backward for one channel value; returns the (u, v) gradient contribution
// (CME cubemapencoder.cu:539-581).
__device__ __forceinline__ void seamless_bwd_channel(const Seamless& s, int c, int C, int L, const float* __restrict__ cubemap,
                                                     float* __restrict__ grad_cubemap, float g, float& gu, float& gv) {
	const size_t i00 = texel(s.f[0], c, s.y[0], s.x[0], C, L), i01 = texel(s.f[1], c, s.y[1], s.x[1], C, L);
	const size_t i10 = texel(s.f[2], c, s.y[2], s.x[2], C, L);
	const float v00 = cubemap[i00], v01 = cubemap[i01], v10 = cubemap[i10];
	float v11;
	if (s.is_vertex) {
		v11 = (v00 + v01 + v10) / 3.f;
		const float extra_g = s.ky * s.kx / 3.f;
		atomicAdd(grad_cubemap + i00, ((1 - s.ky) * (1 - s.kx) + extra_g) * g);
		atomicAdd(grad_cubemap + i01, ((1 - s.ky) * s.kx + extra_g) * g);
		atomicAdd(grad_cubemap + i10, ((s.ky * (1 - s.kx)) + extra_g) * g);
	} else {
		const size_t i11 = texel(s.f[3], c, s.y[3], s.x[3], C, L);
		v11 = cubemap[i11];
		atomicAdd(grad_cubemap + i00, (1 - s.ky) * (1 - s.kx) * g);
		atomicAdd(grad_cubemap + i01, (1 - s.ky) * s.kx * g);
		atomicAdd(grad_cubemap + i10, s.ky * (1 - s.kx) * g);
		atomicAdd(grad_cubemap + i11, s.ky * s.kx * g);
	}
	float lg0 = (1 - s.ky) * (v01 - v00) + s.ky * (v11 - v10);
	float lg1 = (1 - s.kx) * (v10 - v00) + s.kx * (v11 - v01);
	lg0 *= 0.5f * (float)L * g;
	lg1 *= 0.5f * (float)L * g;
	if (s.flag & 1) lg0 = -lg0;
	if (s.flag & 4) lg1 = -lg1;
	lg1 = -lg1;
	gu = lg0;
	gv = lg1;
}

// cubemap_encode_backward (CME cubemapencoder.cu:509-779)
__global__ void __launch_bounds__(256)
cubemap_bwd_kernel(const float* __restrict__ grad_outputs, const float* __restrict__ inputs, const float* __restrict__ cubemap,
                   float* __restrict__ grad_cubemap, float* __restrict__ grad_inputs, float* __restrict__ grad_fail, int interp, int seamless,
                   uint32_t B, int C, int L) {
	const uint32_t n = blockIdx.x * 256 + threadIdx.x;
	if (n >= B) return;
	const float vx = inputs[(size_t)n * 3], vy = inputs[(size_t)n * 3 + 1], vz = inputs[(size_t)n * 3 + 2];
	float gx = 0.f, gy = 0.f, gz = 0.f;
	if (vx == 0.f && vy == 0.f && vz == 0.f) {
		for (int c = 0; c < C; c++) atomicAdd(grad_fail + c, grad_outputs[(size_t)c * B + n]);
	} else if (interp == 0) {
		Plain p;
		plain_index(vx, vy, vz, L, p, true);
		for (int c = 0; c < C; c++) atomicAdd(grad_cubemap + texel(p.f, c, p.uy0, p.ux0, C, L), grad_outputs[(size_t)c * B + n]);
	} else if (seamless == 0) {
		Plain p;
		plain_index(vx, vy, vz, L, p, false);
		for (int c = 0; c < C; c++) {
			const size_t i00 = texel(p.f, c, p.uy0, p.ux0, C, L), i01 = texel(p.f, c, p.uy0, p.ux1, C, L);
			const size_t i10 = texel(p.f, c, p.uy1, p.ux0, C, L), i11 = texel(p.f, c, p.uy1, p.ux1, C, L);
			const float v00 = cubemap[i00], v01 = cubemap[i01], v10 = cubemap[i10], v11 = cubemap[i11];
			const float g = grad_outputs[(size_t)c * B + n];
			atomicAdd(grad_cubemap + i00, (1 - p.ky) * (1 - p.kx) * g);
			atomicAdd(grad_cubemap + i01, (1 - p.ky) * p.kx * g);
			atomicAdd(grad_cubemap + i10, p.ky * (1 - p.kx) * g);
			atomicAdd(grad_cubemap + i11, p.ky * p.kx * g);
			float lg0 = (1 - p.ky) * (v01 - v00) + p.ky * (v11 - v10);
			float lg1 = (1 - p.kx) * (v10 - v00) + p.kx * (v11 - v01);
			lg0 *= 0.5f * (float)L * g;
			lg1 *= 0.5f * (float)L * g;
			lg1 = -lg1;
			float a, b, cc;
			cube_uv_backward(p.f, vx, vy, vz, lg0, lg1, a, b, cc);
			gx += a; gy += b; gz += cc;
		}
	} else {
		float u, v;
		int face;
		cube_uv(vx, vy, vz, u, v, face);
		Seamless s;
		seamless_index(face, L, u, v, s);
		for (int c = 0; c < C; c++) {
			float gu, gv, a, b, cc;
			seamless_bwd_channel(s, c, C, L, cubemap, grad_cubemap, grad_outputs[(size_t)c * B + n], gu, gv);
			cube_uv_backward(face, vx, vy, vz, gu, gv, a, b, cc);
			gx += a; gy += b; gz += cc;
		}
	}
	grad_inputs[(size_t)n * 3] = gx;
	grad_inputs[(size_t)n * 3 + 1] = gy;
	grad_inputs[(size_t)n * 3 + 2] = gz;
}

// ----------------------------------------------------------------------------------------------
// Fused deferred reflection.  cam block (floats):
//   [0..8]   world_view_transform[:3,:3], row-major (wvt[j][i] at 3*j+i)
//   [9..17]  K^-1, row-major
//   [18..26] Rw = R.T of the camera's stored R (= world-to-camera rotation), row-major
//   [27..29] T (world-to-camera translation)        [30..32] rays_o = -Rw^T T (camera centre)
struct ReflPixel {
	float nwx, nwy, nwz, len;   // un-normalised world normal and its length
	float nx, ny, nz;           // normalised (/(len + 1e-6))
	float dx, dy, dz;           // unit view ray
	float dn;                   // d . n
	float rx, ry, rz;           // reflected ray
};
__device__ __forceinline__ void refl_pixel(const float* __restrict__ cam, float nvx, float nvy, float nvz, int px, int py, ReflPixel& o) {
	// gaussian_renderer/__init__.py:148 : n_world_j = sum_i n_view_i * wvt[j][i]
	o.nwx = nvx * cam[0] + nvy * cam[1] + nvz * cam[2];
	o.nwy = nvx * cam[3] + nvy * cam[4] + nvz * cam[5];
	o.nwz = nvx * cam[6] + nvy * cam[7] + nvz * cam[8];
	o.len = sqrtf(o.nwx * o.nwx + o.nwy * o.nwy + o.nwz * o.nwz);
	const float inv = 1.0f / (o.len + 1e-6f);   // :179
	o.nx = o.nwx * inv; o.ny = o.nwy * inv; o.nz = o.nwz * inv;
	// utils/general_utils.py:186-196
	const float x = (float)px, y = (float)py;
	const float pcx = cam[9] * x + cam[10] * y + cam[11] - cam[27];
	const float pcy = cam[12] * x + cam[13] * y + cam[14] - cam[28];
	const float pcz = cam[15] * x + cam[16] * y + cam[17] - cam[29];
	float wx = pcx * cam[18] + pcy * cam[21] + pcz * cam[24] - cam[30];
	float wy = pcx * cam[19] + pcy * cam[22] + pcz * cam[25] - cam[31];
	float wz = pcx * cam[20] + pcy * cam[23] + pcz * cam[26] - cam[32];
	const float dl = sqrtf(wx * wx + wy * wy + wz * wz);
	o.dx = wx / dl; o.dy = wy / dl; o.dz = wz / dl;
	o.dn = o.dx * o.nx + o.dy * o.ny + o.dz * o.nz;
	o.rx = o.dx - 2 * o.nx * o.dn;   // gaussian_renderer/__init__.py:22-24
	o.ry = o.dy - 2 * o.ny * o.dn;
	o.rz = o.dz - 2 * o.nz * o.dn;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// The pixel kernels gather the four bilinear corners of three channels: twelve 4-byte gathers per pixel from the
// reference's planar [6][3][L][L] layout, and the texture (1.2 MB at L = 128) does not live in a 32 KB L1 — 9 M L2 read
// requests per launch at 1080p (TCP_TCC_READ_REQ), which is what bounds a 60 us kernel.  With a texel-interleaved copy
// [6][L][L] of float4 (made by one 5-us kernel per forward, 1.5 MB) a corner is ONE 16-byte gather.
__global__ void __launch_bounds__(256) cubemap_interleave_kernel(const float* __restrict__ cubemap, float4* __restrict__ rgba, int L) {
	const size_t LL = (size_t)L * L;
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= 6 * LL) return;
	const size_t f = t / LL, r = t - f * LL;
	rgba[t] = make_float4(cubemap[(f * 3 + 0) * LL + r], cubemap[(f * 3 + 1) * LL + r], cubemap[(f * 3 + 2) * LL + r], 0.f);
}
// corner k of the seamless lookup, all three channels (the fourth corner of a cube vertex is the mean of the other three)
template <bool RGBA>
__device__ __forceinline__ void fetch_corners(const Seamless& s, int L, const float* __restrict__ cubemap, const float4* __restrict__ rgba, float (&v)[4][3]) {
	const int nk = s.is_vertex ? 3 : 4;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		if (k < nk) {
			if (RGBA) {
				const float4 t = rgba[((size_t)s.f[k] * L + s.y[k]) * L + s.x[k]];
				v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z;
			} else {
#pragma unroll
				for (int c = 0; c < 3; c++) v[k][c] = cubemap[texel(s.f[k], c, s.y[k], s.x[k], 3, L)];
			}
		}
	}
	if (s.is_vertex) {
#pragma unroll
		for (int c = 0; c < 3; c++) v[3][c] = (v[0][c] + v[1][c] + v[2][c]) / 3.f;
	}
}

template <bool RGBA>
__global__ void __launch_bounds__(256)
deferred_refl_fwd_kernel(const float* __restrict__ normal_view, const float* __restrict__ base, const float* __restrict__ strength,
                         const float* __restrict__ cam, const float* __restrict__ cubemap, const float4* __restrict__ rgba,
                         const float* __restrict__ fail_value, int L, int W, int H, float* __restrict__ out_final, float* __restrict__ out_refl,
                         float* __restrict__ out_nworld, uint32_t* __restrict__ sort_keys, uint32_t no_key) {
	const size_t HW = (size_t)W * H;
	const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (pix >= HW) return;
	uint32_t key = no_key;
	const int py = (int)(pix / W), px = (int)(pix - (size_t)py * W);
	// loads that do not depend on the reflected direction are issued together with the normal (one memory round trip less
	// per wave; the kernel is a chain of dependent round trips at five waves per SIMD)
	const float nvx = normal_view[pix], nvy = normal_view[HW + pix], nvz = normal_view[2 * HW + pix];
	const float sv = strength[pix];
	const float b0 = base[pix], b1 = base[HW + pix], b2 = base[2 * HW + pix];
	ReflPixel o;
	refl_pixel(cam, nvx, nvy, nvz, px, py, o);
	float c[3];
	if (o.rx == 0.f && o.ry == 0.f && o.rz == 0.f) {
		c[0] = fail_value[0]; c[1] = fail_value[1]; c[2] = fail_value[2];
	} else {
		float u, v;
		int face;
		cube_uv(o.rx, o.ry, o.rz, u, v, face);
		Seamless s;
		seamless_index(face, L, u, v, s);
		float cv[4][3];
		fetch_corners<RGBA>(s, L, cubemap, rgba, cv);
#pragma unroll
		for (int ch = 0; ch < 3; ch++)
			c[ch] = (1 - s.ky) * ((1 - s.kx) * cv[0][ch] + s.kx * cv[1][ch]) + s.ky * ((1 - s.kx) * cv[2][ch] + s.kx * cv[3][ch]);
		// sort key of the backward's footprint record (deferred_refl_bwd_entries_kernel): the texel of the upper-left corner when the 2x2
		// footprint lies inside one face.  It depends on forward data only, so the sort can run beside the backward's pixel kernel.
		if (s.flag == 0 && s.x[3] == s.x[0] + 1 && s.y[3] == s.y[0] + 1) key = (uint32_t)(((size_t)s.f[0] * L + s.y[0]) * L + s.x[0]);
	}
	if (sort_keys) sort_keys[pix] = key;
	const float bs[3] = {b0, b1, b2};
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		const float rc = sigmoidf_(c[ch]);
		out_refl[ch * HW + pix] = rc;
		out_final[ch * HW + pix] = (1 - sv) * bs[ch] + sv * rc;
	}
	out_nworld[pix] = o.nx;
	out_nworld[HW + pix] = o.ny;
	out_nworld[2 * HW + pix] = o.nz;
}

// Backward of the fused pass.  Four lanes per pixel: lane c < 3 owns colour channel c (lane 3 only helps with
// the shared index math).  The texel gradients go to a channel-INTERLEAVED scratch [6][L][L][4] so that the
// three channel atomics of one texel and its x-neighbour fall into one 64-byte line: float atomics
// execute at the memory side per 64-byte request, so this issues ~4x fewer requests than per-channel planes
// (measured 2.2 ms -> 0.8 ms), and pairing the x-neighbours in one wave instruction (below) removes another third.  `unpack_cubemap_grad_kernel` then adds the scratch into [6,3,L,L].
__device__ __forceinline__ float quad_sum(float v) {
	v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
	v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
	return v;
}
__global__ void __launch_bounds__(256)
deferred_refl_bwd_kernel(const float* __restrict__ normal_view, const float* __restrict__ base, const float* __restrict__ strength,
                         const float* __restrict__ cam, const float* __restrict__ cubemap, const float* __restrict__ fail_value, int L, int W,
                         int H, const float* __restrict__ g_final, const float* __restrict__ g_refl_color, const float* __restrict__ g_nworld,
                         float* __restrict__ g_normal_view, float* __restrict__ g_base, float* __restrict__ g_strength,
                         float* __restrict__ g_scratch, float* __restrict__ g_fail) {
	const size_t HW = (size_t)W * H;
	const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t pix = gid >> 2;
	const int ch = (int)(gid & 3);
	const bool live = pix < HW;           // whole quads are live or dead together
	const size_t p = live ? pix : 0;
	const int py = (int)(p / W), px = (int)(p - (size_t)py * W);
	ReflPixel o;
	refl_pixel(cam, normal_view[p], normal_view[HW + p], normal_view[2 * HW + p], px, py, o);
	const bool fail = (o.rx == 0.f && o.ry == 0.f && o.rz == 0.f);
	const bool chan = live && ch < 3;
	const int c = ch < 3 ? ch : 0;
	Seamless s;
	int face = 0;
	float cval;
	size_t i00 = 0, i01 = 0, i10 = 0, i11 = 0;
	float v00 = 0, v01 = 0, v10 = 0, v11 = 0;
	if (fail) {
		cval = fail_value[c];
		s.kx = 0; s.ky = 0; s.flag = 0; s.is_vertex = false;
	} else {
		float u, v;
		cube_uv(o.rx, o.ry, o.rz, u, v, face);
		seamless_index(face, L, u, v, s);
		i00 = texel(s.f[0], c, s.y[0], s.x[0], 3, L); i01 = texel(s.f[1], c, s.y[1], s.x[1], 3, L);
		i10 = texel(s.f[2], c, s.y[2], s.x[2], 3, L);
		v00 = cubemap[i00]; v01 = cubemap[i01]; v10 = cubemap[i10];
		if (s.is_vertex) v11 = (v00 + v01 + v10) / 3.f;
		else { i11 = texel(s.f[3], c, s.y[3], s.x[3], 3, L); v11 = cubemap[i11]; }
		cval = (1 - s.ky) * ((1 - s.kx) * v00 + s.kx * v01) + s.ky * ((1 - s.kx) * v10 + s.kx * v11);
	}
	const float sv = strength[p];
	const float rc = sigmoidf_(cval);
	float gs = 0.f, grx = 0.f, gry = 0.f, grz = 0.f;
	int tix[4] = {-1, -1, -1, -1};          // scratch index of this lane's channel at the four bilinear corners (-1: none)
	float twg[4] = {0.f, 0.f, 0.f, 0.f};    // and the gradient that goes there
	if (chan) {
		const float gf = g_final[c * HW + p];
		const float b = base[c * HW + p];
		g_base[c * HW + p] = (1 - sv) * gf;
		gs = gf * (rc - b);
		float gc = sv * gf;
		if (g_refl_color) gc += g_refl_color[c * HW + p];
		const float graw = gc * rc * (1 - rc);   // sigmoid'
		if (fail) {
			atomicAdd(g_fail + c, graw);
		} else {
			// interleaved scratch index of texel (f, y, x), channel c; the adds themselves are issued below, outside the
			// divergent region, paired with the neighbouring pixel's lanes
			auto sidx = [&](int k) -> int { return (int)((((((size_t)s.f[k] * L + s.y[k]) * L + s.x[k]) << 2)) + c); };
			if (s.is_vertex) {
				const float extra_g = s.ky * s.kx / 3.f;
				tix[0] = sidx(0); twg[0] = ((1 - s.ky) * (1 - s.kx) + extra_g) * graw;
				tix[1] = sidx(1); twg[1] = ((1 - s.ky) * s.kx + extra_g) * graw;
				tix[2] = sidx(2); twg[2] = ((s.ky * (1 - s.kx)) + extra_g) * graw;
			} else {
				tix[0] = sidx(0); twg[0] = (1 - s.ky) * (1 - s.kx) * graw;
				tix[1] = sidx(1); twg[1] = (1 - s.ky) * s.kx * graw;
				tix[2] = sidx(2); twg[2] = s.ky * (1 - s.kx) * graw;
				tix[3] = sidx(3); twg[3] = s.ky * s.kx * graw;
			}
			float lg0 = (1 - s.ky) * (v01 - v00) + s.ky * (v11 - v10);
			float lg1 = (1 - s.kx) * (v10 - v00) + s.kx * (v11 - v01);
			lg0 *= 0.5f * (float)L * graw;
			lg1 *= 0.5f * (float)L * graw;
			if (s.flag & 1) lg0 = -lg0;
			if (s.flag & 4) lg1 = -lg1;
			lg1 = -lg1;
			cube_uv_backward(face, o.rx, o.ry, o.rz, lg0, lg1, grx, gry, grz);
		}
	}
	// ---- texel adds.  Float atomics are priced per 64-byte memory-side request, and the two x-neighbours of a bilinear
	// footprint are 16 bytes apart in the interleaved scratch.  Quads are paired (pixels A, B = quads 2j, 2j+1): in each
	// of four rounds the eight lanes of a pair serve ONE pixel's row of the footprint — quad A's lanes the left texel,
	// quad B's lanes the right one — so both texels (2 x 3 channels) usually leave as a single request: ~2.5 requests
	// per pixel instead of 4.  The partner's index / value travel with ds_swizzle (xor 4, no LDS traffic).
	{
		const bool inA = ((threadIdx.x >> 2) & 1) == 0;
		const int XOR4 = 0x101F;   // bit-mask mode: and 0x1f, or 0, xor 4
		const int s_i0 = inA ? tix[1] : tix[0], s_i1 = inA ? tix[3] : tix[2];
		const float s_w0 = inA ? twg[1] : twg[0], s_w1 = inA ? twg[3] : twg[2];
		const int r_i0 = __builtin_amdgcn_ds_swizzle(s_i0, XOR4), r_i1 = __builtin_amdgcn_ds_swizzle(s_i1, XOR4);
		const float r_w0 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s_w0), XOR4));
		const float r_w1 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s_w1), XOR4));
		// round 0 / 1: pixel A, upper / lower row;  round 2 / 3: pixel B
		const int i0 = inA ? tix[0] : r_i0, i1 = inA ? tix[2] : r_i1, i2 = inA ? r_i0 : tix[1], i3 = inA ? r_i1 : tix[3];
		const float w0 = inA ? twg[0] : r_w0, w1 = inA ? twg[2] : r_w1, w2 = inA ? r_w0 : twg[1], w3 = inA ? r_w1 : twg[3];
		if (i0 >= 0) atomicAdd(g_scratch + i0, w0);
		if (i1 >= 0) atomicAdd(g_scratch + i1, w1);
		if (i2 >= 0) atomicAdd(g_scratch + i2, w2);
		if (i3 >= 0) atomicAdd(g_scratch + i3, w3);
	}
	// sum the per-channel pieces over the quad (lane 3 contributes zeros)
	gs = quad_sum(gs); grx = quad_sum(grx); gry = quad_sum(gry); grz = quad_sum(grz);
	if (live && ch == 3) g_strength[p] = gs;
	// r = d - 2 n (d.n)  ->  g_n = -2 [ (d.n) g_r + (g_r.n) d ]
	const float grn = grx * o.nx + gry * o.ny + grz * o.nz;
	float gnx = -2.f * (o.dn * grx + grn * o.dx);
	float gny = -2.f * (o.dn * gry + grn * o.dy);
	float gnz = -2.f * (o.dn * grz + grn * o.dz);
	if (g_nworld) { gnx += g_nworld[p]; gny += g_nworld[HW + p]; gnz += g_nworld[2 * HW + p]; }
	// n = nw / (|nw| + eps): g_nw = g_n / (len+eps) - nw (nw.g_n) / (len (len+eps)^2)   (0 subgradient at len = 0)
	const float inv = 1.0f / (o.len + 1e-6f);
	float gwx = gnx * inv, gwy = gny * inv, gwz = gnz * inv;
	if (o.len > 0.f) {
		const float k = (o.nwx * gnx + o.nwy * gny + o.nwz * gnz) * inv * inv / o.len;
		gwx -= o.nwx * k; gwy -= o.nwy * k; gwz -= o.nwz * k;
	}
	if (chan) g_normal_view[c * HW + p] = gwx * cam[c] + gwy * cam[3 + c] + gwz * cam[6 + c];
}

// The same backward for the sorted-footprint path (`binned` in the C ABI), one lane per pixel: nothing here needs the four lanes of the quad version (they
// exist to pair the texel atomics), so the index math runs once per pixel instead of four times.  A pixel whose bilinear
// footprint lies inside one cube face (all but the half-texel rim, ~2/L of the pixels) emits ONE record {g_r, g_g, g_b, kx,
// ky} plus the sort key "texel id of the upper-left corner": the other corners are t+1, t+L, t+L+1 and the four weights
// follow from (kx, ky).  Rim pixels (footprints that wrap onto a neighbouring face, cube vertices) add their corners to
// the staging buffer directly.
struct alignas(32) ReflFootprint {
	float g[3], kx, ky;   // 20 bytes used; padded so that a record never straddles a 32-byte sector when it is gathered
	float pad[3];
};
template <bool RGBA>
__global__ void __launch_bounds__(256)
deferred_refl_bwd_entries_kernel(const float* __restrict__ normal_view, const float* __restrict__ base, const float* __restrict__ strength,
                                 const float* __restrict__ cam, const float* __restrict__ cubemap, const float4* __restrict__ rgba,
                                 const float* __restrict__ fail_value, int L,
                                 int W, int H, const float* __restrict__ g_final, const float* __restrict__ g_refl_color,
                                 const float* __restrict__ g_nworld, float* __restrict__ g_normal_view, float* __restrict__ g_base,
                                 float* __restrict__ g_strength, float* __restrict__ g_fail, float* __restrict__ g_scratch,
                                 ReflFootprint* __restrict__ footprints, uint32_t* __restrict__ keys, const uint32_t* __restrict__ keys_fwd, uint32_t no_key,
                                 void* sort_clear, size_t sort_clear_bytes) {
	const size_t HW = (size_t)W * H;
	const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
	sort_clear_region(sort_clear, sort_clear_bytes, pix, (size_t)gridDim.x * 256u);   // look-back state of the sort that follows
	const bool live = pix < HW;
	const size_t p = live ? pix : 0;
	const int py = (int)(p / W), px = (int)(p - (size_t)py * W);
	// every load that does not depend on the reflected direction is issued here, with the normal
	const float nvx = normal_view[p], nvy = normal_view[HW + p], nvz = normal_view[2 * HW + p];
	const float sv = strength[p];
	float gfin[3], bas[3], grc[3] = {0.f, 0.f, 0.f}, gnw[3] = {0.f, 0.f, 0.f};
#pragma unroll
	for (int c = 0; c < 3; c++) {
		gfin[c] = g_final[c * HW + p];
		bas[c] = base[c * HW + p];
		if (g_refl_color) grc[c] = g_refl_color[c * HW + p];
		if (g_nworld) gnw[c] = g_nworld[c * HW + p];
	}
	ReflPixel o;
	refl_pixel(cam, nvx, nvy, nvz, px, py, o);
	const bool fail = (o.rx == 0.f && o.ry == 0.f && o.rz == 0.f);
	Seamless s;
	int face = 0;
	s.kx = 0; s.ky = 0; s.flag = 0; s.is_vertex = false;
	if (!fail) {
		float u, v;
		cube_uv(o.rx, o.ry, o.rz, u, v, face);
		seamless_index(face, L, u, v, s);
	}
	float graw[3] = {0.f, 0.f, 0.f};
	float gs = 0.f, grx = 0.f, gry = 0.f, grz = 0.f;
	float cv[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
	if (!fail) fetch_corners<RGBA>(s, L, cubemap, rgba, cv);
#pragma unroll
	for (int c = 0; c < 3; c++) {
		float cval;
		const float v00 = cv[0][c], v01 = cv[1][c], v10 = cv[2][c], v11 = cv[3][c];
		if (fail) cval = fail_value[c];
		else cval = (1 - s.ky) * ((1 - s.kx) * v00 + s.kx * v01) + s.ky * ((1 - s.kx) * v10 + s.kx * v11);
		const float rc = sigmoidf_(cval);
		const float gf = gfin[c];
		const float b = bas[c];
		if (live) g_base[c * HW + p] = (1 - sv) * gf;
		gs += gf * (rc - b);
		float gc = sv * gf;
		if (g_refl_color) gc += grc[c];
		graw[c] = gc * rc * (1 - rc);   // sigmoid'
		if (fail) {
			if (live) atomicAdd(g_fail + c, graw[c]);
		} else {
			float lg0 = (1 - s.ky) * (v01 - v00) + s.ky * (v11 - v10);
			float lg1 = (1 - s.kx) * (v10 - v00) + s.kx * (v11 - v01);
			lg0 *= 0.5f * (float)L * graw[c];
			lg1 *= 0.5f * (float)L * graw[c];
			if (s.flag & 1) lg0 = -lg0;
			if (s.flag & 4) lg1 = -lg1;
			lg1 = -lg1;
			float a, bb, cc;
			cube_uv_backward(face, o.rx, o.ry, o.rz, lg0, lg1, a, bb, cc);
			grx += a; gry += bb; grz += cc;
		}
	}
	// flag == 0 already implies the unclamped 2x2 block; the corner test keeps the record format honest regardless
	bool interior = !fail && s.flag == 0 && s.x[3] == s.x[0] + 1 && s.y[3] == s.y[0] + 1;
	// keys_fwd: the forward kernel already wrote the sort keys (same arithmetic on the same inputs) and the sort may be running beside this
	// kernel.  Its key decides whether the pixel has a record; should the two kernels ever disagree, the pixel goes through the rim
	// path below and its record, if the sort expects one, is zeros — nothing is lost or read uninitialised.
	const uint32_t kf = keys_fwd ? keys_fwd[p] : no_key;
	if (keys_fwd) interior = interior && kf == (uint32_t)(((size_t)s.f[0] * L + s.y[0]) * L + s.x[0]);
	{
		// rim pixels (~2/L of all, about one per wave): the wave serves them one at a time.  The pixel's corner texels and
		// weights travel through SGPRs and lanes 0..11 each add one (corner, channel) value, so a rim pixel costs one atomic
		// instruction whose twelve dwords fall into four 16-byte slots — four memory-side requests instead of twelve
		// single-lane ones (float atomics are priced per request: 44 us -> 15 us per launch at C3).
		const int lane = threadIdx.x & 63;
		const int k_of_lane = lane / 3, c_of_lane = lane - 3 * k_of_lane;
		// bilinear weights of the four corners (a cube vertex has three, the fourth is their mean)
		const float extra_g = s.is_vertex ? s.ky * s.kx / 3.f : 0.f;
		const float w4[4] = {(1 - s.ky) * (1 - s.kx) + extra_g, (1 - s.ky) * s.kx + extra_g, s.ky * (1 - s.kx) + extra_g, s.ky * s.kx};
		unsigned long long todo = __ballot(live && !fail && !interior);
		while (todo) {
			const int src = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			uint32_t tk[4];
			float wk[4], gk[3];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int f = __builtin_amdgcn_readlane(s.f[k], src), y = __builtin_amdgcn_readlane(s.y[k], src), x = __builtin_amdgcn_readlane(s.x[k], src);
				tk[k] = (uint32_t)(((size_t)f * L + y) * L + x);
				wk[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w4[k]), src));
			}
#pragma unroll
			for (int c = 0; c < 3; c++) gk[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(graw[c]), src));
			const int corners = __builtin_amdgcn_readlane((int)s.is_vertex, src) ? 3 : 4;
			if (k_of_lane < corners) {
				const uint32_t t = k_of_lane == 0 ? tk[0] : k_of_lane == 1 ? tk[1] : k_of_lane == 2 ? tk[2] : tk[3];
				const float w = k_of_lane == 0 ? wk[0] : k_of_lane == 1 ? wk[1] : k_of_lane == 2 ? wk[2] : wk[3];
				const float g = c_of_lane == 0 ? gk[0] : c_of_lane == 1 ? gk[1] : gk[2];
				atomicAdd(g_scratch + ((size_t)t << 2) + c_of_lane, w * g);
			}
		}
	}
	if (!live) return;
	{
		const uint32_t t00 = (uint32_t)(((size_t)s.f[0] * L + s.y[0]) * L + s.x[0]);
		if (interior) {
			float* f = reinterpret_cast<float*>(footprints + pix);
			*reinterpret_cast<float4*>(f) = make_float4(graw[0], graw[1], graw[2], s.kx);
			f[4] = s.ky;
		} else if (kf != no_key) {
			float* f = reinterpret_cast<float*>(footprints + pix);
			*reinterpret_cast<float4*>(f) = make_float4(0.f, 0.f, 0.f, 0.f);
			f[4] = 0.f;
		}
		if (!keys_fwd) keys[pix] = interior ? t00 : no_key;
	}
	g_strength[p] = gs;
	// r = d - 2 n (d.n)  ->  g_n = -2 [ (d.n) g_r + (g_r.n) d ]
	const float grn = grx * o.nx + gry * o.ny + grz * o.nz;
	float gnx = -2.f * (o.dn * grx + grn * o.dx);
	float gny = -2.f * (o.dn * gry + grn * o.dy);
	float gnz = -2.f * (o.dn * grz + grn * o.dz);
	if (g_nworld) { gnx += gnw[0]; gny += gnw[1]; gnz += gnw[2]; }
	const float inv = 1.0f / (o.len + 1e-6f);
	float gwx = gnx * inv, gwy = gny * inv, gwz = gnz * inv;
	if (o.len > 0.f) {
		const float k = (o.nwx * gnx + o.nwy * gny + o.nwz * gnz) * inv * inv / o.len;
		gwx -= o.nwx * k; gwy -= o.nwy * k; gwz -= o.nwz * k;
	}
#pragma unroll
	for (int c = 0; c < 3; c++) g_normal_view[c * HW + p] = gwx * cam[c] + gwy * cam[3 + c] + gwz * cam[6 + c];
}

// scratch [6][L][L][4] (channel-interleaved) -> grad_cubemap [6][3][L][L] (written, not accumulated)
template <bool ACC>
__global__ void __launch_bounds__(256) unpack_cubemap_grad_kernel(const float4* __restrict__ scratch, float* __restrict__ g_cubemap,
                                                                  float* __restrict__ g_fail, int L) {
	const size_t n = (size_t)6 * L * L;
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (t == 0) {   // the fail-value gradient was accumulated in the four floats behind the texel staging
		const float4 gf = scratch[n];
		g_fail[0] = (ACC ? g_fail[0] : 0.f) + gf.x; g_fail[1] = (ACC ? g_fail[1] : 0.f) + gf.y; g_fail[2] = (ACC ? g_fail[2] : 0.f) + gf.z;
	}
	if (t >= n) return;
	const size_t LL = (size_t)L * L;
	const size_t f = t / LL, r = t - f * LL;
	const float4 g = scratch[t];
	float* o = g_cubemap + f * 3 * LL + r;
	o[0] = (ACC ? o[0] : 0.f) + g.x;
	o[LL] = (ACC ? o[LL] : 0.f) + g.y;
	o[2 * LL] = (ACC ? o[2 * LL] : 0.f) + g.z;
}

// ---- accumulation of the footprints (see gsr_deferred_reflection_backward).  (texel id, pixel) pairs arrive sorted by
// texel id, so equal texels are adjacent and a workgroup's REFL_CHUNK * 256 records cover a narrow range of texels:
//   * every thread gathers the records of REFL_CHUNK consecutive pairs, sums the twelve (corner, channel) contributions
//     of a run of equal texels in registers and, when the texel changes, adds them to an LDS window of REFL_WIN texels
//     that starts at the workgroup's first texel (global atomics only for the rare texel beyond the window);
//   * the touched part of the window then goes to the staging buffer with contiguous atomics.
// LDS float atomics run at ~4 clocks per lane and memory-side ones at ~30 G/s, so the counts are what matters: per pixel
// ~12 / mean-run-length LDS adds (was 12) and per launch ~3 x distinct texels global adds (was 12 x pixels).
// REFL_WIN: at 1080p a workgroup's 2048 sorted pairs span ~100 texels, so the window is mostly slack — and its size decides
// whether this kernel, which runs on the side stream beside the tile backward, finds LDS on a CU that the tile backward
// has filled to 154 of 160 KB: with 4096 texels (49 KB) it waited for the tile backward to drain and ran beside the
// per-Gaussian backward instead (0.143 -> 0.163 ms); with 1024 (12 KB) it hides inside the tile backward (step 1.99 -> 1.97 ms).
#define REFL_CHUNK 8
#ifndef REFL_WIN
#define REFL_WIN 1024u
#endif
__global__ void __launch_bounds__(256) refl_run_combine_kernel(const uint32_t* __restrict__ keys_sorted, const uint32_t* __restrict__ pix_sorted,
                                                               const ReflFootprint* __restrict__ footprints, size_t n, uint32_t L, uint32_t no_key,
                                                               float* __restrict__ g_scratch) {
	__shared__ float win[3][REFL_WIN];
	__shared__ uint32_t s_hi;
	const size_t wg0 = (size_t)blockIdx.x * 256 * REFL_CHUNK;   // < n by the grid size
	const uint32_t t_lo = keys_sorted[wg0];
	if (t_lo == no_key) return;   // the "nothing to add" pairs sort to the end: nothing left for this workgroup
	// only the part of the window this workgroup can reach is cleared: its last texel plus the footprint (L + 1 further)
	const size_t wg_last = min(n, wg0 + (size_t)256 * REFL_CHUNK) - 1;
	const uint32_t t_hi = keys_sorted[wg_last];
	const uint32_t reach = t_hi == no_key ? REFL_WIN : min(REFL_WIN, t_hi - t_lo + L + 2u);
	for (uint32_t i = threadIdx.x; i < reach; i += 256) { win[0][i] = 0.f; win[1][i] = 0.f; win[2][i] = 0.f; }
	if (threadIdx.x == 0) s_hi = 0u;
	__syncthreads();

	const size_t i0 = wg0 + (size_t)threadIdx.x * REFL_CHUNK;
	uint32_t key[REFL_CHUNK];
	float4 ga[REFL_CHUNK];
	float gky[REFL_CHUNK];
#pragma unroll
	for (int j = 0; j < REFL_CHUNK; j++) key[j] = i0 + j < n ? keys_sorted[i0 + j] : no_key;
#pragma unroll
	for (int j = 0; j < REFL_CHUNK; j++) {
		ga[j] = make_float4(0.f, 0.f, 0.f, 0.f);
		gky[j] = 0.f;
		if (key[j] != no_key) {
			const float* f = reinterpret_cast<const float*>(footprints + pix_sorted[i0 + j]);
			ga[j] = *reinterpret_cast<const float4*>(f);
			gky[j] = f[4];
		}
	}
	float acc[4][3];
	uint32_t cur = no_key, hi = 0u;
	auto flush = [&]() {
		if (cur == no_key) return;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint32_t rel = cur - t_lo + (uint32_t)(k & 1) + (uint32_t)(k >> 1) * L;
			if (rel < REFL_WIN) {
				hi = max(hi, rel);
#pragma unroll
				for (int c = 0; c < 3; c++)
					if (acc[k][c] != 0.f) atomicAdd(&win[c][rel], acc[k][c]);
			} else {
				float* dst = g_scratch + (((size_t)t_lo + rel) << 2);
#pragma unroll
				for (int c = 0; c < 3; c++)
					if (acc[k][c] != 0.f) atomicAdd(dst + c, acc[k][c]);
			}
		}
	};
#pragma unroll
	for (int j = 0; j < REFL_CHUNK; j++) {
		if (key[j] != no_key) {
			if (key[j] != cur) {
				flush();
				cur = key[j];
#pragma unroll
				for (int k = 0; k < 4; k++) { acc[k][0] = 0.f; acc[k][1] = 0.f; acc[k][2] = 0.f; }
			}
			const float kx = ga[j].w, ky = gky[j];
			const float w4[4] = {(1 - ky) * (1 - kx), (1 - ky) * kx, ky * (1 - kx), ky * kx};
#pragma unroll
			for (int k = 0; k < 4; k++) {
				acc[k][0] += w4[k] * ga[j].x; acc[k][1] += w4[k] * ga[j].y; acc[k][2] += w4[k] * ga[j].z;
			}
		}
	}
	flush();
	if (hi != 0u) atomicMax(&s_hi, hi);
	__syncthreads();
	const uint32_t top = s_hi;
	for (uint32_t i = threadIdx.x; i <= top; i += 256) {
		float* dst = g_scratch + (((size_t)t_lo + i) << 2);
#pragma unroll
		for (int c = 0; c < 3; c++) {
			const float v = win[c][i];
			if (v != 0.f) atomicAdd(dst + c, v);
		}
	}
}

}  // namespace gsr

using namespace gsr;

extern "C" int gsr_cubemap_forward(const float* inputs, const float* cubemap, const float* fail_value, float* outputs, uint32_t interp,
                                   uint32_t seamless, uint32_t B, uint32_t C, uint32_t L, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (B == 0) return 0;
	if (!inputs || !cubemap || !fail_value || !outputs || C == 0 || L == 0) { set_error("gsr_cubemap_forward: invalid argument"); return GSR_E_INVALID; }
{ StageTimer st_(GSR_STAGE_CUBEMAP_FWD, stream); 	cubemap_fwd_kernel<<<(B + 255) / 256, 256, 0, stream>>>(inputs, cubemap, fail_value, outputs, (int)interp, (int)seamless, B, (int)C, (int)L); }
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_cubemap_backward(const float* grad_outputs, const float* inputs, const float* cubemap, float* grad_cubemap,
                                    float* grad_inputs, float* grad_fail, uint32_t interp, uint32_t seamless, uint32_t B, uint32_t C, uint32_t L,
                                    void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (B == 0) return 0;
	if (!grad_outputs || !inputs || !cubemap || !grad_cubemap || !grad_inputs || !grad_fail || C == 0 || L == 0) {
		set_error("gsr_cubemap_backward: invalid argument");
		return GSR_E_INVALID;
	}
{ StageTimer st_(GSR_STAGE_CUBEMAP_BWD, stream); 	cubemap_bwd_kernel<<<(B + 255) / 256, 256, 0, stream>>>(grad_outputs, inputs, cubemap, grad_cubemap, grad_inputs, grad_fail, (int)interp,
	                                                        (int)seamless, B, (int)C, (int)L); }
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_deferred_reflection_forward_ex(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                  const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                  float* out_final, float* out_refl_color, float* out_normal_world, float* cubemap_rgba,
                                                  uint32_t* sort_keys, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (width <= 0 || height <= 0 || !normal_view || !base_color || !refl_strength || !cam || !cubemap || !fail_value || !out_final ||
	    !out_refl_color || !out_normal_world || L == 0 || ((uintptr_t)cubemap_rgba & 15) != 0) {
		set_error("gsr_deferred_reflection_forward: invalid argument");
		return GSR_E_INVALID;
	}
	const size_t HW = (size_t)width * height;
	StageTimer st_(GSR_STAGE_REFL_FWD, stream);
	const unsigned grid = (unsigned)((HW + 255) / 256);
	if (cubemap_rgba) {
		float4* rgba = reinterpret_cast<float4*>(cubemap_rgba);
		cubemap_interleave_kernel<<<(unsigned)((6 * (size_t)L * L + 255) / 256), 256, 0, stream>>>(cubemap, rgba, (int)L);
		deferred_refl_fwd_kernel<true><<<grid, 256, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap, rgba, fail_value, (int)L, width, height,
		                                                        out_final, out_refl_color, out_normal_world, sort_keys, (uint32_t)(6 * (size_t)L * L));
	} else {
		deferred_refl_fwd_kernel<false><<<grid, 256, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap, nullptr, fail_value, (int)L, width, height,
		                                                         out_final, out_refl_color, out_normal_world, sort_keys, (uint32_t)(6 * (size_t)L * L));
	}
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}
extern "C" int gsr_deferred_reflection_forward(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                               const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                               float* out_final, float* out_refl_color, float* out_normal_world, void* stream_) {
	return gsr_deferred_reflection_forward_ex(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, out_final, out_refl_color,
	                                          out_normal_world, nullptr, nullptr, stream_);
}

// Scratch layout of the sorted-footprint backward (floats): [texel staging ntex*4][fail-value gradient 4][pad 4][footprints 8n][keys_in n][keys_out n][pixels_out n]
// [sort temp], n = H * W.
struct ReflScratch {
	size_t ntex, n, sort_bytes, total_floats;
	int key_bits;
};
// Sort of (texel id, pixel) through gsr_sort.hpp (one clear per sort).  17-bit texel ids at L = 128, 19-bit at L = 256: two
// passes with 9- or 10-bit digits instead of three with 8.  Workgroup shape measured at n = 2 M pairs (whole backward, ms):
// 256x12 0.361, 512x12 0.320, 1024x4 0.313, 1024x6 0.306, 1024x8 0.297, 1024x12 0.309, 1024x16 0.314.
// `small` (the tail on the side stream, beside the tile backward): 256-thread workgroups with 8-bit digits.  The tile backward keeps
// every CU at 16 single-wave workgroups and 152 of its 160 KB of LDS; a 1024-thread pass (39 KB of LDS, 16 waves) can only start on a
// CU that has drained, i.e. when the tile backward is over, while a 256-thread one (one wave per SIMD, ~13 KB) slips in whenever one
// of those workgroups retires.  Slower on an empty chip (one more pass, 4x the look-back chain), but hidden.
#ifndef REFL_SMALL_SORT
#define REFL_SMALL_SORT 1
#endif
static hipError_t refl_sort(void* temp, size_t& bytes, int key_bits, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* pix_out, size_t n, hipStream_t stream,
                            bool pre_cleared = false, bool small = false) {
	rocprim::counting_iterator<uint32_t> pix_in(0);
	if (temp == nullptr) {   // size query: the largest of the drivers' needs (neither the runtime switch nor the stream a tail runs on changes a scratch size)
		size_t own = 0, pub = 0, sm = 0;
		if (key_bits > 16 && key_bits <= 18) (void)onesweep_sort_pairs<1024, 8, 9>(nullptr, own, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		else if (key_bits > 18 && key_bits <= 20) (void)onesweep_sort_pairs<1024, 8, 10>(nullptr, own, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		else (void)onesweep_sort_pairs<1024, 8, 8>(nullptr, own, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		(void)onesweep_sort_pairs<256, 8, 8>(nullptr, sm, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		(void)rocprim::radix_sort_pairs(nullptr, pub, (const uint32_t*)keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, false);
		bytes = own > pub ? own : pub;
		if (sm > bytes) bytes = sm;
		return hipSuccess;
	}
	if (!option_sort_driver())   // gsr_set_option("sort_driver", 0) or an unknown rocPRIM release: the public entry point
		return rocprim::radix_sort_pairs(temp, bytes, (const uint32_t*)keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, false);
	if (small)
		return onesweep_sort_pairs<256, 8, 8>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared);
	if (key_bits > 16 && key_bits <= 18)
		return onesweep_sort_pairs<1024, 8, 9>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared);
	if (key_bits > 18 && key_bits <= 20)
		return onesweep_sort_pairs<1024, 8, 10>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared);
	return onesweep_sort_pairs<1024, 8, 8>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared);
}
static size_t refl_sort_cleared_bytes(int key_bits, size_t n, bool small) {
	if (small) return onesweep_cleared_bytes<256, 8, 8>(n, 0u, (unsigned)key_bits);
	if (key_bits > 16 && key_bits <= 18) return onesweep_cleared_bytes<1024, 8, 9>(n, 0u, (unsigned)key_bits);
	if (key_bits > 18 && key_bits <= 20) return onesweep_cleared_bytes<1024, 8, 10>(n, 0u, (unsigned)key_bits);
	return onesweep_cleared_bytes<1024, 8, 8>(n, 0u, (unsigned)key_bits);
}
static ReflScratch refl_scratch(uint32_t L, int width, int height) {
	ReflScratch r;
	r.ntex = (size_t)6 * L * L;
	r.n = (size_t)width * height;
	r.key_bits = 1;
	while (((size_t)1 << r.key_bits) <= r.ntex) r.key_bits++;   // keys take values 0..ntex (ntex = nothing to add)
	r.sort_bytes = 0;
	(void)refl_sort(nullptr, r.sort_bytes, r.key_bits, nullptr, nullptr, nullptr, r.n, 0);
	r.total_floats = (r.ntex + 2) * 4 + 8 * r.n + 3 * r.n + (r.sort_bytes + 3) / 4 + 128;   // + slack to align the sort temp
	return r;
}
extern "C" size_t gsr_deferred_reflection_scratch_floats(uint32_t L, int width, int height, int binned) {
	if (L == 0 || width <= 0 || height <= 0) return 0;
	if (!binned) return ((size_t)6 * L * L + 1) * 4;
	return refl_scratch(L, width, height).total_floats;
}

// ---- side stream for the texel-gradient tail of the reflection backward.  Only the pixel kernel of that backward feeds the
// rasterizer backward that follows it; the sort / combine / unpack that produce dL_dcubemap (0.13 of 0.22 ms at 1080p, all
// latency- or gather-bound, a few hundred workgroups) feed nothing until the optimizer or the all-reduce.  With async_tail
// they are enqueued on a library-owned stream that forks from the caller's stream after the pixel kernel and run
// beside the (VALU-bound) tile backward; gsr_side_join() makes a stream wait for them.  One side stream per device, so
// successive tails (a batch of views accumulating into one gradient) stay ordered among themselves.
// Priority of the side stream.  Highest: the tail's small workgroups go first whenever a slot frees up, its look-back chains move
// and it is over early (measured, interleaved runs of the C3 step: highest 1.917-1.922 ms; default 1.905-1.920 ms in two runs of four but
// 2.44 / 2.55 ms in the other two — default-priority streams share the runtime's pool of hardware queues and the step then
// depends on which queue the side stream happened to get; lowest 1.98 ms: the tail only runs once the tile backward has drained).
#ifndef GSR_SIDE_PRIO
#define GSR_SIDE_PRIO 1     // 0: default priority; 1: highest; -1: lowest
#endif
namespace {
struct SideStream {
	hipStream_t stream = nullptr;
	hipEvent_t fork = nullptr, fork2 = nullptr, done = nullptr;
	bool recorded = false;      // `done` has been recorded at least once (an event that was never recorded must not be waited on)
};
std::mutex g_side_mu;
SideStream g_side[64];
SideStream* side_stream() {   // (g_side_mu held)
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
	SideStream& s = g_side[dev];
	if (!s.stream) {
#if GSR_SIDE_PRIO
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, GSR_SIDE_PRIO > 0 ? greatest : least) != hipSuccess) { s.stream = nullptr; return nullptr; }
#else
		if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { s.stream = nullptr; return nullptr; }
#endif
		if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.fork2, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
			(void)hipStreamDestroy(s.stream);
			s.stream = nullptr;
			return nullptr;
		}
	}
	return &s;
}
}  // namespace

extern "C" int gsr_side_join(void* stream_) {
	std::lock_guard<std::mutex> lk(g_side_mu);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
	SideStream& s = g_side[dev];
	// Every joining stream waits on the LAST recorded `done` (not cleared by the first waiter: an all-reduce stream and then an
	// optimizer stream may both consume the sink); waiting on an event that has already completed costs nothing on the device.
	if (s.stream && s.recorded) GSR_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream_, s.done, 0));
	return 0;
}

extern "C" int gsr_deferred_reflection_backward_ex(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                   const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                   const float* g_final, const float* g_refl_color, const float* g_normal_world,
                                                   float* g_normal_view, float* g_base, float* g_strength, float* g_cubemap, float* g_fail,
                                                   float* scratch, size_t scratch_floats, int accumulate, int async_tail, const float* cubemap_rgba,
                                                   const uint32_t* sort_keys, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (width <= 0 || height <= 0 || !normal_view || !base_color || !refl_strength || !cam || !cubemap || !fail_value || !g_final ||
	    !g_normal_view || !g_base || !g_strength || !g_cubemap || !g_fail || !scratch || L == 0) {
		set_error("gsr_deferred_reflection_backward: invalid argument");
		return GSR_E_INVALID;
	}
	const size_t HW = (size_t)width * height;
	const ReflScratch rs = refl_scratch(L, width, height);
	const size_t ntex = rs.ntex;
	if (scratch_floats < (ntex + 1) * 4) { set_error("gsr_deferred_reflection_backward: scratch smaller than (6*L*L+1)*4 floats"); return GSR_E_INVALID; }
	float* fail_acc = scratch + ntex * 4;   // [texel staging ntex*4][fail-value gradient 4]
	const bool binned = scratch_floats >= rs.total_floats && rs.n < ((size_t)1 << 30) && rs.ntex < 0xFFFFFFFFull && ((uintptr_t)scratch & 31) == 0;
	GSR_HIP_CHECK(hipMemsetAsync(scratch, 0, (ntex + 1) * 4 * sizeof(float), stream));
	auto st_ = std::make_unique<StageTimer>(GSR_STAGE_REFL_BWD, stream);      // the pixel kernel; the texel-gradient tail is GSR_STAGE_REFL_BWD_TAIL
	const unsigned grid = (unsigned)((HW * 4 + 255) / 256);
	if (!binned) {
		// texel gradients by float atomics straight from the pixel kernel (memory-side, ~2.5 requests per pixel)
		deferred_refl_bwd_kernel<<<grid, 256, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap, fail_value, (int)L, width, height, g_final,
		                                                  g_refl_color, g_normal_world, g_normal_view, g_base, g_strength, scratch, fail_acc);
	} else {
		// sorted footprints: the pixel kernel stores one footprint record per pixel and its texel id as a sort key; a radix sort of
		// (texel id, pixel) makes equal texels adjacent; refl_run_combine_kernel gathers the records in that order, sums runs in
		// registers and a workgroup's texel range in LDS.
		ReflFootprint* fp = reinterpret_cast<ReflFootprint*>(scratch + (ntex + 1) * 4 + 4);   // 32-byte aligned as long as scratch is
		uint32_t* keys_in = reinterpret_cast<uint32_t*>(fp + rs.n);
		uint32_t* keys_out = keys_in + rs.n;
		uint32_t* pix_out = keys_out + rs.n;
		void* sort_temp = reinterpret_cast<void*>(((uintptr_t)(pix_out + rs.n) + 255) & ~(uintptr_t)255);
		const unsigned egrid = (unsigned)((HW + 255) / 256);
		// sort_keys: the forward already wrote the keys (gsr_deferred_reflection_forward_ex), so the sort depends on nothing this call
		// computes: with async_tail it forks BEFORE the pixel kernel and runs beside it (an HBM- and latency-bound kernel without LDS: the
		// 1024-thread passes find room at once) and only the combine waits for the records.  Without them the sort follows the pixel
		// kernel and, on the side stream, has to share the chip with the tile backward: the small shape then (see refl_sort).
		const bool small_sort = async_tail && !sort_keys && REFL_SMALL_SORT;
		const size_t clr = refl_sort_cleared_bytes(rs.key_bits, rs.n, small_sort);
		hipStream_t tail = stream;
		SideStream* side = nullptr;
		std::unique_lock<std::mutex> lk(g_side_mu, std::defer_lock);
		if (async_tail) {
			lk.lock();
			side = side_stream();
			if (side) tail = side->stream;
		}
		size_t sb = rs.sort_bytes;
		std::unique_ptr<StageTimer> tail_timer;     // GSR_STAGE_REFL_BWD_TAIL is timed on the stream it runs on: with async_tail NOT inside GSR_STAGE_REFL_BWD's events
		if (sort_keys) {
			if (side) {
				GSR_HIP_CHECK(hipEventRecord(side->fork, stream));
				GSR_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
			}
			tail_timer = std::make_unique<StageTimer>(GSR_STAGE_REFL_BWD_TAIL, tail);
			GSR_HIP_CHECK(hipMemsetAsync(sort_temp, 0, clr, tail));
			GSR_HIP_CHECK(refl_sort(sort_temp, sb, rs.key_bits, sort_keys, keys_out, pix_out, rs.n, tail, true, false));
		}
		if (cubemap_rgba && ((uintptr_t)cubemap_rgba & 15) == 0)     // the texel-interleaved copy the forward made (same cubemap)
			deferred_refl_bwd_entries_kernel<true><<<egrid, 256, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap,
			                                                                 reinterpret_cast<const float4*>(cubemap_rgba), fail_value, (int)L, width, height, g_final,
			                                                                 g_refl_color, g_normal_world, g_normal_view, g_base, g_strength, fail_acc, scratch, fp,
			                                                                 keys_in, sort_keys, (uint32_t)ntex, sort_keys ? nullptr : sort_temp, sort_keys ? 0 : clr);
		else
			deferred_refl_bwd_entries_kernel<false><<<egrid, 256, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap, nullptr, fail_value, (int)L, width,
			                                                                  height, g_final, g_refl_color, g_normal_world, g_normal_view, g_base, g_strength, fail_acc,
			                                                                  scratch, fp, keys_in, sort_keys, (uint32_t)ntex, sort_keys ? nullptr : sort_temp,
			                                                                  sort_keys ? 0 : clr);
		// the per-pixel gradients are complete here; what follows only produces dL_dcubemap / dL_dfail
		st_.reset();
		if (side) {     // the combine needs the records (and, without forward keys, the sort needs the keys) the pixel kernel just wrote
			hipEvent_t ev = sort_keys ? side->fork2 : side->fork;
			GSR_HIP_CHECK(hipEventRecord(ev, stream));
			GSR_HIP_CHECK(hipStreamWaitEvent(side->stream, ev, 0));
		}
		if (!sort_keys) {
			tail_timer = std::make_unique<StageTimer>(GSR_STAGE_REFL_BWD_TAIL, tail);
			GSR_HIP_CHECK(refl_sort(sort_temp, sb, rs.key_bits, keys_in, keys_out, pix_out, rs.n, tail, true, small_sort));
		}
		{
		const size_t per_wg = (size_t)256 * REFL_CHUNK;
		refl_run_combine_kernel<<<(unsigned)((rs.n + per_wg - 1) / per_wg), 256, 0, tail>>>(keys_out, pix_out, fp, rs.n, L, (uint32_t)ntex, scratch);
		auto unpack = accumulate ? unpack_cubemap_grad_kernel<true> : unpack_cubemap_grad_kernel<false>;
		unpack<<<(unsigned)((ntex + 255) / 256), 256, 0, tail>>>((const float4*)scratch, g_cubemap, g_fail, (int)L);
		}
		tail_timer.reset();
		if (side) {
			GSR_HIP_CHECK(hipEventRecord(side->done, side->stream));
			side->recorded = true;
		}
		GSR_LAUNCH_CHECK(0, stream);
		return 0;
	}
	auto unpack = accumulate ? unpack_cubemap_grad_kernel<true> : unpack_cubemap_grad_kernel<false>;
	unpack<<<(unsigned)((ntex + 255) / 256), 256, 0, stream>>>((const float4*)scratch, g_cubemap, g_fail, (int)L);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_deferred_reflection_backward_accum(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                const float* g_final, const float* g_refl_color, const float* g_normal_world,
                                                float* g_normal_view, float* g_base, float* g_strength, float* g_cubemap, float* g_fail,
                                                float* scratch, size_t scratch_floats, int accumulate, void* stream_) {
	return gsr_deferred_reflection_backward_ex(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, g_final, g_refl_color,
	                                           g_normal_world, g_normal_view, g_base, g_strength, g_cubemap, g_fail, scratch, scratch_floats, accumulate, 0, nullptr,
	                                           nullptr, stream_);
}

extern "C" int gsr_deferred_reflection_backward(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                const float* g_final, const float* g_refl_color, const float* g_normal_world,
                                                float* g_normal_view, float* g_base, float* g_strength, float* g_cubemap, float* g_fail,
                                                float* scratch, size_t scratch_floats, void* stream_) {
	return gsr_deferred_reflection_backward_accum(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, g_final, g_refl_color,
	                                              g_normal_world, g_normal_view, g_base, g_strength, g_cubemap, g_fail, scratch, scratch_floats, 0, stream_);
}

// ---- shading normal alone (the reference's initial stage renders without the reflection chain but still returns
// rend_normal = normalize(allmap[2:5] rotated to world space), gaussian_renderer/__init__.py:148,178-179 of the reference).
__global__ void __launch_bounds__(256) normal_world_fwd_kernel(const float* __restrict__ normal_view, const float* __restrict__ cam, size_t HW,
                                                               float* __restrict__ out) {
	const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (pix >= HW) return;
	const float vx = normal_view[pix], vy = normal_view[HW + pix], vz = normal_view[2 * HW + pix];
	const float wx = vx * cam[0] + vy * cam[1] + vz * cam[2], wy = vx * cam[3] + vy * cam[4] + vz * cam[5], wz = vx * cam[6] + vy * cam[7] + vz * cam[8];
	const float inv = 1.0f / (sqrtf(wx * wx + wy * wy + wz * wz) + 1e-6f);
	out[pix] = wx * inv; out[HW + pix] = wy * inv; out[2 * HW + pix] = wz * inv;
}
__global__ void __launch_bounds__(256) normal_world_bwd_kernel(const float* __restrict__ normal_view, const float* __restrict__ cam, size_t HW,
                                                               const float* __restrict__ g_out, float* __restrict__ g_normal_view) {
	const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (pix >= HW) return;
	const float vx = normal_view[pix], vy = normal_view[HW + pix], vz = normal_view[2 * HW + pix];
	const float wx = vx * cam[0] + vy * cam[1] + vz * cam[2], wy = vx * cam[3] + vy * cam[4] + vz * cam[5], wz = vx * cam[6] + vy * cam[7] + vz * cam[8];
	const float len = sqrtf(wx * wx + wy * wy + wz * wz);
	const float inv = 1.0f / (len + 1e-6f);
	const float gx = g_out[pix], gy = g_out[HW + pix], gz = g_out[2 * HW + pix];
	// n = w / (|w| + eps):  g_w = g / (len + eps) - w (w . g) / (len (len + eps)^2), zero subgradient at len = 0
	float ax = gx * inv, ay = gy * inv, az = gz * inv;
	if (len > 0.f) {
		const float k = (wx * gx + wy * gy + wz * gz) * inv * inv / len;
		ax -= wx * k; ay -= wy * k; az -= wz * k;
	}
#pragma unroll
	for (int c = 0; c < 3; c++) g_normal_view[c * HW + pix] = ax * cam[c] + ay * cam[3 + c] + az * cam[6 + c];
}
extern "C" int gsr_normal_world_forward(const float* normal_view, const float* cam, int width, int height, float* out_normal_world, void* stream_) {
	if (width <= 0 || height <= 0 || !normal_view || !cam || !out_normal_world) { set_error("gsr_normal_world_forward: invalid argument"); return GSR_E_INVALID; }
	const size_t HW = (size_t)width * height;
	normal_world_fwd_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, (hipStream_t)stream_>>>(normal_view, cam, HW, out_normal_world);
	GSR_LAUNCH_CHECK(0, (hipStream_t)stream_);
	return 0;
}
extern "C" int gsr_normal_world_backward(const float* normal_view, const float* cam, int width, int height, const float* g_normal_world,
                                         float* g_normal_view, void* stream_) {
	if (width <= 0 || height <= 0 || !normal_view || !cam || !g_normal_world || !g_normal_view) { set_error("gsr_normal_world_backward: invalid argument"); return GSR_E_INVALID; }
	const size_t HW = (size_t)width * height;
	normal_world_bwd_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, (hipStream_t)stream_>>>(normal_view, cam, HW, g_normal_world, g_normal_view);
	GSR_LAUNCH_CHECK(0, (hipStream_t)stream_);
	return 0;
}
