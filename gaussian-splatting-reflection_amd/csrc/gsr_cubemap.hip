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
#include "gsr_refl.hpp"

namespace gsr {



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
template <bool RGBA>
__global__ void __launch_bounds__(256)
deferred_refl_fwd_kernel(const float* __restrict__ normal_view, const float* __restrict__ base, const float* __restrict__ strength,
                         const float* __restrict__ cam, const float* __restrict__ cubemap, const float4* __restrict__ rgba,
                         const float* __restrict__ fail_value, int L, int W, int H, float* __restrict__ out_final, float* __restrict__ out_refl,
                         float* __restrict__ out_nworld, uint32_t* __restrict__ sort_keys, uint32_t no_key) {
	const size_t HW = (size_t)W * H;
	const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (pix >= HW) return;
	const int py = (int)(pix / W), px = (int)(pix - (size_t)py * W);
	// loads that do not depend on the reflected direction are issued together with the normal (one memory round trip less
	// per wave; the kernel is a chain of dependent round trips at five waves per SIMD)
	const float nvx = normal_view[pix], nvy = normal_view[HW + pix], nvz = normal_view[2 * HW + pix];
	const float sv = strength[pix];
	const float b0 = base[pix], b1 = base[HW + pix], b2 = base[2 * HW + pix];
	ReflFwdOut o;
	refl_forward_pixel<RGBA>(cam, cubemap, rgba, fail_value, L, nvx, nvy, nvz, px, py, sv, b0, b1, b2, no_key, o);   // (gsr_refl.hpp)
	// sort key of the backward's footprint record (deferred_refl_bwd_entries_kernel): it depends on forward data only, so the sort can
	// run beside the backward's pixel kernel
	if (sort_keys) sort_keys[pix] = o.key;
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		out_refl[ch * HW + pix] = o.refl_c[ch];
		out_final[ch * HW + pix] = o.final_c[ch];
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
	const float inv = refl_rcp(o.len + 1e-6f);      // (the same reciprocal forms as refl_backward_pixel, gsr_refl.hpp)
	float gwx = gnx * inv, gwy = gny * inv, gwz = gnz * inv;
	if (o.len > 0.f) {
		const float k = (o.nwx * gnx + o.nwy * gny + o.nwz * gnz) * inv * inv * refl_rcp(o.len);
		gwx -= o.nwx * k; gwy -= o.nwy * k; gwz -= o.nwz * k;
	}
	if (chan) g_normal_view[c * HW + p] = gwx * cam[c] + gwy * cam[3 + c] + gwz * cam[6 + c];
}

// The same backward for the sorted-footprint path (`binned` in the C ABI), one lane per pixel: nothing here needs the four lanes of the quad version (they
// exist to pair the texel atomics), so the index math runs once per pixel instead of four times.  The per-pixel body is
// refl_backward_pixel (gsr_refl.hpp), which the tile backward of the surfel rasterizer also runs as its prologue (fused path).
// Workgroup size 512 (round 4; no LDS, no barrier: any multiple of 64 is correct).  Together with 512-thread workgroups in the key sort
// that runs beside this kernel on the side stream (REFL_SORT_BS): a 1024-thread sort workgroup needs sixteen free wave slots on one CU at
// once and never got them while this kernel was dispatching 256-thread workgroups — the sort's 8-us histogram took the kernel's whole 60 us,
// its first pass started when this kernel ended and the tile backward then waited for the gate in front of the second pass.  With 512 / 512
// the two interleave: histogram 13 us and first pass 43 us beside this kernel (62 -> 81 us), gate open before it ends; tile backward starts
// 93 us after this kernel instead of 110 (profiles/r04_trace_c3_step_refl_sort_512.txt; three interleaved A/B runs: -8 ... -18 us per step).
#ifndef ENTRIES_BS
#define ENTRIES_BS 512
#endif
template <bool RGBA>
__global__ void __launch_bounds__(ENTRIES_BS)
deferred_refl_bwd_entries_kernel(const float* __restrict__ normal_view, const float* __restrict__ base, const float* __restrict__ strength,
                                 const float* __restrict__ cam, const float* __restrict__ cubemap, const float4* __restrict__ rgba,
                                 const float* __restrict__ fail_value, int L,
                                 int W, int H, const float* __restrict__ g_final, const float* __restrict__ g_refl_color,
                                 const float* __restrict__ g_nworld, float* __restrict__ g_normal_view, float* __restrict__ g_base,
                                 float* __restrict__ g_strength, float* __restrict__ g_fail, float* __restrict__ g_scratch,
                                 ReflFootprint* __restrict__ footprints, uint32_t* __restrict__ keys, const uint32_t* __restrict__ keys_fwd, uint32_t no_key,
                                 void* sort_clear, size_t sort_clear_bytes) {
	const size_t HW = (size_t)W * H;
	const size_t pix = (size_t)blockIdx.x * ENTRIES_BS + threadIdx.x;
	sort_clear_region(sort_clear, sort_clear_bytes, pix, (size_t)gridDim.x * ENTRIES_BS);   // look-back state of the sort that follows
	const bool live = pix < HW;
	const size_t p = live ? pix : 0;
	const int py = (int)(p / W), px = (int)(p - (size_t)py * W);
	// every load that does not depend on the reflected direction is issued here, with the normal
	ReflBwdIn in;
	in.nvx = normal_view[p]; in.nvy = normal_view[HW + p]; in.nvz = normal_view[2 * HW + p];
	in.sv = strength[p];
	in.has_grc = g_refl_color != nullptr;
	in.has_gnw = g_nworld != nullptr;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		in.gfin[c] = g_final[c * HW + p];
		in.bas[c] = base[c * HW + p];
		in.grc[c] = g_refl_color ? g_refl_color[c * HW + p] : 0.f;
		in.gnw[c] = g_nworld ? g_nworld[c * HW + p] : 0.f;
	}
	const uint32_t kf = keys_fwd ? keys_fwd[p] : no_key;
	ReflBwdOut o;
	refl_backward_pixel<RGBA>(cam, cubemap, rgba, fail_value, L, px, py, live, in, g_fail, g_scratch, footprints + p, keys_fwd != nullptr, kf,
	                          keys ? keys + p : nullptr, no_key, (int)(threadIdx.x & 63), o);
	if (!live) return;
	g_strength[p] = o.g_strength;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		g_base[c * HW + p] = o.g_base[c];
		g_normal_view[c * HW + p] = o.g_nv[c];
	}
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

extern "C" int gsr_deferred_reflection_forward_keys(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
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
/* the round-2 signature of gsr_deferred_reflection_forward_ex (no sort keys), kept for callers built against the earlier header */
extern "C" int gsr_deferred_reflection_forward_ex(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                  const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                  float* out_final, float* out_refl_color, float* out_normal_world, float* cubemap_rgba, void* stream_) {
	return gsr_deferred_reflection_forward_keys(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, out_final, out_refl_color,
	                                            out_normal_world, cubemap_rgba, nullptr, stream_);
}
extern "C" int gsr_deferred_reflection_forward(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                               const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                               float* out_final, float* out_refl_color, float* out_normal_world, void* stream_) {
	return gsr_deferred_reflection_forward_keys(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, out_final, out_refl_color,
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
// Shape of the 9-bit sort (L = 128).  Round 4: 512 x 16 instead of 1024 x 8 — the same 8192 pairs per workgroup, but a workgroup that finds
// room beside the backward's pixel kernel (see ENTRIES_BS); alone on the chip the two shapes are within 2 us of each other.
#ifndef REFL_SORT_BS
#define REFL_SORT_BS 512
#endif
#ifndef REFL_SORT_IPT
#define REFL_SORT_IPT 16
#endif
static hipError_t refl_sort(void* temp, size_t& bytes, int key_bits, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* pix_out, size_t n, hipStream_t stream,
                            bool pre_cleared = false, bool small = false, hipEvent_t gate = nullptr) {
	rocprim::counting_iterator<uint32_t> pix_in(0);
	if (temp == nullptr) {   // size query: the largest of the drivers' needs (neither the runtime switch nor the stream a tail runs on changes a scratch size)
		size_t own = 0, pub = 0, sm = 0;
		if (key_bits > 16 && key_bits <= 18) (void)onesweep_sort_pairs<REFL_SORT_BS, REFL_SORT_IPT, 9>(nullptr, own, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		else if (key_bits > 18 && key_bits <= 20) (void)onesweep_sort_pairs<1024, 8, 10>(nullptr, own, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		else (void)onesweep_sort_pairs<1024, 8, 8>(nullptr, own, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		(void)onesweep_sort_pairs<256, 8, 8>(nullptr, sm, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream);
		(void)rocprim::radix_sort_pairs(nullptr, pub, (const uint32_t*)keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, false);
		bytes = own > pub ? own : pub;
		if (sm > bytes) bytes = sm;
		return hipSuccess;
	}
	if (!option_sort_driver()) {  // gsr_set_option("sort_driver", 0) or an unknown rocPRIM release: the public entry point (the gate then is the sort's end)
		hipError_t e = rocprim::radix_sort_pairs(temp, bytes, (const uint32_t*)keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, false);
		if (e == hipSuccess && gate) e = hipEventRecord(gate, stream);
		return e;
	}
	if (small)
		return onesweep_sort_pairs<256, 8, 8>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared, nullptr, gate);
	if (key_bits > 16 && key_bits <= 18)
		return onesweep_sort_pairs<REFL_SORT_BS, REFL_SORT_IPT, 9>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared, nullptr, gate);
	if (key_bits > 18 && key_bits <= 20)
		return onesweep_sort_pairs<1024, 8, 10>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared, nullptr, gate);
	return onesweep_sort_pairs<1024, 8, 8>(temp, bytes, keys_in, keys_out, pix_in, pix_out, n, 0u, (unsigned)key_bits, stream, pre_cleared, nullptr, gate);
}
static size_t refl_sort_cleared_bytes(int key_bits, size_t n, bool small) {
	if (small) return onesweep_cleared_bytes<256, 8, 8>(n, 0u, (unsigned)key_bits);
	if (key_bits > 16 && key_bits <= 18) return onesweep_cleared_bytes<REFL_SORT_BS, REFL_SORT_IPT, 9>(n, 0u, (unsigned)key_bits);
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
	hipEvent_t fork = nullptr, fork2 = nullptr, done = nullptr, sorted = nullptr, gate = nullptr;
	bool recorded = false;      // `done` has been recorded at least once (an event that was never recorded must not be waited on)
	bool sorted_recorded = false;
	bool gate_pending = false;  // `gate` = "the last pass of the key sort is next on the side stream": the rasterizer's tile backward waits for it ONCE
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
		    hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.sorted, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s.gate, hipEventDisableTiming) != hipSuccess) {
			(void)hipStreamDestroy(s.stream);
			s.stream = nullptr;
			return nullptr;
		}
	}
	return &s;
}
}  // namespace

// The gate of the key sort (see SideStream::gate_pending): called by the rasterizer backward right before its tile kernel.  A sort pass of
// 1024-thread workgroups that starts AFTER that kernel has filled the chip waits for the whole 0.7 ms of it (its single-wave workgroups
// retire one at a time, never sixteen wave slots of a CU at once), the run combine behind it then runs beside the per-Gaussian backward
// (0.115 -> 0.155 ms) and ends after it; a pass that has started before gets its CUs first (highest stream priority).  So the tile
// kernel is held back until the LAST pass is next in line — measured: 13 us of hold for 55 us back.
namespace gsr {
int side_gate_wait(hipStream_t stream) {
	std::lock_guard<std::mutex> lk(g_side_mu);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
	SideStream& s = g_side[dev];
	if (s.stream && s.gate_pending) {
		s.gate_pending = false;
		GSR_HIP_CHECK(hipStreamWaitEvent(stream, s.gate, 0));
	}
	return 0;
}
}  // namespace gsr

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

namespace gsr {
// ---- host side of the texel-gradient tail, shared by gsr_deferred_reflection_backward* and the fused surfel backward
// (gsr_surfel_backward_refl, gsr_surfel.hip), whose tile kernel runs the pixel part as its prologue.  Order of use:
//   refl_tail_begin   carve the scratch, pick the stream of the tail (locks the side stream when async)
//   [the caller zeroes clear[0..1] on `stream` — refl_tail_clear() does it with two fills — before anything else]
//   refl_tail_sort    with forward keys: fork the side stream and start the sort (it depends on nothing the pixel code computes)
//   [the caller's pixel code on `stream`: records, rim atomics into `staging`, fail-value sums into `fail_acc`]
//   refl_tail_finish  join the pixel code, (sort,) run combine, unpack, `done` event
int refl_tail_begin(ReflTail& t, uint32_t L, int width, int height, float* scratch, size_t scratch_floats, const uint32_t* sort_keys, int async_tail,
                    int accumulate, float* g_cubemap, float* g_fail, hipStream_t stream) {
	const ReflScratch rs = refl_scratch(L, width, height);
	if (scratch_floats < rs.total_floats || rs.n >= ((size_t)1 << 30) || rs.ntex >= 0xFFFFFFFFull || ((uintptr_t)scratch & 31) != 0) {
		set_error("reflection backward (sorted footprints): scratch of %zu floats, 32-byte aligned, needed (gsr_deferred_reflection_scratch_floats(L, W, H, 1))",
		          rs.total_floats);
		return GSR_E_INVALID;
	}
	t.L = L; t.n = rs.n; t.ntex = rs.ntex; t.key_bits = rs.key_bits; t.sort_bytes = rs.sort_bytes;
	t.staging = scratch;
	t.fail_acc = scratch + rs.ntex * 4;   // [texel staging ntex*4][fail-value gradient 4]
	ReflFootprint* fp = reinterpret_cast<ReflFootprint*>(scratch + (rs.ntex + 1) * 4 + 4);   // 32-byte aligned as long as scratch is
	t.footprints = fp;
	t.keys_in = reinterpret_cast<uint32_t*>(fp + rs.n);
	t.keys_out = t.keys_in + rs.n;
	t.pix_out = t.keys_out + rs.n;
	t.sort_temp = reinterpret_cast<void*>(((uintptr_t)(t.pix_out + rs.n) + 255) & ~(uintptr_t)255);
	t.sort_keys = sort_keys;
	// sort_keys: the forward already wrote the keys, so the sort depends on nothing the backward computes: with async_tail it forks
	// BEFORE the pixel code and runs beside it and only the combine waits for the records.  Without keys the sort follows the pixel code and,
	// on the side stream, has to share the chip with the tile backward: the small shape then (see refl_sort).
	t.small_sort = async_tail && !sort_keys && REFL_SMALL_SORT;
	t.clear_ptr[0] = scratch; t.clear_bytes[0] = (rs.ntex + 1) * 4 * sizeof(float);
	t.clear_ptr[1] = t.sort_temp; t.clear_bytes[1] = refl_sort_cleared_bytes(rs.key_bits, rs.n, t.small_sort);
	t.stream = stream; t.tail = stream; t.side = nullptr; t.locked = false;
	t.g_cubemap = g_cubemap; t.g_fail = g_fail; t.accumulate = accumulate;
	if (async_tail) {
		g_side_mu.lock();
		t.locked = true;
		SideStream* side = side_stream();
		if (side) { t.side = side; t.tail = side->stream; }
	}
	return 0;
}
void refl_tail_abort(ReflTail& t) {
	if (t.locked) { g_side_mu.unlock(); t.locked = false; }
}
int refl_tail_clear(ReflTail& t) {
	hipError_t e = hipMemsetAsync(t.clear_ptr[0], 0, t.clear_bytes[0], t.stream);
	if (e == hipSuccess) e = hipMemsetAsync(t.clear_ptr[1], 0, t.clear_bytes[1], t.stream);
	if (e != hipSuccess) { refl_tail_abort(t); set_error("reflection backward: clearing the scratch failed: %s", hipGetErrorString(e)); return GSR_E_HIP; }
	return 0;
}
#define REFL_TAIL_CHECK(expr)                                                                                     \
	do {                                                                                                          \
		hipError_t _e = (expr);                                                                                   \
		if (_e != hipSuccess) {                                                                                   \
			refl_tail_abort(t);                                                                                   \
			gsr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);            \
			return GSR_E_HIP;                                                                                     \
		}                                                                                                         \
	} while (0)
// With forward keys: fork the side stream and start the sort (between the clears and the pixel code).
int refl_tail_sort(ReflTail& t) {
	if (!t.sort_keys) return 0;
	SideStream* side = static_cast<SideStream*>(t.side);
	if (side) {
		REFL_TAIL_CHECK(hipEventRecord(side->fork, t.stream));
		REFL_TAIL_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
	}
	size_t sb = t.sort_bytes;
	StageTimer tt(GSR_STAGE_REFL_BWD_TAIL, t.tail);     // timed on the stream it runs on: with async_tail NOT inside GSR_STAGE_REFL_BWD's events
	REFL_TAIL_CHECK(refl_sort(t.sort_temp, sb, t.key_bits, t.sort_keys, t.keys_out, t.pix_out, t.n, t.tail, true, false, side ? side->gate : nullptr));
	if (side) side->gate_pending = true;
	return 0;
}
// The sort of the footprint keys, EARLY: called by a forward that has just written the keys (gsr_surfel_forward_refl) with the scratch the
// backward will use.  async: on the side stream, forked behind the kernel that wrote the keys — it then runs beside whatever the caller
// enqueues next (the loss; the fills and the pixel kernel of the backward) while the chip still has room: a 1024-thread sort workgroup
// cannot start on a CU the tile backward has filled (round 3), and since round 4 the backward's pixel kernel is short enough that a sort
// forked in the backward no longer finished before that kernel took the chip (its second pass then waited 0.8 ms and the run combine ran
// after the tile backward, beside — and at the expense of — the per-Gaussian backward).  The backward is told with keys_sorted = 1.
int refl_sort_keys_early(uint32_t L, int width, int height, float* scratch, size_t scratch_floats, const uint32_t* sort_keys, int async, hipStream_t stream) {
	ReflTail t;
	int rc = refl_tail_begin(t, L, width, height, scratch, scratch_floats, sort_keys, async, 0, nullptr, nullptr, stream);
	if (rc < 0) return rc;
	SideStream* side = static_cast<SideStream*>(t.side);
	if (side) {
		REFL_TAIL_CHECK(hipEventRecord(side->fork, t.stream));
		REFL_TAIL_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
	}
	{
		StageTimer tt(GSR_STAGE_REFL_BWD_TAIL, t.tail);
		REFL_TAIL_CHECK(hipMemsetAsync(t.clear_ptr[1], 0, t.clear_bytes[1], t.tail));
		size_t sb = t.sort_bytes;
		REFL_TAIL_CHECK(refl_sort(t.sort_temp, sb, t.key_bits, t.sort_keys, t.keys_out, t.pix_out, t.n, t.tail, true, false, side ? side->gate : nullptr));
	}
	if (side) {
		side->gate_pending = true;
		REFL_TAIL_CHECK(hipEventRecord(side->sorted, side->stream));
		side->sorted_recorded = true;
	}
	refl_tail_abort(t);
	return 0;
}
// keys_sorted: the backward's counterpart — the stream its tail runs on is ordered behind the early sort (a no-op when that is the side
// stream itself: the combine is enqueued there behind the sort anyway)
int refl_tail_join_early_sort(ReflTail& t) {
	std::unique_lock<std::mutex> lk(g_side_mu, std::defer_lock);
	if (!t.locked) lk.lock();
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
	SideStream& s = g_side[dev];
	if (s.stream && s.sorted_recorded && t.tail != s.stream) REFL_TAIL_CHECK(hipStreamWaitEvent(t.tail, s.sorted, 0));
	return 0;
}

int refl_tail_finish(ReflTail& t) {
	SideStream* side = static_cast<SideStream*>(t.side);
	if (side) {     // the combine needs the records (and, without forward keys, the sort needs the keys) the pixel code just wrote
		hipEvent_t ev = t.sort_keys ? side->fork2 : side->fork;
		REFL_TAIL_CHECK(hipEventRecord(ev, t.stream));
		REFL_TAIL_CHECK(hipStreamWaitEvent(side->stream, ev, 0));
	}
	{
		StageTimer tt(GSR_STAGE_REFL_BWD_TAIL, t.tail);
		const size_t per_wg = (size_t)256 * REFL_CHUNK;
		const unsigned cgrid = (unsigned)((t.n + per_wg - 1) / per_wg);
		const ReflFootprint* fp = static_cast<const ReflFootprint*>(t.footprints);
		if (!t.sort_keys) {
			size_t sb = t.sort_bytes;
			REFL_TAIL_CHECK(refl_sort(t.sort_temp, sb, t.key_bits, t.keys_in, t.keys_out, t.pix_out, t.n, t.tail, true, t.small_sort));
		}
		refl_run_combine_kernel<<<cgrid, 256, 0, t.tail>>>(t.keys_out, t.pix_out, fp, t.n, t.L, (uint32_t)t.ntex, t.staging);
		auto unpack = t.accumulate ? unpack_cubemap_grad_kernel<true> : unpack_cubemap_grad_kernel<false>;
		unpack<<<(unsigned)((t.ntex + 255) / 256), 256, 0, t.tail>>>((const float4*)t.staging, t.g_cubemap, t.g_fail, (int)t.L);
	}
	if (side) {
		REFL_TAIL_CHECK(hipEventRecord(side->done, side->stream));
		side->recorded = true;
	}
	REFL_TAIL_CHECK(hipGetLastError());
	refl_tail_abort(t);      // (releases the side-stream lock)
	return 0;
}

}  // namespace gsr

extern "C" int gsr_deferred_reflection_backward_keys(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                   const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                   const float* g_final, const float* g_refl_color, const float* g_normal_world,
                                                   float* g_normal_view, float* g_base, float* g_strength, float* g_cubemap, float* g_fail,
                                                   float* scratch, size_t scratch_floats, int accumulate, int async_tail, const float* cubemap_rgba,
                                                   const uint32_t* sort_keys, int keys_sorted, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (width <= 0 || height <= 0 || !normal_view || !base_color || !refl_strength || !cam || !cubemap || !fail_value || !g_final ||
	    !g_normal_view || !g_base || !g_strength || !g_cubemap || !g_fail || !scratch || L == 0 || (keys_sorted && !sort_keys)) {
		set_error("gsr_deferred_reflection_backward: invalid argument");
		return GSR_E_INVALID;
	}
	const size_t HW = (size_t)width * height;
	const ReflScratch rs = refl_scratch(L, width, height);
	const size_t ntex = rs.ntex;
	if (scratch_floats < (ntex + 1) * 4) { set_error("gsr_deferred_reflection_backward: scratch smaller than (6*L*L+1)*4 floats"); return GSR_E_INVALID; }
	const bool binned = scratch_floats >= rs.total_floats && rs.n < ((size_t)1 << 30) && rs.ntex < 0xFFFFFFFFull && ((uintptr_t)scratch & 31) == 0;
	if (keys_sorted && !binned) { set_error("gsr_deferred_reflection_backward: keys_sorted needs the scratch the forward sorted into"); return GSR_E_INVALID; }
	if (!binned) {
		// texel gradients by float atomics straight from the pixel kernel (memory-side, ~2.5 requests per pixel)
		float* fail_acc = scratch + ntex * 4;   // [texel staging ntex*4][fail-value gradient 4]
		GSR_HIP_CHECK(hipMemsetAsync(scratch, 0, (ntex + 1) * 4 * sizeof(float), stream));
		{
			StageTimer st_(GSR_STAGE_REFL_BWD, stream);
			const unsigned grid = (unsigned)((HW * 4 + 255) / 256);
			deferred_refl_bwd_kernel<<<grid, 256, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap, fail_value, (int)L, width, height, g_final,
			                                                  g_refl_color, g_normal_world, g_normal_view, g_base, g_strength, scratch, fail_acc);
		}
		auto unpack = accumulate ? unpack_cubemap_grad_kernel<true> : unpack_cubemap_grad_kernel<false>;
		unpack<<<(unsigned)((ntex + 255) / 256), 256, 0, stream>>>((const float4*)scratch, g_cubemap, g_fail, (int)L);
		GSR_LAUNCH_CHECK(0, stream);
		return 0;
	}
	// sorted footprints: the pixel kernel stores one footprint record per pixel and its texel id as a sort key; a radix sort of
	// (texel id, pixel) makes equal texels adjacent; refl_run_combine_kernel gathers the records in that order, sums runs in
	// registers and a workgroup's texel range in LDS.
	ReflTail t;
	int rc = refl_tail_begin(t, L, width, height, scratch, scratch_floats, sort_keys, async_tail, accumulate, g_cubemap, g_fail, stream);
	if (rc < 0) return rc;
	// the staging buffer is zeroed here; the sort's look-back state by the pixel kernel (no forward keys: the sort follows it) or by a fill
	// in front of the sort (forward keys); keys_sorted: the forward call has sorted (or is still sorting, on the side stream) into this
	// scratch: only the staging buffer is touched
	if (sort_keys && !keys_sorted) rc = refl_tail_clear(t);
	else if (hipMemsetAsync(t.clear_ptr[0], 0, t.clear_bytes[0], stream) != hipSuccess) { refl_tail_abort(t); set_error("hipMemsetAsync failed"); return GSR_E_HIP; }
	if (rc < 0) return rc;
	rc = keys_sorted ? refl_tail_join_early_sort(t) : refl_tail_sort(t);
	if (rc < 0) return rc;
	{
		StageTimer st_(GSR_STAGE_REFL_BWD, stream);      // the pixel kernel; the texel-gradient tail is GSR_STAGE_REFL_BWD_TAIL
		const unsigned egrid = (unsigned)((HW + ENTRIES_BS - 1) / ENTRIES_BS);
		ReflFootprint* fp = static_cast<ReflFootprint*>(t.footprints);
		if (cubemap_rgba && ((uintptr_t)cubemap_rgba & 15) == 0)     // the texel-interleaved copy the forward made (same cubemap)
			deferred_refl_bwd_entries_kernel<true><<<egrid, ENTRIES_BS, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap,
			                                                                 reinterpret_cast<const float4*>(cubemap_rgba), fail_value, (int)L, width, height, g_final,
			                                                                 g_refl_color, g_normal_world, g_normal_view, g_base, g_strength, t.fail_acc, t.staging, fp,
			                                                                 t.keys_in, sort_keys, (uint32_t)ntex, sort_keys ? nullptr : t.sort_temp,
			                                                                 sort_keys ? 0 : t.clear_bytes[1]);
		else
			deferred_refl_bwd_entries_kernel<false><<<egrid, ENTRIES_BS, 0, stream>>>(normal_view, base_color, refl_strength, cam, cubemap, nullptr, fail_value, (int)L, width,
			                                                                  height, g_final, g_refl_color, g_normal_world, g_normal_view, g_base, g_strength, t.fail_acc,
			                                                                  t.staging, fp, t.keys_in, sort_keys, (uint32_t)ntex, sort_keys ? nullptr : t.sort_temp,
			                                                                  sort_keys ? 0 : t.clear_bytes[1]);
	}
	// the per-pixel gradients are complete here; what follows only produces dL_dcubemap / dL_dfail
	return refl_tail_finish(t);
}

/* the round-2 signature of gsr_deferred_reflection_backward_ex (no forward keys): kept so that a caller built against the earlier header
 * keeps working; gsr_deferred_reflection_backward_keys is the full form */
extern "C" int gsr_deferred_reflection_backward_ex(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                   const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                   const float* g_final, const float* g_refl_color, const float* g_normal_world,
                                                   float* g_normal_view, float* g_base, float* g_strength, float* g_cubemap, float* g_fail,
                                                   float* scratch, size_t scratch_floats, int accumulate, int async_tail, const float* cubemap_rgba,
                                                   void* stream_) {
	return gsr_deferred_reflection_backward_keys(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, g_final, g_refl_color,
	                                             g_normal_world, g_normal_view, g_base, g_strength, g_cubemap, g_fail, scratch, scratch_floats, accumulate,
	                                             async_tail, cubemap_rgba, nullptr, 0, stream_);
}

extern "C" int gsr_deferred_reflection_backward_accum(const float* normal_view, const float* base_color, const float* refl_strength, const float* cam,
                                                const float* cubemap, const float* fail_value, uint32_t L, int width, int height,
                                                const float* g_final, const float* g_refl_color, const float* g_normal_world,
                                                float* g_normal_view, float* g_base, float* g_strength, float* g_cubemap, float* g_fail,
                                                float* scratch, size_t scratch_floats, int accumulate, void* stream_) {
	return gsr_deferred_reflection_backward_keys(normal_view, base_color, refl_strength, cam, cubemap, fail_value, L, width, height, g_final, g_refl_color,
	                                             g_normal_world, g_normal_view, g_base, g_strength, g_cubemap, g_fail, scratch, scratch_floats, accumulate, 0, nullptr,
	                                             nullptr, 0, stream_);
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
