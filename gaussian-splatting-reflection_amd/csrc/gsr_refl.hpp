// Per-pixel device code of the cubemap lookup and of the deferred-reflection pass, shared by the stand-alone pixel kernels
// (gsr_cubemap.hip) and by the tile kernels of the surfel rasterizer, which run the same code as their epilogue / prologue
// (gsr_surfel.hip: the fused rasterize + reflect path, round 4).  Behaviour follows submodules/cubemapencoder/src/cubemapencoder.cu
// (CME, LEFT_TOP_AS_ORIGIN branch) and gaussian_renderer/__init__.py:22-35,148,178-179,197-199 + utils/general_utils.py:177-197.
#pragma once
#include "gsr_internal.hpp"

namespace gsr {

// Reciprocals and square roots of this per-pixel code.  Since round 4 it also runs inside the rasterizer's tile kernels, which are bound by
// the NUMBER of vector instructions they issue: an IEEE-rounded fp32 division costs ~10 instructions (v_div_scale x2, v_rcp, 4 FMAs,
// v_div_fmas, v_div_fixup) and the reflection code had sixteen of them per pixel plus two IEEE square roots.  GSR_REFL_FAST = 1: the
// 1-ulp hardware forms (v_rcp_f32, v_rsq_f32, v_sqrt_f32) — the results move by ~1e-7 relative, far inside what the float64 chain of the
// tests allows (2e-5 absolute on the final colour).  0: IEEE division / sqrt as in rounds 1-3.
#ifndef GSR_REFL_FAST
#define GSR_REFL_FAST 1
#endif
__device__ __forceinline__ float refl_rcp(float x) {
#if GSR_REFL_FAST
	return __builtin_amdgcn_rcpf(x);
#else
	return 1.0f / x;
#endif
}
__device__ __forceinline__ float refl_sqrt(float x) {
#if GSR_REFL_FAST
	return __builtin_amdgcn_sqrtf(x);
#else
	return sqrtf(x);
#endif
}

// ----------------------------------------------------------------------------------------------
// Face / uv selection (CME cubemapencoder.cu:147-187)
__device__ __forceinline__ void cube_uv(float x, float y, float z, float& u, float& v, int& index) {
	int max_dim = 0;
	const float x_ = fabsf(x), y_ = fabsf(y), z_ = fabsf(z);
	float max_v = x_;
	if (y_ > max_v) { max_v = y_; max_dim = 1; }
	if (z_ > max_v) { max_v = z_; max_dim = 2; }
#if GSR_REFL_FAST
	// the same table without branches: one reciprocal of the major component m, numerators and sign flips picked by selects
	//   dim 0: u = z/x, v = y/x   dim 1: u = x/y, v = z/y   dim 2: u = x/z, v = y/z;   faces 0, 3: -u, -v   face 1: -u   face 4: -v
	const float m = max_dim == 0 ? x : (max_dim == 1 ? y : z);
	const float a = max_dim == 0 ? z : x, b = max_dim == 1 ? z : y;
	const float r = refl_rcp(m);
	index = 2 * max_dim + (m >= 0.f ? 0 : 1);
	u = a * r; v = b * r;
	if (index == 0 || index == 1 || index == 3) u = -u;
	if (index == 0 || index == 3 || index == 4) v = -v;
#else
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
#endif
}

// Neighbour-face texel across a cube edge (CME cubemapencoder.cu:66-106), as a small table:
// for (face, flag in {1,2,4,8}) -> new face and how (x', y') derive from (L-1, 0, x, y, L-1-x, L-1-y).
// Source selectors: 0 -> 0, 1 -> L-1, 2 -> x, 3 -> y, 4 -> L-1-x, 5 -> L-1-y.
// Branch-free (round 4: this code also runs inside the tile kernels, where a `switch` per selector had become ~60 divergent
// branches with their exec-mask bookkeeping — scalar instructions a wave cannot overlap with its vector ones): selects only.
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
	const int ix = x, iy = y, Lm = L - 1;
	auto sel = [&](unsigned s) -> int {
		int r = 0;
		r = s == 1u ? Lm : r;
		r = s == 2u ? ix : r;
		r = s == 3u ? iy : r;
		r = s == 4u ? Lm - ix : r;
		r = s == 5u ? Lm - iy : r;
		return r;
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
// Compute_Seamless_Index (CME cubemapencoder.cu:189-263).  The common case — the 2x2 footprint inside one face, all but ~2/L of the
// directions — is straight-line; footprints that reach over an edge or a vertex of the cube take ONE divergent region with three
// branch-free edge look-ups (which of its results are used is a matter of selects).
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
	// inside the face: (x0, y0) (x1, y0) (x0, y1) (x1, y1)
#pragma unroll
	for (int i = 0; i < 4; i++) s.f[i] = index;
	s.x[0] = ux_0; s.y[0] = uy_0;
	s.x[1] = ux_1; s.y[1] = uy_0;
	s.x[2] = ux_0; s.y[2] = uy_1;
	s.x[3] = ux_1; s.y[3] = uy_1;
	if (flag != 0) {
		const int fu = flag & 3, fv = flag & 12;
		const bool hu = fu != 0, hv = fv != 0;
		s.is_vertex = hu && hv;
		// texel 1 (u neighbour): across the u edge from (x0, y0) when the footprint leaves the face in u
		int f1 = index, x1 = ux_0, y1 = uy_0;
		edge_table(L, hu ? fu : 1, f1, x1, y1);
		// texel 2 (v neighbour): across the v edge from (x0, y0)
		int f2 = index, x2 = ux_0, y2 = uy_0;
		edge_table(L, hv ? fv : 4, f2, x2, y2);
		// texel 3: u only -> across the u edge from (x0, y1); v only -> across the v edge from (x1, y0); a vertex has no fourth texel
		int f3 = index, x3 = hu ? ux_0 : ux_1, y3 = hu ? uy_1 : uy_0;
		edge_table(L, hu ? fu : fv, f3, x3, y3);
		if (hu) { s.f[1] = f1; s.x[1] = x1; s.y[1] = y1; }
		if (hv) { s.f[2] = f2; s.x[2] = x2; s.y[2] = y2; }
		if (s.is_vertex) { s.f[3] = index; s.x[3] = ux_0; s.y[3] = uy_0; }      // (never read: the fourth sample is the mean of the other three)
		else { s.f[3] = f3; s.x[3] = x3; s.y[3] = y3; }
	}
	s.kx = kx; s.ky = ky; s.flag = flag;
}

// Compute_Cubemap_UV_Backward (CME cubemapencoder.cu:265-292); (gu, gv) are modified as there.
__device__ __forceinline__ void cube_uv_backward(int index, float x, float y, float z, float gu, float gv, float& gx, float& gy, float& gz) {
	const int face = index / 2;
#if GSR_REFL_FAST
	// u = s_u a / m, v = s_v b / m (see cube_uv): d/da = s_u gu / m, d/db = s_v gv / m, d/dm = -(a s_u gu + b s_v gv) / m^2
	if (index == 0 || index == 1 || index == 3) gu = -gu;
	if (index == 0 || index == 3 || index == 4) gv = -gv;
	const float m = face == 0 ? x : (face == 1 ? y : z);
	const float a = face == 0 ? z : x, b = face == 1 ? z : y;
	const float r = refl_rcp(m);
	const float ga = r * gu, gb = r * gv, gm = -(a * gu + b * gv) * (r * r);
	gx = face == 0 ? gm : ga;
	gy = face == 1 ? gm : gb;
	gz = face == 2 ? gm : (face == 0 ? ga : gb);
#else
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
#endif
}

__device__ __forceinline__ size_t texel(int f, int c, int y, int x, int C, int L) { return (((size_t)f * C + c) * L + y) * L + x; }

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
	o.len = refl_sqrt(o.nwx * o.nwx + o.nwy * o.nwy + o.nwz * o.nwz);
	const float inv = refl_rcp(o.len + 1e-6f);   // :179
	o.nx = o.nwx * inv; o.ny = o.nwy * inv; o.nz = o.nwz * inv;
	// utils/general_utils.py:186-196
	const float x = (float)px, y = (float)py;
	const float pcx = cam[9] * x + cam[10] * y + cam[11] - cam[27];
	const float pcy = cam[12] * x + cam[13] * y + cam[14] - cam[28];
	const float pcz = cam[15] * x + cam[16] * y + cam[17] - cam[29];
	float wx = pcx * cam[18] + pcy * cam[21] + pcz * cam[24] - cam[30];
	float wy = pcx * cam[19] + pcy * cam[22] + pcz * cam[25] - cam[31];
	float wz = pcx * cam[20] + pcy * cam[23] + pcz * cam[26] - cam[32];
#if GSR_REFL_FAST
	const float idl = __builtin_amdgcn_rsqf(wx * wx + wy * wy + wz * wz);
	o.dx = wx * idl; o.dy = wy * idl; o.dz = wz * idl;
#else
	const float dl = sqrtf(wx * wx + wy * wy + wz * wz);
	o.dx = wx / dl; o.dy = wy / dl; o.dz = wz / dl;
#endif
	o.dn = o.dx * o.nx + o.dy * o.ny + o.dz * o.nz;
	o.rx = o.dx - 2 * o.nx * o.dn;   // gaussian_renderer/__init__.py:22-24
	o.ry = o.dy - 2 * o.ny * o.dn;
	o.rz = o.dz - 2 * o.nz * o.dn;
}
__device__ __forceinline__ float sigmoidf_(float x) { return refl_rcp(1.0f + __expf(-x)); }

// corner k of the seamless lookup, all three channels (the fourth corner of a cube vertex is the mean of the other three)
template <bool RGBA>
__device__ __forceinline__ void fetch_corners(const Seamless& s, int L, const float* __restrict__ cubemap, const float4* __restrict__ rgba, float (&v)[4][3]) {
	// (a cube vertex has three texels; its slot 3 points at a valid texel whose value is replaced below: four unconditional gathers)
#pragma unroll
	for (int k = 0; k < 4; k++) {
		{
			if (RGBA) {
				const float4 t = rgba[(uint32_t)((s.f[k] * L + s.y[k]) * L + s.x[k])];    // (32-bit texel index: 6 L^2 < 2^32 is checked by the hosts)
				v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z;
			} else {
#pragma unroll
				for (int c = 0; c < 3; c++) v[k][c] = cubemap[texel(s.f[k], c, s.y[k], s.x[k], 3, L)];
			}
		}
	}
	if (s.is_vertex) {
#pragma unroll
		for (int c = 0; c < 3; c++) v[3][c] = GSR_REFL_FAST ? (v[0][c] + v[1][c] + v[2][c]) * 0.33333334f : (v[0][c] + v[1][c] + v[2][c]) / 3.f;
	}
}


struct alignas(32) ReflFootprint {
	float g[3], kx, ky;   // 20 bytes used; padded so that a record never straddles a 32-byte sector when it is gathered
	float pad[3];
};

// ---------------------------------------------------------------------------------------------- one pixel of the deferred-reflection pass
// Forward (gaussian_renderer/__init__.py:22-35,148,178-179,197-199 of the reference): shading normal -> camera ray -> reflect ->
// seamless bilinear cubemap lookup -> sigmoid -> lerp with the base colour by the blended reflection strength.  `key`: the sort key of
// the record the sorted-footprint backward will make for this pixel (the texel under the upper-left corner of its bilinear footprint when
// the 2x2 footprint lies inside one cube face, otherwise no_key); it depends on forward data only.
struct ReflFwdOut {
	float final_c[3], refl_c[3], nx, ny, nz;
	uint32_t key;
};
template <bool RGBA>
__device__ __forceinline__ void refl_forward_pixel(const float* __restrict__ cam, const float* __restrict__ cubemap, const float4* __restrict__ rgba,
                                                   const float* __restrict__ fail_value, int L, float nvx, float nvy, float nvz, int px, int py, float sv,
                                                   float b0, float b1, float b2, uint32_t no_key, ReflFwdOut& out) {
	out.key = no_key;
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
		if (s.flag == 0 && s.x[3] == s.x[0] + 1 && s.y[3] == s.y[0] + 1) out.key = (uint32_t)(((size_t)s.f[0] * L + s.y[0]) * L + s.x[0]);
	}
	const float bs[3] = {b0, b1, b2};
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		const float rc = sigmoidf_(c[ch]);
		out.refl_c[ch] = rc;
		out.final_c[ch] = (1 - sv) * bs[ch] + sv * rc;
	}
	out.nx = o.nx; out.ny = o.ny; out.nz = o.nz;
}

// Backward of one pixel on the sorted-footprint path (see gsr_deferred_reflection_backward in gsr_cubemap.hip; the body of
// deferred_refl_bwd_entries_kernel).  A pixel whose bilinear
// footprint lies inside one cube face (all but the half-texel rim, ~2/L of the pixels) leaves ONE record {g_r, g_g, g_b, kx, ky} in
// `footprint` (keyed by the texel of its upper-left corner: the other corners are t+1, t+L, t+L+1 and the four weights follow from
// (kx, ky)); rim pixels and cube vertices add their corners to the staging buffer `g_scratch` ([6][L][L][4], channel-interleaved)
// directly, the zero reflection vector adds to g_fail.  Returns the per-pixel gradients in registers.
// WAVE-COOPERATIVE: the rim pixels of a wave (about one per wave) are served one at a time by lanes 0..11 (corner texels and weights
// travel through SGPRs; one atomic instruction whose twelve dwords fall into four 16-byte slots), so the function must be called by all
// 64 lanes of the wave, `live` = false for lanes without a pixel (their outputs are unspecified and nothing of theirs is stored).
//   keys_fwd_valid / kf: the forward already wrote this pixel's sort key (same arithmetic on the same inputs) and the sort may be running
//   beside this code: its key decides whether the pixel has a record; should forward and backward ever disagree, the pixel goes through
//   the rim path and its record, which the sort expects, is zeros.  Without it `key_out` receives the key.
struct ReflBwdIn {
	float nvx, nvy, nvz, sv;
	float gfin[3], bas[3], grc[3], gnw[3];
	bool has_grc, has_gnw;
};
struct ReflBwdOut {
	float g_base[3], g_strength, g_nv[3];
};
template <bool RGBA>
__device__ __forceinline__ void refl_backward_pixel(const float* __restrict__ cam, const float* __restrict__ cubemap, const float4* __restrict__ rgba,
                                                    const float* __restrict__ fail_value, int L, int px, int py, bool live, const ReflBwdIn& in,
                                                    float* __restrict__ g_fail, float* __restrict__ g_scratch, ReflFootprint* __restrict__ footprint,
                                                    bool keys_fwd_valid, uint32_t kf, uint32_t* __restrict__ key_out, uint32_t no_key, int lane,
                                                    ReflBwdOut& out) {
	const float sv = in.sv;
	ReflPixel o;
	refl_pixel(cam, in.nvx, in.nvy, in.nvz, px, py, o);
	const bool fail = (o.rx == 0.f && o.ry == 0.f && o.rz == 0.f);
	Seamless s;
	int face = 0;
	s.kx = 0; s.ky = 0; s.flag = 0; s.is_vertex = false;
#pragma unroll
	for (int k = 0; k < 4; k++) { s.f[k] = 0; s.x[k] = 0; s.y[k] = 0; }
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
		const float gf = in.gfin[c];
		const float b = in.bas[c];
		out.g_base[c] = (1 - sv) * gf;
		gs += gf * (rc - b);
		float gc = sv * gf;
		if (in.has_grc) gc += in.grc[c];
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
	const uint32_t t00 = (uint32_t)(((size_t)s.f[0] * L + s.y[0]) * L + s.x[0]);
	if (keys_fwd_valid) interior = interior && kf == t00;
	{
		// rim pixels (~2/L of all, about one per wave): the wave serves them one at a time (float atomics are priced per memory-side
		// request: 44 us -> 15 us per launch at C3 against twelve single-lane adds per rim pixel)
		const int k_of_lane = lane / 3, c_of_lane = lane - 3 * k_of_lane;
		// bilinear weights of the four corners (a cube vertex has three, the fourth is their mean)
		const float extra_g = s.is_vertex ? s.ky * s.kx * 0.33333334f : 0.f;
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
	if (live) {
		if (interior || kf != no_key) {
			// (kf != no_key without `interior`: forward and backward disagree about this footprint — its gradient went through the rim path
			// above and the record the sort expects reads as zeros)
			float* f = reinterpret_cast<float*>(footprint);
			*reinterpret_cast<float4*>(f) = interior ? make_float4(graw[0], graw[1], graw[2], s.kx) : make_float4(0.f, 0.f, 0.f, 0.f);
			f[4] = interior ? s.ky : 0.f;
		}
		if (!keys_fwd_valid) *key_out = interior ? t00 : no_key;
	}
	out.g_strength = gs;
	// r = d - 2 n (d.n)  ->  g_n = -2 [ (d.n) g_r + (g_r.n) d ]
	const float grn = grx * o.nx + gry * o.ny + grz * o.nz;
	float gnx = -2.f * (o.dn * grx + grn * o.dx);
	float gny = -2.f * (o.dn * gry + grn * o.dy);
	float gnz = -2.f * (o.dn * grz + grn * o.dz);
	if (in.has_gnw) { gnx += in.gnw[0]; gny += in.gnw[1]; gnz += in.gnw[2]; }
	// n = nw / (|nw| + eps): g_nw = g_n / (len+eps) - nw (nw.g_n) / (len (len+eps)^2)   (0 subgradient at len = 0)
	const float inv = refl_rcp(o.len + 1e-6f);
	float gwx = gnx * inv, gwy = gny * inv, gwz = gnz * inv;
	if (o.len > 0.f) {
		const float k = (o.nwx * gnx + o.nwy * gny + o.nwz * gnz) * inv * inv * refl_rcp(o.len);
		gwx -= o.nwx * k; gwy -= o.nwy * k; gwz -= o.nwz * k;
	}
#pragma unroll
	for (int c = 0; c < 3; c++) out.g_nv[c] = gwx * cam[c] + gwy * cam[3 + c] + gwz * cam[6 + c];
}

}  // namespace gsr
