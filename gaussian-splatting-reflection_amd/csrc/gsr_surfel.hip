// Variant S — 2D Gaussian surfels (ray-splat intersection, 8-plane auxiliary map, reflection strength,
// env-scope mask, per-Gaussian max blend weight).  MI355X-native restatement of the behaviour of
// submodules/diff-surfel-rasterization (DSR cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu),
// the rasterizer gaussian_renderer/__init__.py:14,130 of the reference calls.  Design notes in DESIGN.md.
//
// Render record (80 bytes, five float4 per Gaussian, written by preprocess, gathered by the tile kernels), in even-aligned
// pairs for packed fp32 math:  {x, y | Tu.x, Tv.x} {Tu.y, Tv.y | Tu.z, Tv.z} {Tw.x, Tw.y | Tw.z, opacity} {n.x, n.y | n.z, refl}
// {r, g | b, mask}   (Tu, Tv, Tw = rows of the 3x3 homography "transMat").
#include "gsr_internal.hpp"
#include "gsr_sort.hpp"
#include "gsr_math.hpp"
#include "gsr_refl.hpp"

namespace gsr {

#define S_REC_F4 5
#define S_ACC_F 20
#define SA_COLOR 0
#define SA_REFL 3
#define SA_NORMAL 4
#define SA_OPAC 7
#define SA_T 8
#define SA_MEAN2D 17

#define S_NEAR 0.2f
#define S_FAR 100.0f
#define S_FILTER_INV_SQ 2.0f

struct SurfelCam {
	const float* view;
	const float* proj;
	const float* campos;
	int W, H;
	float tan_fovx, tan_fovy, focal_x, focal_y;
};

// quat_to_rotmat (DSR auxiliary.h:217-239); the reference's rsqrtf is restated as an exact 1/sqrt.
__device__ __forceinline__ M3 quat_to_rotmat(const float* __restrict__ q, float& w, float& x, float& y, float& z) {
#pragma clang fp contract(off)
	const float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
	w = q[0] * s; x = q[1] * s; y = q[2] * s; z = q[3] * s;
	return m3_make(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y), 2.f * (x * y - w * z),
	               1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x), 2.f * (x * z + w * y), 2.f * (y * z - w * x),
	               1.f - 2.f * (x * x + y * y));
}

// P = world2ndc * ndc2pix as a 3-column x 4-row matrix (DSR forward.cu:99-112, backward.cu:520-533):
// P[j][k] = sum_m world2ndc[m][k] * ndc2pix[j][m], world2ndc[m][k] = proj[m + 4k].
struct P34 {
	float m[3][4];
};
__device__ __forceinline__ P34 make_P(const float* __restrict__ pm, int W, int H) {
#pragma clang fp contract(off)
	const float Wh = (float)(float(W) / 2.0), Wm = (float)(float(W - 1) / 2.0);
	const float Hh = (float)(float(H) / 2.0), Hm = (float)(float(H - 1) / 2.0);
	P34 p;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		p.m[0][k] = pm[0 + 4 * k] * Wh + pm[3 + 4 * k] * Wm;
		p.m[1][k] = pm[1 + 4 * k] * Hh + pm[3 + 4 * k] * Hm;
		p.m[2][k] = pm[3 + 4 * k];
	}
	return p;
}

// compute_transmat (DSR forward.cu:75-115): T = (splat2world^T * world2ndc) * ndc2pix, evaluated in that
// association and left-to-right order (zero terms of the 4-vectors dropped: adding an exact zero is exact).
__device__ __forceinline__ void compute_transmat(float px, float py, float pz, const float* __restrict__ scale, float mod,
                                                 const float* __restrict__ rot, const SurfelCam& cam, M3& T, F3& normal) {
#pragma clang fp contract(off)
	float qw, qx, qy, qz;
	const M3 R = quat_to_rotmat(rot, qw, qx, qy, qz);
	const float s0 = mod * scale[0], s1 = mod * scale[1];
	const float L0[3] = {R.m[0][0] * s0, R.m[0][1] * s0, R.m[0][2] * s0};
	const float L1[3] = {R.m[1][0] * s1, R.m[1][1] * s1, R.m[1][2] * s1};
	const float* pm = cam.proj;
	// AB[j][i], j = 0..3 (ndc component), i = 0..2 (u, v, 1)
	float AB[4][3];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		AB[j][0] = L0[0] * pm[j] + L0[1] * pm[j + 4] + L0[2] * pm[j + 8];
		AB[j][1] = L1[0] * pm[j] + L1[1] * pm[j + 4] + L1[2] * pm[j + 8];
		AB[j][2] = px * pm[j] + py * pm[j + 4] + pz * pm[j + 8] + pm[j + 12];
	}
	const float Wh = (float)(float(cam.W) / 2.0), Wm = (float)(float(cam.W - 1) / 2.0);
	const float Hh = (float)(float(cam.H) / 2.0), Hm = (float)(float(cam.H - 1) / 2.0);
#pragma unroll
	for (int i = 0; i < 3; i++) {
		T.m[0][i] = AB[0][i] * Wh + AB[3][i] * Wm;
		T.m[1][i] = AB[1][i] * Hh + AB[3][i] * Hm;
		T.m[2][i] = AB[3][i];
	}
	const float* vm = cam.view;
	const float nx = R.m[2][0], ny = R.m[2][1], nz = R.m[2][2];
	normal = f3(vm[0] * nx + vm[4] * ny + vm[8] * nz, vm[1] * nx + vm[5] * ny + vm[9] * nz, vm[2] * nx + vm[6] * ny + vm[10] * nz);
}

// Cull record of a surfel (see cull_hit() in gsr_internal.hpp): a superset of the pixels it can blend into.
// alpha = min(0.99, opa * exp(-rho/2)) >= 1/255 needs rho <= rho_max = 2 ln(255 opa), rho = min(rho3d, rho2d):
//   * rho3d <= c2 is the image of the splat-space disc u^2+v^2 <= c2 under the homography pix_h = A (u,v,1)^T,
//     A = rows (Tu, Tv, Tw): an ellipse as long as the disc stays in front of the camera plane.  It is extracted from the
//     DUAL conic Q* = A diag(1,1,-1/c2) A^T = [[S, s],[s^T, sigma]], which needs no inverse of A: centre = s / sigma (the
//     reference's AABB centre formula), covariance-like shape E^-1 = (S - s s^T / sigma) / (-sigma).  (A first version
//     inverted A and read centre and scale off the point conic; for surfels seen edge-on — axis ratios beyond 100:1 —
//     that lost every digit even in double, the footprint came out 10x too short and a contributing pixel was culled:
//     found by tests/test_gpu_fullsize.py.)
//   * rho2d <= c2 is the disc of squared radius c2/2 about the low-pass centre (the cutoff-3 AABB centre).
// c2 carries a 5 % + 0.1 margin over rho_max; the ellipse is grown by 2 % and then dilated by half a pixel
// (E^-1 += 0.25 I), which also bounds its condition number so that the fp32 vote stays accurate for slivers.
// Anything irregular (disc reaching the camera plane, non-positive shape, NaN) is encoded as "always a hit";
// opa < 1/255 can never blend.  Culling with this record leaves every output bit-identical
// (tests/test_gpu_parity.py::test_cull_is_bit_exact, tests/test_gpu_fullsize.py at C3).
__device__ __forceinline__ void surfel_cull_record(const M3& T, float cx, float cy, float opa, float4& c0, float4& c1) {
	c0 = make_float4(cx, cy, 0.f, 0.f);
	c1 = make_float4(0.f, cx, cy, -2.0f);
	if (!(opa >= 1.0f / 255.0f)) return;   // never
	const float c2f = 2.0f * logf(255.0f * opa) * 1.05f + 0.1f;
	c1.w = 0.5f * c2f * 1.02f;             // disc about the low-pass centre
	const double ic2 = 1.0 / (double)c2f;
	const double a00 = T.m[0][0], a01 = T.m[0][1], a02 = T.m[0][2];   // Tu
	const double a10 = T.m[1][0], a11 = T.m[1][1], a12 = T.m[1][2];   // Tv
	const double a20 = T.m[2][0], a21 = T.m[2][1], a22 = T.m[2][2];   // Tw
	// sigma < 0: the cutoff disc stays strictly in front of the camera plane (w > 0); otherwise the projection is unbounded
	const double sigma = a20 * a20 + a21 * a21 - a22 * a22 * ic2;
	if (!(sigma < -1e-6 * a22 * a22 * ic2)) return;   // a stays 0: always a hit
	const double sx = a00 * a20 + a01 * a21 - a02 * a22 * ic2, sy = a10 * a20 + a11 * a21 - a12 * a22 * ic2;
	const double Sxx = a00 * a00 + a01 * a01 - a02 * a02 * ic2, Sxy = a00 * a10 + a01 * a11 - a02 * a12 * ic2,
	             Syy = a10 * a10 + a11 * a11 - a12 * a12 * ic2;
	const double is = 1.0 / sigma;
	const double ex = sx * is, ey = sy * is;
	// E^-1 = (S - s s^T / sigma) / (-sigma), grown by 2 % in length and dilated by half a pixel
	const double g = -is * 1.0404;
	const double Vxx = (Sxx - sx * sx * is) * g + 0.25, Vxy = (Sxy - sx * sy * is) * g, Vyy = (Syy - sy * sy * is) * g + 0.25;
	const double det = Vxx * Vyy - Vxy * Vxy;
	if (!(Vxx > 0.25) || !(Vyy > 0.25) || !(det > 0.0)) return;
	const float fa = (float)(Vyy / det), fb = (float)(-Vxy / det), fc = (float)(Vxx / det);
	if (!(fa > 0.f) || !(fc > 0.f) || !(fa * fc - fb * fb > 0.f) || !(fabsf((float)ex) < 1e7f) || !(fabsf((float)ey) < 1e7f)) return;
	c0 = make_float4((float)ex, (float)ey, fa, fb);
	c1.x = fc;
}

// preprocessCUDA forward (DSR forward.cu:149-253); FMA contraction off (integer outputs bit-exact vs oracle).
// (Occupancy: 88 VGPRs = 5 waves per SIMD.  Forcing 6 / 8 with amdgpu_waves_per_eu was measured on the backward twin of this
// kernel: 0.175 -> 0.249 / 0.321 ms at C3, the registers it gives up cost more than the waves it gains.)
// One Gaussian of the pass; returns false where the reference's kernel returns early.  The render record (5 float4) and the
// cull record (2 float4) are handed back in `o` instead of being stored: the wave stores them together (see the kernel).
__device__ __forceinline__ bool
surfel_preprocess_one(int idx, int D, int M, const float* __restrict__ means, const float* __restrict__ scales, float scale_modifier,
                      const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
                      const float* __restrict__ transMat_precomp, const float* __restrict__ colors_precomp, const float* __restrict__ refl,
                      const uint8_t* __restrict__ env_scope_mask, const SurfelCam& cam, int* __restrict__ radii, const GeomState& g, int gx, int gy,
                      int prefiltered, float* __restrict__ gaussian_weights, float4* o) {
#pragma clang fp contract(off)
	radii[idx] = 0;
	gaussian_weights[idx] = 0.f;   // the tile kernel merges per-wave maxima into it with atomicMax
	g.tiles_touched[idx] = 0;
	reinterpret_cast<uint2*>(g.rect)[idx] = make_uint2(0u, 0u);   // empty tile rectangle: emit_tiles_kernel takes the instance count from its area
	g.depths[idx] = __int_as_float(0x7f7fffff);   // culled: sorts behind every visible Gaussian in the depth pre-sort
	const float mx = means[3 * idx], my = means[3 * idx + 1], mz = means[3 * idx + 2];
	const float* vm = cam.view;
	const float pvx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
	const float pvy = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
	const float pvz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
	if (pvz <= 0.2f) {
		if (prefiltered) g.flags[0] = 1;
		return false;
	}
	M3 T;
	F3 normal;
	if (transMat_precomp == nullptr) {
		compute_transmat(mx, my, mz, scales + 2 * idx, scale_modifier, rotations + 4 * idx, cam, T, normal);
	} else {
#pragma unroll
		for (int c = 0; c < 3; c++)
#pragma unroll
			for (int r = 0; r < 3; r++) T.m[c][r] = transMat_precomp[9 * idx + 3 * c + r];
		normal = f3(0.0f, 0.0f, 1.0f);
	}
	// DUAL_VISIABLE (forward.cu:211-216)
	const float cosv = -((pvx * normal.x) + (pvy * normal.y) + (pvz * normal.z));
	if (cosv == 0) return false;
	const float multiplier = cosv > 0 ? 1.f : -1.f;
	normal = f3(multiplier * normal.x, multiplier * normal.y, multiplier * normal.z);
	// compute_aabb (forward.cu:119-145), cutoff = 3
	const float cutoff = 3.0f;
	const float t0 = cutoff * cutoff, t1 = cutoff * cutoff, t2 = -1.0f;
	const float d = t0 * (T.m[2][0] * T.m[2][0]) + t1 * (T.m[2][1] * T.m[2][1]) + t2 * (T.m[2][2] * T.m[2][2]);
	if (d == 0.0f) return false;
	const float inv_d = 1 / d;
	const float f0 = inv_d * t0, f1 = inv_d * t1, f2 = inv_d * t2;
	const float pxi = f0 * (T.m[0][0] * T.m[2][0]) + f1 * (T.m[0][1] * T.m[2][1]) + f2 * (T.m[0][2] * T.m[2][2]);
	const float pyi = f0 * (T.m[1][0] * T.m[2][0]) + f1 * (T.m[1][1] * T.m[2][1]) + f2 * (T.m[1][2] * T.m[2][2]);
	const float h0x = pxi * pxi - (f0 * (T.m[0][0] * T.m[0][0]) + f1 * (T.m[0][1] * T.m[0][1]) + f2 * (T.m[0][2] * T.m[0][2]));
	const float h0y = pyi * pyi - (f0 * (T.m[1][0] * T.m[1][0]) + f1 * (T.m[1][1] * T.m[1][1]) + f2 * (T.m[1][2] * T.m[1][2]));
	const float ex = sqrtf(fmaxf(1e-4f, h0x)), ey = sqrtf(fmaxf(1e-4f, h0y));
	const float radius = ceilf(fmaxf(fmaxf(ex, ey), cutoff * (float)0.707106));
	uint32_t x0, y0, x1, y1;
	get_rect(pxi, pyi, f2i(radius), gx, gy, x0, y0, x1, y1);
	if ((x1 - x0) * (y1 - y0) == 0) return false;

	float cr, cg, cb;
	if (colors_precomp == nullptr) {
		const float dx = mx - cam.campos[0], dy = my - cam.campos[1], dz = mz - cam.campos[2];
		const float len = sqrtf(dx * dx + dy * dy + dz * dz);
		ShRow s;
		load_sh(shs, idx, M, (D + 1) * (D + 1), s);
		const F3 c = sh_eval(D, s, dx / len, dy / len, dz / len);
		g.clamped[idx] = (uint8_t)((c.x < 0 ? 1 : 0) | (c.y < 0 ? 2 : 0) | (c.z < 0 ? 4 : 0));
		cr = fmaxf(c.x, 0.0f); cg = fmaxf(c.y, 0.0f); cb = fmaxf(c.z, 0.0f);
	} else {
		cr = colors_precomp[3 * idx]; cg = colors_precomp[3 * idx + 1]; cb = colors_precomp[3 * idx + 2];
	}
	g.depths[idx] = pvz;
	radii[idx] = f2i(radius);
	g.rect[2 * idx] = x0 | (y0 << 16);
	g.rect[2 * idx + 1] = x1 | (y1 << 16);
	const float maskv = (env_scope_mask != nullptr && env_scope_mask[idx]) ? 1.0f : 0.0f;
	surfel_cull_record(T, pxi, pyi, opacities[idx], o[5], o[6]);
	float4* rec = o;
	// Render record, laid out in even-aligned PAIRS so that the tile kernels can feed them to packed fp32 instructions
	// (v_pk_mul/add/fma_f32 take a 64-bit aligned SGPR pair) straight from the s_load destination:
	//   {x, y | Tu.x, Tv.x} {Tu.y, Tv.y | Tu.z, Tv.z} {Tw.x, Tw.y | Tw.z, opacity} {n.x, n.y | n.z, refl} {r, g | b, mask}
	rec[0] = make_float4(pxi, pyi, T.m[0][0], T.m[1][0]);
	rec[1] = make_float4(T.m[0][1], T.m[1][1], T.m[0][2], T.m[1][2]);
	rec[2] = make_float4(T.m[2][0], T.m[2][1], T.m[2][2], opacities[idx]);
	rec[3] = make_float4(normal.x, normal.y, normal.z, refl[idx]);
	rec[4] = make_float4(cr, cg, cb, maskv);
	g.tiles_touched[idx] = (y1 - y0) * (x1 - x0);
	return true;
}

// The records are AoS (80 + 32 bytes per Gaussian) because the tile kernels fetch one Gaussian's record at a time; stored by
// the lane that computed them they are 7 store instructions of 64 x 16 bytes at an 80- or 32-byte stride — 384 of this
// kernel's ~420 write requests per wave (TCP_TCC_WRITE_REQ, profiles/r02_pmc_memory_side_before_coalesced_stores.txt), every one of them a partial
// line.  Instead each wave transposes its 64 x 7 float4 through LDS and writes 5 + 2 KB of contiguous memory.
#define S_OUT_F4 7
// Fused rasterize + reflect path: the texel-interleaved copy [6][L][L] of float4 of the cubemap that the tile kernels' reflection code
// gathers from (one 16-byte load per bilinear corner, gsr_refl.hpp) is made here, by the first kernel of the forward, instead of by a
// dispatch of its own in front of the pixel pass (cubemap_interleave_kernel, 5 us).  ntex = 0: nothing to do.
struct CubemapInterleave {
	const float* cubemap;
	float4* rgba;
	uint32_t ntex, LL;
};
__global__ void __launch_bounds__(256)
surfel_preprocess_kernel(int P, int D, int M, const float* __restrict__ means, const float* __restrict__ scales, float scale_modifier,
                         const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
                         const float* __restrict__ transMat_precomp, const float* __restrict__ colors_precomp,
                         const float* __restrict__ refl, const uint8_t* __restrict__ env_scope_mask, SurfelCam cam, int* __restrict__ radii,
                         GeomState g, int gx, int gy, int prefiltered, float* __restrict__ gaussian_weights, CubemapInterleave ci) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	for (uint32_t t = (uint32_t)idx; t < ci.ntex; t += gridDim.x * 256u) {
		const uint32_t f = t / ci.LL, r = t - f * ci.LL;
		const float* c = ci.cubemap + (size_t)f * 3 * ci.LL + r;
		ci.rgba[t] = make_float4(c[0], c[ci.LL], c[2 * (size_t)ci.LL], 0.f);
	}
	// look-back state of the depth sort that follows (gsr_sort.hpp): cleared here instead of by a dispatch of its own
	sort_clear_region(g.depth_sort_temp, g.depth_sort_clear, (size_t)idx, (size_t)gridDim.x * 256u);
	sort_clear_region(g.emit_state, g.emit_state_bytes, (size_t)idx, (size_t)gridDim.x * 256u);   // look-back state of emit_tiles_kernel's scan
	if (idx == 0) { g.flags[1] = 0; g.flags[2] = 0; g.flags[3] = 0; }   // [1] workgroup tickets, [2,3] num_rendered (64 bits) of gaussian_stats_kernel
	__shared__ float4 s_out[4][S_REC_F4 * 65];
	const int lane = threadIdx.x & 63;
	float4* so = s_out[threadIdx.x >> 6];
	float4 o[S_OUT_F4];
#pragma unroll
	for (int k = 0; k < S_OUT_F4; k++) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
	bool live = false;
	if (idx < P)
		live = surfel_preprocess_one(idx, D, M, means, scales, scale_modifier, rotations, opacities, shs, transMat_precomp, colors_precomp, refl,
		                             env_scope_mask, cam, radii, g, gx, gy, prefiltered, gaussian_weights, o);
	if (__ballot(live) == 0ull) return;      // (wave-uniform) nothing of this wave is ever read
	const int g0 = idx - lane;               // first Gaussian of the wave
	const int ng = min(64, P - g0);
	wave_store_rows4<S_REC_F4, false>(so, o, g.rec + (size_t)g0 * S_REC_F4, S_REC_F4, 0, ng, lane);
	wave_store_rows4<2, false>(so, o + S_REC_F4, g.bbox + (size_t)g0 * 2, 2, 0, ng, lane);
}

// Pairs of floats for packed fp32 math (v_pk_*_f32): the k- and l-side (or x- and y-side) of the reference's arithmetic in one register pair.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }

// The five float4 of a render record by meaning (layout: see surfel_preprocess_kernel).
struct SurfelRec {
	float4 r0, r1, r2, r3, r4;
	__device__ __forceinline__ v2f xy() const { return mk2(r0.x, r0.y); }
	__device__ __forceinline__ v2f TuvX() const { return mk2(r0.z, r0.w); }
	__device__ __forceinline__ v2f TuvY() const { return mk2(r1.x, r1.y); }
	__device__ __forceinline__ v2f TuvZ() const { return mk2(r1.z, r1.w); }
	__device__ __forceinline__ v2f Twxy() const { return mk2(r2.x, r2.y); }
	__device__ __forceinline__ float Twz() const { return r2.z; }
	__device__ __forceinline__ float opac() const { return r2.w; }
	__device__ __forceinline__ v2f nxy() const { return mk2(r3.x, r3.y); }
	__device__ __forceinline__ float nz() const { return r3.z; }
	__device__ __forceinline__ float refl() const { return r3.w; }
	__device__ __forceinline__ v2f rg() const { return mk2(r4.x, r4.y); }
	__device__ __forceinline__ float b() const { return r4.z; }
	__device__ __forceinline__ float mask() const { return r4.w; }
};

// renderCUDA forward (DSR forward.cu:258-489), wave-per-quadrant form.
// One 64-thread workgroup (= one wave) owns an 8x8 pixel block of a 16x16 tile and walks the tile's list front
// to back in batches of 64 entries: (1) every lane takes one entry, loads its conservative bounds and votes
// (ballot) whether it can reach this block -> private compacted work list; (2) the wave blends the survivors,
// pulling each 80-byte record through the scalar memory path into SGPRs (it is wave-uniform) while the
// previous one is still being blended; (3) the per-surfel max blend weight goes out with one atomic per
// touched surfel.  The wave stops as soon as ITS 64 pixels are saturated: no workgroup barriers, no waiting
// for the other three quadrants (the reference synchronises the 256 threads of a tile twice per batch).
//
// Instruction budget (tests/microbench/inst_cost.hip, MI355X, per wave64 instruction and SIMD, 4-8 waves resident): plain
// VGPR-operand VOP2/VOP3 ~1.1 ns; anything that reads an SGPR, v_min/v_max/v_cmp/v_cndmask, DPP, packed fp32 and SALU
// ~1.85 ns; v_exp/v_rcp 3.6 ns.  The pair loop is bound by that issue rate (adding resident waves changes nothing), so
// it is written to MINIMISE THE COUNT of ~1.85-ns instructions: predicates are explicit lane masks (no VGPR
// materialisation for ballots), the rare `unstable` case leaves the straight-line path through a wave-uniform branch,
// the 64-lane max for gaussian_weights is done for four pairs at a time (row_max4: 2 DPP per pair instead of 6 + 6 nops),
// the env-scope plane is an accumulated weight, and the median bookkeeping stops once no pixel has T > 0.5.
#define S_WBATCH 64
#define CULL_PAD 0.05f    // the wave's pixel block is padded by this much in the footprint vote (the cull record itself is already dilated by half a pixel)
#ifndef GSR_BWD_WPE
#define GSR_BWD_WPE 4
#endif
#ifndef GSR_BWD_LIST
#define GSR_BWD_LIST 1      // 1: rows read their next entry from an LDS list (render_bwd 0.747 -> 0.732 ms at C3); 0: rows walk their bit masks (round 2)
#endif
#ifndef GSR_BWD_UNROLL
#define GSR_BWD_UNROLL 1
#endif
#ifdef GSR_FWD_WPE
#define GSR_FWD_ATTR __attribute__((amdgpu_waves_per_eu(GSR_FWD_WPE, GSR_FWD_WPE)))
#else
#define GSR_FWD_ATTR
#endif

// Per-pixel compositing state of the forward wave.
struct SurfelFwdPix {
	float T, C2, M2, distortion, median_depth, median_contributor, maskacc;
	v2f Crg, Nxy, NzR, DM;          // accumulators paired like the record: (r, g) | (n.x, n.y) | (n.z, refl) | (depth, m)
	uint32_t last_contributor;
};

// One (wave, surfel) pair of the forward.  `done` = lanes whose pixel has retired (or lies outside the image); returns the
// lanes that blended (ok) and updates `done`.  wq receives the blend weight (0 where the pair did not contribute).
__device__ __forceinline__ lmask surfel_fwd_pair(const SurfelRec& R, const v2f pix, uint32_t contributor, bool med_live, SurfelFwdPix& st,
                                                 lmask& done, float& wq, bool& grazed) {
	float sx, sy, rho3d, depth, alpha, rho2d;
	{
#pragma clang fp contract(off)
		// Plain IEEE mul/sub in the reference's textual order (contraction off): the plane/plane cross product cancels
		// catastrophically in fp32, so the evaluation order is part of the result.
		const v2f Tw = R.Twxy();
		const v2f X = pix * Tw.x - R.TuvX();          // k.x = pix.x*Tw.x - Tu.x | l.x = pix.y*Tw.x - Tv.x
		const v2f Y = pix * Tw.y - R.TuvY();
		const v2f Z = pix * R.Twz() - R.TuvZ();
		const v2f a = Y * Z.yx;                       // (k.y*l.z, l.y*k.z)
		const v2f b = Z * X.yx;                       // (k.z*l.x, l.z*k.x)
		const v2f c = X * Y.yx;                       // (k.x*l.y, l.x*k.y)
		const float ppx = a.x - a.y, ppy = b.x - b.y, pz = c.x - c.y;   // scalar subtractions: no register shuffles
		const lmask unstable = LMASK(fabsf(pz) < 1e-4f);
		if (__builtin_expect(unstable == 0ull, 1)) {
			const float inv = div_nr(1.0f, pz);
			sx = ppx * inv; sy = ppy * inv;
			rho3d = sx * sx + sy * sy;
		} else {                                      // some lane's ray grazes the splat plane: the reference's select form
			grazed = true;                            // (the backward uses another threshold there: it must look at this pair itself)
			const float inv = div_nr(1.0f, selm(unstable, 1.0f, pz));
			sx = selm(unstable, 0.f, ppx * inv); sy = selm(unstable, 0.f, ppy * inv);
			rho3d = selm(unstable, 1e8f, sx * sx + sy * sy);
		}
		const v2f d = R.xy() - pix;
		const v2f d2 = d * d;
		rho2d = S_FILTER_INV_SQ * (d2.x + d2.y);
		const v2f st2 = mk2(sx, sy) * Tw;
		depth = (st2.x + st2.y) + R.Twz();
	}
	const float power = -0.5f * min_raw(rho3d, rho2d);
	alpha = fminf(0.99f, R.opac() * exp_neg(power));
	// reference order of the tests: depth < near, power > 0, alpha < 1/255 (DSR forward.cu:394-410).  power = -rho/2 with
	// rho >= 0 is never > 0 (and a NaN compares false both here and there), so that test has no instruction.
	const lmask live = LMASK(!(depth < S_NEAR)) & LMASK(!(alpha < 1.0f / 255.0f)) & ~done;
	const float test_T = st.T * (1 - alpha);
	const lmask sat = LMASK(test_T < 0.0001f) & live;   // this pixel is saturated: the pair is dropped and the pixel retires
	const lmask ok = live & ~sat;
	done |= sat;
	wq = 0.f;
	if (ok != 0ull) {
		const float w = selm0(ok, alpha * st.T);
		const float dep = selm(ok, depth, 1.0f);
		const float A = 1 - st.T;
		const float m = S_FAR / (S_FAR - S_NEAR) * (1 - S_NEAR * __builtin_amdgcn_rcpf(dep));
		const v2f w2 = mk2(w, w);
		st.distortion += (m * m * A + st.M2 - 2 * m * st.DM.y) * w;
		st.M2 += m * (m * w);
		st.DM = __builtin_elementwise_fma(mk2(dep, m), w2, st.DM);        // depth sum, M1
		if (med_live) {                                                  // median depth: last pair seen while T > 0.5
			const lmask med = ok & LMASK(st.T > 0.5f);
			st.median_depth = selm(med, dep, st.median_depth);
			st.median_contributor = selm(med, (float)contributor, st.median_contributor);
		}
		st.Nxy = __builtin_elementwise_fma(R.nxy(), w2, st.Nxy);
		st.NzR = __builtin_elementwise_fma(mk2(R.nz(), R.refl()), w2, st.NzR);
		st.Crg = __builtin_elementwise_fma(R.rg(), w2, st.Crg);
		st.C2 = fmaf(R.b(), w, st.C2);
		st.maskacc = fmaf(R.mask(), w, st.maskacc);   // > 0 iff a contributor inside the env scope blended (w > 0 for every ok pair)
		st.T = selm(ok, test_T, st.T);
		st.last_contributor = selmu(ok, contributor, st.last_contributor);
		wq = w;
	}
	return ok;
}

// Arguments of the deferred-reflection epilogue of the forward tile kernel (fused rasterize + reflect path, REFL = true): a wave that has
// finished its 8x8 pixel block holds the blended view-space normal, the base colour and the reflection strength of its pixels in
// registers and runs refl_forward_pixel (gsr_refl.hpp: the body of deferred_refl_fwd_kernel) on them before it retires.  The pixel pass
// of its own read those seven planes back (28 B per pixel) in a 40-us HBM-bound kernel; here its ~150 instructions per pixel disappear in a
// VALU-bound kernel of 290 M wave-instructions (+ 5 M) and its four 16-byte texel gathers hide behind the other waves' arithmetic.
struct SurfelReflFwd {
	const float* cam;
	const float* cubemap;
	const float4* rgba;
	const float* fail_value;
	int L;
	uint32_t no_key;
	float* out_final;
	float* out_refl_color;
	float* out_nworld;
	uint32_t* sort_keys;
};
template <bool REFL>
__global__ void __launch_bounds__(64) GSR_FWD_ATTR
surfel_render_fwd_wave_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list, int W, int H, int tiles_x, int ntiles,
                              const float4* __restrict__ rec, const float4* __restrict__ bbox, int cull, const float* __restrict__ bg,
                              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                              float* __restrict__ out_others, float* __restrict__ out_refl, float* __restrict__ gaussian_weights,
                              unsigned long long* __restrict__ blend_mask, size_t mask_stride, SurfelReflFwd rf) {
	const uint32_t slot = xcd_slot(blockIdx.x);   // dispatch slot -> (tile, quadrant), longest lists first
	if (slot >= (uint32_t)ntiles * 4u) return;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_order[slot >> 2]), quad = slot & 3u;   // (readfirstlane: the compiler cannot see that the loaded tile id is wave-uniform)
	const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
	const int lane = threadIdx.x;
	const int bx0 = tile_x * 16 + (quad & 1) * 8, by0 = tile_y * 16 + (quad >> 1) * 8;
	if (bx0 >= W || by0 >= H) return;
	// a 16-lane row (the unit of the DPP reductions) is a 4x4 pixel sub-block: the backward walks one list per sub-block
	const int px = bx0 + sub_px(lane), py = by0 + sub_py(lane);
	const bool inside = px < W && py < H;
	const v2f pix = mk2((float)px, (float)py);
	const uint2 range = ranges[tile];
	const int count = (int)(range.y - range.x);
	const float qx0 = (float)bx0, qy0 = (float)by0, qx1 = qx0 + 7.0f, qy1 = qy0 + 7.0f;

	__shared__ uint32_t s_hid[S_WBATCH];
	__shared__ uint32_t s_hc[S_WBATCH];
	__shared__ float4 s_wmax[S_WBATCH];            // [hit][16-lane row = 4x4 sub-block]: row maxima of the blend weight

	lmask done = ~LMASK(px < W) | ~LMASK(py < H);  // lanes outside the image never blend
	SurfelFwdPix st;
	st.T = 1.0f;
	st.C2 = st.M2 = st.distortion = st.median_depth = st.maskacc = 0.f;
	st.median_contributor = -1.0f;
	st.Crg = st.Nxy = st.NzR = st.DM = mk2(0.f, 0.f);
	st.last_contributor = 0;
	bool med_live = true;
	const size_t batch0 = (size_t)(range.x / S_WBATCH) + tile;     // where this tile's batches sit in blend_mask (see BinningState)
	// (hit ordinal + value slot of this lane's quad) * 16 bytes + row * 4: where the lane parks a row maximum (row_max4)
	const uint32_t wmax_off = (uint32_t)row_reduce_slot(lane) * 16u + (uint32_t)(lane >> 4) * 4u;

	for (int base = 0; base < count; base += S_WBATCH) {
		if (done == ~0ull) break;
		const int nb = min(S_WBATCH, count - base);
		// ---- 1. vote
		bool hit = lane < nb;
		uint32_t id = 0;
		if (hit) {
			id = point_list[range.x + (uint32_t)(base + lane)];
			if (cull) {
				hit = cull_hit(bbox[2 * id], bbox[2 * id + 1], qx0 - CULL_PAD, qx1 + CULL_PAD, qy0 - CULL_PAD, qy1 + CULL_PAD);
			}
		}
		const unsigned long long mm = __ballot(hit);
		const int nh = __popcll(mm);
		if (nh == 0) continue;
		s_wmax[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
		const int kown = __popcll(mm & ((1ull << lane) - 1ull));   // this lane's entry is hit number kown (if it is a hit)
		if (hit) {
			s_hid[kown] = id;
			s_hc[kown] = (uint32_t)(base + lane + 1);  // the pair's contributor number (1-based position in the tile's list)
		}
		__syncthreads();
		const uint32_t hid = lane < nh ? s_hid[lane] : 0u;
		const uint32_t hc = lane < nh ? s_hc[lane] : 0u;
		// ---- 2. blend.  Two SGPR record buffers ping-pong (as in the backward): the s_load of the next record is issued
		// right after the arithmetic of the current one has started; with a single rotating buffer the compiler copies the 20
		// SGPRs twice per pair.  Four pairs per trip: their blend weights share one packed row-max reduction.
		using Rec = SurfelRec;
		auto fetch = [&](int k) -> Rec {
			const float4* q = rec + (size_t)__builtin_amdgcn_readlane(hid, k) * S_REC_F4;
			return Rec{q[0], q[1], q[2], q[3], q[4]};
		};
		Rec A = fetch(0), B = A;
		unsigned long long force = 0ull;     // hits whose pair had a grazing lane (rare): the backward must evaluate them itself
		for (int k = 0; k < nh; k += 4) {
			float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
			lmask any = 0ull;
			bool stop, grazed = false;
			{
				const uint32_t c = __builtin_amdgcn_readlane(hc, k);
				if (k + 1 < nh) B = fetch(k + 1);
				any |= surfel_fwd_pair(A, pix, c, med_live, st, done, w0, grazed);
				stop = done == ~0ull || k + 1 >= nh;
			}
			if (!stop) {
				const uint32_t c = __builtin_amdgcn_readlane(hc, k + 1);
				if (k + 2 < nh) A = fetch(k + 2);
				any |= surfel_fwd_pair(B, pix, c, med_live, st, done, w1, grazed);
				stop = done == ~0ull || k + 2 >= nh;
			}
			if (!stop) {
				const uint32_t c = __builtin_amdgcn_readlane(hc, k + 2);
				if (k + 3 < nh) B = fetch(k + 3);
				any |= surfel_fwd_pair(A, pix, c, med_live, st, done, w2, grazed);
				stop = done == ~0ull || k + 3 >= nh;
			}
			if (!stop) {
				const uint32_t c = __builtin_amdgcn_readlane(hc, k + 3);
				if (k + 4 < nh) A = fetch(k + 4);
				any |= surfel_fwd_pair(B, pix, c, med_live, st, done, w3, grazed);
				stop = done == ~0ull;
			}
			if (grazed) force |= 0xFull << k;    // (the whole group of four: which of them grazed is not worth tracking)
			if (any != 0ull) {
				// gaussian_weights (forward.cu:458-459): row maxima of the four weights; the rows are merged in step 3
				const float z = row_max4(w0, w1, w2, w3);
				reinterpret_cast<float*>(s_wmax)[(k * 16 + wmax_off) >> 2] = z;   // the four lanes of a quad store the same value
				med_live = med_live && LMASK(st.T > 0.5f) != 0ull;
			}
			if (stop) break;
		}
		__syncthreads();
		// ---- 3. per entry (each lane looks at ITS entry of the batch again): merge the four row maxima of its blend weight.
		// w > 0 always, so the IEEE bit pattern orders like a signed int; the reference's check-then-atomicExch is racy,
		// this is the true maximum.  The entries that blended anywhere in a 4x4 sub-block (its row maximum is > 0) form that
		// sub-block's blend mask of the batch: the backward tile kernel walks exactly those (no vote, no pairs that cannot
		// contribute).
		float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
		if (hit) {
			r = s_wmax[kown];
			const float m = fmaxf(fmaxf(r.x, r.y), fmaxf(r.z, r.w));
			// (measured, round 3: leaving the atomic out altogether gains 0.015 ms of 0.50; reading the running maximum first and skipping
			// attempts that cannot raise it is slower, 0.58 ms)
			if (m > 0.f) atomicMax(reinterpret_cast<int*>(gaussian_weights) + id, __float_as_int(m));
		}
		const lmask forced = __ballot(hit && ((force >> kown) & 1ull) != 0ull);
		const lmask b0 = LMASK(r.x > 0.f) | forced, b1 = LMASK(r.y > 0.f) | forced, b2 = LMASK(r.z > 0.f) | forced, b3 = LMASK(r.w > 0.f) | forced;
		if (lane < 4) {
			const lmask mine = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
			blend_mask[(batch0 + (size_t)(base / S_WBATCH)) * 16u + quad * 4u + (uint32_t)lane] = mine;
		}
		__syncthreads();
	}
	if (inside) {
		const size_t HW = (size_t)H * W;
		const size_t p = (size_t)W * py + px;
		const float T = st.T;
		final_T[p] = T;
		final_T[HW + p] = st.DM.y;
		final_T[2 * HW + p] = st.M2;
		n_contrib[p] = st.last_contributor;
		// the reference converts the float -1 of "no median" with cvt.rzi.u32.f32, which saturates to 0; in C++ that
		// conversion is undefined (and clang does exploit it), so the clamp is explicit
		n_contrib[HW + p] = (uint32_t)fmaxf(st.median_contributor, 0.0f);
		const float c0 = st.Crg.x + T * bg[0], c1 = st.Crg.y + T * bg[1], c2 = st.C2 + T * bg[2];
		out_color[p] = c0;
		out_color[HW + p] = c1;
		out_color[2 * HW + p] = c2;
		out_refl[p] = st.NzR.y;
		if (REFL) {
			ReflFwdOut ro;
			refl_forward_pixel<true>(rf.cam, rf.cubemap, rf.rgba, rf.fail_value, rf.L, st.Nxy.x, st.Nxy.y, st.NzR.x, px, py, st.NzR.y, c0, c1, c2, rf.no_key, ro);
			if (rf.sort_keys) rf.sort_keys[p] = ro.key;
#pragma unroll
			for (int ch = 0; ch < 3; ch++) {
				rf.out_final[ch * HW + p] = ro.final_c[ch];
				rf.out_refl_color[ch * HW + p] = ro.refl_c[ch];
			}
			rf.out_nworld[p] = ro.nx;
			rf.out_nworld[HW + p] = ro.ny;
			rf.out_nworld[2 * HW + p] = ro.nz;
		}
		out_others[0 * HW + p] = st.DM.x;
		out_others[1 * HW + p] = 1 - T;
		out_others[2 * HW + p] = st.Nxy.x;
		out_others[3 * HW + p] = st.Nxy.y;
		out_others[4 * HW + p] = st.NzR.x;
		out_others[5 * HW + p] = st.median_depth;
		out_others[6 * HW + p] = st.distortion;
		out_others[7 * HW + p] = st.maskacc > 0.f ? 1.0f : 0.f;
	}
}

// ---------------------------------------------------------------------------------------------------
// renderCUDA backward (DSR backward.cu:143-470).
// Per-pixel recursion state of the back-to-front traversal + this pixel's upstream gradients.
struct SurfelBwdPix {
	float T, T_final, last_dL_dT, bg_dot_dpixel;
	int last_contributor, median_contributor;
	v2f dp01, dn01, dnzr;          // upstream gradients paired like the record: (r, g), (n.x, n.y), (n.z, refl)
	v2f pixs;                      // (-pix.y, pix.x)
	float dp2, dL_ddepth, dL_daccum, dL_dmedian_depth;
	float FD2r, FAr, FDr;          // final_D2, final_A, final_D times dL_dreg
	float A, Dprev, last_alpha;    // blended <attributes, upstream grads> behind this pixel's current position
};
__device__ __forceinline__ void surfel_bwd_init(SurfelBwdPix& s, bool inside, size_t pix, size_t HW, const float* __restrict__ bg,
                                                const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                                                const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths,
                                                const float* __restrict__ dL_drefl_map, const float* __restrict__ dL_dnormal_extra) {
	s.T_final = inside ? final_Ts[pix] : 0.f;
	s.T = s.T_final;
	s.last_contributor = inside ? (int)n_contrib[pix] : 0;
	s.median_contributor = inside ? (int)n_contrib[HW + pix] : 0;
	s.dp01 = s.dn01 = s.dnzr = mk2(0.f, 0.f);
	s.dp2 = s.dL_ddepth = s.dL_daccum = s.dL_dmedian_depth = 0.f;
	float dL_dreg = 0.f;
	if (inside) {
		s.dp01 = mk2(dL_dpixels[pix], dL_dpixels[HW + pix]);
		s.dp2 = dL_dpixels[2 * HW + pix];
		s.dL_ddepth = dL_depths[0 * HW + pix];
		s.dL_daccum = dL_depths[1 * HW + pix];
		s.dn01 = mk2(dL_depths[2 * HW + pix], dL_depths[3 * HW + pix]);
		s.dnzr = mk2(dL_depths[4 * HW + pix], dL_drefl_map[pix]);
		s.dL_dmedian_depth = dL_depths[5 * HW + pix];
		dL_dreg = dL_depths[6 * HW + pix];
		if (dL_dnormal_extra) {   // a second upstream gradient of the normal planes 2..4 (gsr_surfel_backward_ex), added here instead of by a pass over the image
			s.dn01 += mk2(dL_dnormal_extra[pix], dL_dnormal_extra[HW + pix]);
			s.dnzr.x += dL_dnormal_extra[2 * HW + pix];
		}
	}
	s.FDr = (inside ? final_Ts[HW + pix] : 0.f) * dL_dreg;
	s.FD2r = (inside ? final_Ts[2 * HW + pix] : 0.f) * dL_dreg;
	s.FAr = (1 - s.T_final) * dL_dreg;
	s.last_dL_dT = 0;
	s.A = s.Dprev = s.last_alpha = 0.f;
	s.bg_dot_dpixel = bg[0] * s.dp01.x + bg[1] * s.dp01.y + bg[2] * s.dp2;
}
// One iteration of the back-to-front recursion (DSR backward.cu:292-467) — each lane against the surfel R of ITS 16-lane row:
// ray-splat intersection with the backward's `unstable` threshold (1e-6; the forward uses 1e-4 — a quirk of the reference,
// reproduced), then the per-pixel state advance and the 19 gradient contributions (slots SA_*) into v.  Returns the lanes
// that contribute.  Straight-line for all 64 lanes and no early exit (the caller's lists are exact: an iteration without a
// contributing lane is a rare forced entry, and then v is all zeros), predicates as lane masks (see surfel_fwd_pair):
//   * a lane whose pair does not contribute runs with alpha = 0, which is the identity of every recurrence below
//     (T/(1-0), blend weights 0), and has its two root gradients zeroed, so all of v comes out 0 without exec-mask regions
//     or a zero-fill of v; depth and s are sanitised there too (they multiply those zeros and feed the recurrences);
//   * the reference keeps, per output channel x, accum_rec_x = last_alpha*last_x + (1-last_alpha)*accum_rec_x and adds
//     (x - accum_rec_x)*dL_dx to dL_dalpha; only the dot product over channels is ever used, and the recurrence is
//     linear, so ONE scalar recurrence A over D = <attributes, upstream grads> replaces the nine (same value up
//     to summation order);
//   * 1/(1-alpha), 1/depth and 1/p.z are formed once (Newton-refined v_rcp) and multiplied through;
//   * the cross products dL_dk = l x dL_dp, dL_dl = dL_dp x k are formed on the pairs X = (k.x, l.x), Y, Z in their
//     natural order, which yields the results SWAPPED, (-nl, nk) per component; the consumers index accordingly (no
//     register shuffles).
__device__ __forceinline__ lmask surfel_bwd_pair(SurfelBwdPix& s, const SurfelRec& R, const v2f pix, int contributor, lmask inside_m, float* v) {
	v2f X, Y, Z, d;
	float sx, sy, rho3d, rho2d, depth, inv_pz;
	{
#pragma clang fp contract(off)
		const v2f Tw = R.Twxy();
		X = pix * Tw.x - R.TuvX();
		Y = pix * Tw.y - R.TuvY();
		Z = pix * R.Twz() - R.TuvZ();
		const v2f a = Y * Z.yx;
		const v2f b = Z * X.yx;
		const v2f c = X * Y.yx;
		const float ppx = a.x - a.y, ppy = b.x - b.y, pz = c.x - c.y;
		const lmask unstable = LMASK(fabsf(pz) < 1e-6f);
		if (__builtin_expect(unstable == 0ull, 1)) {
			inv_pz = div_nr(1.0f, pz);   // (a bare 1-ulp v_rcp here costs 1e-4 in dL_dscale)
			sx = ppx * inv_pz; sy = ppy * inv_pz;
			rho3d = sx * sx + sy * sy;
		} else {
			inv_pz = div_nr(1.0f, selm(unstable, 1.0f, pz));
			sx = selm(unstable, 0.f, ppx * inv_pz); sy = selm(unstable, 0.f, ppy * inv_pz);
			rho3d = selm(unstable, 1e8f, sx * sx + sy * sy);
		}
		d = R.xy() - pix;
		const v2f d2 = d * d;
		rho2d = S_FILTER_INV_SQ * (d2.x + d2.y);
		const v2f st2 = mk2(sx, sy) * Tw;
		depth = (st2.x + st2.y) + R.Twz();
	}
	const float power = -0.5f * min_raw(rho3d, rho2d);
	const float G = exp_neg(power);   // compensated exp (gsr_internal.hpp): plain exp2(x*log2e) is 3e-7 off, amplified ~200x by this backward
	const float alpha_raw = fminf(0.99f, R.opac() * G);
	const lmask ok = LMASK(!(depth < S_NEAR)) & LMASK(!(alpha_raw < 1.0f / 255.0f)) & LMASK(contributor < s.last_contributor) & inside_m;
	const float alpha = selm0(ok, alpha_raw);
	const float c_d = selm(ok, depth, 1.0f);
	// a rejected pair may carry inf/NaN in s (it overflows when the ray grazes the splat plane); 0 * that must stay 0.  G needs
	// no such care: v_min drops a NaN rho3d, so it is always in [0, 1].
	const v2f sxy = mk2(selm0(ok, sx), selm0(ok, sy));
	const float inv_1ma = div_nr(1.0f, 1.f - alpha);
	s.T *= inv_1ma;
	const float T = s.T;
	const float w = alpha * T;
	const v2f w2 = mk2(w, w);
	// D = <attributes of this surfel, upstream gradients>, pairs first
	v2f Dv = R.rg() * s.dp01;
	Dv = __builtin_elementwise_fma(R.nxy(), s.dn01, Dv);
	Dv = __builtin_elementwise_fma(mk2(R.nz(), R.refl()), s.dnzr, Dv);
	const float D = (Dv.x + Dv.y) + R.b() * s.dp2 + c_d * s.dL_ddepth + s.dL_daccum;
	s.A = s.last_alpha * s.Dprev + (1.f - s.last_alpha) * s.A;
	s.Dprev = D;
	s.last_alpha = alpha;
	float dL_dalpha = D - s.A;
	const v2f vc = w2 * s.dp01, vn = w2 * s.dn01, vz = w2 * s.dnzr;
	v[SA_COLOR + 0] = vc.x;
	v[SA_COLOR + 1] = vc.y;
	v[SA_COLOR + 2] = w * s.dp2;
	v[SA_REFL] = vz.y;
	v[SA_NORMAL + 0] = vn.x;
	v[SA_NORMAL + 1] = vn.y;
	v[SA_NORMAL + 2] = vz.x;
	// distortion regulariser
	const float rc = div_nr(1.0f, c_d);
	const float m_d = S_FAR / (S_FAR - S_NEAR) * (1 - S_NEAR * rc);
	const float dmd_dd = (S_FAR * S_NEAR / (S_FAR - S_NEAR)) * rc * rc;
	const float dL_dweight = s.FD2r + m_d * (m_d * s.FAr - 2 * s.FDr);
	dL_dalpha += dL_dweight - s.last_dL_dT;
	s.last_dL_dT = dL_dweight * alpha + (1 - alpha) * s.last_dL_dT;
	const float dL_dmd = 2.0f * w * (m_d * s.FAr - s.FDr);
	float dL_dz = dL_dmd * dmd_dd + w * s.dL_ddepth;
	dL_dz += (contributor == s.median_contributor - 1) ? s.dL_dmedian_depth : 0.f;
	dL_dalpha *= T;
	dL_dalpha -= s.T_final * inv_1ma * s.bg_dot_dpixel;
	dL_dalpha = selm0(ok, dL_dalpha);
	dL_dz = selm0(ok, dL_dz);
	const float nG = -G * (R.opac() * dL_dalpha);   // dL_dG * -G
	const lmask use3d = LMASK(rho3d <= rho2d);
	// dL_ds = nG * s + dL_dz * Tw.xy on the ray-splat branch, 0 on the low-pass branch
	const v2f dL_ds = __builtin_elementwise_fma(mk2(nG, nG), sxy, mk2(dL_dz, dL_dz) * R.Twxy());
	const v2f dp = mk2(selm0(use3d, dL_ds.x), selm0(use3d, dL_ds.y)) * inv_pz;   // dL_dp.xy
	const v2f dps = dp * sxy;
	const float dpz = -(dps.x + dps.y);
	// With X = (k.x, l.x) etc.:  Z * dp.y - Y * dp.z = (k.z dp.y - k.y dp.z, l.z dp.y - l.y dp.z) = (-nl.x, nk.x) where
	// nk = -dL_dk = -(l x dp) and nl = -dL_dl = -(dp x k) are what the Tu / Tv rows receive.  The accumulator holds -nl in
	// slots SA_T+3..5; the per-Gaussian backward flips the sign when it reads them.
	const v2f dpx2 = mk2(dp.x, dp.x), dpy2 = mk2(dp.y, dp.y), dpz2 = mk2(dpz, dpz);
	const v2f N1 = __builtin_elementwise_fma(Z, dpy2, -(Y * dpz2));      // (-nl.x, nk.x)
	const v2f N2 = __builtin_elementwise_fma(X, dpz2, -(Z * dpx2));      // (-nl.y, nk.y)
	const v2f N3 = __builtin_elementwise_fma(Y, dpx2, -(X * dpy2));      // (-nl.z, nk.z)
	v[SA_T + 0] = N1.y; v[SA_T + 1] = N2.y; v[SA_T + 2] = N3.y;
	v[SA_T + 3] = N1.x; v[SA_T + 4] = N2.x; v[SA_T + 5] = N3.x;       // = -nl
	// third row: dL_dz * (s, 1) - (pix.x * nk + pix.y * nl) = dL_dz * (s, 1) - (N . pixs), pixs = (-pix.y, pix.x)
	const v2f t1 = N1 * s.pixs, t2 = N2 * s.pixs, t3 = N3 * s.pixs;
	v[SA_T + 6] = dL_dz * sxy.x - (t1.y + t1.x);
	v[SA_T + 7] = dL_dz * sxy.y - (t2.y + t2.x);
	v[SA_T + 8] = dL_dz - (t3.y + t3.x);
	const v2f gm = d * selm(use3d, 0.f, nG * S_FILTER_INV_SQ);   // low-pass branch: gradient to the 2D centre only
	v[SA_MEAN2D + 0] = gm.x;
	v[SA_MEAN2D + 1] = gm.y;
	v[SA_OPAC] = G * dL_dalpha;
	return ok;
}

// Sub-block form of the backward: the wave still owns an 8x8 pixel block, but each of its four 16-lane rows is a 4x4 pixel
// sub-block that walks ITS OWN list back to front — the forward left, per (sub-block, batch of 64 list entries), the mask of
// the entries that blended into at least one of its 16 pixels.  In one iteration row r differentiates the next entry of
// its mask; the four rows work on up to four different surfels.  Measured on the C3 scene (tests/blend_stats.py): a shared
// 8x8 list needs 3.13 M wave iterations with 31 % of the lanes carrying a blending pixel; four row lists that advance
// batch by batch need 2.29 M (the ideal for 4x4 sub-blocks is 1.80 M).  What it costs: the record differs per row, so it
// lives in VGPRs — staged per batch through LDS by the lane that owns the entry and picked up with row-uniform
// ds_read_b128 — and the in-row DPP reduction (row_reduce20) yields a total per (row, surfel).  A blended surfel touches 2.3
// of the four rows on average, and sending every row total to memory on its own cost more than the shorter lists gained
// (2.3x the float atomics: 0.99 ms against 0.93 ms for the shared list, 0.76 ms with the atomics switched off).  So the row
// totals of a batch are parked in a slab of (row, entry) slots — slot = (entries of the rows before) + (iteration of the
// row), plain stores, no two writers — and when the batch is done each entry's rows are added up and leave as ONE 80-byte
// row of float atomics, as in the shared-list form.
#ifndef S_CAP
#define S_CAP 48
#endif
// S_CAP: slab slots = (sub-block, entry) pairs differentiated between two flushes; a batch with more is cut (rare)
__device__ __forceinline__ void
surfel_render_bwd_rows_body(const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list, int W, int H, int tiles_x, int ntiles,
                            const float* __restrict__ bg, const float4* __restrict__ rec, int dev_flags, const float* __restrict__ final_Ts,
                            const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths,
                            const float* __restrict__ dL_drefl_map, const float* __restrict__ dL_dnormal_extra, float* __restrict__ acc,
                            const unsigned long long* __restrict__ blend_mask) {
	const uint32_t slot = xcd_slot(blockIdx.x);   // dispatch slot -> (tile, quadrant), longest lists first
	if (slot >= (uint32_t)ntiles * 4u) return;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_order[slot >> 2]), quad = slot & 3u;   // (readfirstlane: the compiler cannot see that the loaded tile id is wave-uniform)
	const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
	const int lane = threadIdx.x, row = lane >> 4;
	const int bx0 = tile_x * 16 + (quad & 1) * 8, by0 = tile_y * 16 + (quad >> 1) * 8;
	if (bx0 >= W || by0 >= H) return;
	const int px = bx0 + sub_px(lane), py = by0 + sub_py(lane);
	const bool inside = px < W && py < H;
	const lmask inside_m = LMASK(px < W) & LMASK(py < H);
	const v2f pixv = mk2((float)px, (float)py);
	const uint2 range = ranges[tile];
	const int count = (int)(range.y - range.x);
	const size_t HW = (size_t)H * W;
	const size_t pix = (size_t)W * py + px;

	__shared__ float s_slab[(S_CAP + 2) * S_ACC_F];     // [slot][20 floats]; slot S_CAP takes the stores of rows that have run out, slot S_CAP + 1 stays zero
	__shared__ float4 s_rec[S_WBATCH * S_REC_F4];       // records of the batch's blended entries (indexed by position in the batch)
	__shared__ uint32_t s_cw[S_WBATCH];                 // per entry of the chunk (compacted): the slab slot of each of the four rows (6 bits each) and, from bit 24, its batch position
#if GSR_BWD_LIST
	__shared__ uint8_t s_list[64];                      // by slab slot (a row's slots are consecutive, in the order it visits them: last entry first): the batch position
#endif
	static_assert(S_CAP + 1 < 64, "slab slots are packed into 6 bits");

	SurfelBwdPix st;
	surfel_bwd_init(st, inside, pix, HW, bg, final_Ts, n_contrib, dL_dpixels, dL_depths, dL_drefl_map, dL_dnormal_extra);
	st.pixs = mk2(-pixv.y, pixv.x);
	int wave_last = st.last_contributor;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off));
	wave_last = __builtin_amdgcn_readfirstlane(wave_last);   // (uniform after the butterfly; tells the compiler so)
	if (wave_last == 0) return;
	if (lane < S_ACC_F) s_slab[(S_CAP + 1) * S_ACC_F + lane] = 0.f;
	// byte offset inside a slab slot where this lane parks its row totals: quad q of a row holds value slot(q) of every reduced register
	const uint32_t slab_lane = (uint32_t)row_reduce_slot(lane) * 4u;
	const unsigned long long lt = (1ull << lane) - 1ull, gt = ~((2ull << lane) - 1ull);   // batch positions below / above this lane's
	// the flush: lanes 0..59 = 3 entries x 20 floats per pass
	const uint32_t fl_k = (uint32_t)lane / S_ACC_F, fl_d = (uint32_t)lane - fl_k * S_ACC_F;
	const bool fl_on = lane < 3 * S_ACC_F && fl_d < S_ACC_F - 1;
	float* const fl_acc = acc + fl_d;
	const char* const slab_b = reinterpret_cast<const char*>(s_slab);

	const size_t batch0 = (size_t)(range.x / S_WBATCH) + tile;
	for (int b = (min(wave_last, count) - 1) / S_WBATCH; b >= 0; b--) {
		const unsigned long long* mp = blend_mask + ((batch0 + (size_t)b) * 16u + quad * 4u);
		const lmask m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3];
		const lmask any = (m0 | m1) | (m2 | m3);
		if (any == 0ull) continue;
		// ---- 1. stage: lane l owns batch position l; if its entry blended anywhere in the block it fetches the record for the wave
		uint32_t id = 0u;
		if ((any >> lane) & 1ull) {
			id = point_list[range.x + (uint32_t)(b * S_WBATCH + lane)];
			const float4* q = rec + (size_t)id * S_REC_F4;
			const float4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3];
			float4 r4 = q[4];
			r4.w = __uint_as_float(id);     // the record's last float (the env-scope mask) is not read by the backward: the flush finds the surfel id there
			float4* d = s_rec + lane * S_REC_F4;
			d[0] = r0; d[1] = r1; d[2] = r2; d[3] = r3; d[4] = r4;
		}
		// a row that has run out re-reads some staged record (finite values; all its lanes are masked off)
		const uint32_t jany = 63u - (uint32_t)__builtin_clzll(any);
		// ---- 2. chunks: normally the whole batch; a batch with more than S_CAP (row, entry) pairs is cut at multiples of 8 positions
		lmask todo = any;
		while (todo != 0ull) {
			lmask c = ~0ull;
			if (__popcll(m0 & todo) + __popcll(m1 & todo) + __popcll(m2 & todo) + __popcll(m3 & todo) > S_CAP) {
				c = 0ull;
#pragma unroll 1
				for (int e = 7; e >= 0; e--) {          // from the top: the list is walked back to front
					const lmask em = (0xFFull << (8 * e)) & todo;
					if (em == 0ull) continue;
					const lmask cc = c | em;
					if (c != 0ull && __popcll(m0 & cc) + __popcll(m1 & cc) + __popcll(m2 & cc) + __popcll(m3 & cc) > S_CAP) break;
					c = cc;
				}
			}
			const lmask c0 = m0 & todo & c, c1 = m1 & todo & c, c2 = m2 & todo & c, c3 = m3 & todo & c, call = any & todo & c;
			todo &= ~c;
			const int n0 = __popcll(c0), n1 = __popcll(c1), n2 = __popcll(c2), n3 = __popcll(c3);
			const int nmax = __builtin_amdgcn_readfirstlane(max(max(n0, n1), max(n2, n3)));
			const int nent = __popcll(call);
			// slab slot of (row r, entry): the rows' slots are laid out one row after the other, in the order the row visits them;
			// a row that skips the entry is pointed at the zero slot
			if ((call >> lane) & 1ull) {
				const uint32_t s0 = ((c0 >> lane) & 1ull) ? (uint32_t)__popcll(c0 & gt) : (uint32_t)(S_CAP + 1);
				const uint32_t s1 = ((c1 >> lane) & 1ull) ? (uint32_t)(n0 + __popcll(c1 & gt)) : (uint32_t)(S_CAP + 1);
				const uint32_t s2 = ((c2 >> lane) & 1ull) ? (uint32_t)(n0 + n1 + __popcll(c2 & gt)) : (uint32_t)(S_CAP + 1);
				const uint32_t s3 = ((c3 >> lane) & 1ull) ? (uint32_t)(n0 + n1 + n2 + __popcll(c3 & gt)) : (uint32_t)(S_CAP + 1);
				const int k = __popcll(call & lt);
				s_cw[k] = s0 | (s1 << 6) | (s2 << 12) | (s3 << 18) | ((uint32_t)lane << 24);
#if GSR_BWD_LIST
				// the row's visiting order is its slab-slot order
				if ((c0 >> lane) & 1ull) s_list[s0] = (uint8_t)lane;
				if ((c1 >> lane) & 1ull) s_list[s1] = (uint8_t)lane;
				if ((c2 >> lane) & 1ull) s_list[s2] = (uint8_t)lane;
				if ((c3 >> lane) & 1ull) s_list[s3] = (uint8_t)lane;
#endif
			}
			__syncthreads();
			// ---- 3. differentiate: every row takes the LAST remaining entry of its own mask.  The mask is kept bit-reversed so that
			// "last entry" is the lowest set bit (v_ffbl) and dropping it is x & (x - 1).
#if !GSR_BWD_LIST
			unsigned long long left = __builtin_bitreverse64(row == 0 ? c0 : (row == 1 ? c1 : (row == 2 ? c2 : c3)));
#else
			// (round 3) the row reads its next entry from the list the staging lanes wrote (one ds_read_u8) instead of walking its bit mask
			// (two v_ffbl, a 64-bit x & (x - 1), min / sub / or: 11 VALU instructions per iteration in a loop bound by VALU issue)
			const int nrow = row == 0 ? n0 : (row == 1 ? n1 : (row == 2 ? n2 : n3));
#endif
			const uint32_t rowbase = (uint32_t)(row == 0 ? 0 : (row == 1 ? n0 : (row == 2 ? n0 + n1 : n0 + n1 + n2)));   // first slab slot of the row
			// byte offset of the row's next slab slot
			uint32_t myslot = rowbase * (S_ACC_F * 4u) + slab_lane;
#pragma unroll GSR_BWD_UNROLL
			for (int t = 0; t < nmax; t++) {
#if !GSR_BWD_LIST
				const uint32_t lo = (uint32_t)left, hi = (uint32_t)(left >> 32);
				const lmask valid = LMASK((lo | hi) != 0u);
				const uint32_t tz = min(ffbl_raw(lo), ffbl_raw(hi) | 32u);   // count of trailing zeros; garbage if left == 0
				const uint32_t j = selmu(valid, 63u - tz, jany);
				left &= left - 1ull;
#else
				const lmask valid = LMASK(t < nrow);
				const uint32_t j = selmu(valid, (uint32_t)s_list[min(rowbase + (uint32_t)t, 63u)], jany);
#endif
				const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + __umul24(j, S_REC_F4 * 16u));   // (24-bit multiply: full rate)
				const SurfelRec R{q[0], q[1], q[2], q[3], q[4]};
				float v[S_ACC_F];
				v[S_ACC_F - 1] = 0.f;
				// (no early exit: the masks are exact, an entry without a blending lane is a rare forced one, and then v is all zeros)
				surfel_bwd_pair(st, R, pixv, b * S_WBATCH + (int)j, inside_m & valid, v);
				// 20 values -> 5 registers of per-row totals; every lane of quad q of row r parks value slot(q) of each register
				float z[5];
				row_reduce20(v, z);
				float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(s_slab) + selmu(valid, myslot, (uint32_t)(S_CAP * S_ACC_F * 4) + slab_lane));
				myslot += S_ACC_F * 4u;
#pragma unroll
				for (int g = 0; g < 5; g++) slab[4 * g] = z[g];
			}
			// ---- 4. flush: lane -> (entry of the chunk, float d), three entries per pass; the entry's rows are added up and leave as
			// 80 contiguous bytes of float atomics per surfel
			__syncthreads();
#pragma unroll 1
			for (int k0 = 0; k0 < nent; k0 += 3) {
				const uint32_t k = (uint32_t)k0 + fl_k;
				if (fl_on && k < (uint32_t)nent) {
					const uint32_t w = s_cw[k];
					const uint32_t eid = __float_as_uint(s_rec[(w >> 24) * S_REC_F4 + 4].w);
					const char* base = slab_b + fl_d * 4u;
					const float a0 = *reinterpret_cast<const float*>(base + (w & 0x3Fu) * (S_ACC_F * 4u));
					const float a1 = *reinterpret_cast<const float*>(base + ((w >> 6) & 0x3Fu) * (S_ACC_F * 4u));
					const float a2 = *reinterpret_cast<const float*>(base + ((w >> 12) & 0x3Fu) * (S_ACC_F * 4u));
					const float a3 = *reinterpret_cast<const float*>(base + ((w >> 18) & 0x3Fu) * (S_ACC_F * 4u));
					if (!(dev_flags & 1)) atomicAdd(fl_acc + (size_t)eid * S_ACC_F, (a0 + a1) + (a2 + a3));
				}
			}
			__syncthreads();
		}
	}
}

// 97 VGPRs, 9.6 KB of LDS per wave: 4 waves per SIMD.  (The shared-list form this kernel replaced, measured at C3 with
// amdgpu_waves_per_eu = 3 / 4 / 5 / 6 / 8: 1.29 / 1.24 / 1.25 / 1.29 / 1.46 ms — occupancy beyond 4 buys nothing for a VALU-issue-bound kernel.)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSR_BWD_WPE, GSR_BWD_WPE)))
surfel_render_bwd_rows_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list, int W, int H, int tiles_x, int ntiles,
                              const float* __restrict__ bg, const float4* __restrict__ rec, int dev_flags, const float* __restrict__ final_Ts,
                              const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths,
                              const float* __restrict__ dL_drefl_map, const float* __restrict__ dL_dnormal_extra, float* __restrict__ acc,
                              const unsigned long long* __restrict__ blend_mask) {
	surfel_render_bwd_rows_body(ranges, tile_order, point_list, W, H, tiles_x, ntiles, bg, rec, dev_flags, final_Ts, n_contrib, dL_dpixels, dL_depths,
	                            dL_drefl_map, dL_dnormal_extra, acc, blend_mask);
}

// quat_to_rotmat_vjp (DSR auxiliary.h:242-286)
__device__ __forceinline__ void quat_vjp(float w, float x, float y, float z, const M3& v_R, float* v_quat) {
	v_quat[0] = 2.f * (x * (v_R.m[1][2] - v_R.m[2][1]) + y * (v_R.m[2][0] - v_R.m[0][2]) + z * (v_R.m[0][1] - v_R.m[1][0]));
	v_quat[1] = 2.f * (-2.f * x * (v_R.m[1][1] + v_R.m[2][2]) + y * (v_R.m[0][1] + v_R.m[1][0]) + z * (v_R.m[0][2] + v_R.m[2][0]) + w * (v_R.m[1][2] - v_R.m[2][1]));
	v_quat[2] = 2.f * (x * (v_R.m[0][1] + v_R.m[1][0]) - 2.f * y * (v_R.m[0][0] + v_R.m[2][2]) + z * (v_R.m[1][2] + v_R.m[2][1]) + w * (v_R.m[2][0] - v_R.m[0][2]));
	v_quat[3] = 2.f * (x * (v_R.m[0][2] + v_R.m[2][0]) + y * (v_R.m[1][2] + v_R.m[2][1]) - 2.f * z * (v_R.m[0][0] + v_R.m[1][1]) + w * (v_R.m[0][1] - v_R.m[1][0]));
}

// compute_transmat_aabb + preprocessCUDA backward (DSR backward.cu:473-660), one thread per surfel.
// Writes every output element (zeros for culled surfels).
// ACC: the parameter gradients (mean3D, sh, opacity, scale, rotation, refl strength) are ADDED to the output tensors instead
// of written: several views accumulate into one gradient buffer on the device (gsr_surfel_backward_accum).
template <bool ACC>
__global__ void __launch_bounds__(256)
surfel_preprocess_bwd_kernel(int P, int D, int M, const float* __restrict__ means, const int* __restrict__ radii, const float* __restrict__ shs,
                             const uint8_t* __restrict__ clamped, const float* __restrict__ scales, const float* __restrict__ rotations,
                             const float4* __restrict__ rec, SurfelCam cam, const float* __restrict__ acc, float* __restrict__ dL_dmean2D,
                             float* __restrict__ dL_dnormal, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
                             float* __restrict__ dL_drefl, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dtransMat,
                             float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
	// Outputs are AoS rows of the reference's tensors; the wave stores them together through LDS (wave_store_rows): stored by
	// the lane that computed them this kernel issued ~970 write requests per wave, three times what its bytes need.
	__shared__ __attribute__((aligned(16))) float s_tile[4][1368];     // 9 planes of 65 dwords or 4 planes of 65 float4, 16-byte aligned per wave
	const int lane = threadIdx.x & 63;
	float* tile = s_tile[threadIdx.x >> 6];
	const int idx_raw = blockIdx.x * 256 + threadIdx.x;
	const int g0 = idx_raw - lane, nrows = min(64, P - g0);      // the wave's first row, its rows inside the arrays
	if (nrows <= 0) return;                                      // (wave-uniform)
	const bool in_range = idx_raw < P;
	const int idx = in_range ? idx_raw : P - 1;                  // lanes past the end recompute the last row; their rows are not stored
	const float4* a4 = reinterpret_cast<const float4*>(acc + (size_t)idx * S_ACC_F);
	const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3], a4v = a4[4];
	const float gcol[3] = {a0.x, a0.y, a0.z};
	const float gnrm[3] = {a1.x, a1.y, a1.z};
	// render-accumulated dL_dtransMat; the tile kernel accumulates the Tv row with the opposite sign (see surfel_bwd_pair)
	float dT[9] = {a2.x, a2.y, a2.z, -a2.w, -a3.x, -a3.y, a3.z, a3.w, a4v.x};
	const float gm2x = a4v.y, gm2y = a4v.z;
	if (in_range) {
		put<ACC>(dL_drefl + idx, a0.w);
		put<ACC>(dL_dopacity + idx, a1.w);
	}

	float dmean[3] = {0.f, 0.f, 0.f}, dscale[2] = {0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
	float out_m2x = gm2x, out_m2y = gm2y;
	float dTout[9];
#pragma unroll
	for (int i = 0; i < 9; i++) dTout[i] = dT[i];
	const bool visible = radii[idx] > 0;
	const float mx = means[3 * idx], my = means[3 * idx + 1], mz = means[3 * idx + 2];
	if (visible) {
		const int Wb = f2i(cam.focal_x * cam.tan_fovx * 2);  // DSR backward.cu:637-638
		const int Hb = f2i(cam.focal_y * cam.tan_fovy * 2);
		const bool precomp = (scales == nullptr);
		M3 T;
		F3 normal = f3(0.f, 0.f, 0.f);
		P34 Pm;
		M3 R;
		float qw = 0, qx = 0, qy = 0, qz = 0, sc0 = 0, sc1 = 0;
		const float4* rc = rec + (size_t)idx * S_REC_F4;
		if (precomp) {
			const float4 r0 = rc[0], r1 = rc[1], r2 = rc[2];
			T = m3_make(r0.z, r1.x, r1.z, r0.w, r1.y, r1.w, r2.x, r2.y, r2.z);   // (Tu | Tv | Tw) from the paired record layout
		} else {
			R = quat_to_rotmat(rotations + 4 * idx, qw, qx, qy, qz);
			sc0 = scales[2 * idx]; sc1 = scales[2 * idx + 1];   // scale_to_mat(scale, 1.0f): scale_modifier ignored (backward.cu:511)
			const float L0[3] = {R.m[0][0] * sc0, R.m[0][1] * sc0, R.m[0][2] * sc0};
			const float L1[3] = {R.m[1][0] * sc1, R.m[1][1] * sc1, R.m[1][2] * sc1};
			Pm = make_P(cam.proj, Wb, Hb);
#pragma unroll
			for (int j = 0; j < 3; j++) {
				T.m[j][0] = L0[0] * Pm.m[j][0] + L0[1] * Pm.m[j][1] + L0[2] * Pm.m[j][2];
				T.m[j][1] = L1[0] * Pm.m[j][0] + L1[1] * Pm.m[j][1] + L1[2] * Pm.m[j][2];
				T.m[j][2] = mx * Pm.m[j][0] + my * Pm.m[j][1] + mz * Pm.m[j][2] + Pm.m[j][3];
			}
			const float* vm = cam.view;
			const float nx = R.m[2][0], ny = R.m[2][1], nz = R.m[2][2];
			normal = f3(vm[0] * nx + vm[4] * ny + vm[8] * nz, vm[1] * nx + vm[5] * ny + vm[9] * nz, vm[2] * nx + vm[6] * ny + vm[10] * nz);
		}
		M3 dL_dT = m3_make(dT[0], dT[1], dT[2], dT[3], dT[4], dT[5], dT[6], dT[7], dT[8]);
		bool early = false;
		if (gm2x != 0 || gm2y != 0) {
			// gradient of the AABB centre w.r.t. T (backward.cu:545-573)
			const float tv[3] = {9.0f, 9.0f, -1.0f};
			const float d = tv[0] * (T.m[2][0] * T.m[2][0]) + tv[1] * (T.m[2][1] * T.m[2][1]) + tv[2] * (T.m[2][2] * T.m[2][2]);
			const float invd = 1.0f / d;
			float fv[3], dL_df[3], dT3[3];
#pragma unroll
			for (int r = 0; r < 3; r++) fv[r] = tv[r] * invd;
#pragma unroll
			for (int r = 0; r < 3; r++) {
				dL_dT.m[0][r] += gm2x * fv[r] * T.m[2][r];
				dL_dT.m[1][r] += gm2y * fv[r] * T.m[2][r];
				dT3[r] = gm2x * fv[r] * T.m[0][r] + gm2y * fv[r] * T.m[1][r];
				dL_df[r] = gm2x * T.m[0][r] * T.m[2][r] + gm2y * T.m[1][r] * T.m[2][r];
			}
			const float dL_dd = (float)((double)(dL_df[0] * fv[0] + dL_df[1] * fv[1] + dL_df[2] * fv[2]) * (-1.0 / (double)d));
#pragma unroll
			for (int r = 0; r < 3; r++) dL_dT.m[2][r] += dT3[r] + dL_dd * (tv[r] * T.m[2][r] * 2.0f);
			if (precomp) {
#pragma unroll
				for (int c = 0; c < 3; c++)
#pragma unroll
					for (int r = 0; r < 3; r++) dTout[3 * c + r] = dL_dT.m[c][r];
				early = true;
			}
		}
		if (!precomp && !early) {
			// dL_dM = P * dL_dT^T  (4 rows x 3 cols); only rows 0..2 are used
			float dM[3][3];
#pragma unroll
			for (int j = 0; j < 3; j++)
#pragma unroll
				for (int r = 0; r < 3; r++) dM[j][r] = Pm.m[0][r] * dL_dT.m[0][j] + Pm.m[1][r] * dL_dT.m[1][j] + Pm.m[2][r] * dL_dT.m[2][j];
			const float* vm = cam.view;
			float tnx = vm[0] * gnrm[0] + vm[1] * gnrm[1] + vm[2] * gnrm[2];
			float tny = vm[4] * gnrm[0] + vm[5] * gnrm[1] + vm[6] * gnrm[2];
			float tnz = vm[8] * gnrm[0] + vm[9] * gnrm[1] + vm[10] * gnrm[2];
			const float pvx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
			const float pvy = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
			const float pvz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
			const float cosv = -((pvx * normal.x) + (pvy * normal.y) + (pvz * normal.z));
			const float mult = cosv > 0 ? 1.f : -1.f;
			tnx *= mult; tny *= mult; tnz *= mult;
			M3 dL_dR;
#pragma unroll
			for (int r = 0; r < 3; r++) {
				dL_dR.m[0][r] = dM[0][r] * sc0;
				dL_dR.m[1][r] = dM[1][r] * sc1;
			}
			dL_dR.m[2][0] = tnx; dL_dR.m[2][1] = tny; dL_dR.m[2][2] = tnz;
			quat_vjp(qw, qx, qy, qz, dL_dR, drot);
			dscale[0] = dM[0][0] * R.m[0][0] + dM[0][1] * R.m[0][1] + dM[0][2] * R.m[0][2];
			dscale[1] = dM[1][0] * R.m[1][0] + dM[1][1] * R.m[1][1] + dM[1][2] * R.m[1][2];
			dmean[0] = dM[2][0]; dmean[1] = dM[2][1]; dmean[2] = dM[2][2];
		}
		// densification signal overwrites dL_dmean2D.xy (backward.cu:656-659); it reads dL_dtransMat as stored
		// transMats[idx*9+8] = Tw.z of the forward.  Recomputed (same expression, same order, no contraction: the same bits) unless
		// the transform was supplied: reading it back costs a 64-byte sector per Gaussian for 4 bytes.
		float depth;
		{
#pragma clang fp contract(off)
			depth = precomp ? rc[2].z : ((mx * cam.proj[3] + my * cam.proj[7]) + mz * cam.proj[11]) + cam.proj[15];
		}
		out_m2x = (float)((double)(dTout[2] * depth) * 0.5 * (double)float(Wb));
		out_m2y = (float)((double)(dTout[5] * depth) * 0.5 * (double)float(Hb));
	}
	if (shs != nullptr) {
		if (M == 16) {
			// the row is the outer product w x dL_dRGB (gsr_math.hpp): three passes of one 64-byte sector per row
			float w[16];
			F3 grgb = f3(0.f, 0.f, 0.f);
#pragma unroll
			for (int k = 0; k < 16; k++) w[k] = 0.f;
			// (a surfel that blended into no pixel has a zero colour gradient: its SH row of the gradient and the view-direction term are
			// zeros whatever the coefficients are, so their 192 bytes are not read)
			if (visible && (gcol[0] != 0.f || gcol[1] != 0.f || gcol[2] != 0.f)) {
				ShRow s;
				load_sh(shs, idx, M, (D + 1) * (D + 1), s);
				const F3 dir = f3(mx - cam.campos[0], my - cam.campos[1], mz - cam.campos[2]);
				grgb = f3(gcol[0], gcol[1], gcol[2]);
				const F3 dm = sh_backward_weights(D, s, dir, clamped[idx], grgb, w);
				dmean[0] += dm.x; dmean[1] += dm.y; dmean[2] += dm.z;
			}
			float4* sh_out = reinterpret_cast<float4*>(dL_dsh) + (size_t)g0 * 12;
#pragma unroll
			for (int t = 0; t < 3; t++) {
				const float4 q4[4] = {sh_row_f4(4 * t, w, grgb), sh_row_f4(4 * t + 1, w, grgb), sh_row_f4(4 * t + 2, w, grgb), sh_row_f4(4 * t + 3, w, grgb)};
				wave_store_rows4<4, ACC>(reinterpret_cast<float4*>(tile), q4, sh_out, 12, 4 * t, nrows, lane);
			}
		} else if (in_range) {
			if (visible) {
				ShRow s;
				load_sh(shs, idx, M, (D + 1) * (D + 1), s);
				const F3 dir = f3(mx - cam.campos[0], my - cam.campos[1], mz - cam.campos[2]);
				const F3 dm = sh_backward<ACC>(idx, D, M, s, dir, clamped[idx], f3(gcol[0], gcol[1], gcol[2]), dL_dsh);
				dmean[0] += dm.x; dmean[1] += dm.y; dmean[2] += dm.z;
			} else if (!ACC) {
				float* out = dL_dsh + (size_t)idx * M * 3;
				for (int q = 0; q < M * 3; q++) out[q] = 0.f;
			}
		}
	}
	const float m2d[3] = {out_m2x, out_m2y, 0.f};
	// (per-view outputs the caller did not ask for — NULL — are not written: 60 of this kernel's ~690 bytes per surfel when shs and
	// scales / rotations are the inputs, as in the reference's training path)
	if (dL_dcolor != nullptr) wave_store_rows<3, false>(tile, gcol, dL_dcolor + (size_t)g0 * 3, nrows, lane);
	if (dL_dnormal != nullptr) wave_store_rows<3, false>(tile, gnrm, dL_dnormal + (size_t)g0 * 3, nrows, lane);
	wave_store_rows<3, false>(tile, m2d, dL_dmean2D + (size_t)g0 * 3, nrows, lane);
	wave_store_rows<3, ACC>(tile, dmean, dL_dmean3D + (size_t)g0 * 3, nrows, lane);
	if (dL_dtransMat != nullptr) wave_store_rows<9, false>(tile, dTout, dL_dtransMat + (size_t)g0 * 9, nrows, lane);
	if (in_range) {
		put<ACC>(dL_dscale + 2 * idx, dscale[0]); put<ACC>(dL_dscale + 2 * idx + 1, dscale[1]);
		float4* rq = reinterpret_cast<float4*>(dL_drot) + idx;
		if (ACC) {
			if (drot[0] != 0.f || drot[1] != 0.f || drot[2] != 0.f || drot[3] != 0.f) {
				const float4 r0 = *rq;
				*rq = make_float4(r0.x + drot[0], r0.y + drot[1], r0.z + drot[2], r0.w + drot[3]);
			}
		} else {
			*rq = make_float4(drot[0], drot[1], drot[2], drot[3]);
		}
	}
}

}  // namespace gsr

using namespace gsr;

static SurfelCam make_scam(const float* view, const float* proj, const float* campos, int W, int H, float tan_fovx, float tan_fovy) {
	SurfelCam c;
	c.view = view; c.proj = proj; c.campos = campos; c.W = W; c.H = H;
	c.tan_fovx = tan_fovx; c.tan_fovy = tan_fovy;
	c.focal_y = H / (2.0f * tan_fovy);   // DSR rasterizer_impl.cu:228-229
	c.focal_x = W / (2.0f * tan_fovx);
	return c;
}

extern "C" int gsr_surfel_forward_refl(gsr_alloc_fn alloc, void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                                  const float* means3D, const uint8_t* env_scope_mask, const float* shs, const float* colors_precomp,
                                  const float* refl_strengths, const float* opacities, const float* scales, float scale_modifier,
                                  const float* rotations, const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                                  const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_others,
                                  float* out_refl_strength_map, int* radii, float* gaussian_weights, const gsr_refl_forward* refl, int debug,
                                  void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (!alloc || P < 0 || width <= 0 || height <= 0 || !background || !out_color || !out_others || !out_refl_strength_map) {
		set_error("gsr_surfel_forward: invalid argument");
		return GSR_E_INVALID;
	}
	if (refl && (!refl->cam || !refl->cubemap || !refl->fail_value || refl->L == 0 || !refl->cubemap_rgba || ((uintptr_t)refl->cubemap_rgba & 15) != 0 ||
	             !refl->out_final || !refl->out_refl_color || !refl->out_normal_world)) {
		set_error("gsr_surfel_forward_refl: incomplete reflection descriptor (cam, cubemap, fail_value, L, 16-byte aligned cubemap_rgba and the three outputs are required)");
		return GSR_E_INVALID;
	}
	const size_t HW = (size_t)width * height;
	if (P == 0) {
		GSR_HIP_CHECK(hipMemsetAsync(out_color, 0, HW * 3 * 4, stream));
		GSR_HIP_CHECK(hipMemsetAsync(out_others, 0, HW * 8 * 4, stream));
		GSR_HIP_CHECK(hipMemsetAsync(out_refl_strength_map, 0, HW * 4, stream));
		if (refl)    // an empty scene through the stand-alone pixel pass: the same arithmetic on zero planes
			return gsr_deferred_reflection_forward_keys(out_others + 2 * HW, out_color, out_refl_strength_map, refl->cam, refl->cubemap, refl->fail_value, refl->L,
			                                            width, height, refl->out_final, refl->out_refl_color, refl->out_normal_world, refl->cubemap_rgba,
			                                            refl->sort_keys, stream_) < 0 ? GSR_E_HIP : 0;
		return 0;
	}
	if (!means3D || !opacities || !refl_strengths || !viewmatrix || !projmatrix || !cam_pos || !radii || !gaussian_weights ||
	    (!shs && !colors_precomp) || ((!scales || !rotations) && !transMat_precomp)) {
		set_error("gsr_surfel_forward: missing required input pointer");
		return GSR_E_INVALID;
	}
	if (D < 0 || D > 3 || (shs && (D + 1) * (D + 1) > M)) { set_error("gsr_surfel_forward: SH degree %d not supported with M=%d", D, M); return GSR_E_INVALID; }
	if (shs && ((M * 3) & 3) == 0) GSR_REQUIRE_ALIGNED16(shs, "shs (rows of a multiple of 16 bytes)");
	const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
	const int ntiles = tiles_x * tiles_y;

	size_t geom_bytes = 0, img_bytes = 0;
	const size_t scan_bytes = scan_temp_bytes(P);
	carve_geom(nullptr, P, S_REC_F4, 0, S_ACC_F, scan_bytes, &geom_bytes);
	carve_image(nullptr, HW, ntiles, 3, 2, &img_bytes);
	void* gbuf = alloc(alloc_user, GSR_BUF_GEOM, geom_bytes);
	void* ibuf = alloc(alloc_user, GSR_BUF_IMAGE, img_bytes);
	if (!gbuf || !ibuf) { set_error("workspace allocation failed (%zu / %zu bytes)", geom_bytes, img_bytes); return GSR_E_ALLOC; }
	GeomState geom = carve_geom(gbuf, P, S_REC_F4, 0, S_ACC_F, scan_bytes, nullptr);
	ImageState img = carve_image(ibuf, HW, ntiles, 3, 2, nullptr);

	if (prefiltered) GSR_HIP_CHECK(hipMemsetAsync(geom.flags, 0, 4 * sizeof(int), stream));   // the flag is only written and read then
	const SurfelCam cam = make_scam(viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy);
	CubemapInterleave ci{nullptr, nullptr, 0u, 1u};
	if (refl) {
		if ((size_t)6 * refl->L * refl->L >= 0xFFFFFFFFull) { set_error("gsr_surfel_forward_refl: cubemap too large"); return GSR_E_INVALID; }
		ci = CubemapInterleave{refl->cubemap, reinterpret_cast<float4*>(refl->cubemap_rgba), 6u * refl->L * refl->L, refl->L * refl->L};
	}
{ StageTimer st_(GSR_STAGE_PREPROCESS, stream); 	surfel_preprocess_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
	                                                              transMat_precomp, colors_precomp, refl_strengths, env_scope_mask, cam, radii, geom,
	                                                              tiles_x, tiles_y, prefiltered, gaussian_weights, ci); }
	GSR_LAUNCH_CHECK(debug, stream);

	BinningState bin;
	const int R = run_binning(alloc, alloc_user, P, tiles_x, tiles_y, geom, img, &bin, prefiltered, debug, stream);
	if (R < 0) return R;

{ StageTimer st_(GSR_STAGE_RENDER_FWD, stream);
	const int nunits = (int)xcd_grid((uint32_t)ntiles * 4u);
	if (refl) {
		const SurfelReflFwd rf{refl->cam, refl->cubemap, reinterpret_cast<const float4*>(refl->cubemap_rgba), refl->fail_value, (int)refl->L, 6u * refl->L * refl->L,
		                       refl->out_final, refl->out_refl_color, refl->out_normal_world, refl->sort_keys};
		surfel_render_fwd_wave_kernel<true><<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, geom.rec, geom.bbox,
		                                                               option_cull(), background, img.final_T, img.n_contrib, out_color, out_others,
		                                                               out_refl_strength_map, gaussian_weights, bin.blend_mask, bin.mask_stride, rf);
	} else {
		surfel_render_fwd_wave_kernel<false><<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, geom.rec, geom.bbox,
		                                                                option_cull(), background, img.final_T, img.n_contrib, out_color, out_others,
		                                                                out_refl_strength_map, gaussian_weights, bin.blend_mask, bin.mask_stride, SurfelReflFwd{});
	} }
	GSR_LAUNCH_CHECK(debug, stream);
	if (refl && refl->sort_keys && refl->scratch) {
		const int rc = refl_sort_keys_early(refl->L, width, height, refl->scratch, refl->scratch_floats, refl->sort_keys, refl->async_sort, stream);
		if (rc < 0) return rc;
	}
	return R;
}

extern "C" int gsr_surfel_forward(gsr_alloc_fn alloc, void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                                  const float* means3D, const uint8_t* env_scope_mask, const float* shs, const float* colors_precomp,
                                  const float* refl_strengths, const float* opacities, const float* scales, float scale_modifier,
                                  const float* rotations, const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                                  const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_others,
                                  float* out_refl_strength_map, int* radii, float* gaussian_weights, int debug, void* stream_) {
	return gsr_surfel_forward_refl(alloc, alloc_user, P, D, M, background, width, height, means3D, env_scope_mask, shs, colors_precomp, refl_strengths, opacities,
	                               scales, scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color,
	                               out_others, out_refl_strength_map, radii, gaussian_weights, nullptr, debug, stream_);
}

extern "C" int gsr_surfel_backward_ex(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* refl_strengths, const float* scales,
                                   float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                                   void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix, const float* dL_dothers,
                                   const float* dL_drefl_strength_map, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                                   float* dL_drefl_strengths, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale,
                                   float* dL_drot, int accumulate, const float* dL_dnormal_extra, int debug, void* stream_) {
	(void)colors_precomp; (void)refl_strengths; (void)scale_modifier; (void)transMat_precomp;
	hipStream_t stream = (hipStream_t)stream_;
	if (P < 0 || R < 0 || width <= 0 || height <= 0) { set_error("gsr_surfel_backward: invalid size"); return GSR_E_INVALID; }
	if (P == 0) return 0;
	if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer) || !dL_dpix || !dL_dothers || !dL_drefl_strength_map || !dL_dmean2D ||
	    !dL_dopacity || !dL_drefl_strengths || !dL_dmean3D || !dL_dscale || !dL_drot || (!shs && !dL_dcolor) || (!scales && !dL_dtransMat) ||
	    (shs && !dL_dsh) || !radii || !means3D) {
		set_error("gsr_surfel_backward: missing required pointer");
		return GSR_E_INVALID;
	}
	if (shs && ((M * 3) & 3) == 0) GSR_REQUIRE_ALIGNED16(shs, "shs (rows of a multiple of 16 bytes)");
	if (shs && ((M * 3) & 3) == 0) GSR_REQUIRE_ALIGNED16(dL_dsh, "dL_dsh");
	GSR_REQUIRE_ALIGNED16(dL_drot, "dL_drot");
	const size_t HW = (size_t)width * height;
	const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
	const int ntiles = tiles_x * tiles_y;
	GeomState geom = carve_geom(geom_buffer, P, S_REC_F4, 0, S_ACC_F, scan_temp_bytes(P), nullptr);
	ImageState img = carve_image(image_buffer, HW, ntiles, 3, 2, nullptr);
	BinningState bin = carve_binning(binning_buffer, R, ntiles, 0, nullptr);

	GSR_HIP_CHECK(hipMemsetAsync(geom.acc, 0, (size_t)P * S_ACC_F * sizeof(float), stream));
	if (R > 0) {
		// a key sort of the reflection backward on the library's side stream must have reached its last pass before this kernel takes
		// every wave slot of the chip (side_gate_wait, gsr_cubemap.hip); nothing pending: no wait
		{ const int rc = side_gate_wait(stream); if (rc < 0) return rc; }
{ StageTimer st_(GSR_STAGE_RENDER_BWD, stream);
		const int nunits = (int)xcd_grid((uint32_t)ntiles * 4u);
		surfel_render_bwd_rows_kernel<<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, background, geom.rec,
		                                                         option_dev(), img.final_T, img.n_contrib, dL_dpix, dL_dothers, dL_drefl_strength_map, dL_dnormal_extra,
		                                                         geom.acc, bin.blend_mask); }
		GSR_LAUNCH_CHECK(debug, stream);
	}
	const SurfelCam cam = make_scam(viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy);
	// scales == NULL selects the transMat_precomp path in the per-surfel backward (DSR backward.cu:639)
{ StageTimer st_(GSR_STAGE_PREPROCESS_BWD, stream);
	auto kern = accumulate ? surfel_preprocess_bwd_kernel<true> : surfel_preprocess_bwd_kernel<false>;
	kern<<<(P + 255) / 256, 256, 0, stream>>>(P, D, M, means3D, radii, shs, geom.clamped, scales, rotations, geom.rec, cam, geom.acc, dL_dmean2D, dL_dnormal,
	                                          dL_dopacity, dL_dcolor, dL_drefl_strengths, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot); }
	GSR_LAUNCH_CHECK(debug, stream);
	return 0;
}

extern "C" int gsr_surfel_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* refl_strengths, const float* scales,
                                   float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                                   void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix, const float* dL_dothers,
                                   const float* dL_drefl_strength_map, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                                   float* dL_drefl_strengths, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale,
                                   float* dL_drot, int debug, void* stream_) {
	return gsr_surfel_backward_ex(P, D, M, R, background, width, height, means3D, shs, colors_precomp, refl_strengths, scales, scale_modifier, rotations,
	                              transMat_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer,
	                              dL_dpix, dL_dothers, dL_drefl_strength_map, dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_drefl_strengths, dL_dmean3D,
	                              dL_dtransMat, dL_dsh, dL_dscale, dL_drot, 0, nullptr, debug, stream_);
}

extern "C" int gsr_surfel_backward_accum(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* refl_strengths, const float* scales,
                                   float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                                   void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix, const float* dL_dothers,
                                   const float* dL_drefl_strength_map, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                                   float* dL_drefl_strengths, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale,
                                   float* dL_drot, int accumulate, int debug, void* stream_) {
	return gsr_surfel_backward_ex(P, D, M, R, background, width, height, means3D, shs, colors_precomp, refl_strengths, scales, scale_modifier, rotations,
	                              transMat_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer,
	                              dL_dpix, dL_dothers, dL_drefl_strength_map, dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_drefl_strengths, dL_dmean3D,
	                              dL_dtransMat, dL_dsh, dL_dscale, dL_drot, accumulate, nullptr, debug, stream_);
}
