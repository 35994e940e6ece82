// Variant G — 3D Gaussians (EWA projection, anti-aliasing, inverse depth, normal / reflection-strength
// channels).  MI355X-native restatement of the behaviour of submodules/diff-gaussian-rasterization
// (DGR cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu); design notes in DESIGN.md.
//
// Data layout: preprocess writes one 64-byte render record per Gaussian
//   f4[0] = (x, y, conic.x, conic.y)  f4[1] = (conic.z, opacity, r, g)
//   f4[2] = (b, nx, ny, nz)           f4[3] = (refl, 1/depth, -, -)
// which the tile kernels gather with four lanes per record (64 contiguous bytes) into LDS.
// A 16x16 tile is one 256-thread workgroup = 4 waves, each wave owning an 8x8 pixel quadrant so that
// wave-uniform skips ("no lane of this wave touches Gaussian j") fire as often as possible.
#include "gsr_internal.hpp"
#include "gsr_sort.hpp"
#include "gsr_math.hpp"

namespace gsr {

#define G_REC_F4 4
#define G_ACC_F 16
// accumulator slots (floats) of the backward tile kernel / per-Gaussian backward
#define GA_COLOR 0
#define GA_NORMAL 3
#define GA_REFL 6
#define GA_INVD 7
#define GA_MEAN2D 8
#define GA_MEAN2DP 10
#define GA_CONIC 12
#define GA_OPAC 15

struct GaussCam {
	const float* view;
	const float* proj;
	const float* campos;
	int W, H;
	float tan_fovx, tan_fovy, focal_x, focal_y;
};

// computeCov3D (DGR forward.cu:114-148): quaternion used as given (not normalised).
__device__ __forceinline__ void cov3d_from_scale_rot(const float* __restrict__ scale, float mod, const float* __restrict__ rot, float* cov3D) {
#pragma clang fp contract(off)
	M3 S = m3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
	S.m[0][0] = mod * scale[0];
	S.m[1][1] = mod * scale[1];
	S.m[2][2] = mod * scale[2];
	const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	M3 R = m3_make(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
	               1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
	               1.f - 2.f * (x * x + y * y));
	M3 Mm = m3_mul(S, R);
	M3 Sigma = m3_mul(m3_T(Mm), Mm);
	cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
	cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

// Shared by forward and backward: view-space point with the 1.3*tanfov clamp, J, W, T = W*J (DGR forward.cu:80-99).
struct Cov2DCtx {
	float tx, ty, tz, txtz, tytz, limx, limy;
	M3 T, Wm;
};
__device__ __forceinline__ Cov2DCtx cov2d_ctx(float mx, float my, float mz, const GaussCam& c) {
#pragma clang fp contract(off)
	const float* vm = c.view;
	Cov2DCtx k;
	float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
	float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
	const float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
	k.limx = 1.3f * c.tan_fovx;
	k.limy = 1.3f * c.tan_fovy;
	k.txtz = tx / tz;
	k.tytz = ty / tz;
	tx = fminf(k.limx, fmaxf(-k.limx, k.txtz)) * tz;
	ty = fminf(k.limy, fmaxf(-k.limy, k.tytz)) * tz;
	k.tx = tx; k.ty = ty; k.tz = tz;
	M3 J = m3_make(c.focal_x / tz, 0.0f, -(c.focal_x * tx) / (tz * tz), 0.0f, c.focal_y / tz, -(c.focal_y * ty) / (tz * tz), 0, 0, 0);
	k.Wm = m3_make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
	k.T = m3_mul(k.Wm, J);
	return k;
}

// preprocessCUDA forward (DGR forward.cu:151-269).  FMA contraction off: radii, tile rects and sort keys
// must match the oracle bit for bit.
// One Gaussian of the pass; returns false where the reference's kernel returns early.  The render record (4 float4), the cull
// record (2 float4) and the 3D covariance (6 floats) are handed back instead of being stored: the wave stores them together
// (see surfel_preprocess_kernel).
__device__ __forceinline__ bool
gauss_preprocess_one(int idx, int D, int M, const float* __restrict__ means, const float* __restrict__ scales, float scale_modifier,
                     const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
                     const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp, const float* __restrict__ normals,
                     const float* __restrict__ refl, const GaussCam& cam, int* __restrict__ radii, const GeomState& g, int gx, int gy, int prefiltered,
                     int antialiasing, float4* o, float* cov3D) {
#pragma clang fp contract(off)
	radii[idx] = 0;
	g.tiles_touched[idx] = 0;
	reinterpret_cast<uint2*>(g.rect)[idx] = make_uint2(0u, 0u);   // empty tile rectangle: emit_tiles_kernel takes the instance count from its area
	g.depths[idx] = __int_as_float(0x7f7fffff);   // culled: sorts behind every visible Gaussian in the depth pre-sort
	const float mx = means[3 * idx], my = means[3 * idx + 1], mz = means[3 * idx + 2];
	const float* vm = cam.view;
	const float* pm = cam.proj;
	const float pvz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
	if (pvz <= 0.2f) {  // in_frustum (DGR auxiliary.h:151-176)
		if (prefiltered) g.flags[0] = 1;
		return false;
	}
	const float hx = pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12];
	const float hy = pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13];
	const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
	const float p_w = 1.0f / (hw + 0.0000001f);
	const float projx = hx * p_w, projy = hy * p_w;

	if (cov3D_precomp != nullptr) {
#pragma unroll
		for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * idx + i];
	} else {
		cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3D);
	}
	const Cov2DCtx k = cov2d_ctx(mx, my, mz, cam);
	const M3 Vrk = m3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	const M3 cov = m3_mul(m3_mul(m3_T(k.T), m3_T(Vrk)), k.T);
	float cx = cov.m[0][0], cy = cov.m[0][1], cz = cov.m[1][1];
	const float h_var = 0.3f;
	const float det_cov = cx * cz - cy * cy;
	cx += h_var;
	cz += h_var;
	const float det = cx * cz - cy * cy;
	float h_convolution_scaling = 1.0f;
	if (antialiasing) h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det));
	if (det == 0.0f) return false;
	const float det_inv = 1.f / det;
	const float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
	const float mid = 0.5f * (cx + cz);
	const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
	const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
	const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
	// ndc2Pix (DGR auxiliary.h:40-43) is written in double there
	const float pix_x = (float)((((double)projx + 1.0) * cam.W - 1.0) * 0.5);
	const float pix_y = (float)((((double)projy + 1.0) * cam.H - 1.0) * 0.5);
	uint32_t x0, y0, x1, y1;
	get_rect(pix_x, pix_y, f2i(my_radius), gx, gy, x0, y0, x1, y1);
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
	radii[idx] = f2i(my_radius);
	g.rect[2 * idx] = x0 | (y0 << 16);
	g.rect[2 * idx + 1] = x1 | (y1 << 16);
	float4* rec = o;
	rec[0] = make_float4(pix_x, pix_y, conx, cony);
	rec[1] = make_float4(conz, opacities[idx] * h_convolution_scaling, cr, cg);
	rec[2] = make_float4(cb, normals[3 * idx], normals[3 * idx + 1], normals[3 * idx + 2]);
	rec[3] = make_float4(refl[idx], 1.0f / pvz, 0.f, 0.f);
	{
		// Cull record: alpha = opac * exp(-q/2) >= 1/255 needs q = d^T conic d <= q_max = 2 ln(255 opac); the record is
		// that ellipse (5 % + 0.1 margin on q_max) normalised to d^T E d <= 1.  No disc part for this variant.
		const float opac = opacities[idx] * h_convolution_scaling;
		float4 c0 = make_float4(pix_x, pix_y, 0.f, 0.f), c1 = make_float4(0.f, 0.f, 0.f, -2.0f);
		if (opac >= 1.0f / 255.0f) {
			const float inv_q = 1.0f / (2.0f * logf(255.0f * opac) * 1.05f + 0.1f);
			c0.z = conx * inv_q; c0.w = cony * inv_q; c1.x = conz * inv_q; c1.w = -1.0f;
		}
		o[G_REC_F4] = c0;
		o[G_REC_F4 + 1] = c1;
	}
	g.tiles_touched[idx] = (y1 - y0) * (x1 - x0);
	return true;
}

__global__ void __launch_bounds__(256)
gauss_preprocess_kernel(int P, int D, int M, const float* __restrict__ means, const float* __restrict__ scales, float scale_modifier,
                        const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
                        const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp, const float* __restrict__ normals,
                        const float* __restrict__ refl, GaussCam cam, int* __restrict__ radii, GeomState g, int gx, int gy, int prefiltered,
                        int antialiasing) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	// look-back state of the depth sort that follows (gsr_sort.hpp): cleared here instead of by a dispatch of its own
	sort_clear_region(g.depth_sort_temp, g.depth_sort_clear, (size_t)idx, (size_t)gridDim.x * 256u);
	sort_clear_region(g.emit_state, g.emit_state_bytes, (size_t)idx, (size_t)gridDim.x * 256u);   // look-back state of emit_tiles_kernel's scan
	if (idx == 0) { g.flags[1] = 0; g.flags[2] = 0; g.flags[3] = 0; }   // [1] workgroup tickets, [2,3] num_rendered (64 bits) of gaussian_stats_kernel
	__shared__ float4 s_out[4][6 * 65];     // per wave: 4 planes of 65 float4 (record), then 2 (cull record), then 6 planes of 65 dwords (cov3D)
	const int lane = threadIdx.x & 63;
	float4* so = s_out[threadIdx.x >> 6];
	float4 o[G_REC_F4 + 2];
	float cov3D[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int k = 0; k < G_REC_F4 + 2; k++) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
	bool live = false;
	if (idx < P)
		live = gauss_preprocess_one(idx, D, M, means, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp, normals, refl, cam,
		                            radii, g, gx, gy, prefiltered, antialiasing, o, cov3D);
	if (__ballot(live) == 0ull) return;      // (wave-uniform) nothing of this wave is ever read
	const int g0 = idx - lane;               // first Gaussian of the wave
	const int ng = min(64, P - g0);
	wave_store_rows4<G_REC_F4, false>(so, o, g.rec + (size_t)g0 * G_REC_F4, G_REC_F4, 0, ng, lane);
	wave_store_rows4<2, false>(so, o + G_REC_F4, g.bbox + (size_t)g0 * 2, 2, 0, ng, lane);
	// (round 4: the 3D covariance is not kept for the backward any more — 24 bytes written here and read there per Gaussian, for a value
	// that the backward recomputes from the scale and rotation rows it reads anyway, with the same instructions: cov3d_from_scale_rot)
}

// Per (pixel, Gaussian) falloff shared VERBATIM by the forward and backward tile kernels: the backward
// recovers T by dividing by (1 - alpha), so alpha must be the same bits in both.  Explicit fmaf + no
// further contraction makes the instruction sequence independent of the surrounding code.
// Returns the lanes whose pair passes the reference's two tests (power > 0, alpha < 1/255: DGR forward.cu:357-366) as a
// lane mask (see gsr_internal.hpp: explicit masks instead of bools keep hipcc from materialising ballots in VGPRs).
__device__ __forceinline__ lmask gauss_pair(float4 r0, float conz, float opac, float pixx, float pixy, float& dx, float& dy, float& G,
                                            float& alpha) {
#pragma clang fp contract(off)
	dx = r0.x - pixx;
	dy = r0.y - pixy;
	const float q = fmaf(r0.z * dx, dx, (conz * dy) * dy);
	const float power = fmaf(-0.5f, q, -((r0.w * dx) * dy));
	// straight-line (no early return): G and alpha are defined in every lane, the callers zero them where the pair is rejected
	G = __expf(power);
	alpha = fminf(0.99f, opac * G);
	return LMASK(!(power > 0.0f)) & LMASK(!(alpha < 1.0f / 255.0f));
}

// renderCUDA forward (DGR forward.cu:274-411), wave-per-quadrant form (see surfel_render_fwd_wave_kernel in
// gsr_surfel.hip for the design: one wave = one 8x8 pixel block, ballot-compacted private work list from
// conservative cull bounds, per-Gaussian record through the scalar memory path, no workgroup barriers).
#define G_WBATCH 64
#define G_SUB 16      // hits between two flushes of the backward's gradient slab (power of two)
template <bool INVDEPTH>
// The vote tests the cull ellipse against the box of the wave's pixel CENTRES, and alpha >= 1/255 is exactly q <= q_max for
// this variant (the screen-space blur is part of the conic), so the 5 % + 0.1 margin on q_max already makes the record
// conservative; the pad only covers the float rounding of the edge minimisation.
#define G_CULL_PAD 0.05f
__global__ void __launch_bounds__(64)
gauss_render_fwd_wave_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list, int W, int H, int tiles_x, int ntiles,
                             const float4* __restrict__ rec, const float4* __restrict__ bbox, int cull, const float* __restrict__ bg,
                             float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                             float* __restrict__ out_normal, float* __restrict__ out_refl, float* __restrict__ out_invdepth,
                             unsigned long long* __restrict__ blend_mask, size_t mask_stride) {
	const uint32_t slot = xcd_slot(blockIdx.x);   // dispatch slot -> (tile, quadrant), longest lists first
	if (slot >= (uint32_t)ntiles * 4u) return;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_order[slot >> 2]), quad = slot & 3u;   // (readfirstlane: the compiler cannot see that the loaded tile id is wave-uniform)
	const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
	const int lane = threadIdx.x;
	const int bx0 = tile_x * 16 + (quad & 1) * 8, by0 = tile_y * 16 + (quad >> 1) * 8;
	if (bx0 >= W || by0 >= H) return;
	const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
	const bool inside = px < W && py < H;
	const float pixx = (float)px, pixy = (float)py;
	const uint2 range = ranges[tile];
	const int count = (int)(range.y - range.x);
	const float qx0 = (float)bx0, qy0 = (float)by0, qx1 = qx0 + 7.0f, qy1 = qy0 + 7.0f;

	__shared__ uint32_t s_hid[G_WBATCH];
	__shared__ uint32_t s_hj[G_WBATCH];

	lmask done = ~LMASK(px < W) | ~LMASK(py < H);   // lanes outside the image never blend
	float T = 1.0f;
	uint32_t last_contributor = 0;
	float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, RS = 0, ID = 0;
	const size_t batch0 = (size_t)(range.x / G_WBATCH) + tile;     // where this tile's batches sit in blend_mask (see BinningState)

	for (int base = 0; base < count; base += G_WBATCH) {
		if (done == ~0ull) break;
		const int nb = min(G_WBATCH, count - base);
		bool hit = lane < nb;
		uint32_t id = 0;
		if (hit) {
			id = point_list[range.x + (uint32_t)(base + lane)];
			if (cull) {
				hit = cull_hit(bbox[2 * id], bbox[2 * id + 1], qx0 - G_CULL_PAD, qx1 + G_CULL_PAD, qy0 - G_CULL_PAD, qy1 + G_CULL_PAD);
			}
		}
		const unsigned long long mm = __ballot(hit);
		const int nh = __popcll(mm);
		if (nh == 0) continue;
		const int kown = __popcll(mm & ((1ull << lane) - 1ull));   // this lane's entry is hit number kown (if it is a hit)
		if (hit) {
			s_hid[kown] = id;
			s_hj[kown] = (uint32_t)lane;
		}
		__syncthreads();
		const uint32_t hid = lane < nh ? s_hid[lane] : 0u;
		const uint32_t hj = lane < nh ? s_hj[lane] : 0u;
		__syncthreads();
		unsigned long long blendk = 0ull;    // hits that blended into at least one pixel of the block
		// two SGPR record buffers ping-pong so that the next record's s_load stays in flight for a whole pair
		struct Rec { float4 r0, r1, r2, r3; };
		auto fetch = [&](int k) -> Rec {
			const float4* q = rec + (size_t)__builtin_amdgcn_readlane(hid, k) * G_REC_F4;
			return Rec{q[0], q[1], q[2], q[3]};
		};
		auto blend = [&](int k, const Rec& R, auto&& prefetch_next) -> bool {   // true: every pixel of the block has retired
			const uint32_t contributor = (uint32_t)(base + (int)__builtin_amdgcn_readlane(hj, k) + 1);
			// straight-line for all 64 lanes: a rejected pair blends with weight 0 (see surfel_render_fwd_wave_kernel)
			float dx, dy, G, alpha;
			const lmask live = gauss_pair(R.r0, R.r1.x, R.r1.y, pixx, pixy, dx, dy, G, alpha) & ~done;
			__builtin_amdgcn_sched_barrier(0);
			prefetch_next();
			__builtin_amdgcn_sched_barrier(0);
			const float test_T = T * (1 - alpha);
			const lmask sat = LMASK(test_T < 0.0001f) & live;   // saturated: the pair is dropped and the pixel retires
			const lmask ok = live & ~sat;
			done |= sat;
			if (ok != 0ull) {
				const float w = selm0(ok, alpha * T);
				C0 = fmaf(R.r1.z, w, C0); C1 = fmaf(R.r1.w, w, C1); C2 = fmaf(R.r2.x, w, C2);
				N0 = fmaf(R.r2.y, w, N0); N1 = fmaf(R.r2.z, w, N1); N2 = fmaf(R.r2.w, w, N2);
				RS = fmaf(R.r3.x, w, RS);
				if (INVDEPTH) ID = fmaf(R.r3.y, w, ID);
				T = selm(ok, test_T, T);
				last_contributor = selmu(ok, contributor, last_contributor);
				blendk |= 1ull << k;
			}
			return done == ~0ull;
		};
		Rec A = fetch(0), B = A;
		for (int k = 0; k < nh; k += 2) {
			if (blend(k, A, [&]() { if (k + 1 < nh) B = fetch(k + 1); })) break;
			if (k + 1 >= nh) break;
			if (blend(k + 1, B, [&]() { if (k + 2 < nh) A = fetch(k + 2); })) break;
		}
		// the batch's blend mask (bit = position in the batch): the backward tile kernel walks exactly these entries
		const lmask blended = __ballot(hit && ((blendk >> kown) & 1ull) != 0ull);
		if (lane == 0) blend_mask[(size_t)quad * mask_stride + batch0 + (size_t)(base / G_WBATCH)] = blended;
	}
	if (inside) {
		const size_t HW = (size_t)H * W;
		const size_t pix = (size_t)W * py + px;
		final_T[pix] = T;
		n_contrib[pix] = last_contributor;
		out_color[pix] = C0 + T * bg[0];
		out_color[HW + pix] = C1 + T * bg[1];
		out_color[2 * HW + pix] = C2 + T * bg[2];
		out_normal[pix] = N0;
		out_normal[HW + pix] = N1;
		out_normal[2 * HW + pix] = N2;
		out_refl[pix] = RS;
		if (INVDEPTH) out_invdepth[pix] = ID;
	}
}

// renderCUDA backward (DGR backward.cu:452-690), wave-per-quadrant form, shared list.  One 64-thread workgroup (= one wave)
// owns an 8x8 pixel block of a tile and walks the tile's list back to front in the forward's batches of 64 entries:
//   1. each lane takes one list entry and looks it up in the batch's blend mask (written by the forward: the entries that
//      blended into this block) -> ballot-compacted private work list: no footprint vote, no pair that cannot contribute;
//   2. the wave differentiates the survivors one at a time; the record of the current one is wave-uniform and arrives
//      through the scalar memory path into SGPRs (two buffers ping-pong), nothing is staged in LDS;
//   3. per contributing Gaussian the 16 gradient values (the reference: 16 float atomicAdds per (pixel, Gaussian) pair) are
//      reduced over each 16-lane row with exchange-type DPP (row_reduce16) and parked in the slab row of (hit, row);
//   4. every G_SUB hits the wave adds the four row totals and flushes them as one 64-byte row of float atomics per
//      Gaussian into acc[P][16].
// No workgroup barriers and no waiting for sibling quadrants (the reference synchronises the 256 threads of a tile twice
// per batch).  Variant S moved on to one list per 4x4 sub-block (surfel_render_bwd_rows_kernel); for this variant's much
// cheaper pair that form was measured and did not pay: the per-row blend masks cost the forward more than the shorter
// lists gave the backward (masks tracked with scalar instructions, 1 M Gaussians: fwd 0.265 -> 0.311 ms, bwd 0.534 -> 0.511 ms; 5 M: 0.44 ->
// 0.53, 0.84 -> 0.74; tracked per lane with two vector instructions per pair and an OR over the rows per batch: 1 M 0.263 -> 0.279,
// 0.529 -> 0.503; 5 M 0.45 -> 0.50, 0.84 -> 0.74; 100 k both kernels slower).
template <bool INVDEPTH>
__global__ void __launch_bounds__(64)
gauss_render_bwd_wave_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list, int W, int H, int tiles_x, int ntiles,
                             const float* __restrict__ bg, const float4* __restrict__ rec, const float4* __restrict__ bbox, int cull,
                             const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                             const float* __restrict__ dL_dnormal_map, const float* __restrict__ dL_drefl_map,
                             const float* __restrict__ dL_invdepths, float* __restrict__ acc, const unsigned long long* __restrict__ blend_mask,
                             size_t mask_stride) {
	const uint32_t slot = xcd_slot(blockIdx.x);   // dispatch slot -> (tile, quadrant), longest lists first
	if (slot >= (uint32_t)ntiles * 4u) return;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_order[slot >> 2]), quad = slot & 3u;   // (readfirstlane: the compiler cannot see that the loaded tile id is wave-uniform)
	const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
	const int lane = threadIdx.x;
	const int bx0 = tile_x * 16 + (quad & 1) * 8, by0 = tile_y * 16 + (quad >> 1) * 8;
	if (bx0 >= W || by0 >= H) return;
	const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
	const bool inside = px < W && py < H;
	const float pixx = (float)px, pixy = (float)py;
	const uint2 range = ranges[tile];
	const int count = (int)(range.y - range.x);
	const size_t HW = (size_t)H * W;
	const size_t pix = (size_t)W * py + px;
	__shared__ float s_slab[G_SUB * 4 * G_ACC_F];   // [hit in sub-batch][16-lane row][16 floats]
	__shared__ uint32_t s_hid[G_WBATCH];
	__shared__ uint32_t s_hj[G_WBATCH];

	const lmask inside_m = LMASK(px < W) & LMASK(py < H);
	// where this lane parks its row totals: quad q of a row holds value slot(q) of every reduced register (row_reduce_slot)
	const uint32_t slab_lane = (uint32_t)(lane >> 4) * G_ACC_F + (uint32_t)row_reduce_slot(lane);
	const float T_final = inside ? final_Ts[pix] : 0.f;
	float T = T_final;
	const int last_contributor = inside ? (int)n_contrib[pix] : 0;
	float dp0 = 0, dp1 = 0, dp2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, dr = 0, di = 0;
	if (inside) {
		dp0 = dL_dpixels[pix]; dp1 = dL_dpixels[HW + pix]; dp2 = dL_dpixels[2 * HW + pix];
		dn0 = dL_dnormal_map[pix]; dn1 = dL_dnormal_map[HW + pix]; dn2 = dL_dnormal_map[2 * HW + pix];
		dr = dL_drefl_map[pix];
		if (INVDEPTH) di = dL_invdepths[pix];
	}
	// The reference keeps accum_rec_x = last_alpha*last_x + (1-last_alpha)*accum_rec_x per output channel and adds
	// (x - accum_rec_x)*dL_dx to dL_dalpha.  Only dot products over channels are used and the recurrence is linear,
	// so two scalar recurrences replace the eight: A1 over D1 = <attributes, upstream grads> (-> dL_dalpha) and A2
	// over D2, the same dot with colour weights 3,2,1 and no other channels: the reference's dL_dalpha_means2d is the
	// running sum of the partial dL_dalpha INSIDE its colour loop (DGR backward.cu:613-614), i.e. 3 t0 + 2 t1 + t2.
	float A1 = 0, A2 = 0, D1p = 0, D2p = 0, last_alpha = 0;
	const float dp0x3 = 3.0f * dp0, dp1x2 = 2.0f * dp1;
	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
	const float bg_dot_dpixel = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;

	int wave_last = last_contributor;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off));
	wave_last = __builtin_amdgcn_readfirstlane(wave_last);   // (uniform after the butterfly; tells the compiler so)
	if (wave_last == 0) return;
	// back to front through the forward's batches; per batch the forward left the mask of the entries that blended into this
	// block: only those are differentiated (no footprint vote, no cull-record traffic, no pair that cannot contribute)
	const size_t batch0 = (size_t)(range.x / G_WBATCH) + tile;
	for (int b = (min(wave_last, count) - 1) / G_WBATCH; b >= 0; b--) {
		const unsigned long long bits = blend_mask[(size_t)quad * mask_stride + batch0 + (size_t)b];
		if (bits == 0ull) continue;
		const int pos = b * G_WBATCH + (G_WBATCH - 1 - lane);      // lane l looks at batch slot 63 - l: ascending lanes = descending positions
		const bool hit = ((bits >> (G_WBATCH - 1 - lane)) & 1ull) != 0ull && pos < wave_last;
		const unsigned long long mm = __ballot(hit);
		const int nh = __popcll(mm);
		if (nh == 0) continue;
		if (hit) {
			const int k = __popcll(mm & ((1ull << lane) - 1ull));
			s_hid[k] = point_list[range.x + (uint32_t)pos];
			s_hj[k] = (uint32_t)pos;
		}
		__syncthreads();
		const uint32_t hid = lane < nh ? s_hid[lane] : 0u;
		const uint32_t hj = lane < nh ? s_hj[lane] : 0u;
		unsigned long long touched = 0ull;
		struct Rec { float4 r0, r1, r2, r3; };
		auto fetch = [&](int k) -> Rec {
			const float4* q = rec + (size_t)__builtin_amdgcn_readlane(hid, k) * G_REC_F4;
			return Rec{q[0], q[1], q[2], q[3]};
		};
		auto differentiate = [&](int k, const Rec& R, auto&& prefetch_next) {
			const int contributor = (int)__builtin_amdgcn_readlane(hj, k);   // 0-based position in the tile's list
			float dx, dy, Gp, alpha_p;
			const lmask ok = gauss_pair(R.r0, R.r1.x, R.r1.y, pixx, pixy, dx, dy, Gp, alpha_p) & LMASK(contributor < last_contributor) & inside_m;
			__builtin_amdgcn_sched_barrier(0);
			prefetch_next();
			__builtin_amdgcn_sched_barrier(0);
			if (ok == 0ull) return;
			// Straight-line for all 64 lanes: a rejected pair runs with alpha = 0 (the identity of the recurrences) and
			// G = 0 (exp of a positive power may be inf), and has its root gradients zeroed, so every v[] comes out 0 without
			// exec-mask regions.
			float v[G_ACC_F];
			const float alpha = selm0(ok, alpha_p), G = selm0(ok, Gp);
			const float inv_1ma = div_nr(1.0f, 1.f - alpha);
			T *= inv_1ma;
			const float w = alpha * T;
			float D1 = R.r1.z * dp0 + R.r1.w * dp1 + R.r2.x * dp2 + R.r2.y * dn0 + R.r2.z * dn1 + R.r2.w * dn2 + R.r3.x * dr;
			if (INVDEPTH) D1 += R.r3.y * di;
			const float D2 = R.r1.z * dp0x3 + R.r1.w * dp1x2 + R.r2.x * dp2;
			A1 = last_alpha * D1p + (1.f - last_alpha) * A1;
			A2 = last_alpha * D2p + (1.f - last_alpha) * A2;
			D1p = D1; D2p = D2; last_alpha = alpha;
			const float bgterm = -T_final * inv_1ma * bg_dot_dpixel;
			const float dL_dalpha = selm0(ok, (D1 - A1) * T + bgterm);
			const float dL_dalpha_means2d = selm0(ok, (D2 - A2) * T + bgterm);
			v[GA_COLOR + 0] = w * dp0;
			v[GA_COLOR + 1] = w * dp1;
			v[GA_COLOR + 2] = w * dp2;
			v[GA_NORMAL + 0] = w * dn0;
			v[GA_NORMAL + 1] = w * dn1;
			v[GA_NORMAL + 2] = w * dn2;
			v[GA_REFL] = w * dr;
			v[GA_INVD] = INVDEPTH ? w * di : 0.f;
			const float dL_dG = R.r1.y * dL_dalpha;
			const float dL_dG_means2d = R.r1.y * dL_dalpha_means2d;
			const float gdx = G * dx, gdy = G * dy;
			const float dG_ddelx = (-gdx * R.r0.z - gdy * R.r0.w) * ddelx_dx;
			const float dG_ddely = (-gdy * R.r1.x - gdx * R.r0.w) * ddely_dy;
			v[GA_MEAN2D + 0] = dL_dG * dG_ddelx;
			v[GA_MEAN2D + 1] = dL_dG * dG_ddely;
			v[GA_MEAN2DP + 0] = dL_dG_means2d * dG_ddelx;
			v[GA_MEAN2DP + 1] = dL_dG_means2d * dG_ddely;
			const float hg = -0.5f * dL_dG;
			v[GA_CONIC + 0] = hg * gdx * dx;
			v[GA_CONIC + 1] = hg * gdx * dy;
			v[GA_CONIC + 2] = hg * gdy * dy;
			v[GA_OPAC] = G * dL_dalpha;
			// 16 values -> 4 registers of row totals (exchange-type DPP, row_reduce16); the four 16-lane rows park
			// theirs in separate slab rows and the flush adds them
			// (the four lanes of a quad hold the same totals and store them to the same address: cheaper than masking three off)
			float z[4];
			row_reduce16(v, z);
			float* slab = s_slab + (k & (G_SUB - 1)) * 4 * G_ACC_F + slab_lane;
#pragma unroll
			for (int g = 0; g < 4; g++) slab[4 * g] = z[g];
			touched |= 1ull << k;
		};
		auto flush = [&](int k_last) {
			const int k0 = k_last & ~(G_SUB - 1);
			__syncthreads();
			if (touched != 0ull) {
				const int n = (k_last - k0 + 1) * G_ACC_F;
				for (int item = lane; item < n; item += 64) {
					const int kk = item / G_ACC_F, d = item - kk * G_ACC_F;
					if ((touched >> (k0 + kk)) & 1ull) {
						const float* row = s_slab + kk * 4 * G_ACC_F + d;
						atomicAdd(acc + (size_t)s_hid[k0 + kk] * G_ACC_F + d, (row[0] + row[G_ACC_F]) + (row[2 * G_ACC_F] + row[3 * G_ACC_F]));
					}
				}
			}
			__syncthreads();
		};
		Rec A = fetch(0), B = A;
		for (int k = 0; k < nh; k += 2) {
			differentiate(k, A, [&]() { if (k + 1 < nh) B = fetch(k + 1); });
			if (k + 1 >= nh) { flush(k); break; }
			differentiate(k + 1, B, [&]() { if (k + 2 < nh) A = fetch(k + 2); });
			if (((k + 1) & (G_SUB - 1)) == G_SUB - 1 || k + 2 >= nh) flush(k + 1);
		}
	}
}

// computeCov2DCUDA + preprocessCUDA backward + computeCov3D backward fused into one per-Gaussian pass
// (DGR backward.cu:147-326, 330-393, 399-449).  Every output element is written (zeros for culled
// Gaussians), so the caller does not need the reference's 11 zero-filled tensors.
// ACC (round 4, as surfel_preprocess_bwd_kernel): the PARAMETER gradients (mean3D, sh, opacity, scale, rotation, normal, refl strength) are
// ADDED to the output tensors instead of written — several views accumulate into one gradient buffer on the device
// (gsr_gauss_backward_accum); the per-view outputs (dL_dmean2D_pixels, dL_dcolor, dL_dcov3D, the intermediates) are written either way.
template <bool ACC>
__global__ void __launch_bounds__(256)
gauss_preprocess_bwd_kernel(int P, int D, int M, const float* __restrict__ means, const int* __restrict__ radii, const float* __restrict__ shs,
                            const uint8_t* __restrict__ clamped, const float* __restrict__ opacities, const float* __restrict__ scales,
                            const float* __restrict__ rotations, float scale_modifier, const float* __restrict__ cov3Ds, GaussCam cam,
                            const float* __restrict__ acc, int has_invdepth, int antialiasing, float* __restrict__ dL_dmean2D,
                            float* __restrict__ dL_dmean2D_pixels, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
                            float* __restrict__ dL_dcolor, float* __restrict__ dL_dnormals, float* __restrict__ dL_drefl,
                            float* __restrict__ dL_dinvdepth, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
                            float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
	// The (P,3) / (P,6) / (P,16,3) output rows are stored by the wave together (wave_store_rows, gsr_internal.hpp).
	__shared__ __attribute__((aligned(16))) float s_tile[4][1048];     // per wave: 6 planes of 65 dwords or 4 planes of 65 float4
	const int lane = threadIdx.x & 63;
	float* tile = s_tile[threadIdx.x >> 6];
	const int idx_raw = blockIdx.x * 256 + threadIdx.x;
	const int g0 = idx_raw - lane, nrows = min(64, P - g0);      // the wave's first row, its rows inside the arrays
	if (nrows <= 0) return;                                      // (wave-uniform)
	const bool in_range = idx_raw < P;
	const int idx = in_range ? idx_raw : P - 1;                  // lanes past the end recompute the last row; their rows are not stored
	const float4* a4 = reinterpret_cast<const float4*>(acc + (size_t)idx * G_ACC_F);
	const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3];
	// pass-through outputs of the tile kernel
	if (in_range) {
		put<ACC>(dL_drefl + idx, a1.z);
		if (has_invdepth) dL_dinvdepth[idx] = a1.w;
		if (dL_dconic != nullptr) reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(a3.x, a3.y, 0.f, a3.z);
	}
	{
		const float c3[3] = {a0.x, a0.y, a0.z}, n3[3] = {a0.w, a1.x, a1.y}, m3[3] = {a2.x, a2.y, 0.f}, p3[3] = {a2.z, a2.w, 0.f};
		// (outputs the caller did not ask for — NULL — are not written: the tile kernel's intermediate dL_dmean2D / dL_dconic, which the
		// reference's binding never returns, and the gradients of inputs that were not supplied)
		if (dL_dcolor != nullptr) wave_store_rows<3, false>(tile, c3, dL_dcolor + (size_t)g0 * 3, nrows, lane);
		wave_store_rows<3, ACC>(tile, n3, dL_dnormals + (size_t)g0 * 3, nrows, lane);
		if (dL_dmean2D != nullptr) wave_store_rows<3, false>(tile, m3, dL_dmean2D + (size_t)g0 * 3, nrows, lane);
		wave_store_rows<3, false>(tile, p3, dL_dmean2D_pixels + (size_t)g0 * 3, nrows, lane);
	}
	float dL_dopac = a3.w;

	float dmean[3] = {0.f, 0.f, 0.f};
	float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float dscale[3] = {0.f, 0.f, 0.f};
	float drot[4] = {0.f, 0.f, 0.f, 0.f};
	const bool visible = radii[idx] > 0;
	const float mx = means[3 * idx], my = means[3 * idx + 1], mz = means[3 * idx + 2];
	if (visible) {
		// ---- computeCov2DCUDA
		float c3[6];
		if (cov3Ds != nullptr) {       // supplied by the caller (cov3D_precomp)
#pragma unroll
			for (int i = 0; i < 6; i++) c3[i] = cov3Ds[6 * idx + i];
		} else {                       // the forward's value, recomputed (no contraction in there: the same bits)
			cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, c3);
		}
		const Cov2DCtx k = cov2d_ctx(mx, my, mz, cam);
		const float x_grad_mul = (k.txtz < -k.limx || k.txtz > k.limx) ? 0.f : 1.f;
		const float y_grad_mul = (k.tytz < -k.limy || k.tytz > k.limy) ? 0.f : 1.f;
		const M3 Vrk = m3_make(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
		const M3& T = k.T;
		const M3& Wm = k.Wm;
		const M3 cov2D = m3_mul(m3_mul(m3_T(T), m3_T(Vrk)), T);
		float c_xx = cov2D.m[0][0], c_xy = cov2D.m[0][1], c_yy = cov2D.m[1][1];
		const float h_var = 0.3f;
		float d_inside_root = 0.f;
		if (antialiasing) {
			const float det_cov = c_xx * c_yy - c_xy * c_xy;
			c_xx += h_var;
			c_yy += h_var;
			const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
			const float h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
			const float dL_dopacity_v = dL_dopac;
			const float d_h_convolution_scaling = dL_dopacity_v * opacities[idx];
			dL_dopac = dL_dopacity_v * h_convolution_scaling;
			d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
		} else {
			c_xx += h_var;
			c_yy += h_var;
		}
		float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
		if (antialiasing) {
			const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
			const float dn = (w * w + w * (x + y) + x * y - z * z);
			const float denom_f = d_inside_root / (dn * dn);
			dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
			dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
			dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
		}
		const float dcx = a3.x, dcy = a3.y, dcz = a3.z;  // dL_dconic slots x, y, w
		const float denom = c_xx * c_yy - c_xy * c_xy;
		const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
		if (denom2inv != 0) {
			dL_dc_xx += denom2inv * (-c_yy * c_yy * dcx + 2 * c_xy * c_yy * dcy + (denom - c_xx * c_yy) * dcz);
			dL_dc_yy += denom2inv * (-c_xx * c_xx * dcz + 2 * c_xx * c_xy * dcy + (denom - c_xx * c_yy) * dcx);
			dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dcx - (denom + 2 * c_xy * c_xy) * dcy + c_xx * c_xy * dcz);
			dcov[0] = (T.m[0][0] * T.m[0][0] * dL_dc_xx + T.m[0][0] * T.m[1][0] * dL_dc_xy + T.m[1][0] * T.m[1][0] * dL_dc_yy);
			dcov[3] = (T.m[0][1] * T.m[0][1] * dL_dc_xx + T.m[0][1] * T.m[1][1] * dL_dc_xy + T.m[1][1] * T.m[1][1] * dL_dc_yy);
			dcov[5] = (T.m[0][2] * T.m[0][2] * dL_dc_xx + T.m[0][2] * T.m[1][2] * dL_dc_xy + T.m[1][2] * T.m[1][2] * dL_dc_yy);
			dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_dc_xx + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_dc_xy + 2 * T.m[1][0] * T.m[1][1] * dL_dc_yy;
			dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_dc_xx + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_dc_xy + 2 * T.m[1][0] * T.m[1][2] * dL_dc_yy;
			dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_dc_xx + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_dc_xy + 2 * T.m[1][1] * T.m[1][2] * dL_dc_yy;
		}
		const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_dc_xx +
		                      (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc_xy;
		const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_dc_xx +
		                      (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc_xy;
		const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_dc_xx +
		                      (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc_xy;
		const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc_yy +
		                      (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_dc_xy;
		const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc_yy +
		                      (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_dc_xy;
		const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc_yy +
		                      (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_dc_xy;
		const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
		const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
		const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
		const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
		const float tz = 1.f / k.tz, tz2 = tz * tz, tz3 = tz2 * tz;
		const float h_x = cam.focal_x, h_y = cam.focal_y;
		const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * k.tx) * tz3 * dL_dJ02 + (2 * h_y * k.ty) * tz3 * dL_dJ12;
		if (has_invdepth) dL_dtz -= a1.w / (k.tz * k.tz);
		const float* vm = cam.view;
		dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;   // assignment (DGR backward.cu:325)
		dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
		dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

		// ---- preprocessCUDA backward: projection Jacobian of the 2-D mean
		const float* proj = cam.proj;
		const float m_hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
		const float m_w = 1.0f / (m_hw + 0.0000001f);
		const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
		const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
		const float g2x = a2.x, g2y = a2.y;
		dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
		dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
		dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

		// ---- computeCov3D backward (unnormalised quaternion, DGR backward.cu:392)
		if (scales != nullptr) {
			const float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
			const M3 R = m3_make(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
			                     1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
			                     1.f - 2.f * (x * x + y * y));
			const float sx = scale_modifier * scales[3 * idx], sy = scale_modifier * scales[3 * idx + 1], sz = scale_modifier * scales[3 * idx + 2];
			M3 S = m3_make(sx, 0, 0, 0, sy, 0, 0, 0, sz);
			M3 Mm = m3_mul(S, R);
			M3 dSigma = m3_make(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
#pragma unroll
			for (int c = 0; c < 3; c++)
#pragma unroll
				for (int n = 0; n < 3; n++) Mm.m[c][n] *= 2.0f;
			const M3 dL_dM = m3_mul(Mm, dSigma);
			const M3 Rt = m3_T(R);
			M3 dMt = m3_T(dL_dM);
			dscale[0] = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
			dscale[1] = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
			dscale[2] = Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2];
#pragma unroll
			for (int n = 0; n < 3; n++) { dMt.m[0][n] *= sx; dMt.m[1][n] *= sy; dMt.m[2][n] *= sz; }
			drot[0] = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
			drot[1] = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
			drot[2] = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
			drot[3] = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
		}
	}
	if (in_range) put<ACC>(dL_dopacity + idx, dL_dopac);
	// ---- SH backward (also writes the dL_dsh row, zeros when not visible)
	if (shs != nullptr) {
		if (M == 16) {
			// the row is the outer product w x dL_dRGB (gsr_math.hpp): three passes of one 64-byte sector per row
			float w[16];
			F3 grgb = f3(0.f, 0.f, 0.f);
#pragma unroll
			for (int k = 0; k < 16; k++) w[k] = 0.f;
			// (a Gaussian that blended into no pixel has a zero colour gradient: its SH row of the gradient and the view-direction term
			// are zeros whatever the coefficients are, so their 192 bytes are not read)
			if (visible && (a0.x != 0.f || a0.y != 0.f || a0.z != 0.f)) {
				ShRow s;
				load_sh(shs, idx, M, (D + 1) * (D + 1), s);
				const F3 dir = f3(mx - cam.campos[0], my - cam.campos[1], mz - cam.campos[2]);
				grgb = f3(a0.x, a0.y, a0.z);
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
				const F3 dm = sh_backward<ACC>(idx, D, M, s, dir, clamped[idx], f3(a0.x, a0.y, a0.z), dL_dsh);
				dmean[0] += dm.x; dmean[1] += dm.y; dmean[2] += dm.z;
			} else if (!ACC) {
				float* out = dL_dsh + (size_t)idx * M * 3;
				for (int q = 0; q < M * 3; q++) out[q] = 0.f;
			}
		}
	}
	wave_store_rows<3, ACC>(tile, dmean, dL_dmean3D + (size_t)g0 * 3, nrows, lane);
	if (dL_dcov3D != nullptr) wave_store_rows<6, false>(tile, dcov, dL_dcov3D + (size_t)g0 * 6, nrows, lane);
	wave_store_rows<3, ACC>(tile, dscale, dL_dscale + (size_t)g0 * 3, nrows, lane);
	if (in_range) put4<ACC>(reinterpret_cast<float4*>(dL_drot) + idx, drot[0], drot[1], drot[2], drot[3]);
}

}  // namespace gsr

using namespace gsr;

static GaussCam make_cam(const float* view, const float* proj, const float* campos, int W, int H, float tan_fovx, float tan_fovy) {
	GaussCam c;
	c.view = view; c.proj = proj; c.campos = campos; c.W = W; c.H = H;
	c.tan_fovx = tan_fovx; c.tan_fovy = tan_fovy;
	c.focal_y = H / (2.0f * tan_fovy);   // DGR rasterizer_impl.cu:228-229
	c.focal_x = W / (2.0f * tan_fovx);
	return c;
}

extern "C" int gsr_gauss_forward(gsr_alloc_fn alloc, void* alloc_user, int P, int D, int M, const float* background, int width, int height,
                                 const float* means3D, const float* shs, const float* colors_precomp, const float* normals,
                                 const float* refl_strengths, const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                 const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_normal_map,
                                 float* out_refl_strength_map, float* out_invdepth, int antialiasing, int* radii, int debug, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (!alloc || P < 0 || width <= 0 || height <= 0 || !background || !out_color || !out_normal_map || !out_refl_strength_map) {
		set_error("gsr_gauss_forward: invalid argument");
		return GSR_E_INVALID;
	}
	const size_t HW = (size_t)width * height;
	if (P == 0) {  // reference: outputs stay zero, num_rendered = 0 (DGR rasterize_points.cu:99-100)
		GSR_HIP_CHECK(hipMemsetAsync(out_color, 0, HW * 3 * 4, stream));
		GSR_HIP_CHECK(hipMemsetAsync(out_normal_map, 0, HW * 3 * 4, stream));
		GSR_HIP_CHECK(hipMemsetAsync(out_refl_strength_map, 0, HW * 4, stream));
		if (out_invdepth) GSR_HIP_CHECK(hipMemsetAsync(out_invdepth, 0, HW * 4, stream));
		return 0;
	}
	if (!means3D || !opacities || !normals || !refl_strengths || !viewmatrix || !projmatrix || !cam_pos || !radii ||
	    (!shs && !colors_precomp) || ((!scales || !rotations) && !cov3D_precomp)) {
		set_error("gsr_gauss_forward: missing required input pointer");
		return GSR_E_INVALID;
	}
	if (D < 0 || D > 3 || (shs && (D + 1) * (D + 1) > M)) { set_error("gsr_gauss_forward: SH degree %d not supported with M=%d", D, M); return GSR_E_INVALID; }
	if (shs && ((M * 3) & 3) == 0) GSR_REQUIRE_ALIGNED16(shs, "shs (rows of a multiple of 16 bytes)");
	const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
	const int ntiles = tiles_x * tiles_y;

	size_t geom_bytes = 0, img_bytes = 0;
	const size_t scan_bytes = scan_temp_bytes(P);
	carve_geom(nullptr, P, G_REC_F4, 0, G_ACC_F, scan_bytes, &geom_bytes);
	carve_image(nullptr, HW, ntiles, 1, 1, &img_bytes);
	void* gbuf = alloc(alloc_user, GSR_BUF_GEOM, geom_bytes);
	void* ibuf = alloc(alloc_user, GSR_BUF_IMAGE, img_bytes);
	if (!gbuf || !ibuf) { set_error("workspace allocation failed (%zu / %zu bytes)", geom_bytes, img_bytes); return GSR_E_ALLOC; }
	GeomState geom = carve_geom(gbuf, P, G_REC_F4, 0, G_ACC_F, scan_bytes, nullptr);
	ImageState img = carve_image(ibuf, HW, ntiles, 1, 1, nullptr);

	if (prefiltered) GSR_HIP_CHECK(hipMemsetAsync(geom.flags, 0, 4 * sizeof(int), stream));   // the flag is only written and read then
	const GaussCam cam = make_cam(viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy);
{ StageTimer st_(GSR_STAGE_PREPROCESS, stream); 	gauss_preprocess_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
	                                                             colors_precomp, normals, refl_strengths, cam, radii, geom, tiles_x, tiles_y,
	                                                             prefiltered, antialiasing); }
	GSR_LAUNCH_CHECK(debug, stream);

	BinningState bin;
	const int R = run_binning(alloc, alloc_user, P, tiles_x, tiles_y, geom, img, &bin, prefiltered, debug, stream);
	if (R < 0) return R;

	const int nunits = (int)xcd_grid((uint32_t)ntiles * 4u);
	{
		StageTimer st_(GSR_STAGE_RENDER_FWD, stream);
		if (out_invdepth)
			gauss_render_fwd_wave_kernel<true><<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, geom.rec, geom.bbox,
			                                                              option_cull(), background, img.final_T, img.n_contrib, out_color, out_normal_map,
			                                                              out_refl_strength_map, out_invdepth, bin.blend_mask, bin.mask_stride);
		else
			gauss_render_fwd_wave_kernel<false><<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, geom.rec, geom.bbox,
			                                                               option_cull(), background, img.final_T, img.n_contrib, out_color, out_normal_map,
			                                                               out_refl_strength_map, nullptr, bin.blend_mask, bin.mask_stride);
	}
	GSR_LAUNCH_CHECK(debug, stream);
	return R;
}

extern "C" int gsr_gauss_backward_accum(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                  const float* shs, const float* colors_precomp, const float* normals, const float* refl_strengths,
                                  const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                  float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer, void* binning_buffer, void* image_buffer,
                                  const float* dL_dpix, const float* dL_dnormal_map, const float* dL_drefl_strength_map,
                                  const float* dL_invdepths, float* dL_dmean2D, float* dL_dmean2D_pixels, float* dL_dconic, float* dL_dopacity,
                                  float* dL_dcolor, float* dL_dnormals, float* dL_drefl_strengths, float* dL_dinvdepth, float* dL_dmean3D,
                                  float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int antialiasing, int accumulate, int debug, void* stream_) {
	(void)colors_precomp; (void)normals; (void)refl_strengths;
	hipStream_t stream = (hipStream_t)stream_;
	if (P < 0 || R < 0 || width <= 0 || height <= 0) { set_error("gsr_gauss_backward: invalid size"); return GSR_E_INVALID; }
	if (P == 0) return 0;
	if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer) || !dL_dpix || !dL_dnormal_map || !dL_drefl_strength_map ||
	    !dL_dmean2D_pixels || !dL_dopacity || (!shs && !dL_dcolor) || !dL_dnormals || !dL_drefl_strengths || !dL_dmean3D || (!scales && !dL_dcov3D) ||
	    !dL_dscale || !dL_drot || (shs && !dL_dsh) || (dL_invdepths && !dL_dinvdepth) || !radii || !means3D || !opacities) {
		set_error("gsr_gauss_backward: missing required pointer");
		return GSR_E_INVALID;
	}
	if (shs && ((M * 3) & 3) == 0) GSR_REQUIRE_ALIGNED16(shs, "shs (rows of a multiple of 16 bytes)");
	if (shs && ((M * 3) & 3) == 0) GSR_REQUIRE_ALIGNED16(dL_dsh, "dL_dsh");
	GSR_REQUIRE_ALIGNED16(dL_drot, "dL_drot");
	GSR_REQUIRE_ALIGNED16(dL_dconic, "dL_dconic");
	const size_t HW = (size_t)width * height;
	const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
	const int ntiles = tiles_x * tiles_y;
	GeomState geom = carve_geom(geom_buffer, P, G_REC_F4, 0, G_ACC_F, scan_temp_bytes(P), nullptr);
	ImageState img = carve_image(image_buffer, HW, ntiles, 1, 1, nullptr);
	BinningState bin = carve_binning(binning_buffer, R, ntiles, 0, nullptr);

	GSR_HIP_CHECK(hipMemsetAsync(geom.acc, 0, (size_t)P * G_ACC_F * sizeof(float), stream));
	if (R > 0) {
		const int nunits = (int)xcd_grid((uint32_t)ntiles * 4u);
		{
			StageTimer st_(GSR_STAGE_RENDER_BWD, stream);
			if (dL_invdepths)
				gauss_render_bwd_wave_kernel<true><<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, background, geom.rec,
				                                                              geom.bbox, option_cull(), img.final_T, img.n_contrib, dL_dpix, dL_dnormal_map,
				                                                              dL_drefl_strength_map, dL_invdepths, geom.acc, bin.blend_mask, bin.mask_stride);
			else
				gauss_render_bwd_wave_kernel<false><<<nunits, 64, 0, stream>>>(img.ranges, img.tile_order, bin.point_list, width, height, tiles_x, ntiles, background, geom.rec,
				                                                               geom.bbox, option_cull(), img.final_T, img.n_contrib, dL_dpix, dL_dnormal_map,
				                                                               dL_drefl_strength_map, nullptr, geom.acc, bin.blend_mask, bin.mask_stride);
		}
		GSR_LAUNCH_CHECK(debug, stream);
	}
	const GaussCam cam = make_cam(viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy);
	const float* cov3D_ptr = cov3D_precomp;      // NULL: the kernel recomputes it from scales / rotations
{ StageTimer st_(GSR_STAGE_PREPROCESS_BWD, stream);
	auto kern = accumulate ? gauss_preprocess_bwd_kernel<true> : gauss_preprocess_bwd_kernel<false>;
	kern<<<(P + 255) / 256, 256, 0, stream>>>(P, D, M, means3D, radii, shs, geom.clamped, opacities, scales, rotations,
	                                                                 scale_modifier, cov3D_ptr, cam, geom.acc, dL_invdepths ? 1 : 0, antialiasing,
	                                                                 dL_dmean2D, dL_dmean2D_pixels, dL_dconic, dL_dopacity, dL_dcolor, dL_dnormals,
	                                                                 dL_drefl_strengths, dL_dinvdepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
	                                                                 dL_drot); }
	GSR_LAUNCH_CHECK(debug, stream);
	return 0;
}

extern "C" int gsr_gauss_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                  const float* shs, const float* colors_precomp, const float* normals, const float* refl_strengths,
                                  const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                  float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer, void* binning_buffer, void* image_buffer,
                                  const float* dL_dpix, const float* dL_dnormal_map, const float* dL_drefl_strength_map,
                                  const float* dL_invdepths, float* dL_dmean2D, float* dL_dmean2D_pixels, float* dL_dconic, float* dL_dopacity,
                                  float* dL_dcolor, float* dL_dnormals, float* dL_drefl_strengths, float* dL_dinvdepth, float* dL_dmean3D,
                                  float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int antialiasing, int debug, void* stream_) {
	return gsr_gauss_backward_accum(P, D, M, R, background, width, height, means3D, shs, colors_precomp, normals, refl_strengths, opacities, scales,
	                                scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geom_buffer,
	                                binning_buffer, image_buffer, dL_dpix, dL_dnormal_map, dL_drefl_strength_map, dL_invdepths, dL_dmean2D,
	                                dL_dmean2D_pixels, dL_dconic, dL_dopacity, dL_dcolor, dL_dnormals, dL_drefl_strengths, dL_dinvdepth, dL_dmean3D,
	                                dL_dcov3D, dL_dsh, dL_dscale, dL_drot, antialiasing, 0, debug, stream_);
}
