// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.hpp header).
//
// Variant S: CPU restatement of submodules/diff-surfel-rasterization (DSR) —
//   cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h and the tensor
//   plumbing of rasterize_points.cu.  This is the rasterizer gaussian_renderer/__init__.py:14,130
//   actually calls.  Parity unpinned by reference artefacts (no tests, not compilable here);
//   pinned by float64 finite differences + known-answer tests in tests/.
//
// Deliberately reproduced quirks:
//   * forward `unstable = |p.z| < 1e-4` (forward.cu:373) vs backward `< 1e-6` (backward.cu:302)
//   * backward rebuilds T with scale_to_mat(scale, 1.0f) — scale_modifier ignored (backward.cu:511)
//   * backward W,H = int(focal*tan*2) (backward.cu:637-638)
//   * dL_dmean2D.xy overwritten with the densification signal (backward.cu:656-659)
//   * median contributor stored by float->uint conversion of a float initialised to -1
//   * rsqrtf in quat_to_rotmat is restated as 1/sqrt (CUDA's rsqrtf is a 2-ulp approximation)
//   * gaussian_weights: the reference's check-then-atomicExch max is racy; the oracle computes
//     the true max (an upper bound of anything the reference can return)
#include "oracle_common.hpp"

namespace orc {

static const float near_n = 0.2f;          // DSR auxiliary.h:41-44 (float consts initialised from double literals)
static const float far_n = 100.0f;
static const float FilterSize = (float)0.707106;
static const float FilterInvSquare = 2.0f;

template <class R> struct SurfelIn {
	int P, D, M, W, H;
	const R *bg, *means3D;
	const uint8_t* env_scope_mask;
	const R *shs, *colors_precomp, *refl, *opacities, *scales, *rotations, *transMat_precomp;
	const R *view, *proj, *campos;
	R scale_modifier, tan_fovx, tan_fovy;
	bool prefiltered;
};

template <class R> struct SurfelState {
	int P = 0, W = 0, H = 0, gx = 0, gy = 0;
	std::vector<R> depths, means2D, transMat, normal_opacity, rgb;
	std::vector<uint8_t> clamped;
	std::vector<int> radii;
	std::vector<uint32_t> tiles_touched;
	Binning bin;
	std::vector<R> final_T;            // 3 planes: T, M1, M2
	std::vector<uint32_t> n_contrib;   // 2 planes: last, median
	bool trap = false;
};

// DSR auxiliary.h:217-239
template <class R> static M3<R> quat_to_rotmat(const R* quat) {
	// glm::vec4 quat = (q.x,q.y,q.z,q.w) = tensor order (r,x,y,z)
	R s = R(1) / std::sqrt(quat[3] * quat[3] + quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2]);
	R w = quat[0] * s, x = quat[1] * s, y = quat[2] * s, z = quat[3] * s;
	return mat3<R>(R(1) - R(2) * (y * y + z * z), R(2) * (x * y + w * z), R(2) * (x * z - w * y),
	               R(2) * (x * y - w * z), R(1) - R(2) * (x * x + z * z), R(2) * (y * z + w * x),
	               R(2) * (x * z + w * y), R(2) * (y * z - w * x), R(1) - R(2) * (x * x + y * y));
}

// DSR auxiliary.h:242-286
template <class R> static void quat_to_rotmat_vjp(const R* quat, const M3<R>& v_R, R* v_quat) {
	R s = R(1) / std::sqrt(quat[3] * quat[3] + quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2]);
	R w = quat[0] * s, x = quat[1] * s, y = quat[2] * s, z = quat[3] * s;
	v_quat[0] = R(2) * (x * (v_R[1][2] - v_R[2][1]) + y * (v_R[2][0] - v_R[0][2]) + z * (v_R[0][1] - v_R[1][0]));
	v_quat[1] = R(2) * (R(-2) * x * (v_R[1][1] + v_R[2][2]) + y * (v_R[0][1] + v_R[1][0]) + z * (v_R[0][2] + v_R[2][0]) + w * (v_R[1][2] - v_R[2][1]));
	v_quat[2] = R(2) * (x * (v_R[0][1] + v_R[1][0]) - R(2) * y * (v_R[0][0] + v_R[2][2]) + z * (v_R[1][2] + v_R[2][1]) + w * (v_R[2][0] - v_R[0][2]));
	v_quat[3] = R(2) * (x * (v_R[0][2] + v_R[2][0]) + y * (v_R[1][2] + v_R[2][1]) - R(2) * z * (v_R[0][0] + v_R[1][1]) + w * (v_R[0][1] - v_R[1][0]));
}

// DSR auxiliary.h:289-296
template <class R> static M3<R> scale_to_mat(const R* scale, R glob_scale) {
	M3<R> S = mat3<R>(1, 0, 0, 0, 1, 0, 0, 0, 1);
	S[0][0] = glob_scale * scale[0];
	S[1][1] = glob_scale * scale[1];
	return S;
}

template <class R> static Mat<R, 4, 4> world2ndc_of(const R* pm) {
	// glm::mat4(p0,p4,p8,p12, p1,p5,p9,p13, ...) : column c = (p[c], p[c+4], p[c+8], p[c+12])
	Mat<R, 4, 4> m;
	for (int c = 0; c < 4; c++)
		for (int r = 0; r < 4; r++) m[c][r] = pm[c + 4 * r];
	return m;
}
template <class R> static Mat<R, 3, 4> ndc2pix_of(int W, int H) {
	Mat<R, 3, 4> m;
	m[0][0] = (R)(float(W) / 2.0); m[0][1] = 0; m[0][2] = 0; m[0][3] = (R)(float(W - 1) / 2.0);
	m[1][0] = 0; m[1][1] = (R)(float(H) / 2.0); m[1][2] = 0; m[1][3] = (R)(float(H - 1) / 2.0);
	m[2][0] = 0; m[2][1] = 0; m[2][2] = 0; m[2][3] = 1;
	return m;
}

// DSR forward.cu:75-115
template <class R>
static void compute_transmat(V3<R> p_orig, const R* scale, R mod, const R* rot, const R* projmatrix, const R* viewmatrix, int W, int H,
                             M3<R>& T, V3<R>& normal) {
	M3<R> Rm = quat_to_rotmat(rot);
	M3<R> S = scale_to_mat(scale, mod);
	M3<R> L = mul(Rm, S);
	Mat<R, 3, 4> splat2world;
	for (int r = 0; r < 3; r++) { splat2world[0][r] = L[0][r]; splat2world[1][r] = L[1][r]; }
	splat2world[0][3] = 0; splat2world[1][3] = 0;
	splat2world[2][0] = p_orig.x; splat2world[2][1] = p_orig.y; splat2world[2][2] = p_orig.z; splat2world[2][3] = 1;
	Mat<R, 4, 4> world2ndc = world2ndc_of(projmatrix);
	Mat<R, 3, 4> ndc2pix = ndc2pix_of<R>(W, H);
	T = mul(mul(transpose(splat2world), world2ndc), ndc2pix);
	normal = transformVec4x3(V3<R>{L[2][0], L[2][1], L[2][2]}, viewmatrix);
}

// DSR forward.cu:119-145
template <class R> static bool compute_aabb(const M3<R>& T, R cutoff, V2<R>& point_image, V2<R>& extent) {
	V3<R> t = {cutoff * cutoff, cutoff * cutoff, R(-1)};
	R d = dot(t, col(T, 2) * col(T, 2));
	if (d == R(0)) return false;
	V3<R> f = (R(1) / d) * t;
	V2<R> p = {dot(f, col(T, 0) * col(T, 2)), dot(f, col(T, 1) * col(T, 2))};
	V2<R> h0 = {p.x * p.x - dot(f, col(T, 0) * col(T, 0)), p.y * p.y - dot(f, col(T, 1) * col(T, 1))};
	V2<R> h = {std::sqrt(std::max(R(1e-4f), h0.x)), std::sqrt(std::max(R(1e-4f), h0.y))};
	point_image = p;
	extent = h;
	return true;
}

// preprocessCUDA forward: DSR forward.cu:149-253
template <class R> static void preprocess_fwd(const SurfelIn<R>& in, SurfelState<R>& st) {
	const int P = in.P;
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++) {
		st.radii[idx] = 0;
		st.tiles_touched[idx] = 0;
		V3<R> p_view;
		bool trap = false;
		if (!in_frustum(idx, in.means3D, in.view, in.proj, in.prefiltered, p_view, trap)) {
			if (trap) st.trap = true;
			continue;
		}
		M3<R> T;
		V3<R> normal;
		if (in.transMat_precomp == nullptr) {
			V3<R> p_orig = {in.means3D[3 * idx], in.means3D[3 * idx + 1], in.means3D[3 * idx + 2]};
			compute_transmat(p_orig, in.scales + 2 * idx, in.scale_modifier, in.rotations + 4 * idx, in.proj, in.view, in.W, in.H, T, normal);
			for (int c = 0; c < 3; c++)
				for (int r = 0; r < 3; r++) st.transMat[9 * idx + 3 * c + r] = T[c][r];
		} else {
			for (int c = 0; c < 3; c++)
				for (int r = 0; r < 3; r++) T[c][r] = in.transMat_precomp[9 * idx + 3 * c + r];
			normal = {R(0), R(0), R(1)};
		}
		// DUAL_VISIABLE (forward.cu:211-216)
		R cosv = -((p_view.x * normal.x) + (p_view.y * normal.y) + (p_view.z * normal.z));
		if (cosv == R(0)) continue;
		R multiplier = cosv > 0 ? R(1) : R(-1);
		normal = multiplier * normal;
		R cutoff = R(3.0f);
		V2<R> point_image, extent;
		if (!compute_aabb(T, cutoff, point_image, extent)) continue;
		R radius = std::ceil(std::max(std::max(extent.x, extent.y), cutoff * R(FilterSize)));
		uint32_t rmin[2], rmax[2];
		getRect(point_image, f2i_sat(radius), rmin, rmax, st.gx, st.gy);
		if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
		if (in.colors_precomp == nullptr) {
			V3<R> c = sh_forward(idx, in.D, in.M, in.means3D, in.campos, in.shs, st.clamped.data());
			st.rgb[idx * 3 + 0] = c.x; st.rgb[idx * 3 + 1] = c.y; st.rgb[idx * 3 + 2] = c.z;
		}
		st.depths[idx] = p_view.z;
		st.radii[idx] = f2i_sat(radius);
		st.means2D[2 * idx] = point_image.x;
		st.means2D[2 * idx + 1] = point_image.y;
		st.normal_opacity[4 * idx + 0] = normal.x;
		st.normal_opacity[4 * idx + 1] = normal.y;
		st.normal_opacity[4 * idx + 2] = normal.z;
		st.normal_opacity[4 * idx + 3] = in.opacities[idx];
		st.tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
	}
}

// renderCUDA forward: DSR forward.cu:258-489
template <class R>
static void render_fwd(const SurfelIn<R>& in, SurfelState<R>& st, const R* features, const R* transMats, R* out_color, R* out_others,
                       R* out_refl, R* gaussian_weights) {
	const int W = in.W, H = in.H;
	const size_t HW = (size_t)H * W;
	std::vector<double> gw(in.P, 0.0);
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
	for (int ty = 0; ty < st.gy; ty++)
		for (int tx = 0; tx < st.gx; tx++) {
			uint32_t rs = st.bin.ranges[2 * (ty * st.gx + tx)], re = st.bin.ranges[2 * (ty * st.gx + tx) + 1];
			std::vector<double> tile_w(re - rs, 0.0);
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++) {
					int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
					if (!(px < W && py < H)) continue;
					uint32_t pix_id = W * py + px;
					V2<R> pixf = {(R)px, (R)py};
					R T = 1;
					uint32_t contributor = 0, last_contributor = 0;
					R C[3] = {0, 0, 0}, refl_strength = 0, mask = 0;
					R N[3] = {0, 0, 0}, D = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
					R median_contributor = -1;
					for (uint32_t e = rs; e < re; e++) {
						contributor++;
						int id = st.bin.point_list[e];
						V2<R> xy = {st.means2D[2 * id], st.means2D[2 * id + 1]};
						const R* tm = transMats + 9 * id;
						V3<R> Tu = {tm[0], tm[1], tm[2]}, Tv = {tm[3], tm[4], tm[5]}, Tw = {tm[6], tm[7], tm[8]};
						V3<R> k = pixf.x * Tw - Tu;
						V3<R> l = pixf.y * Tw - Tv;
						V3<R> p = cross(k, l);
						bool unstable = std::fabs(p.z) < R(1e-4f);
						V2<R> s;
						if (!unstable) {
							R inv_pz = R(1) / p.z;
							s = {p.x * inv_pz, p.y * inv_pz};
						} else s = {R(0), R(0)};
						R rho3d = unstable ? R(1e8f) : (s.x * s.x + s.y * s.y);
						V2<R> d = {xy.x - pixf.x, xy.y - pixf.y};
						R rho2d = R(FilterInvSquare) * (d.x * d.x + d.y * d.y);
						R rho = std::min(rho3d, rho2d);
						R depth = (s.x * Tw.x + s.y * Tw.y) + Tw.z;
						if (depth < R(near_n)) continue;
						const R* no = &st.normal_opacity[4 * id];
						R opa = no[3];
						R power = R(-0.5f) * rho;
						if (power > R(0)) continue;
						R alpha = std::min(R(0.99f), opa * std::exp(power));
						if (alpha < R(1.0f / 255.0f)) continue;
						R test_T = T * (1 - alpha);
						if (test_T < R(0.0001f)) break;
						R w = alpha * T;
						R A = 1 - T;
						R m = R(far_n) / (R(far_n) - R(near_n)) * (1 - R(near_n) / depth);
						distortion += (m * m * A + M2 - 2 * m * M1) * w;
						D += depth * w;
						M1 += m * w;
						M2 += m * m * w;
						if (T > R(0.5)) {
							median_depth = depth;
							median_contributor = (R)contributor;
						}
						for (int ch = 0; ch < 3; ch++) N[ch] += no[ch] * w;
						for (int ch = 0; ch < 3; ch++) C[ch] += features[id * 3 + ch] * w;
						refl_strength += in.refl[id] * w;
						if (in.env_scope_mask && in.env_scope_mask[id]) mask = 1;
						T = test_T;
						last_contributor = contributor;
						// forward.cu:458-459 (racy max in the reference; true max here).  The maximum over the tile's pixels is taken
						// privately per list entry and merged once per tile (a maximum does not depend on the order): one critical
						// section per tile instead of one per blended pair
						tile_w[e - rs] = std::max(tile_w[e - rs], (double)w);
					}
					st.final_T[pix_id] = T;
					st.n_contrib[pix_id] = last_contributor;
					for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * in.bg[ch];
					out_refl[pix_id] = refl_strength;
					st.n_contrib[pix_id + HW] = f2u_sat(median_contributor);
					st.final_T[pix_id + HW] = M1;
					st.final_T[pix_id + 2 * HW] = M2;
					out_others[pix_id + 0 * HW] = D;
					out_others[pix_id + 1 * HW] = 1 - T;
					for (int ch = 0; ch < 3; ch++) out_others[pix_id + (2 + ch) * HW] = N[ch];
					out_others[pix_id + 5 * HW] = median_depth;
					out_others[pix_id + 6 * HW] = distortion;
					out_others[pix_id + 7 * HW] = mask;
				}
#pragma omp critical(gw_max)
			for (uint32_t e = rs; e < re; e++) {
				const int id = st.bin.point_list[e];
				if (tile_w[e - rs] > gw[id]) gw[id] = tile_w[e - rs];
			}
		}
	for (int i = 0; i < in.P; i++) gaussian_weights[i] = (R)gw[i];
}

// Rasterizer::forward: DSR rasterizer_impl.cu:198-355
template <class R>
static SurfelState<R>* surfel_forward(const SurfelIn<R>& in, R* out_color, R* out_others, R* out_refl, int* radii_out, R* gaussian_weights,
                                      int* num_rendered) {
	auto* st = new SurfelState<R>();
	const int P = in.P, W = in.W, H = in.H;
	st->P = P; st->W = W; st->H = H;
	st->gx = (W + BLOCK_X - 1) / BLOCK_X;
	st->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	st->depths.assign(P, 0); st->means2D.assign(2 * (size_t)P, 0); st->transMat.assign(9 * (size_t)P, 0);
	st->normal_opacity.assign(4 * (size_t)P, 0); st->rgb.assign(3 * (size_t)P, 0); st->clamped.assign(3 * (size_t)P, 0);
	st->radii.assign(P, 0); st->tiles_touched.assign(P, 0);
	st->final_T.assign((size_t)W * H * 3, 0); st->n_contrib.assign((size_t)W * H * 2, 0);
	preprocess_fwd(in, *st);
	std::vector<float> m2f, df;
	to_float(st->means2D, m2f);
	to_float(st->depths, df);
	build_binning(P, st->gx, st->gy, st->tiles_touched.data(), st->radii.data(), m2f.data(), df.data(), st->bin);
	const R* feat = in.colors_precomp ? in.colors_precomp : st->rgb.data();
	const R* tm = in.transMat_precomp ? in.transMat_precomp : st->transMat.data();
	render_fwd(in, *st, feat, tm, out_color, out_others, out_refl, gaussian_weights);
	if (radii_out) std::memcpy(radii_out, st->radii.data(), sizeof(int) * P);
	*num_rendered = st->bin.num_rendered;
	return st;
}

struct SurfelGrads {
	std::vector<double> transMat, mean2D, normal3D, opacity, colors, refl;
};

// renderCUDA backward: DSR backward.cu:143-470
template <class R>
static void render_bwd(const SurfelIn<R>& in, const SurfelState<R>& st, const R* colors, const R* transMats, const R* dL_dpixels,
                       const R* dL_depths, const R* dL_drefl_map, SurfelGrads& g) {
	const int W = in.W, H = in.H;
	const size_t HW = (size_t)H * W;
	// Accumulation (the reference's atomicAdd targets): every value a pixel contributes is summed in double.  The 256 pixels of a
	// tile walk the same list, so a tile first sums into a PRIVATE row per list entry (TileAccum, oracle_common.hpp) and adds
	// each row to the shared per-Gaussian vectors once, when the tile is done: R x 19 atomic adds instead of one per (pixel,
	// Gaussian, value) — the per-pixel arithmetic below is untouched, only where its results are added changed (round 3: the
	// all-atomic form made 256 host cores slower than one).
	const TileAccum::Target targets[] = {{&g.transMat, 9}, {&g.mean2D, 3}, {&g.normal3D, 3}, {&g.opacity, 1}, {&g.colors, 3}, {&g.refl, 1}};
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
	for (int ty = 0; ty < st.gy; ty++)
		for (int tx = 0; tx < st.gx; tx++) {
			uint32_t rs = st.bin.ranges[2 * (ty * st.gx + tx)], re = st.bin.ranges[2 * (ty * st.gx + tx) + 1];
			TileAccum acc(targets, sizeof(targets) / sizeof(targets[0]), re - rs);
			auto add = [&acc](std::vector<double>& v, size_t i, R val) { acc.add(v, i, (double)val); };
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++) {
					int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
					if (!(px < W && py < H)) continue;
					uint32_t pix_id = W * py + px;
					V2<R> pixf = {(R)px, (R)py};
					const R T_final = st.final_T[pix_id];
					R T = T_final;
					uint32_t contributor = re - rs;
					const int last_contributor = (int)st.n_contrib[pix_id];
					R accum_rec[3] = {0, 0, 0}, dL_dpixel[3], dL_drefl_strength, accum_refl_rec = 0;
					const int median_contributor = (int)st.n_contrib[pix_id + HW];
					R dL_ddepth = dL_depths[0 * HW + pix_id];
					R dL_daccum = dL_depths[1 * HW + pix_id];
					R dL_dreg = dL_depths[6 * HW + pix_id];
					R dL_dnormal2D[3];
					for (int i = 0; i < 3; i++) dL_dnormal2D[i] = dL_depths[(2 + i) * HW + pix_id];
					R dL_dmedian_depth = dL_depths[5 * HW + pix_id];
					R last_depth = 0, last_normal[3] = {0, 0, 0}, accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = {0, 0, 0};
					const R final_D = st.final_T[pix_id + HW];
					const R final_D2 = st.final_T[pix_id + 2 * HW];
					const R final_A = 1 - T_final;
					R last_dL_dT = 0;
					for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
					dL_drefl_strength = dL_drefl_map[pix_id];
					R last_alpha = 0, last_color[3] = {0, 0, 0}, last_refl = 0;
					for (uint32_t e = re; e-- > rs;) {
						contributor--;
						if ((int)contributor >= last_contributor) continue;
						int id = st.bin.point_list[e];
						acc.entry(e - rs);
						V2<R> xy = {st.means2D[2 * id], st.means2D[2 * id + 1]};
						const R* tm = transMats + 9 * id;
						V3<R> Tu = {tm[0], tm[1], tm[2]}, Tv = {tm[3], tm[4], tm[5]}, Tw = {tm[6], tm[7], tm[8]};
						V3<R> k = pixf.x * Tw - Tu;
						V3<R> l = pixf.y * Tw - Tv;
						V3<R> p = cross(k, l);
						bool unstable = std::fabs(p.z) < R(1e-6f);  // differs from forward on purpose
						V2<R> s;
						if (!unstable) {
							R inv_pz = R(1) / p.z;
							s = {p.x * inv_pz, p.y * inv_pz};
						} else s = {R(0), R(0)};
						R rho3d = unstable ? R(1e8f) : (s.x * s.x + s.y * s.y);
						V2<R> d = {xy.x - pixf.x, xy.y - pixf.y};
						R rho2d = R(FilterInvSquare) * (d.x * d.x + d.y * d.y);
						R rho = std::min(rho3d, rho2d);
						R c_d = (s.x * Tw.x + s.y * Tw.y) + Tw.z;
						if (c_d < R(near_n)) continue;
						const R* no = &st.normal_opacity[4 * id];
						R normal[3] = {no[0], no[1], no[2]};
						R opa = no[3];
						R power = R(-0.5f) * rho;
						if (power > R(0)) continue;
						const R G = std::exp(power);
						const R alpha = std::min(R(0.99f), opa * G);
						if (alpha < R(1.0f / 255.0f)) continue;
						T = T / (R(1) - alpha);
						const R dchannel_dcolor = alpha * T;
						R dL_dalpha = 0;
						for (int ch = 0; ch < 3; ch++) {
							const R c = colors[id * 3 + ch];
							accum_rec[ch] = last_alpha * last_color[ch] + (R(1) - last_alpha) * accum_rec[ch];
							last_color[ch] = c;
							const R dL_dchannel = dL_dpixel[ch];
							dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
							add(g.colors, (size_t)id * 3 + ch, dchannel_dcolor * dL_dchannel);
						}
						accum_refl_rec = last_alpha * last_refl + (R(1) - last_alpha) * accum_refl_rec;
						last_refl = in.refl[id];
						dL_dalpha += (in.refl[id] - accum_refl_rec) * dL_drefl_strength;
						add(g.refl, id, dchannel_dcolor * dL_drefl_strength);
						R dL_dz = 0, dL_dweight = 0;
						const R m_d = R(far_n) / (R(far_n) - R(near_n)) * (1 - R(near_n) / c_d);
						const R dmd_dd = (R(far_n) * R(near_n)) / ((R(far_n) - R(near_n)) * c_d * c_d);
						if (contributor == (uint32_t)(median_contributor - 1)) dL_dz += dL_dmedian_depth;
						dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
						dL_dalpha += dL_dweight - last_dL_dT;
						last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
						const R dL_dmd = R(2) * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
						dL_dz += dL_dmd * dmd_dd;
						accum_depth_rec = last_alpha * last_depth + (R(1) - last_alpha) * accum_depth_rec;
						last_depth = c_d;
						dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
						accum_alpha_rec = last_alpha * R(1.0) + (R(1) - last_alpha) * accum_alpha_rec;
						dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
						for (int ch = 0; ch < 3; ch++) {
							accum_normal_rec[ch] = last_alpha * last_normal[ch] + (R(1) - last_alpha) * accum_normal_rec[ch];
							last_normal[ch] = normal[ch];
							dL_dalpha += (normal[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
							add(g.normal3D, (size_t)id * 3 + ch, alpha * T * dL_dnormal2D[ch]);
						}
						dL_dalpha *= T;
						last_alpha = alpha;
						R bg_dot_dpixel = 0;
						for (int i = 0; i < 3; i++) bg_dot_dpixel += in.bg[i] * dL_dpixel[i];
						dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot_dpixel;
						const R dL_dG = opa * dL_dalpha;
						dL_dz += alpha * T * dL_ddepth;
						if (rho3d <= rho2d) {
							const V2<R> dL_ds = {dL_dG * -G * s.x + dL_dz * Tw.x, dL_dG * -G * s.y + dL_dz * Tw.y};
							const V3<R> dz_dTw = {s.x, s.y, R(1)};
							const R dsx_pz = dL_ds.x / p.z, dsy_pz = dL_ds.y / p.z;
							const V3<R> dL_dp = {dsx_pz, dsy_pz, -(dsx_pz * s.x + dsy_pz * s.y)};
							const V3<R> dL_dk = cross(l, dL_dp);
							const V3<R> dL_dl = cross(dL_dp, k);
							const V3<R> dL_dTu = {-dL_dk.x, -dL_dk.y, -dL_dk.z};
							const V3<R> dL_dTv = {-dL_dl.x, -dL_dl.y, -dL_dl.z};
							const V3<R> dL_dTw = {pixf.x * dL_dk.x + pixf.y * dL_dl.x + dL_dz * dz_dTw.x,
							                      pixf.x * dL_dk.y + pixf.y * dL_dl.y + dL_dz * dz_dTw.y,
							                      pixf.x * dL_dk.z + pixf.y * dL_dl.z + dL_dz * dz_dTw.z};
							add(g.transMat, (size_t)id * 9 + 0, dL_dTu.x); add(g.transMat, (size_t)id * 9 + 1, dL_dTu.y); add(g.transMat, (size_t)id * 9 + 2, dL_dTu.z);
							add(g.transMat, (size_t)id * 9 + 3, dL_dTv.x); add(g.transMat, (size_t)id * 9 + 4, dL_dTv.y); add(g.transMat, (size_t)id * 9 + 5, dL_dTv.z);
							add(g.transMat, (size_t)id * 9 + 6, dL_dTw.x); add(g.transMat, (size_t)id * 9 + 7, dL_dTw.y); add(g.transMat, (size_t)id * 9 + 8, dL_dTw.z);
						} else {
							const R dG_ddelx = -G * R(FilterInvSquare) * d.x;
							const R dG_ddely = -G * R(FilterInvSquare) * d.y;
							add(g.mean2D, (size_t)id * 3 + 0, dL_dG * dG_ddelx);
							add(g.mean2D, (size_t)id * 3 + 1, dL_dG * dG_ddely);
							add(g.transMat, (size_t)id * 9 + 6, s.x * dL_dz);
							add(g.transMat, (size_t)id * 9 + 7, s.y * dL_dz);
							add(g.transMat, (size_t)id * 9 + 8, dL_dz);
						}
						add(g.opacity, id, G * dL_dalpha);
					}
				}
			acc.flush(st.bin.point_list.data() + rs);
		}
}

// compute_transmat_aabb + preprocessCUDA backward: DSR backward.cu:473-660
template <class R>
static void preprocess_bwd(int idx, const SurfelIn<R>& in, const SurfelState<R>& st, const R* transMats, R focal_x, R focal_y,
                           R* dL_dtransMats, const R* dL_dnormal3Ds, R* dL_dcolors, R* dL_dshs, R* dL_dmean2Ds, R* dL_dmean3Ds,
                           R* dL_dscales, R* dL_drots) {
	if (!(st.radii[idx] > 0)) return;
	const int W = f2i_sat(focal_x * in.tan_fovx * 2);
	const int H = f2i_sat(focal_y * in.tan_fovy * 2);
	const R* Ts_precomp = in.scales ? nullptr : transMats;
	bool early_return = false;
	{
		M3<R> T;
		V3<R> normal;
		Mat<R, 3, 4> Pm;
		M3<R> Rm;
		V3<R> p_orig{0, 0, 0};
		const R* rot = nullptr;
		R scale[2] = {0, 0};
		if (Ts_precomp != nullptr) {
			for (int c = 0; c < 3; c++)
				for (int r = 0; r < 3; r++) T[c][r] = Ts_precomp[9 * idx + 3 * c + r];
			normal = {R(0), R(0), R(0)};
		} else {
			p_orig = {in.means3D[3 * idx], in.means3D[3 * idx + 1], in.means3D[3 * idx + 2]};
			rot = in.rotations + 4 * idx;
			scale[0] = in.scales[2 * idx]; scale[1] = in.scales[2 * idx + 1];
			Rm = quat_to_rotmat(rot);
			M3<R> S = scale_to_mat(scale, R(1.0f));  // scale_modifier ignored (backward.cu:511)
			M3<R> L = mul(Rm, S);
			Mat<R, 3, 4> Mm;
			for (int r = 0; r < 3; r++) { Mm[0][r] = L[0][r]; Mm[1][r] = L[1][r]; }
			Mm[0][3] = 0; Mm[1][3] = 0;
			Mm[2][0] = p_orig.x; Mm[2][1] = p_orig.y; Mm[2][2] = p_orig.z; Mm[2][3] = 1;
			Mat<R, 4, 4> world2ndc = world2ndc_of(in.proj);
			Mat<R, 3, 4> ndc2pix = ndc2pix_of<R>(W, H);
			Pm = mul(world2ndc, ndc2pix);
			T = mul(transpose(Mm), Pm);
			normal = transformVec4x3(V3<R>{L[2][0], L[2][1], L[2][2]}, in.view);
		}
		M3<R> dL_dT;
		for (int c = 0; c < 3; c++)
			for (int r = 0; r < 3; r++) dL_dT[c][r] = dL_dtransMats[9 * idx + 3 * c + r];
		V3<R> dL_dmean2D = {dL_dmean2Ds[3 * idx], dL_dmean2Ds[3 * idx + 1], dL_dmean2Ds[3 * idx + 2]};
		if (dL_dmean2D.x != 0 || dL_dmean2D.y != 0) {
			V3<R> t_vec = {R(9.0f), R(9.0f), R(-1.0f)};
			R d = dot(t_vec, col(T, 2) * col(T, 2));
			V3<R> f_vec = t_vec * (R(1.0f) / d);
			V3<R> dL_dT0 = (dL_dmean2D.x * f_vec) * col(T, 2);
			V3<R> dL_dT1 = (dL_dmean2D.y * f_vec) * col(T, 2);
			V3<R> dL_dT3 = (dL_dmean2D.x * f_vec) * col(T, 0) + (dL_dmean2D.y * f_vec) * col(T, 1);
			V3<R> dL_df = (dL_dmean2D.x * col(T, 0)) * col(T, 2) + (dL_dmean2D.y * col(T, 1)) * col(T, 2);
			R dL_dd = (R)((double)dot(dL_df, f_vec) * (-1.0 / (double)d));
			V3<R> dd_dT3 = (t_vec * col(T, 2)) * R(2.0f);
			dL_dT3 = dL_dT3 + dL_dd * dd_dT3;
			for (int r = 0; r < 3; r++) {
				dL_dT[0][r] += (&dL_dT0.x)[r];
				dL_dT[1][r] += (&dL_dT1.x)[r];
				dL_dT[2][r] += (&dL_dT3.x)[r];
			}
			if (Ts_precomp != nullptr) {
				for (int c = 0; c < 3; c++)
					for (int r = 0; r < 3; r++) dL_dtransMats[9 * idx + 3 * c + r] = dL_dT[c][r];
				early_return = true;
			}
		}
		if (!early_return && Ts_precomp == nullptr) {
			Mat<R, 3, 4> dL_dM = mul(Pm, transpose(dL_dT));
			V3<R> dL_dn = {dL_dnormal3Ds[3 * idx], dL_dnormal3Ds[3 * idx + 1], dL_dnormal3Ds[3 * idx + 2]};
			V3<R> dL_dtn = transformVec4x3Transpose(dL_dn, in.view);
			V3<R> p_view = transformPoint4x3(p_orig, in.view);
			R cosv = -((p_view.x * normal.x) + (p_view.y * normal.y) + (p_view.z * normal.z));
			R multiplier = cosv > 0 ? R(1) : R(-1);
			dL_dtn = multiplier * dL_dtn;
			M3<R> dL_dRS;
			for (int r = 0; r < 3; r++) { dL_dRS[0][r] = dL_dM[0][r]; dL_dRS[1][r] = dL_dM[1][r]; }
			dL_dRS[2][0] = dL_dtn.x; dL_dRS[2][1] = dL_dtn.y; dL_dRS[2][2] = dL_dtn.z;
			M3<R> dL_dR;
			for (int r = 0; r < 3; r++) {
				dL_dR[0][r] = dL_dRS[0][r] * scale[0];
				dL_dR[1][r] = dL_dRS[1][r] * scale[1];
				dL_dR[2][r] = dL_dRS[2][r];
			}
			quat_to_rotmat_vjp(rot, dL_dR, dL_drots + 4 * idx);
			dL_dscales[2 * idx + 0] = dot(col(dL_dRS, 0), col(Rm, 0));
			dL_dscales[2 * idx + 1] = dot(col(dL_dRS, 1), col(Rm, 1));
			dL_dmean3Ds[3 * idx + 0] = dL_dM[2][0];
			dL_dmean3Ds[3 * idx + 1] = dL_dM[2][1];
			dL_dmean3Ds[3 * idx + 2] = dL_dM[2][2];
		}
	}
	if (in.shs) sh_backward(idx, in.D, in.M, in.means3D, in.campos, in.shs, st.clamped.data(), dL_dcolors, dL_dmean3Ds, dL_dshs);
	// densification hack (backward.cu:656-659), double arithmetic as written there
	R depth = transMats[idx * 9 + 8];
	dL_dmean2Ds[3 * idx + 0] = (R)((double)(dL_dtransMats[idx * 9 + 2] * depth) * 0.5 * (double)float(W));
	dL_dmean2Ds[3 * idx + 1] = (R)((double)(dL_dtransMats[idx * 9 + 5] * depth) * 0.5 * (double)float(H));
}

// Rasterizer::backward + RasterizeGaussiansBackwardCUDA: DSR rasterizer_impl.cu:358-466, rasterize_points.cu:153-267
template <class R>
static void surfel_backward(const SurfelIn<R>& in, const SurfelState<R>& st, const R* dL_dpix, const R* dL_depths, const R* dL_drefl_map,
                            R* dL_dmean2D /*P*3*/, R* dL_dnormal /*P*3*/, R* dL_dopacity /*P*/, R* dL_dcolor /*P*3*/, R* dL_drefl /*P*/,
                            R* dL_dmean3D /*P*3*/, R* dL_dtransMat /*P*9*/, R* dL_dsh /*P*M*3*/, R* dL_dscale /*P*2*/, R* dL_drot /*P*4*/) {
	const int P = in.P;
	SurfelGrads g;
	g.transMat.assign(9 * (size_t)P, 0); g.mean2D.assign(3 * (size_t)P, 0); g.normal3D.assign(3 * (size_t)P, 0);
	g.opacity.assign(P, 0); g.colors.assign(3 * (size_t)P, 0); g.refl.assign(P, 0);
	const R focal_y = R(in.H) / (R(2) * in.tan_fovy);
	const R focal_x = R(in.W) / (R(2) * in.tan_fovx);
	const R* color_ptr = in.colors_precomp ? in.colors_precomp : st.rgb.data();
	const R* tm = in.transMat_precomp ? in.transMat_precomp : st.transMat.data();
	render_bwd(in, st, color_ptr, tm, dL_dpix, dL_depths, dL_drefl_map, g);
	auto put = [](const std::vector<double>& s, R* d) { for (size_t i = 0; i < s.size(); i++) d[i] = (R)s[i]; };
	put(g.transMat, dL_dtransMat); put(g.mean2D, dL_dmean2D); put(g.normal3D, dL_dnormal); put(g.opacity, dL_dopacity);
	put(g.colors, dL_dcolor); put(g.refl, dL_drefl);
	std::fill(dL_dmean3D, dL_dmean3D + 3 * (size_t)P, R(0));
	std::fill(dL_dsh, dL_dsh + (size_t)P * in.M * 3, R(0));
	std::fill(dL_dscale, dL_dscale + 2 * (size_t)P, R(0));
	std::fill(dL_drot, dL_drot + 4 * (size_t)P, R(0));
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
		preprocess_bwd(idx, in, st, tm, focal_x, focal_y, dL_dtransMat, dL_dnormal, dL_dcolor, dL_dsh, dL_dmean2D, dL_dmean3D, dL_dscale, dL_drot);
}

template <class R> static void copy_out(const std::vector<R>& s, void* d) { std::memcpy(d, s.data(), s.size() * sizeof(R)); }

template <class R> static int surfel_get(SurfelState<R>* st, const char* name, void* dst) {
	std::string n(name);
	if (n == "depths") copy_out(st->depths, dst);
	else if (n == "means2D") copy_out(st->means2D, dst);
	else if (n == "transMat") copy_out(st->transMat, dst);
	else if (n == "normal_opacity") copy_out(st->normal_opacity, dst);
	else if (n == "rgb") copy_out(st->rgb, dst);
	else if (n == "clamped") copy_out(st->clamped, dst);
	else if (n == "radii") copy_out(st->radii, dst);
	else if (n == "tiles_touched") copy_out(st->tiles_touched, dst);
	else if (n == "point_offsets") copy_out(st->bin.point_offsets, dst);
	else if (n == "keys_unsorted") copy_out(st->bin.keys_unsorted, dst);
	else if (n == "keys") copy_out(st->bin.keys, dst);
	else if (n == "point_list") copy_out(st->bin.point_list, dst);
	else if (n == "ranges") copy_out(st->bin.ranges, dst);
	else if (n == "final_T") copy_out(st->final_T, dst);
	else if (n == "n_contrib") copy_out(st->n_contrib, dst);
	else return -1;
	return 0;
}

// checkFrustum / markVisible: DSR rasterizer_impl.cu:54-66,141-153
template <class R> static void mark_visible(int P, const R* means3D, const R* view, const R* proj, uint8_t* present) {
	for (int idx = 0; idx < P; idx++) {
		V3<R> p_view;
		bool trap = false;
		present[idx] = in_frustum(idx, means3D, view, proj, false, p_view, trap) ? 1 : 0;
	}
}

}  // namespace orc

using namespace orc;

#define SURFEL_API(SUF, R)                                                                                                         \
	extern "C" void* orc_surfel_forward_##SUF(int P, int D, int M, const R* bg, int W, int H, const R* means3D,                     \
	                                          const uint8_t* env_scope_mask, const R* shs, const R* colors_precomp, const R* refl,  \
	                                          const R* opacities, const R* scales, R scale_modifier, const R* rotations,            \
	                                          const R* transMat_precomp, const R* view, const R* proj, const R* campos, R tan_fovx, \
	                                          R tan_fovy, int prefiltered, R* out_color, R* out_others, R* out_refl, int* radii,    \
	                                          R* gaussian_weights, int* num_rendered) {                                             \
		SurfelIn<R> in{P, D, M, W, H, bg, means3D, env_scope_mask, shs, colors_precomp, refl, opacities, scales, rotations,          \
		               transMat_precomp, view, proj, campos, scale_modifier, tan_fovx, tan_fovy, prefiltered != 0};                 \
		return surfel_forward<R>(in, out_color, out_others, out_refl, radii, gaussian_weights, num_rendered);                       \
	}                                                                                                                              \
	extern "C" void orc_surfel_backward_##SUF(                                                                                     \
	    void* handle, int P, int D, int M, const R* bg, int W, int H, const R* means3D, const R* shs, const R* colors_precomp,      \
	    const R* refl, const R* scales, R scale_modifier, const R* rotations, const R* transMat_precomp, const R* view, const R* proj, \
	    const R* campos, R tan_fovx, R tan_fovy, const R* dL_dpix, const R* dL_depths, const R* dL_drefl_map, R* dL_dmean2D,        \
	    R* dL_dnormal, R* dL_dopacity, R* dL_dcolor, R* dL_drefl, R* dL_dmean3D, R* dL_dtransMat, R* dL_dsh, R* dL_dscale,          \
	    R* dL_drot) {                                                                                                               \
		SurfelIn<R> in{P, D, M, W, H, bg, means3D, nullptr, shs, colors_precomp, refl, nullptr, scales, rotations, transMat_precomp, \
		               view, proj, campos, scale_modifier, tan_fovx, tan_fovy, false};                                              \
		surfel_backward<R>(in, *(SurfelState<R>*)handle, dL_dpix, dL_depths, dL_drefl_map, dL_dmean2D, dL_dnormal, dL_dopacity,     \
		                   dL_dcolor, dL_drefl, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot);                              \
	}                                                                                                                              \
	extern "C" int orc_surfel_get_##SUF(void* handle, const char* name, void* dst) { return surfel_get<R>((SurfelState<R>*)handle, name, dst); } \
	extern "C" int orc_surfel_trapped_##SUF(void* handle) { return ((SurfelState<R>*)handle)->trap ? 1 : 0; }                       \
	extern "C" void orc_surfel_free_##SUF(void* handle) { delete (SurfelState<R>*)handle; }                                         \
	extern "C" void orc_mark_visible_##SUF(int P, const R* means3D, const R* view, const R* proj, uint8_t* present) {               \
		mark_visible<R>(P, means3D, view, proj, present);                                                                           \
	}

SURFEL_API(f32, float)
SURFEL_API(f64, double)
