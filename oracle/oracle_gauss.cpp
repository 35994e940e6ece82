// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.hpp header).
//
// Variant G: CPU restatement of submodules/diff-gaussian-rasterization (DGR) —
//   cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h and the
//   tensor plumbing of rasterize_points.cu.  Parity unpinned by reference artefacts (no tests,
//   not compilable here); pinned by float64 finite differences + known-answer tests in tests/.
#include "oracle_common.hpp"

namespace orc {

template <class R> struct GaussIn {
	int P, D, M, W, H;
	const R *bg, *means3D, *shs, *colors_precomp, *normals, *refl, *opacities, *scales, *rotations, *cov3D_precomp;
	const R *view, *proj, *campos;
	R scale_modifier, tan_fovx, tan_fovy;
	bool prefiltered, antialiasing;
};

template <class R> struct GaussState {
	int P = 0, W = 0, H = 0, gx = 0, gy = 0;
	std::vector<R> depths, means2D, cov3D, conic_opacity, rgb;
	std::vector<uint8_t> clamped;
	std::vector<int> radii;
	std::vector<uint32_t> tiles_touched;
	Binning bin;
	std::vector<R> final_T;
	std::vector<uint32_t> n_contrib;
	bool trap = false;
};

// DGR forward.cu:114-148
template <class R> static void computeCov3D(const R* scale, R mod, const R* rot, R* cov3D) {
	M3<R> S = mat3<R>(1, 0, 0, 0, 1, 0, 0, 0, 1);
	S[0][0] = mod * scale[0];
	S[1][1] = mod * scale[1];
	S[2][2] = mod * scale[2];
	R r = rot[0], x = rot[1], y = rot[2], z = rot[3];  // quaternion NOT normalised (forward.cu:123)
	M3<R> Rm = mat3<R>(R(1) - R(2) * (y * y + z * z), R(2) * (x * y - r * z), R(2) * (x * z + r * y),
	                   R(2) * (x * y + r * z), R(1) - R(2) * (x * x + z * z), R(2) * (y * z - r * x),
	                   R(2) * (x * z - r * y), R(2) * (y * z + r * x), R(1) - R(2) * (x * x + y * y));
	M3<R> Mm = mul(S, Rm);
	M3<R> Sigma = mul(transpose(Mm), Mm);
	cov3D[0] = Sigma[0][0]; cov3D[1] = Sigma[0][1]; cov3D[2] = Sigma[0][2];
	cov3D[3] = Sigma[1][1]; cov3D[4] = Sigma[1][2]; cov3D[5] = Sigma[2][2];
}

// DGR forward.cu:74-109
template <class R>
static V3<R> computeCov2D(V3<R> mean, R focal_x, R focal_y, R tan_fovx, R tan_fovy, const R* cov3D, const R* vm) {
	V3<R> t = transformPoint4x3(mean, vm);
	const R limx = R(1.3f) * tan_fovx, limy = R(1.3f) * tan_fovy;
	const R txtz = t.x / t.z, tytz = t.y / t.z;
	t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
	t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
	M3<R> J = mat3<R>(focal_x / t.z, 0, -(focal_x * t.x) / (t.z * t.z), 0, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
	M3<R> Wm = mat3<R>(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
	M3<R> T = mul(Wm, J);
	M3<R> Vrk = mat3<R>(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	M3<R> cov = mul(mul(transpose(T), transpose(Vrk)), T);
	return {cov[0][0], cov[0][1], cov[1][1]};
}

// preprocessCUDA forward: DGR forward.cu:151-269
template <class R> static void preprocess_fwd(const GaussIn<R>& in, GaussState<R>& st, R focal_x, R focal_y) {
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
		V3<R> p_orig = {in.means3D[3 * idx], in.means3D[3 * idx + 1], in.means3D[3 * idx + 2]};
		V4<R> p_hom = transformPoint4x4(p_orig, in.proj);
		R p_w = R(1) / (p_hom.w + R(0.0000001f));
		V3<R> p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};
		const R* cov3D;
		if (in.cov3D_precomp != nullptr) cov3D = in.cov3D_precomp + idx * 6;
		else {
			computeCov3D(in.scales + 3 * idx, in.scale_modifier, in.rotations + 4 * idx, st.cov3D.data() + idx * 6);
			cov3D = st.cov3D.data() + idx * 6;
		}
		V3<R> cov = computeCov2D(p_orig, focal_x, focal_y, in.tan_fovx, in.tan_fovy, cov3D, in.view);
		const R h_var = R(0.3f);
		const R det_cov = cov.x * cov.z - cov.y * cov.y;
		cov.x += h_var;
		cov.z += h_var;
		const R det_cov_plus_h_cov = cov.x * cov.z - cov.y * cov.y;
		R h_convolution_scaling = R(1);
		if (in.antialiasing) h_convolution_scaling = std::sqrt(std::max(R(0.000025f), det_cov / det_cov_plus_h_cov));
		const R det = det_cov_plus_h_cov;
		if (det == R(0)) continue;
		R det_inv = R(1) / det;
		V3<R> conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};
		R mid = R(0.5f) * (cov.x + cov.z);
		R lambda1 = mid + std::sqrt(std::max(R(0.1f), mid * mid - det));
		R lambda2 = mid - std::sqrt(std::max(R(0.1f), mid * mid - det));
		R my_radius = std::ceil(R(3) * std::sqrt(std::max(lambda1, lambda2)));
		V2<R> point_image = {ndc2Pix(p_proj.x, in.W), ndc2Pix(p_proj.y, in.H)};
		uint32_t rmin[2], rmax[2];
		getRect(point_image, f2i_sat(my_radius), rmin, rmax, st.gx, st.gy);
		if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
		if (in.colors_precomp == nullptr) {
			V3<R> c = sh_forward(idx, in.D, in.M, in.means3D, in.campos, in.shs, st.clamped.data());
			st.rgb[idx * 3 + 0] = c.x; st.rgb[idx * 3 + 1] = c.y; st.rgb[idx * 3 + 2] = c.z;
		}
		st.depths[idx] = p_view.z;
		st.radii[idx] = f2i_sat(my_radius);
		st.means2D[2 * idx] = point_image.x;
		st.means2D[2 * idx + 1] = point_image.y;
		R opacity = in.opacities[idx];
		st.conic_opacity[4 * idx + 0] = conic.x;
		st.conic_opacity[4 * idx + 1] = conic.y;
		st.conic_opacity[4 * idx + 2] = conic.z;
		st.conic_opacity[4 * idx + 3] = opacity * h_convolution_scaling;
		st.tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
	}
}

// renderCUDA forward: DGR forward.cu:274-411 (per pixel; the block-level vote only affects speed)
template <class R>
static void render_fwd(const GaussIn<R>& in, GaussState<R>& st, const R* features, R* out_color, R* out_normal,
                       R* out_refl, R* invdepth) {
	const int W = in.W, H = in.H;
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
	for (int ty = 0; ty < st.gy; ty++)
		for (int tx = 0; tx < st.gx; tx++) {
			uint32_t rs = st.bin.ranges[2 * (ty * st.gx + tx)], re = st.bin.ranges[2 * (ty * st.gx + tx) + 1];
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++) {
					int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
					if (!(px < W && py < H)) continue;
					uint32_t pix_id = W * py + px;
					V2<R> pixf = {(R)px, (R)py};
					R T = 1;
					uint32_t contributor = 0, last_contributor = 0;
					R C[3] = {0, 0, 0}, nm[3] = {0, 0, 0}, refl_strength = 0, expected_invdepth = 0;
					for (uint32_t e = rs; e < re; e++) {
						contributor++;
						int id = st.bin.point_list[e];
						R xyx = st.means2D[2 * id], xyy = st.means2D[2 * id + 1];
						R dx = xyx - pixf.x, dy = xyy - pixf.y;
						const R* co = &st.conic_opacity[4 * id];
						R power = R(-0.5f) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
						if (power > R(0)) continue;
						R alpha = std::min(R(0.99f), co[3] * std::exp(power));
						if (alpha < R(1.0f / 255.0f)) continue;
						R test_T = T * (1 - alpha);
						if (test_T < R(0.0001f)) break;  // done = true
						for (int ch = 0; ch < 3; ch++) C[ch] += features[id * 3 + ch] * alpha * T;
						for (int ax = 0; ax < 3; ax++) nm[ax] += in.normals[id * 3 + ax] * alpha * T;
						refl_strength += in.refl[id] * alpha * T;
						if (invdepth) expected_invdepth += (1 / st.depths[id]) * alpha * T;
						T = test_T;
						last_contributor = contributor;
					}
					st.final_T[pix_id] = T;
					st.n_contrib[pix_id] = last_contributor;
					for (int ch = 0; ch < 3; ch++) out_color[ch * H * W + pix_id] = C[ch] + T * in.bg[ch];
					for (int ax = 0; ax < 3; ax++) out_normal[ax * H * W + pix_id] = nm[ax];
					out_refl[pix_id] = refl_strength;
					if (invdepth) invdepth[pix_id] = expected_invdepth;
				}
		}
}

// Rasterizer::forward: DGR rasterizer_impl.cu:198-349
template <class R>
static GaussState<R>* gauss_forward(const GaussIn<R>& in, R* out_color, R* out_normal, R* out_refl, R* invdepth, int* radii_out,
                                    int* num_rendered) {
	auto* st = new GaussState<R>();
	const int P = in.P, W = in.W, H = in.H;
	st->P = P; st->W = W; st->H = H;
	st->gx = (W + BLOCK_X - 1) / BLOCK_X;
	st->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const R focal_y = R(H) / (R(2) * in.tan_fovy);
	const R focal_x = R(W) / (R(2) * in.tan_fovx);
	st->depths.assign(P, 0); st->means2D.assign(2 * (size_t)P, 0); st->cov3D.assign(6 * (size_t)P, 0);
	st->conic_opacity.assign(4 * (size_t)P, 0); st->rgb.assign(3 * (size_t)P, 0); st->clamped.assign(3 * (size_t)P, 0);
	st->radii.assign(P, 0); st->tiles_touched.assign(P, 0);
	st->final_T.assign((size_t)W * H, 0); st->n_contrib.assign((size_t)W * H, 0);
	preprocess_fwd(in, *st, focal_x, focal_y);
	std::vector<float> m2f, df;
	to_float(st->means2D, m2f);
	to_float(st->depths, df);
	build_binning(P, st->gx, st->gy, st->tiles_touched.data(), st->radii.data(), m2f.data(), df.data(), st->bin);
	const R* feat = in.colors_precomp ? in.colors_precomp : st->rgb.data();
	render_fwd(in, *st, feat, out_color, out_normal, out_refl, invdepth);
	if (radii_out) std::memcpy(radii_out, st->radii.data(), sizeof(int) * P);
	*num_rendered = st->bin.num_rendered;
	return st;
}

template <class R> struct GaussGrads {
	// accumulated in double regardless of R (order-independent "ideal" sum of the reference's atomicAdds)
	std::vector<double> mean2D, mean2D_pixels, conic, opacity, colors, normals, refl, invdepth;
};

// renderCUDA backward: DGR backward.cu:452-690
template <class R>
static void render_bwd(const GaussIn<R>& in, const GaussState<R>& st, const R* colors, const R* dL_dpixels, const R* dL_dnormal_map,
                       const R* dL_drefl_map, const R* dL_invdepths, GaussGrads<R>& g) {
	const int W = in.W, H = in.H;
	const R ddelx_dx = R(0.5 * W), ddely_dy = R(0.5 * H);
	// tile-private accumulation rows, flushed once per (tile, list entry): see TileAccum in oracle_common.hpp and oracle_surfel.cpp
	const TileAccum::Target targets[] = {{&g.mean2D, 3}, {&g.mean2D_pixels, 3}, {&g.conic, 4}, {&g.opacity, 1}, {&g.colors, 3}, {&g.normals, 3},
	                                     {&g.refl, 1}, {&g.invdepth, 1}};
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
					R accum_rec[3] = {0, 0, 0}, accum_norm_rec[3] = {0, 0, 0};
					R dL_dpixel[3], dL_dnormal[3], dL_drefl_strength, dL_invdepth = 0;
					R accum_refl_rec = 0, accum_invdepth_rec = 0;
					for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * H * W + pix_id];
					for (int i = 0; i < 3; i++) dL_dnormal[i] = dL_dnormal_map[i * H * W + pix_id];
					dL_drefl_strength = dL_drefl_map[pix_id];
					if (dL_invdepths) dL_invdepth = dL_invdepths[pix_id];
					R last_alpha = 0, last_color[3] = {0, 0, 0}, last_normal[3] = {0, 0, 0}, last_refl = 0, last_invdepth = 0;
					for (uint32_t e = re; e-- > rs;) {
						contributor--;
						if ((int)contributor >= last_contributor) continue;
						int id = st.bin.point_list[e];
						acc.entry(e - rs);
						R dx = st.means2D[2 * id] - pixf.x, dy = st.means2D[2 * id + 1] - pixf.y;
						const R* co = &st.conic_opacity[4 * id];
						const R power = R(-0.5f) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
						if (power > R(0)) continue;
						const R G = std::exp(power);
						const R alpha = std::min(R(0.99f), co[3] * G);
						if (alpha < R(1.0f / 255.0f)) continue;
						T = T / (R(1) - alpha);
						const R dchannel_dcolor = alpha * T;
						R dL_dalpha = 0, dL_dalpha_means2d = 0;
						for (int ch = 0; ch < 3; ch++) {
							const R c = colors[id * 3 + ch];
							accum_rec[ch] = last_alpha * last_color[ch] + (R(1) - last_alpha) * accum_rec[ch];
							last_color[ch] = c;
							const R dL_dchannel = dL_dpixel[ch];
							dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
							dL_dalpha_means2d += dL_dalpha;  // running sum INSIDE the loop (backward.cu:613-614)
							add(g.colors, (size_t)id * 3 + ch, dchannel_dcolor * dL_dchannel);
						}
						for (int ax = 0; ax < 3; ax++) {
							const R n = in.normals[id * 3 + ax];
							accum_norm_rec[ax] = last_alpha * last_normal[ax] + (R(1) - last_alpha) * accum_norm_rec[ax];
							last_normal[ax] = n;
							const R dL_dchannel = dL_dnormal[ax];
							dL_dalpha += (n - accum_norm_rec[ax]) * dL_dchannel;
							add(g.normals, (size_t)id * 3 + ax, dchannel_dcolor * dL_dchannel);
						}
						accum_refl_rec = last_alpha * last_refl + (R(1) - last_alpha) * accum_refl_rec;
						last_refl = in.refl[id];
						dL_dalpha += (in.refl[id] - accum_refl_rec) * dL_drefl_strength;
						add(g.refl, id, dchannel_dcolor * dL_drefl_strength);
						if (dL_invdepths) {
							const R invd = R(1) / st.depths[id];
							accum_invdepth_rec = last_alpha * last_invdepth + (R(1) - last_alpha) * accum_invdepth_rec;
							last_invdepth = invd;
							dL_dalpha += (invd - accum_invdepth_rec) * dL_invdepth;
							add(g.invdepth, id, dchannel_dcolor * dL_invdepth);
						}
						dL_dalpha *= T;
						dL_dalpha_means2d *= T;
						last_alpha = alpha;
						R bg_dot_dpixel = 0;
						for (int i = 0; i < 3; i++) bg_dot_dpixel += in.bg[i] * dL_dpixel[i];
						dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot_dpixel;
						dL_dalpha_means2d += (-T_final / (R(1) - alpha)) * bg_dot_dpixel;
						const R dL_dG = co[3] * dL_dalpha;
						const R dL_dG_means2d = co[3] * dL_dalpha_means2d;
						const R gdx = G * dx, gdy = G * dy;
						const R dG_ddelx = -gdx * co[0] - gdy * co[1];
						const R dG_ddely = -gdy * co[2] - gdx * co[1];
						add(g.mean2D, (size_t)id * 3 + 0, dL_dG * dG_ddelx * ddelx_dx);
						add(g.mean2D, (size_t)id * 3 + 1, dL_dG * dG_ddely * ddely_dy);
						add(g.mean2D_pixels, (size_t)id * 3 + 0, dL_dG_means2d * dG_ddelx * ddelx_dx);
						add(g.mean2D_pixels, (size_t)id * 3 + 1, dL_dG_means2d * dG_ddely * ddely_dy);
						add(g.conic, (size_t)id * 4 + 0, R(-0.5f) * gdx * dx * dL_dG);
						add(g.conic, (size_t)id * 4 + 1, R(-0.5f) * gdx * dy * dL_dG);
						add(g.conic, (size_t)id * 4 + 3, R(-0.5f) * gdy * dy * dL_dG);
						add(g.opacity, id, G * dL_dalpha);
					}
				}
			acc.flush(st.bin.point_list.data() + rs);
		}
}

template <class R> static inline R sq(R x) { return x * x; }

// computeCov2DCUDA: DGR backward.cu:147-326
template <class R>
static void cov2d_bwd(int idx, const GaussIn<R>& in, const GaussState<R>& st, const R* cov3Ds, R h_x, R h_y, const R* dL_dconics,
                      R* dL_dopacity, const R* dL_dinvdepth, R* dL_dmeans, R* dL_dcov) {
	if (!(st.radii[idx] > 0)) return;
	const R* cov3D = cov3Ds + 6 * idx;
	const R* vm = in.view;
	V3<R> mean = {in.means3D[3 * idx], in.means3D[3 * idx + 1], in.means3D[3 * idx + 2]};
	V3<R> dL_dconic = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
	V3<R> t = transformPoint4x3(mean, vm);
	const R limx = R(1.3f) * in.tan_fovx, limy = R(1.3f) * in.tan_fovy;
	const R txtz = t.x / t.z, tytz = t.y / t.z;
	t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
	t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
	const R x_grad_mul = (txtz < -limx || txtz > limx) ? R(0) : R(1);
	const R y_grad_mul = (tytz < -limy || tytz > limy) ? R(0) : R(1);
	M3<R> J = mat3<R>(h_x / t.z, 0, -(h_x * t.x) / (t.z * t.z), 0, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
	M3<R> Wm = mat3<R>(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
	M3<R> Vrk = mat3<R>(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	M3<R> T = mul(Wm, J);
	M3<R> cov2D = mul(mul(transpose(T), transpose(Vrk)), T);
	R c_xx = cov2D[0][0], c_xy = cov2D[0][1], c_yy = cov2D[1][1];
	const R h_var = R(0.3f);
	R d_inside_root = 0;
	if (in.antialiasing) {
		const R det_cov = c_xx * c_yy - c_xy * c_xy;
		c_xx += h_var;
		c_yy += h_var;
		const R det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
		const R h_convolution_scaling = std::sqrt(std::max(R(0.000025f), det_cov / det_cov_plus_h_cov));
		const R dL_dopacity_v = dL_dopacity[idx];
		const R d_h_convolution_scaling = dL_dopacity_v * in.opacities[idx];
		dL_dopacity[idx] = dL_dopacity_v * h_convolution_scaling;
		d_inside_root = (det_cov / det_cov_plus_h_cov) <= R(0.000025f) ? R(0) : d_h_convolution_scaling / (R(2) * h_convolution_scaling);
	} else {
		c_xx += h_var;
		c_yy += h_var;
	}
	R dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
	if (in.antialiasing) {
		const R x = c_xx, y = c_yy, z = c_xy, w = h_var;  // post-"+= h_var" values (backward.cu:235-238)
		const R denom_f = d_inside_root / sq(w * w + w * (x + y) + x * y - z * z);
		dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
		dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
		dL_dc_xy = R(-2) * w * z * (w + x + y) * denom_f;
	}
	R denom = c_xx * c_yy - c_xy * c_xy;
	R denom2inv = R(1) / ((denom * denom) + R(0.0000001f));
	if (denom2inv != 0) {
		dL_dc_xx += denom2inv * (-c_yy * c_yy * dL_dconic.x + 2 * c_xy * c_yy * dL_dconic.y + (denom - c_xx * c_yy) * dL_dconic.z);
		dL_dc_yy += denom2inv * (-c_xx * c_xx * dL_dconic.z + 2 * c_xx * c_xy * dL_dconic.y + (denom - c_xx * c_yy) * dL_dconic.x);
		dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dL_dconic.x - (denom + 2 * c_xy * c_xy) * dL_dconic.y + c_xx * c_xy * dL_dconic.z);
		dL_dcov[6 * idx + 0] = (T[0][0] * T[0][0] * dL_dc_xx + T[0][0] * T[1][0] * dL_dc_xy + T[1][0] * T[1][0] * dL_dc_yy);
		dL_dcov[6 * idx + 3] = (T[0][1] * T[0][1] * dL_dc_xx + T[0][1] * T[1][1] * dL_dc_xy + T[1][1] * T[1][1] * dL_dc_yy);
		dL_dcov[6 * idx + 5] = (T[0][2] * T[0][2] * dL_dc_xx + T[0][2] * T[1][2] * dL_dc_xy + T[1][2] * T[1][2] * dL_dc_yy);
		dL_dcov[6 * idx + 1] = 2 * T[0][0] * T[0][1] * dL_dc_xx + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_dc_xy + 2 * T[1][0] * T[1][1] * dL_dc_yy;
		dL_dcov[6 * idx + 2] = 2 * T[0][0] * T[0][2] * dL_dc_xx + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_dc_xy + 2 * T[1][0] * T[1][2] * dL_dc_yy;
		dL_dcov[6 * idx + 4] = 2 * T[0][2] * T[0][1] * dL_dc_xx + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_dc_xy + 2 * T[1][1] * T[1][2] * dL_dc_yy;
	} else {
		for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
	}
	R dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_dc_xx +
	            (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc_xy;
	R dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_dc_xx +
	            (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc_xy;
	R dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_dc_xx +
	            (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc_xy;
	R dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc_yy +
	            (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_dc_xy;
	R dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc_yy +
	            (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_dc_xy;
	R dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc_yy +
	            (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_dc_xy;
	R dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
	R dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
	R dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
	R dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
	R tz = R(1) / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
	R dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
	R dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
	R dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
	if (dL_dinvdepth) dL_dtz -= dL_dinvdepth[idx] / (t.z * t.z);
	V3<R> dL_dmean = transformVec4x3Transpose(V3<R>{dL_dtx, dL_dty, dL_dtz}, vm);
	dL_dmeans[3 * idx + 0] = dL_dmean.x;  // assignment (backward.cu:325)
	dL_dmeans[3 * idx + 1] = dL_dmean.y;
	dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

// computeCov3D backward: DGR backward.cu:330-393
template <class R> static void cov3d_bwd(int idx, const R* scale, R mod, const R* rot, const R* dL_dcov3Ds, R* dL_dscales, R* dL_drots) {
	R r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	M3<R> Rm = mat3<R>(R(1) - R(2) * (y * y + z * z), R(2) * (x * y - r * z), R(2) * (x * z + r * y),
	                   R(2) * (x * y + r * z), R(1) - R(2) * (x * x + z * z), R(2) * (y * z - r * x),
	                   R(2) * (x * z - r * y), R(2) * (y * z + r * x), R(1) - R(2) * (x * x + y * y));
	M3<R> S = mat3<R>(1, 0, 0, 0, 1, 0, 0, 0, 1);
	V3<R> s = {mod * scale[0], mod * scale[1], mod * scale[2]};
	S[0][0] = s.x; S[1][1] = s.y; S[2][2] = s.z;
	M3<R> Mm = mul(S, Rm);
	const R* d = dL_dcov3Ds + 6 * idx;
	M3<R> dL_dSigma = mat3<R>(d[0], R(0.5f) * d[1], R(0.5f) * d[2], R(0.5f) * d[1], d[3], R(0.5f) * d[4], R(0.5f) * d[2], R(0.5f) * d[4], d[5]);
	M3<R> twoM = Mm;
	for (int c = 0; c < 3; c++)
		for (int n = 0; n < 3; n++) twoM[c][n] = R(2) * Mm[c][n];
	M3<R> dL_dM = mul(twoM, dL_dSigma);
	M3<R> Rt = transpose(Rm);
	M3<R> dL_dMt = transpose(dL_dM);
	dL_dscales[3 * idx + 0] = dot(col(Rt, 0), col(dL_dMt, 0));
	dL_dscales[3 * idx + 1] = dot(col(Rt, 1), col(dL_dMt, 1));
	dL_dscales[3 * idx + 2] = dot(col(Rt, 2), col(dL_dMt, 2));
	for (int n = 0; n < 3; n++) { dL_dMt[0][n] *= s.x; dL_dMt[1][n] *= s.y; dL_dMt[2][n] *= s.z; }
	R q0 = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
	R q1 = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
	R q2 = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
	R q3 = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
	dL_drots[4 * idx + 0] = q0; dL_drots[4 * idx + 1] = q1; dL_drots[4 * idx + 2] = q2; dL_drots[4 * idx + 3] = q3;  // unnormalised (backward.cu:392)
}

// preprocessCUDA backward: DGR backward.cu:399-449
template <class R>
static void preprocess_bwd(int idx, const GaussIn<R>& in, const GaussState<R>& st, const R* dL_dmean2D, R* dL_dmeans, const R* dL_dcolor,
                           const R* dL_dcov3D, R* dL_dsh, R* dL_dscale, R* dL_drot) {
	if (!(st.radii[idx] > 0)) return;
	const R* proj = in.proj;
	V3<R> m = {in.means3D[3 * idx], in.means3D[3 * idx + 1], in.means3D[3 * idx + 2]};
	V4<R> m_hom = transformPoint4x4(m, proj);
	R m_w = R(1) / (m_hom.w + R(0.0000001f));
	R mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
	R mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
	R gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
	V3<R> dL_dmean;
	dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
	dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
	dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
	dL_dmeans[3 * idx + 0] += dL_dmean.x;
	dL_dmeans[3 * idx + 1] += dL_dmean.y;
	dL_dmeans[3 * idx + 2] += dL_dmean.z;
	if (in.shs) sh_backward(idx, in.D, in.M, in.means3D, in.campos, in.shs, st.clamped.data(), dL_dcolor, dL_dmeans, dL_dsh);
	if (in.scales) cov3d_bwd(idx, in.scales + 3 * idx, in.scale_modifier, in.rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
}

// Rasterizer::backward + RasterizeGaussiansBackwardCUDA: DGR rasterizer_impl.cu:353-472, rasterize_points.cu:142-264.
// All outputs are caller-allocated and are zero-initialised here (the binding uses torch::zeros).
template <class R>
static void gauss_backward(const GaussIn<R>& in, const GaussState<R>& st, const R* dL_dpix, const R* dL_dnormal_map,
                           const R* dL_drefl_map, const R* dL_invdepths, R* dL_dmean2D /*P*3*/, R* dL_dmean2D_pixels /*P*3*/,
                           R* dL_dconic /*P*4*/, R* dL_dopacity /*P*/, R* dL_dcolor /*P*3*/, R* dL_dnormals /*P*3*/,
                           R* dL_drefl /*P*/, R* dL_dinvdepth /*P or null*/, R* dL_dmean3D /*P*3*/, R* dL_dcov3D /*P*6*/,
                           R* dL_dsh /*P*M*3*/, R* dL_dscale /*P*3*/, R* dL_drot /*P*4*/) {
	const int P = in.P;
	GaussGrads<R> g;
	g.mean2D.assign(3 * (size_t)P, 0); g.mean2D_pixels.assign(3 * (size_t)P, 0); g.conic.assign(4 * (size_t)P, 0);
	g.opacity.assign(P, 0); g.colors.assign(3 * (size_t)P, 0); g.normals.assign(3 * (size_t)P, 0); g.refl.assign(P, 0);
	g.invdepth.assign(P, 0);
	const R* color_ptr = in.colors_precomp ? in.colors_precomp : st.rgb.data();
	render_bwd(in, st, color_ptr, dL_dpix, dL_dnormal_map, dL_drefl_map, dL_invdepths, g);
	auto put = [](const std::vector<double>& s, R* d) { if (d) for (size_t i = 0; i < s.size(); i++) d[i] = (R)s[i]; };
	put(g.mean2D, dL_dmean2D); put(g.mean2D_pixels, dL_dmean2D_pixels); put(g.conic, dL_dconic); put(g.opacity, dL_dopacity);
	put(g.colors, dL_dcolor); put(g.normals, dL_dnormals); put(g.refl, dL_drefl);
	if (dL_dinvdepth) put(g.invdepth, dL_dinvdepth);
	std::fill(dL_dmean3D, dL_dmean3D + 3 * (size_t)P, R(0));
	std::fill(dL_dcov3D, dL_dcov3D + 6 * (size_t)P, R(0));
	std::fill(dL_dsh, dL_dsh + (size_t)P * in.M * 3, R(0));
	std::fill(dL_dscale, dL_dscale + 3 * (size_t)P, R(0));
	std::fill(dL_drot, dL_drot + 4 * (size_t)P, R(0));
	const R focal_y = R(in.H) / (R(2) * in.tan_fovy);
	const R focal_x = R(in.W) / (R(2) * in.tan_fovx);
	const R* cov3D_ptr = in.cov3D_precomp ? in.cov3D_precomp : st.cov3D.data();
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
		cov2d_bwd(idx, in, st, cov3D_ptr, focal_x, focal_y, dL_dconic, dL_dopacity, dL_invdepths ? dL_dinvdepth : nullptr, dL_dmean3D, dL_dcov3D);
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
		preprocess_bwd(idx, in, st, dL_dmean2D, dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}

template <class R> static void copy_out(const std::vector<R>& s, void* d) { std::memcpy(d, s.data(), s.size() * sizeof(R)); }

template <class R> static int gauss_get(GaussState<R>* st, const char* name, void* dst) {
	std::string n(name);
	if (n == "depths") copy_out(st->depths, dst);
	else if (n == "means2D") copy_out(st->means2D, dst);
	else if (n == "cov3D") copy_out(st->cov3D, dst);
	else if (n == "conic_opacity") copy_out(st->conic_opacity, dst);
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

}  // namespace orc

using namespace orc;

#define GAUSS_API(SUF, R)                                                                                                          \
	extern "C" void* orc_gauss_forward_##SUF(int P, int D, int M, const R* bg, int W, int H, const R* means3D, const R* shs,        \
	                                         const R* colors_precomp, const R* normals, const R* refl, const R* opacities,          \
	                                         const R* scales, R scale_modifier, const R* rotations, const R* cov3D_precomp,         \
	                                         const R* view, const R* proj, const R* campos, R tan_fovx, R tan_fovy, int prefiltered, \
	                                         int antialiasing, R* out_color, R* out_normal, R* out_refl, R* out_invdepth,           \
	                                         int* radii, int* num_rendered) {                                                       \
		GaussIn<R> in{P, D, M, W, H, bg, means3D, shs, colors_precomp, normals, refl, opacities, scales, rotations, cov3D_precomp,   \
		              view, proj, campos, scale_modifier, tan_fovx, tan_fovy, prefiltered != 0, antialiasing != 0};                 \
		return gauss_forward<R>(in, out_color, out_normal, out_refl, out_invdepth, radii, num_rendered);                            \
	}                                                                                                                              \
	extern "C" void orc_gauss_backward_##SUF(                                                                                      \
	    void* handle, int P, int D, int M, const R* bg, int W, int H, const R* means3D, const R* shs, const R* colors_precomp,      \
	    const R* normals, const R* refl, const R* opacities, const R* scales, R scale_modifier, const R* rotations,                 \
	    const R* cov3D_precomp, const R* view, const R* proj, const R* campos, R tan_fovx, R tan_fovy, int antialiasing,            \
	    const R* dL_dpix, const R* dL_dnormal_map, const R* dL_drefl_map, const R* dL_invdepths, R* dL_dmean2D,                     \
	    R* dL_dmean2D_pixels, R* dL_dconic, R* dL_dopacity, R* dL_dcolor, R* dL_dnormals, R* dL_drefl, R* dL_dinvdepth,             \
	    R* dL_dmean3D, R* dL_dcov3D, R* dL_dsh, R* dL_dscale, R* dL_drot) {                                                         \
		GaussIn<R> in{P, D, M, W, H, bg, means3D, shs, colors_precomp, normals, refl, opacities, scales, rotations, cov3D_precomp,   \
		              view, proj, campos, scale_modifier, tan_fovx, tan_fovy, false, antialiasing != 0};                            \
		gauss_backward<R>(in, *(GaussState<R>*)handle, dL_dpix, dL_dnormal_map, dL_drefl_map, dL_invdepths, dL_dmean2D,             \
		                  dL_dmean2D_pixels, dL_dconic, dL_dopacity, dL_dcolor, dL_dnormals, dL_drefl, dL_dinvdepth, dL_dmean3D,    \
		                  dL_dcov3D, dL_dsh, dL_dscale, dL_drot);                                                                   \
	}                                                                                                                              \
	extern "C" int orc_gauss_get_##SUF(void* handle, const char* name, void* dst) { return gauss_get<R>((GaussState<R>*)handle, name, dst); } \
	extern "C" int orc_gauss_trapped_##SUF(void* handle) { return ((GaussState<R>*)handle)->trap ? 1 : 0; }                         \
	extern "C" void orc_gauss_free_##SUF(void* handle) { delete (GaussState<R>*)handle; }

GAUSS_API(f32, float)
GAUSS_API(f64, double)

// Direct access to the shared SH restatement (pinned against the reference's utils/sh_utils.py golden vectors).
#define SH_API(SUF, R)                                                                                                              \
	extern "C" void orc_sh_forward_##SUF(int N, int deg, int M, const R* means, const R* campos, const R* shs, R* rgb, uint8_t* clamped) { \
		for (int i = 0; i < N; i++) {                                                                                               \
			V3<R> c = sh_forward<R>(i, deg, M, means, campos, shs, clamped);                                                        \
			rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;                                                           \
		}                                                                                                                           \
	}                                                                                                                               \
	extern "C" void orc_sh_backward_##SUF(int N, int deg, int M, const R* means, const R* campos, const R* shs, const uint8_t* clamped, \
	                                      const R* dL_dcolor, R* dL_dmeans, R* dL_dshs) {                                           \
		for (int i = 0; i < N; i++) sh_backward<R>(i, deg, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dshs);          \
	}
SH_API(f32, float)
SH_API(f64, double)
