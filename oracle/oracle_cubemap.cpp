// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.hpp header).
//
// CPU restatement of submodules/cubemapencoder/src/cubemapencoder.cu (CME), LEFT_TOP_AS_ORIGIN
// branch (cubemapencoder.cu:20).  Parity unpinned by reference artefacts; pinned by
// known-answer cube-face/edge/corner cases and float64 finite differences in tests/.
// Note: the reference builds this file with -use_fast_math (CME/setup.py:10); the oracle uses
// IEEE division, so float comparisons against it carry a tolerance.
#include <cmath>
#include <cstdint>
#include <vector>

namespace orc {

// CME cubemapencoder.cu:66-106
static void EdgeTable(int L, int flag, int* index_xy) {
	int input_face = index_xy[0], input_x = index_xy[1], input_y = index_xy[2];
	if (input_face == 0) {
		if (flag == 1) { index_xy[0] = 4; index_xy[1] = L - 1; index_xy[2] = input_y; }
		else if (flag == 2) { index_xy[0] = 5; index_xy[1] = 0; index_xy[2] = input_y; }
		else if (flag == 4) { index_xy[0] = 3; index_xy[1] = L - 1; index_xy[2] = input_x; }
		else { index_xy[0] = 2; index_xy[1] = L - 1; index_xy[2] = input_x; }
	} else if (input_face == 1) {
		if (flag == 1) { index_xy[0] = 5; index_xy[1] = L - 1; index_xy[2] = input_y; }
		else if (flag == 2) { index_xy[0] = 4; index_xy[1] = 0; index_xy[2] = input_y; }
		else if (flag == 4) { index_xy[0] = 3; index_xy[1] = 0; index_xy[2] = L - 1 - input_x; }
		else { index_xy[0] = 2; index_xy[1] = 0; index_xy[2] = L - 1 - input_x; }
	} else if (input_face == 2) {
		if (flag == 1) { index_xy[0] = 1; index_xy[1] = L - 1 - input_y; index_xy[2] = L - 1; }
		else if (flag == 2) { index_xy[0] = 0; index_xy[1] = input_y; index_xy[2] = L - 1; }
		else if (flag == 4) { index_xy[0] = 4; index_xy[1] = input_x; index_xy[2] = L - 1; }
		else { index_xy[0] = 5; index_xy[1] = L - 1 - input_x; index_xy[2] = L - 1; }
	} else if (input_face == 3) {
		if (flag == 1) { index_xy[0] = 1; index_xy[1] = L - 1 - input_y; index_xy[2] = 0; }
		else if (flag == 2) { index_xy[0] = 0; index_xy[1] = input_y; index_xy[2] = 0; }
		else if (flag == 4) { index_xy[0] = 4; index_xy[1] = input_x; index_xy[2] = 0; }
		else { index_xy[0] = 5; index_xy[1] = L - 1 - input_x; index_xy[2] = 0; }
	} else if (input_face == 4) {
		if (flag == 1) { index_xy[0] = 1; index_xy[1] = L - 1; index_xy[2] = input_y; }
		else if (flag == 2) { index_xy[0] = 0; index_xy[1] = 0; index_xy[2] = input_y; }
		else if (flag == 4) { index_xy[0] = 3; index_xy[1] = input_x; index_xy[2] = 0; }
		else { index_xy[0] = 2; index_xy[1] = input_x; index_xy[2] = 0; }
	} else {
		if (flag == 1) { index_xy[0] = 0; index_xy[1] = L - 1; index_xy[2] = input_y; }
		else if (flag == 2) { index_xy[0] = 1; index_xy[1] = 0; index_xy[2] = input_y; }
		else if (flag == 4) { index_xy[0] = 3; index_xy[1] = L - 1 - input_x; index_xy[2] = L - 1; }
		else { index_xy[0] = 2; index_xy[1] = L - 1 - input_x; index_xy[2] = L - 1; }
	}
}

// CME cubemapencoder.cu:147-187
template <class S> static void Compute_Cubemap_UV(S x, S y, S z, S* uv, int* index) {
	int max_dim = 0;
	S x_ = std::fabs(x), y_ = std::fabs(y), z_ = std::fabs(z);
	S max_v = x_;
	if (y_ > max_v) { max_v = y_; max_dim = 1; }
	if (z_ > max_v) { max_v = z_; max_dim = 2; }
	if (max_dim == 0) {
		uv[0] = z / x; uv[1] = y / x;
		if (x >= S(0)) { *index = 0; uv[0] = -uv[0]; uv[1] = -uv[1]; }
		else { *index = 1; uv[0] = -uv[0]; }
	} else if (max_dim == 1) {
		uv[0] = x / y; uv[1] = z / y;
		if (y >= S(0)) { *index = 2; }
		else { *index = 3; uv[0] = -uv[0]; uv[1] = -uv[1]; }
	} else {
		uv[0] = x / z; uv[1] = y / z;
		if (z >= S(0)) { *index = 4; uv[1] = -uv[1]; }
		else { *index = 5; }
	}
}

// CME cubemapencoder.cu:189-263
template <class S> static bool Compute_Seamless_Index(int index, int L, const S* uv, int* index_xy, S* kxky, int* out_flag) {
	S loc_uv[2] = {uv[0], uv[1]};
	int uy_0, ux_0, uy_1, ux_1;
	S kx, ky;
	int flag = 0;
	bool is_vertex = false;
	loc_uv[1] = -loc_uv[1];
	loc_uv[0] = (loc_uv[0] * S(0.5f) + S(0.5f)) * S(L);
	loc_uv[1] = (loc_uv[1] * S(0.5f) + S(0.5f)) * S(L);
	ux_0 = int(std::floor(loc_uv[0] - S(0.5f))); uy_0 = int(std::floor(loc_uv[1] - S(0.5f)));
	ux_1 = ux_0 + 1; uy_1 = uy_0 + 1;
	kx = loc_uv[0] - S(ux_0) - S(0.5f);
	ky = loc_uv[1] - S(uy_0) - S(0.5f);
	if (ux_0 < 0) ux_0 = 0;
	if (ux_0 >= L) ux_0 = L - 1;
	if (ux_1 < 0) ux_1 = 0;
	if (ux_1 >= L) ux_1 = L - 1;
	if (uy_0 < 0) uy_0 = 0;
	if (uy_0 >= L) uy_0 = L - 1;
	if (uy_1 < 0) uy_1 = 0;
	if (uy_1 >= L) uy_1 = L - 1;
	if (loc_uv[0] < S(0.5f)) { flag |= 0x01; kx = S(0.5f) - loc_uv[0]; }
	else if (loc_uv[0] >= S(L) - S(0.5f)) { flag |= 0x02; }
	if (loc_uv[1] < S(0.5f)) { flag |= 0x04; ky = S(0.5f) - loc_uv[1]; }
	else if (loc_uv[1] >= S(L) - S(0.5f)) { flag |= 0x08; }
	if ((flag & 0x03) && (flag & 0x0C)) {
		is_vertex = true;
		index_xy[0] = index; index_xy[1] = ux_0; index_xy[2] = uy_0;
		index_xy[3] = index; index_xy[4] = ux_0; index_xy[5] = uy_0; EdgeTable(L, flag & 0x03, &index_xy[3]);
		index_xy[6] = index; index_xy[7] = ux_0; index_xy[8] = uy_0; EdgeTable(L, flag & 0x0C, &index_xy[6]);
		index_xy[9] = index; index_xy[10] = ux_0; index_xy[11] = uy_0;  // unused in the reference (uninitialised there)
	} else if (flag & 0x03) {
		index_xy[0] = index; index_xy[1] = ux_0; index_xy[2] = uy_0;
		index_xy[3] = index; index_xy[4] = ux_0; index_xy[5] = uy_0; EdgeTable(L, flag, &index_xy[3]);
		index_xy[6] = index; index_xy[7] = ux_0; index_xy[8] = uy_1;
		index_xy[9] = index; index_xy[10] = ux_0; index_xy[11] = uy_1; EdgeTable(L, flag, &index_xy[9]);
	} else if (flag & 0x0C) {
		index_xy[0] = index; index_xy[1] = ux_0; index_xy[2] = uy_0;
		index_xy[3] = index; index_xy[4] = ux_1; index_xy[5] = uy_0;
		index_xy[6] = index; index_xy[7] = ux_0; index_xy[8] = uy_0; EdgeTable(L, flag, &index_xy[6]);
		index_xy[9] = index; index_xy[10] = ux_1; index_xy[11] = uy_0; EdgeTable(L, flag, &index_xy[9]);
	} else {
		index_xy[0] = index; index_xy[1] = ux_0; index_xy[2] = uy_0;
		index_xy[3] = index; index_xy[4] = ux_1; index_xy[5] = uy_0;
		index_xy[6] = index; index_xy[7] = ux_0; index_xy[8] = uy_1;
		index_xy[9] = index; index_xy[10] = ux_1; index_xy[11] = uy_1;
	}
	kxky[0] = kx; kxky[1] = ky;
	*out_flag = flag;
	return is_vertex;
}

// CME cubemapencoder.cu:265-292
template <class S> static void Compute_Cubemap_UV_Backward(int index, S x, S y, S z, S* uv, S* grad_xyz) {
	int face = index / 2;
	if (face == 0) {
		if (index == 0) { uv[0] = -uv[0]; uv[1] = -uv[1]; }
		else { uv[0] = -uv[0]; }
		grad_xyz[0] = -(z * uv[0] + y * uv[1]) / (x * x);
		grad_xyz[1] = S(1) / x * uv[1];
		grad_xyz[2] = S(1) / x * uv[0];
	} else if (face == 1) {
		if (index == 2) {}
		else { uv[0] = -uv[0]; uv[1] = -uv[1]; }
		grad_xyz[0] = S(1) / y * uv[0];
		grad_xyz[1] = -(x * uv[0] + z * uv[1]) / (y * y);
		grad_xyz[2] = S(1) / y * uv[1];
	} else {
		if (index == 4) { uv[1] = -uv[1]; }
		grad_xyz[0] = S(1) / z * uv[0];
		grad_xyz[1] = S(1) / z * uv[1];
		grad_xyz[2] = -(x * uv[0] + y * uv[1]) / (z * z);
	}
}

// Non-seamless bilinear footprint: CME cubemapencoder.cu:356-378
template <class S> static void plain_bilinear(S vx, S vy, S vz, int L, int* cube_idx, int* ux0, int* ux1, int* uy0, int* uy1, S* kx, S* ky) {
	S uv[2];
	Compute_Cubemap_UV(vx, vy, vz, uv, cube_idx);
	uv[1] = -uv[1];
	uv[0] = (uv[0] * S(0.5f) + S(0.5f)) * S(L);
	uv[1] = (uv[1] * S(0.5f) + S(0.5f)) * S(L);
	int ux_0 = int(std::floor(uv[0] - S(0.5f))), uy_0 = int(std::floor(uv[1] - S(0.5f)));
	int ux_1 = ux_0 + 1, uy_1 = uy_0 + 1;
	*kx = uv[0] - S(ux_0) - S(0.5f);
	*ky = uv[1] - S(uy_0) - S(0.5f);
	auto cl = [L](int v) { return v < 0 ? 0 : (v >= L ? L - 1 : v); };
	*ux0 = cl(ux_0); *ux1 = cl(ux_1); *uy0 = cl(uy_0); *uy1 = cl(uy_1);
}
template <class S> static void nearest_texel(S vx, S vy, S vz, int L, int* cube_idx, int* ux, int* uy) {
	S uv[2];
	Compute_Cubemap_UV(vx, vy, vz, uv, cube_idx);
	uv[1] = -uv[1];
	uv[0] = (uv[0] * S(0.5f) + S(0.5f)) * S(L);
	uv[1] = (uv[1] * S(0.5f) + S(0.5f)) * S(L);
	int x = int(uv[0]), y = int(uv[1]);
	auto cl = [L](int v) { return v < 0 ? 0 : (v >= L ? L - 1 : v); };
	*ux = cl(x); *uy = cl(y);
}

// cubemap_encode_forward: CME cubemapencoder.cu:297-488.  outputs [C,B]
template <class S>
static void cubemap_forward(const S* inputs, const S* cubemap, const S* fail_value, S* outputs, int interp, int seamless, int B, int C, int L) {
	auto tex = [&](int f, int c, int y, int x) -> S { return cubemap[(((size_t)f * C + c) * L + y) * L + x]; };
#pragma omp parallel for schedule(static)
	for (int n = 0; n < B; n++) {
		S vx = inputs[n * 3 + 0], vy = inputs[n * 3 + 1], vz = inputs[n * 3 + 2];
		if (vx == S(0) && vy == S(0) && vz == S(0)) {
			for (int iC = 0; iC < C; iC++) outputs[(size_t)iC * B + n] = fail_value[iC];
			continue;
		}
		if (interp == 0) {
			int f, ux, uy;
			nearest_texel(vx, vy, vz, L, &f, &ux, &uy);
			for (int iC = 0; iC < C; iC++) outputs[(size_t)iC * B + n] = tex(f, iC, uy, ux);
		} else if (seamless == 0) {
			int f, ux0, ux1, uy0, uy1;
			S kx, ky;
			plain_bilinear(vx, vy, vz, L, &f, &ux0, &ux1, &uy0, &uy1, &kx, &ky);
			for (int iC = 0; iC < C; iC++) {
				S v00 = tex(f, iC, uy0, ux0), v01 = tex(f, iC, uy0, ux1), v10 = tex(f, iC, uy1, ux0), v11 = tex(f, iC, uy1, ux1);
				outputs[(size_t)iC * B + n] = (1 - ky) * ((1 - kx) * v00 + kx * v01) + ky * ((1 - kx) * v10 + kx * v11);
			}
		} else {
			S uv[2], kxky[2];
			int cube_idx, index_xy[12], flag;
			Compute_Cubemap_UV(vx, vy, vz, uv, &cube_idx);
			bool is_vertex = Compute_Seamless_Index(cube_idx, L, uv, index_xy, kxky, &flag);
			for (int iC = 0; iC < C; iC++) {
				S v00 = tex(index_xy[0], iC, index_xy[2], index_xy[1]);
				S v01 = tex(index_xy[3], iC, index_xy[5], index_xy[4]);
				S v10 = tex(index_xy[6], iC, index_xy[8], index_xy[7]);
				S v11 = is_vertex ? (v00 + v01 + v10) / S(3) : tex(index_xy[9], iC, index_xy[11], index_xy[10]);
				outputs[(size_t)iC * B + n] = (1 - kxky[1]) * ((1 - kxky[0]) * v00 + kxky[0] * v01) + kxky[1] * ((1 - kxky[0]) * v10 + kxky[0] * v11);
			}
		}
	}
}

// cubemap_encode_backward: CME cubemapencoder.cu:509-779.  grad_cubemap and grad_fail are accumulated
// into (callers pass zeros, cubemap_encoder.py:53-55); texel sums are taken in double (the reference's
// float atomicAdd order is non-deterministic).
template <class S>
static void cubemap_backward(const S* grad_outputs, const S* inputs, const S* cubemap, S* grad_cubemap, S* grad_inputs, S* grad_fail,
                             int interp, int seamless, int B, int C, int L) {
	std::vector<double> gc((size_t)6 * C * L * L, 0.0), gf(C, 0.0);
	auto tidx = [&](int f, int c, int y, int x) -> size_t { return (((size_t)f * C + c) * L + y) * L + x; };
	if (interp == 0)
		for (size_t i = 0; i < (size_t)B * 3; i++) grad_inputs[i] = 0;
	for (int n = 0; n < B; n++) {
		S vx = inputs[n * 3 + 0], vy = inputs[n * 3 + 1], vz = inputs[n * 3 + 2];
		if (vx == S(0) && vy == S(0) && vz == S(0)) {
			for (int iC = 0; iC < C; iC++) gf[iC] += (double)grad_outputs[(size_t)iC * B + n];
			if (interp != 0) { grad_inputs[n * 3 + 0] = 0; grad_inputs[n * 3 + 1] = 0; grad_inputs[n * 3 + 2] = 0; }
			continue;
		}
		if (interp == 0) {
			int f, ux, uy;
			nearest_texel(vx, vy, vz, L, &f, &ux, &uy);
			for (int iC = 0; iC < C; iC++) gc[tidx(f, iC, uy, ux)] += (double)grad_outputs[(size_t)iC * B + n];
			continue;
		}
		S grad_view_[3] = {0, 0, 0};
		if (seamless == 0) {
			int f, ux0, ux1, uy0, uy1;
			S kx, ky;
			plain_bilinear(vx, vy, vz, L, &f, &ux0, &ux1, &uy0, &uy1, &kx, &ky);
			for (int iC = 0; iC < C; iC++) {
				S v00 = cubemap[tidx(f, iC, uy0, ux0)], v01 = cubemap[tidx(f, iC, uy0, ux1)];
				S v10 = cubemap[tidx(f, iC, uy1, ux0)], v11 = cubemap[tidx(f, iC, uy1, ux1)];
				S grad_input = grad_outputs[(size_t)iC * B + n];
				gc[tidx(f, iC, uy0, ux0)] += (double)((1 - ky) * (1 - kx) * grad_input);
				gc[tidx(f, iC, uy0, ux1)] += (double)((1 - ky) * kx * grad_input);
				gc[tidx(f, iC, uy1, ux0)] += (double)(ky * (1 - kx) * grad_input);
				gc[tidx(f, iC, uy1, ux1)] += (double)(ky * kx * grad_input);
				S loc_grad[2];
				loc_grad[0] = (1 - ky) * (v01 - v00) + ky * (v11 - v10);
				loc_grad[1] = (1 - kx) * (v10 - v00) + kx * (v11 - v01);
				loc_grad[0] *= S(0.5f) * S(L) * grad_input;
				loc_grad[1] *= S(0.5f) * S(L) * grad_input;
				loc_grad[1] = -loc_grad[1];
				S lgv[3];
				Compute_Cubemap_UV_Backward(f, vx, vy, vz, loc_grad, lgv);
				grad_view_[0] += lgv[0]; grad_view_[1] += lgv[1]; grad_view_[2] += lgv[2];
			}
		} else {
			S uv[2], kxky[2];
			int cube_idx, index_xy[12], flag;
			Compute_Cubemap_UV(vx, vy, vz, uv, &cube_idx);
			bool is_vertex = Compute_Seamless_Index(cube_idx, L, uv, index_xy, kxky, &flag);
			for (int iC = 0; iC < C; iC++) {
				S grad_input = grad_outputs[(size_t)iC * B + n];
				size_t i00 = tidx(index_xy[0], iC, index_xy[2], index_xy[1]);
				size_t i01 = tidx(index_xy[3], iC, index_xy[5], index_xy[4]);
				size_t i10 = tidx(index_xy[6], iC, index_xy[8], index_xy[7]);
				S v00 = cubemap[i00], v01 = cubemap[i01], v10 = cubemap[i10], v11;
				if (is_vertex) {
					v11 = (v00 + v01 + v10) / S(3);
					S extra_g = kxky[1] * kxky[0] / S(3);
					gc[i00] += (double)(((1 - kxky[1]) * (1 - kxky[0]) + extra_g) * grad_input);
					gc[i01] += (double)(((1 - kxky[1]) * kxky[0] + extra_g) * grad_input);
					gc[i10] += (double)(((kxky[1] * (1 - kxky[0])) + extra_g) * grad_input);
				} else {
					size_t i11 = tidx(index_xy[9], iC, index_xy[11], index_xy[10]);
					v11 = cubemap[i11];
					gc[i00] += (double)((1 - kxky[1]) * (1 - kxky[0]) * grad_input);
					gc[i01] += (double)((1 - kxky[1]) * kxky[0] * grad_input);
					gc[i10] += (double)(kxky[1] * (1 - kxky[0]) * grad_input);
					gc[i11] += (double)(kxky[1] * kxky[0] * grad_input);
				}
				S loc_grad[2];
				loc_grad[0] = (1 - kxky[1]) * (v01 - v00) + kxky[1] * (v11 - v10);
				loc_grad[1] = (1 - kxky[0]) * (v10 - v00) + kxky[0] * (v11 - v01);
				loc_grad[0] *= S(0.5f) * S(L) * grad_input;
				loc_grad[1] *= S(0.5f) * S(L) * grad_input;
				if (flag & 0x01) loc_grad[0] = -loc_grad[0];
				if (flag & 0x04) loc_grad[1] = -loc_grad[1];
				loc_grad[1] = -loc_grad[1];
				S lgv[3];
				Compute_Cubemap_UV_Backward(cube_idx, vx, vy, vz, loc_grad, lgv);
				grad_view_[0] += lgv[0]; grad_view_[1] += lgv[1]; grad_view_[2] += lgv[2];
			}
		}
		grad_inputs[n * 3 + 0] = grad_view_[0];
		grad_inputs[n * 3 + 1] = grad_view_[1];
		grad_inputs[n * 3 + 2] = grad_view_[2];
	}
	for (size_t i = 0; i < gc.size(); i++) grad_cubemap[i] += (S)gc[i];
	for (int i = 0; i < C; i++) grad_fail[i] += (S)gf[i];
}

}  // namespace orc

#define CUBE_API(SUF, S)                                                                                                              \
	extern "C" void orc_cubemap_forward_##SUF(const S* inputs, const S* cubemap, const S* fail_value, S* outputs, int interp,         \
	                                          int seamless, int B, int C, int L) {                                                    \
		orc::cubemap_forward<S>(inputs, cubemap, fail_value, outputs, interp, seamless, B, C, L);                                     \
	}                                                                                                                                 \
	extern "C" void orc_cubemap_backward_##SUF(const S* grad_outputs, const S* inputs, const S* cubemap, S* grad_cubemap,             \
	                                           S* grad_inputs, S* grad_fail, int interp, int seamless, int B, int C, int L) {         \
		orc::cubemap_backward<S>(grad_outputs, inputs, cubemap, grad_cubemap, grad_inputs, grad_fail, interp, seamless, B, C, L);     \
	}
CUBE_API(f32, float)
CUBE_API(f64, double)
