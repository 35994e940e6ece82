// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference CUDA rasterizers (gssales/gaussian-splatting-reflection).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product path (gaussian-splatting-reflection_amd/) never links, imports or calls it.
//
// PARITY PINNING: the reference ships no tests, golden vectors or fixtures for this path and
// cannot be compiled here (CUDA-only sources + un-vendored glm submodule), so the render /
// backward restatement is "parity unpinned" by reference artefacts.  It is pinned instead by
//   (i)  golden vectors generated from the importable reference utilities (eval_sh, camera
//        matrices; tests/golden/),
//   (ii) hand-derived known-answer cases, and
//   (iii) float64 finite differences of this restatement (every function is templated on the
//        scalar type so the same text runs in double).
//
// Shared helpers follow
//   DGR = submodules/diff-gaussian-rasterization/cuda_rasterizer
//   DSR = submodules/diff-surfel-rasterization/cuda_rasterizer
// glm semantics are restated by hand: matrices are column-major, m[col][row]; products
// accumulate left-to-right over k; vec*vec is component-wise.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>

namespace orc {

constexpr int BLOCK_X = 16;   // DGR/DSR config.h:16-17
constexpr int BLOCK_Y = 16;
constexpr int BLOCK_SIZE = BLOCK_X * BLOCK_Y;

template <class R> struct V2 { R x, y; };
template <class R> struct V3 { R x, y, z; };
template <class R> struct V4 { R x, y, z, w; };

template <class R> inline V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> inline V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> inline V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class R> inline V3<R> operator*(R f, V3<R> a) { return {f * a.x, f * a.y, f * a.z}; }
template <class R> inline V3<R> operator*(V3<R> a, R f) { return {a.x * f, a.y * f, a.z * f}; }
template <class R> inline V3<R> operator/(V3<R> a, R f) { return {a.x / f, a.y / f, a.z / f}; }
template <class R> inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // glm compute_dot: left-to-right
template <class R> inline R length(V3<R> a) { return std::sqrt(dot(a, a)); }
template <class R> inline V3<R> cross(V3<R> a, V3<R> b) {  // DSR auxiliary.h:159
	return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Column-major matrix with C columns and N rows: m[col][row] (glm::mat<C,N>).
template <class R, int C, int N> struct Mat {
	R m[C][N];
	R* operator[](int c) { return m[c]; }
	const R* operator[](int c) const { return m[c]; }
};
// glm: (K cols x N rows) * (M cols x K rows) -> (M cols x N rows), res[j][i] = sum_k a[k][i] * b[j][k]
template <class R, int K, int N, int M>
inline Mat<R, M, N> mul(const Mat<R, K, N>& a, const Mat<R, M, K>& b) {
	Mat<R, M, N> r;
	for (int j = 0; j < M; j++)
		for (int i = 0; i < N; i++) {
			R s = a.m[0][i] * b.m[j][0];
			for (int k = 1; k < K; k++) s = s + a.m[k][i] * b.m[j][k];
			r.m[j][i] = s;
		}
	return r;
}
template <class R, int C, int N> inline Mat<R, N, C> transpose(const Mat<R, C, N>& a) {
	Mat<R, N, C> r;
	for (int c = 0; c < C; c++)
		for (int n = 0; n < N; n++) r.m[n][c] = a.m[c][n];
	return r;
}
template <class R> using M3 = Mat<R, 3, 3>;
// glm::mat3(a..i): fills column by column
template <class R> inline M3<R> mat3(R a, R b, R c, R d, R e, R f, R g, R h, R i) {
	M3<R> r;
	r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
	r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
	r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
	return r;
}
template <class R> inline V3<R> col(const M3<R>& a, int c) { return {a.m[c][0], a.m[c][1], a.m[c][2]}; }

// GPU float->int conversion saturates and maps NaN to 0 (cvt.rzi.s32.f32 / v_cvt_i32_f32);
// a plain C cast is UB out of range, so restate the device behaviour explicitly.
template <class R> inline int f2i_sat(R v) {
	if (v != v) return 0;
	if (v >= R(2147483647.0)) return 2147483647;
	if (v <= R(-2147483648.0)) return (int)0x80000000;
	return (int)v;
}
template <class R> inline uint32_t f2u_sat(R v) {
	if (v != v) return 0u;
	if (v <= R(0)) return 0u;
	if (v >= R(4294967295.0)) return 0xFFFFFFFFu;
	return (uint32_t)v;
}

// SH constants: DGR auxiliary.h:21-38, DSR auxiliary.h:47-64
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                              -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// DGR auxiliary.h:40-43 (double arithmetic inside, as written there)
template <class R> inline R ndc2Pix(R v, int S) { return (R)((((double)v + 1.0) * S - 1.0) * 0.5); }

// DGR auxiliary.h:45-55 / DSR auxiliary.h:71-81; radius arrives as int
template <class R>
inline void getRect(V2<R> p, int max_radius, uint32_t rect_min[2], uint32_t rect_max[2], int gx, int gy) {
	rect_min[0] = (uint32_t)std::min(gx, std::max(0, f2i_sat((p.x - R(max_radius)) / R(BLOCK_X))));
	rect_min[1] = (uint32_t)std::min(gy, std::max(0, f2i_sat((p.y - R(max_radius)) / R(BLOCK_Y))));
	rect_max[0] = (uint32_t)std::min(gx, std::max(0, f2i_sat((p.x + R(max_radius) + R(BLOCK_X - 1)) / R(BLOCK_X))));
	rect_max[1] = (uint32_t)std::min(gy, std::max(0, f2i_sat((p.y + R(max_radius) + R(BLOCK_Y - 1)) / R(BLOCK_Y))));
}

// DGR auxiliary.h:70-109
template <class R> inline V3<R> transformPoint4x3(V3<R> p, const R* m) {
	return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
	        m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
template <class R> inline V4<R> transformPoint4x4(V3<R> p, const R* m) {
	return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
	        m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
template <class R> inline V3<R> transformVec4x3(V3<R> p, const R* m) {
	return {m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
	        m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
template <class R> inline V3<R> transformVec4x3Transpose(V3<R> p, const R* m) {
	return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
	        m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
// DGR auxiliary.h:119-129
template <class R> inline V3<R> dnormvdv(V3<R> v, V3<R> dv) {
	R sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	R invsum32 = R(1) / std::sqrt(sum2 * sum2 * sum2);
	V3<R> r;
	r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return r;
}

// in_frustum: DGR auxiliary.h:151-176 / DSR auxiliary.h:189-214.  Returns false if culled.
// `trap` is set when prefiltered && culled (the reference executes __trap()).
template <class R> inline bool in_frustum(int idx, const R* orig_points, const R* view, const R* proj, bool prefiltered,
                                          V3<R>& p_view, bool& trap) {
	V3<R> p = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
	p_view = transformPoint4x3(p, view);
	if (p_view.z <= R(0.2f)) {
		if (prefiltered) trap = true;
		return false;
	}
	return true;
}

// computeColorFromSH forward: DGR forward.cu:20-71 / DSR forward.cu:20-71 (identical text).
// shs layout (P, M, 3); clamped (P*3) bytes.
template <class R>
inline V3<R> sh_forward(int idx, int deg, int max_coeffs, const R* means, const R* campos, const R* shs, uint8_t* clamped) {
	V3<R> pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	V3<R> cam = {campos[0], campos[1], campos[2]};
	V3<R> dir = pos - cam;
	dir = dir / length(dir);
	auto sh = [&](int k) -> V3<R> {
		const R* s = shs + ((size_t)idx * max_coeffs + k) * 3;
		return V3<R>{s[0], s[1], s[2]};
	};
	V3<R> result = R(SH_C0) * sh(0);
	if (deg > 0) {
		R x = dir.x, y = dir.y, z = dir.z;
		result = result - (R(SH_C1) * y) * sh(1) + (R(SH_C1) * z) * sh(2) - (R(SH_C1) * x) * sh(3);
		if (deg > 1) {
			R xx = x * x, yy = y * y, zz = z * z;
			R xy = x * y, yz = y * z, xz = x * z;
			result = result + (R(SH_C2[0]) * xy) * sh(4) + (R(SH_C2[1]) * yz) * sh(5) +
			         (R(SH_C2[2]) * (R(2) * zz - xx - yy)) * sh(6) + (R(SH_C2[3]) * xz) * sh(7) +
			         (R(SH_C2[4]) * (xx - yy)) * sh(8);
			if (deg > 2) {
				result = result + (R(SH_C3[0]) * y * (R(3) * xx - yy)) * sh(9) + (R(SH_C3[1]) * xy * z) * sh(10) +
				         (R(SH_C3[2]) * y * (R(4) * zz - xx - yy)) * sh(11) +
				         (R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy)) * sh(12) +
				         (R(SH_C3[4]) * x * (R(4) * zz - xx - yy)) * sh(13) + (R(SH_C3[5]) * z * (xx - yy)) * sh(14) +
				         (R(SH_C3[6]) * x * (xx - R(3) * yy)) * sh(15);
			}
		}
	}
	result.x += R(0.5f);
	result.y += R(0.5f);
	result.z += R(0.5f);
	clamped[3 * idx + 0] = (result.x < 0);
	clamped[3 * idx + 1] = (result.y < 0);
	clamped[3 * idx + 2] = (result.z < 0);
	return {std::max(result.x, R(0)), std::max(result.y, R(0)), std::max(result.z, R(0))};
}

// computeColorFromSH backward: DGR backward.cu:23-142 / DSR backward.cu:20-139.
// dL_dshs (P,M,3) written for k < (deg+1)^2; dL_dmeans[idx] += view-direction path.
template <class R>
inline void sh_backward(int idx, int deg, int max_coeffs, const R* means, const R* campos, const R* shs,
                        const uint8_t* clamped, const R* dL_dcolor, R* dL_dmeans, R* dL_dshs) {
	V3<R> pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	V3<R> cam = {campos[0], campos[1], campos[2]};
	V3<R> dir_orig = pos - cam;
	V3<R> dir = dir_orig / length(dir_orig);
	auto sh = [&](int k) -> V3<R> {
		const R* s = shs + ((size_t)idx * max_coeffs + k) * 3;
		return V3<R>{s[0], s[1], s[2]};
	};
	V3<R> dL_dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
	dL_dRGB.x *= clamped[3 * idx + 0] ? R(0) : R(1);
	dL_dRGB.y *= clamped[3 * idx + 1] ? R(0) : R(1);
	dL_dRGB.z *= clamped[3 * idx + 2] ? R(0) : R(1);
	V3<R> dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
	R x = dir.x, y = dir.y, z = dir.z;
	auto put = [&](int k, R f) {
		R* d = dL_dshs + ((size_t)idx * max_coeffs + k) * 3;
		d[0] = f * dL_dRGB.x; d[1] = f * dL_dRGB.y; d[2] = f * dL_dRGB.z;
	};
	put(0, R(SH_C0));
	if (deg > 0) {
		put(1, -R(SH_C1) * y);
		put(2, R(SH_C1) * z);
		put(3, -R(SH_C1) * x);
		dRGBdx = (-R(SH_C1)) * sh(3);
		dRGBdy = (-R(SH_C1)) * sh(1);
		dRGBdz = R(SH_C1) * sh(2);
		if (deg > 1) {
			R xx = x * x, yy = y * y, zz = z * z;
			R xy = x * y, yz = y * z, xz = x * z;
			put(4, R(SH_C2[0]) * xy);
			put(5, R(SH_C2[1]) * yz);
			put(6, R(SH_C2[2]) * (R(2) * zz - xx - yy));
			put(7, R(SH_C2[3]) * xz);
			put(8, R(SH_C2[4]) * (xx - yy));
			dRGBdx = dRGBdx + ((R(SH_C2[0]) * y) * sh(4) + (R(SH_C2[2]) * R(2) * -x) * sh(6) + (R(SH_C2[3]) * z) * sh(7) +
			                   (R(SH_C2[4]) * R(2) * x) * sh(8));
			dRGBdy = dRGBdy + ((R(SH_C2[0]) * x) * sh(4) + (R(SH_C2[1]) * z) * sh(5) + (R(SH_C2[2]) * R(2) * -y) * sh(6) +
			                   (R(SH_C2[4]) * R(2) * -y) * sh(8));
			dRGBdz = dRGBdz + ((R(SH_C2[1]) * y) * sh(5) + (R(SH_C2[2]) * R(2) * R(2) * z) * sh(6) + (R(SH_C2[3]) * x) * sh(7));
			if (deg > 2) {
				put(9, R(SH_C3[0]) * y * (R(3) * xx - yy));
				put(10, R(SH_C3[1]) * xy * z);
				put(11, R(SH_C3[2]) * y * (R(4) * zz - xx - yy));
				put(12, R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy));
				put(13, R(SH_C3[4]) * x * (R(4) * zz - xx - yy));
				put(14, R(SH_C3[5]) * z * (xx - yy));
				put(15, R(SH_C3[6]) * x * (xx - R(3) * yy));
				dRGBdx = dRGBdx + ((R(SH_C3[0]) * sh(9)) * (R(3) * R(2) * xy) + (R(SH_C3[1]) * sh(10)) * yz +
				                   (R(SH_C3[2]) * sh(11)) * (R(-2) * xy) + (R(SH_C3[3]) * sh(12)) * (R(-3) * R(2) * xz) +
				                   (R(SH_C3[4]) * sh(13)) * (R(-3) * xx + R(4) * zz - yy) +
				                   (R(SH_C3[5]) * sh(14)) * (R(2) * xz) + (R(SH_C3[6]) * sh(15)) * (R(3) * (xx - yy)));
				dRGBdy = dRGBdy + ((R(SH_C3[0]) * sh(9)) * (R(3) * (xx - yy)) + (R(SH_C3[1]) * sh(10)) * xz +
				                   (R(SH_C3[2]) * sh(11)) * (R(-3) * yy + R(4) * zz - xx) +
				                   (R(SH_C3[3]) * sh(12)) * (R(-3) * R(2) * yz) + (R(SH_C3[4]) * sh(13)) * (R(-2) * xy) +
				                   (R(SH_C3[5]) * sh(14)) * (R(-2) * yz) + (R(SH_C3[6]) * sh(15)) * (R(-3) * R(2) * xy));
				dRGBdz = dRGBdz + ((R(SH_C3[1]) * sh(10)) * xy + (R(SH_C3[2]) * sh(11)) * (R(4) * R(2) * yz) +
				                   (R(SH_C3[3]) * sh(12)) * (R(3) * (R(2) * zz - xx - yy)) +
				                   (R(SH_C3[4]) * sh(13)) * (R(4) * R(2) * xz) + (R(SH_C3[5]) * sh(14)) * (xx - yy));
			}
		}
	}
	V3<R> dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
	V3<R> dL_dmean = dnormvdv(dir_orig, dL_ddir);
	dL_dmeans[3 * idx + 0] += dL_dmean.x;
	dL_dmeans[3 * idx + 1] += dL_dmean.y;
	dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

// getHigherMsb: DGR/DSR rasterizer_impl.cu:35-50
inline uint32_t getHigherMsb(uint32_t n) {
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step;
		else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

// Binning shared by both variants: DGR/DSR rasterizer_impl.cu:70-138, 282-325.
// tiles_touched -> inclusive scan -> duplicateWithKeys (y outer, x inner) -> stable sort of
// (tile<<32 | depth bits) over bits [0, 32+getHigherMsb(tiles)) -> identifyTileRanges.
struct Binning {
	std::vector<uint32_t> point_offsets;  // P, inclusive scan
	std::vector<uint64_t> keys_unsorted, keys;
	std::vector<uint32_t> vals_unsorted, point_list;
	std::vector<uint32_t> ranges;  // tiles * 2
	int num_rendered = 0;
};

inline void build_binning(int P, int gx, int gy, const uint32_t* tiles_touched, const int* radii, const float* means2D /*P*2*/,
                          const float* depths, Binning& b) {
	b.point_offsets.resize(P);
	uint32_t acc = 0;
	for (int i = 0; i < P; i++) { acc += tiles_touched[i]; b.point_offsets[i] = acc; }
	int R = P > 0 ? (int)b.point_offsets[P - 1] : 0;
	b.num_rendered = R;
	b.keys_unsorted.assign(R, 0); b.vals_unsorted.assign(R, 0);
	for (int idx = 0; idx < P; idx++) {
		if (radii[idx] > 0) {
			uint32_t off = (idx == 0) ? 0 : b.point_offsets[idx - 1];
			uint32_t rmin[2], rmax[2];
			getRect(V2<float>{means2D[2 * idx], means2D[2 * idx + 1]}, radii[idx], rmin, rmax, gx, gy);
			for (int y = rmin[1]; y < (int)rmax[1]; y++)
				for (int x = rmin[0]; x < (int)rmax[0]; x++) {
					uint64_t key = (uint64_t)(y * gx + x);
					key <<= 32;
					uint32_t dbits; std::memcpy(&dbits, &depths[idx], 4);
					key |= dbits;
					b.keys_unsorted[off] = key; b.vals_unsorted[off] = (uint32_t)idx; off++;
				}
		}
	}
	int bit = (int)getHigherMsb((uint32_t)(gx * gy));
	uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
	std::vector<uint32_t> order(R);
	for (int i = 0; i < R; i++) order[i] = i;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) {
		return (b.keys_unsorted[a] & mask) < (b.keys_unsorted[c] & mask);
	});
	b.keys.resize(R); b.point_list.resize(R);
	for (int i = 0; i < R; i++) { b.keys[i] = b.keys_unsorted[order[i]]; b.point_list[i] = b.vals_unsorted[order[i]]; }
	b.ranges.assign((size_t)gx * gy * 2, 0);
	for (int idx = 0; idx < R; idx++) {
		uint32_t currtile = (uint32_t)(b.keys[idx] >> 32);
		if (idx == 0) b.ranges[2 * currtile] = 0;
		else {
			uint32_t prevtile = (uint32_t)(b.keys[idx - 1] >> 32);
			if (currtile != prevtile) { b.ranges[2 * prevtile + 1] = idx; b.ranges[2 * currtile] = idx; }
		}
		if (idx == R - 1) b.ranges[2 * currtile + 1] = R;
	}
}

template <class R> inline void to_float(const std::vector<R>& s, std::vector<float>& d) {
	d.resize(s.size());
	for (size_t i = 0; i < s.size(); i++) d[i] = (float)s[i];
}

// Tile-private accumulation for the render backward kernels (round 3).  The reference adds every (pixel, Gaussian) contribution to
// global per-Gaussian arrays with atomicAdd; the oracle sums the same contributions in double.  All pixels of a tile walk ONE list,
// so a tile sums into a private row per list entry and adds that row to the shared vectors once (`flush`): the number of atomic
// read-modify-writes falls from (pixels x entries x values) to (entries x values) per tile and the all-core run scales, while the
// per-pixel arithmetic — what the restatement is about — is exactly what it was.  Sums are in double in both forms, so results
// move by ~1e-16 relative (summation order) at most.
struct TileAccum {
	struct Target { std::vector<double>* v; int stride; };
	const Target* targets;
	int ntargets, width = 0;
	int base[16];
	std::vector<double> rows;
	double* row = nullptr;
	TileAccum(const Target* t, int n, size_t entries) : targets(t), ntargets(n) {
		for (int k = 0; k < n; k++) { base[k] = width; width += t[k].stride; }
		rows.assign(entries * (size_t)width, 0.0);
	}
	void entry(size_t e) { row = rows.data() + e * (size_t)width; }            // the list entry the following add() calls belong to
	void add(const std::vector<double>& v, size_t i, double val) {             // i = id * stride + component, as the callers index the shared vector
		for (int k = 0; k < ntargets; k++)
			if (targets[k].v == &v) { row[base[k] + (int)(i % (size_t)targets[k].stride)] += val; return; }
	}
	void flush(const uint32_t* ids) {                                          // ids[e] = Gaussian of list entry e
		const size_t n = width ? rows.size() / (size_t)width : 0;
		for (size_t e = 0; e < n; e++) {
			const double* r = rows.data() + e * (size_t)width;
			for (int k = 0; k < ntargets; k++) {
				double* dst = targets[k].v->data() + (size_t)ids[e] * targets[k].stride;
				for (int c = 0; c < targets[k].stride; c++) {
					const double val = r[base[k] + c];
					if (val != 0.0) {
#pragma omp atomic
						dst[c] += val;
					}
				}
			}
		}
	}
};

}  // namespace orc
