// Surface pass of render(): the per-pixel chain the reference runs as ~25 torch ops after the
// rasterizer (gaussian_renderer/__init__.py:151-176 + utils/point_utils.py:9-37):
//   depth_expected = nan_to_num(allmap[0] / clamp(alpha, 1e-3));  depth_median = nan_to_num(allmap[5])
//   surf_depth     = depth_expected * (1 - depth_ratio) + depth_ratio * depth_median
//   points         = surf_depth * rays_d + rays_o                           (depths_to_points)
//   surf_normal    = normalize(cross(P[y+1] - P[y-1], P[x+1] - P[x-1])) on interior pixels, 0 on the border,
//                    times alpha.detach()                                   (depth_to_normal)
// One 16x16 pixel tile per workgroup; the unprojected points of the tile + halo live in LDS, so the forward reads 3 and
// writes 4 floats per pixel and the backward reads 6 and writes 8 (HBM-bound, one pass each).  The backward is written
// as a GATHER (each pixel collects from the four neighbours whose cross product used its point), so there are no atomics.
#include "gsr_internal.hpp"

namespace gsr {

#define SF_T 16

struct RayMat { float m[9]; float o[3]; };   // rays_d = (x, y, 1) . m (row-major 3x3), rays_o = o

__device__ __forceinline__ RayMat load_raymat(const float* __restrict__ raymat) {   // wave-uniform: 12 scalar loads
	RayMat r;
#pragma unroll
	for (int i = 0; i < 9; i++) r.m[i] = raymat[i];
#pragma unroll
	for (int i = 0; i < 3; i++) r.o[i] = raymat[9 + i];
	return r;
}
__device__ __forceinline__ float nan_to_num00(float v) {   // torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest
	if (v != v) return 0.f;
	if (v == __int_as_float(0x7f800000)) return 0.f;
	if (v == __int_as_float(0xff800000)) return -3.4028234663852886e38f;
	return v;
}
__device__ __forceinline__ bool finite_(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }
__device__ __forceinline__ float surf_depth_of(float D, float A, float med, float ratio) {
	const float e = nan_to_num00(D / fmaxf(A, 1e-3f));
	return e * (1.f - ratio) + ratio * nan_to_num00(med);
}
__device__ __forceinline__ void ray_dir(const RayMat& r, int x, int y, float& dx, float& dy, float& dz) {
	const float fx = (float)x, fy = (float)y;
	dx = fx * r.m[0] + fy * r.m[3] + r.m[6];
	dy = fx * r.m[1] + fy * r.m[4] + r.m[7];
	dz = fx * r.m[2] + fy * r.m[5] + r.m[8];
}

__global__ void __launch_bounds__(256)
surface_fwd_kernel(const float* __restrict__ allmap, const float* __restrict__ raymat, float ratio, int H, int W, float* __restrict__ surf_depth,
                   float* __restrict__ surf_normal) {
	__shared__ float P[SF_T + 2][SF_T + 2][3];
	const RayMat ray = load_raymat(raymat);
	const size_t HW = (size_t)H * W;
	const int x0 = blockIdx.x * SF_T, y0 = blockIdx.y * SF_T;
	for (int i = threadIdx.x; i < (SF_T + 2) * (SF_T + 2); i += 256) {
		const int ly = i / (SF_T + 2), lx = i - ly * (SF_T + 2);
		const int gx = x0 + lx - 1, gy = y0 + ly - 1;
		float px = 0.f, py = 0.f, pz = 0.f;
		if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
			const size_t o = (size_t)gy * W + gx;
			const float sd = surf_depth_of(allmap[o], allmap[HW + o], allmap[5 * HW + o], ratio);
			float dx, dy, dz;
			ray_dir(ray, gx, gy, dx, dy, dz);
			px = sd * dx + ray.o[0]; py = sd * dy + ray.o[1]; pz = sd * dz + ray.o[2];
			if (lx >= 1 && lx <= SF_T && ly >= 1 && ly <= SF_T) surf_depth[o] = sd;
		}
		P[ly][lx][0] = px; P[ly][lx][1] = py; P[ly][lx][2] = pz;
	}
	__syncthreads();
	const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
	const int gx = x0 + tx, gy = y0 + ty;
	if (gx >= W || gy >= H) return;
	const size_t o = (size_t)gy * W + gx;
	float nx = 0.f, ny = 0.f, nz = 0.f;
	if (gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1) {
		const int lx = tx + 1, ly = ty + 1;
		const float ax = P[ly + 1][lx][0] - P[ly - 1][lx][0], ay = P[ly + 1][lx][1] - P[ly - 1][lx][1], az = P[ly + 1][lx][2] - P[ly - 1][lx][2];
		const float bx = P[ly][lx + 1][0] - P[ly][lx - 1][0], by = P[ly][lx + 1][1] - P[ly][lx - 1][1], bz = P[ly][lx + 1][2] - P[ly][lx - 1][2];
		const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
		const float inv = 1.0f / fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);   // F.normalize(eps=1e-12)
		const float A = allmap[HW + o];
		nx = cx * inv * A; ny = cy * inv * A; nz = cz * inv * A;
	}
	surf_normal[o] = nx; surf_normal[HW + o] = ny; surf_normal[2 * HW + o] = nz;
}

// Backward.  LDS: points of the tile + halo 2 (from the saved surf_depth), then the cotangents g_dx, g_dy of the two
// central differences for the tile + halo 1, then each pixel gathers
//   g_P(q) = g_dx(q - ey) - g_dx(q + ey) + g_dy(q - ex) - g_dy(q + ex)
// and chains g_P . rays_d (+ the direct surf_depth cotangent) back to allmap planes 0 (depth sum), 1 (alpha) and 5 (median).
__global__ void __launch_bounds__(256)
surface_bwd_kernel(const float* __restrict__ allmap, const float* __restrict__ raymat, float ratio, int H, int W, const float* __restrict__ surf_depth,
                   const float* __restrict__ g_surf_depth, const float* __restrict__ g_surf_normal, float* __restrict__ g_allmap) {
	__shared__ float P[SF_T + 4][SF_T + 4][3];
	__shared__ float G[SF_T + 2][SF_T + 2][6];
	const RayMat ray = load_raymat(raymat);
	const size_t HW = (size_t)H * W;
	const int x0 = blockIdx.x * SF_T, y0 = blockIdx.y * SF_T;
	for (int i = threadIdx.x; i < (SF_T + 4) * (SF_T + 4); i += 256) {
		const int ly = i / (SF_T + 4), lx = i - ly * (SF_T + 4);
		const int gx = x0 + lx - 2, gy = y0 + ly - 2;
		float px = 0.f, py = 0.f, pz = 0.f;
		if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
			const float sd = surf_depth[(size_t)gy * W + gx];
			float dx, dy, dz;
			ray_dir(ray, gx, gy, dx, dy, dz);
			px = sd * dx + ray.o[0]; py = sd * dy + ray.o[1]; pz = sd * dz + ray.o[2];
		}
		P[ly][lx][0] = px; P[ly][lx][1] = py; P[ly][lx][2] = pz;
	}
	__syncthreads();
	for (int i = threadIdx.x; i < (SF_T + 2) * (SF_T + 2); i += 256) {
		const int ly = i / (SF_T + 2), lx = i - ly * (SF_T + 2);
		const int gx = x0 + lx - 1, gy = y0 + ly - 1;
		float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
		if (g_surf_normal && gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1) {
			const int px = lx + 1, py = ly + 1;   // position in P
			const float ax = P[py + 1][px][0] - P[py - 1][px][0], ay = P[py + 1][px][1] - P[py - 1][px][1], az = P[py + 1][px][2] - P[py - 1][px][2];
			const float bx = P[py][px + 1][0] - P[py][px - 1][0], by = P[py][px + 1][1] - P[py][px - 1][1], bz = P[py][px + 1][2] - P[py][px - 1][2];
			const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
			const float len = sqrtf(cx * cx + cy * cy + cz * cz);
			const size_t o = (size_t)gy * W + gx;
			const float A = allmap[HW + o];                  // alpha.detach(): scales the cotangent, receives none
			const float gnx = g_surf_normal[o] * A, gny = g_surf_normal[HW + o] * A, gnz = g_surf_normal[2 * HW + o] * A;
			float gcx, gcy, gcz;
			if (len > 1e-12f) {
				const float inv = 1.0f / len;
				const float nx = cx * inv, ny = cy * inv, nz = cz * inv;
				const float d = nx * gnx + ny * gny + nz * gnz;
				gcx = (gnx - nx * d) * inv; gcy = (gny - ny * d) * inv; gcz = (gnz - nz * d) * inv;
			} else {
				gcx = gnx * 1e12f; gcy = gny * 1e12f; gcz = gnz * 1e12f;
			}
			// c = a x b  ->  g_a = b x g_c,  g_b = g_c x a
			g[0] = by * gcz - bz * gcy; g[1] = bz * gcx - bx * gcz; g[2] = bx * gcy - by * gcx;
			g[3] = gcy * az - gcz * ay; g[4] = gcz * ax - gcx * az; g[5] = gcx * ay - gcy * ax;
		}
#pragma unroll
		for (int k = 0; k < 6; k++) G[ly][lx][k] = g[k];
	}
	__syncthreads();
	const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
	const int gx = x0 + tx, gy = y0 + ty;
	if (gx >= W || gy >= H) return;
	const size_t o = (size_t)gy * W + gx;
	const int lx = tx + 1, ly = ty + 1;
	// a = P[y+1] - P[y-1] at pixel (y, x): P(q) enters a(q - ey) with +, a(q + ey) with -; likewise b along x
	const float gpx = G[ly - 1][lx][0] - G[ly + 1][lx][0] + G[ly][lx - 1][3] - G[ly][lx + 1][3];
	const float gpy = G[ly - 1][lx][1] - G[ly + 1][lx][1] + G[ly][lx - 1][4] - G[ly][lx + 1][4];
	const float gpz = G[ly - 1][lx][2] - G[ly + 1][lx][2] + G[ly][lx - 1][5] - G[ly][lx + 1][5];
	float dx, dy, dz;
	ray_dir(ray, gx, gy, dx, dy, dz);
	float gsd = gpx * dx + gpy * dy + gpz * dz;
	if (g_surf_depth) gsd += g_surf_depth[o];
	const float D = allmap[o], A = allmap[HW + o], med = allmap[5 * HW + o];
	const float cl = fmaxf(A, 1e-3f);
	const float e = D / cl;
	const float ge = finite_(e) ? gsd * (1.f - ratio) : 0.f;          // nan_to_num passes gradient only where finite
	g_allmap[o] = ge / cl;
	g_allmap[HW + o] = (A >= 1e-3f) ? -ge * D / (cl * cl) : 0.f;       // clamp(min): gradient where alpha >= min
	g_allmap[2 * HW + o] = 0.f; g_allmap[3 * HW + o] = 0.f; g_allmap[4 * HW + o] = 0.f;
	g_allmap[5 * HW + o] = finite_(med) ? gsd * ratio : 0.f;
	g_allmap[6 * HW + o] = 0.f; g_allmap[7 * HW + o] = 0.f;
}

}  // namespace gsr

using namespace gsr;

extern "C" int gsr_surface_forward(const float* allmap, const float* raymat, float depth_ratio, int H, int W, float* surf_depth,
                                   float* surf_normal, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (H < 0 || W < 0 || !raymat) { set_error("gsr_surface_forward: invalid argument"); return GSR_E_INVALID; }
	if (H == 0 || W == 0) return 0;
	if (!allmap || !surf_depth || !surf_normal) { set_error("gsr_surface_forward: NULL buffer"); return GSR_E_INVALID; }
	dim3 grid((W + SF_T - 1) / SF_T, (H + SF_T - 1) / SF_T);
	{
		StageTimer st_(GSR_STAGE_SURFACE_FWD, stream);
		surface_fwd_kernel<<<grid, 256, 0, stream>>>(allmap, raymat, depth_ratio, H, W, surf_depth, surf_normal);
	}
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_surface_backward(const float* allmap, const float* raymat, float depth_ratio, int H, int W, const float* surf_depth,
                                    const float* g_surf_depth, const float* g_surf_normal, float* g_allmap, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (H < 0 || W < 0 || !raymat) { set_error("gsr_surface_backward: invalid argument"); return GSR_E_INVALID; }
	if (H == 0 || W == 0) return 0;
	if (!allmap || !surf_depth || !g_allmap) { set_error("gsr_surface_backward: NULL buffer"); return GSR_E_INVALID; }
	dim3 grid((W + SF_T - 1) / SF_T, (H + SF_T - 1) / SF_T);
	{
		StageTimer st_(GSR_STAGE_SURFACE_BWD, stream);
		surface_bwd_kernel<<<grid, 256, 0, stream>>>(allmap, raymat, depth_ratio, H, W, surf_depth, g_surf_depth, g_surf_normal,
		                                             g_allmap);
	}
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}
