// Training-step kernels around the rasterizer: the photometric loss of the reference's
// train loop (train.py:167-173: (1 - lambda) * L1 + lambda * (1 - SSIM), utils/loss_utils.py:40-97) and the Adam update
// of its eight parameter groups (scene/gaussian_model.py:196-209, torch.optim.Adam(lr=0, eps=1e-15)).
//
// Both are streaming, HBM-bound passes:
//   * SSIM + L1: one 16x16 output tile per workgroup; the 26x26 halo of both images is staged in LDS once, the 11x11
//     Gaussian window (sigma 1.5, zero padding 5: F.conv2d(padding=5, groups=C)) is applied separably to the five moment
//     planes x, y, x^2, y^2, xy from LDS, and the kernel leaves only what the backward needs: the three partial-derivative
//     planes of the SSIM map and two block-reduced sums.  ~8 B read + 12 B written per pixel-channel.
//   * Adam: one float4 per lane over ONE flat parameter / gradient / moment buffer (28 B per parameter), learning rate
//     by segment table, so the 59 floats per Gaussian + cubemap update in a single launch.
#include <cstring>
#include "gsr_internal.hpp"
#include <algorithm>

namespace gsr {

#define SSIM_R 5
#define SSIM_T 32                          // output tile: 32 x 32 pixels per workgroup, four per thread
#define SSIM_HALO (SSIM_T + 2 * SSIM_R)   // 42
#define SSIM_B 4                           // outputs per work item along the filtered direction: 4 + 10 loads instead of 4 x 11

struct SsimWindow { float g[2 * SSIM_R + 1]; };

// utils/loss_utils.py:46-48: exp(-(x - 5)^2 / (2 sigma^2)) as float32, normalised by the float32 sum
static SsimWindow make_window() {
	SsimWindow w;
	float sum = 0.f;
	for (int i = 0; i < 11; i++) {
		w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
		sum += w.g[i];
	}
	for (int i = 0; i < 11; i++) w.g[i] /= sum;
	return w;
}

// Stage the zero-padded halo tiles of up to three planes into LDS.  All of a thread's global loads are issued before the
// first LDS store (the loop is fully unrolled into registers): with three workgroups per CU a load-store-load-store
// sequence leaves the kernel waiting on seven dependent HBM round trips per tile.
#define SSIM_LOADS ((SSIM_HALO * SSIM_HALO + 255) / 256)
template <int NP>
__device__ __forceinline__ void ssim_load_tiles(const float* const (&src)[NP], int H, int W, int x0, int y0, float (*const (&dst)[NP])[SSIM_HALO + 1]) {
	float v[NP][SSIM_LOADS];
#pragma unroll
	for (int j = 0; j < SSIM_LOADS; j++) {
		const int i = threadIdx.x + 256 * j;
		const int ly = i / SSIM_HALO, lx = i - ly * SSIM_HALO;
		const int gx = x0 + lx - SSIM_R, gy = y0 + ly - SSIM_R;
		const bool in = i < SSIM_HALO * SSIM_HALO && gx >= 0 && gx < W && gy >= 0 && gy < H;
		const size_t o = in ? (size_t)gy * W + gx : 0;
#pragma unroll
		for (int p = 0; p < NP; p++) v[p][j] = in ? src[p][o] : 0.f;
	}
#pragma unroll
	for (int j = 0; j < SSIM_LOADS; j++) {
		const int i = threadIdx.x + 256 * j;
		const int ly = i / SSIM_HALO, lx = i - ly * SSIM_HALO;
		if (i < SSIM_HALO * SSIM_HALO) {
#pragma unroll
			for (int p = 0; p < NP; p++) dst[p][ly][lx] = v[p][j];
		}
	}
}

// Forward: per-pixel SSIM (utils/loss_utils.py:75-92) and |x - y|, block-reduced into sums[0] (L1) and sums[1] (SSIM);
// optionally the SSIM map and the three planes d ssim / d mu1, d ssim / d E[x^2], d ssim / d E[xy] for the backward.
// Both separable passes are register-blocked: a work item produces SSIM_B neighbouring outputs from one sliding window of
// SSIM_B + 10 LDS reads (the kernel is LDS-read bound: 29 reads per output pixel instead of 90), every output still
// accumulating its eleven taps in the same order.
__global__ void __launch_bounds__(256)
ssim_l1_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W, SsimWindow win, float C1, float C2,
                   float* __restrict__ partials, float* __restrict__ ssim_map, float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                   float* __restrict__ dm_dsigma12) {
	__shared__ float ta[SSIM_HALO][SSIM_HALO + 1], tb[SSIM_HALO][SSIM_HALO + 1];
	__shared__ float hs[5][SSIM_HALO][SSIM_T + 1];
	__shared__ float red[2][4];
	const size_t plane = (size_t)blockIdx.z * H * W;
	const int x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
	{
		const float* const src[2] = {img1 + plane, img2 + plane};
		float (*const dst[2])[SSIM_HALO + 1] = {ta, tb};
		ssim_load_tiles<2>(src, H, W, x0, y0, dst);
	}
	__syncthreads();
	// horizontal pass: 42 rows x (32 / 4) strips x 5 moments
	for (int i = threadIdx.x; i < SSIM_HALO * (SSIM_T / SSIM_B); i += 256) {
		const int ly = i % SSIM_HALO, lx = (i / SSIM_HALO) * SSIM_B;   // lanes run down the rows: row pitch 43 / 33 words is odd, no bank conflicts
		float a[SSIM_B + 10], b[SSIM_B + 10], aa[SSIM_B + 10], bb[SSIM_B + 10], ab[SSIM_B + 10];
#pragma unroll
		for (int k = 0; k < SSIM_B + 10; k++) {
			a[k] = ta[ly][lx + k]; b[k] = tb[ly][lx + k];
			aa[k] = a[k] * a[k]; bb[k] = b[k] * b[k]; ab[k] = a[k] * b[k];   // once per window element, not once per tap
		}
#pragma unroll
		for (int c = 0; c < SSIM_B; c++) {
			float s1 = 0, s2 = 0, s11 = 0, s22 = 0, s12 = 0;
#pragma unroll
			for (int k = 0; k < 11; k++) {
				const float w = win.g[k];
				s1 += w * a[c + k]; s2 += w * b[c + k]; s11 += w * aa[c + k]; s22 += w * bb[c + k]; s12 += w * ab[c + k];
			}
			hs[0][ly][lx + c] = s1; hs[1][ly][lx + c] = s2; hs[2][ly][lx + c] = s11; hs[3][ly][lx + c] = s22; hs[4][ly][lx + c] = s12;
		}
	}
	__syncthreads();
	// vertical pass: thread (tx, tq) owns the four rows 4 tq .. 4 tq + 3 of column tx
	const int tx = threadIdx.x & 31, tq = threadIdx.x >> 5;
	const int gx = x0 + tx;
	float col[5][SSIM_B + 10];
#pragma unroll
	for (int m = 0; m < 5; m++)
#pragma unroll
		for (int k = 0; k < SSIM_B + 10; k++) col[m][k] = hs[m][tq * SSIM_B + k][tx];
	float l1 = 0.f, ssim_sum = 0.f;
#pragma unroll
	for (int r = 0; r < SSIM_B; r++) {
		const int ty = tq * SSIM_B + r, gy = y0 + ty;
		if (gx < W && gy < H) {
			float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
			for (int k = 0; k < 11; k++) {
				const float w = win.g[k];
				mu1 += w * col[0][r + k]; mu2 += w * col[1][r + k];
				e11 += w * col[2][r + k]; e22 += w * col[3][r + k]; e12 += w * col[4][r + k];
			}
			const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
			const float sigma1_sq = e11 - mu1_sq, sigma2_sq = e22 - mu2_sq, sigma12 = e12 - mu12;
			const float A = 2.f * mu12 + C1, B = 2.f * sigma12 + C2, Cc = mu1_sq + mu2_sq + C1, D = sigma1_sq + sigma2_sq + C2;
			const float inv = 1.0f / (Cc * D);
			const float sv = A * B * inv;
			const size_t o = plane + (size_t)gy * W + gx;
			l1 += fabsf(ta[ty + SSIM_R][tx + SSIM_R] - tb[ty + SSIM_R][tx + SSIM_R]);
			ssim_sum += sv;
			if (ssim_map) ssim_map[o] = sv;
			if (dm_dmu1) {
				// with mu1, E[x^2], E[xy] as the independent conv outputs of image 1 (sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2)
				dm_dmu1[o] = 2.f * mu2 * (B - A) * inv - 2.f * mu1 * sv * (D - Cc) * inv;
				dm_dsigma1_sq[o] = -sv / D;
				dm_dsigma12[o] = 2.f * A * inv;
			}
		}
	}
	// block sums -> one partial pair per block (thousands of blocks adding into ONE address serialise at the memory side:
	// measured 0.32 ms for the 1080p loss with atomics); ssim_l1_reduce_kernel adds them up in a fixed order
	float r4[4] = {l1, ssim_sum, 0.f, 0.f};
	wave_sum4(r4);
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 63) { red[0][wave] = r4[0]; red[1][wave] = r4[1]; }
	__syncthreads();
	if (threadIdx.x < 2) {
		const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
		partials[2 * blk + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
	}
}

// second stage: one workgroup, double accumulation, fixed order -> the loss value is bitwise reproducible
__global__ void __launch_bounds__(256) ssim_l1_reduce_kernel(const float* __restrict__ partials, size_t nblocks, float* __restrict__ sums) {
	__shared__ double red[2][256];
	double a = 0.0, b = 0.0;
	for (size_t i = threadIdx.x; i < nblocks; i += 256) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
	red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { sums[0] = (float)red[0][0]; sums[1] = (float)red[1][0]; }
}

// Backward: dL/dimg1 = w_l1 * sign(x - y) + w_ssim * [ conv(dm_dmu1) + 2 x conv(dm_dsigma1_sq) + y conv(dm_dsigma12) ]
// (the window is symmetric, so the adjoint of the zero-padded correlation is the same correlation of the zero-padded planes).
__global__ void __launch_bounds__(256)
ssim_l1_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W, SsimWindow win,
                   const float* __restrict__ weights /* [2]: dL/d(sum |x-y|), dL/d(sum ssim) */, const float* __restrict__ dm_dmu1,
                   const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
	__shared__ float t0[SSIM_HALO][SSIM_HALO + 1], t1[SSIM_HALO][SSIM_HALO + 1], t2[SSIM_HALO][SSIM_HALO + 1];
	__shared__ float hs[3][SSIM_HALO][SSIM_T + 1];
	const size_t plane = (size_t)blockIdx.z * H * W;
	const int x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
	{
		const float* const src[3] = {dm_dmu1 + plane, dm_dsigma1_sq + plane, dm_dsigma12 + plane};
		float (*const dst[3])[SSIM_HALO + 1] = {t0, t1, t2};
		ssim_load_tiles<3>(src, H, W, x0, y0, dst);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < SSIM_HALO * (SSIM_T / SSIM_B); i += 256) {
		const int ly = i % SSIM_HALO, lx = (i / SSIM_HALO) * SSIM_B;   // lanes run down the rows: row pitch 43 / 33 words is odd, no bank conflicts
		float v0[SSIM_B + 10], v1[SSIM_B + 10], v2[SSIM_B + 10];
#pragma unroll
		for (int k = 0; k < SSIM_B + 10; k++) { v0[k] = t0[ly][lx + k]; v1[k] = t1[ly][lx + k]; v2[k] = t2[ly][lx + k]; }
#pragma unroll
		for (int c = 0; c < SSIM_B; c++) {
			float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
			for (int k = 0; k < 11; k++) {
				const float w = win.g[k];
				s0 += w * v0[c + k]; s1 += w * v1[c + k]; s2 += w * v2[c + k];
			}
			hs[0][ly][lx + c] = s0; hs[1][ly][lx + c] = s1; hs[2][ly][lx + c] = s2;
		}
	}
	__syncthreads();
	const int tx = threadIdx.x & 31, tq = threadIdx.x >> 5;
	const int gx = x0 + tx;
	if (gx >= W) return;
	float col[3][SSIM_B + 10];
#pragma unroll
	for (int m = 0; m < 3; m++)
#pragma unroll
		for (int k = 0; k < SSIM_B + 10; k++) col[m][k] = hs[m][tq * SSIM_B + k][tx];
	const float w_l1 = weights[0], w_ssim = weights[1];
#pragma unroll
	for (int r = 0; r < SSIM_B; r++) {
		const int gy = y0 + tq * SSIM_B + r;
		if (gy >= H) break;
		float c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
		for (int k = 0; k < 11; k++) {
			const float w = win.g[k];
			c0 += w * col[0][r + k]; c1 += w * col[1][r + k]; c2 += w * col[2][r + k];
		}
		const size_t o = plane + (size_t)gy * W + gx;
		const float x = img1[o], y = img2[o];
		const float d = x - y;
		const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // torch.abs backward: sign(), 0 at 0
		dL_dimg1[o] = w_l1 * sgn + w_ssim * (c0 + 2.f * x * c1 + y * c2);
	}
}

// ---------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, amsgrad = False, weight_decay = 0, maximize = False; torch/optim/adam.py _single_tensor_adam):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#define ADAM_MAX_SEG 16
struct AdamSegs {
	int n;
	unsigned long long end[ADAM_MAX_SEG];   // exclusive end offset of segment i (begin = end[i-1])
	float lr[ADAM_MAX_SEG], lr2[ADAM_MAX_SEG];
	unsigned int period[ADAM_MAX_SEG], split[ADAM_MAX_SEG];
};
__device__ __forceinline__ float adam_lr(const AdamSegs& s, unsigned long long i) {
	int k = 0;
	while (k + 1 < s.n && i >= s.end[k]) k++;
	const unsigned long long begin = k ? s.end[k - 1] : 0ull;
	if (s.period[k] == 0u) return s.lr[k];
	return ((i - begin) % s.period[k]) < s.split[k] ? s.lr[k] : s.lr2[k];
}
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float bc1,
                                         float sqrt_bc2) {
	m = b1 * m + (1.f - b1) * g;
	v = b2 * v + (1.f - b2) * (g * g);
	const float denom = sqrtf(v) / sqrt_bc2 + eps;
	p -= (lr / bc1) * (m / denom);
}
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
            unsigned long long first, unsigned long long n, AdamSegs segs, float b1, float b2, float eps, float inv_bc1, float inv_sqrt_bc2) {
	// (inv_bc1, inv_sqrt_bc2 carry bc1 and sqrt(bc2) themselves: the divisions stay in the kernel, as in torch's addcdiv path)
	// elements [first, n) of the buffers (first is a multiple of 4): a rank that owns one shard of the flat buffer steps only that
	const unsigned long long i4 = first + ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4ull;
	if (i4 >= n) return;
	if (i4 + 4 <= n) {
		float4 p = *reinterpret_cast<float4*>(param + i4);
		const float4 g = *reinterpret_cast<const float4*>(grad + i4);
		float4 m = *reinterpret_cast<float4*>(exp_avg + i4), v = *reinterpret_cast<float4*>(exp_avg_sq + i4);
		adam_one(p.x, g.x, m.x, v.x, adam_lr(segs, i4), b1, b2, eps, inv_bc1, inv_sqrt_bc2);
		adam_one(p.y, g.y, m.y, v.y, adam_lr(segs, i4 + 1), b1, b2, eps, inv_bc1, inv_sqrt_bc2);
		adam_one(p.z, g.z, m.z, v.z, adam_lr(segs, i4 + 2), b1, b2, eps, inv_bc1, inv_sqrt_bc2);
		adam_one(p.w, g.w, m.w, v.w, adam_lr(segs, i4 + 3), b1, b2, eps, inv_bc1, inv_sqrt_bc2);
		*reinterpret_cast<float4*>(param + i4) = p;
		*reinterpret_cast<float4*>(exp_avg + i4) = m;
		*reinterpret_cast<float4*>(exp_avg_sq + i4) = v;
	} else {
		for (unsigned long long i = i4; i < n; i++) {
			float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
			adam_one(p, grad[i], m, v, adam_lr(segs, i), b1, b2, eps, inv_bc1, inv_sqrt_bc2);
			param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
		}
	}
}

}  // namespace gsr

using namespace gsr;

// ---- normal-consistency term of the reference's training loss (train.py:182-189):
//   normal_error = (1 - sum_c rend_normal[c] * surf_normal[c]) [* env_scope_mask];  loss = lambda * mean(normal_error)
// five elementwise torch kernels each way at 1080p; here one pass each way.  The forward leaves sum(normal_error) (block
// partials added in a fixed order, in double: bitwise reproducible); the caller scales by lambda / HW.  The backward takes
// the upstream gradient of that SUM as a one-float device tensor (no host synchronisation) and writes both gradients fully.
__global__ void __launch_bounds__(256)
normal_loss_fwd_kernel(const float* __restrict__ rend, const float* __restrict__ surf, const float* __restrict__ mask, size_t HW, float* __restrict__ partials) {
	__shared__ float red[4];
	float acc = 0.f;
	for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (size_t)gridDim.x * 256) {
		float e = 1.0f - (rend[p] * surf[p] + rend[HW + p] * surf[HW + p] + rend[2 * HW + p] * surf[2 * HW + p]);
		if (mask) e *= mask[p];
		acc += e;
	}
	float r4[4] = {acc, 0.f, 0.f, 0.f};
	wave_sum4(r4);
	if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = r4[0];
	__syncthreads();
	if (threadIdx.x == 0) {
		partials[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
		partials[2 * blockIdx.x + 1] = 0.f;
	}
}
__global__ void __launch_bounds__(256)
normal_loss_bwd_kernel(const float* __restrict__ rend, const float* __restrict__ surf, const float* __restrict__ mask, size_t HW,
                       const float* __restrict__ g_sum, float* __restrict__ g_rend, float* __restrict__ g_surf) {
	const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (p >= HW) return;
	const float g = -g_sum[0] * (mask ? mask[p] : 1.0f);
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const float r = rend[c * HW + p], s = surf[c * HW + p];
		g_rend[c * HW + p] = g * s;
		g_surf[c * HW + p] = g * r;
	}
}
#define NORMAL_LOSS_BLOCKS 1024
extern "C" size_t gsr_normal_loss_scratch_floats(void) { return 2 * NORMAL_LOSS_BLOCKS; }
extern "C" int gsr_normal_loss_forward(const float* rend_normal, const float* surf_normal, const float* mask, int H, int W, float* sum2,
                                       float* scratch, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (H <= 0 || W <= 0 || !rend_normal || !surf_normal || !sum2 || !scratch) { set_error("gsr_normal_loss_forward: invalid argument"); return GSR_E_INVALID; }
	const size_t HW = (size_t)H * W;
	const unsigned blocks = (unsigned)std::min<size_t>(NORMAL_LOSS_BLOCKS, (HW + 255) / 256);
	normal_loss_fwd_kernel<<<blocks, 256, 0, stream>>>(rend_normal, surf_normal, mask, HW, scratch);
	ssim_l1_reduce_kernel<<<1, 256, 0, stream>>>(scratch, (size_t)blocks, sum2);      // sum2[0] = sum(normal_error), sum2[1] = 0
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}
extern "C" int gsr_normal_loss_backward(const float* rend_normal, const float* surf_normal, const float* mask, int H, int W, const float* g_sum,
                                        float* g_rend_normal, float* g_surf_normal, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (H <= 0 || W <= 0 || !rend_normal || !surf_normal || !g_sum || !g_rend_normal || !g_surf_normal) {
		set_error("gsr_normal_loss_backward: invalid argument");
		return GSR_E_INVALID;
	}
	const size_t HW = (size_t)H * W;
	normal_loss_bwd_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, stream>>>(rend_normal, surf_normal, mask, HW, g_sum, g_rend_normal, g_surf_normal);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" size_t gsr_ssim_l1_scratch_floats(int C, int H, int W) {
	if (C <= 0 || H <= 0 || W <= 0) return 0;
	return 2 * (size_t)C * ((H + SSIM_T - 1) / SSIM_T) * ((W + SSIM_T - 1) / SSIM_T);
}

extern "C" int gsr_ssim_l1_forward(const float* img1, const float* img2, int C, int H, int W, float C1, float C2, float* sums, float* scratch,
                                   float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (C < 0 || H < 0 || W < 0 || !sums) { set_error("gsr_ssim_l1_forward: invalid argument"); return GSR_E_INVALID; }
	if (C == 0 || H == 0 || W == 0) { GSR_HIP_CHECK(hipMemsetAsync(sums, 0, 2 * sizeof(float), stream)); return 0; }
	if (!scratch) { set_error("gsr_ssim_l1_forward: scratch of gsr_ssim_l1_scratch_floats(C,H,W) floats is required"); return GSR_E_INVALID; }
	if (!img1 || !img2 || ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr)) || ((dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))) {
		set_error("gsr_ssim_l1_forward: NULL image or partial set of derivative planes");
		return GSR_E_INVALID;
	}
	static const SsimWindow win = make_window();
	dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C);
	{
		StageTimer st_(GSR_STAGE_LOSS_FWD, stream);
		ssim_l1_fwd_kernel<<<grid, 256, 0, stream>>>(img1, img2, H, W, win, C1, C2, scratch, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
		ssim_l1_reduce_kernel<<<1, 256, 0, stream>>>(scratch, (size_t)grid.x * grid.y * grid.z, sums);
	}
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_ssim_l1_backward(const float* img1, const float* img2, int C, int H, int W, const float* weights, const float* dm_dmu1,
                                    const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (C < 0 || H < 0 || W < 0) { set_error("gsr_ssim_l1_backward: invalid argument"); return GSR_E_INVALID; }
	if (C == 0 || H == 0 || W == 0) return 0;
	if (!img1 || !img2 || !weights || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) {
		set_error("gsr_ssim_l1_backward: NULL argument");
		return GSR_E_INVALID;
	}
	static const SsimWindow win = make_window();
	dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C);
	{
		StageTimer st_(GSR_STAGE_LOSS_BWD, stream);
		ssim_l1_bwd_kernel<<<grid, 256, 0, stream>>>(img1, img2, H, W, win, weights, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
	}
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_adam_step_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, const gsr_adam_segment* segments,
                                   int num_segments, float beta1, float beta2, float eps, int step, uint64_t range_begin, uint64_t range_end,
                                   void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (n == 0) return 0;
	if (range_begin > range_end || range_end > n || (range_begin & 3u) != 0 || (range_end != n && (range_end & 3u) != 0)) {
		set_error("gsr_adam_step_range: [range_begin, range_end) must lie inside [0, n) with multiples of 4 as bounds");
		return GSR_E_INVALID;
	}
	if (!param || !grad || !exp_avg || !exp_avg_sq || !segments || num_segments < 1 || num_segments > ADAM_MAX_SEG || step < 1) {
		set_error("gsr_adam_step: invalid argument (NULL buffer, step < 1 or more than 16 segments)");
		return GSR_E_INVALID;
	}
	if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) != 0) {
		set_error("gsr_adam_step: buffers must be 16-byte aligned");
		return GSR_E_INVALID;
	}
	AdamSegs s;
	memset(&s, 0, sizeof(s));
	s.n = num_segments;
	uint64_t prev = 0;
	for (int i = 0; i < num_segments; i++) {
		if (segments[i].begin != prev || segments[i].end < segments[i].begin || (segments[i].period != 0 && segments[i].split > segments[i].period)) {
			set_error("gsr_adam_step: segments must tile [0, n) in order");
			return GSR_E_INVALID;
		}
		prev = segments[i].end;
		s.end[i] = segments[i].end; s.lr[i] = segments[i].lr; s.lr2[i] = segments[i].lr2;
		s.period[i] = segments[i].period; s.split[i] = segments[i].split;
	}
	if (prev != n) { set_error("gsr_adam_step: segments must tile [0, n) in order"); return GSR_E_INVALID; }
	// bias corrections in double on the host, as Python floats are in torch/optim/adam.py
	const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
	if (range_begin == range_end) return 0;
	const unsigned long long nvec = (range_end - range_begin + 3) / 4;
	{
		StageTimer st_(GSR_STAGE_ADAM, stream);
		adam_kernel<<<(unsigned)((nvec + 255) / 256), 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, range_begin, range_end, s, beta1, beta2, eps,
		                                                               (float)bc1, (float)sqrt(bc2));
	}
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, const gsr_adam_segment* segments,
                             int num_segments, float beta1, float beta2, float eps, int step, void* stream_) {
	return gsr_adam_step_range(param, grad, exp_avg, exp_avg_sq, n, segments, num_segments, beta1, beta2, eps, step, 0, n, stream_);
}
