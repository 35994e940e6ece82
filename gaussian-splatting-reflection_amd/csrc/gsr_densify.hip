// Densification bookkeeping: the data-movement side of the reference's adaptive density control
// (scene/gaussian_model.py:418-584, train.py:239-255).  The reference edits every parameter tensor and both Adam moment
// tensors of every group once per operation (prune -> clone -> split -> prune: ~80 boolean-index / cat passes per
// densification step).  Here the host composes ONE row map `new row -> old row` (the masks are a few P-sized boolean
// vectors) and the device applies it in ONE streaming gather per flat buffer; the only arithmetic is the per-view
// statistics update and the sampling of split children.
#include "gsr_internal.hpp"

namespace gsr {

// add_densification_stats (scene/gaussian_model.py:578-584) + the max_radii2D update of train.py:243, one pass over P.
__global__ void __launch_bounds__(256)
densify_stats_kernel(int P, const float* __restrict__ grad_means2D, const int* __restrict__ radii, const float* __restrict__ weights,
                     float* __restrict__ xyz_gradient_accum, float* __restrict__ denom, float* __restrict__ accum_w, float* __restrict__ denom_w,
                     float* __restrict__ max_radii2D) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P) return;
	const int r = radii[i];
	if (r > 0) {   // visibility_filter = radii > 0
		const float gx = grad_means2D[3 * i], gy = grad_means2D[3 * i + 1], gz = grad_means2D[3 * i + 2];
		xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy + gz * gz);   // torch.norm(grad[update_filter], dim=-1)
		denom[i] += 1.0f;
		max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
	}
	const float w = weights[i];
	if (w > 0.0f) {
		accum_w[i] += w;
		denom_w[i] += 1.0f;
	}
}

// Row gather over several row-major blocks that share one row map: dst_g[r, :] = map[r] >= 0 ? src_g[map[r], :] : 0.
#define GATHER_MAX_GROUPS 16
struct GatherGroups {
	int n;
	unsigned long long src_off[GATHER_MAX_GROUPS], dst_off[GATHER_MAX_GROUPS];   // float offsets of the blocks
	unsigned int width[GATHER_MAX_GROUPS];                                          // floats per row
	unsigned long long first[GATHER_MAX_GROUPS + 1];                                // prefix of n_rows * width (work items)
};
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ row_map, GatherGroups g) {
	const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	if (t >= g.first[g.n]) return;
	int k = 0;
	while (t >= g.first[k + 1]) k++;
	const unsigned long long e = t - g.first[k];
	const unsigned int w = g.width[k];
	const unsigned long long r = e / w, c = e - r * w;
	const int s = row_map[r];
	dst[g.dst_off[k] + e] = s >= 0 ? src[g.src_off[k] + (unsigned long long)s * w + c] : 0.0f;
}

// Children of densify_and_split (scene/gaussian_model.py:508-534): child j of selected surfel s gets
//   xyz = R(q_s) . (exp(scale_s) * noise_j, 0) + xyz_s,   scaling = log(exp(scale_s) / (0.8 N))
// R = build_rotation (utils/general_utils.py:78-99: quaternion normalised by its norm).  `noise` is standard normal,
// supplied by the caller (the reference draws torch.normal(0, stds) = stds * N(0,1)); S = 2 (surfel) or 3 scale components.
__global__ void __launch_bounds__(256)
split_children_kernel(int n_children, int S, int N, const int* __restrict__ parent, const float* __restrict__ xyz, const float* __restrict__ scaling,
                      const float* __restrict__ rotation, const float* __restrict__ noise, float* __restrict__ child_xyz,
                      float* __restrict__ child_scaling) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n_children) return;
	const int p = parent[j];
	float s[3] = {0.f, 0.f, 0.f}, smp[3] = {0.f, 0.f, 0.f};
	for (int k = 0; k < S; k++) {
		s[k] = expf(scaling[(size_t)p * S + k]);
		smp[k] = s[k] * noise[(size_t)j * S + k];
		child_scaling[(size_t)j * S + k] = logf(s[k] / (0.8f * (float)N));
	}
	const float qr = rotation[4 * (size_t)p], qx = rotation[4 * (size_t)p + 1], qy = rotation[4 * (size_t)p + 2], qz = rotation[4 * (size_t)p + 3];
	const float nrm = sqrtf(qr * qr + qx * qx + qy * qy + qz * qz);
	const float r = qr / nrm, x = qx / nrm, y = qy / nrm, z = qz / nrm;
	const float R00 = 1 - 2 * (y * y + z * z), R01 = 2 * (x * y - r * z), R02 = 2 * (x * z + r * y);
	const float R10 = 2 * (x * y + r * z), R11 = 1 - 2 * (x * x + z * z), R12 = 2 * (y * z - r * x);
	const float R20 = 2 * (x * z - r * y), R21 = 2 * (y * z + r * x), R22 = 1 - 2 * (x * x + y * y);
	child_xyz[3 * (size_t)j] = R00 * smp[0] + R01 * smp[1] + R02 * smp[2] + xyz[3 * (size_t)p];
	child_xyz[3 * (size_t)j + 1] = R10 * smp[0] + R11 * smp[1] + R12 * smp[2] + xyz[3 * (size_t)p + 1];
	child_xyz[3 * (size_t)j + 2] = R20 * smp[0] + R21 * smp[1] + R22 * smp[2] + xyz[3 * (size_t)p + 2];
}

}  // namespace gsr

using namespace gsr;

extern "C" int gsr_densification_stats(int P, const float* grad_means2D, const int* radii, const float* gaussian_weights, float* xyz_gradient_accum,
                                       float* denom, float* accum_w, float* denom_w, float* max_radii2D, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (P < 0) { set_error("gsr_densification_stats: invalid argument"); return GSR_E_INVALID; }
	if (P == 0) return 0;
	if (!grad_means2D || !radii || !gaussian_weights || !xyz_gradient_accum || !denom || !accum_w || !denom_w || !max_radii2D) {
		set_error("gsr_densification_stats: NULL buffer");
		return GSR_E_INVALID;
	}
	densify_stats_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, grad_means2D, radii, gaussian_weights, xyz_gradient_accum, denom, accum_w, denom_w,
	                                                         max_radii2D);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_gather_rows(const float* src, float* dst, const int* row_map, uint64_t n_rows, const gsr_gather_group* groups, int num_groups,
                               void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (num_groups < 0 || num_groups > GATHER_MAX_GROUPS) { set_error("gsr_gather_rows: at most 16 groups"); return GSR_E_INVALID; }
	if (n_rows == 0 || num_groups == 0) return 0;
	if (!src || !dst || !row_map || !groups) { set_error("gsr_gather_rows: NULL argument"); return GSR_E_INVALID; }
	GatherGroups g;
	g.n = num_groups;
	g.first[0] = 0;
	for (int k = 0; k < num_groups; k++) {
		if (groups[k].width == 0) { set_error("gsr_gather_rows: zero-width group"); return GSR_E_INVALID; }
		g.src_off[k] = groups[k].src_offset; g.dst_off[k] = groups[k].dst_offset; g.width[k] = groups[k].width;
		g.first[k + 1] = g.first[k] + n_rows * (unsigned long long)groups[k].width;
	}
	const unsigned long long total = g.first[num_groups];
	gather_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(src, dst, row_map, g);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

extern "C" int gsr_split_children(int n_children, int scale_dims, int N, const int* parent, const float* xyz, const float* scaling,
                                  const float* rotation, const float* noise, float* child_xyz, float* child_scaling, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	if (n_children < 0 || (scale_dims != 2 && scale_dims != 3) || N < 1) { set_error("gsr_split_children: invalid argument"); return GSR_E_INVALID; }
	if (n_children == 0) return 0;
	if (!parent || !xyz || !scaling || !rotation || !noise || !child_xyz || !child_scaling) { set_error("gsr_split_children: NULL buffer"); return GSR_E_INVALID; }
	split_children_kernel<<<(n_children + 255) / 256, 256, 0, stream>>>(n_children, scale_dims, N, parent, xyz, scaling, rotation, noise, child_xyz,
	                                                                   child_scaling);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}
