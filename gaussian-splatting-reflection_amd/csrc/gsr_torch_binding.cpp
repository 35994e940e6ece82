// Compiled torch/pybind binding over the C ABI of libgsr_hip.so (optional: the package works without it through ctypes).
//
// This is what remains of the reference's extension sources once the kernels live behind include/gsr_hip.h: marshaling only.
// It exports, per variant, the three functions of the reference's pybind modules with their positional arguments and return
// tuples:
//   S:  submodules/diff-surfel-rasterization/ext.cpp:15-19   (rasterize_points.cu:39-151 forward, 153-267 backward, 269-288 markVisible)
//   G:  submodules/diff-gaussian-rasterization/ext.cpp:15-19 (rasterize_points.cu:38-140, 142-264, 266-285)
// as surfel_rasterize_gaussians / surfel_rasterize_gaussians_backward / gauss_... / mark_visible of one module `_gsr_C`.
// Differences from the reference's marshaling, all permitted by the library: gradient tensors are torch::empty (every
// element is written), the stream is torch's current stream (the reference uses the legacy default stream), the three
// std::function resize callbacks are one C callback.
// Build: python gaussian-splatting-reflection_amd/csrc/build.py --binding   (hipcc, links libgsr_hip.so).  Select with GSR_BINDING=pybind.
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <tuple>
#include <vector>

#include "../../include/gsr_hip.h"

namespace {

struct Workspace {
	torch::Tensor bufs[3];
	explicit Workspace(const torch::Tensor& like) {
		for (auto& b : bufs) b = torch::empty({0}, like.options().dtype(torch::kByte));
	}
};
void* resize_cb(void* user, int which, size_t bytes) {   // replaces resizeFunctional (DSR rasterize_points.cu:31-37)
	auto* ws = static_cast<Workspace*>(user);
	if (which < 0 || which > 2) return nullptr;
	ws->bufs[which].resize_({(int64_t)(bytes < 256 ? 256 : bytes)});
	return ws->bufs[which].data_ptr();
}

// contiguous float view; an empty tensor maps to NULL (the reference's `.contiguous().data<float>()` is nullptr then)
struct FloatArg {
	torch::Tensor keep;
	const float* p;
	FloatArg(const torch::Tensor& t, const char* name) : p(nullptr) {
		if (t.numel() == 0) return;
		TORCH_CHECK(t.scalar_type() == torch::kFloat32, "expected scalar type Float but found ", t.scalar_type(), " for ", name);
		keep = t.contiguous();
		p = keep.data_ptr<float>();
	}
};
// (PyTorch-ROCm presents HIP devices as "cuda": the guards and stream getters are the ...MasqueradingAsCUDA flavours)
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
void* current_stream(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }
void check(int rc, const char* what) { TORCH_CHECK(rc >= 0, what, " failed (code ", rc, "): ", gsr_last_error()); }

#define REQUIRE_CUDA(t) TORCH_CHECK((t).is_cuda(), #t " must be a CUDA tensor")

}  // namespace

// ---------------------------------------------------------------------------------------------- variant S
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
SurfelForward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& env_scope_mask, const torch::Tensor& colors,
              const torch::Tensor& refl_strengths, const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
              const float scale_modifier, const torch::Tensor& transMat_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
              const float tan_fovx, const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
              const torch::Tensor& campos, const bool prefiltered, const bool debug) {
	TORCH_CHECK(means3D.ndimension() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
	REQUIRE_CUDA(background); REQUIRE_CUDA(means3D); REQUIRE_CUDA(colors); REQUIRE_CUDA(refl_strengths); REQUIRE_CUDA(opacity); REQUIRE_CUDA(scales);
	REQUIRE_CUDA(rotations); REQUIRE_CUDA(transMat_precomp); REQUIRE_CUDA(viewmatrix); REQUIRE_CUDA(projmatrix); REQUIRE_CUDA(sh); REQUIRE_CUDA(campos);
	const int P = (int)means3D.size(0), H = image_height, W = image_width;
	const DeviceGuard guard(means3D.device());
	auto f = means3D.options().dtype(torch::kFloat32);
	auto out_color = torch::empty({3, H, W}, f), out_others = torch::empty({8, H, W}, f), out_refl = torch::empty({1, H, W}, f);
	auto radii = torch::empty({P}, means3D.options().dtype(torch::kInt32)), gw = torch::empty({P}, f);
	Workspace ws(means3D);
	const int M = sh.numel() ? (int)sh.size(1) : 0;
	torch::Tensor mask;
	const uint8_t* mask_p = nullptr;
	if (env_scope_mask.numel()) {
		TORCH_CHECK(env_scope_mask.scalar_type() == torch::kBool, "expected scalar type Bool but found ", env_scope_mask.scalar_type(), " for env_scope_mask");
		mask = env_scope_mask.contiguous();
		mask_p = reinterpret_cast<const uint8_t*>(mask.data_ptr<bool>());
	}
	FloatArg bg(background, "background"), m3(means3D, "means3D"), shc(sh, "sh"), col(colors, "colors"), refl(refl_strengths, "refl_strengths"),
	    opa(opacity, "opacity"), sca(scales, "scales"), rot(rotations, "rotations"), tm(transMat_precomp, "transMat_precomp"),
	    vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix"), cp(campos, "campos");
	const int rendered = gsr_surfel_forward(resize_cb, &ws, P, degree, M, bg.p, W, H, m3.p, mask_p, shc.p, col.p, refl.p, opa.p, sca.p, scale_modifier, rot.p,
	                                        tm.p, vm.p, pm.p, cp.p, tan_fovx, tan_fovy, prefiltered, out_color.data_ptr<float>(),
	                                        out_others.data_ptr<float>(), out_refl.data_ptr<float>(), radii.data_ptr<int>(), gw.data_ptr<float>(),
	                                        debug, current_stream(means3D));
	check(rendered, "gsr_surfel_forward");
	return std::make_tuple(rendered, out_color, out_others, radii, ws.bufs[0], ws.bufs[1], ws.bufs[2], out_refl, gw);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
SurfelBackward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& colors,
               const torch::Tensor& refl_strengths, const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
               const torch::Tensor& transMat_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_others,
               const torch::Tensor& dL_dout_refl_strength_map, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
               const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug,
               const int unused) {
	// unused (extension; 0 = the reference's behaviour): bit 0 = dL_dcolors, bit 1 = dL_dtransMat belong to inputs the caller did not supply
	// and are neither computed nor allocated (empty tensors in the tuple).  dL_dnormal3D is internal to the reference's backward: never allocated.
	const int P = (int)means3D.size(0), H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
	const int M = sh.numel() ? (int)sh.size(1) : 0;
	const DeviceGuard guard(means3D.device());
	auto f = means3D.options().dtype(torch::kFloat32);
	auto mk = [&](std::vector<int64_t> shape) { return P ? torch::empty(shape, f) : torch::zeros(shape, f); };   // the library writes every element
	const bool no_col = (unused & 1) && sh.numel() != 0, no_tm = (unused & 2) && scales.numel() != 0;
	auto dL_dmeans3D = mk({P, 3}), dL_dmeans2D = mk({P, 3}), dL_dcolors = no_col ? torch::empty({0}, f) : mk({P, 3}), dL_dopacity = mk({P, 1}),
	     dL_dtransMat = no_tm ? torch::empty({0}, f) : mk({P, 9}), dL_dsh = mk({P, M, 3}), dL_dscales = mk({P, 2}), dL_drotations = mk({P, 4}),
	     dL_drefl = mk({P, 1});
	if (P != 0) {
		torch::Tensor grefl_t = dL_dout_refl_strength_map.numel() ? dL_dout_refl_strength_map : torch::zeros({1, H, W}, f);
		FloatArg bg(background, "background"), m3(means3D, "means3D"), shc(sh, "sh"), col(colors, "colors"), refl(refl_strengths, "refl_strengths"),
		    sca(scales, "scales"), rot(rotations, "rotations"), tm(transMat_precomp, "transMat_precomp"), vm(viewmatrix, "viewmatrix"),
		    pm(projmatrix, "projmatrix"), cp(campos, "campos"), gcol(dL_dout_color, "dL_dout_color"), goth(dL_dout_others, "dL_dout_others"),
		    grefl(grefl_t, "dL_dout_refl_strength_map");
		auto rad = radii.contiguous();
		check(gsr_surfel_backward(P, degree, M, R, bg.p, W, H, m3.p, shc.p, col.p, refl.p, sca.p, scale_modifier, rot.p, tm.p, vm.p, pm.p, cp.p, tan_fovx,
		                          tan_fovy, rad.data_ptr<int>(), geomBuffer.data_ptr(), binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr,
		                          imageBuffer.data_ptr(), gcol.p, goth.p, grefl.p, dL_dmeans2D.data_ptr<float>(), nullptr,
		                          dL_dopacity.data_ptr<float>(), no_col ? nullptr : dL_dcolors.data_ptr<float>(), dL_drefl.data_ptr<float>(),
		                          dL_dmeans3D.data_ptr<float>(), no_tm ? nullptr : dL_dtransMat.data_ptr<float>(), M ? dL_dsh.data_ptr<float>() : nullptr,
		                          dL_dscales.data_ptr<float>(),
		                          dL_drotations.data_ptr<float>(), debug, current_stream(means3D)),
		      "gsr_surfel_backward");
	}
	return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_drefl, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations);
}

// ---------------------------------------------------------------------------------------------- variant G
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
GaussForward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& normals,
             const torch::Tensor& refl_strengths, const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
             const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
             const float tan_fovx, const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
             const torch::Tensor& campos, const bool prefiltered, const bool antialiasing, const bool debug) {
	TORCH_CHECK(means3D.ndimension() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
	REQUIRE_CUDA(means3D);
	const int P = (int)means3D.size(0), H = image_height, W = image_width;
	const DeviceGuard guard(means3D.device());
	auto f = means3D.options().dtype(torch::kFloat32);
	auto out_color = torch::empty({3, H, W}, f), out_normal = torch::empty({3, H, W}, f), out_inv = torch::empty({1, H, W}, f), out_refl = torch::empty({1, H, W}, f);
	auto radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
	Workspace ws(means3D);
	const int M = sh.numel() ? (int)sh.size(1) : 0;
	FloatArg bg(background, "background"), m3(means3D, "means3D"), shc(sh, "sh"), col(colors, "colors"), nrm(normals, "normals"),
	    refl(refl_strengths, "refl_strengths"), opa(opacity, "opacity"), sca(scales, "scales"), rot(rotations, "rotations"), cov(cov3D_precomp, "cov3D_precomp"),
	    vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix"), cp(campos, "campos");
	const int rendered = gsr_gauss_forward(resize_cb, &ws, P, degree, M, bg.p, W, H, m3.p, shc.p, col.p, nrm.p, refl.p, opa.p, sca.p, scale_modifier, rot.p, cov.p,
	                                       vm.p, pm.p, cp.p, tan_fovx, tan_fovy, prefiltered, out_color.data_ptr<float>(), out_normal.data_ptr<float>(),
	                                       out_refl.data_ptr<float>(), out_inv.data_ptr<float>(), antialiasing, radii.data_ptr<int>(), debug,
	                                       current_stream(means3D));
	check(rendered, "gsr_gauss_forward");
	return std::make_tuple(rendered, out_color, radii, ws.bufs[0], ws.bufs[1], ws.bufs[2], out_inv, out_normal, out_refl);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
GaussBackward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& colors,
              const torch::Tensor& normals, const torch::Tensor& refl_strengths, const torch::Tensor& opacities, const torch::Tensor& scales,
              const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
              const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const torch::Tensor& dL_dout_color,
              const torch::Tensor& dL_dout_invdepth, const torch::Tensor& dL_dout_normal_map, const torch::Tensor& dL_dout_refl_strength_map,
              const torch::Tensor& sh, const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
              const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool antialiasing, const bool debug, const int unused) {
	// unused: bit 0 = dL_dcolors, bit 1 = dL_dcov3D (see SurfelBackward).  dL_dmean2D / dL_dconic are intermediates the reference's binding
	// never returns (DGR rasterize_points.cu:263): not allocated.
	const int P = (int)means3D.size(0), H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
	const int M = sh.numel() ? (int)sh.size(1) : 0;
	const DeviceGuard guard(means3D.device());
	auto f = means3D.options().dtype(torch::kFloat32);
	auto mk = [&](std::vector<int64_t> shape) { return P ? torch::empty(shape, f) : torch::zeros(shape, f); };
	const bool no_col = (unused & 1) && sh.numel() != 0, no_cov = (unused & 2) && scales.numel() != 0;
	auto dL_dmeans3D = mk({P, 3}), dL_dmeans2D_pixels = mk({P, 3}), dL_dcolors = no_col ? torch::empty({0}, f) : mk({P, 3}), dL_dnormals = mk({P, 3}),
	     dL_dopacity = mk({P, 1}), dL_dcov3D = no_cov ? torch::empty({0}, f) : mk({P, 6}), dL_dsh = mk({P, M, 3}), dL_dscales = mk({P, 3}),
	     dL_drotations = mk({P, 4}), dL_drefl = mk({P, 1});
	// depth / refl-strength backward are active whenever the incoming grad tensors are non-empty (DGR rasterize_points.cu:196-216)
	const bool has_inv = dL_dout_invdepth.numel() != 0, has_refl = dL_dout_refl_strength_map.numel() != 0;
	auto dL_dinvdepths = has_inv ? mk({P, 1}) : torch::zeros({0, 1}, f);
	if (P != 0) {
		torch::Tensor grefl_t = has_refl ? dL_dout_refl_strength_map : torch::zeros({1, H, W}, f);
		FloatArg bg(background, "background"), m3(means3D, "means3D"), shc(sh, "sh"), col(colors, "colors"), nrm(normals, "normals"),
		    refl(refl_strengths, "refl_strengths"), opa(opacities, "opacities"), sca(scales, "scales"), rot(rotations, "rotations"),
		    cov(cov3D_precomp, "cov3D_precomp"), vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix"), cp(campos, "campos"),
		    gcol(dL_dout_color, "dL_dout_color"), gnrm(dL_dout_normal_map, "dL_dout_normal_map"), grefl(grefl_t, "dL_dout_refl_strength_map"),
		    ginv(dL_dout_invdepth, "dL_dout_invdepth");
		auto rad = radii.contiguous();
		check(gsr_gauss_backward(P, degree, M, R, bg.p, W, H, m3.p, shc.p, col.p, nrm.p, refl.p, opa.p, sca.p, scale_modifier, rot.p, cov.p, vm.p, pm.p, cp.p,
		                         tan_fovx, tan_fovy, rad.data_ptr<int>(), geomBuffer.data_ptr(), binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr,
		                         imageBuffer.data_ptr(), gcol.p, gnrm.p, grefl.p, has_inv ? ginv.p : nullptr, nullptr,
		                         dL_dmeans2D_pixels.data_ptr<float>(), nullptr, dL_dopacity.data_ptr<float>(), no_col ? nullptr : dL_dcolors.data_ptr<float>(),
		                         dL_dnormals.data_ptr<float>(), dL_drefl.data_ptr<float>(), has_inv ? dL_dinvdepths.data_ptr<float>() : nullptr,
		                         dL_dmeans3D.data_ptr<float>(), no_cov ? nullptr : dL_dcov3D.data_ptr<float>(), M ? dL_dsh.data_ptr<float>() : nullptr,
		                         dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(), antialiasing, debug, current_stream(means3D)),
		      "gsr_gauss_backward");
	}
	if (!has_refl) dL_drefl = torch::zeros({0, 1}, f);
	return std::make_tuple(dL_dmeans2D_pixels, dL_dcolors, dL_dnormals, dL_drefl, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor MarkVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
	const int P = (int)means3D.size(0);
	auto present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
	if (P != 0) {
		const DeviceGuard guard(means3D.device());
		FloatArg m3(means3D, "means3D"), vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix");
		check(gsr_mark_visible(P, m3.p, vm.p, pm.p, reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), current_stream(means3D)), "gsr_mark_visible");
	}
	return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
	m.def("surfel_rasterize_gaussians", &SurfelForward);
	m.def("surfel_rasterize_gaussians_backward", &SurfelBackward);      // (last argument: `unused` bit mask, 0 = the reference's behaviour)
	m.def("gauss_rasterize_gaussians", &GaussForward);
	m.def("gauss_rasterize_gaussians_backward", &GaussBackward);
	m.def("mark_visible", &MarkVisible);
}
