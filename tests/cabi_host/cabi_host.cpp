// Test infrastructure: a torch-free host program against the C ABI of libgsr_hip.so (include/gsr_hip.h) — what a maintainer's binding
// in any language does: device memory from the HIP runtime, the three workspace buffers through the allocation callback, plain pointers
// and sizes in, num_rendered out.  tests/test_gpu_cabi_host.py writes the scene as raw little-endian arrays, builds this file with
// hipcc, runs it and compares its outputs with the Python path.
//   cabi_host <dir>: reads <dir>/{meta.txt, means3D, shs, opacities, scales, rotations, refl, mask, view, proj, campos, bg, g_color, g_others, g_refl}.bin
//   writes <dir>/out_{color, others, refl, radii, weights, dmeans3D, dsh, dopacity, dscales, drot, drefl, dmeans2D}.bin and num_rendered.txt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gsr_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static std::vector<char> slurp(const std::string& path) {
	std::vector<char> v;
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(3); }
	fseek(f, 0, SEEK_END);
	v.resize((size_t)ftell(f));
	fseek(f, 0, SEEK_SET);
	if (!v.empty() && fread(v.data(), 1, v.size(), f) != v.size()) exit(3);
	fclose(f);
	return v;
}
template <class T> static T* upload(const std::string& dir, const char* name, size_t count) {
	const std::vector<char> h = slurp(dir + "/" + name + ".bin");
	if (h.size() != count * sizeof(T)) { fprintf(stderr, "%s: %zu bytes, expected %zu\n", name, h.size(), count * sizeof(T)); exit(3); }
	T* d = nullptr;
	if (hipMalloc((void**)&d, h.size() ? h.size() : 4) != hipSuccess || hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) exit(2);
	return d;
}
template <class T> static T* device(size_t count) {
	T* d = nullptr;
	if (hipMalloc((void**)&d, count ? count * sizeof(T) : 4) != hipSuccess) exit(2);
	return d;
}
template <class T> static void download(const std::string& dir, const char* name, const T* d, size_t count) {
	std::vector<T> h(count);
	if (hipMemcpy(h.data(), d, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) exit(2);
	FILE* f = fopen((dir + "/out_" + name + ".bin").c_str(), "wb");
	if (!f || fwrite(h.data(), sizeof(T), count, f) != count) exit(3);
	fclose(f);
}

struct Workspace { void* buf[3] = {nullptr, nullptr, nullptr}; };
static void* resize(void* user, int which, size_t bytes) {      // the reference's three resizeFunctional callbacks (rasterize_points.cu:31-37) as one
	Workspace* w = static_cast<Workspace*>(user);
	if (which < 0 || which > 2) return nullptr;
	if (w->buf[which]) (void)hipFree(w->buf[which]);
	if (hipMalloc(&w->buf[which], bytes ? bytes : 256) != hipSuccess) return nullptr;
	return w->buf[which];
}

int main(int argc, char** argv) {
	if (argc != 2) { fprintf(stderr, "usage: cabi_host <dir>\n"); return 1; }
	const std::string dir = argv[1];
	int P = 0, D = 0, M = 0, W = 0, H = 0;
	float tanx = 0, tany = 0;
	{
		FILE* f = fopen((dir + "/meta.txt").c_str(), "r");
		if (!f || fscanf(f, "%d %d %d %d %d %f %f", &P, &D, &M, &W, &H, &tanx, &tany) != 7) { fprintf(stderr, "bad meta.txt\n"); return 3; }
		fclose(f);
	}
	const size_t HW = (size_t)W * H;
	float* means = upload<float>(dir, "means3D", (size_t)P * 3);
	float* shs = upload<float>(dir, "shs", (size_t)P * M * 3);
	float* opac = upload<float>(dir, "opacities", P);
	float* scales = upload<float>(dir, "scales", (size_t)P * 2);
	float* rot = upload<float>(dir, "rotations", (size_t)P * 4);
	float* refl = upload<float>(dir, "refl", P);
	uint8_t* mask = upload<uint8_t>(dir, "mask", P);
	float* view = upload<float>(dir, "view", 16);
	float* proj = upload<float>(dir, "proj", 16);
	float* campos = upload<float>(dir, "campos", 3);
	float* bg = upload<float>(dir, "bg", 3);
	float* g_color = upload<float>(dir, "g_color", HW * 3);
	float* g_others = upload<float>(dir, "g_others", HW * 8);
	float* g_refl = upload<float>(dir, "g_refl", HW);
	float *color = device<float>(HW * 3), *others = device<float>(HW * 8), *reflmap = device<float>(HW), *weights = device<float>(P);
	int* radii = device<int>(P);
	hipStream_t stream;
	CHECK_HIP(hipStreamCreate(&stream));
	Workspace ws;
	const int R = gsr_surfel_forward(resize, &ws, P, D, M, bg, W, H, means, mask, shs, nullptr, refl, opac, scales, 1.0f, rot, nullptr, view, proj, campos, tanx,
	                                 tany, 0, color, others, reflmap, radii, weights, 0, stream);
	if (R < 0) { fprintf(stderr, "gsr_surfel_forward: %d (%s)\n", R, gsr_last_error()); return 4; }
	float *dmeans2D = device<float>((size_t)P * 3), *dopacity = device<float>(P), *drefl = device<float>(P), *dmeans3D = device<float>((size_t)P * 3);
	float *dsh = device<float>((size_t)P * M * 3), *dscales = device<float>((size_t)P * 2), *drot = device<float>((size_t)P * 4);
	// per-view gradients of inputs that were not supplied (colors_precomp, transMat_precomp) and the internal dL_dnormal: NULL = not wanted
	const int rc = gsr_surfel_backward(P, D, M, R, bg, W, H, means, shs, nullptr, refl, scales, 1.0f, rot, nullptr, view, proj, campos, tanx, tany, radii, ws.buf[0],
	                                   ws.buf[1], ws.buf[2], g_color, g_others, g_refl, dmeans2D, nullptr, dopacity, nullptr, drefl, dmeans3D, nullptr, dsh, dscales,
	                                   drot, 0, stream);
	if (rc < 0) { fprintf(stderr, "gsr_surfel_backward: %d (%s)\n", rc, gsr_last_error()); return 5; }
	CHECK_HIP(hipStreamSynchronize(stream));
	download(dir, "color", color, HW * 3); download(dir, "others", others, HW * 8); download(dir, "refl", reflmap, HW);
	download(dir, "radii", radii, P); download(dir, "weights", weights, P);
	download(dir, "dmeans2D", dmeans2D, (size_t)P * 3); download(dir, "dopacity", dopacity, P); download(dir, "drefl", drefl, P);
	download(dir, "dmeans3D", dmeans3D, (size_t)P * 3); download(dir, "dsh", dsh, (size_t)P * M * 3); download(dir, "dscales", dscales, (size_t)P * 2);
	download(dir, "drot", drot, (size_t)P * 4);
	FILE* f = fopen((dir + "/num_rendered.txt").c_str(), "w");
	if (!f) return 3;
	fprintf(f, "%d\n", R);
	fclose(f);
	// a mistake must come back as a code and a message, not as a crash
	const int bad = gsr_surfel_forward(resize, &ws, P, D, M, bg, W, H, nullptr, mask, shs, nullptr, refl, opac, scales, 1.0f, rot, nullptr, view, proj, campos, tanx,
	                                   tany, 0, color, others, reflmap, radii, weights, 0, stream);
	if (bad != GSR_E_INVALID || std::string(gsr_last_error()).empty()) { fprintf(stderr, "missing pointer: expected GSR_E_INVALID, got %d\n", bad); return 6; }
	printf("cabi_host: num_rendered %d\n", R);
	return 0;
}
