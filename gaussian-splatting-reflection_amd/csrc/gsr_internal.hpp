// Internal declarations shared by the gfx950 kernels of libgsr_hip.so.
// Wave = 64 lanes everywhere; tiles are 16x16 pixels = 4 waves of 8x8 pixels.
#pragma once
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <hip/hip_runtime.h>
#include "../../include/gsr_hip.h"

#define GSR_TILE 16
#define GSR_TILE_PIX 256

namespace gsr {

// ------------------------------------------------------------------ error plumbing
void set_error(const char* fmt, ...);
#define GSR_HIP_CHECK(expr)                                                                       \
	do {                                                                                          \
		hipError_t _e = (expr);                                                                   \
		if (_e != hipSuccess) {                                                                   \
			gsr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
			return GSR_E_HIP;                                                                     \
		}                                                                                         \
	} while (0)
// After a kernel launch: always catch launch errors; with debug also synchronise (reference CHECK_CUDA,
// DSR auxiliary.h:300-307).
#define GSR_LAUNCH_CHECK(debug, stream)                                  \
	do {                                                                 \
		GSR_HIP_CHECK(hipGetLastError());                                \
		if (debug) GSR_HIP_CHECK(hipStreamSynchronize(stream));          \
	} while (0)

// 128-bit accesses: the per-Gaussian kernels read an SH row whose length is a multiple of 16 bytes as float4, and store dL_dsh rows
// (M = 16) and dL_drot rows as float4.  torch allocations are 256-byte aligned; VIEWS into a packed buffer (gradient sinks) need not be.
// Checked at the C ABI so that a misaligned caller gets GSR_E_INVALID instead of a faulting or sector-splitting kernel.
static inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
#define GSR_REQUIRE_ALIGNED16(ptr, what)                                                                      \
	do {                                                                                                      \
		if ((ptr) && gsr::misaligned16(ptr)) {                                                                \
			gsr::set_error("%s: %s must be 16-byte aligned (float4 accesses); got %p", __func__, what, (const void*)(ptr)); \
			return GSR_E_INVALID;                                                                             \
		}                                                                                                     \
	} while (0)

// ------------------------------------------------------------------ per-stage timing (off by default)
// RAII: records a start event at construction and a stop event at destruction on `stream` when profiling
// is enabled (gsr_profile_enable); a no-op otherwise.
struct StageTimer {
	int stage;
	hipStream_t stream;
	void* rec;
	StageTimer(int stage, hipStream_t stream);
	~StageTimer();
};

// ------------------------------------------------------------------ workspace carving
// Bump allocation with 256-byte alignment inside the three opaque buffers (the reference's
// obtain()/required(), DSR rasterizer_impl.h:21-73, with a private layout).
struct Carver {
	char* base;
	size_t off;
	explicit Carver(void* b) : base((char*)b), off(0) {}
	template <class T> T* take(size_t count) {
		off = (off + 255) & ~(size_t)255;
		T* p = base ? (T*)(base + off) : nullptr;
		off += count * sizeof(T);
		return p;
	}
	size_t size() const { return (off + 255) & ~(size_t)255; }
};

// Per-Gaussian state common to both variants.  `rec` is the render record gathered by the tile
// kernels: REC_F4 float4 per Gaussian (G: 4 = 64 B, S: 5 = 80 B), written by preprocess.
struct GeomState {
	float* depths;           // P        view-space z (sort key low word)
	uint32_t* rect;          // P        packed tile rect: xmin | ymin<<8 ... see pack_rect (2 words)
	uint32_t* tiles_touched; // P
	uint32_t* point_offsets; // P        inclusive scan
	uint8_t* clamped;        // P        bit c set = SH colour channel c clamped at 0
	float4* rec;             // P*REC_F4
	float4* bbox;            // 2P       cull record: conservative screen-space footprint of the pixels a Gaussian can
	                         //          blend into, {ex, ey, a, b}{c, disc_x, disc_y, disc_r2}: ellipse d^T[[a,b],[b,c]]d <= 1
	                         //          about (ex,ey) united with a disc; see cull_hit() below
	float* aux;              // G: cov3D P*6.  S: unused
	float* acc;              // backward accumulator P*ACC_F (zeroed by backward)
	int* flags;              // 4 ints: [0] prefiltered-trap flag, [2,3] num_rendered (64 bits): zeroed by the preprocess kernel, accumulated by
	                         //         gaussian_stats_kernel
	uint32_t* depth_sorted;  // P        depth bits in ascending order (output of the depth pre-sort; keys only)
	unsigned long long* order;   // P    at each position of the depth order (stable: ties by index): Gaussian index (low word) and its tile
	                         //          rectangle packed to 4 x 8 bits (high word; 0 on grids beyond 255 tiles per axis) — OrderRect in gsr_common.hip
	unsigned long long* emit_state;   // one word per 256 Gaussians: decoupled look-back state of emit_tiles_kernel's scan of the instance counts;
	size_t emit_state_bytes;          //   the LAST word is the scan's ticket counter (multiple of 16; zeroed by the preprocess kernel)
	void* scan_temp;         // temp of the two scans (shared) followed by the temp of the P-sized depth sort
	size_t scan_temp_bytes;
	void* depth_sort_temp;   // = scan_temp + scan part; its first depth_sort_clear bytes are zeroed by the preprocess kernel (they start with
	                         //   the digit counts of the depth sort, which gaussian_stats_kernel then accumulates)
	size_t depth_sort_bytes, depth_sort_clear;
};
struct ImageState {
	uint2* ranges;       // tiles
	float* final_T;      // planes_T * H*W
	uint32_t* n_contrib; // planes_n * H*W
	uint32_t* tile_order;    // tiles   tile ids by (approximately) descending list length: longest-first dispatch of the tile kernels
};
struct BinningState {
	uint32_t* point_list;        // R  Gaussian index of each instance, ordered by (tile, depth, index)   [first: the backward finds it without sizes]
	unsigned long long* blend_mask;  // 16 x mask_stride words, written by the forward tile kernel and read by the backward one, which then neither votes
	                                 // nor touches pairs that cannot contribute.  Batch b (64 list entries) of tile t is batch index range.x / 64 + t + b.
	                                 // Variant S: [batch index][quadrant][4x4 sub-block] = the entries that blended into at least one pixel of that
	                                 // sub-block; variant G: [quadrant][batch index] (first 4 x mask_stride words) per 8x8 quadrant.
	size_t mask_stride;          // R / 64 + tiles + 1
	uint32_t* tile_keys;         // R  tile id of each instance, sorted
	uint32_t* tile_keys_unsorted;
	uint32_t* vals_unsorted;
	void* sort_temp;
	size_t sort_temp_bytes;
};

GeomState carve_geom(void* buf, size_t P, int rec_f4, int aux_floats, int acc_floats, size_t scan_temp_bytes, size_t* total);
ImageState carve_image(void* buf, size_t HW, size_t tiles, int planes_T, int planes_n, size_t* total);
BinningState carve_binning(void* buf, size_t R, size_t tiles, size_t sort_temp_bytes, size_t* total);

int option_cull();   // 1 (default): per-wave bounding-box culling in the tile kernels; 0: evaluate every list entry
int option_dev();    // development ablation bits (0 in production): 1 = skip the gradient atomics of the surfel backward
size_t scan_temp_bytes(size_t P);
size_t sort_temp_bytes(size_t R, int end_bit);
int run_tile_order(const ImageState& img, size_t tiles, hipStream_t stream);
uint32_t higher_msb(uint32_t n);

// Binning pipeline shared by both variants (reference: DSR/DGR rasterizer_impl.cu:282-325):
// inclusive scan of tiles_touched -> num_rendered (pinned 4-byte readback) -> binning buffer via alloc
// -> key/value emission -> radix sort -> tile ranges.  Returns num_rendered or <0.
int run_binning(gsr_alloc_fn alloc, void* alloc_user, int P, int tiles_x, int tiles_y, const GeomState& geom,
                const ImageState& img, BinningState* out_binning, int prefiltered, int debug, hipStream_t stream);

// Texel-gradient tail of the deferred-reflection backward (sort of the per-pixel footprint records by texel, run combine, unpack:
// gsr_cubemap.hip) around the pixel kernel of gsr_deferred_reflection_backward*.  See refl_tail_begin in gsr_cubemap.hip.
struct ReflTail {
	float* staging;          // [6][L][L][4] channel-interleaved texel gradients (rim pixels add here directly; the combine adds the rest)
	float* fail_acc;         // 4 floats behind it: gradient of the fail value
	void* footprints;        // ReflFootprint[n], one 32-byte record per pixel
	uint32_t *keys_in, *keys_out, *pix_out;
	void* sort_temp;
	void* clear_ptr[2];      // what must be zero before the pixel code runs: [0] staging + fail_acc, [1] the sort's look-back state
	size_t clear_bytes[2];
	size_t n, ntex, sort_bytes;
	int key_bits;
	uint32_t L;
	const uint32_t* sort_keys;   // keys written by the forward, or NULL (then the pixel code writes keys_in)
	bool small_sort, locked;
	hipStream_t stream, tail;
	void* side;
	float *g_cubemap, *g_fail;
	int accumulate;
};
int refl_tail_begin(ReflTail& t, uint32_t L, int width, int height, float* scratch, size_t scratch_floats, const uint32_t* sort_keys, int async_tail,
                    int accumulate, float* g_cubemap, float* g_fail, hipStream_t stream);
int refl_tail_clear(ReflTail& t);
int refl_tail_sort(ReflTail& t);
int refl_tail_finish(ReflTail& t);
void refl_tail_abort(ReflTail& t);
int side_gate_wait(hipStream_t stream);
int refl_sort_keys_early(uint32_t L, int width, int height, float* scratch, size_t scratch_floats, const uint32_t* sort_keys, int async, hipStream_t stream);

}  // namespace gsr

// ------------------------------------------------------------------ device helpers
#ifdef __HIPCC__
namespace gsr {

__device__ __forceinline__ int f2i(float v) { return (int)v; }  // v_cvt_i32_f32: saturating, NaN -> 0

// getRect (DSR auxiliary.h:71-81 / DGR auxiliary.h:45-55)
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, uint32_t& x0, uint32_t& y0,
                                         uint32_t& x1, uint32_t& y1) {
	const float r = (float)max_radius;
	x0 = (uint32_t)min(gx, max(0, f2i((px - r) / 16.0f)));
	y0 = (uint32_t)min(gy, max(0, f2i((py - r) / 16.0f)));
	x1 = (uint32_t)min(gx, max(0, f2i((px + r + 15.0f) / 16.0f)));
	y1 = (uint32_t)min(gy, max(0, f2i((py + r + 15.0f) / 16.0f)));
}

// ---- wave64 sums.  Classic GCN reduction: row_shr 1,2,4,8 inside each row of 16 lanes, then
// row_bcast:15 / row_bcast:31 across the four rows; the total lands in lane 63.  hipcc does not fuse
// __builtin_amdgcn_update_dpp into the add (it emits v_mov_b32_dpp + v_pk_add_f32 + a zeroing move), so the
// multi-value forms below issue the fused `v_add_f32_dpp` directly: one VALU instruction per value per
// step.  Values are interleaved inside one asm statement so that consecutive DPP reads of a register are
// always >= 3 instructions after its last write (gfx9 VALU-write -> DPP-read hazard: 2 wait states; the
// compiler's hazard recogniser does not look inside asm, hence the leading s_nop).
#define GSR_DPP1(CTRL, i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " " CTRL "\n\t"
#define GSR_DPP4(CTRL) GSR_DPP1(CTRL, 0) GSR_DPP1(CTRL, 1) GSR_DPP1(CTRL, 2) GSR_DPP1(CTRL, 3)
#define GSR_SHR1 "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define GSR_SHR2 "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define GSR_SHR4 "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define GSR_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define GSR_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define GSR_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
// Sum 4 independent values across the wave; totals valid in lane 63 only.  Must be called with all 64 lanes active.
__device__ __forceinline__ void wave_sum4(float* v) {
	asm volatile("s_nop 1\n\t" GSR_DPP4(GSR_SHR1) GSR_DPP4(GSR_SHR2) GSR_DPP4(GSR_SHR4) GSR_DPP4(GSR_SHR8) GSR_DPP4(GSR_BC15) GSR_DPP4(GSR_BC31)
	             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// ---- in-row packed reduction (exchange-type DPP only).  An earlier version folded 64 -> 16 lanes with
// v_permlane32_swap / v_permlane16_swap (fewer instructions); measured on MI355X those swaps cost ~4x a DPP add, so
// that was the SLOWER way to pack on this part (ablation numbers in DESIGN.md).
// Four values (a, b, c, d) per group are reduced over each 16-lane row: level 1 pairs lane i with 15-i (row_mirror) and
// keeps a in lanes 0-7 / b in lanes 8-15, level 2 pairs i with 7-i inside each half (row_half_mirror) and keeps the first
// pair's result in lanes 0-3 of each half / the second pair's in lanes 4-7; two quad_perm adds finish the row sum.
// Result: every lane of quad q of a row holds the ROW total of value {a, c, b, d}[q]; the four row totals are combined
// by the caller (separate LDS rows, added at the flush).
#define GSR_Q1(CTRL, i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " " CTRL " row_mask:0xf bank_mask:0xf\n\t"
// No selects: DPP's bank_mask enables the destination per bank of four lanes, so level 1 is two adds into ONE register —
// `a + mirror(a)` written to lanes 0-7 (banks 0,1), `b + mirror(b)` to lanes 8-15 (banks 2,3) — and level 2 likewise with
// banks {0,2} / {1,3}.  2+2 (level 1) + 2 (level 2) = 6 instructions per group of four values, + 2 quad_perm adds per output.
#define GSR_DPPB(CTRL, BANK, dst, src) "v_add_f32_dpp %" #dst ", %" #src ", %" #src " " CTRL " row_mask:0xf bank_mask:" BANK "\n\t"
// The surfel backward's 20 values in as few issue slots as the hazards allow: the level-1 adds of all five
// groups go into two asm blocks and the level-2 + quad adds into one, ordered so that no DPP reads a register written
// less than three instructions earlier; only the block heads need an s_nop (the compiler's hazard recogniser does not look
// inside asm and the preceding instruction may have written an input).  40 DPP adds + 3 s_nop instead of 40 + 11.
#define GSR_L1(CTRL, d, a, b) GSR_DPPB(CTRL, "0x3", d, a) GSR_DPPB(CTRL, "0xc", d, b)
__device__ __forceinline__ void row_reduce20(const float* v, float* z) {
	float p0, q0, p1, q1, p2, q2, p3, q3, p4, q4;
	asm volatile("s_nop 1\n\t" GSR_L1("row_mirror", 0, 6, 7) GSR_L1("row_mirror", 1, 8, 9) GSR_L1("row_mirror", 2, 10, 11)
	             GSR_L1("row_mirror", 3, 12, 13) GSR_L1("row_mirror", 4, 14, 15) GSR_L1("row_mirror", 5, 16, 17)
	             : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(p2), "=&v"(q2)
	             : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]));
	asm volatile("s_nop 1\n\t" GSR_L1("row_mirror", 0, 4, 5) GSR_L1("row_mirror", 1, 6, 7) GSR_L1("row_mirror", 2, 8, 9) GSR_L1("row_mirror", 3, 10, 11)
	             : "=&v"(p3), "=&v"(q3), "=&v"(p4), "=&v"(q4)
	             : "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]));
#define GSR_L2(d, a, b) GSR_DPPB("row_half_mirror", "0x5", d, a) GSR_DPPB("row_half_mirror", "0xa", d, b)
#define GSR_QA(CTRL) GSR_Q1(CTRL, 0) GSR_Q1(CTRL, 1) GSR_Q1(CTRL, 2) GSR_Q1(CTRL, 3) GSR_Q1(CTRL, 4)
	asm volatile("s_nop 1\n\t" GSR_L2(0, 5, 6) GSR_L2(1, 7, 8) GSR_L2(2, 9, 10) GSR_L2(3, 11, 12) GSR_L2(4, 13, 14)
	             GSR_QA("quad_perm:[2,3,0,1]") GSR_QA("quad_perm:[1,0,3,2]")
	             : "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3]), "=&v"(z[4])
	             : "v"(p0), "v"(q0), "v"(p1), "v"(q1), "v"(p2), "v"(q2), "v"(p3), "v"(q3), "v"(p4), "v"(q4));
}
// 16 values (variant G's backward): 32 DPP adds + 2 s_nop.
__device__ __forceinline__ void row_reduce16(const float* v, float* z) {
	float p0, q0, p1, q1, p2, q2, p3, q3;
	asm volatile("s_nop 1\n\t" GSR_L1("row_mirror", 0, 8, 9) GSR_L1("row_mirror", 1, 10, 11) GSR_L1("row_mirror", 2, 12, 13) GSR_L1("row_mirror", 3, 14, 15)
	             GSR_L1("row_mirror", 4, 16, 17) GSR_L1("row_mirror", 5, 18, 19) GSR_L1("row_mirror", 6, 20, 21) GSR_L1("row_mirror", 7, 22, 23)
	             : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(p2), "=&v"(q2), "=&v"(p3), "=&v"(q3)
	             : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
	               "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
#define GSR_QB(CTRL) GSR_Q1(CTRL, 0) GSR_Q1(CTRL, 1) GSR_Q1(CTRL, 2) GSR_Q1(CTRL, 3)
	asm volatile("s_nop 1\n\t" GSR_L2(0, 4, 5) GSR_L2(1, 6, 7) GSR_L2(2, 8, 9) GSR_L2(3, 10, 11)
	             GSR_QB("quad_perm:[2,3,0,1]") GSR_QB("quad_perm:[1,0,3,2]")
	             : "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3])
	             : "v"(p0), "v"(q0), "v"(p1), "v"(q1), "v"(p2), "v"(q2), "v"(p3), "v"(q3));
}
// lane -> pixel inside the wave's 8x8 block: 16-lane row r is the 4x4 sub-block (r & 1, r >> 1), lanes inside it row-major
__device__ __forceinline__ int sub_px(int lane) { return ((lane >> 4) & 1) * 4 + (lane & 3); }
__device__ __forceinline__ int sub_py(int lane) { return (lane >> 5) * 4 + ((lane >> 2) & 3); }
// ---- wave-cooperative stores of per-Gaussian rows.  The reference's tensors are AoS ((P,3), (P,9), (P,16,3), 80-byte records ...):
// stored by the lane that computed them, a row of N dwords is N store instructions of 64 x 4 bytes at a stride of 4N bytes —
// 64 partial-line write requests each.  Here the wave's 64 rows, which are contiguous in memory, go through a wave-private
// LDS tile (N planes of 65 elements) and leave as N instructions of 64 consecutive elements.  `out` points at row 0 of the
// wave, rows beyond `nrows` are not written; ACC adds to memory instead.  All 64 lanes must be active.
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int N, bool ACC>
__device__ __forceinline__ void wave_store_rows(float* lds, const float* v, float* __restrict__ out, int nrows, int lane) {
#pragma unroll
	for (int k = 0; k < N; k++) lds[k * 65 + lane] = v[k];
	wave_lds_sync();
#pragma unroll
	for (int j = 0; j < N; j++) {
		const int e = lane + 64 * j, r = e / N, k = e - r * N;
		if (r < nrows) {
			const float x = lds[k * 65 + r];
			if (ACC) {
				if (x != 0.f) out[e] += x;      // (adding zero is a no-op: the rows of Gaussians this view did not touch are neither read nor written)
			} else {
				out[e] = x;
			}
		}
	}
	wave_lds_sync();
}
// float4 elements; the row is N consecutive float4 starting at float4 `col0` of a row of `pitch` float4
template <int N, bool ACC>
__device__ __forceinline__ void wave_store_rows4(float4* lds, const float4* v, float4* __restrict__ out, int pitch, int col0, int nrows, int lane) {
#pragma unroll
	for (int k = 0; k < N; k++) lds[k * 65 + lane] = v[k];
	wave_lds_sync();
#pragma unroll
	for (int j = 0; j < N; j++) {
		const int e = lane + 64 * j, r = e / N, k = e - r * N;
		if (r < nrows) {
			const float4 x = lds[k * 65 + r];
			float4* o = out + (size_t)r * pitch + col0 + k;
			if (ACC) {
				if (x.x != 0.f || x.y != 0.f || x.z != 0.f || x.w != 0.f) {
					const float4 y = *o;
					*o = make_float4(y.x + x.x, y.y + x.y, y.z + x.z, y.w + x.w);
				}
			} else {
				*o = x;
			}
		}
	}
	wave_lds_sync();
}
// one element of a per-Gaussian parameter gradient: written, or (ACC: several views accumulate into one gradient buffer on the device) added
template <bool ACC> __device__ __forceinline__ void put(float* p, float v) {
	if (ACC) { if (v != 0.f) *p += v; }     // (a view adds nothing to the Gaussians it did not touch: no read, no write)
	else *p = v;
}
template <bool ACC> __device__ __forceinline__ void put4(float4* p, float a, float b, float c, float d) {
	if (ACC) {
		if (a != 0.f || b != 0.f || c != 0.f || d != 0.f) {
			const float4 r = *p;
			*p = make_float4(r.x + a, r.y + b, r.z + c, r.w + d);
		}
	} else {
		*p = make_float4(a, b, c, d);
	}
}
// v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0 (which __builtin_ctz leaves undefined)
__device__ __forceinline__ uint32_t ffbl_raw(uint32_t x) {
	uint32_t r;
	asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
	return r;
}
// which of the four values of a row_reduce4 call quad q of a row ends up with
__device__ __forceinline__ int row_reduce_slot(int lane) {
	const int q = (lane >> 2) & 3;          // quads 0,1,2,3 hold a, c, b, d
	return ((q & 1) << 1) | (q >> 1);
}

// ---- lane masks.  A per-lane predicate lives in an SGPR pair; the tile kernels keep their predicates as explicit 64-bit
// masks (`lmask`: ballots of single compares combined with scalar and/or/andn2) and select with v_cndmask on that pair.
// Written out because `__ballot(a && b)` makes hipcc materialise the combined predicate in a VGPR and compare it again
// (v_cndmask 0/1 + v_cmp_ne, two ~4-cycle instructions per ballot) although the SGPR pair is already the answer when all
// 64 lanes are active, which they always are in these kernels.
typedef unsigned long long lmask;
#define LMASK(cond) __builtin_amdgcn_ballot_w64(cond)      // `cond` must be ONE compare for the fold to v_cmp -> sgpr pair
__device__ __forceinline__ float selm(lmask m, float a, float b) {   // lane bit set ? a : b
	float r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
	return r;
}
__device__ __forceinline__ float selm0(lmask m, float a) {           // lane bit set ? a : 0
	float r;
	asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
	return r;
}
__device__ __forceinline__ uint32_t selmu(lmask m, uint32_t a, uint32_t b) {
	uint32_t r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
	return r;
}

// fminf without the canonicalising v_max(x, x) hipcc puts in front of v_min_f32 when it cannot prove an operand is not a
// signalling NaN (results of fp32 arithmetic never are): one ~4-cycle instruction per use in the pair loops.
__device__ __forceinline__ float min_raw(float a, float b) {
	float r;
	asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// Row maxima of four non-negative values (one per processed pair) in 8 DPP instructions: the max-flavoured twin of
// row_reduce_groups + quad_sum.  On return every lane of quad q of a 16-lane row holds the ROW maximum of value
// {a, c, b, d}[q] (row_reduce_slot).  All 64 lanes must be active.
__device__ __forceinline__ float row_max4(float a, float b, float c, float d) {
	float p, q, z;
	asm volatile("s_nop 1\n\t"
	             "v_max_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0x3\n\t"
	             "v_max_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0xc\n\t"
	             "v_max_f32_dpp %1, %4, %4 row_mirror row_mask:0xf bank_mask:0x3\n\t"
	             "v_max_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
	             : "=&v"(p), "=&v"(q) : "v"(a), "v"(b), "v"(c), "v"(d));
	asm volatile("s_nop 1\n\t"
	             "v_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
	             "v_max_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
	             "s_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\t"
	             "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	             : "=&v"(z) : "v"(p), "v"(q));
	return z;
}

// exp(x) for x in [-88, 0]: exp2 of a two-float product x*log2(e) (hi from the multiply, lo from the FMA residual
// plus the low word of log2(e)) with a first-order correction for lo.  ~1.5 ulp, 5 instructions; the plain
// exp2(x*log2e) form loses |x| * 2^-24 relative (3e-7 at x = -5.5), which the ill-conditioned surfel backward
// amplifies ~200x, and the ocml expf() costs ~3x more.
__device__ __forceinline__ float exp_neg(float x) {
	const float hi = x * 1.44269504f;
	const float lo = fmaf(x, 1.44269504f, -hi) + x * 1.92596299e-8f;
	const float e = __builtin_amdgcn_exp2f(hi);
	return fmaf(e, 0.693147181f * lo, e);
}

// a / b with one Newton-Raphson correction on v_rcp_f32: correctly rounded except in rare last-bit cases,
// ~8 ns per wave instead of ~18 ns for the IEEE v_div_scale/v_div_fmas/v_div_fixup sequence (measured,
// tests/microbench/valu_rate.hip).  Only for operands known to be finite, normal and far from overflow
// (1 - alpha >= 0.01, |p.z| >= 1e-6, depth >= 0.2 in the tile kernels).
__device__ __forceinline__ float div_nr(float a, float b) {
	const float r = __builtin_amdgcn_rcpf(b);
	const float q = a * r;
	const float e = fmaf(-b, q, a);
	return fmaf(e, r, q);
}


// ---- per-wave culling vote.  A Gaussian's cull record (two float4, written by preprocess) describes a superset of
// the pixels it can blend into: the ellipse d^T [[a,b],[b,c]] d <= 1 about (ex, ey), united with a disc of squared
// radius r2 about (dcx, dcy).  Encodings: a <= 0 (or NaN) = unbounded (always a hit); r2 == -1 = no disc;
// r2 == -2 = can never blend (opacity < 1/255).  The vote asks whether that set meets the wave's 8x8 pixel block
// [x0,x1]x[y0,y1] (already inflated by half a pixel by the caller): the minimum of a convex quadratic over a box is
// attained at the centre if it lies inside, otherwise on one of the four edges.  The vote runs once per (wave, list
// entry) with one entry per lane, so its ~40 instructions cost less than one per entry.
#ifndef GSR_CULL_RCP
#define GSR_CULL_RCP 1     // 1-ulp v_rcp instead of two IEEE divisions in the footprint vote: tile forward 0.503 -> 0.496 ms at C3
#endif
__device__ __forceinline__ bool cull_hit(const float4 c0, const float4 c1, float x0, float x1, float y0, float y1) {
	if (c1.w == -2.0f) return false;
	const float a = c0.z, b = c0.w, c = c1.x;
	if (!(a > 0.0f)) return true;
	const float ex = c0.x, ey = c0.y;
	bool hit = ex >= x0 && ex <= x1 && ey >= y0 && ey <= y1;
	if (!hit) {
		float best = __int_as_float(0x7f800000);
#if GSR_CULL_RCP
		const float boc = b * __builtin_amdgcn_rcpf(c), boa = b * __builtin_amdgcn_rcpf(a);   // 1-ulp reciprocals: the point they place on an edge
#else                                                                                         // moves by 1e-7 of its offset, the form there by its square
		const float boc = b / c, boa = b / a;
#endif
#pragma unroll
		for (int e = 0; e < 2; e++) {
			const float dx = (e ? x1 : x0) - ex;
			const float dy = fminf(fmaxf(-boc * dx, y0 - ey), y1 - ey);
			best = fminf(best, a * dx * dx + 2.0f * b * dx * dy + c * dy * dy);
			const float dy2 = (e ? y1 : y0) - ey;
			const float dx2 = fminf(fmaxf(-boa * dy2, x0 - ex), x1 - ex);
			best = fminf(best, a * dx2 * dx2 + 2.0f * b * dx2 * dy2 + c * dy2 * dy2);
		}
		hit = !(best > 1.0f);   // NaN -> hit
	}
	if (!hit && c1.w >= 0.0f) {
		const float dx = fminf(fmaxf(c1.y, x0), x1) - c1.y;
		const float dy = fminf(fmaxf(c1.z, y0), y1) - c1.z;
		hit = !(dx * dx + dy * dy > c1.w);
	}
	return hit;
}

// Dispatch order of the tile kernels' units (unit = one 8x8 quadrant of a tile = one wave).
//  * Longest first: tiles are sorted by descending list length (tile_order) and slots are dealt in that order, so the
//    waves still running at the end of the kernel are the short ones (classic LPT list scheduling).  A simulation with
//    the C3 list lengths gives a makespan of 1.02x the ideal against 1.07-1.26x for image order and 1.96x for the
//    first version of this code, which gave each XCD one contiguous band of the image: the bands at the top and bottom of
//    the view hold far fewer instances and their XCDs idled half of the time (measured: render fwd 1.15 -> 0.73 ms,
//    bwd 2.24 -> 1.52 ms when the bands went away).
//  * XCD-aware: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs; the four quadrants of a tile
//    gather the same records, so slots are dealt in chunks of 4: one tile = one XCD (its own L2).
// Pure performance; any bijective mapping is correct.  The caller launches xcd_grid(nunits) workgroups and drops slots
// >= nunits.
#define XCD_CHUNK 4
__device__ __forceinline__ uint32_t xcd_slot(uint32_t bid) {
	const uint32_t xcd = bid & 7u, j = bid >> 3;
	return ((j / XCD_CHUNK) * 8u + xcd) * XCD_CHUNK + (j % XCD_CHUNK);
}
__host__ __device__ __forceinline__ uint32_t xcd_grid(uint32_t nunits) {
	const uint32_t g = 8u * XCD_CHUNK;
	return (nunits + g - 1) / g * g;
}

}  // namespace gsr
#endif
