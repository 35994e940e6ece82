// Shared host plumbing + the binning pipeline (tile/depth key emission, rocPRIM radix sort,
// tile ranges) used by both rasterizer variants.  Integer work; results are bit-exact with the
// reference's CUB pipeline (DSR/DGR rasterizer_impl.cu:70-138, 282-325): any stable LSD radix sort
// of the same keys gives the same permutation.
#include "gsr_internal.hpp"
#include <cstdarg>
#include <string>
#include <initializer_list>
#include <vector>
#include <mutex>
#include <rocprim/device/device_radix_sort.hpp>
#include "gsr_sort.hpp"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace gsr {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}


// ---------------------------------------------------------------- per-stage timing
struct ProfRec {
	int stage;
	hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec*> g_prof_recs;
static std::vector<ProfRec*> g_prof_free;
static std::mutex g_prof_mu;

StageTimer::StageTimer(int stage_, hipStream_t stream_) : stage(stage_), stream(stream_), rec(nullptr) {
	if (!g_prof_on) return;
	std::lock_guard<std::mutex> lk(g_prof_mu);
	ProfRec* r = nullptr;
	if (!g_prof_free.empty()) {
		r = g_prof_free.back();
		g_prof_free.pop_back();
	} else {
		r = new ProfRec();
		if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) { delete r; return; }
	}
	r->stage = stage;
	(void)hipEventRecord(r->e0, stream);
	rec = r;
}
StageTimer::~StageTimer() {
	if (!rec) return;
	ProfRec* r = (ProfRec*)rec;
	(void)hipEventRecord(r->e1, stream);
	std::lock_guard<std::mutex> lk(g_prof_mu);
	g_prof_recs.push_back(r);
}

static int g_opt_cull = 1;
int option_cull() { return g_opt_cull; }
static int g_opt_dev = 0;
int option_dev() { return g_opt_dev; }
static int g_opt_emit_items = 0;   // 0: by the Gaussian count (EMIT_ITEMS2_FROM); 1 / 2: forced (tests)
static int g_opt_mailbox = 1;
static int option_mailbox() { return g_opt_mailbox; }
static int g_opt_sort_driver = GSR_ONESWEEP_DRIVER;
int option_sort_driver() { return GSR_ONESWEEP_DRIVER && g_opt_sort_driver; }

// getHigherMsb (DSR/DGR rasterizer_impl.cu:35-50)
uint32_t higher_msb(uint32_t n) {
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

// rocPRIM switches radix_sort_pairs to a merge sort below 1 Mi items (~22 small launches, 165 us for 1e6 pairs on MI355X);
// Onesweep (histogram + one launch per 8 key bits) is ~3x faster there, so the switch point is lowered to 64 Ki items.
// (11-bit digits — three passes over the 31 depth bits instead of four; they fit LDS only with the `match` ranking —
// were measured too: 0.43 ms against 0.27 ms for the whole two-level sort.)
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;
// Both sorts go through gsr_sort.hpp (rocPRIM's Onesweep device code, one clear per sort instead of three dispatches per
// pass).  Workgroup shapes: the depth sort of P keys is launch-latency-bound (a pass over 1 M pairs moves 16 MB) and
// rocPRIM's tuned shape for 4-byte pairs (1024 threads x 16 items: 61 workgroups at P = 1 M) leaves most CUs idle;
// 1024 x 4 measured best (whole two-level sort at P = 1 M, ms: tuned 0.274, 256x8 0.270, 512x4 0.251, 1024x2 0.250,
// 1024x3 0.233, 1024x4 0.220, 1024x6 0.251, 1024x8 0.235; it also wins at 0.3 M, 2 M and 5 M).  The second-level sort of
// R tile ids keeps the tuned 1024 x 16 (1024x4 0.240, 1024x8 0.226, 512x8 0.258 against 0.220).
#ifndef DEPTH_SORT_BS
#define DEPTH_SORT_BS 1024
#endif
#ifndef DEPTH_SORT_IPT
#define DEPTH_SORT_IPT 4
#endif
#define DEPTH_SORT_SHAPE DEPTH_SORT_BS, DEPTH_SORT_IPT, 8
#define DEPTH_KEY_BITS 31u             // depths are positive floats (culled Gaussians carry FLT_MAX): their bit patterns order like the values
#define DEPTH_KEY_PLACES 4u            // 8-bit digits
// (Round 3, measured and dropped: keys = float bits minus the bits of the near plane have 27 significant bits for every depth below 13 107,
// i.e. three 9-bit places instead of four 8-bit ones, with a host-side fallback to the 31-bit sort for scenes that reach beyond.  A 9-bit
// pass over 1 M pairs takes 29 us against 17.7 us for an 8-bit one — the ranking works bit by bit and its LDS counters double — so three of
// them cost more than four: depth passes 86 vs 71 us at C3, the same at C5.)
#ifndef TILE_SORT_IPT
#define TILE_SORT_IPT 16
#endif
#ifndef TILE_SORT_BITS
#define TILE_SORT_BITS 8
#endif
#ifndef TILE_SORT_BS
#define TILE_SORT_BS 1024
#endif
#define TILE_SORT_SHAPE TILE_SORT_BS, TILE_SORT_IPT, TILE_SORT_BITS
// From this many instances on the tile sort uses half as many items per thread (twice as many workgroups): measured at C5 (15.2 M instances),
// whole sort stage 0.501 -> 0.485-0.495 ms; at C3 (3.95 M) the large shape wins by 12-30 us (profiles/r04_ab_tile_sort_shapes_emit_block.txt).
#ifndef TILE_SORT_SMALL_FROM
#define TILE_SORT_SMALL_FROM 9000000
#endif
#define TILE_SORT_SHAPE_SMALL TILE_SORT_BS, (TILE_SORT_IPT / 2), TILE_SORT_BITS
// ... and, where the tile ids fit 14 bits (up to 16 384 tiles: still two passes), 7-bit digits: sort stage at C5 0.475-0.487 -> 0.444-0.471 ms
// (three interleaved runs, profiles/r04_ab_tile_sort_7bit_digits_c5.txt); at C3 sizes 7 against 8 bits was +-2 % either way (round 3).
#define TILE_SORT_SHAPE_SMALL7 TILE_SORT_BS, (TILE_SORT_IPT / 2), 7
static bool tile_sort_small(size_t R) { return R >= (size_t)TILE_SORT_SMALL_FROM; }
static bool tile_sort_7bit(size_t R, int end_bit) { return tile_sort_small(R) && end_bit > 7 && end_bit <= 14; }
static const size_t SORT_MAX_ITEMS = ((size_t)1 << 30) - 1;   // gsr_sort.hpp handles one rocPRIM batch; beyond it rocPRIM itself

// What the depth sort carries as its VALUE (round 4): the Gaussian's index (low word) and its tile rectangle packed to 4 x 8 bits
// (high word: x0 | y0 << 8 | x1 << 16 | y1 << 24).  Key emission then reads index and rectangle of the Gaussians in depth order as ONE
// coalesced 8-byte stream; before, it read the 4-byte index coalesced and gathered the 8-byte rectangle through it — a whole line per
// Gaussian for 8 bytes: 135 MB of traffic against 44 MB of algorithmic bytes at C3, 763 against 180 MB at C5, where the kernel was bound by
// exactly that.  The first pass reads the rectangles in INDEX order (coalesced, through this iterator); the three further passes move 8
// instead of 4 bytes of value per Gaussian.  Grids beyond 255 tiles per axis (images beyond 4080 pixels) do not fit 8 bits per
// coordinate: `packed` = 0 leaves the high word empty and emit_tiles_kernel<true> gathers the rectangle as before.
struct OrderRect {
	const uint32_t* rect;
	int packed;
	__host__ __device__ unsigned long long operator()(uint32_t i) const {
		uint32_t hi = 0u;
		if (packed) {
			const uint2 r = reinterpret_cast<const uint2*>(rect)[i];
			hi = (r.x & 0xFFu) | ((r.x >> 16) << 8) | ((r.y & 0xFFu) << 16) | ((r.y >> 16) << 24);
		}
		return ((unsigned long long)hi << 32) | (unsigned long long)i;
	}
};
using OrderIn = rocprim::transform_iterator<rocprim::counting_iterator<uint32_t>, OrderRect, unsigned long long>;
static OrderIn order_in(const uint32_t* rect, int packed) { return OrderIn(rocprim::counting_iterator<uint32_t>(0), OrderRect{rect, packed}); }

// tiles_touched read through the depth order: element i of the sequence the second scan runs over
struct TouchedInOrder {
	const uint32_t* tiles_touched;
	__host__ __device__ uint32_t operator()(uint32_t idx) const { return tiles_touched[idx]; }
};
// bytes of the P-sized temp region: the two scans share the first part, the depth pre-sort has the second one to itself
// (its look-back state is cleared by the preprocess kernel, before the first scan runs)
static size_t scan_part_bytes(size_t P) {
	size_t a = 0, b = 0;
	(void)rocprim::inclusive_scan(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, P, rocprim::plus<uint32_t>(), 0, false);
	auto it = rocprim::make_transform_iterator((const uint32_t*)nullptr, TouchedInOrder{nullptr});
	(void)rocprim::inclusive_scan(nullptr, b, it, (uint32_t*)nullptr, P, rocprim::plus<uint32_t>(), 0, false);
	return (std::max(a, b) + 255) & ~(size_t)255;
}
// temp bytes: the larger of the two drivers' needs, so that the runtime switch (option_sort_driver) never changes a workspace size
static size_t depth_sort_bytes(size_t P) {
	size_t c = 0, d = 0;
	if (P <= SORT_MAX_ITEMS)
		(void)onesweep_sort_pairs<DEPTH_SORT_SHAPE>(nullptr, c, (const uint32_t*)nullptr, (uint32_t*)nullptr, order_in(nullptr, 0),
		                                            (unsigned long long*)nullptr, P, 0u, DEPTH_KEY_BITS, 0);
	(void)rocprim::radix_sort_pairs<SortConfig>(nullptr, d, (const uint32_t*)nullptr, (uint32_t*)nullptr, order_in(nullptr, 0),
	                                            (unsigned long long*)nullptr, P, 0, 31, 0, false);
	return std::max(c, d);
}
size_t scan_temp_bytes(size_t P) { return scan_part_bytes(P) + depth_sort_bytes(P); }
size_t sort_temp_bytes(size_t R, int end_bit) {
	size_t bytes = 0, pub = 0;
	if (R <= SORT_MAX_ITEMS) {
		if (tile_sort_7bit(R, end_bit))
			(void)onesweep_sort_pairs<TILE_SORT_SHAPE_SMALL7>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
			                                                  R, 0u, (unsigned)end_bit, 0);
		else if (tile_sort_small(R))
			(void)onesweep_sort_pairs<TILE_SORT_SHAPE_SMALL>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
			                                                 R, 0u, (unsigned)end_bit, 0);
		else
			(void)onesweep_sort_pairs<TILE_SORT_SHAPE>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, R,
			                                           0u, (unsigned)end_bit, 0);
	}
	(void)rocprim::radix_sort_pairs<SortConfig>(nullptr, pub, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, R, 0,
	                                            end_bit, 0, false);
	return std::max(bytes, pub);
}

GeomState carve_geom(void* buf, size_t P, int rec_f4, int aux_floats, int acc_floats, size_t scan_bytes, size_t* total) {
	Carver c(buf);
	GeomState g;
	g.depths = c.take<float>(P);
	g.rect = c.take<uint32_t>(2 * P);
	g.tiles_touched = c.take<uint32_t>(P);
	g.point_offsets = c.take<uint32_t>(P);
	g.clamped = c.take<uint8_t>(P);
	g.rec = c.take<float4>(P * rec_f4);
	g.bbox = c.take<float4>(2 * P);
	g.aux = c.take<float>(P * aux_floats);
	g.acc = c.take<float>(P * acc_floats);
	g.flags = c.take<int>(4);
	g.depth_sorted = c.take<uint32_t>(P);
	g.order = c.take<unsigned long long>(P);
	g.emit_state_bytes = ((((P + 255) / 256 + 1) * sizeof(unsigned long long)) + 15) & ~(size_t)15;   // (enough for any EMIT_BLOCK >= 256)
	g.emit_state = c.take<unsigned long long>(g.emit_state_bytes / sizeof(unsigned long long));
	g.scan_temp = c.take<char>(scan_bytes);
	g.scan_temp_bytes = scan_bytes;
	g.depth_sort_temp = g.scan_temp ? static_cast<char*>(g.scan_temp) + scan_part_bytes(P) : nullptr;
	g.depth_sort_bytes = depth_sort_bytes(P);
	g.depth_sort_clear = P <= SORT_MAX_ITEMS ? onesweep_cleared_bytes<DEPTH_SORT_SHAPE>(P, 0u, DEPTH_KEY_BITS) : 0;
	if (total) *total = c.size();
	return g;
}
ImageState carve_image(void* buf, size_t HW, size_t tiles, int planes_T, int planes_n, size_t* total) {
	Carver c(buf);
	ImageState s;
	s.ranges = c.take<uint2>(tiles);
	s.final_T = c.take<float>(HW * planes_T);
	s.n_contrib = c.take<uint32_t>(HW * planes_n);
	s.tile_order = c.take<uint32_t>(tiles);
	if (total) *total = c.size();
	return s;
}
BinningState carve_binning(void* buf, size_t R, size_t tiles, size_t sort_bytes, size_t* total) {
	Carver c(buf);
	BinningState b;
	b.point_list = c.take<uint32_t>(R);
	b.mask_stride = R / 64 + tiles + 1;
	b.blend_mask = c.take<unsigned long long>(16 * b.mask_stride);
	b.tile_keys = c.take<uint32_t>(R);
	b.tile_keys_unsorted = c.take<uint32_t>(R);
	b.vals_unsorted = c.take<uint32_t>(R);
	b.sort_temp = c.take<char>(sort_bytes);
	b.sort_temp_bytes = sort_bytes;
	if (total) *total = c.size();
	return b;
}

// duplicateWithKeys + SortPairs (DSR/DGR rasterizer_impl.cu:70-111, 308-313), restructured as a TWO-LEVEL sort.
// The reference sorts R (tile << 32 | depth bits) 64-bit keys: 6 radix passes over R pairs (152 B per instance).  The
// same order — tile, then depth bits, then Gaussian index (stable) — is obtained by
//   1. a stable sort of the P Gaussians by depth bits (4 passes over P pairs, P << R),
//   2. emitting the (tile, index) instances in THAT order, and
//   3. a stable sort of the R instances by tile id alone (2 passes for <= 16 tile bits),
// since a stable sort by tile keeps instances of one tile in emission (= depth, index) order.  point_list is bit-identical
// to the reference's; the 64-bit keys exist only as a debug reconstruction (gsr_debug_fetch "keys").
// Emission order inside a Gaussian: y outer / x inner, as the reference.
// (A wave-cooperative version — the 64 Gaussians of a wave own one contiguous output run; lanes take instances begin + l,
// + 64, ..., find the owner by bisection over the start offsets in LDS and store 256 contiguous bytes per instruction —
// was measured slower: 75 us against 57 us at R = 3.9 M.)
// Per-Gaussian statistics between preprocess and the sorts, ONE dispatch (round 3; it replaces rocPRIM's 64-bit reduce of tiles_touched
// and the histogram dispatch of the depth sort):
//   * num_rendered = sum of tiles_touched in 64 bits (the reference's 32-bit InclusiveSum, rasterizer_impl.cu:282, wraps silently),
//   * the counts of the four 8-bit digits of the depth keys (per-workgroup LDS histogram, non-zero bins flushed with global atomics: what
//     rocPRIM's onesweep_histograms does), laid out as the sort's pass kernels expect them.
// flags[2,3] were zeroed by the preprocess kernel, depth_counts by its clear of the depth sort's temp region.
// (The same fold was tried for the tile-id sort — emit_tiles_kernel counting the digits of the ids it writes in an LDS histogram — and
// lost: the high digit of a tile id has 32 values, the 64 lanes of a wave hit the same few bins and LDS atomics serialise; key emission went
// from 0.044 to 0.114 ms at C3 and from 0.12 to 0.38 ms at C5 to save an 8-us dispatch.  The depth keys' low digits are spread over 256 bins.)
#define STATS_BLOCK 1024
__global__ void __launch_bounds__(STATS_BLOCK) gaussian_stats_kernel(int P, const float* __restrict__ depths, const uint32_t* __restrict__ tiles_touched,
                                                                      uint32_t* __restrict__ depth_counts, int* __restrict__ flags, int* __restrict__ mailbox,
                                                                      uint32_t seq) {
	__shared__ uint32_t s_hist[DEPTH_KEY_PLACES * 256u];
	__shared__ unsigned long long s_sum[STATS_BLOCK / 64];
	const uint32_t t = threadIdx.x;
	s_hist[t] = 0u;                       // (STATS_BLOCK == DEPTH_KEY_PLACES * 256)
	__syncthreads();
	unsigned long long sum = 0ull;
	for (size_t i = (size_t)blockIdx.x * STATS_BLOCK + t; i < (size_t)P; i += (size_t)gridDim.x * STATS_BLOCK) {
		sum += tiles_touched[i];
		if (depth_counts != nullptr) {
			const uint32_t key = __float_as_uint(depths[i]);
			atomicAdd(&s_hist[key & 255u], 1u);
			atomicAdd(&s_hist[256u + ((key >> 8) & 255u)], 1u);
			atomicAdd(&s_hist[512u + ((key >> 16) & 255u)], 1u);
			// top digit (bit 31 is outside the sorted range): the sign-less exponent's high bits — nearly the same value in every lane, and LDS
			// atomics to one address serialise: one add per DISTINCT value of the wave instead
			const uint32_t top = (key >> 24) & 127u;
			unsigned long long todo = __ballot(1);
			while (todo != 0ull) {
				const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)top, __ffsll((long long)todo) - 1);
				const unsigned long long same = __ballot(top == d0) & todo;
				if ((t & 63u) == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&s_hist[768u + d0], (uint32_t)__popcll(same));
				todo &= ~same;
			}
		}
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
	if ((t & 63u) == 0u) s_sum[t >> 6] = sum;
	__syncthreads();
	if (t == 0u) {
		for (uint32_t w = 1; w < STATS_BLOCK / 64; w++) sum += s_sum[w];
		if (sum != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(flags + 2), sum);
		// mailbox (pinned, host-coherent): the LAST workgroup to get here hands num_rendered to the waiting host thread itself — no copy
		// packet and no barrier between this kernel and the depth sort behind it (4 us of copy + 6 us of gap on the forward's critical path)
		if (mailbox != nullptr) {
			const uint32_t ticket = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(flags + 1), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			if (ticket == gridDim.x - 1u) {
				const unsigned long long total = __hip_atomic_load(reinterpret_cast<unsigned long long*>(flags + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(mailbox + 2, (int)(uint32_t)(total & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(mailbox + 3, (int)(uint32_t)(total >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(reinterpret_cast<uint32_t*>(mailbox + 4), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
	if (depth_counts != nullptr) {
		const uint32_t h = s_hist[t];
		if (h != 0u) atomicAdd(&depth_counts[t], h);
	}
}
static_assert(STATS_BLOCK == DEPTH_KEY_PLACES * 256u, "one histogram bin per thread");

#ifndef EMIT_BLOCK
#define EMIT_BLOCK 1024    // fewer, larger workgroups shorten the look-back chains of the scan (emit at C3 / C5: 256 -> 0.052 / 0.205 ms, 512 -> 0.048 / 0.179, 1024 -> 0.046 / 0.169)
#endif
#define EMIT_BIG 8u   // a Gaussian with more instances than this is emitted by its whole wave (C3: 32 -> 0.046 ms, 16 -> 0.045, 8 -> 0.042)
// State word of the scan inside emit_tiles_kernel: flag in bits 62-63 (0 nothing yet, 1 = this workgroup's own instance count, 2 = inclusive
// prefix up to and including this workgroup), value in the low 32 bits.  One 64-bit relaxed agent-scope atomic carries flag and value
// together, so no fence is needed (an agent-scope release fence writes back the XCD's L2 on this part: gsr_sort.hpp).
__device__ __forceinline__ unsigned long long emit_pack(uint32_t flag, uint32_t v) { return ((unsigned long long)flag << 62) | (unsigned long long)v; }
__device__ __forceinline__ void emit_publish(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long emit_peek(unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// duplicateWithKeys (rasterizer_impl.cu:70-111) for the Gaussians in depth order.  Round 3: the exclusive prefix sum of the instance
// counts in that order — where each Gaussian's run starts — is computed HERE, as a single-pass chained scan (per-workgroup scan +
// decoupled look-back over the workgroups in front, wave-parallel), instead of a rocPRIM inclusive_scan in front of this kernel: that
// scan read tiles_touched through the depth order (a random 4-byte gather per Gaussian) and cost two dispatches, 19 us at C3 and 97 us
// at C5; the count is the area of the tile rectangle, which this kernel gathers anyway (culled Gaussians carry an empty rectangle).
// EMIT_ITEMS: Gaussians per thread (consecutive positions of the depth order).  Two halve the number of workgroups and of look-back hops but
// double what a workgroup does between its ticket and its last store: emission 0.041 -> 0.060 ms at C3 (977 workgroups: two rounds over the
// chip become one, twice as long) and 0.116 -> 0.101 ms at C5 (4883 workgroups: ten rounds become five) — chosen by the Gaussian count.
#ifndef EMIT_ITEMS2_FROM
#define EMIT_ITEMS2_FROM 3000000
#endif
template <bool GATHER, int EMIT_ITEMS>
__global__ void __launch_bounds__(EMIT_BLOCK) emit_tiles_kernel(int P, const unsigned long long* __restrict__ order, const uint32_t* __restrict__ rect,
                                                         unsigned long long* __restrict__ scan_state, uint32_t* __restrict__ ticket,
                                                         uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals, uint32_t tiles_x,
                                                         uint2* __restrict__ ranges, uint32_t tiles, void* sort_clear, size_t sort_clear_bytes,
                                                         unsigned long long* __restrict__ blend_mask, size_t blend_words) {
	__shared__ uint32_t s_wsum[EMIT_BLOCK / 64];
	__shared__ uint32_t s_base;
	__shared__ uint32_t s_bid;
	// Position of this workgroup in the chained scan below: an atomic TICKET, not blockIdx.x.  The look-back spins until every workgroup
	// in front has published; that terminates only if those workgroups have started, which the order of the tickets guarantees (a
	// workgroup holding ticket t exists, hence so do the holders of 0..t-1) and the order of the block ids does not — HIP makes no
	// promise about dispatch order, and rocPRIM's own Onesweep passes take an atomic block id on gfx950 for the same reason
	// (gsr_sort.hpp).  One atomic per workgroup on a word the preprocess kernel has cleared.
	if (threadIdx.x == 0) s_bid = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__syncthreads();
	const int bid = (int)s_bid;
	const size_t gtid = (size_t)bid * EMIT_BLOCK + threadIdx.x;       // for the grid-stride clears below
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	// index and rectangle of the Gaussians at positions i0 .. i0 + EMIT_ITEMS - 1 of the depth order: one coalesced load of 8 bytes each
	// (OrderRect); the area of a rectangle is the number of instances (tiles_touched)
	const int i0 = (bid * EMIT_BLOCK + (int)threadIdx.x) * EMIT_ITEMS;
	uint32_t idx[EMIT_ITEMS], x0[EMIT_ITEMS], y0[EMIT_ITEMS], x1[EMIT_ITEMS], y1[EMIT_ITEMS], cnt[EMIT_ITEMS];
	unsigned long long ov[EMIT_ITEMS];
	if (EMIT_ITEMS == 2 && i0 + 1 < P) {       // (order is 16-byte aligned: carve_geom; i0 is even)
		const ulonglong2 o2 = *reinterpret_cast<const ulonglong2*>(order + i0);
		ov[0] = o2.x; ov[EMIT_ITEMS - 1] = o2.y;
	} else {
#pragma unroll
		for (int e = 0; e < EMIT_ITEMS; e++) ov[e] = i0 + e < P ? order[i0 + e] : 0ull;
	}
	uint32_t mine = 0u;
#pragma unroll
	for (int e = 0; e < EMIT_ITEMS; e++) {
		idx[e] = (uint32_t)ov[e];
		x0[e] = y0[e] = x1[e] = y1[e] = 0u;
		if (i0 + e < P) {
			if (GATHER) {
				const uint2 r = reinterpret_cast<const uint2*>(rect)[idx[e]];
				x0[e] = r.x & 0xFFFFu; y0[e] = r.x >> 16; x1[e] = r.y & 0xFFFFu; y1[e] = r.y >> 16;
			} else {
				const uint32_t hi = (uint32_t)(ov[e] >> 32);
				x0[e] = hi & 0xFFu; y0[e] = (hi >> 8) & 0xFFu; x1[e] = (hi >> 16) & 0xFFu; y1[e] = hi >> 24;
			}
		}
		cnt[e] = (x1[e] - x0[e]) * (y1[e] - y0[e]);
		mine += cnt[e];
	}
	// (the forward tile kernel only writes the blend masks of the batches it reaches: the rest must read as "nothing blended")
	for (size_t t = gtid; t < blend_words; t += (size_t)gridDim.x * EMIT_BLOCK) blend_mask[t] = 0ull;
	// the tile ranges (tile_ranges_kernel fills the non-empty ones after the sort) and the look-back state of the tile-id
	// sort that follows are cleared here: a dispatch of its own costs ~5 us whatever it does
	for (uint32_t t = (uint32_t)gtid; t < tiles; t += gridDim.x * EMIT_BLOCK) ranges[t] = make_uint2(0u, 0u);
	sort_clear_region(sort_clear, sort_clear_bytes, gtid, (size_t)gridDim.x * EMIT_BLOCK);
	// ---- scan of the counts: inside the wave, across the waves, across the workgroups in front
	uint32_t incl = mine;
#pragma unroll
	for (uint32_t o = 1; o < 64; o <<= 1) {
		const uint32_t v = __shfl_up(incl, o, 64);
		if (lane >= o) incl += v;
	}
	if (lane == 63u) s_wsum[wave] = incl;
	__syncthreads();
	if (wave == 0u) {
		uint32_t total = 0u;
		for (uint32_t w = 0; w < EMIT_BLOCK / 64; w++) total += s_wsum[w];
		uint32_t prefix = 0u;
		if (bid == 0) {
			if (lane == 0u) emit_publish(scan_state, emit_pack(2u, total));
		} else {
			if (lane == 0u) emit_publish(scan_state + bid, emit_pack(1u, total));
			int j = bid - 1;                         // lane l examines workgroup j - l
			for (;;) {
				const int tgt = j - (int)lane;
				const unsigned long long st = tgt >= 0 ? emit_peek(scan_state + tgt) : emit_pack(2u, 0u);   // in front of workgroup 0: prefix 0
				const uint32_t flag = (uint32_t)(st >> 62);
				const unsigned long long done = __ballot(flag == 2u), empty = __ballot(flag == 0u);
				// the nearest workgroup that already knows its inclusive prefix ends the walk; everything nearer must at least have published its count
				const int stop = done != 0ull ? __ffsll((long long)done) - 1 : 63;
				const unsigned long long need = stop == 63 ? ~0ull : ((2ull << stop) - 1ull);
				if ((empty & need) != 0ull) continue;      // somebody nearer has not published yet: look again (it holds a smaller ticket, so it is running)
				uint32_t v = (int)lane <= stop ? (uint32_t)st : 0u;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
				prefix += v;
				if (done != 0ull) break;
				j -= 64;
			}
			if (lane == 0u) emit_publish(scan_state + bid, emit_pack(2u, prefix + total));
		}
		if (lane == 0u) s_base = prefix;
	}
	__syncthreads();
	uint32_t wbase = s_base;
	for (uint32_t w = 0; w < wave; w++) wbase += s_wsum[w];
	uint32_t off = wbase + incl - mine;
#pragma unroll
	for (int e = 0; e < EMIT_ITEMS; e++) {
		// small Gaussians (the mean is 4 instances): one thread writes its own short run
		if (cnt[e] != 0u && cnt[e] <= EMIT_BIG) {
			uint32_t o = off;
			for (uint32_t y = y0[e]; y < y1[e]; y++)
				for (uint32_t x = x0[e]; x < x1[e]; x++) {
					tile_keys[o] = y * tiles_x + x;
					vals[o] = idx[e];
					o++;
				}
		}
		// large ones (a near Gaussian covers up to every tile of the image; one thread looping over thousands of instances
		// was the whole tail of this kernel) are emitted by all 64 lanes of the wave, 256 contiguous bytes per store
		unsigned long long big = __ballot(cnt[e] > EMIT_BIG);
		while (big) {
			const int src = __ffsll((long long)big) - 1;
			big &= big - 1;
			const uint32_t b_off = (uint32_t)__builtin_amdgcn_readlane((int)off, src), b_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt[e], src);
			const uint32_t b_idx = (uint32_t)__builtin_amdgcn_readlane((int)idx[e], src);
			const uint32_t b_x0 = (uint32_t)__builtin_amdgcn_readlane((int)x0[e], src), b_y0 = (uint32_t)__builtin_amdgcn_readlane((int)y0[e], src);
			const uint32_t b_w = (uint32_t)__builtin_amdgcn_readlane((int)(x1[e] - x0[e]), src);
			for (uint32_t q = lane; q < b_cnt; q += 64u) {
				const uint32_t dy = q / b_w, dx = q - dy * b_w;   // emission order inside a Gaussian: y outer / x inner
				tile_keys[b_off + q] = (b_y0 + dy) * tiles_x + (b_x0 + dx);
				vals[b_off + q] = b_idx;
			}
		}
		off += cnt[e];
	}
}

// identifyTileRanges (DSR/DGR rasterizer_impl.cu:116-138)
__global__ void __launch_bounds__(256) tile_ranges_kernel(int L, const uint32_t* __restrict__ tile_keys, uint2* __restrict__ ranges) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= L) return;
	const uint32_t currtile = tile_keys[idx];
	if (idx == 0) ranges[currtile].x = 0;
	else {
		const uint32_t prevtile = tile_keys[idx - 1];
		if (currtile != prevtile) {
			ranges[prevtile].y = idx;
			ranges[currtile].x = idx;
		}
	}
	if (idx == L - 1) ranges[currtile].y = L;
}

// Longest-first dispatch order of the tiles in ONE single-workgroup kernel: a counting sort on the list length quantised
// to 1024 buckets (list scheduling only needs an approximate order; a full rocPRIM sort of ~8 k keys costs 30 us in
// launches).  Order inside a bucket is whatever the LDS atomics produce; any permutation is correct.
#define TILE_ORDER_REGS 16   // list lengths a thread keeps in registers (covers 16 384 tiles; beyond that they are re-read)
__global__ void __launch_bounds__(1024) tile_order_kernel(int tiles, const uint2* __restrict__ ranges, uint32_t* __restrict__ order) {
	__shared__ uint32_t hist[1024];
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_max;
	const int t = threadIdx.x;
	hist[t] = 0;
	if (t == 0) s_max = 0;
	// one pass over the ranges: the lengths stay in registers for the three phases (the first version re-read them from
	// global memory in each phase: three dependent memory round trips in a one-workgroup kernel)
	uint32_t len[TILE_ORDER_REGS];
	uint32_t m = 0;
#pragma unroll
	for (int j = 0; j < TILE_ORDER_REGS; j++) {
		const int i = t + 1024 * j;
		const uint2 r = i < tiles ? ranges[i] : make_uint2(0u, 0u);
		len[j] = r.y - r.x;
		m = max(m, len[j]);
	}
	for (int i = t + 1024 * TILE_ORDER_REGS; i < tiles; i += 1024) m = max(m, ranges[i].y - ranges[i].x);
	__syncthreads();
	atomicMax(&s_max, m);
	__syncthreads();
	int shift = 0;
	while ((s_max >> shift) >= 1024u) shift++;
#pragma unroll
	for (int j = 0; j < TILE_ORDER_REGS; j++)
		if (t + 1024 * j < tiles) atomicAdd(&hist[1023u - (len[j] >> shift)], 1u);   // bucket 0 = longest
	for (int i = t + 1024 * TILE_ORDER_REGS; i < tiles; i += 1024) atomicAdd(&hist[1023u - ((ranges[i].y - ranges[i].x) >> shift)], 1u);
	__syncthreads();
	// exclusive scan of the 1024 buckets: inside each wave with shuffles, across the 16 waves through LDS
	const uint32_t mine = hist[t];
	uint32_t incl = mine;
	const int lane = t & 63, wave = t >> 6;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t o = __shfl_up(incl, off, 64);
		if (lane >= off) incl += o;
	}
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int w = 0; w < wave; w++) base += s_wave[w];
	hist[t] = base + incl - mine;
	__syncthreads();
#pragma unroll
	for (int j = 0; j < TILE_ORDER_REGS; j++) {
		const int i = t + 1024 * j;
		if (i < tiles) order[atomicAdd(&hist[1023u - (len[j] >> shift)], 1u)] = (uint32_t)i;
	}
	for (int i = t + 1024 * TILE_ORDER_REGS; i < tiles; i += 1024) {
		const uint32_t b = 1023u - ((ranges[i].y - ranges[i].x) >> shift);
		order[atomicAdd(&hist[b], 1u)] = (uint32_t)i;
	}
}

// debug only: the reference's sorted 64-bit keys, rebuilt from the sorted tile ids and the depth of each instance
__global__ void __launch_bounds__(256) rebuild_keys_kernel(int L, const uint32_t* __restrict__ tile_keys, const uint32_t* __restrict__ point_list,
                                                           const float* __restrict__ depths, uint64_t* __restrict__ keys) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= L) return;
	keys[i] = ((uint64_t)tile_keys[i] << 32) | (uint64_t)__float_as_uint(depths[point_list[i]]);
}

// One pinned host word and one event per (host thread, device) for the num_rendered readback (the reference does a blocking
// 4-byte cudaMemcpy on the default stream, rasterizer_impl.cu:286).  Keyed by the current device: an event created on one
// device must not be recorded on another device's stream.
struct Readback {
	int* word = nullptr;       // host address of the pinned words: [0] trap flag, [2,3] num_rendered, [4] sequence number of the mailbox write
	int* dev_word = nullptr;   // the same memory as the device sees it
	uint32_t seq = 0;
	hipEvent_t done = nullptr;
};
static Readback* readback_slot() {
	static thread_local Readback slots[64];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
	Readback& r = slots[dev];
	if (!r.word) {
		if (hipHostMalloc((void**)&r.word, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
			if (hipHostGetDevicePointer((void**)&r.dev_word, r.word, 0) != hipSuccess) r.dev_word = nullptr;   // (then the copy path below is used)
		} else {
			(void)hipGetLastError();
			r.dev_word = nullptr;      // no coherent mapping on this system: plain pinned memory and the copy + event path
			if (hipHostMalloc((void**)&r.word, 64, hipHostMallocDefault) != hipSuccess) { r.word = nullptr; return nullptr; }
		}
		memset(r.word, 0, 64);
	}
	if (!r.done && hipEventCreateWithFlags(&r.done, hipEventDisableTiming) != hipSuccess) { r.done = nullptr; return nullptr; }
	return &r;
}

static int rect_packs(int tiles_x, int tiles_y) { return (tiles_x <= 255 && tiles_y <= 255) ? 1 : 0; }   // see OrderRect

int run_binning(gsr_alloc_fn alloc, void* alloc_user, int P, int tiles_x, int tiles_y, const GeomState& geom, const ImageState& img,
                BinningState* out_binning, int prefiltered, int debug, hipStream_t stream) {
	Readback* rb = readback_slot();
	if (!rb) { set_error("pinned word / event for the num_rendered readback could not be created"); return GSR_E_HIP; }
	int* host = rb->word;
	int* mailbox = nullptr;
	uint32_t seq = 0;
	const bool own_depth_sort = option_sort_driver() && (size_t)P <= SORT_MAX_ITEMS;
	{
		StageTimer st_(GSR_STAGE_SCAN, stream);
		// num_rendered = sum of tiles_touched, reduced in 64 bits (the reference's 32-bit InclusiveSum, rasterizer_impl.cu:282,
		// wraps silently at 2^32 instances; its per-Gaussian offsets are not needed here: the instances are emitted in depth
		// order from the second scan below, and gsr_debug_fetch("point_offsets") computes them on demand)
		// one dispatch: the 64-bit sum and the depth keys' digit counts (gaussian_stats_kernel)
		const unsigned blocks = (unsigned)std::min<size_t>(512, ((size_t)P + 4 * STATS_BLOCK - 1) / (4 * STATS_BLOCK));
		// the trap flag flags[0] is only ever set, and cleared, with `prefiltered`: that (debugging) mode copies the four words back instead
		mailbox = (!prefiltered && rb->dev_word != nullptr && option_mailbox()) ? rb->dev_word : nullptr;
		seq = ++rb->seq;
		gaussian_stats_kernel<<<blocks, STATS_BLOCK, 0, stream>>>(P, geom.depths, geom.tiles_touched,
		                                                          own_depth_sort ? reinterpret_cast<uint32_t*>(geom.depth_sort_temp) : nullptr, geom.flags, mailbox, seq);
		GSR_LAUNCH_CHECK(debug, stream);
		if (!mailbox) GSR_HIP_CHECK(hipMemcpyAsync(host, geom.flags, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
	}
	// the host waits on THIS point only, not on the level-1 work enqueued behind it.  (With the mailbox no event is recorded: an event record
	// is a 5-us bubble between the statistics kernel and the depth sort; the fallback of the spin below is the stream itself.)
	hipEvent_t readback_done = rb->done;
	if (!mailbox) GSR_HIP_CHECK(hipEventRecord(readback_done, stream));
	{
		// level 1 (independent of num_rendered, so it runs while the host waits for the read-back): depth order of the
		// Gaussians (31 key bits: depths are positive floats, their bit patterns order like the values); the scan of the
		// instance counts taken in that order happens inside emit_tiles_kernel
		StageTimer st_(GSR_STAGE_SORT, stream);
		size_t tmp = geom.depth_sort_bytes;
		if (own_depth_sort)   // look-back state cleared by the preprocess kernel, digit counts accumulated by gaussian_stats_kernel: four dispatches
			GSR_HIP_CHECK(onesweep_sort_pairs<DEPTH_SORT_SHAPE>(geom.depth_sort_temp, tmp, reinterpret_cast<const uint32_t*>(geom.depths), geom.depth_sorted,
			                                                   order_in(geom.rect, rect_packs(tiles_x, tiles_y)), geom.order, (size_t)P, 0u, DEPTH_KEY_BITS, stream, true,
			                                                   reinterpret_cast<const uint32_t*>(geom.depth_sort_temp)));
		else
			GSR_HIP_CHECK(rocprim::radix_sort_pairs<SortConfig>(geom.depth_sort_temp, tmp, reinterpret_cast<const uint32_t*>(geom.depths), geom.depth_sorted,
			                                                   order_in(geom.rect, rect_packs(tiles_x, tiles_y)), geom.order, (size_t)P, 0u, 31u, stream, false));
	}
	if (mailbox) {
		// spin on the sequence number the last workgroup of gaussian_stats_kernel stores; the way out, should the store not be seen while
		// the kernel runs, is the stream running dry (the depth sort behind it, ~0.07 ms later): once the kernel has completed its writes are
		// visible in any case
		volatile uint32_t* box = reinterpret_cast<volatile uint32_t*>(host);
		unsigned spins = 0;
		while (box[4] != seq) {
			if ((++spins & 63u) == 0u && hipStreamQuery(stream) != hipErrorNotReady) break;
		}
		if (box[4] != seq) {
			GSR_HIP_CHECK(hipStreamSynchronize(stream));
			if (box[4] != seq) { set_error("num_rendered mailbox was not written"); return GSR_E_HIP; }
		}
		__atomic_thread_fence(__ATOMIC_ACQUIRE);
	} else {
		GSR_HIP_CHECK(hipEventSynchronize(readback_done));
	}
	unsigned long long total64;
	memcpy(&total64, host + 2, sizeof(total64));
	if (prefiltered && host[0] != 0) { set_error("Point is filtered although prefiltered is set. This shouldn't happen!"); return GSR_E_PREFILTERED; }
	if (total64 > 0x7fffffffull) { set_error("num_rendered = %llu does not fit the int the API returns", total64); return GSR_E_INVALID; }
	const int R = (int)total64;
	const uint32_t tiles = (uint32_t)tiles_x * (uint32_t)tiles_y;
	const int bit = (int)higher_msb(tiles);
	const size_t sort_bytes = R > 0 ? sort_temp_bytes((size_t)R, bit) : 0;
	size_t total = 0;
	carve_binning(nullptr, (size_t)R, (size_t)tiles, sort_bytes, &total);
	void* buf = alloc(alloc_user, GSR_BUF_BINNING, total);
	if (!buf && total > 0) { set_error("binning buffer allocation of %zu bytes failed", total); return GSR_E_ALLOC; }
	BinningState b = carve_binning(buf, (size_t)R, (size_t)tiles, sort_bytes, nullptr);
	*out_binning = b;

	if (R == 0) GSR_HIP_CHECK(hipMemsetAsync(img.ranges, 0, (size_t)tiles * sizeof(uint2), stream));   // otherwise emit_tiles_kernel clears them
	if (R > 0) {
		{ StageTimer st_(GSR_STAGE_EMIT_KEYS, stream);
		const bool own_sort = option_sort_driver() && (size_t)R <= SORT_MAX_ITEMS;
		const size_t clear_bytes = !own_sort ? 0 : tile_sort_7bit((size_t)R, bit) ? onesweep_cleared_bytes<TILE_SORT_SHAPE_SMALL7>((size_t)R, 0u, (unsigned)bit)
		                         : tile_sort_small((size_t)R) ? onesweep_cleared_bytes<TILE_SORT_SHAPE_SMALL>((size_t)R, 0u, (unsigned)bit)
		                                                                       : onesweep_cleared_bytes<TILE_SORT_SHAPE>((size_t)R, 0u, (unsigned)bit);
		uint32_t* ticket = reinterpret_cast<uint32_t*>(geom.emit_state + (geom.emit_state_bytes / sizeof(unsigned long long) - 1));   // last state word: never a scan position
		const int items = g_opt_emit_items ? g_opt_emit_items : (P >= EMIT_ITEMS2_FROM ? 2 : 1);
		auto emit = rect_packs(tiles_x, tiles_y) ? (items == 2 ? emit_tiles_kernel<false, 2> : emit_tiles_kernel<false, 1>)
		                                         : (items == 2 ? emit_tiles_kernel<true, 2> : emit_tiles_kernel<true, 1>);
		emit<<<(P + EMIT_BLOCK * items - 1) / (EMIT_BLOCK * items), EMIT_BLOCK, 0, stream>>>(P, geom.order, geom.rect, geom.emit_state, ticket, b.tile_keys_unsorted,
		                                                       b.vals_unsorted, (uint32_t)tiles_x, img.ranges, tiles, b.sort_temp, clear_bytes, b.blend_mask,
		                                                       16 * b.mask_stride); }
		GSR_LAUNCH_CHECK(debug, stream);
		size_t sb = b.sort_temp_bytes;
		{ StageTimer st_(GSR_STAGE_SORT, stream);   // level 2: stable by tile id only
		if (option_sort_driver() && (size_t)R <= SORT_MAX_ITEMS && tile_sort_7bit((size_t)R, bit))
			GSR_HIP_CHECK(onesweep_sort_pairs<TILE_SORT_SHAPE_SMALL7>(b.sort_temp, sb, (const uint32_t*)b.tile_keys_unsorted, b.tile_keys,
			                                                         (const uint32_t*)b.vals_unsorted, b.point_list, (size_t)R, 0u, (unsigned)bit, stream, true));
		else if (option_sort_driver() && (size_t)R <= SORT_MAX_ITEMS && tile_sort_small((size_t)R))
			GSR_HIP_CHECK(onesweep_sort_pairs<TILE_SORT_SHAPE_SMALL>(b.sort_temp, sb, (const uint32_t*)b.tile_keys_unsorted, b.tile_keys,
			                                                        (const uint32_t*)b.vals_unsorted, b.point_list, (size_t)R, 0u, (unsigned)bit, stream, true));
		else if (option_sort_driver() && (size_t)R <= SORT_MAX_ITEMS)
			GSR_HIP_CHECK(onesweep_sort_pairs<TILE_SORT_SHAPE>(b.sort_temp, sb, (const uint32_t*)b.tile_keys_unsorted, b.tile_keys,
			                                                  (const uint32_t*)b.vals_unsorted, b.point_list, (size_t)R, 0u, (unsigned)bit, stream, true));
		else
			GSR_HIP_CHECK(rocprim::radix_sort_pairs<SortConfig>(b.sort_temp, sb, b.tile_keys_unsorted, b.tile_keys, b.vals_unsorted, b.point_list, (size_t)R,
			                                                   0u, (unsigned)bit, stream, false)); }
		if (debug) GSR_HIP_CHECK(hipStreamSynchronize(stream));
		{ StageTimer st_(GSR_STAGE_RANGES, stream);
		tile_ranges_kernel<<<(R + 255) / 256, 256, 0, stream>>>(R, b.tile_keys, img.ranges); }
		GSR_LAUNCH_CHECK(debug, stream);
	}
	{ StageTimer st_(GSR_STAGE_RANGES, stream);
	const int rc = run_tile_order(img, tiles, stream);
	if (rc < 0) return rc; }
	return R;
}

int run_tile_order(const ImageState& img, size_t tiles, hipStream_t stream) {
	if (tiles == 0) return 0;
	tile_order_kernel<<<1, 1024, 0, stream>>>((int)tiles, img.ranges, img.tile_order);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ pts, const float* __restrict__ vm,
                                                           uint8_t* __restrict__ present) {
#pragma clang fp contract(off)
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	const float x = pts[3 * idx], y = pts[3 * idx + 1], z = pts[3 * idx + 2];
	const float pz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14];
	present[idx] = (pz <= 0.2f) ? 0 : 1;
}

__global__ void __launch_bounds__(256) widen_clamped_kernel(int P, const uint8_t* __restrict__ packed, uint8_t* __restrict__ out) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	const uint8_t b = packed[idx];
	out[3 * idx + 0] = b & 1;
	out[3 * idx + 1] = (b >> 1) & 1;
	out[3 * idx + 2] = (b >> 2) & 1;
}
// Gather up to 9 floats per record by explicit position (debug views of records whose fields are not contiguous).
struct RecPick { int n; int pos[9]; };
__global__ void __launch_bounds__(256) pick_rec_kernel(int P, const float* __restrict__ rec, int stride, RecPick pick, float* __restrict__ out) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P * pick.n) return;
	const int idx = i / pick.n, k = i % pick.n;
	out[i] = rec[(size_t)idx * stride + pick.pos[k]];
}
// Gather `nf` floats starting at float `first` of each record (stride rec_f4*4 floats) into a dense [P, nf] array.
__global__ void __launch_bounds__(256) gather_rec_kernel(int P, const float* __restrict__ rec, int stride, int first, int nf,
                                                         float* __restrict__ out) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P * nf) return;
	const int idx = i / nf, k = i % nf;
	out[i] = rec[(size_t)idx * stride + first + k];
}

}  // namespace gsr

using namespace gsr;

extern "C" const char* gsr_last_error(void) { return g_err; }
extern "C" int gsr_version(void) { return GSR_ABI_VERSION; }


extern "C" int gsr_set_option(const char* name, int value) {
	if (std::string(name) == "cull") { g_opt_cull = value ? 1 : 0; return 0; }
	if (std::string(name) == "dev") { g_opt_dev = value; return 0; }
	if (std::string(name) == "emit_items") { g_opt_emit_items = (value == 1 || value == 2) ? value : 0; return 0; }   // Gaussians per thread of key emission; 0 = by size
	if (std::string(name) == "mailbox") { g_opt_mailbox = value ? 1 : 0; return 0; }   // 0: num_rendered comes back through a copy + event (round 2)
	if (std::string(name) == "sort_driver") { g_opt_sort_driver = value ? 1 : 0; return 0; }   // 0: public rocprim::radix_sort_pairs everywhere
	set_error("gsr_set_option: unknown option '%s'", name);
	return GSR_E_INVALID;
}
extern "C" int gsr_profile_enable(int on) {
	std::lock_guard<std::mutex> lk(g_prof_mu);
	for (ProfRec* r : g_prof_recs) g_prof_free.push_back(r);
	g_prof_recs.clear();
	g_prof_on = on != 0;
	return 0;
}
extern "C" int gsr_profile_collect(float* ms_out, int* launches_out) {
	std::lock_guard<std::mutex> lk(g_prof_mu);
	for (int i = 0; i < GSR_STAGE_COUNT; i++) { if (ms_out) ms_out[i] = 0.f; if (launches_out) launches_out[i] = 0; }
	for (ProfRec* r : g_prof_recs) {
		float ms = 0.f;
		if (hipEventSynchronize(r->e1) == hipSuccess && hipEventElapsedTime(&ms, r->e0, r->e1) == hipSuccess) {
			if (ms_out) ms_out[r->stage] += ms;
			if (launches_out) launches_out[r->stage] += 1;
		}
		g_prof_free.push_back(r);
	}
	g_prof_recs.clear();
	return 0;
}

extern "C" int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                                void* stream_) {
	(void)projmatrix;
	hipStream_t stream = (hipStream_t)stream_;
	if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("gsr_mark_visible: invalid argument"); return GSR_E_INVALID; }
	if (P == 0) return 0;
	mark_visible_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, means3D, viewmatrix, present);
	GSR_LAUNCH_CHECK(0, stream);
	return 0;
}

// Record layouts (must match gsr_gauss.hip / gsr_surfel.hip):
//   G (16 floats): xy(0,1) conic(2,3,4) opacity(5) rgb(6,7,8) normal(9,10,11) refl(12) invdepth(13) pad pad
//   S (20 floats): xy(0,1) Tu(2,3,4) Tv(5,6,7) Tw(8,9,10) normal(11,12,13) opacity(14) rgb(15,16,17) refl(18) mask(19)
extern "C" int gsr_debug_fetch(int variant, const char* name, int P, int R, int width, int height, const void* geom_buffer,
                               const void* binning_buffer, const void* image_buffer, void* dst, void* stream_) {
	hipStream_t stream = (hipStream_t)stream_;
	const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
	const size_t HW = (size_t)width * height, tiles = (size_t)tiles_x * tiles_y;
	const int rec_f4 = variant == 0 ? 5 : 4;
	const int stride = rec_f4 * 4;
	GeomState g = carve_geom((void*)geom_buffer, P, rec_f4, 0, variant == 0 ? 20 : 16, scan_temp_bytes(P), nullptr);
	ImageState im = carve_image((void*)image_buffer, HW, tiles, variant == 0 ? 3 : 1, variant == 0 ? 2 : 1, nullptr);
	BinningState b = carve_binning((void*)binning_buffer, R, tiles, 0, nullptr);
	auto d2d = [&](const void* src, size_t bytes) -> int {
		if (bytes == 0) return 0;
		GSR_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
		return 0;
	};
	auto gather = [&](int first, int nf) -> int {
		if (P == 0) return 0;
		gather_rec_kernel<<<(P * nf + 255) / 256, 256, 0, stream>>>(P, (const float*)g.rec, stride, first, nf, (float*)dst);
		GSR_LAUNCH_CHECK(0, stream);
		return 0;
	};
	std::string n(name);
	if (n == "depths") return d2d(g.depths, (size_t)P * 4);
	if (n == "means2D") return gather(0, 2);     // the first two floats of the render record of both variants (not kept as an array of its own since round 4)

	if (n == "tiles_touched") return d2d(g.tiles_touched, (size_t)P * 4);
	if (n == "cull") return d2d(g.bbox, (size_t)P * 32);   // two float4 per Gaussian (see cull_hit)
	if (n == "point_offsets") {   // not kept by the forward (it only needs the total): inclusive scan of tiles_touched on demand
		if (P == 0) return 0;
		size_t tmp = (size_t)(static_cast<char*>(g.depth_sort_temp) - static_cast<char*>(g.scan_temp));
		GSR_HIP_CHECK(rocprim::inclusive_scan(g.scan_temp, tmp, g.tiles_touched, g.point_offsets, (size_t)P, rocprim::plus<uint32_t>(), stream, false));
		return d2d(g.point_offsets, (size_t)P * 4);
	}
	if (n == "clamped") {
		if (P == 0) return 0;
		widen_clamped_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, g.clamped, (uint8_t*)dst);
		GSR_LAUNCH_CHECK(0, stream);
		return 0;
	}
	auto pick = [&](std::initializer_list<int> pos) -> int {
		if (P == 0) return 0;
		RecPick pk;
		pk.n = (int)pos.size();
		int i = 0;
		for (int v : pos) pk.pos[i++] = v;
		pick_rec_kernel<<<(P * pk.n + 255) / 256, 256, 0, stream>>>(P, (const float*)g.rec, stride, pk, (float*)dst);
		GSR_LAUNCH_CHECK(0, stream);
		return 0;
	};
	// variant S record: {x, y | Tu.x, Tv.x} {Tu.y, Tv.y | Tu.z, Tv.z} {Tw.x, Tw.y | Tw.z, opacity} {n.x, n.y | n.z, refl} {r, g | b, mask}
	if (n == "rgb") return variant == 0 ? gather(16, 3) : gather(6, 3);
	if (n == "geom4") {  // G: conic.xyz + opacity ; S: normal.xyz + opacity
		return variant == 0 ? pick({12, 13, 14, 11}) : gather(2, 4);
	}
	if (n == "transMat" && variant == 0) return pick({2, 4, 6, 3, 5, 7, 8, 9, 10});
	if (n == "cov3D" && variant == 1) { set_error("gsr_debug_fetch: the 3D covariance is not kept by the forward (round 4: the backward recomputes it)"); return GSR_E_INVALID; }
	if (n == "point_list") return d2d(b.point_list, (size_t)R * 4);
	if (n == "keys") {   // reference-format sorted keys, rebuilt (the product path sorts tile ids only)
		if (R == 0) return 0;
		rebuild_keys_kernel<<<(R + 255) / 256, 256, 0, stream>>>(R, b.tile_keys, b.point_list, g.depths, (uint64_t*)dst);
		GSR_LAUNCH_CHECK(0, stream);
		return 0;
	}
	if (n == "ranges") return d2d(im.ranges, tiles * 8);
	if (n == "final_T") return d2d(im.final_T, HW * 4 * (variant == 0 ? 3 : 1));
	if (n == "n_contrib") return d2d(im.n_contrib, HW * 4 * (variant == 0 ? 2 : 1));
	set_error("gsr_debug_fetch: unknown array '%s'", name);
	return GSR_E_INVALID;
}
