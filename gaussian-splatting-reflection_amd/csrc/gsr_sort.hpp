// Onesweep radix sort of (uint32 key, 4- or 8-byte value) pairs with ONE clear per sort.
//
// rocPRIM's radix_sort_pairs issues, per digit pass, a memset of the decoupled look-back states and a memset of the ordered
// block-id counter (gfx950 takes the atomic block-id path) in front of the pass kernel, plus one memset for the digit
// histograms: 3 dispatches per pass.  A dispatch costs ~5 us on this part whatever it does, and the sorts of this library
// are small enough (1-4 M pairs) to be bound by exactly that: the eight passes of a training step carried 17 fills.  This
// driver gives every pass its own look-back states and block-id counter inside one temp region and clears the region once;
// the device code is rocPRIM's own (rocprim::detail::onesweep_histograms / onesweep_iteration, header-only, ROCm 7.2), instantiated
// with a fixed workgroup shape instead of the architecture dispatch.  Round 3: the per-place scan of the digit histograms is done by
// every workgroup of a pass in LDS (no scan dispatch), and a caller that has the keys in its hands in an earlier kernel can supply
// the digit counts itself (no histogram dispatch): a sort is then exactly one dispatch per digit place.
// Stable, ascending, keys compared on bits [begin_bit, end_bit).
// (Folding the histogram scans into the histogram kernel's last workgroup — ticket counter + __threadfence, the classic
// "last block" pattern — was measured and is far slower: 0.30 ms against 0.185 ms for the two-level sort.  An agent-scope
// release fence writes back and invalidates the XCD's L2 on this multi-die part, once per workgroup.)
//
// What this file assumes about rocPRIM's PRIVATE device code, and how each assumption is guarded:
//   * the signatures of detail::onesweep_histograms / onesweep_iteration and of block_id_wrapper:
//     checked by the compiler; the driver is only compiled for the rocPRIM release it was written against
//     (GSR_ONESWEEP_DRIVER below), any other release takes the public rocprim::radix_sort_pairs for every sort;
//   * an all-zero onesweep_lookback_state means "empty" and is 4 bytes: static_asserts below;
//   * the temp layout is THIS driver's own (it passes every pointer explicitly), not rocPRIM's;
//   * a runtime switch (gsr_set_option("sort_driver", 0)) forces the public path, and tests/test_gpu_api_paths.py asserts that
//     both drivers produce bit-identical point lists and tile ranges at small and full size.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/rocprim_version.hpp>

#if ROCPRIM_VERSION / 100 == 4002      // rocPRIM 4.2.x (ROCm 7.2): the release whose detail:: entry points are used below
#define GSR_ONESWEEP_DRIVER 1
#else
#define GSR_ONESWEEP_DRIVER 0          // unknown internals: public rocprim::radix_sort_pairs only
#endif

namespace gsr {

int option_sort_driver();   // 1 (default): one-clear Onesweep driver of this file; 0: rocprim::radix_sort_pairs (gsr_set_option("sort_driver", ...))

#if GSR_ONESWEEP_DRIVER
using SortOffset = unsigned int;
using SortBlockId = rocprim::detail::block_id_wrapper<unsigned int, true>;
using SortLookback = rocprim::detail::onesweep_lookback_state;
static_assert(sizeof(SortLookback) == 4, "onesweep_lookback_state is expected to be one 32-bit word (flag in the top two bits)");
static_assert(SortLookback::EMPTY == 0, "a zero-filled look-back state must mean EMPTY: the driver clears the states with zeros");

template <unsigned BS, unsigned IPT, unsigned BITS>
__global__ void __launch_bounds__(BS) sort_histogram_kernel(const uint32_t* keys, SortOffset* digit_counts, SortOffset size, SortOffset full_blocks,
                                                             unsigned begin_bit, unsigned end_bit) {
	rocprim::detail::onesweep_histograms<BS, IPT, BITS, false>(keys, digit_counts, size, full_blocks, rocprim::identity_decomposer{}, begin_bit, end_bit);
}
// One digit pass.  `digit_counts` are the RAW digit counts of this place (what the histogram kernel — or whoever accumulated them, see
// onesweep_sort_pairs — left): every workgroup turns them into exclusive offsets itself, in LDS (2^BITS <= BS values: one wave-level
// scan and one cross-wave step), instead of a separate one-workgroup-per-place scan kernel in front of the passes.  A dispatch costs
// ~5 us on this part whatever it does and these sorts are launch-latency-bound; 256 or 512 redundant adds per workgroup are free.
template <unsigned BS, unsigned IPT, unsigned BITS, class ValuesIn, class Value>
__global__ void __launch_bounds__(BS) sort_pass_kernel(const uint32_t* keys_in, uint32_t* keys_out, ValuesIn values_in, Value* values_out, unsigned size,
                                                        const SortOffset* digit_counts, SortOffset* digit_offsets_out, SortLookback* lookback, unsigned bit,
                                                        unsigned current_bits, unsigned full_blocks, SortBlockId block_id) {
	constexpr unsigned radix = 1u << BITS;
	static_assert(radix <= BS && BS % 64 == 0 && BS <= 1024, "one digit per thread");
	__shared__ SortOffset s_offsets[radix];
	__shared__ SortOffset s_wave_total[BS / 64];
	{
		const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
		const SortOffset mine = t < radix ? digit_counts[t] : 0u;
		SortOffset incl = mine;
#pragma unroll
		for (unsigned off = 1; off < 64; off <<= 1) {
			const SortOffset o = __shfl_up(incl, off, 64);
			if (lane >= off) incl += o;
		}
		if (lane == 63) s_wave_total[wave] = incl;
		__syncthreads();
		SortOffset base = 0;
		for (unsigned w = 0; w < wave; w++) base += s_wave_total[w];
		if (t < radix) s_offsets[t] = base + incl - mine;
		__syncthreads();
	}
	rocprim::detail::onesweep_iteration<BS, IPT, BITS, false, rocprim::block_radix_rank_algorithm::match>(
	    keys_in, keys_out, values_in, values_out, size, s_offsets, digit_offsets_out, lookback, rocprim::identity_decomposer{}, bit, current_bits,
	    full_blocks, block_id);
}

// Bytes at the start of the temp region that must be zero when the sort starts (multiple of 256).  A caller that has a
// kernel running in front of the sort anyway can clear them there (sort_clear_region below) and pass pre_cleared = true:
// one dispatch less.
template <unsigned BS, unsigned IPT, unsigned BITS>
size_t onesweep_cleared_bytes(size_t size_, unsigned begin_bit, unsigned end_bit) {
	constexpr unsigned radix = 1u << BITS, items_per_block = BS * IPT;
	if (size_ >= ((size_t)1 << 30) || end_bit <= begin_bit) return 0;
	const unsigned size = (unsigned)size_;
	const unsigned places = (end_bit - begin_bit + BITS - 1) / BITS;
	const unsigned blocks = (size + items_per_block - 1) / items_per_block;
	auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
	return up((size_t)radix * places * sizeof(SortOffset)) + up((size_t)radix * sizeof(SortOffset)) +
	       up((size_t)radix * (blocks ? blocks : 1) * sizeof(SortLookback)) * places + up((size_t)places * 64);
}
// grid-stride clear of `bytes` (multiple of 16) at `ptr` (16-byte aligned) by the calling kernel's threads
__device__ __forceinline__ void sort_clear_region(void* ptr, size_t bytes, size_t thread, size_t threads) {
	uint4* p = static_cast<uint4*>(ptr);
	for (size_t i = thread; i < bytes / 16; i += threads) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// temp == nullptr: returns the required bytes in `bytes` and does nothing else.  size < 2^30.
// ext_counts != nullptr: the raw digit counts — 2^BITS per place, place p counting digit (key >> (begin_bit + p * BITS)) & (2^BITS - 1)
// over ALL `size` keys — have already been accumulated there by a kernel that had the keys in its hands anyway (the per-Gaussian
// statistics kernel for the depth keys, key emission for the tile ids): no histogram dispatch.
template <unsigned BS, unsigned IPT, unsigned BITS, class ValuesIn, class Value>
hipError_t onesweep_sort_pairs(void* temp, size_t& bytes, const uint32_t* keys_in, uint32_t* keys_out, ValuesIn values_in, Value* values_out, size_t size_,
                               unsigned begin_bit, unsigned end_bit, hipStream_t stream, bool pre_cleared = false, const SortOffset* ext_counts = nullptr,
                               hipEvent_t before_last_pass = nullptr) {
	static_assert(sizeof(Value) == 4 || sizeof(Value) == 8, "4- or 8-byte values");
	constexpr unsigned radix = 1u << BITS, items_per_block = BS * IPT;
	if (size_ >= ((size_t)1 << 30) || end_bit <= begin_bit) return hipErrorInvalidValue;
	const unsigned size = (unsigned)size_;
	const unsigned places = (end_bit - begin_bit + BITS - 1) / BITS;
	const unsigned blocks = (size + items_per_block - 1) / items_per_block;
	const unsigned full_blocks = size % items_per_block == 0 ? blocks : blocks - 1;
	auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
	// cleared region: [digit offsets radix*places][offsets of the next batch radix][per pass: look-back radix*blocks][block-id counters places]
	const size_t o_digits = 0;
	const size_t o_next = o_digits + up((size_t)radix * places * sizeof(SortOffset));
	const size_t o_lookback = o_next + up((size_t)radix * sizeof(SortOffset));
	const size_t lookback_pass = up((size_t)radix * (blocks ? blocks : 1) * sizeof(SortLookback));
	const size_t o_ids = o_lookback + lookback_pass * places;
	const size_t cleared = o_ids + up((size_t)places * 64);   // one counter per 64 bytes
	const size_t o_keys_tmp = cleared;
	const size_t o_vals_tmp = o_keys_tmp + up((size_t)size * 4);
	const size_t total = o_vals_tmp + up((size_t)size * sizeof(Value));
	if (temp == nullptr) { bytes = total; return hipSuccess; }
	if (bytes < total) return hipErrorInvalidValue;
	if (size == 0) return hipSuccess;
	char* base = static_cast<char*>(temp);
	SortOffset* digits = reinterpret_cast<SortOffset*>(base + o_digits);
	SortOffset* next = reinterpret_cast<SortOffset*>(base + o_next);
	uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(base + o_keys_tmp);
	Value* values_tmp = reinterpret_cast<Value*>(base + o_vals_tmp);

	if (!pre_cleared) {
		hipError_t e = hipMemsetAsync(base, 0, cleared, stream);
		if (e != hipSuccess) return e;
	}
	const SortOffset* counts = ext_counts;
	if (counts == nullptr) {
		// the histogram has its own items per thread: with the passes' 8 (tile sort at C5 sizes) it launches twice the workgroups, each of
		// which flushes 2^BITS counters per place with atomics — 39 us against 25 us for 15 M keys
		constexpr unsigned HIPT = IPT < 16 ? 16 : IPT, hist_items = BS * HIPT;
		const unsigned hist_blocks = (size + hist_items - 1) / hist_items;
		const unsigned hist_full = size % hist_items == 0 ? hist_blocks : hist_blocks - 1;
		sort_histogram_kernel<BS, HIPT, BITS><<<hist_blocks, BS, 0, stream>>>(keys_in, digits, size, hist_full, begin_bit, end_bit);
		counts = digits;
	}

	bool to_output = (places - 1) % 2 == 0, from_input = true;
	unsigned place = 0;
	for (unsigned bit = begin_bit; bit < end_bit; bit += BITS, ++place) {
		const unsigned current_bits = (end_bit - bit) < BITS ? (end_bit - bit) : BITS;
		SortLookback* lookback = reinterpret_cast<SortLookback*>(base + o_lookback + lookback_pass * place);
		SortBlockId block_id = SortBlockId::create(base + o_ids + (size_t)place * 64);
		const SortOffset* d_in = counts + (size_t)place * radix;
		uint32_t* k_out = to_output ? keys_out : keys_tmp;
		Value* v_out = to_output ? values_out : values_tmp;
		if (before_last_pass != nullptr && place + 1 == places) {     // (a caller that must not let another kernel take the chip before the LAST pass has started)
			hipError_t e = hipEventRecord(before_last_pass, stream);
			if (e != hipSuccess) return e;
		}
		if (from_input) {
			sort_pass_kernel<BS, IPT, BITS, ValuesIn, Value><<<blocks, BS, 0, stream>>>(keys_in, k_out, values_in, v_out, size, d_in, next, lookback, bit,
			                                                                            current_bits, full_blocks, block_id);
		} else {
			const uint32_t* k_in = to_output ? keys_tmp : keys_out;
			const Value* v_in = to_output ? values_tmp : values_out;
			sort_pass_kernel<BS, IPT, BITS, const Value*, Value><<<blocks, BS, 0, stream>>>(k_in, k_out, v_in, v_out, size, d_in, next, lookback, bit,
			                                                                                current_bits, full_blocks, block_id);
		}
		from_input = false;
		to_output = !to_output;
	}
	return hipGetLastError();
}
#else   // !GSR_ONESWEEP_DRIVER: nothing to clear, and the sort entry point reports "not available" so that callers take rocPRIM's own
template <unsigned BS, unsigned IPT, unsigned BITS>
size_t onesweep_cleared_bytes(size_t, unsigned, unsigned) { return 0; }
__device__ __forceinline__ void sort_clear_region(void*, size_t, size_t, size_t) {}
template <unsigned BS, unsigned IPT, unsigned BITS, class ValuesIn, class Value>
hipError_t onesweep_sort_pairs(void* temp, size_t& bytes, const uint32_t*, uint32_t*, ValuesIn, Value*, size_t, unsigned, unsigned, hipStream_t, bool = false,
                               const unsigned* = nullptr, hipEvent_t = nullptr) {
	if (temp == nullptr) bytes = 0;
	return hipErrorNotSupported;
}
#endif

}  // namespace gsr
