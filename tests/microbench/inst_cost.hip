// Micro-benchmark (development aid, not a test): issue cost of the instruction kinds the tile kernels are made of,
// in shader cycles per wave64 instruction, as a function of resident waves per SIMD and of the EXEC mask.
//   hipcc --offload-arch=gfx950 -O3 inst_cost.hip -o inst_cost && ./inst_cost
// Each kernel is one asm loop of 32 instructions of one kind on 8 independent registers, timed in-kernel with
// s_memtime (tick = shader cycle); a workgroup holds w waves per SIMD (blocks of 256*w threads, one or two per CU).
// Printed: median cycles per instruction per wave, and that figure divided by w = cycles the SIMD spends per
// wave-instruction when w waves share it (the throughput figure the kernels are priced with).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY32(I) R8(I) R8(I) R8(I) R8(I)

#define LOOP_HEAD                                                                                        \
	"s_mov_b64 s[20:21], exec\n\t"                                                                      \
	"s_mov_b64 exec, %[mask]\n\t"                                                                       \
	"s_mov_b32 s30, 0x3f800100\n\t s_mov_b32 s31, 0x3f000000\n\t" \
	"s_mov_b32 s24, %[iters]\n\t"                                                                       \
	"s_barrier\n\t"                                                                                     \
	"s_memtime s[22:23]\n\t s_waitcnt lgkmcnt(0)\n\t"                                                   \
	"1:\n\t"
#define LOOP_TAIL                                                                                        \
	"s_sub_u32 s24, s24, 1\n\t s_cmp_lg_u32 s24, 0\n\t s_cbranch_scc1 1b\n\t"                           \
	"s_memtime s[26:27]\n\t s_waitcnt lgkmcnt(0)\n\t"                                                   \
	"s_mov_b64 exec, s[20:21]\n\t"                                                                      \
	"s_sub_u32 s26, s26, s22\n\t"                                                                       \
	"v_mov_b32 %[cyc], s26\n\t"
#define CLOB "s20", "s21", "s22", "s23", "s24", "s26", "s27", "s30", "s31", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "vcc", "scc", "memory", "v40", "v41", "v42", "v43"

#define DEF_KERNEL_F(NAME, INST)                                                                         \
	__global__ void NAME(uint32_t* out, int iters, unsigned long long mask, float a, float b, const float* mem) { \
		float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
		uint32_t cyc;                                                                                   \
		__shared__ float lds[1024];                                                                     \
		lds[threadIdx.x] = x0;                                                                          \
		uint32_t la = (threadIdx.x & 1023) * 4;                                                         \
		asm volatile(LOOP_HEAD BODY32(INST) "2:\n\t" LOOP_TAIL                                                   \
		             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), [cyc] "=v"(cyc) \
		             : "v"(a), "v"(b), [iters] "s"(iters), [mask] "s"(mask), [mem] "s"(mem), [la] "v"(la), [z] "v"(0u), [la4] "v"((threadIdx.x & 255u) * 16u)  \
		             : CLOB);                                                                           \
		out[blockIdx.x * blockDim.x + threadIdx.x] = cyc;                                               \
		if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678f) out[0] = 1;                           \
	}
#define DEF_KERNEL_P(NAME, INST)                                                                         \
	__global__ void NAME(uint32_t* out, int iters, unsigned long long mask, float a_, float b_, const float* mem) { \
		float t = threadIdx.x + 1.f;                                                                    \
		v2f x0 = {t, t + 1}, x1 = {t + 2, t + 3}, x2 = {t + 4, t + 5}, x3 = {t + 6, t + 7}, x4 = {t + 8, t + 9}, x5 = {t + 10, t + 11}, x6 = {t + 12, t + 13}, x7 = {t + 14, t + 15}; \
		v2f a = {a_, a_}, b = {b_, b_};                                                                 \
		uint32_t cyc;                                                                                   \
		uint32_t la = (threadIdx.x & 1023) * 4;                                                         \
		asm volatile(LOOP_HEAD BODY32(INST) "2:\n\t" LOOP_TAIL                                                   \
		             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), [cyc] "=v"(cyc) \
		             : "v"(a), "v"(b), [iters] "s"(iters), [mask] "s"(mask), [mem] "s"(mem), [la] "v"(la), [z] "v"(0u), [la4] "v"((threadIdx.x & 255u) * 16u)  \
		             : CLOB);                                                                           \
		out[blockIdx.x * blockDim.x + threadIdx.x] = cyc;                                               \
		if (x0.x + x1.x + x2.x + x3.x + x4.y + x5.y + x6.y + x7.y == 12345.678f) out[0] = 1;           \
	}

#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %9, %10\n\t"
#define I_FMA_S(i) "v_fma_f32 %" #i ", %" #i ", s30, %10\n\t"
#define I_FMA_SS(i) "v_fma_f32 %" #i ", %" #i ", s30, s30\n\t"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %9\n\t"
#define I_MUL_S(i) "v_mul_f32 %" #i ", s30, %" #i "\n\t"
#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %10\n\t"
#define I_MOV(i) "v_mov_b32 %" #i ", %9\n\t"
#define I_MAX(i) "v_max_f32 %" #i ", %" #i ", %10\n\t"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %9, %10\n\t"
#define I_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define I_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n\t"
#define I_CMP_S(i) "v_cmp_lt_f32 s[34:35], %" #i ", %9\n\t"
#define I_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n\t v_cndmask_b32 %" #i ", %" #i ", %10, vcc\n\t"
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define I_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPB(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_mirror row_mask:0xf bank_mask:0x3\n\t"
#define I_DPPQ(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_RDL(i) "v_readlane_b32 s6" #i ", %" #i ", 5\n\t"
#define I_RDFL(i) "v_readfirstlane_b32 s6" #i ", %" #i "\n\t"
#define I_SALU(i) "s_add_u32 s4" #i ", s4" #i ", 1\n\t"
#define I_SMOV(i) "s_mov_b32 s4" #i ", s30\n\t"
#define I_FMA_SALU(i) "v_fma_f32 %" #i ", %" #i ", %9, %10\n\t s_add_u32 s4" #i ", s4" #i ", 1\n\t"
#define I_FMA_2SALU(i) "v_fma_f32 %" #i ", %" #i ", %9, %10\n\t s_add_u32 s4" #i ", s4" #i ", 1\n\t s_and_b32 s6" #i ", s4" #i ", 7\n\t"
#define I_NOP(i) "s_nop 0\n\t"
#define I_SLOAD4(i) "s_load_dwordx4 s[36:39], %[mem], 0x0\n\t s_waitcnt lgkmcnt(0)\n\t"
#define I_SLOAD16(i) "s_load_dwordx16 s[36:51], %[mem], 0x0\n\t s_waitcnt lgkmcnt(0)\n\t"
#define I_SLOAD4_NW(i) "s_load_dwordx4 s[36:39], %[mem], 0x" #i "0\n\t"
#define I_LDSW(i) "ds_write_b32 %[la], %" #i "\n\t"
#define I_LDSR(i) "ds_read_b32 %" #i ", %[la]\n\t"
#define I_LDSR_W(i) "ds_read_b32 %" #i ", %[la]\n\t s_waitcnt lgkmcnt(0)\n\t"
#define I_BALLOT(i) "v_cmp_lt_f32 s[34:35], %" #i ", %9\n\t s_bcnt1_i32_b64 s36, s[34:35]\n\t"
#define I_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %9, %10\n\t"
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %9\n\t"
#define I_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %10\n\t"
#define I_PKFMA_S(i) "v_pk_fma_f32 %" #i ", %" #i ", s[30:31], %10\n\t"
#define I_PKMUL_S(i) "v_pk_mul_f32 %" #i ", %" #i ", s[30:31]\n\t"
#define I_PKFMA_OPSEL(i) "v_pk_fma_f32 %" #i ", %" #i ", %9, %10 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n\t"


#define I_CND_S(i) "v_cndmask_b32 %" #i ", %" #i ", %9, s[34:35]\n\t"
#define I_SAND_CND(i) "s_and_b64 s[34:35], s[36:37], s[38:39]\n\t v_cndmask_b32 %" #i ", %" #i ", %9, s[34:35]\n\t"
#define I_CMP_CND4(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n\t v_cndmask_b32 %" #i ", %" #i ", %10, vcc\n\t v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t v_cndmask_b32 %" #i ", %" #i ", %10, vcc\n\t v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define I_CMPS_CND(i) "v_cmp_lt_f32 s[34:35], %" #i ", %9\n\t v_cndmask_b32 %" #i ", %" #i ", %10, s[34:35]\n\t"
#define I_MUL_LIT(i) "v_mul_f32 %" #i ", 0x3fb8aa3b, %" #i "\n\t"
#define I_MUL_INL(i) "v_mul_f32 %" #i ", 0.5, %" #i "\n\t"
#define I_FMAC(i) "v_fmac_f32 %" #i ", %9, %10\n\t"
#define I_FMAC_S(i) "v_fmac_f32 %" #i ", s30, %10\n\t"
#define I_MIN(i) "v_min_f32 %" #i ", %" #i ", %10\n\t"
#define I_SUB(i) "v_sub_f32 %" #i ", %" #i ", %10\n\t"
#define I_MAXI(i) "v_max_i32 %" #i ", %" #i ", %10\n\t"
#define I_ANDB(i) "v_and_b32 %" #i ", %" #i ", %10\n\t"
#define I_MAXDPP(i) "v_max_f32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define I_RDL_S(i) "v_readlane_b32 s6" #i ", %" #i ", s24\n\t"
#define I_GLOAD_U(i) "global_load_dwordx4 v[40:43], %[z], %[mem]\n\t"
#define I_GLOAD_UW(i) "global_load_dwordx4 v[40:43], %[z], %[mem]\n\t s_waitcnt vmcnt(0)\n\t"
#define I_LDSR128_U(i) "ds_read_b128 v[40:43], %[z]\n\t"
#define I_LDSR128_UW(i) "ds_read_b128 v[40:43], %[z]\n\t s_waitcnt lgkmcnt(0)\n\t"
#define I_LDSW128(i) "ds_write_b128 %[la4], v[40:43]\n\t"
#define I_SCMP_BR(i) "s_cmp_lg_u32 s24, 0\n\t s_cbranch_scc0 2f\n\t"
#define I_SAND64(i) "s_and_b64 s[34:35], s[36:37], s[38:39]\n\t"
#define I_FMA_NOP(i) "v_fma_f32 %" #i ", %" #i ", %9, %10\n\t s_nop 1\n\t"

DEF_KERNEL_F(k_fma, I_FMA)
DEF_KERNEL_F(k_fma_s, I_FMA_S)
DEF_KERNEL_F(k_fma_ss, I_FMA_SS)
DEF_KERNEL_F(k_mul, I_MUL)
DEF_KERNEL_F(k_mul_s, I_MUL_S)
DEF_KERNEL_F(k_add, I_ADD)
DEF_KERNEL_F(k_mov, I_MOV)
DEF_KERNEL_F(k_max, I_MAX)
DEF_KERNEL_F(k_med3, I_MED3)
DEF_KERNEL_F(k_cnd, I_CND)
DEF_KERNEL_F(k_cmp, I_CMP)
DEF_KERNEL_F(k_cmp_s, I_CMP_S)
DEF_KERNEL_F(k_cmpcnd, I_CMPCND)
DEF_KERNEL_F(k_exp, I_EXP)
DEF_KERNEL_F(k_rcp, I_RCP)
DEF_KERNEL_F(k_dpp, I_DPP)
DEF_KERNEL_F(k_dppb, I_DPPB)
DEF_KERNEL_F(k_dppq, I_DPPQ)
DEF_KERNEL_F(k_movdpp, I_MOVDPP)
DEF_KERNEL_F(k_rdl, I_RDL)
DEF_KERNEL_F(k_rdfl, I_RDFL)
DEF_KERNEL_F(k_salu, I_SALU)
DEF_KERNEL_F(k_smov, I_SMOV)
DEF_KERNEL_F(k_fma_salu, I_FMA_SALU)
DEF_KERNEL_F(k_fma_2salu, I_FMA_2SALU)
DEF_KERNEL_F(k_nop, I_NOP)
DEF_KERNEL_F(k_sload4, I_SLOAD4)
DEF_KERNEL_F(k_sload16, I_SLOAD16)
DEF_KERNEL_F(k_sload4_nw, I_SLOAD4_NW)
DEF_KERNEL_F(k_ldsw, I_LDSW)
DEF_KERNEL_F(k_ldsr, I_LDSR)
DEF_KERNEL_F(k_ldsr_w, I_LDSR_W)
DEF_KERNEL_F(k_ballot, I_BALLOT)

DEF_KERNEL_F(k_cnd_s, I_CND_S)
DEF_KERNEL_F(k_sand_cnd, I_SAND_CND)
DEF_KERNEL_F(k_cmp_cnd4, I_CMP_CND4)
DEF_KERNEL_F(k_cmps_cnd, I_CMPS_CND)
DEF_KERNEL_F(k_mul_lit, I_MUL_LIT)
DEF_KERNEL_F(k_mul_inl, I_MUL_INL)
DEF_KERNEL_F(k_fmac, I_FMAC)
DEF_KERNEL_F(k_fmac_s, I_FMAC_S)
DEF_KERNEL_F(k_min, I_MIN)
DEF_KERNEL_F(k_sub, I_SUB)
DEF_KERNEL_F(k_maxi, I_MAXI)
DEF_KERNEL_F(k_andb, I_ANDB)
DEF_KERNEL_F(k_maxdpp, I_MAXDPP)
DEF_KERNEL_F(k_rdl_s, I_RDL_S)
DEF_KERNEL_F(k_gload_u, I_GLOAD_U)
DEF_KERNEL_F(k_gload_uw, I_GLOAD_UW)
DEF_KERNEL_F(k_ldsr128_u, I_LDSR128_U)
DEF_KERNEL_F(k_ldsr128_uw, I_LDSR128_UW)
DEF_KERNEL_F(k_ldsw128, I_LDSW128)
DEF_KERNEL_F(k_sand64, I_SAND64)
DEF_KERNEL_F(k_fma_nop, I_FMA_NOP)
DEF_KERNEL_P(k_pkfma, I_PKFMA)
DEF_KERNEL_P(k_pkmul, I_PKMUL)
DEF_KERNEL_P(k_pkadd, I_PKADD)
DEF_KERNEL_P(k_pkfma_s, I_PKFMA_S)
DEF_KERNEL_P(k_pkmul_s, I_PKMUL_S)
DEF_KERNEL_P(k_pkfma_opsel, I_PKFMA_OPSEL)

typedef void (*kern_t)(uint32_t*, int, unsigned long long, float, float, const float*);
struct Entry { const char* name; kern_t k; int per; };   // per = instructions per macro instance

static bool g_small_blocks = false;
static double run(kern_t k, int w, unsigned long long mask, int iters, uint32_t* d_out, const float* d_mem, double* wall_ms) {
	// w waves per SIMD: block of 256*w threads (<= 1024), grid 256 (w <= 4) or 512 (w == 8, two blocks of 1024 per CU)
	const int threads = g_small_blocks ? 64 : (w <= 4 ? 256 * w : 1024);
	const int grid = g_small_blocks ? 256 * 4 * w : (w <= 4 ? 256 : 512);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	k<<<grid, threads>>>(d_out, iters, mask, 1.0001f, 0.5f, d_mem);
	(void)hipEventRecord(e0);
	k<<<grid, threads>>>(d_out, iters, mask, 1.0001f, 0.5f, d_mem);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	*wall_ms = ms;
	std::vector<uint32_t> h((size_t)grid * threads);
	(void)hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
	std::vector<uint32_t> c;
	for (size_t i = 0; i < h.size(); i += 64) c.push_back(h[i]);
	std::sort(c.begin(), c.end());
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return (double)c[c.size() / 2];
}

int main() {
	uint32_t* d_out;
	float* d_mem;
	(void)hipMalloc(&d_out, 512 * 1024 * 4);
	(void)hipMalloc(&d_mem, 4096);
	(void)hipMemset(d_mem, 0, 4096);
	const int iters = 20000;
	const Entry es[] = {
	    {"v_fma_f32 vvv", k_fma, 1}, {"v_fma_f32 v,s,v", k_fma_s, 1}, {"v_fma_f32 v,s,s", k_fma_ss, 1}, {"v_mul_f32", k_mul, 1}, {"v_mul_f32 s,v", k_mul_s, 1},
	    {"v_add_f32", k_add, 1}, {"v_mov_b32", k_mov, 1}, {"v_max_f32", k_max, 1}, {"v_med3_f32", k_med3, 1}, {"v_cndmask vcc", k_cnd, 1},
	    {"v_cmp -> vcc", k_cmp, 1}, {"v_cmp -> sgpr pair", k_cmp_s, 1}, {"v_cmp+v_cndmask (2)", k_cmpcnd, 2}, {"v_exp_f32", k_exp, 1}, {"v_rcp_f32", k_rcp, 1},
	    {"v_add_f32_dpp row_mirror", k_dpp, 1}, {"v_add_f32_dpp bank 0x3", k_dppb, 1}, {"v_add_f32_dpp quad_perm", k_dppq, 1}, {"v_mov_b32_dpp", k_movdpp, 1},
	    {"v_readlane_b32", k_rdl, 1}, {"v_readfirstlane_b32", k_rdfl, 1}, {"s_add_u32", k_salu, 1}, {"s_mov_b32", k_smov, 1},
	    {"v_fma + s_add (2)", k_fma_salu, 2}, {"v_fma + 2 SALU (3)", k_fma_2salu, 3}, {"s_nop 0", k_nop, 1},
	    {"s_load_dwordx4 + wait", k_sload4, 1}, {"s_load_dwordx16 + wait", k_sload16, 1}, {"s_load_dwordx4 no wait", k_sload4_nw, 1},
	    {"ds_write_b32", k_ldsw, 1}, {"ds_read_b32", k_ldsr, 1}, {"ds_read_b32 + wait", k_ldsr_w, 1}, {"v_cmp->s + s_bcnt1 (2)", k_ballot, 2},
	    {"v_cndmask sgpr-pair stale", k_cnd_s, 1}, {"s_and_b64 + v_cndmask (2)", k_sand_cnd, 2}, {"v_cmp vcc + 4 v_cndmask (5)", k_cmp_cnd4, 5}, {"v_cmp s[] + v_cndmask (2)", k_cmps_cnd, 2},
	    {"v_mul_f32 literal", k_mul_lit, 1}, {"v_mul_f32 inline const", k_mul_inl, 1}, {"v_fmac_f32 (VOP2)", k_fmac, 1}, {"v_fmac_f32 s,v", k_fmac_s, 1}, {"v_min_f32", k_min, 1}, {"v_sub_f32", k_sub, 1},
	    {"v_max_i32", k_maxi, 1}, {"v_and_b32", k_andb, 1}, {"v_max_f32_dpp row_shr", k_maxdpp, 1}, {"v_readlane_b32 lane in sgpr", k_rdl_s, 1},
	    {"global_load_dwordx4 uniform", k_gload_u, 1}, {"global_load_dwordx4 unif+wait", k_gload_uw, 1}, {"ds_read_b128 uniform", k_ldsr128_u, 1}, {"ds_read_b128 unif + wait", k_ldsr128_uw, 1},
	    {"ds_write_b128", k_ldsw128, 1}, {"s_and_b64", k_sand64, 1}, {"v_fma + s_nop 1 (2)", k_fma_nop, 2},
	    {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_add_f32", k_pkadd, 1}, {"v_pk_fma_f32 v,s2,v", k_pkfma_s, 1}, {"v_pk_mul_f32 v,s2", k_pkmul_s, 1},
	    {"v_pk_fma_f32 op_sel", k_pkfma_opsel, 1},
	};
	printf("# cycles per macro instance per wave (median over waves) | the same divided by waves per SIMD | wall ms\n");
	printf("%-28s", "kind");
	for (int w : {1, 2, 4, 8}) printf("   w=%d: /wave /SIMD wall/SIMD", w);
	printf("\n");
	for (const Entry& e : es) {
		printf("%-28s", e.name);
		for (int w : {1, 2, 4, 8}) {
			double wall;
			const double cyc = run(e.k, w, ~0ull, iters, d_out, d_mem, &wall);
			const double per = cyc / (iters * 32.0);
			printf("   %6.2f %5.2f %5.2fns", per, per / w, wall * 1e6 / (iters * 32.0 * e.per) / w * e.per);
		}
		printf("\n");
		fflush(stdout);
	}
	printf("\n# EXEC-mask dependence (w = 4): does a wave64 instruction with an empty 32-lane half issue in one pass?\n");
	const unsigned long long masks[] = {~0ull, 0x00000000ffffffffull, 0xffffffff00000000ull, 0x0000ffff0000ffffull, 0x000000000000ffffull, 0x1ull};
	const Entry ms[] = {{"v_fma_f32 vvv", k_fma, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_exp_f32", k_exp, 1}, {"v_add_f32_dpp row_mirror", k_dpp, 1}, {"v_cndmask vcc", k_cnd, 1}, {"ds_write_b32", k_ldsw, 1}};
	for (const Entry& e : ms) {
		printf("%-28s", e.name);
		for (unsigned long long m : masks) {
			double wall;
			const double cyc = run(e.k, 4, m, iters, d_out, d_mem, &wall);
			printf("   exec=%016llx: %6.2f", m, cyc / (iters * 32.0) / 4);
		}
		printf("\n");
		fflush(stdout);
	}
	printf("\n# one-wave workgroups (grid = 1024 w blocks of 64 threads, as the tile kernels launch): cycles per instruction per SIMD\n");
	g_small_blocks = true;
	const Entry ss[] = {{"v_fma_f32 vvv", k_fma, 1}, {"v_fma_f32 v,s,v", k_fma_s, 1}, {"v_mul_f32", k_mul, 1}, {"v_pk_fma_f32 v,s2,v", k_pkfma_s, 1}, {"v_add_f32_dpp row_mirror", k_dpp, 1}, {"v_exp_f32", k_exp, 1}, {"v_cmp+v_cndmask (2)", k_cmpcnd, 2}, {"s_add_u32", k_salu, 1}, {"v_fma + 2 SALU (3)", k_fma_2salu, 3}, {"s_load_dwordx16 + wait", k_sload16, 1}};
	printf("%-28s", "kind");
	for (int w = 1; w <= 8; w++) printf("  w=%d ", w);
	printf("\n");
	for (const Entry& e : ss) {
		printf("%-28s", e.name);
		for (int w = 1; w <= 8; w++) {
			double wall;
			const double cyc = run(e.k, w, ~0ull, iters, d_out, d_mem, &wall);
			printf(" %5.2f/%4.2fns", cyc / (iters * 32.0) / w, wall * 1e6 / (iters * 32.0) / w);
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
