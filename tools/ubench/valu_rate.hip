// valu_rate.hip -- issue cost of single VALU/SALU instructions on gfx950, per SIMD,
// at 1..8 waves per SIMD.  Each wave runs ITER x 16 independent copies of one
// instruction (inline asm, distinct destinations) and stamps s_memtime around it.
// Output: cycles per wave-instruction per SIMD = elapsed / (ITER*16*waves_per_simd).
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define ITER 4096

#define REP16(OP) \
	OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

#define KERNEL(NAME, ASMSTR) \
__global__ void __launch_bounds__(256) NAME(unsigned long long *out, float seed) \
{ \
	float a = seed + threadIdx.x, b = seed * 3.0f; \
	float r[16]; \
	for(int i = 0; i < 16; i++) r[i] = a + i; \
	unsigned long long q0 = __builtin_amdgcn_s_memrealtime(); \
	unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
	for(int it = 0; it < ITER; it++) \
	{ \
		asm volatile(ASMSTR \
			: "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
			  "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) \
			: "v"(a), "v"(b) : "vcc", "scc", "s4", "s5", "s6", "s7"); \
	} \
	unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
	unsigned long long q1 = __builtin_amdgcn_s_memrealtime(); \
	float s = 0; for(int i = 0; i < 16; i++) s += r[i]; \
	if(s == 12345.678f) out[0] = 1; \
	if((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = ((t1 - t0) << 24) | (q1 - q0); \
}

#define L(n, txt) txt "\n"
// one line per destination %0..%15; %16 = a, %17 = b
#define ASM16(fmt_pre, fmt_post) \
	fmt_pre "%0" fmt_post "\n" fmt_pre "%1" fmt_post "\n" fmt_pre "%2" fmt_post "\n" fmt_pre "%3" fmt_post "\n" \
	fmt_pre "%4" fmt_post "\n" fmt_pre "%5" fmt_post "\n" fmt_pre "%6" fmt_post "\n" fmt_pre "%7" fmt_post "\n" \
	fmt_pre "%8" fmt_post "\n" fmt_pre "%9" fmt_post "\n" fmt_pre "%10" fmt_post "\n" fmt_pre "%11" fmt_post "\n" \
	fmt_pre "%12" fmt_post "\n" fmt_pre "%13" fmt_post "\n" fmt_pre "%14" fmt_post "\n" fmt_pre "%15" fmt_post "\n"

KERNEL(k_add_f32,   ASM16("v_add_f32 ", ", %16, %17"))
KERNEL(k_mul_f32,   ASM16("v_mul_f32 ", ", %16, %17"))
KERNEL(k_fma_f32,   ASM16("v_fma_f32 ", ", %16, %17, %16"))
KERNEL(k_add_u32,   ASM16("v_add_u32 ", ", %16, %17"))
KERNEL(k_and_b32,   ASM16("v_and_b32 ", ", %16, %17"))
KERNEL(k_lshl,      ASM16("v_lshlrev_b32 ", ", 3, %17"))
KERNEL(k_bfe,       ASM16("v_bfe_u32 ", ", %16, 8, 8"))
KERNEL(k_mov,       ASM16("v_mov_b32 ", ", %16"))
KERNEL(k_cndmask,   ASM16("v_cndmask_b32 ", ", %16, %17, vcc"))
KERNEL(k_cndmask64, ASM16("v_cndmask_b32 ", ", %16, %17, s[4:5]"))
KERNEL(k_cmp,       "v_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\n" \
                    "v_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\n" \
                    "v_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\n" \
                    "v_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\nv_cmp_lt_f32 vcc, %16, %17\n")
KERNEL(k_cmp_e64,   "v_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\nv_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\n" \
                    "v_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\nv_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\n" \
                    "v_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\nv_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\n" \
                    "v_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\nv_cmp_lt_f32 s[4:5], %16, %17\nv_cmp_lt_f32 s[6:7], %16, %17\n")
// packed fp32: %0 names a single VGPR; the pair forms use explicit even-aligned register pairs
#define PK16(op) \
	op " v[32:33], v[64:65], v[66:67]\n" op " v[34:35], v[64:65], v[66:67]\n" op " v[36:37], v[64:65], v[66:67]\n" op " v[38:39], v[64:65], v[66:67]\n" \
	op " v[40:41], v[64:65], v[66:67]\n" op " v[42:43], v[64:65], v[66:67]\n" op " v[44:45], v[64:65], v[66:67]\n" op " v[46:47], v[64:65], v[66:67]\n" \
	op " v[48:49], v[64:65], v[66:67]\n" op " v[50:51], v[64:65], v[66:67]\n" op " v[52:53], v[64:65], v[66:67]\n" op " v[54:55], v[64:65], v[66:67]\n" \
	op " v[56:57], v[64:65], v[66:67]\n" op " v[58:59], v[64:65], v[66:67]\n" op " v[60:61], v[64:65], v[66:67]\n" op " v[62:63], v[64:65], v[66:67]\n"
#define KERNEL_PK(NAME, ASMSTR) \
__global__ void __launch_bounds__(256) NAME(unsigned long long *out, float seed) \
{ \
	unsigned long long q0 = __builtin_amdgcn_s_memrealtime(); \
	unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
	for(int it = 0; it < ITER; it++) \
	{ \
		asm volatile(ASMSTR ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
			"v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","s4","s5","vcc","scc"); \
	} \
	unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
	unsigned long long q1 = __builtin_amdgcn_s_memrealtime(); \
	if(seed == 12345.678f) out[0] = 1; \
	if((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = ((t1 - t0) << 24) | (q1 - q0); \
}
KERNEL_PK(k_pk_add_f32, PK16("v_pk_add_f32"))
KERNEL_PK(k_pk_mul_f32, PK16("v_pk_mul_f32"))
KERNEL_PK(k_pk_fma_f32, "v_pk_fma_f32 v[32:33], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[34:35], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[36:37], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[38:39], v[64:65], v[66:67], v[64:65]\n" \
	"v_pk_fma_f32 v[40:41], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[42:43], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[44:45], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[46:47], v[64:65], v[66:67], v[64:65]\n" \
	"v_pk_fma_f32 v[48:49], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[50:51], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[52:53], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[54:55], v[64:65], v[66:67], v[64:65]\n" \
	"v_pk_fma_f32 v[56:57], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[58:59], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[60:61], v[64:65], v[66:67], v[64:65]\nv_pk_fma_f32 v[62:63], v[64:65], v[66:67], v[64:65]\n")
KERNEL(k_min3_f32,  ASM16("v_min3_f32 ", ", %16, %17, %16"))
KERNEL(k_med3_i32,  ASM16("v_med3_i32 ", ", %16, %17, %16"))
KERNEL(k_xor_b32,   ASM16("v_xor_b32 ", ", %16, %17"))
KERNEL(k_sub_f32,   ASM16("v_sub_f32 ", ", %16, %17"))
KERNEL(k_lshl_add,  ASM16("v_lshl_add_u32 ", ", %16, 2, %17"))
KERNEL(k_pk_add_u16, ASM16("v_pk_add_u16 ", ", %16, %17"))
KERNEL(k_dot2_u16,  ASM16("v_dot2_u32_u16 ", ", %16, %17, 0"))
KERNEL(k_bitop3,    ASM16("v_bitop3_b32 ", ", %16, %17, %16 bitop3:0x48"))
KERNEL(k_add_dpp,   ASM16("v_add_f32_dpp ", ", %16, %17 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_cmp_u32,   "v_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\n" \
                    "v_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\n" \
                    "v_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\n" \
                    "v_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\nv_cmp_lt_u32 vcc, %16, %17\n")
KERNEL(k_f64_fma,   "v_fma_f64 v[32:33], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[34:35], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[36:37], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[38:39], v[64:65], v[66:67], v[64:65]\n" \
	"v_fma_f64 v[40:41], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[42:43], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[44:45], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[46:47], v[64:65], v[66:67], v[64:65]\n" \
	"v_fma_f64 v[48:49], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[50:51], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[52:53], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[54:55], v[64:65], v[66:67], v[64:65]\n" \
	"v_fma_f64 v[56:57], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[58:59], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[60:61], v[64:65], v[66:67], v[64:65]\nv_fma_f64 v[62:63], v[64:65], v[66:67], v[64:65]\n")
KERNEL_PK(k_mad_u64, "v_mad_u64_u32 v[32:33], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[34:35], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[36:37], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[38:39], s[4:5], v64, v66, v[64:65]\n" \
	"v_mad_u64_u32 v[40:41], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[42:43], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[44:45], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[46:47], s[4:5], v64, v66, v[64:65]\n" \
	"v_mad_u64_u32 v[48:49], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[50:51], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[52:53], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[54:55], s[4:5], v64, v66, v[64:65]\n" \
	"v_mad_u64_u32 v[56:57], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[58:59], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[60:61], s[4:5], v64, v66, v[64:65]\nv_mad_u64_u32 v[62:63], s[4:5], v64, v66, v[64:65]\n")
// round 3: the opcodes the issue model (tools/issue_model.py) classes without a measurement of their own
KERNEL(k_or_b32,    ASM16("v_or_b32 ", ", %16, %17"))
KERNEL(k_sub_u32,   ASM16("v_sub_u32 ", ", %16, %17"))
KERNEL(k_min_u32,   ASM16("v_min_u32 ", ", %16, %17"))
KERNEL(k_cvt_i32,   ASM16("v_cvt_i32_f32 ", ", %16"))
KERNEL(k_mul_u24,   ASM16("v_mul_u32_u24 ", ", %16, %17"))
KERNEL(k_mad_i24,   ASM16("v_mad_i32_i24 ", ", %16, %17, %16"))
KERNEL(k_and_or,    ASM16("v_and_or_b32 ", ", %16, %17, %16"))
KERNEL(k_lshl_or,   ASM16("v_lshl_or_b32 ", ", %16, 3, %17"))
KERNEL(k_perm,      ASM16("v_perm_b32 ", ", %16, %17, %16"))
KERNEL(k_cvt_pk_u8, ASM16("v_cvt_pk_u8_f32 ", ", %16, 1, %17"))
KERNEL(k_lerp_u8,   ASM16("v_lerp_u8 ", ", %16, %17, %16"))
#define F64_16(op, args) \
	op " v[32:33], " args "\n" op " v[34:35], " args "\n" op " v[36:37], " args "\n" op " v[38:39], " args "\n" \
	op " v[40:41], " args "\n" op " v[42:43], " args "\n" op " v[44:45], " args "\n" op " v[46:47], " args "\n" \
	op " v[48:49], " args "\n" op " v[50:51], " args "\n" op " v[52:53], " args "\n" op " v[54:55], " args "\n" \
	op " v[56:57], " args "\n" op " v[58:59], " args "\n" op " v[60:61], " args "\n" op " v[62:63], " args "\n"
KERNEL_PK(k_f64_mul, F64_16("v_mul_f64", "v[64:65], v[66:67]"))
KERNEL_PK(k_f64_add, F64_16("v_add_f64", "v[64:65], v[66:67]"))
KERNEL_PK(k_cvt_f64_f32, F64_16("v_cvt_f64_f32", "v64"))
KERNEL_PK(k_lshr_b64, F64_16("v_lshrrev_b64", "3, v[64:65]"))
#define F32FROM64_16(op) \
	op " v32, v[64:65]\n" op " v33, v[64:65]\n" op " v34, v[64:65]\n" op " v35, v[64:65]\n" op " v36, v[64:65]\n" op " v37, v[64:65]\n" op " v38, v[64:65]\n" op " v39, v[64:65]\n" \
	op " v40, v[64:65]\n" op " v41, v[64:65]\n" op " v42, v[64:65]\n" op " v43, v[64:65]\n" op " v44, v[64:65]\n" op " v45, v[64:65]\n" op " v46, v[64:65]\n" op " v47, v[64:65]\n"
KERNEL_PK(k_cvt_f32_f64, F32FROM64_16("v_cvt_f32_f64"))
KERNEL_PK(k_cvt_i32_f64, F32FROM64_16("v_cvt_i32_f64"))
KERNEL(k_max_f32,   ASM16("v_max_f32 ", ", %16, %17"))
KERNEL(k_cvt,       ASM16("v_cvt_f32_i32 ", ", %16"))
KERNEL(k_mul_lo,    ASM16("v_mul_lo_u32 ", ", %16, %17"))
KERNEL(k_mul_hi,    ASM16("v_mul_hi_u32 ", ", %16, %17"))
KERNEL(k_mad_u32,   ASM16("v_mad_u32_u24 ", ", %16, %17, %16"))
KERNEL(k_add3,      ASM16("v_add3_u32 ", ", %16, %17, %16"))
KERNEL(k_rcp,       ASM16("v_rcp_f32 ", ", %16"))
KERNEL(k_sqrt,      ASM16("v_sqrt_f32 ", ", %16"))
KERNEL(k_salu,      "s_add_u32 s4, s4, 1\ns_add_u32 s5, s5, 1\ns_add_u32 s6, s6, 1\ns_add_u32 s7, s7, 1\ns_add_u32 s4, s4, 1\ns_add_u32 s5, s5, 1\ns_add_u32 s6, s6, 1\ns_add_u32 s7, s7, 1\n" \
                    "s_add_u32 s4, s4, 1\ns_add_u32 s5, s5, 1\ns_add_u32 s6, s6, 1\ns_add_u32 s7, s7, 1\ns_add_u32 s4, s4, 1\ns_add_u32 s5, s5, 1\ns_add_u32 s6, s6, 1\ns_add_u32 s7, s7, 1\n")
KERNEL(k_salu64,    "s_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\ns_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\n" \
                    "s_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\ns_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\n" \
                    "s_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\ns_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\n" \
                    "s_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\ns_and_b64 s[4:5], s[4:5], exec\ns_or_b64 s[6:7], s[6:7], exec\n")
KERNEL(k_cmpcnd,    "v_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %0, %16, %17, vcc\nv_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %1, %16, %17, vcc\nv_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %2, %16, %17, vcc\nv_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %3, %16, %17, vcc\n" \
                    "v_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %4, %16, %17, vcc\nv_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %5, %16, %17, vcc\nv_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %6, %16, %17, vcc\nv_cmp_lt_f32 vcc, %16, %17\nv_cndmask_b32 %7, %16, %17, vcc\n")
// saveexec / restore pairs around one VALU op, as hipcc emits for a divergent if
KERNEL(k_ifblock,   "v_cmp_lt_f32 vcc, %16, %17\ns_and_saveexec_b64 s[4:5], vcc\nv_add_f32 %0, %16, %17\ns_or_b64 exec, exec, s[4:5]\nv_cmp_lt_f32 vcc, %16, %17\ns_and_saveexec_b64 s[4:5], vcc\nv_add_f32 %1, %16, %17\ns_or_b64 exec, exec, s[4:5]\n" \
                    "v_cmp_lt_f32 vcc, %16, %17\ns_and_saveexec_b64 s[4:5], vcc\nv_add_f32 %2, %16, %17\ns_or_b64 exec, exec, s[4:5]\nv_cmp_lt_f32 vcc, %16, %17\ns_and_saveexec_b64 s[4:5], vcc\nv_add_f32 %3, %16, %17\ns_or_b64 exec, exec, s[4:5]\n")
// alternating VALU / SALU: does the scalar stream ride along for free?
KERNEL(k_mix_vs,    "v_add_f32 %0, %16, %17\ns_add_u32 s4, s4, 1\nv_add_f32 %1, %16, %17\ns_add_u32 s5, s5, 1\nv_add_f32 %2, %16, %17\ns_add_u32 s6, s6, 1\nv_add_f32 %3, %16, %17\ns_add_u32 s7, s7, 1\n" \
                    "v_add_f32 %4, %16, %17\ns_add_u32 s4, s4, 1\nv_add_f32 %5, %16, %17\ns_add_u32 s5, s5, 1\nv_add_f32 %6, %16, %17\ns_add_u32 s6, s6, 1\nv_add_f32 %7, %16, %17\ns_add_u32 s7, s7, 1\n")

template<int OPS> static void run(const char *name, void (*k)(unsigned long long *, float), unsigned long long *d, int ncu)
{
	printf("%-12s", name);
	const int wlist[] = { 1, 2, 4, 5, 6, 8 };
	for(int wps : wlist)
	{
		// wps workgroups of 256 threads per CU: one wave of each on every SIMD
		int threads = 256;
		int nblk = ncu * wps;
		int nw = ncu * 4 * wps;
		hipMemset(d, 0, 8 * (nw + 1));
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipLaunchKernelGGL(k, dim3(nblk), dim3(threads), 0, 0, d, 1.0f);
		const int NL = 8;                      // back-to-back launches under the host's clock
		hipEventRecord(e0, 0);
		for(int l = 0; l < NL; l++) hipLaunchKernelGGL(k, dim3(nblk), dim3(threads), 0, 0, d, 1.0f);
		hipEventRecord(e1, 0);
		hipDeviceSynchronize();
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		ms /= NL;
		hipEventDestroy(e0); hipEventDestroy(e1);
		std::vector<unsigned long long> h(nw + 1);
		hipMemcpy(h.data(), d, 8 * (nw + 1), hipMemcpyDeviceToHost);
		double s = 0, q = 0; for(int i = 1; i <= nw; i++) { s += (double)(h[i] >> 24); q += (double)(h[i] & 0xffffff); }
		double per = s / nw / ((double)ITER * OPS * wps);
		printf("  w%d: %5.2f", wps, per);
		if(getenv("VALU_RATE_CALIBRATE"))
			// s_memtime ticks against the 100 MHz s_memrealtime and against the host's event clock:
			// tick frequency, and wave-instructions per SIMD per nanosecond of the whole launch
			printf(" [tick %.0f MHz, launch %.3f ms = %.3f instr/ns/SIMD]", s / q * 100.0, ms,
				(double)ITER * OPS * wps / (ms * 1e6));
	}
	printf("   s_memtime ticks per wave-instruction per SIMD\n");
}

int main(int argc, char **argv)
{
	setvbuf(stdout, NULL, _IONBF, 0);
	const char *only = argc > 1 ? argv[1] : NULL;
	printf("start\n");
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	int ncu = p.multiProcessorCount;
	printf("%s, %d CUs; s_memtime ticks (100 MHz realtime? see below)\n", p.gcnArchName, ncu);
	unsigned long long *d; hipMalloc(&d, 8 * (ncu * 32 + 1));
#define R(n) if(only == NULL || strcmp(only, #n) == 0) run<16>(#n, k_##n, d, ncu)
	R(add_f32); R(mul_f32); R(fma_f32); R(add_u32); R(and_b32); R(lshl); R(bfe); R(mov); R(cndmask); R(cndmask64);
	R(mad_u64); R(pk_add_f32); R(pk_mul_f32); R(pk_fma_f32); R(min3_f32); R(med3_i32); R(xor_b32); R(sub_f32); R(lshl_add); R(pk_add_u16); R(dot2_u16); R(bitop3); R(add_dpp); R(cmp_u32); R(f64_fma);
	R(cmp); R(cmp_e64); R(max_f32); R(cvt); R(mul_lo); R(mul_hi); R(mad_u32); R(add3); R(rcp); R(sqrt); R(salu); R(salu64);
	R(cmpcnd); R(ifblock);
	R(or_b32); R(sub_u32); R(min_u32); R(cvt_i32); R(mul_u24); R(mad_i24); R(and_or); R(lshl_or); R(perm); R(cvt_pk_u8); R(lerp_u8);
	R(f64_mul); R(f64_add); R(cvt_f64_f32); R(lshr_b64); R(cvt_f32_f64); R(cvt_i32_f64);
	if(only == NULL || strcmp(only, "mix_vs") == 0) run<16>("mix_vs(8v+8s)", k_mix_vs, d, ncu);
	return 0;
}
