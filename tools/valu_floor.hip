// valu_floor.hip -- what does one wave64 vector instruction cost a gfx950 SIMD?
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_floor.hip -o tools/valu_floor && tools/valu_floor > profiles/r02_valu_floor.json
//
// Every test kernel issues ONE opcode (or a fixed mix) back to back on eight independent register chains, with
// k = 1, 2, 4, 5, 8 waves resident per SIMD (k blocks of 256 threads per CU; every block is resident, the grid never
// exceeds the chip).  All waves meet at a spin barrier (with a deadline, so a placement surprise cannot hang the GPU), then
// run their loop for a fixed number of shader cycles (s_memtime) and count what they issued.  Each wave records the SIMD
// it ran on (HW_ID / XCC_ID); the host groups the waves by SIMD and reports
//      cycles per wave-instruction = window / (instructions issued by all waves of that SIMD in the window)
// as the median over the SIMDs that held exactly k waves.  That is the issue cost the trace kernel's instruction
// stream pays: DESIGN.md section 3.1 prices the kernel against these numbers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

struct Stamp {
    unsigned long long t0, t1;  // s_memtime around the counted loop
    unsigned long long r0, r1;  // s_memrealtime (100 MHz)
    uint32_t hw_id, xcc_id;
    uint32_t rounds;            // passes of 4 x BODY the wave completed inside the window
    uint32_t late;              // 1: the start barrier ran into its deadline
};

#define R8(x) x x x x x x x x

// %0..%7 double chains, %8..%15 32-bit chains, %16/%17 double sources, %18 u32/f32 VGPR source, %19 SGPR source
#define OPERANDS                                                                                                            \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), \
      "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)                                                                                \
    : "v"(b), "v"(c), "v"(w), "s"(sw)                                                                                       \
    : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"

// "ins chain, chain tail" on the eight double / 32-bit chains
#define EIGHT_D(ins, tail) \
    ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" ins " %4, %4" tail "\n" ins " %5, %5" tail "\n" ins " %6, %6" tail "\n" ins " %7, %7" tail "\n"
#define EIGHT_U(ins, tail) \
    ins " %8, %8" tail "\n" ins " %9, %9" tail "\n" ins " %10, %10" tail "\n" ins " %11, %11" tail "\n" ins " %12, %12" tail "\n" ins " %13, %13" tail "\n" ins " %14, %14" tail "\n" ins " %15, %15" tail "\n"
// "ins chain tail" (one-operand forms: the chain is written, sources come from the tail)
#define EIGHT_UW(ins, tail) \
    ins " %8" tail "\n" ins " %9" tail "\n" ins " %10" tail "\n" ins " %11" tail "\n" ins " %12" tail "\n" ins " %13" tail "\n" ins " %14" tail "\n" ins " %15" tail "\n"
#define EIGHT_DW(ins, tail) \
    ins " %0" tail "\n" ins " %1" tail "\n" ins " %2" tail "\n" ins " %3" tail "\n" ins " %4" tail "\n" ins " %5" tail "\n" ins " %6" tail "\n" ins " %7" tail "\n"
// compares into eight SGPR pairs
#define EIGHT_CMP_D(ins, tail) \
    ins " s[40:41], %0" tail "\n" ins " s[42:43], %1" tail "\n" ins " s[44:45], %2" tail "\n" ins " s[46:47], %3" tail "\n" ins " s[48:49], %4" tail "\n" ins " s[50:51], %5" tail "\n" ins " s[52:53], %6" tail "\n" ins " s[54:55], %7" tail "\n"
#define EIGHT_CMP_U(ins, tail) \
    ins " s[40:41], %8" tail "\n" ins " s[42:43], %9" tail "\n" ins " s[44:45], %10" tail "\n" ins " s[46:47], %11" tail "\n" ins " s[48:49], %12" tail "\n" ins " s[50:51], %13" tail "\n" ins " s[52:53], %14" tail "\n" ins " s[54:55], %15" tail "\n"

// partial waves: does the SIMD skip the 16-lane passes that have no active lane?  (s[54:55] keeps the mask; the bodies used here leave it alone)
#define EXEC_SET(lo, hi) "s_mov_b64 s[54:55], exec\ns_mov_b32 exec_lo, " lo "\ns_mov_b32 exec_hi, " hi "\n"
#define EXEC_RESTORE "s_mov_b64 exec, s[54:55]\n"

struct OpDesc { const char *name; int per_body; };

// X(id, name, instructions per BODY, asm text of one BODY)
#define OPS(X)                                                                                                                  \
    X(FMA_F64, "v_fma_f64", 64, R8(EIGHT_D("v_fma_f64", ", %16, %17")))                                                       \
    X(MUL_F64, "v_mul_f64", 64, R8(EIGHT_D("v_mul_f64", ", %16")))                                                            \
    X(ADD_F64, "v_add_f64", 64, R8(EIGHT_D("v_add_f64", ", %16")))                                                            \
    X(MAX_F64, "v_max_f64", 64, R8(EIGHT_D("v_max_f64", ", %16")))                                                            \
    X(CMP_F64, "v_cmp_lt_f64 (sgpr pair dst)", 64, R8(EIGHT_CMP_D("v_cmp_lt_f64", ", %16")))                                   \
    X(CMP_F64_VCC, "v_cmp_lt_f64 (vcc dst)", 64, R8(EIGHT_DW("v_cmp_lt_f64 vcc,", ", %16")))                                  \
    X(CLASS_F64, "v_cmp_class_f64", 64, R8(EIGHT_CMP_D("v_cmp_class_f64", ", %18")))                                          \
    X(LDEXP_F64, "v_ldexp_f64", 64, R8(EIGHT_D("v_ldexp_f64", ", %18")))                                                      \
    X(RCP_F64, "v_rcp_f64", 64, R8(EIGHT_D("v_rcp_f64", "")))                                                                 \
    X(RSQ_F64, "v_rsq_f64", 64, R8(EIGHT_D("v_rsq_f64", "")))                                                                 \
    X(SQRT_F64, "v_sqrt_f64", 64, R8(EIGHT_D("v_sqrt_f64", "")))                                                              \
    X(DIV_SCALE_F64, "v_div_scale_f64", 64,                                                                                   \
      R8("v_div_scale_f64 %0, vcc, %0, %16, %0\nv_div_scale_f64 %1, vcc, %1, %16, %1\nv_div_scale_f64 %2, vcc, %2, %16, %2\nv_div_scale_f64 %3, vcc, %3, %16, %3\n" \
         "v_div_scale_f64 %4, vcc, %4, %16, %4\nv_div_scale_f64 %5, vcc, %5, %16, %5\nv_div_scale_f64 %6, vcc, %6, %16, %6\nv_div_scale_f64 %7, vcc, %7, %16, %7\n")) \
    X(DIV_FMAS_F64, "v_div_fmas_f64", 64, R8(EIGHT_D("v_div_fmas_f64", ", %16, %17")))                                        \
    X(DIV_FIXUP_F64, "v_div_fixup_f64", 64, R8(EIGHT_D("v_div_fixup_f64", ", %16, %17")))                                     \
    X(CVT_F32_F64, "v_cvt_f32_f64", 64,                                                                                       \
      R8("v_cvt_f32_f64 %8, %0\nv_cvt_f32_f64 %9, %1\nv_cvt_f32_f64 %10, %2\nv_cvt_f32_f64 %11, %3\nv_cvt_f32_f64 %12, %4\nv_cvt_f32_f64 %13, %5\nv_cvt_f32_f64 %14, %6\nv_cvt_f32_f64 %15, %7\n")) \
    X(CVT_F64_U32, "v_cvt_f64_u32", 64,                                                                                       \
      R8("v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %9\nv_cvt_f64_u32 %2, %10\nv_cvt_f64_u32 %3, %11\nv_cvt_f64_u32 %4, %12\nv_cvt_f64_u32 %5, %13\nv_cvt_f64_u32 %6, %14\nv_cvt_f64_u32 %7, %15\n")) \
    X(MOV_B32, "v_mov_b32", 64, R8(EIGHT_UW("v_mov_b32", ", %18")))                                                           \
    X(MOV_B64, "v_mov_b64", 64, R8(EIGHT_DW("v_mov_b64", ", %16")))                                                           \
    X(CNDMASK_VCC, "v_cndmask_b32 (vcc, nobody writes it)", 64, R8(EIGHT_U("v_cndmask_b32", ", %18, vcc")))                    \
    X(CNDMASK_SGPR, "v_cndmask_b32 (sgpr pair mask)", 64, R8(EIGHT_U("v_cndmask_b32", ", %18, s[40:41]")))                    \
    X(CMP_CNDMASK, "v_cmp_lt_f32 vcc + v_cndmask_b32 (pair, cost per instruction)", 64,                                       \
      R8("v_cmp_lt_f32 vcc, %8, %18\nv_cndmask_b32 %9, %9, %18, vcc\nv_cmp_lt_f32 vcc, %10, %18\nv_cndmask_b32 %11, %11, %18, vcc\n" \
         "v_cmp_lt_f32 vcc, %12, %18\nv_cndmask_b32 %13, %13, %18, vcc\nv_cmp_lt_f32 vcc, %14, %18\nv_cndmask_b32 %15, %15, %18, vcc\n")) \
    X(CMP64_CNDMASK2, "v_cmp_lt_f64 vcc + 2 x v_cndmask_b32 (a double select, cost per instruction)", 72,                     \
      R8("v_cmp_lt_f64 vcc, %0, %16\nv_cndmask_b32 %8, %8, %18, vcc\nv_cndmask_b32 %9, %9, %18, vcc\nv_cmp_lt_f64 vcc, %1, %16\nv_cndmask_b32 %10, %10, %18, vcc\n" \
         "v_cndmask_b32 %11, %11, %18, vcc\nv_cmp_lt_f64 vcc, %2, %16\nv_cndmask_b32 %12, %12, %18, vcc\nv_cndmask_b32 %13, %13, %18, vcc\n")) \
    X(AND_B32, "v_and_b32", 64, R8(EIGHT_U("v_and_b32", ", %18")))                                                            \
    X(XOR_B32, "v_xor_b32", 64, R8(EIGHT_U("v_xor_b32", ", %18")))                                                            \
    X(ADD_U32, "v_add_u32", 64, R8(EIGHT_U("v_add_u32", ", %18")))                                                            \
    X(ADD_CO_U32, "v_add_co_u32 (vcc carry out)", 64, R8(EIGHT_UW("v_add_co_u32", ", vcc, %18, %18")))                        \
    X(LSHL_ADD_U64, "v_lshl_add_u64", 64, R8(EIGHT_D("v_lshl_add_u64", ", 0, %16")))                                          \
    X(LSHLREV_B32, "v_lshlrev_b32", 64, R8(EIGHT_UW("v_lshlrev_b32", ", 1, %18")))                                            \
    X(LSHL_B64, "v_lshlrev_b64", 64, R8(EIGHT_DW("v_lshlrev_b64", ", 1, %16")))                                               \
    X(LSHR_B64, "v_lshrrev_b64", 64, R8(EIGHT_DW("v_lshrrev_b64", ", 1, %16")))                                               \
    X(MUL_LO_U32, "v_mul_lo_u32", 64, R8(EIGHT_U("v_mul_lo_u32", ", %18")))                                                   \
    X(MUL_HI_U32, "v_mul_hi_u32", 64, R8(EIGHT_U("v_mul_hi_u32", ", %18")))                                                   \
    X(MAD_U64_U32, "v_mad_u64_u32", 64,                                                                                       \
      R8("v_mad_u64_u32 %0, vcc, %8, %18, %0\nv_mad_u64_u32 %1, vcc, %9, %18, %1\nv_mad_u64_u32 %2, vcc, %10, %18, %2\nv_mad_u64_u32 %3, vcc, %11, %18, %3\n" \
         "v_mad_u64_u32 %4, vcc, %12, %18, %4\nv_mad_u64_u32 %5, vcc, %13, %18, %5\nv_mad_u64_u32 %6, vcc, %14, %18, %6\nv_mad_u64_u32 %7, vcc, %15, %18, %7\n")) \
    X(FMA_F32, "v_fma_f32 (vgpr sources)", 64, R8(EIGHT_U("v_fma_f32", ", %18, %18")))                                        \
    X(FMAC_F32, "v_fmac_f32", 64, R8(EIGHT_UW("v_fmac_f32", ", %18, %18")))                                                   \
    X(FMA_F32_S, "v_fma_f32 (one sgpr source)", 64, R8(EIGHT_U("v_fma_f32", ", %19, %18")))                                   \
    X(ADD_F32, "v_add_f32", 64, R8(EIGHT_U("v_add_f32", ", %18")))                                                            \
    X(SUB_F32_S, "v_subrev_f32 (sgpr source)", 64, R8(EIGHT_UW("v_sub_f32", ", %19, %18")))                                   \
    X(MUL_F32, "v_mul_f32", 64, R8(EIGHT_U("v_mul_f32", ", %18")))                                                            \
    X(MAX_F32, "v_max_f32", 64, R8(EIGHT_U("v_max_f32", ", %18")))                                                            \
    X(MIN_F32, "v_min_f32", 64, R8(EIGHT_U("v_min_f32", ", %18")))                                                            \
    X(MAX3_F32, "v_max3_f32", 64, R8(EIGHT_U("v_max3_f32", ", %18, %18")))                                                    \
    X(MED3_F32, "v_med3_f32", 64, R8(EIGHT_U("v_med3_f32", ", %18, %18")))                                                    \
    X(CMP_F32, "v_cmp_lt_f32 (sgpr pair dst)", 64, R8(EIGHT_CMP_U("v_cmp_lt_f32", ", %18")))                                   \
    X(PK_FMA_F32, "v_pk_fma_f32", 64, R8(EIGHT_D("v_pk_fma_f32", ", %16, %17")))                                              \
    X(PK_MUL_F32, "v_pk_mul_f32", 64, R8(EIGHT_D("v_pk_mul_f32", ", %16")))                                                   \
    X(PK_ADD_F32, "v_pk_add_f32", 64, R8(EIGHT_D("v_pk_add_f32", ", %16")))                                                   \
    X(RCP_F32, "v_rcp_f32", 64, R8(EIGHT_U("v_rcp_f32", "")))                                                                 \
    X(PK_FMA_OPSEL, "v_pk_fma_f32, one source half splat by op_sel", 64, R8(EIGHT_D("v_pk_fma_f32", ", %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]"))) \
    X(PK_ADD_S, "v_pk_add_f32 (sgpr pair source)", 64, R8(EIGHT_D("v_pk_add_f32", ", s[40:41]")))                             \
    X(PK_FMA_S, "v_pk_fma_f32 (sgpr pair source)", 64, R8(EIGHT_D("v_pk_fma_f32", ", s[40:41], %17")))                        \
    X(PK_FMA_DEP, "v_pk_fma_f32, ONE dependent chain per wave", 64,                                                            \
      R8("v_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %0, %0, %16, %17\n" \
         "v_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %0, %0, %16, %17\nv_pk_fma_f32 %0, %0, %16, %17\n")) \
    X(PK_FMA_OPSEL_DEP, "v_pk_fma_f32 op_sel splat, ONE dependent chain per wave", 64,                                         \
      R8("v_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\nv_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" \
         "v_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\nv_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" \
         "v_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\nv_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" \
         "v_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\nv_pk_fma_f32 %0, %0, %16, %17 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n")) \
    X(FMA_F32_DEP, "v_fma_f32, ONE dependent chain per wave", 64,                                                              \
      R8("v_fma_f32 %8, %8, %18, %18\nv_fma_f32 %8, %8, %18, %18\nv_fma_f32 %8, %8, %18, %18\nv_fma_f32 %8, %8, %18, %18\n"     \
         "v_fma_f32 %8, %8, %18, %18\nv_fma_f32 %8, %8, %18, %18\nv_fma_f32 %8, %8, %18, %18\nv_fma_f32 %8, %8, %18, %18\n"))   \
    X(MUL_F32_DEP, "v_mul_f32, ONE dependent chain per wave", 64,                                                              \
      R8("v_mul_f32 %8, %8, %18\nv_mul_f32 %8, %8, %18\nv_mul_f32 %8, %8, %18\nv_mul_f32 %8, %8, %18\n"                         \
         "v_mul_f32 %8, %8, %18\nv_mul_f32 %8, %8, %18\nv_mul_f32 %8, %8, %18\nv_mul_f32 %8, %8, %18\n"))                       \
    X(FMA_F64_DEP, "v_fma_f64, ONE dependent chain per wave", 64,                                                              \
      R8("v_fma_f64 %0, %0, %16, %17\nv_fma_f64 %0, %0, %16, %17\nv_fma_f64 %0, %0, %16, %17\nv_fma_f64 %0, %0, %16, %17\n"     \
         "v_fma_f64 %0, %0, %16, %17\nv_fma_f64 %0, %0, %16, %17\nv_fma_f64 %0, %0, %16, %17\nv_fma_f64 %0, %0, %16, %17\n"))   \
    X(READLANE, "v_readlane_b32", 64,                                                                                         \
      R8("v_readlane_b32 s40, %8, 1\nv_readlane_b32 s41, %9, 2\nv_readlane_b32 s42, %10, 3\nv_readlane_b32 s43, %11, 4\n"      \
         "v_readlane_b32 s44, %12, 5\nv_readlane_b32 s45, %13, 6\nv_readlane_b32 s46, %14, 7\nv_readlane_b32 s47, %15, 8\n"))  \
    X(READFIRSTLANE, "v_readfirstlane_b32", 64,                                                                               \
      R8("v_readfirstlane_b32 s40, %8\nv_readfirstlane_b32 s41, %9\nv_readfirstlane_b32 s42, %10\nv_readfirstlane_b32 s43, %11\n" \
         "v_readfirstlane_b32 s44, %12\nv_readfirstlane_b32 s45, %13\nv_readfirstlane_b32 s46, %14\nv_readfirstlane_b32 s47, %15\n")) \
    X(WRITELANE, "v_writelane_b32", 64,                                                                                       \
      R8("v_writelane_b32 %8, s40, 1\nv_writelane_b32 %9, s41, 2\nv_writelane_b32 %10, s42, 3\nv_writelane_b32 %11, s43, 4\n"  \
         "v_writelane_b32 %12, s44, 5\nv_writelane_b32 %13, s45, 6\nv_writelane_b32 %14, s46, 7\nv_writelane_b32 %15, s47, 8\n")) \
    X(MOV_DPP, "v_mov_b32 dpp row_shr:1", 64, R8(EIGHT_UW("v_mov_b32_dpp", ", %18 row_shr:1 row_mask:0xf bank_mask:0xf")))     \
    X(DS_BPERMUTE, "ds_bpermute_b32", 64,                                                                                     \
      R8("ds_bpermute_b32 %9, %8, %18\nds_bpermute_b32 %10, %8, %18\nds_bpermute_b32 %11, %8, %18\nds_bpermute_b32 %12, %8, %18\n" \
         "ds_bpermute_b32 %13, %8, %18\nds_bpermute_b32 %14, %8, %18\nds_bpermute_b32 %15, %8, %18\nds_bpermute_b32 %9, %8, %18\n") "s_waitcnt lgkmcnt(0)\n") \
    X(DS_READ_B64, "ds_read_b64", 64,                                                                                         \
      R8("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:512\nds_read_b64 %2, %8 offset:1024\nds_read_b64 %3, %8 offset:1536\n" \
         "ds_read_b64 %4, %8 offset:2048\nds_read_b64 %5, %8 offset:2560\nds_read_b64 %6, %8 offset:3072\nds_read_b64 %7, %8 offset:3584\n") "s_waitcnt lgkmcnt(0)\n") \
    X(SALU, "s_add_u32", 64,                                                                                                  \
      R8("s_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\ns_add_u32 s44, s44, 1\ns_add_u32 s45, s45, 1\ns_add_u32 s46, s46, 1\ns_add_u32 s47, s47, 1\n")) \
    X(SALU_B64, "s_and_b64", 64,                                                                                              \
      R8("s_and_b64 s[40:41], s[40:41], s[42:43]\ns_or_b64 s[44:45], s[44:45], s[46:47]\ns_and_b64 s[48:49], s[48:49], s[50:51]\ns_or_b64 s[52:53], s[52:53], s[54:55]\n" \
         "s_and_b64 s[40:41], s[40:41], s[42:43]\ns_or_b64 s[44:45], s[44:45], s[46:47]\ns_and_b64 s[48:49], s[48:49], s[50:51]\ns_or_b64 s[52:53], s[52:53], s[54:55]\n")) \
    X(VALU_SALU, "v_mul_f64 + s_add_u32 alternating (cost per PAIR)", 32,                                                     \
      R8("v_mul_f64 %0, %0, %16\ns_add_u32 s40, s40, 1\nv_mul_f64 %1, %1, %16\ns_add_u32 s41, s41, 1\nv_mul_f64 %2, %2, %16\ns_add_u32 s42, s42, 1\nv_mul_f64 %3, %3, %16\ns_add_u32 s43, s43, 1\n")) \
    X(FMA_F64_LO16, "v_fma_f64, exec = lanes 0-15", 64, EXEC_SET("0xffff", "0") R8(EIGHT_D("v_fma_f64", ", %16, %17")) EXEC_RESTORE)        \
    X(FMA_F64_LO32, "v_fma_f64, exec = lanes 0-31", 64, EXEC_SET("0xffffffff", "0") R8(EIGHT_D("v_fma_f64", ", %16, %17")) EXEC_RESTORE)    \
    X(FMA_F64_SPREAD16, "v_fma_f64, exec = every fourth lane (16 lanes)", 64,                                                  \
      EXEC_SET("0x11111111", "0x11111111") R8(EIGHT_D("v_fma_f64", ", %16, %17")) EXEC_RESTORE)                               \
    X(FMA_F64_ONE, "v_fma_f64, exec = lane 0", 64, EXEC_SET("1", "0") R8(EIGHT_D("v_fma_f64", ", %16, %17")) EXEC_RESTORE)                \
    X(FMA_F32_LO16, "v_fma_f32 (vgpr sources), exec = lanes 0-15", 64, EXEC_SET("0xffff", "0") R8(EIGHT_U("v_fma_f32", ", %18, %18")) EXEC_RESTORE) \
    X(FMA_F32_LO32, "v_fma_f32 (vgpr sources), exec = lanes 0-31", 64, EXEC_SET("0xffffffff", "0") R8(EIGHT_U("v_fma_f32", ", %18, %18")) EXEC_RESTORE) \
    X(CNDMASK_LO16, "v_cndmask_b32 (sgpr pair mask), exec = lanes 0-15", 64, EXEC_SET("0xffff", "0") R8(EIGHT_U("v_cndmask_b32", ", %18, s[40:41]")) EXEC_RESTORE) \
    X(MIX_HALF, "mix: 4 f64 (mul add fma mul) : 4 full-rate (fma_f32 and add_u32 mul_f32)", 64,                                \
      R8("v_mul_f64 %0, %0, %16\nv_fma_f32 %8, %8, %18, %18\nv_add_f64 %1, %1, %16\nv_and_b32 %9, %9, %18\n"                   \
         "v_fma_f64 %2, %2, %16, %17\nv_add_u32 %11, %11, %18\nv_mul_f64 %3, %3, %16\nv_mul_f32 %14, %14, %18\n"))             \
    X(MIX_TRACE, "mix like the trace kernel: 5 f64 + 3 (cmp,cndmask,max) + 8 full-rate per 16", 128,                          \
      R8("v_mul_f64 %0, %0, %16\nv_fma_f32 %8, %8, %18, %18\nv_cndmask_b32 %9, %9, %18, vcc\nv_add_f64 %1, %1, %16\n"          \
         "v_fma_f32 %10, %10, %18, %18\nv_and_b32 %11, %11, %18\nv_cmp_lt_f32 s[40:41], %12, %18\nv_fma_f64 %2, %2, %16, %17\n") \
      R8("v_mov_b32 %13, %18\nv_add_u32 %14, %14, %18\nv_mul_f64 %3, %3, %16\nv_mov_b32 %15, %18\n"                             \
         "v_fma_f32 %8, %8, %18, %18\nv_add_f64 %4, %4, %16\nv_mul_f32 %9, %9, %18\nv_max_f32 %10, %10, %18\n"))

#define X_ENUM(id, name, n, text) OP_##id,
enum Op { OPS(X_ENUM) OP_COUNT };
#define X_DESC(id, name, n, text) {name, n},
static const OpDesc op_desc[OP_COUNT] = {OPS(X_DESC)};

template <int OP>
__global__ __launch_bounds__(256) void floor_kernel(Stamp *out, unsigned int *arrive, unsigned int expected, unsigned long long window,
                                                    double b, double c, uint32_t w, uint32_t sw, double *sink) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = b + i;
    __syncthreads();
    double a0 = b + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t u0 = (threadIdx.x & 63u) * 8u, u1 = w + 1, u2 = w + 2, u3 = w + 3, u4 = w + 4, u5 = w + 5, u6 = w + 6, u7 = w + 7;
    if (OP != OP_DS_READ_B64 && OP != OP_DS_BPERMUTE) u0 = w + threadIdx.x;
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    // start barrier over the whole grid, with a deadline of ~20 ms
    uint32_t late = 0;
    if ((threadIdx.x & 63) == 0) atomicAdd(arrive, 1u);
    {
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
            if (__builtin_amdgcn_s_memtime() - tb > 50000000ull) { late = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    uint32_t rounds = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1;
    do {
        for (int j = 0; j < 4; j++) {
            switch (OP) {
#define X_CASE(id, name, n, text) case OP_##id: asm volatile(text OPERANDS); break;
                OPS(X_CASE)
            }
        }
        rounds++;
        t1 = __builtin_amdgcn_s_memtime();
    } while (t1 - t0 < window);
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) out[wave] = Stamp{t0, t1, r0, r1, hw_id, xcc_id, rounds, late};
    // keep every chain alive
    const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + lds[threadIdx.x];
    if (s == 12345.678) sink[0] = s;
}

template <int OP>
static void launch(Stamp *d, unsigned int *arrive, int blocks, unsigned long long window, double *sink) {
    hipLaunchKernelGGL(floor_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, arrive, (unsigned)blocks * 4u, window, 1.0000001, 0.9999999,
                       0x3f800001u, 0x3f800001u, sink);
}
typedef void (*launch_fn)(Stamp *, unsigned int *, int, unsigned long long, double *);
template <int... I>
static std::vector<launch_fn> table(std::integer_sequence<int, I...>) { return {&launch<I>...}; }

int main(int argc, char **argv) {
    const unsigned long long window = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000000ull;  // shader cycles per test
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const auto fns = table(std::make_integer_sequence<int, OP_COUNT>{});
    const int ks[] = {1, 2, 4, 5, 6, 8};
    const char *only = argc > 2 ? argv[2] : nullptr;  // run the ops whose name contains this
    Stamp *d;
    double *sink;
    unsigned int *arrive;
    const int max_waves = ncu * 8 * 4;
    CHECK(hipMalloc(&d, sizeof(Stamp) * max_waves));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMalloc(&arrive, 4));
    std::vector<Stamp> h(max_waves);
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"window_cycles\": %llu, \"note\": \"cyc = shader cycles (s_memtime) a SIMD spends per wave64 "
           "instruction with k waves resident on it (all started together, counted over a fixed window), median over the SIMDs that held exactly k waves; "
           "clock_mhz from s_memrealtime\", \"results\": [\n", prop.name, prop.gcnArchName, ncu, window);
    bool first = true;
    for (int op = 0; op < OP_COUNT; op++) {
        if (only && !strstr(op_desc[op].name, only)) continue;
        for (int k : ks) {
            const int blocks = ncu * k;
            launch_fn f = fns[op];
            CHECK(hipMemset(arrive, 0, 4));
            f(d, arrive, blocks, window / 8, sink);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(arrive, 0, 4));
            f(d, arrive, blocks, window, sink);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), d, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
            std::map<uint32_t, std::vector<int>> by_simd;
            int late = 0;
            for (int wv = 0; wv < blocks * 4; wv++) {
                const uint32_t key = (h[wv].xcc_id & 0xf) << 16 | ((h[wv].hw_id >> 4) & 0x3) | (h[wv].hw_id & 0xff00);
                by_simd[key].push_back(wv);
                late += h[wv].late;
            }
            std::vector<double> cyc, mhz;
            for (auto &kv : by_simd) {
                if ((int)kv.second.size() != k) continue;
                unsigned long long s0 = ~0ull, e1 = 0;
                double instr = 0;
                for (int wv : kv.second) {
                    s0 = std::min(s0, h[wv].t0);
                    e1 = std::max(e1, h[wv].t1);
                    instr += (double)h[wv].rounds * 4.0 * op_desc[op].per_body;
                }
                cyc.push_back((double)(e1 - s0) / instr);
                const Stamp &w0 = h[kv.second[0]];
                mhz.push_back((double)(w0.t1 - w0.t0) / (double)(w0.r1 - w0.r0) * 100.0);
            }
            if (cyc.empty()) continue;
            std::sort(cyc.begin(), cyc.end());
            std::sort(mhz.begin(), mhz.end());
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_wave_instr\": %.3f, \"p10\": %.3f, \"p90\": %.3f, \"simds\": %zu, \"simds_seen\": %zu, "
                   "\"late_waves\": %d, \"clock_mhz\": %.0f}",
                   first ? "" : ",\n", op_desc[op].name, k, cyc[cyc.size() / 2], cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10], cyc.size(), by_simd.size(),
                   late, mhz[mhz.size() / 2]);
            first = false;
            fflush(stdout);
        }
    }
    printf("\n]}\n");
    return 0;
}
