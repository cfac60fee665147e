// tools/ubench/valu.hip — MEASUREMENT TOOL, not part of the product library.
// Issue cost (cycles per wave64 instruction per SIMD) of the VALU / LDS instructions the MurmurHash3 and L2 kernels are made
// of, on the GPU this runs on.  bench.py's roofline.int_ops ceiling quotes these numbers (profiles/r02_ubench_valu.txt).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/valu tools/ubench/valu.hip ; tools/ubench/valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
enum Op { ADD, XOR, ALIGNBIT, LSHL, LSHL_ADD, ADD64, MUL_LO, MUL_HI, MAD_U64_U32, MUL_U24, MAD_U24, CNDMASK, BFE, PERM, MIN_U32, LSHL64, DS_READ_U8, DS_READ_B32, DS_WRITE_B8, DS_READ_U16, CND_E64, CND_VCCSET, CMP_VCC, CMP_E64, CMP_CND, BFI, ADD3, AND_OR, MAX_I32, SALU_AND64, SALU_ADD, MIX_VS, READLANE, CMP_CND_E64, SUBREV_ASHR, CND_E64_VCC, CMP1_CND7, CMP1_CND7_S, CMP1_CND7_E64VCC, CMP_ADD_CND, CND_VCC_SPACED, NOPS };
static const char *names[] = {"v_add_u32", "v_xor_b32", "v_alignbit_b32", "v_lshlrev_b32", "v_lshl_add_u32", "v_add_co_u32+v_addc_co_u32 (pair)", "v_mul_lo_u32", "v_mul_hi_u32",
                              "v_mad_u64_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_cndmask_b32", "v_bfe_u32", "v_perm_b32", "v_min_u32", "v_lshlrev_b64", "ds_read_u8", "ds_read_b32", "ds_write_b8", "ds_read_u16", "v_cndmask_b32_e64 (sgpr pair mask)", "v_cndmask_b32 vcc (vcc set by s_mov in loop)", "v_cmp_lt_u32 vcc", "v_cmp_lt_u32_e64 s[..]", "v_cmp vcc + v_cndmask vcc (pair=2 instr)", "v_bfi_b32", "v_add3_u32", "v_and_or_b32", "v_max_i32", "s_and_b64 (x8)", "s_add_i32 (x8)", "4 v_add_u32 + 4 s_add_i32 interleaved", "v_readlane_b32 (x8)", "v_cmp_e64 s + v_cndmask_e64 s (pair)", "v_sub_u32 + v_ashrrev_i32 (pair: mask in vgpr)", "v_cndmask_b32_e64 with vcc operand (stale vcc)", "1 v_cmp vcc + 7 v_cndmask_e32 vcc", "1 v_cmp_e64 s + 7 v_cndmask_e64 s", "1 v_cmp vcc + 7 v_cndmask_e64 vcc", "v_cmp vcc; v_add; v_cndmask vcc; v_add (x2)", "v_cndmask vcc, v_add, v_cndmask vcc, v_add ... (stale vcc)"};

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters, uint32_t seed)
{
  __shared__ uint32_t lds[256 * 16];
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77, a3 = a0 * 3, a4 = a0 + 5, a5 = a0 ^ 91, a6 = a0 + 13, a7 = a0 * 7;
  uint32_t b = seed | 1u, c = seed * 31u + 3u;
  uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;
  for (int i = threadIdx.x; i < 256 * 16; i += 256) lds[i] = i * seed;
  __syncthreads();
  uint32_t addr = (threadIdx.x * 4) & 0x3ffc;
  const uint64_t t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %8, 7\n v_alignbit_b32 %1, %1, %8, 7\n v_alignbit_b32 %2, %2, %8, 7\n v_alignbit_b32 %3, %3, %8, 7\n v_alignbit_b32 %4, %4, %8, 7\n v_alignbit_b32 %5, %5, %8, 7\n v_alignbit_b32 %6, %6, %8, 7\n v_alignbit_b32 %7, %7, %8, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == ADD64) asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc");
    if (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n v_bfe_u32 %4, %4, 3, 9\n v_bfe_u32 %5, %5, 3, 9\n v_bfe_u32 %6, %6, 3, 9\n v_bfe_u32 %7, %7, 3, 9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    if (OP == MIN_U32) asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == LSHL64) asm volatile("v_lshlrev_b64 %0, 5, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 5, %2\n v_lshlrev_b64 %3, 5, %3\n v_lshlrev_b64 %0, 5, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 5, %2\n v_lshlrev_b64 %3, 5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
    if (OP == DS_READ_U8) { asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:64\n ds_read_u8 %2, %8 offset:128\n ds_read_u8 %3, %8 offset:192\n ds_read_u8 %4, %8 offset:256\n ds_read_u8 %5, %8 offset:320\n ds_read_u8 %6, %8 offset:384\n ds_read_u8 %7, %8 offset:448\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr) : "memory"); }
    if (OP == DS_READ_U16) { asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:64\n ds_read_u16 %2, %8 offset:128\n ds_read_u16 %3, %8 offset:192\n ds_read_u16 %4, %8 offset:256\n ds_read_u16 %5, %8 offset:320\n ds_read_u16 %6, %8 offset:384\n ds_read_u16 %7, %8 offset:448\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr) : "memory"); }
    if (OP == DS_READ_B32) { asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr) : "memory"); }
    if (OP == DS_WRITE_B8) { asm volatile("ds_write_b8 %8, %0\n ds_write_b8 %8, %1 offset:64\n ds_write_b8 %8, %2 offset:128\n ds_write_b8 %8, %3 offset:192\n ds_write_b8 %8, %4 offset:256\n ds_write_b8 %8, %5 offset:320\n ds_write_b8 %8, %6 offset:384\n ds_write_b8 %8, %7 offset:448\n s_waitcnt lgkmcnt(0)" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(addr) : "memory"); }

    if (OP == CND_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");
    if (OP == CND_VCCSET) asm volatile("s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CMP_VCC) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CMP_E64) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cmp_lt_u32_e64 s[22:23], %1, %8\n v_cmp_lt_u32_e64 s[24:25], %2, %8\n v_cmp_lt_u32_e64 s[26:27], %3, %8\n v_cmp_lt_u32_e64 s[20:21], %4, %8\n v_cmp_lt_u32_e64 s[22:23], %5, %8\n v_cmp_lt_u32_e64 s[24:25], %6, %8\n v_cmp_lt_u32_e64 s[26:27], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (OP == CMP_CND) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CMP_CND_E64) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cmp_lt_u32_e64 s[22:23], %1, %8\n v_cmp_lt_u32_e64 s[24:25], %2, %8\n v_cmp_lt_u32_e64 s[26:27], %3, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cndmask_b32_e64 %2, %2, %8, s[24:25]\n v_cndmask_b32_e64 %3, %3, %8, s[26:27]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (OP == SUBREV_ASHR) asm volatile("v_sub_u32 %4, %0, %8\n v_ashrrev_i32 %0, 31, %4\n v_sub_u32 %5, %1, %8\n v_ashrrev_i32 %1, 31, %5\n v_sub_u32 %6, %2, %8\n v_ashrrev_i32 %2, 31, %6\n v_sub_u32 %7, %3, %8\n v_ashrrev_i32 %3, 31, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == BFI) asm volatile("v_bfi_b32 %0, %9, %8, %0\n v_bfi_b32 %1, %9, %8, %1\n v_bfi_b32 %2, %9, %8, %2\n v_bfi_b32 %3, %9, %8, %3\n v_bfi_b32 %4, %9, %8, %4\n v_bfi_b32 %5, %9, %8, %5\n v_bfi_b32 %6, %9, %8, %6\n v_bfi_b32 %7, %9, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    if (OP == MAX_I32) asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == SALU_AND64) asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");
    if (OP == SALU_ADD) asm volatile("s_add_i32 s20, s20, 3\n s_add_i32 s21, s21, 3\n s_add_i32 s22, s22, 3\n s_add_i32 s23, s23, 3\n s_add_i32 s24, s24, 3\n s_add_i32 s25, s25, 3\n s_add_i32 s26, s26, 3\n s_add_i32 s27, s27, 3" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
    if (OP == MIX_VS) asm volatile("v_add_u32 %0, %0, %4\n s_add_i32 s20, s20, 3\n v_add_u32 %1, %1, %4\n s_add_i32 s21, s21, 3\n v_add_u32 %2, %2, %4\n s_add_i32 s22, s22, 3\n v_add_u32 %3, %3, %4\n s_add_i32 s23, s23, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21", "s22", "s23", "scc");
    if (OP == READLANE) asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");

    if (OP == CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CMP1_CND7) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CMP1_CND7_S) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");
    if (OP == CMP1_CND7_E64VCC) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CMP_ADD_CND) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_add_u32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (OP == CND_VCC_SPACED) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_add_u32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
  }
  const uint64_t t1 = clock64();
  uint32_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3) ^ (uint32_t)((w0 ^ w1 ^ w2 ^ w3) >> 32);
  if (sink == 0x12345 && iters < 0) out[1] = sink;
  if ((threadIdx.x & 63) == 0) out[2 + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> void run(uint64_t *d, int cus, int occ, int iters, double clockGHz)
{
  const int blocks = cus * occ;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters / 8, 7u);      // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(2 + (size_t)blocks * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<uint64_t> t(h.begin() + 2, h.end());
  std::sort(t.begin(), t.end());
  const double medTicks = (double)t[t.size() / 2];
  const double instr = (double)iters * 8.0;                 // wave-instructions per wave (ADD64: 8 instructions = 4 pairs)
  // per-SIMD issue cost seen by one wave / by the SIMD (occ waves share it)
  printf("%-36s occ %d  ticks/instr/wave %7.2f  -> per SIMD %6.2f ticks/instr   wall %8.3f ms  (%.2f cyc/instr/SIMD at %.2f GHz)\n", names[OP], occ, medTicks / instr,
         medTicks / instr / occ, ms, ms * 1e-3 * clockGHz * 1e9 / (instr * occ), clockGHz);
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate / 1e6;
  printf("device %s, %d CUs, clockRate %.3f GHz; one block = 4 waves = one wave per SIMD; occ = waves per SIMD\n", p.name, cus, ghz);
  printf("(ticks = clock64() = s_memtime; 'cyc/instr/SIMD' from wall time assumes the nominal clock)\n");
  uint64_t *d; hipMalloc(&d, (2 + (size_t)cus * 8 * 4) * 8);
  const int iters = 20000;
  for (int occ : {2, 8}) {
    run<ADD>(d, cus, occ, iters, ghz); run<XOR>(d, cus, occ, iters, ghz); run<ALIGNBIT>(d, cus, occ, iters, ghz); run<LSHL>(d, cus, occ, iters, ghz);
    run<LSHL_ADD>(d, cus, occ, iters, ghz); run<ADD64>(d, cus, occ, iters, ghz); run<MUL_LO>(d, cus, occ, iters, ghz); run<MUL_HI>(d, cus, occ, iters, ghz);
    run<MAD_U64_U32>(d, cus, occ, iters, ghz); run<MUL_U24>(d, cus, occ, iters, ghz); run<MAD_U24>(d, cus, occ, iters, ghz); run<CNDMASK>(d, cus, occ, iters, ghz);
    run<BFE>(d, cus, occ, iters, ghz); run<PERM>(d, cus, occ, iters, ghz); run<MIN_U32>(d, cus, occ, iters, ghz); run<LSHL64>(d, cus, occ, iters, ghz);
    run<CND_E64>(d, cus, occ, iters, ghz); run<CND_VCCSET>(d, cus, occ, iters, ghz); run<CMP_VCC>(d, cus, occ, iters, ghz); run<CMP_E64>(d, cus, occ, iters, ghz);
    run<CMP_CND>(d, cus, occ, iters, ghz); run<CMP_CND_E64>(d, cus, occ, iters, ghz); run<SUBREV_ASHR>(d, cus, occ, iters, ghz); run<BFI>(d, cus, occ, iters, ghz); run<ADD3>(d, cus, occ, iters, ghz);
    run<AND_OR>(d, cus, occ, iters, ghz); run<MAX_I32>(d, cus, occ, iters, ghz); run<SALU_AND64>(d, cus, occ, iters, ghz); run<SALU_ADD>(d, cus, occ, iters, ghz); run<MIX_VS>(d, cus, occ, iters, ghz);
    run<READLANE>(d, cus, occ, iters, ghz);
    run<CND_E64_VCC>(d, cus, occ, iters, ghz); run<CMP1_CND7>(d, cus, occ, iters, ghz); run<CMP1_CND7_S>(d, cus, occ, iters, ghz); run<CMP1_CND7_E64VCC>(d, cus, occ, iters, ghz);
    run<CMP_ADD_CND>(d, cus, occ, iters, ghz); run<CND_VCC_SPACED>(d, cus, occ, iters, ghz);
    run<DS_READ_U8>(d, cus, occ, iters / 4, ghz); run<DS_READ_U16>(d, cus, occ, iters / 4, ghz); run<DS_READ_B32>(d, cus, occ, iters / 4, ghz); run<DS_WRITE_B8>(d, cus, occ, iters / 4, ghz);
    printf("\n");
  }
  return 0;
}
