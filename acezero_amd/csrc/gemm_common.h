// gemm_common.h -- device helpers shared by the GEMM-shaped kernels (head_kernels.hip, encoder_kernels.hip):
// bf16 packing, LDS swizzles, LDS-DMA address-space typedefs, counted waits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Ablation bits of the argument structs (`dbg` fields; tools/ablate_rowgemm.hip, tools/loss_ablate.py, tools/enc_ablate.sh) are read only in
// the diagnostics build: in the product library they are the constant 0 and the branches they guard do not exist.
#ifdef ACEZ_DIAG
#define ACEZ_DBG(x) (x)
#else
#define ACEZ_DBG(x) 0
#endif

namespace acez {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even fp32 -> bf16 (same as torch .to(bfloat16)); NaN stays NaN.
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// two fp32 -> packed bf16x2, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950; the compiler emits
// it for the __bf16 conversion). Branch-free: the bit-twiddling f2bf above costs an exec-mask branch per value.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = pack2(a, b);
  r.y = pack2(c, d);
  return r;
}
__device__ __forceinline__ void unpack4(uint2 v, float* o) {
  o[0] = __uint_as_float(v.x << 16);
  o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16);
  o[3] = __uint_as_float(v.y & 0xffff0000u);
}


// [row][64] bf16 K-stage tiles: 16-byte chunk index XOR (row >> 1) & 7 -> the ds_read_b128 fragment reads of 32
// consecutive rows are conflict-free. Applied to the per-lane SOURCE chunk of the LDS-DMA and, identically, to the reads.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
// [rows][128] bf16 epilogue staging tiles: chunk index XOR row & 15 (conflict-free for the accumulator layout and for
// the row-wise copy)
__device__ __forceinline__ int st_off(int row, int col) { return row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7)); }

// Cross-lane sums with DPP (no LDS traffic, unlike __shfl_xor = ds_bpermute). row16_sum: sum over the 16 lanes of a DPP row (the
// lanes that share l >> 4), every lane of the row gets it. wave_sum63: sum over the wavefront, valid in lanes 48..63. Fixed
// orders -> deterministic.
#define ACEZ_DPP_ADD(v, ctrl, rowmask) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rowmask), 0xF, false)))
__device__ __forceinline__ float row16_sum(float v) {
  v = ACEZ_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  v = ACEZ_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  v = ACEZ_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
  v = ACEZ_DPP_ADD(v, 0x140, 0xF);   // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum63(float v) {
  v = row16_sum(v);
  v = ACEZ_DPP_ADD(v, 0x142, 0xA);   // row_bcast:15 -> rows 1 and 3 add the sum of the row before
  v = ACEZ_DPP_ADD(v, 0x143, 0xC);   // row_bcast:31 -> rows 2 and 3 add the sum of rows 0..1
  return v;
}

// Maximum of NON-NEGATIVE values over the wavefront, valid in lane 63 (the same DPP sequence as wave_sum63; lanes a DPP step does not
// reach read 0, the identity of max on non-negative values). No LDS traffic: as six __shfl_xor (= ds_bpermute) rounds the fp16 mode's
// gradient-magnitude bookkeeping cost the input-gradient chain six serial LDS round trips per layer in every multiplier wave (+6 us per step).
#define ACEZ_DPP_MAX(v, ctrl, rowmask) fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rowmask), 0xF, false)))
__device__ __forceinline__ float wave_max63_nonneg(float v) {
  v = ACEZ_DPP_MAX(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  v = ACEZ_DPP_MAX(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  v = ACEZ_DPP_MAX(v, 0x141, 0xF);   // row_half_mirror
  v = ACEZ_DPP_MAX(v, 0x140, 0xF);   // row_mirror
  v = ACEZ_DPP_MAX(v, 0x142, 0xA);   // row_bcast:15 -> rows 1 and 3 fold the row before
  v = ACEZ_DPP_MAX(v, 0x143, 0xC);   // row_bcast:31 -> rows 2 and 3 fold rows 0..1
  return v;
}

// ---- the two 16-bit operand formats of the head (acez_train_config.compute_dtype): bf16 (default; BASELINE.json north_star) and fp16
// (what the reference's autocast uses, ace_trainer.py:517-518; three more mantissa bits, a narrower exponent: gradients are
// propagated scaled, head_api.hip grad_scale). Same MFMA rate on gfx950 (v_mfma_f32_16x16x32_{bf16,f16}), fp32 accumulation in both.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack2h(float a, float b) {   // v_cvt_pk_f16_f32: round to nearest even
  f16x2_t v;
  v[0] = (_Float16)a;
  v[1] = (_Float16)b;
  return __builtin_bit_cast(uint32_t, v);
}
struct EltBf16 {
  typedef bf16x8 frag;
  static constexpr bool is_f16 = false;
  static __device__ __forceinline__ float to_f(uint16_t h) { return bf2f(h); }
  static __device__ __forceinline__ uint16_t from_f(float f) { return f2bf(f); }
  static __device__ __forceinline__ uint32_t pk2(float a, float b) { return pack2(a, b); }
  static __device__ __forceinline__ uint2 pk4(float a, float b, float c, float d) { return pack4(a, b, c, d); }
  static __device__ __forceinline__ void un2(uint32_t v, float& a, float& b) { a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u); }
  static __device__ __forceinline__ void un4(uint2 v, float* o) { unpack4(v, o); }
  static __device__ __forceinline__ f32x4 mfma16(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct EltF16 {
  typedef f16x8 frag;
  static constexpr bool is_f16 = true;
  static __device__ __forceinline__ float to_f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ uint16_t from_f(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
  static __device__ __forceinline__ uint32_t pk2(float a, float b) { return pack2h(a, b); }
  static __device__ __forceinline__ uint2 pk4(float a, float b, float c, float d) { uint2 r; r.x = pack2h(a, b); r.y = pack2h(c, d); return r; }
  static __device__ __forceinline__ void un2(uint32_t v, float& a, float& b) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, v);
    a = (float)h[0]; b = (float)h[1];
  }
  static __device__ __forceinline__ void un4(uint2 v, float* o) { un2(v.x, o[0], o[1]); un2(v.y, o[2], o[3]); }
  static __device__ __forceinline__ f32x4 mfma16(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
#define ACEZ_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define ACEZ_VMCNT_C(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")   // n: integral constant expression

}  // namespace acez
