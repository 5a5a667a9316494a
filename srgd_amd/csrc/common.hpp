// Shared device/host helpers for the gfx950 kernels of the Real-SRGD sampling path.
// Written for MI355X (CDNA4, wave64) only - no portability layer.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string>

namespace srgd {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- error plumbing (message retrievable through srgd_last_error) ---------------------
void set_error(const std::string& msg);
#define SRGD_FAIL(msg)                                  \
  do {                                                  \
    ::srgd::set_error(std::string(msg));                \
    return -1;                                          \
  } while (0)
#define SRGD_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess) {                                                        \
      ::srgd::set_error(std::string(#call) + ": " + hipGetErrorString(e__));       \
      return -1;                                                                    \
    }                                                                               \
  } while (0)
#define SRGD_TRY(call)          \
  do {                          \
    int r__ = (call);           \
    if (r__ != 0) return r__;   \
  } while (0)

// ---- element conversion ---------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// A 16-byte vector of activations: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

// SiLU. PRECISE selects the accurate expf (fp32 parity mode); the bf16 mode uses the fast exp.
template <bool PRECISE> __device__ __forceinline__ float silu(float x) {
  if (PRECISE) return x / (1.0f + expf(-x));
  // v_exp_f32 + v_rcp_f32 (1 ulp each; the result is rounded to bf16 by every caller).  Round 3: __frcp_rn compiled to the
  // IEEE division sequence (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup): 16 VALU instructions per SiLU instead of 6.
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane of the row: two quad permutes, then
// row_half_mirror and row_mirror (after the quad steps a quad is uniform, so the mirrors fetch "the other quad / half").
// VALU-only - 4 DPP moves instead of 4 ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);    // row_half_mirror
  v += dpp_move<0x140>(v);    // row_mirror
  return v;
}

// OCP MX-fp8 quantisation of 8 consecutive channels held by this lane; lanes (lane & ~3) .. (lane | 3) hold one 32-channel
// block and must all be active.  Returns the 8 e4m3 bytes; *scale_byte = E8M0 shared exponent of the block
// (floor(log2 max|y|) - 8 + 127).  Same arithmetic in every kernel that writes MX-fp8 (quant_mxfp8.hip and the fused
// epilogues), so a fused twin equals the separate quantisation pass bit for bit.
__device__ __forceinline__ uint2 mx_quant8(const float (&y)[8], int* scale_byte) {
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(y[j]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  // floor(log2 amax) = biased exponent - 127 for a normal float; zero / denormal blocks get the smallest scale.  Scale rule: the
  // smallest power of two for which the block maximum does not saturate - floor(log2 amax) - 8, one step up when the maximum's
  // mantissa exceeds 1.75 (it would land above 448).  The OCP conversion recipe stops at floor(log2 amax) - 8 and clamps such
  // maxima (up to 12 % error on the block's largest element); on BASELINE configs[4] this rule is worth +2.5 dB (34.1 -> 36.6).
  const int bexp = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  const int over = ((__float_as_uint(amax) & 0x7fffffu) > 0x600000u) ? 1 : 0;
  const int sb = min(max(bexp - 8 + over, 0), 254);
  const float inv = __uint_as_float((unsigned)(254 - sb) << 23);      // 2^(127 - sb)
  float t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // saturate, but let a NaN through (fmaxf / fminf return the other operand): it becomes the e4m3 NaN byte 0x7f and the
    // MFMA carries it on - a numerical blow-up stays visible in fp8 mode instead of turning into a finite -448
    const float u = y[j] * inv;
    t[j] = (u != u) ? u : fminf(fmaxf(u, -448.f), 448.f);
  }
  unsigned w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], w1, true);
  *scale_byte = sb;
  return make_uint2(w0, w1);
}
// the MX-fp8 twin of a bf16x8 chunk that is being stored: element offset `elem` (a multiple of 8) of a [.., C] tensor
__device__ __forceinline__ void mx_store_twin(const bf16x8& v, unsigned char* q, unsigned char* s, size_t elem, int lane_in_quad) {
  float y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) y[j] = (float)v[j];
  int sb;
  const uint2 w = mx_quant8(y, &sb);
  *reinterpret_cast<uint2*>(q + elem) = w;
  if (lane_in_quad == 0) s[elem >> 5] = (unsigned char)sb;
}

// ---- register-direct MFMA epilogues (conv3x3_bf16, conv3x3_mxfp8, conv1x1_epilogue.hpp) --------------------------------------
// The convolution kernels issue their 16x16 MFMAs with the WEIGHT fragment as the A operand and the PIXEL fragment as B, so that
// D holds, for lane (r16 = lane & 15, g = lane >> 4), rows 4 g .. 4 g + 3 of the weight block for pixel r16.  With four weight
// blocks J = 0..3 per wave (64 rows) the host stores the rows of a 128-row weight tile in the order
//     tile row 64 wn + 16 J + 4 g + e   <-   output channel 64 wn + 16 g + 4 J + e
// so that the sixteen accumulator registers (J, e) of a lane are the SIXTEEN CONSECUTIVE output channels 64 wn + 16 g + 4 J + e of
// its pixel: 32 bytes of bf16 = two 16-byte stores straight from the registers, an MX-fp8 32-channel block = the lane pair
// (g, g ^ 1) = lanes l and l ^ 16, a GroupNorm group of 16 channels = one lane.  No LDS transposition, no barrier.
static inline int regepi_row_channel(int n) {            // host: output channel (inside the 128-channel tile) stored in tile row n
  const int wn = n >> 6, J = (n >> 4) & 3, g = (n >> 2) & 3, e = n & 3;
  return wn * 64 + g * 16 + J * 4 + e;
}
// v (+ | max) the same register of lane ^ 16 / lane ^ 32, in every lane: one half-exchange of two copies (v_permlane16_swap swaps
// the odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper half with the lower half) and
// one VALU op - no ds_bpermute (that form goes through the LDS crossbar).  After the swap the first register holds the even
// row's (lower half's) value and the second the odd row's (upper half's) in BOTH partner lanes: sums are formed in the same
// order everywhere.  Inline asm: the builtins of this toolchain (ROCm 7.2 clang) return their first result twice.  The s_nop
// covers the VALU-write -> v_permlane-read hazard (2 wait states), which the compiler does not insert inside asm.
__device__ __forceinline__ void permlane16_swap2(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void permlane32_swap2(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float xor16_sum(float v) { float a = v, b = v; permlane16_swap2(a, b); return a + b; }
__device__ __forceinline__ float xor32_sum(float v) { float a = v, b = v; permlane32_swap2(a, b); return a + b; }
__device__ __forceinline__ float xor16_max(float v) { float a = v, b = v; permlane16_swap2(a, b); return fmaxf(a, b); }
// A 16-byte store from registers through a raw buffer descriptor (SGPR quad {base lo, base hi & 0xffff, bytes in range, 0x00020000}),
// per-lane byte offset + wave-uniform byte offset, followed by two wait states.  Why asm: compiled from the builtin, the next
// packed VALU write to the data registers may follow the store directly (LLVM's store-data hazard rule exempts stores whose
// soffset is an SGPR), and on gfx950 lanes 12-15 of every row then stored the NEW contents of the second data register
// (found by the conv3x3_bf16 kernel test, round 5).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT = false>
__device__ __forceinline__ void buffer_store16(const u32x4& data, const u32x4& rsrc, int voffset, int soffset) {
  if (NT) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(data), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(data), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ u32x4 make_raw_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)(size_t)base;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t));
}
// OCP MX-fp8 quantisation of the 16 consecutive channels a lane holds in the register-direct layout; the 32-channel block is
// shared with lane ^ 16 (all lanes active).  Same scale rule and arithmetic as mx_quant8: bit-identical to quant_mxfp8.
__device__ __forceinline__ u32x4 mx_quant16_pair(const float (&y)[16], int* scale_byte) {
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) amax = fmaxf(amax, fabsf(y[j]));
  amax = xor16_max(amax);
  const int bexp = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  const int over = ((__float_as_uint(amax) & 0x7fffffu) > 0x600000u) ? 1 : 0;
  const int sb = min(max(bexp - 8 + over, 0), 254);
  const float inv = __uint_as_float((unsigned)(254 - sb) << 23);      // 2^(127 - sb)
  float t[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float u = y[j] * inv;
    t[j] = (u != u) ? u : fminf(fmaxf(u, -448.f), 448.f);
  }
  u32x4 w = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned d = 0;
    d = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * k + 0], t[4 * k + 1], d, false);
    d = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * k + 2], t[4 * k + 3], d, true);
    w[k] = d;
  }
  *scale_byte = sb;
  return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Integer tuning knob from the environment (`dflt` when unset).  Call sites keep the result in a function-local
// `static const int`, whose initialisation C++11 makes thread-safe: engines driven from several host threads
//  read every knob exactly once, race-free.
static inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Kernel classes for the per-class timing that bench.py reads (srgd_profile_*).
enum KClass {
  KC_CONV = 0,     // generic implicit-GEMM convolutions (1x1, 2x2/s2, pixel-shuffle, fp32 mode, odd shapes)
  KC_CONV3,        // conv3x3_bf16_kernel: the 3x3 bf16 fast path (dominant kernel)
  KC_CONV1,        // conv1x1_bf16_kernel: pointwise bf16 streaming GEMM (res_conv + GN tail, to_qkv/to_out, resamplers)
  KC_INIT,         // 7x7 input convolution reading the canvases
  KC_GN,           // GroupNorm finalize + apply(+SiLU, +residual)
  KC_RMS,          // RMSNorm
  KC_LINATTN,      // linear attention
  KC_FULLATTN,     // softmax attention
  KC_FINAL,        // 1x1 output conv + CFG + DDPM update
  KC_CANVAS,       // canvas prepare / re-noise / finish / RNG
  KC_COND,         // conditioning MLPs
  KC_CONVQ,        // conv3x3_mxfp8_kernel: block-scaled MX-fp8 3x3 convolution (fp8 mode)
  KC_QUANT,        // bf16 -> MX-fp8 quantisation passes (fp8 mode)
  KC_CONV1Q,       // conv1x1_mxfp8_kernel: pointwise layers on the MX matrix cores (fp8 mode)
  KC_CONV3S,       // conv3x3_split_kernel: 3x3 convolutions in split-operand precision (f16x3 mode)
  KC_CONVS,        // conv_igemm_split_kernel: the other convolutions of the f16x3 mode
  KC_CONV1S,       // conv1x1_split_kernel: pointwise layers of the f16x3 mode (streaming GEMM, LDS-DMA ring)
  KC_COUNT
};

// One-time per-device setup of a kernel family (hipFuncSetAttribute for > 64 KiB of dynamic LDS), safe when several host threads
// drive engines of their own: `if (DeviceSetup once(flags); once.need) { ...setup... }` - the first caller on a device runs the
// block under a process-wide mutex and ends it with `once.done()`; only then is the flag published (release) when the block
// is left - a block abandoned half-way (SRGD_HIP returning on a failed hipFuncSetAttribute) leaves the flag clear, so the next
// call repeats the setup and reports the real error instead of failing at launch.  No other thread can launch the kernel
// before its attribute is set; afterwards the check is one acquire load.
struct DeviceSetup {
  bool need = false;
  explicit DeviceSetup(bool (&flags)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { need = true; return; }   // unknown device: redo the setup
    flag_ = &flags[dev];
    if (__atomic_load_n(flag_, __ATOMIC_ACQUIRE)) return;
    mutex().lock();
    locked_ = true;
    need = !__atomic_load_n(flag_, __ATOMIC_ACQUIRE);
  }
  void done() { done_ = true; }
  ~DeviceSetup() {
    if (need && done_ && flag_) __atomic_store_n(flag_, true, __ATOMIC_RELEASE);
    if (locked_) mutex().unlock();
  }
  DeviceSetup(const DeviceSetup&) = delete;
  DeviceSetup& operator=(const DeviceSetup&) = delete;

 private:
  static std::mutex& mutex() { static std::mutex m; return m; }
  bool* flag_ = nullptr;
  bool locked_ = false;
  bool done_ = false;
};

}  // namespace srgd
