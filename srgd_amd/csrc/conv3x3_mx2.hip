// 3x3 / stride 1 / pad 1 convolution in a TWO-MFMA split-operand arithmetic for gfx950 (MI355X) - a prototype: kernel ABI impl 14 / 15
// (include/srgd_hip_kernels.h) and the 3x3 layers below the tile's resolution of SRGD_PRECISION_F16MX2 (engine.hip: the blocks at
// the tile's own resolution stay on conv3x3_split.hip).  The op: Block.proj, reference model.py:246.
//
// Arithmetic (emulation: oracle/split_emulation.py mixed_split_conv2d(mode="f16mx2")):
//     x = x_hi + x_lo,  w * s = w_hi + w_lo                      (x_hi, w_hi, w_lo f16 as in conv3x3_split.hip; x_lo = x - x_hi in fp32;
//                                                                 s = the layer's power of two)
//     conv(x, w) ~= [ x_hi . w_hi  +  Q(x_lo) . Q(w_hi)  +  Q(x_hi) . Q(w_lo) ] / s
// where Q is OCP MX-fp8 (e4m3 elements, one E8M0 scale per 32 input channels - the engine's scale rule, common.hpp mx_quant8).  The
// leading term runs on v_mfma_f32_16x16x32_f16 (exact products); the two cross terms are 2^-11 of it and need only the ~2^-4 an
// e4m3 pair carries, so they run on v_mfma_scale_f32_16x16x128_f8f6f4 at twice the f16 rate: three products for the matrix time
// of two.  Priced on the CPU first (profiles/r6/split_numerics_cheaper_variants.jsonl, reference configs[0] fixture: 1.3e-4 max-abs at
// the final pixel, f16x3 5.2e-6, bar 1e-3); measured on the GPU against the reference's fixtures and per shape in
// profiles/r6/conv3x3_mx2_prototype.txt (384-438 TF algorithmic where conv3x3_split runs 332-378; matrix pipe 47 % busy - the kernel is
// bound by its LDS operand reads, not by the MFMAs it saved).
//
// Kernel = conv3x3_split.hip's (8 x 32 pixel patch x 128 channels per 512-thread workgroup, K walked in 32-channel chunks, the halo
// patch of a chunk staged once for all nine taps, fp32 input through VGPRs, weights by LDS-DMA, register-direct epilogue) with:
//   * staging: a thread's 8 channels are split into f16 hi / lo; hi goes to the 16-bit halo image as before; lo AND hi are
//     quantised to e4m3 over the pixel's 32-channel block (the four lanes of a quad hold it: mx_quant8_finite below) and written to four
//     16-byte-per-pixel planes [product p][half h] (p = 0: x_lo, p = 1: x_hi), scale bytes [pixel][p];
//   * one scaled MFMA covers TWO taps: its four 32-wide K blocks are (tap t, x_lo.w_hi), (tap t, x_hi.w_lo), (tap t+1, x_lo.w_hi),
//     (tap t+1, x_hi.w_lo).  With the operand map decoded for conv3x3_mxfp8.hip - lane (r, g) supplies K bytes [16g, 16g+16) and
//     [64+16g, 64+16g+16), and the scale of K block g - lane group g reads plane g of tap t and plane g of tap t+1 (a compile-time
//     offset apart), and the scale of (tap t + (g >> 1), product g & 1).  Taps pair as (0,1) (2,3) (4,5) (6,7) (8,-): five scaled MFMAs
//     per chunk and accumulator block where 4.5 would do;
//   * weight unit per (tap, chunk, n-tile) = 16.5 KB: f16 hi tile (8 KB, conv3x3_split's image) | four e4m3 planes (2 KB each:
//     [p][h], p = 0: Q(w_hi), p = 1: Q(w_lo)) | 512 scale bytes [p][wn][r16][J]; a 4-slot LDS-DMA ring = the tap pair being consumed
//     + the pair in flight; the K loop advances in PAIR steps (one barrier per pair: ~1,000 MFMA cycles per wave between barriers,
//     one pair-step of DMA lead);
//   * both MFMA kinds accumulate into the same 16 x 16 fp32 blocks (same D layout), issued as inline asm with the accumulator tied;
//   * GNIN instances apply the producer's GroupNorm + SiLU to the fp32 pieces ahead of the split (coefficients through an LDS slot).
// LDS: 2 x 46 KB of A (24 KB hi image + four 5.3 KB planes + scales) + 4 x 16.5 KB of B + 0.5 KB of GNIN coefficients = 158.5 KB - one
// workgroup per CU.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int PH = 8, PW = 32;
constexpr int HP = PH + 2, WP = PW + 2;        // halo patch: 10 x 34 = 340 pixels
constexpr int KC = 32;
constexpr int BN3 = 128;
constexpr int NT3 = 512;
constexpr int A_IMG = 24 * 1024;               // f16 hi halo image (1,536 16-byte pieces, 1,360 used)
constexpr int A_PLANE = HP * WP * 16;          // one e4m3 plane: 16 B per halo pixel (5,440)
constexpr int A_Q = A_IMG;                     // the four planes [p][h]
constexpr int A_SC = A_IMG + 4 * A_PLANE;      // pixel scales [P][p]
constexpr int A_BUF = A_SC + 768;              // 47,104
constexpr int B_HI = BN3 * KC * 2;             // 8 KiB f16 tile
constexpr int B_PLANE = BN3 * 16;              // 2 KiB
constexpr int B_Q = B_HI;                      // four planes [p][h]
constexpr int B_SC = B_HI + 4 * B_PLANE;       // 16,384: 512 scale bytes [p][wn][r16][J]
constexpr int B_UNIT = B_SC + 512;             // 16,896
constexpr int B_RING = 4;                      // two tap pairs: the one being consumed and the one in flight
constexpr int LDS_BYTES = 2 * A_BUF + B_RING * B_UNIT;   // 161,792
constexpr int COEF_BYTES = 2 * 256;            // GNIN: 32 scales | 32 shifts (fp32) of a channel chunk, double-buffered

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct Mx2Args {
  const float* in0; const float* in1; int C0, C1;
  int B, H, W;
  const void* w;          // pack_conv3x3_mx2
  const float* bias;
  float w_inv_scale;
  int Cout;
  float* out;
  float* gn_partial; int groups;
  const float* gn_in_a;   // GNIN: y = silu(a[b][c] * x + b[b][c]) applied to the input while it is staged ([B][Cin] fp32)
  int gn_in_b_off;        // byte offset of the shift array from gn_in_a (one allocation)
};

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BARRIER()                        \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

__device__ __forceinline__ int row_swz(int row) { return (row >> 1) & 3; }

// 8 fp32 -> f16 hi (packed, for the 16-bit image) and the hi / lo VALUES as floats for the e4m3 quantisation; lo = x - hi in fp32
// (not rounded to f16 first: at e4m3's 4 significand bits that rounding is invisible, and it costs three conversions per pair)
__device__ __forceinline__ void split8f(const u32x4& r0, const u32x4& r1, u32x4& hi, float (&yh)[8], float (&yl)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned ua = k < 2 ? r0[2 * k] : r1[2 * k - 4], ub = k < 2 ? r0[2 * k + 1] : r1[2 * k - 3];
    const float a = __builtin_amdgcn_fmed3f(__uint_as_float(ua), -65504.f, 65504.f);
    const float b = __builtin_amdgcn_fmed3f(__uint_as_float(ub), -65504.f, 65504.f);
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    hi[k] = __builtin_bit_cast(unsigned, h);
    yh[2 * k] = hf[0]; yh[2 * k + 1] = hf[1];
    yl[2 * k] = a - hf[0]; yl[2 * k + 1] = b - hf[1];
  }
}

// mx_quant8 (common.hpp) for FINITE inputs: the same scale rule and rounding, without the saturation and NaN pass-through it
// carries for the fp8 modes' activations (the rule never lets the block maximum exceed 448 after scaling, and these inputs are f16
// values or differences of them) - the staging of this kernel is VALU work next to the matrix pipe, and those were half of it
__device__ __forceinline__ uint2 mx_quant8_finite(const float (&y)[8], int* scale_byte) {
  float amax = fmaxf(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))),
                     fmaxf(fmaxf(fabsf(y[4]), fabsf(y[5])), fmaxf(fabsf(y[6]), fabsf(y[7]))));
  amax = fmaxf(amax, dpp_move<0xB1>(amax));      // quad_perm [1,0,3,2]
  amax = fmaxf(amax, dpp_move<0x4E>(amax));      // quad_perm [2,3,0,1]: the quad's 32 channels
  const unsigned bits = __float_as_uint(amax);
  const int bexp = (int)((bits >> 23) & 0xffu);
  const int over = ((bits & 0x7fffffu) > 0x600000u) ? 1 : 0;
  const int sb = max(bexp - 8 + over, 0);
  const float inv = __uint_as_float((unsigned)(254 - sb) << 23);      // 2^(127 - sb)
  unsigned w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * inv, y[1] * inv, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * inv, y[3] * inv, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[4] * inv, y[5] * inv, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[6] * inv, y[7] * inv, w1, true);
  *scale_byte = sb;
  return make_uint2(w0, w1);
}

template <bool STATS, bool GNIN>
__global__ __launch_bounds__(NT3, 2) void conv3x3_mx2_kernel(Mx2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, q16 = lane >> 4;

  // ---- tile coordinates (conv3x3_split.hip)
  const int n_tiles = p.Cout / BN3;
  const int tiles_x = p.W / PW, tiles_y = p.H / PH;
  const int m_tiles = p.B * tiles_y * tiles_x;
  const int nwg = m_tiles * n_tiles;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, rem = nwg & 7, x = wg & 7, k = wg >> 3;
    wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
  }
  const int nt = wg % n_tiles, mt = wg / n_tiles;
  const int b = mt / (tiles_y * tiles_x);
  const int trem = mt - b * tiles_y * tiles_x;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * PH, x0 = tx * PW;
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / KC;

  // ---- A staging (conv3x3_split.hip): thread t owns 16-byte pieces t, t + 512, t + 1024 of the hi image; piece g = halo pixel
  // g >> 2, stored chunk position g & 3 holding SOURCE octet (g & 3) ^ row_swz(pixel)
  int a_pix0, a_pix1, a_pix2;
#define K_A_DECL(J)                                                           \
  {                                                                           \
    const int g = tid + NT3 * J;                                              \
    const int P = g >> 2;                                                     \
    const int py = P / WP, px = P - py * WP;                                  \
    const int y = y0 + py - 1, x = x0 + px - 1;                               \
    const bool ok = P < HP * WP && y >= 0 && y < p.H && x >= 0 && x < p.W;    \
    a_pix##J = ok ? y * p.W + x : -1;                                         \
  }
  K_A_DECL(0) K_A_DECL(1) K_A_DECL(2)
#undef K_A_DECL
  const int a_sub = (tid & 3) ^ row_swz(tid >> 2);       // the source octet (8 channels) this thread stages
  const size_t img_elems0 = (size_t)p.H * p.W * p.C0, img_elems1 = (size_t)p.H * p.W * p.C1;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.in0 + (size_t)b * img_elems0), 0, (int)(img_elems0 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in1 ? p.in1 + (size_t)b * img_elems1 : p.in0), 0, p.in1 ? (int)(img_elems1 * 4) : 0, 0x00020000);
  const size_t w_tile_stride = (size_t)n_tiles * B_UNIT;
  const char* w_base = (const char*)p.w + (size_t)nt * B_UNIT;
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, (int)((size_t)(9 * CC - 1) * w_tile_stride + B_UNIT), 0x00020000);

  char* const sA0 = smem;
  char* const sB0 = smem + 2 * A_BUF;
  const int tid16 = tid * 16;                      // the weight DMAs' per-lane offset; lane * 4 and (after the loop) the lane id are derived from it

  u32x4 ra00, ra01, ra10, ra11, ra20, ra21;
  auto load_piece = [&](int cc, int a_pix, u32x4& lo16, u32x4& hi16) {
    const int c = cc * KC;
    const bool first = c < p.C0;
    const int Cs = first ? p.C0 : p.C1;
    const int coff = first ? c : c - p.C0;
    const int voff = a_pix >= 0 ? (a_pix * Cs + coff + a_sub * 8) * 4 : 0x7ffffff0;
    if (first) {
      lo16 = __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, 0, 0);
      hi16 = __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, 16, 0);
    } else {
      lo16 = __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, 0, 0);
      hi16 = __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, 16, 0);
    }
  };
  auto load_a = [&](int cc) {
    load_piece(cc, a_pix0, ra00, ra01);
    load_piece(cc, a_pix1, ra10, ra11);
    load_piece(cc, a_pix2, ra20, ra21);
  };
  // GNIN instances (conv3x3_split.hip): the PRODUCER's GroupNorm-apply + SiLU (reference Block.forward model.py:250-259 between two
  // convolutions) on the fp32 halo pieces ahead of the split; out-of-image pixels stay zero.  The chunk's 64 coefficients come in
  // through ONE 4-byte-per-lane LDS-DMA per wave (every wave issues it: identical bytes, identical vmcnt counts), as in
  // conv3x3_bf16.hip - registers for them do not exist here.
  char* const sCoef = smem + LDS_BYTES;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(GNIN ? p.gn_in_a + (size_t)b * Cin : p.in0), 0, GNIN ? p.gn_in_b_off + Cin * 4 : 0, 0x00020000);
  auto coef_dma = [&](int cc) {
    const int l4 = (tid16 >> 2) & 255;          // lane * 4, rebuilt from the DMA offset register: `lane` itself is not kept through the K loop
    const int voff = l4 < 128 ? l4 : p.gn_in_b_off + l4 - 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (lds_ptr)(sCoef + (cc & 1) * 256), 4, voff, cc * KC * 4, 0, 0);
  };
  auto act8 = [&](int cc, u32x4& r0, u32x4& r1) {
    const char* sc = sCoef + (cc & 1) * 256 + a_sub * 32;
    const f32x4 ga0 = *reinterpret_cast<const f32x4*>(sc), ga1 = *reinterpret_cast<const f32x4*>(sc + 16);
    const f32x4 gb0 = *reinterpret_cast<const f32x4*>(sc + 128), gb1 = *reinterpret_cast<const f32x4*>(sc + 144);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned u0 = r0[k], u1 = r1[k];       // (element copied out first: conv3x3_split.hip)
      const float t0 = __builtin_fmaf(ga0[k], __uint_as_float(u0), gb0[k]), t1 = __builtin_fmaf(ga1[k], __uint_as_float(u1), gb1[k]);
      r0[k] = __float_as_uint(t0 * __builtin_amdgcn_rcpf(1.0f + __expf(-t0)));
      r1[k] = __float_as_uint(t1 * __builtin_amdgcn_rcpf(1.0f + __expf(-t1)));
    }
  };
  // split, quantise, write: hi image piece; e4m3 of x_lo -> plane (0, h), of x_hi -> plane (1, h), h = octet >> 1, 8 bytes at
  // (octet & 1) * 8 of the pixel's 16; the quad's lane 0 writes the two scale bytes
  const int q_off = (a_sub >> 1) * A_PLANE + (a_sub & 1) * 8;
  auto store_piece = [&](int cc, int j, u32x4 r0, u32x4 r1) {
    if constexpr (GNIN) {
      if ((j == 0 ? a_pix0 : (j == 1 ? a_pix1 : a_pix2)) >= 0) act8(cc, r0, r1);
    }
    u32x4 hi;
    float yh[8], yl[8];
    split8f(r0, r1, hi, yh, yl);
    int sl, sh;
    const uint2 ql = mx_quant8_finite(yl, &sl);
    const uint2 qh = mx_quant8_finite(yh, &sh);
    char* A = sA0 + (cc & 1) * A_BUF;
    *reinterpret_cast<u32x4*>(A + (tid + NT3 * j) * 16) = hi;
    const int P = (tid >> 2) + (NT3 / 4) * j;
    if (j < 2 || P < HP * WP) {                   // (pieces 1,360 .. 1,535 of the hi image are padding; the planes have none)
      *reinterpret_cast<uint2*>(A + A_Q + q_off + P * 16) = ql;
      *reinterpret_cast<uint2*>(A + A_Q + 2 * A_PLANE + q_off + P * 16) = qh;
      if ((tid & 3) == 0) *reinterpret_cast<unsigned short*>(A + A_SC + P * 2) = (unsigned short)(sl | (sh << 8));
    }
  };
  // weight unit (tap, cc) -> its ring slot: 16 KB as two 1 KB pieces per wave + 512 scale bytes (waves of equal parity
  // copy the same 256: every wave issues the same three instructions)
  const int w_tap_stride = (int)(CC * w_tile_stride);
  // ring slot of tap t of a chunk of parity par: pair k = t >> 1 sits in slots 2 ((k ^ par) & 1) + (t & 1) - five pairs per chunk, so
  // the parity of the pair count carries over the chunk boundary (tap 8 of this chunk and taps 0, 1 of the next never share slots)
  auto slot_off = [&](int par, int tap) { return ((((tap >> 1) & 1) ^ par) * 2 + (tap & 1)) * B_UNIT; };
  auto issue_b = [&](int cc, int tap) {
    if (tap >= 9) { tap -= 9; cc += 1; }
    char* dst = sB0 + slot_off(cc & 1, tap);
    const int so = tap * w_tap_stride + cc * (int)w_tile_stride;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(dst + wave * 1024), 16, tid16, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(dst + 8192 + wave * 1024), 16, tid16, so + 8192, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(dst + B_SC + (wave & 1) * 256), 4, (tid16 >> 2) & 255, so + B_SC + (wave & 1) * 256, 0, 0);
  };

  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0,
        c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;

  // ---- operand addresses
  // f16 (conv3x3_split.hip): weight row n = wn*64 + J*16 + r16, chunk q16; halo pixel P = lp + Pc
  const int b_base = (wn * 64 + r16) * 64 + ((q16 ^ row_swz(r16)) << 4);
  const int lp = 2 * wm * WP + r16, lp8 = lp << 3, lp64 = lp * 64, q16s = q16 << 4;
  auto pc_of = [&](int tap, int i) {
    const int dy = tap / 3, dx = tap - dy * 3;
    return ((i >> 1) + dy) * WP + (i & 1) * 16 + dx;
  };
  auto a_addr = [&](int tap, int i) {
    const int Pc = pc_of(tap, i);
    return lp64 + (((lp8 + Pc * 8) & 0x30) ^ q16s) + Pc * 64;
  };
  // e4m3: lane group g = q16 reads plane g (= [p = g >> 1][h = g & 1]) of both operands
  const int a8_base = A_Q + q16 * A_PLANE + lp * 16;                 // + Pc * 16
  const int b8_base = B_Q + q16 * B_PLANE + (wn * 64 + r16) * 16;    // + J * 256
  // scales: lane group g supplies K block g = (tap t + (g >> 1), product g & 1)
  const int asc_base = A_SC + lp * 2 + (q16 & 1);                    // + Pc(tap) * 2
  const int bsc_base = B_SC + (q16 & 1) * 256 + (wn * 16 + r16) * 4; // dword of the four J bytes, in the unit of tap t + (g >> 1)
  const bool second_tap = (q16 >> 1) != 0;

  typedef u32x4 frag;
#define K_SB __builtin_amdgcn_sched_barrier(0)
#define K_F16MM(C_, WT_, PX_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C_) : "v"(WT_), "v"(PX_))
#define K_QMM_OPSEL_0 "op_sel_hi:[0,0,0]"
#define K_QMM_OPSEL_1 "op_sel:[1,0,0] op_sel_hi:[0,0,0]"
#define K_QMM_OPSEL_2 "op_sel_hi:[1,0,0]"
#define K_QMM_OPSEL_3 "op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#define K_QMM(C_, WT_, SW_, PX_, SP_, J_)                                                              \
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " K_QMM_OPSEL_##J_            \
               : "+v"(C_) : "v"(WT_), "v"(PX_), "v"(SW_), "v"(SP_))

  // One K-step.  PAIR: 0 = f16 MFMAs only (even taps 0..6: the scaled MFMA of the pair is issued with the odd tap), 1 = the scaled
  // MFMA covers taps (tap - 1, tap), 2 = tap 8 alone (K blocks 2 and 3 are zero).
  auto compute = [&](int cc, int tap) {
    const char* A = sA0 + (cc & 1) * A_BUF;
    const char* Bt = sB0 + slot_off(cc & 1, tap);
    const frag bh0 = *reinterpret_cast<const frag*>(Bt + b_base), bh1 = *reinterpret_cast<const frag*>(Bt + b_base + 1024),
               bh2 = *reinterpret_cast<const frag*>(Bt + b_base + 2048), bh3 = *reinterpret_cast<const frag*>(Bt + b_base + 3072);
    const bool pair = (tap & 1) != 0, single = tap == 8;
    if (!pair && !single) {
#define K_ROWF(I, C0_, C1_, C2_, C3_)                                                         \
  {                                                                                            \
    const frag ah = *reinterpret_cast<const frag*>(A + a_addr(tap, I));                        \
    K_F16MM(C0_, bh0, ah); K_F16MM(C1_, bh1, ah); K_F16MM(C2_, bh2, ah); K_F16MM(C3_, bh3, ah); \
  }
      K_ROWF(0, c00, c01, c02, c03)
      K_ROWF(1, c10, c11, c12, c13)
      K_ROWF(2, c20, c21, c22, c23)
      K_ROWF(3, c30, c31, c32, c33)
#undef K_ROWF
      return;
    }
    // scaled MFMA of taps (t0, t1) = (tap - 1, tap), or (8, -)
    const int t0 = single ? 8 : tap - 1;
    const char* Bt0 = sB0 + slot_off(cc & 1, t0);                  // unit of tap t0
    const char* Bt1 = Bt;                                          // unit of tap t1 = tap (single: unused)
    v8i w0, w1, w2, w3;
#define K_LOADW(J)                                                                            \
  {                                                                                            \
    const v4i f0 = *reinterpret_cast<const v4i*>(Bt0 + b8_base + J * 256);                     \
    const v4i f1 = single ? v4i{0, 0, 0, 0} : *reinterpret_cast<const v4i*>(Bt1 + b8_base + J * 256); \
    w##J = v8i{f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};                        \
  }
    K_LOADW(0) K_LOADW(1) K_LOADW(2) K_LOADW(3)
#undef K_LOADW
    // weight scales: K block g lives in the unit of tap t0 (g < 2) or t1 (g >= 2); tap 8 alone: blocks 2, 3 are zero data, any
    // finite scale will do - the unit of tap 8 again
    const int sbw = *reinterpret_cast<const int*>((second_tap && !single ? Bt1 : Bt0) + bsc_base);
#define K_ROWQ(I, C0_, C1_, C2_, C3_)                                                         \
  {                                                                                            \
    const frag ah = *reinterpret_cast<const frag*>(A + a_addr(tap, I));                        \
    const int Pc0 = pc_of(t0, I), Pc1 = single ? Pc0 : pc_of(tap, I);                          \
    const v4i f0 = *reinterpret_cast<const v4i*>(A + a8_base + Pc0 * 16);                      \
    const v4i f1 = single ? v4i{0, 0, 0, 0} : *reinterpret_cast<const v4i*>(A + a8_base + Pc1 * 16); \
    const v8i px = v8i{f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};                \
    const int sp = *reinterpret_cast<const unsigned char*>(A + asc_base + (second_tap ? Pc1 : Pc0) * 2); \
    K_F16MM(C0_, bh0, ah); K_F16MM(C1_, bh1, ah); K_F16MM(C2_, bh2, ah); K_F16MM(C3_, bh3, ah); \
    K_QMM(C0_, w0, sbw, px, sp, 0); K_QMM(C1_, w1, sbw, px, sp, 1);                            \
    K_QMM(C2_, w2, sbw, px, sp, 2); K_QMM(C3_, w3, sbw, px, sp, 3);                            \
  }
    K_ROWQ(0, c00, c01, c02, c03)
    K_SB;
    K_ROWQ(1, c10, c11, c12, c13)
    K_SB;
    K_ROWQ(2, c20, c21, c22, c23)
    K_SB;
    K_ROWQ(3, c30, c31, c32, c33)
#undef K_ROWQ
  };

  // ---- prologue: [coefficients of chunk 0,] B[0], A(0) through registers
  if constexpr (GNIN) {
    coef_dma(0);
    issue_b(0, 0);
    issue_b(0, 1);
    load_a(0);
    WAIT_VM(6);                                  // the coefficient slot and B[0] have landed (the six fp32 loads may still fly) ...
    BARRIER();                                   // ... in every wave: the slot is complete
  } else {
    issue_b(0, 0);
    issue_b(0, 1);
    load_a(0);
  }
  store_piece(0, 0, ra00, ra01);
  store_piece(0, 1, ra10, ra11);
  store_piece(0, 2, ra20, ra21);
  WAIT_VM(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BARRIER();

  // ---- main loop, in steps of a TAP PAIR (0,1) (2,3) (4,5) (6,7) (8): issue the weight units of the next pair into the two slots
  // the previous pair left (its readers are behind the barrier that ended the last step); pair 0 also issues the six fp32 loads of
  // chunk cc+1 behind them (the counted wait lets them fly); the pair's MFMAs; pairs 1..3 split one piece each into the other A
  // buffer; wait for the next pair's units; barrier.  One barrier per ~1,000 MFMA cycles and wave, one pair-step of DMA lead.
  for (int cc = 0; cc < CC; ++cc) {
    const bool more = cc + 1 < CC;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (GNIN && k == 0 && more) coef_dma(cc + 1);              // oldest request of the step: the counted wait below covers it
      if (k < 3) { issue_b(cc, 2 * k + 2); issue_b(cc, 2 * k + 3); }
      else if (k == 3) issue_b(cc, 8);
      else if (more) { issue_b(cc, 9); issue_b(cc, 10); }
      if (k == 0 && more) {
        __builtin_amdgcn_sched_barrier(0);
        load_a(cc + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      compute(cc, 2 * k);
      if (k < 4) compute(cc, 2 * k + 1);
      if (more) {
        if (k == 1) store_piece(cc + 1, 0, ra00, ra01);
        if (k == 2) store_piece(cc + 1, 1, ra10, ra11);
        if (k == 3) store_piece(cc + 1, 2, ra20, ra21);
      }
      if (k == 0 && more) WAIT_VM(6); else WAIT_VM(0);
      if (k == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (k < 4 || more) BARRIER();
    }
  }
#undef K_QMM
#undef K_QMM_OPSEL_0
#undef K_QMM_OPSEL_1
#undef K_QMM_OPSEL_2
#undef K_QMM_OPSEL_3
#undef K_F16MM
  // asm MFMAs: no compiler-inserted wait states ahead of the first VALU read of their results
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
  asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));

  // ------------------------------- epilogue (conv3x3_split.hip: register-direct, fp32, full-line stores) --------------------------
  int tidE = tid16;                                     // per-lane epilogue addresses from an opaque copy: not carried through the K loop
  asm volatile("" : "+v"(tidE));
  const int laneE = (tidE >> 4) & 63, r16E = laneE & 15, q16E = laneE >> 4;
  const int chw = nt * BN3 + wn * 64;
  const int chl = q16E * 16;
  f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0, bs2 = bs0, bs3 = bs0;
  if (p.bias) {
    const float* bp = p.bias + chw + chl;
    bs0 = *reinterpret_cast<const f32x4*>(bp);
    bs1 = *reinterpret_cast<const f32x4*>(bp + 4);
    bs2 = *reinterpret_cast<const f32x4*>(bp + 8);
    bs3 = *reinterpret_cast<const f32x4*>(bp + 12);
  }
  const u32x4 rso = make_raw_rsrc(p.out + ((size_t)(b * p.H + y0 + 2 * wm) * p.W + x0) * p.Cout + chw, (unsigned)(2 * p.W * p.Cout * 4));
  constexpr int STG_ROW = 272;
  static_assert(8 * 16 * STG_ROW <= A_BUF, "store staging fits the idle A buffer");
  char* const stg = smem + (CC & 1) * A_BUF + wave * (16 * STG_ROW);
  const int stg_w = r16E * STG_ROW + q16E * 64;
  const int stg_r = (laneE >> 4) * STG_ROW + (laneE & 15) * 16;
  const int line_off = (laneE >> 4) * p.Cout * 4 + (laneE & 15) * 16;
  const float ws = p.w_inv_scale;
  f32x4 s1v = {0.f, 0.f, 0.f, 0.f}, s2v = s1v;
  asm volatile("" : "+v"(bs0), "+v"(bs1), "+v"(bs2), "+v"(bs3));
#define K_EMIT(MI, C0_, C1_, C2_, C3_)                                                          \
  do {                                                                                             \
    const int so_ = ((((MI) >> 1) * p.W + ((MI) & 1) * 16) * p.Cout) * 4;                          \
    const f32x4 v0 = C0_ * ws + bs0, v1 = C1_ * ws + bs1, v2 = C2_ * ws + bs2, v3 = C3_ * ws + bs3; \
    if (STATS) {                                                                                   \
      s1v += (v0 + v1) + (v2 + v3);                                                                \
      s2v = __builtin_elementwise_fma(v0, v0, s2v);                                                \
      s2v = __builtin_elementwise_fma(v1, v1, s2v);                                                \
      s2v = __builtin_elementwise_fma(v2, v2, s2v);                                                \
      s2v = __builtin_elementwise_fma(v3, v3, s2v);                                                \
    }                                                                                              \
    *reinterpret_cast<f32x4*>(stg + stg_w) = v0;                                                   \
    *reinterpret_cast<f32x4*>(stg + stg_w + 16) = v1;                                              \
    *reinterpret_cast<f32x4*>(stg + stg_w + 32) = v2;                                              \
    *reinterpret_cast<f32x4*>(stg + stg_w + 48) = v3;                                              \
    asm volatile("" ::: "memory");                                                                 \
    const u32x4 w0_ = *reinterpret_cast<const u32x4*>(stg + stg_r);                                \
    const u32x4 w1_ = *reinterpret_cast<const u32x4*>(stg + stg_r + 4 * STG_ROW);                  \
    const u32x4 w2_ = *reinterpret_cast<const u32x4*>(stg + stg_r + 8 * STG_ROW);                  \
    const u32x4 w3_ = *reinterpret_cast<const u32x4*>(stg + stg_r + 12 * STG_ROW);                 \
    asm volatile("" ::: "memory");                                                                 \
    buffer_store16(w0_, rso, line_off, so_);                                                       \
    buffer_store16(w1_, rso, line_off, so_ + 4 * p.Cout * 4);                                      \
    buffer_store16(w2_, rso, line_off, so_ + 8 * p.Cout * 4);                                      \
    buffer_store16(w3_, rso, line_off, so_ + 12 * p.Cout * 4);                                     \
  } while (0)
  K_EMIT(0, c00, c01, c02, c03);
  K_EMIT(1, c10, c11, c12, c13);
  K_EMIT(2, c20, c21, c22, c23);
  K_EMIT(3, c30, c31, c32, c33);
#undef K_EMIT
  if (STATS) {
    const int cpg = p.Cout / p.groups;
    float a1 = row16_sum((s1v[0] + s1v[1]) + (s1v[2] + s1v[3]));
    float a2 = row16_sum((s2v[0] + s2v[1]) + (s2v[2] + s2v[3]));
    if (cpg >= 32) { a1 = xor16_sum(a1); a2 = xor16_sum(a2); }
    if (cpg >= 64) { a1 = xor32_sum(a1); a2 = xor32_sum(a2); }
    const int rows_per_group = cpg >= 64 ? 4 : cpg >> 4;
    if (r16E == 0 && (q16E & (rows_per_group - 1)) == 0) {
      const int tpg = cpg >= BN3 ? cpg / BN3 : 1;
      const int wpt = cpg >= BN3 ? 8 : 4;
      const int nslots = tiles_y * tiles_x * tpg * wpt;
      const int slot = (trem * tpg + (cpg >= BN3 ? nt % tpg : 0)) * wpt + (cpg >= BN3 ? wave : wm);
      const int g = cpg >= BN3 ? chw / cpg : (chw + chl) >> __builtin_ctz(cpg);
      float* dst = p.gn_partial + ((size_t)(b * p.groups + g) * nslots + slot) * 2;
      *reinterpret_cast<f32x2*>(dst) = f32x2{a1, a2};
    }
  }
}

float f16_bits_to_f32_mx2(unsigned short h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const int e = (h >> 10) & 31;
  const uint32_t m = h & 0x3ffu;
  float v;
  if (e == 0) v = std::ldexp((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = std::ldexp((float)(m | 0x400u), e - 25);
  uint32_t u;
  std::memcpy(&u, &v, 4);
  u |= sign;
  std::memcpy(&v, &u, 4);
  return v;
}

}  // namespace

static size_t conv3x3_mx2_packed_bytes(int Cin, int Cout) { return (size_t)9 * (Cin / KC) * (Cout / BN3) * B_UNIT; }

// OIHW fp32 -> [tap][cc][ntile][unit]; unit = f16 hi tile (conv3x3_split's image and row order) | e4m3 planes [p][h] of Q(w_hi) (p = 0)
// and Q(w_lo) (p = 1), 16 bytes per row and plane | scale bytes [p][wn][r16][J] (row n = 64 wn + 16 J + r16); blocks = the 32 input
// channels of the chunk, the engine's scale rule (mx_block_exponent)
void pack_conv3x3_mx2(const float* src_oihw, int Cin, int Cout, float scale, std::vector<unsigned char>& out) {
  const int CC = Cin / KC, NTL = Cout / BN3;
  out.assign(conv3x3_mx2_packed_bytes(Cin, Cout), 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned char* u = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * B_UNIT;
        unsigned short* hi_t = reinterpret_cast<unsigned short*>(u);
        for (int n = 0; n < BN3; ++n) {
          const int o = nt * BN3 + regepi_row_channel(n);
          float vh[KC], vl[KC];
          for (int k = 0; k < KC; ++k) {
            const float v = src_oihw[(((size_t)o * Cin + cc * KC + k) * 3 + tap / 3) * 3 + tap % 3] * scale;
            unsigned short hb, lb;
            split_halves_host(v, true, &hb, &lb);
            const int c = k >> 3, e = k & 7, cs = c ^ ((n >> 1) & 3);
            hi_t[n * KC + cs * 8 + e] = hb;
            vh[k] = f16_bits_to_f32_mx2(hb);
            vl[k] = f16_bits_to_f32_mx2(lb);
          }
          for (int pr = 0; pr < 2; ++pr) {
            const float* v = pr == 0 ? vh : vl;
            float amax = 0.f;
            for (int k = 0; k < KC; ++k) amax = std::max(amax, std::fabs(v[k]));
            const int ex = mx_block_exponent(amax);
            const float inv = std::ldexp(1.0f, -ex);
            for (int k = 0; k < KC; ++k)
              u[B_Q + (pr * 2 + (k >> 4)) * B_PLANE + n * 16 + (k & 15)] = e4m3_encode(v[k] * inv);
            u[B_SC + pr * 256 + ((n >> 6) * 16 + (n & 15)) * 4 + ((n >> 4) & 3)] = (unsigned char)(ex + 127);
          }
        }
      }
}

int conv3x3_mx2(const ConvArgs& a, const void* packed_w, float w_inv_scale, hipStream_t st, const float* gn_in_a, const float* gn_in_b) {
  if (!conv3x3_split_eligible(a)) SRGD_FAIL("conv3x3_mx2: shape not eligible");
  if (a.bias && ((size_t)a.bias & 15)) SRGD_FAIL("conv3x3_mx2: the bias array must be 16-byte aligned");
  if ((size_t)9 * ((a.C0 + a.C1) / KC) * (a.Cout / BN3) * B_UNIT >= (1ull << 31)) SRGD_FAIL("conv3x3_mx2: packed weights beyond 2 GiB");
  Mx2Args p;
  p.in0 = (const float*)a.in0; p.in1 = (const float*)a.in1; p.C0 = a.C0; p.C1 = a.C1;
  p.B = a.B; p.H = a.Hin; p.W = a.Win; p.w = packed_w; p.bias = a.bias; p.w_inv_scale = w_inv_scale; p.Cout = a.Cout;
  p.out = (float*)a.out; p.gn_partial = a.gn_partial; p.groups = a.groups;
  const bool gnin = gn_in_a != nullptr;
  if (gnin && (a.C1 != 0 || !gn_in_b || gn_in_b < gn_in_a || (size_t)((const char*)gn_in_b - (const char*)gn_in_a) > (1u << 30)))
    SRGD_FAIL("conv3x3_mx2: fused input GroupNorm needs one source and scale / shift arrays in one allocation");
  p.gn_in_a = gn_in_a;
  p.gn_in_b_off = gnin ? (int)((const char*)gn_in_b - (const char*)gn_in_a) : 0;
  const int grid = a.B * (a.Hin / PH) * (a.Win / PW) * (a.Cout / BN3);
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define K_SET(S_, G_) SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mx2_kernel<S_, G_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + COEF_BYTES));
    K_SET(true, false) K_SET(false, false) K_SET(true, true) K_SET(false, true)
#undef K_SET
    once.done();
  }
  const bool stats = a.gn_partial != nullptr;
#define K_GO(S_, G_) hipLaunchKernelGGL((conv3x3_mx2_kernel<S_, G_>), dim3(grid), dim3(NT3), LDS_BYTES + COEF_BYTES, st, p)
  if (stats && gnin) K_GO(true, true); else if (stats) K_GO(true, false); else if (gnin) K_GO(false, true); else K_GO(false, false);
#undef K_GO
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
