// 3x3 / stride 1 / pad 1 convolution on the block-scaled MX matrix cores of gfx950 (MI355X):
//   v_mfma_scale_f32_16x16x128_f8f6f4 with OCP e4m3 operands (activations AND weights), one E8M0 scale per 32 K-elements,
//   fp32 accumulate, bf16 out.  This is the BASELINE configs[4] compute path ("fp8 (CDNA4 MFMA) conv ... weights"): the op
//   it accelerates is Block.proj (reference model.py:246) and the last-stage 3x3 resamplers (:647, :668).  The plain fp8
//   MFMAs run at the bf16 rate on this part; only the scaled K = 128 form doubles it (3965 vs 1785 TFLOP/s measured on
//   register operands, tools/probe_mxfp8.hip).
//
// Operand maps, decoded on the hardware (tools/probe_mxfp8_v2.hip; there is no ISA text for them in this image):
//   A: lane l = (row l & 15, g = l >> 4) supplies 32 bytes: K elements [16g, 16g+16) and [64+16g, 64+16g+16) of its row
//   B: the same for column l & 15;   D: lane l, register r -> row 4*(l >> 4) + r, column l & 15
//   scale: byte `opsel` of lane l's scale VGPR scales K block (l >> 4) (K elements [32 (l>>4), 32 (l>>4) + 32)) of row l & 15
// So with channels as K, a lane reads 16-byte chunks g and 4+g of a pixel's 128-byte chunk row, and scale byte g.
//
// Implicit GEMM like conv3x3_bf16.hip (halo patch staged once per channel chunk, all 9 taps read it at shifted addresses,
// all staging by LDS-DMA, counted vmcnt + raw barriers), re-tiled for 128-channel (= one MFMA K) chunks:
//   * workgroup = 256 threads = 4 waves (2 along M x 2 along N), output tile = 8 x 32 pixel patch (M = 256) x 128 channels,
//     wave tile 128 x 64 = 8 x 4 MFMA blocks (128 accumulator registers); <= 256 VGPRs, 79 KiB LDS -> two workgroups per CU.
//     (The first version used M = 128 tiles: at twice the bf16 MFMA rate every workgroup then re-streamed the weight tiles at
//     ~68 GB/s per CU, the L2 -> LDS ceiling of this part, and the 128-channel layers ran at 1.2x the bf16 kernel instead of
//     1.5x; M = 256 halves the weight bytes per FLOP.)
//   * operand addresses: halo pixel P = 136 wm + Pc + r16 with Pc a compile-time constant per (tap, fragment); 136 = 17 * 8,
//     so the swizzle term (P & 6) needs only (Pc & 7) and the lane, and Pc * 128 rides in the ds_read offset field
//   * per chunk: the (8+2) x (32+2) halo patch x 128 B (42.5 KiB) + its scale bytes (4 B / pixel); per K-step (tap, chunk) one
//     16 KiB weight tile + 512 B of weight scales, double-buffered; rows XOR-swizzled (chunk ^= row & 6): every ds_read_b128 of
//     the operand pattern is conflict-free at every tap shift (exhaustive search over the lane groups of ds_read_b128)
//   * the MFMAs take the WEIGHT fragment (and its scales) as the A operand and the PIXEL fragment as B (symmetric maps), so a lane
//     holds four consecutive weight rows of ONE pixel per block; with the tile rows stored in regepi_row_channel order
//     (common.hpp) that is 16 consecutive output channels per pixel block - the epilogue is register-direct as in
//     conv3x3_bf16.hip: + bias, GroupNorm partial sums (DPP + permlane swaps, one slot per wave), two 16-byte bf16 stores and,
//     for the MX-fp8 twin, one 16-byte e4m3 store + one scale byte per lane pair.  No barrier; the bf16 values take one wave-private
//     hop through the (dead) halo-patch area so that they leave as full 128-byte lines.
// Diagnostic build (tools/build_variant.py only): -DSRGD_MXFP8_STAMPS=1 adds per-phase s_memtime stamps.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"
#if SRGD_MXFP8_STAMPS
#include "stamps.hpp"
#endif

namespace srgd {
namespace {

#ifndef SRGD_MXFP8_STAMPS
#define SRGD_MXFP8_STAMPS 0
#endif
#ifndef SRGD_MXFP8_DIAG_GNVALU
#define SRGD_MXFP8_DIAG_GNVALU 0
#endif
constexpr bool QSTAMPS = SRGD_MXFP8_STAMPS != 0;
constexpr int QPH = 8, QPW = 32;                 // output patch
constexpr int QHP = QPH + 2, QWP = QPW + 2;      // halo patch: 10 x 34 = 340 pixels
constexpr int QKC = 128;                         // channels per chunk = K of one MFMA
constexpr int QBN = 128;
constexpr int NW = 4;                            // waves per workgroup (an 8-wave instance with 64 x 64 wave tiles measured 3-7 % slower: DESIGN 4.3)
constexpr int QA_PIECES = 11;                    // 1 KiB LDS-DMA pieces per wave and chunk
constexpr int QA_BYTES = 4 * QA_PIECES * 1024;   // 44 KiB (340 px * 128 B = 43,520 used)
constexpr int QAS_BYTES = 2048;                  // activation scales: 4 B per halo pixel (1,360 used), 2 dword pieces per wave
constexpr int QB_TILE = QBN * QKC;               // 16 KiB of e4m3 weights per K-step
constexpr int QB_BYTES = QB_TILE + 512;          // + 512 scale bytes ([wn][r16][g][J]): one 16.5 KiB unit per (tap, chunk, n-tile)
constexpr int QRING = 2;
constexpr int QLDS = QA_BYTES + QAS_BYTES + QRING * QB_BYTES;   // 80,896 B: two workgroups per CU (<= 81,920)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

// voffset: per-lane byte offset (VGPR); soffset: wave-uniform byte offset (SGPR) - keeping the uniform part out of the VGPRs
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset, int soffset = 0) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds_wave_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset, int soffset = 0) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds_wave_base, 4, voffset, soffset, 0, 0);
}

struct ConvQArgs {
  const unsigned char* q0; const unsigned char* s0; int C0;     // MX-fp8 source 0: [B,H,W,C0] e4m3, [B,H,W,C0/32] E8M0
  const unsigned char* q1; const unsigned char* s1; int C1;     // optional source 1 (channel concat)
  int B, H, W;
  const unsigned char* w;     // packed [tap][cc][ntile][16.5 KiB]
  const float* bias;
  int Cout;
  bf16* out;
  float* gn_partial; int groups;
  unsigned char* oq; unsigned char* os;   // optional MX-fp8 twin of the output (ConvArgs::out_q / out_s)
};

#if SRGD_MXFP8_STAMPS
__device__ unsigned long long g_convq_timeline[(size_t)STAMP_REC * STAMP_MAX_WAVES];      // stamps.hpp
#endif

// lane id from v_mbcnt, as volatile asm: never hoisted or CSE'd, so no VGPR carries it (or the thread id) across the K loop
__device__ __forceinline__ int lane_id_opaque() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

#define QWAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define QBARRIER()                       \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// 2 x 2 waves, wave tile 128 pixels x 64 channels (128 accumulators, <= 256 VGPRs), two waves per SIMD, two workgroups per CU.
template <bool STATS>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv3x3_mxfp8_kernel(ConvQArgs p) {
  constexpr int NT = NW * 64;               // threads
  constexpr int NWM = NW / 2;               // wave rows (along pixels); 2 wave columns (along channels)
  constexpr int RPW = QPH / NWM;            // patch rows per wave: 4
  constexpr int NM = RPW * 2;               // 16-pixel fragments per wave: 8
  constexpr int A_PIECES = (4 * QA_PIECES + NW - 1) / NW;      // 1 KiB halo-patch pieces per wave and chunk: 11
  static_assert(NM == 8 && NT == 256, "conv3x3_mxfp8: 4-wave shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sAs = smem + QA_BYTES;
  char* const sB0 = smem + QA_BYTES + QAS_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- tile coordinates (XCD-aware remap as in conv3x3_bf16.hip)
  const int n_tiles = p.Cout / QBN;
  const int tiles_x = p.W / QPW, tiles_y = p.H / QPH;
  const int m_tiles = p.B * tiles_y * tiles_x;
  const int nwg = m_tiles * n_tiles;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, rem = nwg & 7, x = wg & 7, k = wg >> 3;
    wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
  }
  const int nt = wg % n_tiles;
  const int mt = wg / n_tiles;
  const int b = mt / (tiles_y * tiles_x);
  const int trem = mt - b * tiles_y * tiles_x;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * QPH, x0 = tx * QPW;
  const int CC0 = p.C0 / QKC, CC = (p.C0 + p.C1) / QKC;

  // ---- A staging: 44 pieces of 1 KiB per chunk, wave w issues pieces w, w+4, ..., w+40; 16-byte chunk index in the LDS image
  // = piece*64 + lane -> pixel P = idx >> 3, stored position idx & 7 holds logical chunk (idx & 7) ^ (P & 6).
  // Per-lane state: global pixel offset of the first piece (or -1) is recomputed per piece from (py, px): cheap integer
  // arithmetic once per chunk, nothing kept in registers across the K loop except the tile origin.
  auto halo_pix = [&](int P) {                    // pixel offset y*W + x of halo position P, -1 outside the image / patch
    const int py = P / QWP, px = P - py * QWP;
    const int y = y0 + py - 1, x = x0 + px - 1;
    const bool ok = P < QHP * QWP && y >= 0 && y < p.H && x >= 0 && x < p.W;
    return ok ? y * p.W + x : -1;
  };
  const size_t npix = (size_t)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rq0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.q0 + (size_t)b * npix * p.C0), 0, (int)(npix * p.C0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.s0 + (size_t)b * npix * (p.C0 / 32)), 0, (int)(npix * (p.C0 / 32)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rq1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.q1 ? p.q1 + (size_t)b * npix * p.C1 : p.q0), 0, p.q1 ? (int)(npix * p.C1) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.s1 ? p.s1 + (size_t)b * npix * (p.C1 / 32) : p.s0), 0, p.s1 ? (int)(npix * (p.C1 / 32)) : 0, 0x00020000);
  const size_t w_step_stride = (size_t)n_tiles * QB_BYTES;             // bytes between consecutive (tap, cc) tiles
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.w + (size_t)nt * QB_BYTES), 0, (int)((size_t)(9 * CC - 1) * w_step_stride + QB_BYTES), 0x00020000);

  auto issue_a = [&](int cc) {                    // 11 + 2 DMA instructions per wave
    const bool first = cc < CC0;
    const int Cs = first ? p.C0 : p.C1;
    const int ccl = first ? cc : cc - CC0;
    // per-chunk recomputation of the 13 staging offsets from a freshly derived lane id (they are invariant across chunks, and
    // hoisted out of the K loop they cost 26 long-lived VGPRs; so would a copy of the lane id kept for this purpose alone)
    const int opq_a = lane_id_opaque();
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
      const int idx = (wave + NW * j) * 64 + opq_a;
      const int P = idx >> 3;
      const int pix = halo_pix(P);
      const int sub = (idx & 7) ^ (P & 6);
      const int voff = pix >= 0 ? pix * Cs + ccl * QKC + sub * 16 : 0x7ffffff0;
      char* dst = sA + (wave + NW * j) * 1024;
      if (first) dma16(rq0, dst, voff); else dma16(rq1, dst, voff);
    }
    const int Cs32 = Cs / 32;
#pragma unroll
    for (int k = 0; k < 8 / NW; ++k) {            // scale dwords of halo pixels (wave + NW k)*64 .. +63 (8 pieces cover 512 >= 340)
      const int pix = halo_pix((wave + NW * k) * 64 + opq_a);
      const int voff = pix >= 0 ? pix * Cs32 + ccl * 4 : 0x7ffffff0;
      if (first) dma4(rs0, sAs + (wave + NW * k) * 256, voff); else dma4(rs1, sAs + (wave + NW * k) * 256, voff);
    }
  };
  // weight unit (tap, cc) into ring slot `slot`.  `tap` is a compile-time constant at every call site (unrolled tap loop), so the
  // unit's offset is one scalar multiply-add (round 2 divided the runtime step index by 9: ~30 SALU instructions per step on
  // the issue path of a wave that should be feeding the matrix pipe).
  const int w_tap_stride = (int)(CC * w_step_stride);
  auto issue_b = [&](int cc, int tap, int slot) {
    const int base = tap * w_tap_stride + cc * (int)w_step_stride;
    char* dst = sB0 + slot * QB_BYTES;
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) dma16(rsw, dst + (wave + NW * j) * 1024, lane * 16, base + (wave + NW * j) * 1024);
    // the 512 scale bytes: waves 0 and 1 would do; waves 2 and 3 repeat their transfers (same bytes to the same place) so
    // that every wave issues the same five instructions and the tap loop stays branch-free
    dma4(rsw, dst + QB_TILE + (wave & 1) * 256, lane * 4, base + QB_TILE + (wave & 1) * 256);
  };

  // accumulators: [8 pixel blocks (patch row 4 wm + (i >> 1), x half i & 1)][4 channel blocks of 16]
  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0, c20 = 0, c21 = 0, c22 = 0, c23 = 0,
        c30 = 0, c31 = 0, c32 = 0, c33 = 0, c40 = 0, c41 = 0, c42 = 0, c43 = 0, c50 = 0, c51 = 0, c52 = 0, c53 = 0,
        c60 = 0, c61 = 0, c62 = 0, c63 = 0, c70 = 0, c71 = 0, c72 = 0, c73 = 0;

  // ---- operand addresses.  Row P of the halo patch / weight row n, logical chunks g and 4+g, swizzle chunk ^= row & 6.
  // A: P = 136 wm + Pc + r16 with Pc = (patch row + dy) * 34 + 16 (x half) + dx a compile-time constant per (tap, fragment).
  // 136 = 17 * 8, so (P & 6) depends only on (Pc & 7) and the lane: eight per-lane bases (one per value of Pc & 7) are
  // computed once, and every fragment read is base[Pc & 7] + an immediate - no address arithmetic inside the K loop
  // (the first version recomputed ~100 VALU instructions per tap next to 32 MFMAs).
  const int lanepix = RPW * wm * QWP + r16;
  const int apix = lanepix * 128;                                   // byte offset of the lane's pixel row (before Pc)
  const int r7 = lanepix & 7;                                       // P & 6 = ((Pc & 7) + (lanepix & 7)) & 6  (NW = 4: 136 wm vanishes)
  // the eight swizzled per-lane bases (one per value of Pc & 7) and their (address ^ 64) partners: 16 registers, and NO address
  // arithmetic in the K loop (round 2 recomputed them per fragment - 4-5 VALU instructions x 8 fragments per step - because the
  // kernel had no registers to spare; the tied MFMAs freed 54)
#define K_QABASE(K) const int ab##K = apix + ((g ^ (((K) + r7) & 6)) << 4), ac##K = ab##K ^ 64;
  K_QABASE(0) K_QABASE(1) K_QABASE(2) K_QABASE(3) K_QABASE(4) K_QABASE(5) K_QABASE(6) K_QABASE(7)
#undef K_QABASE
  const int asb = lanepix * 4 + g;                                  // scale byte of (pixel, channel block g)
  // B: n = 64 wn + 16 J + r16 -> n & 6 = r16 & 6
  const int bb = (wn * 64 + r16) * 128 + ((g ^ (r16 & 6)) << 4);
  // weight scales: the unit's 512 scale bytes are laid out [wn][r16][g][J] (pack_conv3x3_mxfp8), so ONE dword per lane holds the
  // scales of its four column blocks J for K block g, and the MFMA's opsel picks byte J: 1 ds_read_b32 and 1 VGPR per tap instead
  // of 4 ds_read_u8 and 4 VGPRs (the kernel lives at the 256-register budget)
  const int bsb = QB_TILE + ((wn * 16 + r16) * 4 + g) * 4;

  // One K-step: the 4 weight fragments (64 output channels of this wave) stay in registers, the 8 pixel fragments stream
  // through one at a time - 24 ds_read_b128 per 32 MFMAs, and 128 + 32 + 16 operand/accumulator registers.
  auto compute = [&](int tap, int slot) {
#if SRGD_MXFP8_DIAG_GNVALU
    // Pricing build (tools/build_variant.py -DSRGD_MXFP8_DIAG_GNVALU=1; results unchanged - the arithmetic runs on dummy registers):
    // the vector work a GroupNorm-apply + SiLU + MX quantisation of the staged halo patch would add to this kernel.  Per 128-channel
    // chunk a wave would rewrite 340 x 128 / 4 = 10,880 elements = 170 per lane; the bf16 kernel's transform costs 28 instructions
    // per 4 elements (8 of them v_exp / v_rcp), the 32-channel maximum + scale + e4m3 pack ~10 more per 4: ~1,600 per chunk and
    // lane = ~180 per tap, issued here as four independent chains of the same instruction mix.
    {
      float g0 = (float)tap, g1 = g0 + 1.f, g2 = g0 + 2.f, g3 = g0 + 3.f;
      asm volatile(".rept 6\n\t"
                   "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1\n\t"
                   "v_mul_f32 %0, 0xbfb8aa3b, %0\n\tv_mul_f32 %1, 0xbfb8aa3b, %1\n\tv_mul_f32 %2, 0xbfb8aa3b, %2\n\tv_mul_f32 %3, 0xbfb8aa3b, %3\n\t"
                   "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_add_f32 %2, 1.0, %2\n\tv_add_f32 %3, 1.0, %3\n\t"
                   "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                   "v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_mul_f32 %2, %2, %0\n\tv_mul_f32 %3, %3, %1\n\t"
                   "v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_and_b32 %1, 0xffff0000, %0\n\tv_lshlrev_b32 %3, 16, %2\n\t"
                   ".endr" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));
    }
#endif
    const char* Bt = sB0 + slot * QB_BYTES;
    const int dy = tap / 3, dx = tap - dy * 3;
    v8i b0, b1, b2, b3;
    const int sbw = *reinterpret_cast<const int*>(Bt + bsb);
#define K_QLOAD_B(J)                                                              \
    {                                                                                \
      const v4i lo = *reinterpret_cast<const v4i*>(Bt + bb + J * 2048);              \
      const v4i hi = *reinterpret_cast<const v4i*>(Bt + (bb ^ 64) + J * 2048);       \
      b##J = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};            \
    }
    K_QLOAD_B(0) K_QLOAD_B(1) K_QLOAD_B(2) K_QLOAD_B(3)
#undef K_QLOAD_B
#define K_QLOAD_A(I)                                                              \
    v8i a##I; int sa##I;                                                             \
    {                                                                                \
      const int Pc = ((I >> 1) + dy) * QWP + (I & 1) * 16 + dx;                      \
      const int k7 = Pc & 7;                                                         \
      const int o = k7 == 0 ? ab0 : k7 == 1 ? ab1 : k7 == 2 ? ab2 : k7 == 3 ? ab3 : k7 == 4 ? ab4 : k7 == 5 ? ab5 : k7 == 6 ? ab6 : ab7; \
      const int o2 = k7 == 0 ? ac0 : k7 == 1 ? ac1 : k7 == 2 ? ac2 : k7 == 3 ? ac3 : k7 == 4 ? ac4 : k7 == 5 ? ac5 : k7 == 6 ? ac6 : ac7; \
      const v4i lo = *reinterpret_cast<const v4i*>(sA + o + Pc * 128);               \
      const v4i hi = *reinterpret_cast<const v4i*>(sA + o2 + Pc * 128);              \
      a##I = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};            \
      sa##I = *reinterpret_cast<const unsigned char*>(sAs + asb + Pc * 4);           \
    }
    // The MFMA goes out as inline asm with the accumulator tied (D = C, "+v"): through the builtin hipcc picks a fresh
    // destination for every scaled MFMA (this LLVM has no tied form of v_mfma_scale), so the 128 accumulators migrate through the
    // register file during a tap and the allocator - at 256 registers - spills loop-carried addresses into the K loop (the
    // reload sat behind s_waitcnt vmcnt(0), i.e. behind the weight DMA just issued).  Volatile asm also keeps each tap's MFMAs
    // in their tap: left to itself hipcc SINKS the register-only chains of all nine taps below the chunk's last barrier.
    // Hazards the compiler no longer sees: the operand fragments are overwritten by ds_reads no sooner than four MFMAs
    // (>= 128 cycles) later, the accumulators are next touched by MFMAs with the same D (interlocked) or by the epilogue behind
    // its s_nop block, and the inputs come from LDS reads whose lgkmcnt waits the compiler still inserts.
    // Operand order: srcA = weight fragment B_ (scale dword sbw, byte J picked by op_sel / op_sel_hi index 0: bit 0 -> op_sel[0],
    // bit 1 -> op_sel_hi[0]), srcB = pixel fragment A_ (scale byte 0 of SA_): D[i][j], i = weight row, j = pixel.
#define K_QMM_OPSEL_0 "op_sel_hi:[0,0,0]"
#define K_QMM_OPSEL_1 "op_sel:[1,0,0] op_sel_hi:[0,0,0]"
#define K_QMM_OPSEL_2 "op_sel_hi:[1,0,0]"
#define K_QMM_OPSEL_3 "op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#define K_QMM(C_, A_, SA_, B_, J_)                                                                      \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " K_QMM_OPSEL_##J_            \
                 : "+v"(C_) : "v"(B_), "v"(A_), "v"(sbw), "v"(SA_))
#define K_QROW(I)                                                                 \
    K_QMM(c##I##0, a##I, sa##I, b0, 0); K_QMM(c##I##1, a##I, sa##I, b1, 1);    \
    K_QMM(c##I##2, a##I, sa##I, b2, 2); K_QMM(c##I##3, a##I, sa##I, b3, 3);
    // software pipeline over the pixel fragments, fenced for the scheduler (left alone it hoists all eight fragment loads to
    // the top of the step and spills ~130 registers into the loop): the loads of fragment i+1 are issued ahead of the 4 MFMAs
    // (128 cycles of matrix pipe) of fragment i; two fragments live at a time
    K_QLOAD_A(0)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(1)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(0)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(2)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(1)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(3)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(2)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(4)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(3)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(5)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(4)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(6)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(5)
    __builtin_amdgcn_sched_barrier(0);
    K_QLOAD_A(7)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(6)
    __builtin_amdgcn_sched_barrier(0);
    K_QROW(7)
#undef K_QROW
#undef K_QMM
#undef K_QMM_OPSEL_0
#undef K_QMM_OPSEL_1
#undef K_QMM_OPSEL_2
#undef K_QMM_OPSEL_3
#undef K_QLOAD_A
  };

  [[maybe_unused]] unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, r0 = 0;
  if constexpr (QSTAMPS) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // ---- prologue: A(0), B[0]
  issue_a(0);
  issue_b(0, 0, 0);
  QWAIT_VM(0);
  QBARRIER();

  if constexpr (QSTAMPS) t1 = __builtin_amdgcn_s_memtime();
  // ---- main loop.  Per K-step: issue B[s+1] into the other ring slot; compute(s) (32 MFMAs per wave, ~2,000 cycles with the
  // SIMD's second wave: plenty for a 16.5 KiB L2 hit to land); wait for it; barrier.  Every step issues one weight unit (the
  // last step re-fetches the final one into the slot nobody reads any more) so that the unrolled tap loop is branch-free:
  // with data-dependent branches around the DMAs hipcc tail-merges the MFMA blocks of different taps.  The halo patch is
  // single-buffered (two would not leave room for two workgroups per CU): at a chunk boundary the next patch is fetched after
  // the barrier that retires the last tap's reads, and the co-resident workgroup keeps the matrix pipe busy meanwhile.
  for (int cc = 0; cc < CC; ++cc) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int slot = (cc + tap) & 1;               // step s = cc * 9 + tap lives in ring slot s % 2
      if (tap < 8) issue_b(cc, tap + 1, slot ^ 1);
      else issue_b(min(cc + 1, CC - 1), cc + 1 < CC ? 0 : 8, slot ^ 1);   // (the last step re-fetches its own unit into the idle slot)
      compute(tap, slot);
      QWAIT_VM(0);
      QBARRIER();
    }
    if (cc + 1 < CC) {
      issue_a(cc + 1);                             // every wave passed the barrier above: the old patch is dead
      QWAIT_VM(0);
      QBARRIER();
    }
  }
  if constexpr (QSTAMPS) t2 = __builtin_amdgcn_s_memtime();
  // The MFMAs are inline asm (compute()): the compiler does not know that the accumulators were written by the matrix pipe and
  // inserts none of the wait states a VALU read of an XDL result needs (<= 18 for a 16-pass MFMA).  The accumulators are
  // threaded through these statements, so every epilogue read comes after >= 32 wait states behind the last MFMA.
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
  asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));
  asm volatile("" : "+v"(c40), "+v"(c41), "+v"(c42), "+v"(c43), "+v"(c50), "+v"(c51), "+v"(c52), "+v"(c53));
  asm volatile("" : "+v"(c60), "+v"(c61), "+v"(c62), "+v"(c63), "+v"(c70), "+v"(c71), "+v"(c72), "+v"(c73));

  // ------------------------------- epilogue (register-direct) --------------------------
  // Accumulator block (mi, J), register e of lane (r16, g) = pixel (patch row 4 wm + (mi >> 1), x = 16 (mi & 1) + r16), tile row
  // 64 wn + 16 J + 4 g + e = output channel 64 wn + 16 g + 4 J + e (pack_conv3x3_mxfp8: regepi_row_channel): 16 consecutive channels
  // per pixel block.  Per-lane addresses come from a lane id re-derived HERE (v_mbcnt, opaque to the optimiser): derived from
  // `tid` they are computed ahead of the K loop and carried through it - in registers this kernel does not have.
  const int laneE = lane_id_opaque(), r16E = laneE & 15, gE = laneE >> 4;
  const int chw = nt * QBN + wn * 64;                     // first output channel of the wave (uniform)
  const int chl = gE * 16;                                // the lane's 16-channel run
  f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0, bs2 = bs0, bs3 = bs0;
  if (p.bias) {
    const float* bp = p.bias + chw + chl;
    bs0 = *reinterpret_cast<const f32x4*>(bp);
    bs1 = *reinterpret_cast<const f32x4*>(bp + 4);
    bs2 = *reinterpret_cast<const f32x4*>(bp + 8);
    bs3 = *reinterpret_cast<const f32x4*>(bp + 12);
  }
  const size_t pix_base = (size_t)(b * p.H + y0 + RPW * wm) * p.W + x0;      // the wave's first pixel
  const u32x4 rso = make_raw_rsrc(p.out + pix_base * p.Cout + chw, (unsigned)(RPW * p.W * p.Cout * 2));
  const u32x4 rsq = make_raw_rsrc(p.oq ? p.oq + pix_base * p.Cout + chw : (unsigned char*)p.out, p.oq ? (unsigned)(RPW * p.W * p.Cout) : 0u);
  const int lane_elem = r16E * p.Cout + chl;              // element offset of the lane's run from the wave's base
  // bf16 stores leave as FULL 128-byte lines (the wave's 64 channels of a pixel): straight from the accumulators a lane owns 32 bytes
  // of a pixel and a store instruction touches 64 scattered 16-byte pieces - the vector memory path takes them one piece at a time,
  // and the pricing builds (profiles/r6/conv3x3_mxfp8_persistent_pricing.txt, knobs in profiles/r6/pricing_and_ab_knobs.patch) showed the stores costing
  // out_bytes / ~6 TB/s ON TOP of the arithmetic in every shape (128 -> 128 @256^2: 1,864 -> 2,627 TF without them).  Each 16-pixel
  // block goes through 16 staging rows of the wave (144-byte pitch: conflict-free 16-byte writes) in the halo-patch area, which
  // is dead behind the K loop's last barrier; LDS executes a wave's instructions in order, so no barrier - only compiler fences
  // (the optimiser's memory model is per thread: conv1x1_split.hip).
  constexpr int STG_ROW = 144;
  char* const stg = sA + wave * (16 * STG_ROW);
  const int stg_w = r16E * STG_ROW + gE * 32;             // the lane's 32 bytes of pixel r16
  const int stg_r = (laneE >> 3) * STG_ROW + (laneE & 7) * 16;      // read-back: pixel lane >> 3 (and + 8), 16-byte piece lane & 7
  const int line_off = (laneE >> 3) * p.Cout * 2 + (laneE & 7) * 16;
  f32x4 s1v = {0.f, 0.f, 0.f, 0.f}, s2v = s1v;            // GroupNorm sums, per register position
  asm volatile("" : "+v"(bs0), "+v"(bs1), "+v"(bs2), "+v"(bs3));      // (all bias loads are waited for ahead of the first asm store)
#define K_QEMIT(MI)                                                                             \
  do {                                                                                             \
    const int eo_ = (((MI) >> 1) * p.W + ((MI) & 1) * 16) * p.Cout;       /* elements from the wave's base */ \
    const f32x4 v0 = c##MI##0 + bs0, v1 = c##MI##1 + bs1, v2 = c##MI##2 + bs2, v3 = c##MI##3 + bs3; \
    if (STATS) {                                                                                   \
      s1v += (v0 + v1) + (v2 + v3);                                                                \
      s2v = __builtin_elementwise_fma(v0, v0, s2v);                                                \
      s2v = __builtin_elementwise_fma(v1, v1, s2v);                                                \
      s2v = __builtin_elementwise_fma(v2, v2, s2v);                                                \
      s2v = __builtin_elementwise_fma(v3, v3, s2v);                                                \
    }                                                                                              \
    const u32x4 lo_ = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])}; \
    const u32x4 hi_ = {pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3]), pack_bf16x2(v3[0], v3[1]), pack_bf16x2(v3[2], v3[3])}; \
    {                                                                                              \
      /* through the wave's staging rows: 16 pixels x 128 B in, 8 full 128-byte lines per store instruction out */ \
      *reinterpret_cast<u32x4*>(stg + stg_w) = lo_;                                                \
      *reinterpret_cast<u32x4*>(stg + stg_w + 16) = hi_;                                           \
      asm volatile("" ::: "memory");                                                               \
      const u32x4 w0_ = *reinterpret_cast<const u32x4*>(stg + stg_r);                              \
      const u32x4 w1_ = *reinterpret_cast<const u32x4*>(stg + stg_r + 8 * STG_ROW);                \
      asm volatile("" ::: "memory");                                                               \
      buffer_store16(w0_, rso, line_off, eo_ * 2);                                                 \
      buffer_store16(w1_, rso, line_off, eo_ * 2 + 8 * p.Cout * 2);                                \
    }                                                                                              \
    if (p.oq) {                                                                                    \
      /* MX-fp8 twin of the STORED bf16 values (identical to quant_mxfp8 of the output): the 32-channel block = lanes l, l ^ 16 */ \
      float y_[16];                                                                                \
      _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                           \
        y_[2 * k_] = __uint_as_float(lo_[k_] << 16);          y_[2 * k_ + 1] = __uint_as_float(lo_[k_] & 0xffff0000u);     \
        y_[8 + 2 * k_] = __uint_as_float(hi_[k_] << 16);      y_[8 + 2 * k_ + 1] = __uint_as_float(hi_[k_] & 0xffff0000u); \
      }                                                                                            \
      int sb_;                                                                                     \
      const u32x4 w_ = mx_quant16_pair(y_, &sb_);                                                  \
      buffer_store16(w_, rsq, lane_elem, eo_);                                                     \
      if ((gE & 1) == 0) p.os[((pix_base * p.Cout + chw) >> 5) + ((size_t)(eo_ + lane_elem) >> 5)] = (unsigned char)sb_; \
    }                                                                                              \
  } while (0)
  { K_QEMIT(0); K_QEMIT(1); K_QEMIT(2); K_QEMIT(3); K_QEMIT(4); K_QEMIT(5); K_QEMIT(6); K_QEMIT(7); }
#undef K_QEMIT
  if (STATS) {
    // per-(sample, group) sums of this wave's 128 pixels x 64 channels (conv3x3_bf16.hip): cpg 16 -> one group per lane row,
    // 32 -> row pairs, >= 64 -> the wave; one slot per contributing wave: the 2 waves (wm) of the group's column half, or all 4
    const int cpg = p.Cout / p.groups;                    // 16, 32, 64 or a multiple of 128
    float a1 = row16_sum((s1v[0] + s1v[1]) + (s1v[2] + s1v[3]));
    float a2 = row16_sum((s2v[0] + s2v[1]) + (s2v[2] + s2v[3]));
    if (cpg >= 32) { a1 = xor16_sum(a1); a2 = xor16_sum(a2); }
    if (cpg >= 64) { a1 = xor32_sum(a1); a2 = xor32_sum(a2); }
    const int rows_per_group = cpg >= 64 ? 4 : cpg >> 4;
    if (r16E == 0 && (gE & (rows_per_group - 1)) == 0) {
      const int tpg = cpg >= QBN ? cpg / QBN : 1;
      const int wpt = cpg >= QBN ? NW : NWM;
      const int nslots = tiles_y * tiles_x * tpg * wpt;
      const int slot = (trem * tpg + (cpg >= QBN ? nt % tpg : 0)) * wpt + (cpg >= QBN ? wave : wm);
      const int grp = cpg >= QBN ? chw / cpg : (chw + chl) >> __builtin_ctz(cpg);
      float* dst = p.gn_partial + ((size_t)(b * p.groups + grp) * nslots + slot) * 2;
      *reinterpret_cast<f32x2*>(dst) = f32x2{a1, a2};
    }
  }
  if constexpr (QSTAMPS) {
    t3 = __builtin_amdgcn_s_memtime();
#if SRGD_MXFP8_STAMPS
    if (laneE == 0) stamp_record(g_convq_timeline, blockIdx.x * NW + wave, r0, t1 - t0, t2 - t1, t3 - t2, 0);
#endif
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
// float -> OCP e4m3 "fn" byte, round-to-nearest-even, saturating at +-448 (no NaN produced for finite input)
unsigned char e4m3_encode(float x) {
  const float r = round_through_e4m3(x);          // engine.hip: the same rounding the bf16_w8 mode uses
  if (r != r) return 0x7f;
  const unsigned char sgn = std::signbit(r) ? 0x80 : 0;
  const float a = std::fabs(r);
  if (a == 0.f) return sgn;
  int ex;
  const float m = std::frexp(a, &ex);             // a = m * 2^ex, m in [0.5, 1)
  int e = ex - 1 + 7;                             // biased exponent of the leading bit
  if (e <= 0) return sgn | (unsigned char)std::lround(a * 512.0f);           // subnormal: multiples of 2^-9
  const int mant = (int)std::lround((m * 2.0f - 1.0f) * 8.0f);                // exact: r is representable
  return sgn | (unsigned char)((e << 3) | mant);
}

// E8M0 exponent of a block with maximum magnitude amax: floor(log2 amax) - emax(e4m3) with emax = 8 (the OCP MX recipe), one step
// up when the maximum would saturate (mantissa > 1.75 -> above 448 after scaling); the rule of mx_quant8 (common.hpp), clamped
int mx_block_exponent(float amax) {
  if (!(amax > 0.f)) return -127;
  int ex;
  const float m = std::frexp(amax, &ex);          // amax = m * 2^ex, m in [0.5, 1) -> floor(log2 amax) = ex - 1
  int e = ex - 1 - 8 + (m * 2.0f > 1.75f ? 1 : 0);
  return e < -127 ? -127 : (e > 127 ? 127 : e);
}

bool conv3x3_mxfp8_eligible(const ConvArgs& a) {
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.mode != CONV_PLAIN || a.residual || a.gn_res_src) return false;
  if (a.C0 % QKC || a.C1 % QKC || a.Cout % QBN || a.Cout != a.CoutPad) return false;
  if (a.Hin % QPH || a.Win % QPW) return false;
  if (a.gn_partial) {
    const int cpg = a.Cout / a.groups;
    if (a.Cout % a.groups) return false;
    if (!(cpg == 16 || cpg == 32 || cpg == 64 || cpg % QBN == 0)) return false;
  }
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.C0, a.C1) >= (1ull << 31)) return false;
  if ((size_t)a.Hin * a.Win * (size_t)a.Cout * 2 >= (1ull << 31)) return false;
  if ((size_t)9 * ((a.C0 + a.C1) / QKC) * (a.Cout / QBN) * QB_BYTES >= (1ull << 31)) return false;
  return true;
}

int conv3x3_mxfp8_stats_slots(const ConvArgs& a) {
  if (a.groups <= 0) return 0;
  const int cpg = a.Cout / a.groups;
  return (a.Hin / QPH) * (a.Win / QPW) * (cpg >= QBN ? (cpg / QBN) * NW : NW / 2);      // one slot per contributing wave
}

// OIHW fp32 -> [tap][cc][ntile][16.5 KiB]: 128 rows x 128 B of e4m3 (swizzled LDS image) + 512 E8M0 bytes [wn][r16][blk][J].
// Tile row n holds output channel regepi_row_channel(n) (common.hpp: the register-direct epilogue's row order).
// One scale per (output channel, tap, 32 input channels): w = q * 2^(byte - 127).
void pack_conv3x3_mxfp8(const float* src_oihw, int Cin, int Cout, std::vector<unsigned char>& out) {
  const int CC = Cin / QKC, NTL = Cout / QBN;
  out.assign((size_t)9 * CC * NTL * QB_BYTES, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned char* unit = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * QB_BYTES;
        for (int n = 0; n < QBN; ++n) {
          const int o = nt * QBN + regepi_row_channel(n);
          for (int blk = 0; blk < 4; ++blk) {
            float amax = 0.f;
            for (int e = 0; e < 32; ++e) {
              const int ci = cc * QKC + blk * 32 + e;
              amax = std::max(amax, std::fabs(src_oihw[(((size_t)o * Cin + ci) * 3 + tap / 3) * 3 + tap % 3]));
            }
            const int ex = mx_block_exponent(amax);
            // scale bytes as [wn = n / 64][r16 = n % 16][blk][J = (n / 16) % 4]: see the kernel's `bsb`
            unit[QB_TILE + (((n >> 6) * 16 + (n & 15)) * 4 + blk) * 4 + ((n >> 4) & 3)] = (unsigned char)(ex + 127);
            const float inv = std::ldexp(1.0f, -ex);
            for (int e = 0; e < 32; ++e) {
              const int k = blk * 32 + e;                      // channel inside the chunk = K index
              const int ci = cc * QKC + k;
              const float v = src_oihw[(((size_t)o * Cin + ci) * 3 + tap / 3) * 3 + tap % 3] * inv;
              const int chunk = (k >> 4) ^ (n & 6);            // stored 16-byte chunk position (row_lo of the kernel)
              unit[n * 128 + chunk * 16 + (k & 15)] = e4m3_encode(v);
            }
          }
        }
      }
}

int conv3x3_mxfp8(const ConvArgs& a, const void* q0, const void* s0, const void* q1, const void* s1, const void* packed_w,
                  hipStream_t st) {
  if (!conv3x3_mxfp8_eligible(a)) SRGD_FAIL("conv3x3_mxfp8: shape not eligible");
  if (!q0 || !s0 || (a.C1 && (!q1 || !s1))) SRGD_FAIL("conv3x3_mxfp8: missing quantised operand");
  ConvQArgs p;
  p.q0 = (const unsigned char*)q0; p.s0 = (const unsigned char*)s0; p.C0 = a.C0;
  p.q1 = a.C1 ? (const unsigned char*)q1 : nullptr; p.s1 = a.C1 ? (const unsigned char*)s1 : nullptr; p.C1 = a.C1;
  p.B = a.B; p.H = a.Hin; p.W = a.Win; p.w = (const unsigned char*)packed_w; p.bias = a.bias; p.Cout = a.Cout;
  p.out = (bf16*)a.out; p.gn_partial = a.gn_partial; p.groups = a.groups;
  p.oq = (unsigned char*)a.out_q; p.os = (unsigned char*)a.out_s;
  const int grid = a.B * (a.Hin / QPH) * (a.Win / QPW) * (a.Cout / QBN);
  if (a.bias && ((size_t)a.bias & 15)) SRGD_FAIL("conv3x3_mxfp8: the bias array must be 16-byte aligned");
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mxfp8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mxfp8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
    once.done();
  }
  if (a.gn_partial) hipLaunchKernelGGL((conv3x3_mxfp8_kernel<true>), dim3(grid), dim3(NW * 64), QLDS, st, p);
  else hipLaunchKernelGGL((conv3x3_mxfp8_kernel<false>), dim3(grid), dim3(NW * 64), QLDS, st, p);
  SRGD_HIP(hipGetLastError());
#if SRGD_MXFP8_STAMPS
  {                                                     // stamp build: synchronous, prints the phase means and the slot timeline
    SRGD_HIP(hipStreamSynchronize(st));
    unsigned long long* dtl = nullptr;
    SRGD_HIP(hipGetSymbolAddress((void**)&dtl, HIP_SYMBOL(g_convq_timeline)));
    const StampSummary r = stamp_summary(dtl, grid, NW, 2);
    const int steps = 9 * ((a.C0 + a.C1) / QKC);
    fprintf(stderr, "[conv3x3_mxfp8 stamps] C %d+%d -> %d @%dx%d grid %d: prologue %.0f  main %.0f (%.0f per K-step)  epilogue %.0f  total %.0f  "
                    "(s_memtime ticks per workgroup)  in-kernel clock %.3f GHz | launch span %.1f us, workgroup %.2f us (wave-exit skew %.2f us), "
                    "slot gap exit -> next entry %.2f us, slot occupancy %.3f on %zu CUs\n",
            a.C0, a.C1, a.Cout, a.Hin, a.Win, grid, r.phase[0], r.phase[1], r.phase[1] / steps, r.phase[2], r.phase[0] + r.phase[1] + r.phase[2],
            r.clock_ghz, r.span_us, r.wg_us, r.skew_us, r.gap_us, r.occupancy, r.cus);
  }
#endif
  return 0;
}

}  // namespace srgd
