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
//   * epilogue as in the bf16 kernel: + bias, GroupNorm partial sums, LDS transpose, 16-byte stores.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace srgd {
namespace {

#ifndef SRGD_MXFP8_EPI_PRIO
#define SRGD_MXFP8_EPI_PRIO 0
#endif
constexpr int QPH = 8, QPW = 32;                 // output patch
constexpr int QHP = QPH + 2, QWP = QPW + 2;      // halo patch: 10 x 34 = 340 pixels
constexpr int QKC = 128;                         // channels per chunk = K of one MFMA
constexpr int QBN = 128;
constexpr int QNW_DEFAULT = 4;                   // waves per workgroup of the shipped instance (kernel template parameter NW; 8 measured 3-7 % slower)
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
  unsigned long long* stamps;             // diagnostics (SRGD_MXFP8_STAMPS=1): per-phase s_memtime deltas summed over workgroups; null otherwise
};

// phase accumulators of the diagnostic mode: [prologue, main loop, epilogue, total, workgroups, s_memrealtime ticks]
__device__ unsigned long long g_convq_stamps[8];

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

// NW = waves per workgroup.  4 (round 2): 2 x 2 waves, wave tile 128 x 64 (128 accumulators, <= 256 VGPRs), two waves per SIMD.
// 8 (round 3): 4 x 2 waves, wave tile 64 x 64 (64 accumulators, <= 128 VGPRs), FOUR waves per SIMD at the same two workgroups per
// CU - the bf16 kernel's shape.  Round 3's diagnostics showed the 4-wave kernel is bound by neither LDS, DMA nor
// barriers; the hypothesis behind this shape - a wave's MFMA stream has gaps and with two waves per SIMD nothing fills them while
// the co-resident workgroup is in its prologue, epilogue or a chunk-boundary patch reload - did not hold: correct (same tests),
// and 3-7 % SLOWER on all twelve shapes (profiles/r3/conv3x3_mxfp8_w8_ab.txt; 33 % more ds_read bytes per FLOP, pixel fragments
// single-buffered to fit 64 VGPRs next to 64 accumulator AGPRs).  Kept as an A/B switch (SRGD_MXFP8_WAVES=8); 4 ships.
template <bool STATS, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv3x3_mxfp8_kernel(ConvQArgs p) {
  constexpr int NT = NW * 64;               // threads
  constexpr int NWM = NW / 2;               // wave rows (along pixels); 2 wave columns (along channels)
  constexpr int RPW = QPH / NWM;            // patch rows per wave: 4 or 2
  constexpr int NM = RPW * 2;               // 16-pixel fragments per wave: 8 or 4
  constexpr int A_PIECES = (4 * QA_PIECES + NW - 1) / NW;      // 1 KiB halo-patch pieces per wave and chunk: 11 or 6
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

  auto issue_a = [&](int cc) {                    // 11 + 2 (NW = 4) or <= 6 + 1 (NW = 8) DMA instructions per wave
    const bool first = cc < CC0;
    const int Cs = first ? p.C0 : p.C1;
    const int ccl = first ? cc : cc - CC0;
    // per-chunk recomputation of the 13 staging offsets from a freshly derived lane id (they are invariant across chunks, and
    // hoisted out of the K loop they cost 26 long-lived VGPRs; so would a copy of the lane id kept for this purpose alone)
    const int opq_a = lane_id_opaque();
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
      if (NW * j + wave >= 4 * QA_PIECES) break;  // (NW = 8: 44 pieces over 8 waves - the last round only has waves 0..3; uniform)
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
#define SRGD_QABASE(K) const int ab##K = apix + ((g ^ (((K) + r7) & 6)) << 4), ac##K = ab##K ^ 64;
  SRGD_QABASE(0) SRGD_QABASE(1) SRGD_QABASE(2) SRGD_QABASE(3) SRGD_QABASE(4) SRGD_QABASE(5) SRGD_QABASE(6) SRGD_QABASE(7)
#undef SRGD_QABASE
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
    const char* Bt = sB0 + slot * QB_BYTES;
    const int dy = tap / 3, dx = tap - dy * 3;
    int r7t = r7;                                  // NW = 8: refreshed per step, so that no fragment address survives a step in a register
    if constexpr (NW == 8) asm volatile("" : "+v"(r7t));
    v8i b0, b1, b2, b3;
#ifdef SRGD_MXFP8_DIAG_NOB                  // timing-only diagnostic (wrong results): weight fragments made up in registers
    const int sbw = r7;
#define SRGD_QLOAD_B(J) b##J = v8i{r7, lane, r7 + J, lane, r7, lane + J, r7, lane};
#else
    const int sbw = *reinterpret_cast<const int*>(Bt + bsb);
#define SRGD_QLOAD_B(J)                                                              \
    {                                                                                \
      const v4i lo = *reinterpret_cast<const v4i*>(Bt + bb + J * 2048);              \
      const v4i hi = *reinterpret_cast<const v4i*>(Bt + (bb ^ 64) + J * 2048);       \
      b##J = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};            \
    }
#endif
    SRGD_QLOAD_B(0) SRGD_QLOAD_B(1) SRGD_QLOAD_B(2) SRGD_QLOAD_B(3)
#undef SRGD_QLOAD_B
    // SRGD_MXFP8_DIAG_NOLDS (timing-only diagnostic build, wrong results): fragments 1..7 are register copies of fragment 0 -
    // 2/3 of the step's ds_read traffic gone - to price the kernel's LDS-bandwidth limit (tools/build_variant.py)
#ifndef SRGD_MXFP8_DIAG_NOLDS
#define SRGD_MXFP8_DIAG_NOLDS 0
#endif
#define SRGD_QLOAD_A(I)                                                              \
    v8i a##I; int sa##I;                                                             \
    if (SRGD_MXFP8_DIAG_NOLDS && (I) > 0) { a##I = a0; sa##I = sa0; } else           \
    {                                                                                \
      const int Pc = ((I >> 1) + dy) * QWP + (I & 1) * 16 + dx;                      \
      const int k7 = Pc & 7;                                                         \
      /* NW = 4: one of the 16 precomputed bases; NW = 8 (128-register budget): 3 + 1 VALU from the opaque copy of r7 */ \
      const int o = NW == 8 ? apix + ((g ^ ((k7 + r7t) & 6)) << 4)                                                      \
                  : k7 == 0 ? ab0 : k7 == 1 ? ab1 : k7 == 2 ? ab2 : k7 == 3 ? ab3 : k7 == 4 ? ab4 : k7 == 5 ? ab5 : k7 == 6 ? ab6 : ab7; \
      const int o2 = NW == 8 ? (o ^ 64)                                                                                 \
                   : k7 == 0 ? ac0 : k7 == 1 ? ac1 : k7 == 2 ? ac2 : k7 == 3 ? ac3 : k7 == 4 ? ac4 : k7 == 5 ? ac5 : k7 == 6 ? ac6 : ac7; \
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
    // a barrier, and the inputs come from LDS reads whose lgkmcnt waits the compiler still inserts.
    // opsel of the weight scale (byte J of sbw): bit 0 -> op_sel[1], bit 1 -> op_sel_hi[1]
#define SRGD_QMM_OPSEL_0 "op_sel_hi:[0,0,0]"
#define SRGD_QMM_OPSEL_1 "op_sel:[0,1,0] op_sel_hi:[0,0,0]"
#define SRGD_QMM_OPSEL_2 "op_sel_hi:[0,1,0]"
#define SRGD_QMM_OPSEL_3 "op_sel:[0,1,0] op_sel_hi:[0,1,0]"
    // NW = 8: the accumulators live in AGPRs ("+a"; gfx950's register file is unified, 128 per wave at four waves per SIMD):
    // 64 AGPRs + <= 64 VGPRs are two allocation problems the register allocator can solve; as one class of 128 with 4- and
    // 8-register tuples it spilled 120-160 registers (accumulators included) into the K loop.
#define SRGD_QMM(C_, A_, SA_, B_, J_)                                                                      \
    if constexpr (NW == 8)                                                                                 \
      asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " SRGD_QMM_OPSEL_##J_          \
                   : "+a"(C_) : "v"(A_), "v"(B_), "v"(SA_), "v"(sbw));                                     \
    else                                                                                                   \
      asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " SRGD_QMM_OPSEL_##J_          \
                   : "+v"(C_) : "v"(A_), "v"(B_), "v"(SA_), "v"(sbw))
#define SRGD_QROW(I)                                                                 \
    SRGD_QMM(c##I##0, a##I, sa##I, b0, 0); SRGD_QMM(c##I##1, a##I, sa##I, b1, 1);    \
    SRGD_QMM(c##I##2, a##I, sa##I, b2, 2); SRGD_QMM(c##I##3, a##I, sa##I, b3, 3);
    // software pipeline over the pixel fragments, fenced for the scheduler (left alone it hoists all eight fragment loads to
    // the top of the step and spills ~130 registers into the loop): the loads of fragment i+1 are issued ahead of the 4 MFMAs
    // (128 cycles of matrix pipe) of fragment i; two fragments live at a time
    if constexpr (NW == 8) {
      // 64 VGPRs next to the 64 accumulator AGPRs: the four weight fragments (32) stay, the pixel fragments stream through ONE
      // at a time (9 registers) - with four waves per SIMD another wave's MFMAs cover the fragment's LDS latency
      SRGD_QLOAD_A(0)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(0)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QLOAD_A(1)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(1)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QLOAD_A(2)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(2)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QLOAD_A(3)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(3)
      return;
    }
    SRGD_QLOAD_A(0)
    __builtin_amdgcn_sched_barrier(0);
    SRGD_QLOAD_A(1)
    __builtin_amdgcn_sched_barrier(0);
    SRGD_QROW(0)
    __builtin_amdgcn_sched_barrier(0);
    SRGD_QLOAD_A(2)
    __builtin_amdgcn_sched_barrier(0);
    SRGD_QROW(1)
    __builtin_amdgcn_sched_barrier(0);
    SRGD_QLOAD_A(3)
    __builtin_amdgcn_sched_barrier(0);
    SRGD_QROW(2)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NM == 8) {
      SRGD_QLOAD_A(4)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(3)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QLOAD_A(5)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(4)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QLOAD_A(6)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(5)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QLOAD_A(7)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(6)
      __builtin_amdgcn_sched_barrier(0);
      SRGD_QROW(7)
    } else {
      SRGD_QROW(3)
    }
#undef SRGD_QROW
#undef SRGD_QMM
#undef SRGD_QMM_OPSEL_0
#undef SRGD_QMM_OPSEL_1
#undef SRGD_QMM_OPSEL_2
#undef SRGD_QMM_OPSEL_3
#undef SRGD_QLOAD_A
  };

  unsigned long long t0 = 0, t1 = 0, t2 = 0, r0 = 0;
  if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // ---- prologue: A(0), B[0]
  issue_a(0);
  issue_b(0, 0, 0);
#ifdef SRGD_MXFP8_DIAG_NOPROLOGUE_WAIT      // timing-only diagnostic (wrong results): the tile does not wait for its first patch and
  QWAIT_VM(18);                             // weight unit - the upper bound of what a persistent, prefetching workgroup could hide
#else
  QWAIT_VM(0);
#endif
  QBARRIER();

  if (p.stamps) t1 = __builtin_amdgcn_s_memtime();
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
#ifndef SRGD_MXFP8_DIAG_NODMA              // timing-only diagnostic (wrong results): no weight DMA inside the K loop
      if (tap < 8) issue_b(cc, tap + 1, slot ^ 1);
      else issue_b(min(cc + 1, CC - 1), cc + 1 < CC ? 0 : 8, slot ^ 1);   // (the last step re-fetches its own unit into the idle slot)
#endif
      compute(tap, slot);
#ifdef SRGD_MXFP8_DIAG_LATE_WAIT           // timing-only diagnostic (wrong results): the step does not wait for its weight DMA -
      QWAIT_VM(10);                        // two more steps' worth may stay in flight - to price the DMA latency on the critical path
#else
      QWAIT_VM(0);
#endif
#ifndef SRGD_MXFP8_DIAG_NO_BARRIER         // timing-only diagnostic (wrong results): no per-step workgroup barrier
      QBARRIER();
#endif
    }
#ifdef SRGD_MXFP8_DIAG_NOAPATCH              // timing-only diagnostic (wrong results): the halo patch is staged once per tile
    if (false) {
#else
    if (cc + 1 < CC) {
#endif
      issue_a(cc + 1);                             // every wave passed the barrier above: the old patch is dead
      QWAIT_VM(0);
      QBARRIER();
    }
  }
  QWAIT_VM(0);                                     // (nothing is in flight here; kept next to the epilogue's reuse of the ring)
  if (p.stamps) t2 = __builtin_amdgcn_s_memtime();
#if SRGD_MXFP8_EPI_PRIO
  __builtin_amdgcn_s_setprio(SRGD_MXFP8_EPI_PRIO);   // see conv3x3_bf16.hip: the epilogue competes with the co-resident workgroup's MFMA stream for issue slots
#endif
  // The MFMAs are inline asm (compute()): the compiler does not know that the accumulators were written by the matrix pipe and
  // inserts none of the wait states a VALU read of an XDL result needs (<= 18 for a 16-pass MFMA).  The accumulators are
  // threaded through these statements, so every epilogue read comes after >= 32 wait states behind the last MFMA.
  if constexpr (NW == 8) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(c00), "+a"(c01), "+a"(c02), "+a"(c03), "+a"(c10), "+a"(c11), "+a"(c12), "+a"(c13));
    asm volatile("" : "+a"(c20), "+a"(c21), "+a"(c22), "+a"(c23), "+a"(c30), "+a"(c31), "+a"(c32), "+a"(c33));
  } else {
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
    asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));
  }
  if constexpr (NM == 8) {
    asm volatile("" : "+v"(c40), "+v"(c41), "+v"(c42), "+v"(c43), "+v"(c50), "+v"(c51), "+v"(c52), "+v"(c53));
    asm volatile("" : "+v"(c60), "+v"(c61), "+v"(c62), "+v"(c63), "+v"(c70), "+v"(c71), "+v"(c72), "+v"(c73));
  }

  // ------------------------------- epilogue -------------------------------------------
  // tile transposed through LDS ([256 pixels][128 ch] bf16, rows padded to 272 B), stored as whole 256-byte channel rows
  constexpr int EROW = QBN * 2 + 16;
  // Per-lane epilogue addresses come from a lane id re-derived HERE (v_mbcnt, opaque to the optimiser): derived from `tid` they are
  // computed ahead of the K loop and carried through it - in registers this kernel does not have (256-VGPR budget): the <STATS>
  // instance spilled 6 of them, and the reload of one at the top of every chunk put an s_waitcnt vmcnt(0) right behind the
  // weight DMA of tap 0 (one exposed L2 round trip per chunk = per tile on the 128-channel layers).
  const int laneE = lane_id_opaque(), tidE = wave * 64 + laneE, r16E = laneE & 15, gE = laneE >> 4;
  float s1[4], s2[4];                                     // (the loop's last barrier retired every operand read)
  const bool odd_lane = (r16E & 1) != 0;
  const int pair_off = odd_lane ? 2 * EROW - 2 : 0;       // odd lane: rows 2-3, the even channel's column
#define SRGD_QACC(MI, NI) (NI == 0 ? c##MI##0 : NI == 1 ? c##MI##1 : NI == 2 ? c##MI##2 : c##MI##3)
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    s1[ni] = 0.f;
    s2[ni] = 0.f;
    const int cl = wn * 64 + ni * 16 + r16E;               // column inside the tile
    const float bias = p.bias ? p.bias[nt * QBN + cl] : 0.f;
    f32x2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};              // the column's sums as a register pair (even | odd rows)
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
      const f32x4 av = mi == 0 ? SRGD_QACC(0, ni) : mi == 1 ? SRGD_QACC(1, ni) : mi == 2 ? SRGD_QACC(2, ni) : mi == 3 ? SRGD_QACC(3, ni)
                     : mi == 4 ? SRGD_QACC(4, ni) : mi == 5 ? SRGD_QACC(5, ni) : mi == 6 ? SRGD_QACC(6, ni) : SRGD_QACC(7, ni);
      // D map: column = laneE & 15, row = (laneE >> 4) * 4 + reg -> pixel (patch row RPW wm + (mi >> 1), x = 16 (mi & 1) + 4 gE + reg)
      char* trow = smem + ((RPW * wm + (mi >> 1)) * QPW + (mi & 1) * 16 + gE * 4) * EROW + cl * 2;
      // written for instruction count (conv3x3_bf16.hip: the epilogue is 35-40 % of a 9-step tile's instruction stream here):
      // packed fp32 adds / fmas on register pairs, one v_cvt_pk_bf16_f32 per two values, ds_write_b16 + ds_write_b16_d16_hi
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const f32x2 b2 = {bias, bias};
      const f32x2 v01 = f32x2{av[0], av[1]} + b2, v23 = f32x2{av[2], av[3]} + b2;
      if (STATS) {
        s1p += v01 + v23;
        s2p = __builtin_elementwise_fma(v01, v01, s2p);
        s2p = __builtin_elementwise_fma(v23, v23, s2p);
      }
      const bf16x2_t t01 = __builtin_convertvector(v01, bf16x2_t), t23 = __builtin_convertvector(v23, bf16x2_t);
      // lane-pair exchange + two conflict-free ds_write_b32 instead of four ds_write_b16 (conv3x3_bf16.hip): the even lane takes
      // rows 0-1, the odd lane rows 2-3 of both lanes' channels
      const unsigned own01 = __builtin_bit_cast(unsigned, t01), own23 = __builtin_bit_cast(unsigned, t23);
      const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_lane ? own01 : own23), 0xB1, 0xf, 0xf, true);
      const unsigned lo_ch = odd_lane ? recv : own01, hi_ch = odd_lane ? own23 : recv;
      char* prow = trow + pair_off;
      *reinterpret_cast<unsigned*>(prow) = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x05040100u);
      *reinterpret_cast<unsigned*>(prow + EROW) = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x07060302u);
    }
    if (STATS) {
      s1[ni] = s1p[0] + s1p[1];
      s2[ni] = s2p[0] + s2p[1];
    }
  }
#undef SRGD_QACC
  // column sums behind the staged tile: one barrier publishes both, and nothing below waits for the output stores
  // (conv3x3_bf16.hip: reducing after the stores cost ~6,000 cycles per tile behind a vmcnt(0))
  float* const cs = reinterpret_cast<float*>(smem + QPH * QPW * EROW);      // [NWM (wm)][128][2] floats at byte 69,632 (2 or 4 KiB: < QLDS)
  if (STATS) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float t1 = s1[ni], t2 = s2[ni];
      t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 32, 64);
      t1 += __shfl_xor(t1, 16, 64);
      t2 += __shfl_xor(t2, 16, 64);
      if (laneE < 16) {
        const int cl = wn * 64 + ni * 16 + r16E;
        cs[(wm * QBN + cl) * 2 + 0] = t1;
        cs[(wm * QBN + cl) * 2 + 1] = t2;
      }
    }
  }
  __syncthreads();
  {
    // 16-byte chunk q = tidE + NT i: NT / 16 pixels per iteration = whole patch rows (NT = 256: half a row), so pixel
    // (row, column) advances by a constant: ONE per-lane address + a uniform stride instead of a 64-bit computation per store
    constexpr int PXI = NT / 16;                           // pixels per iteration: 16 (half a patch row) or 32 (one row)
    const int pix0 = tidE >> 4, c16 = tidE & 15;
    const int py0 = pix0 / QPW, px0 = pix0 - py0 * QPW;    // (py0 = 0: pix0 < 32)
    const size_t o0 = ((size_t)(b * p.H + y0 + py0) * p.W + x0 + px0) * p.Cout + nt * QBN + c16 * 8;
    const char* src = smem + pix0 * EROW + c16 * 16;
#pragma unroll
    for (int i = 0; i < (QPH * QPW * 16) / NT; ++i) {
      // NT = 512: row i, same column; NT = 256: row i / 2, column + 16 (i & 1)
      const size_t oo = o0 + (PXI == 32 ? (size_t)i * p.W * p.Cout : (size_t)(i >> 1) * p.W * p.Cout + (size_t)(i & 1) * 16 * p.Cout);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i * PXI * EROW);
      *reinterpret_cast<bf16x8*>(p.out + oo) = v;
      if (p.oq) mx_store_twin(v, p.oq, p.os, oo, tidE & 3);
    }
  }
  if (STATS) {
    const int cpg = p.Cout / p.groups;                    // multiple of 16, divides or is a multiple of 128
    const int span = cpg >= QBN ? QBN : cpg;              // columns of this tile that belong to one group: 16, 32, 64 or 128
    float a1 = 0.f, a2 = 0.f;
    if (tidE < QBN) {
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        a1 += cs[(k * QBN + tidE) * 2 + 0];
        a2 += cs[(k * QBN + tidE) * 2 + 1];
      }
      for (int o = 1; o < span && o < 64; o <<= 1) {
        a1 += __shfl_xor(a1, o, 64);
        a2 += __shfl_xor(a2, o, 64);
      }
    }
    if (span == QBN) {                                    // a group spans both waves: combine through LDS, raw barriers
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      QBARRIER();
      if (tidE < QBN && laneE == 0) { cs[wave * 2 + 0] = a1; cs[wave * 2 + 1] = a2; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      QBARRIER();
      if (tidE == 0) { a1 = cs[0] + cs[2]; a2 = cs[1] + cs[3]; }
    }
    if (tidE < QBN && (tidE % span) == 0) {
      const int tiles_per_group = cpg >= QBN ? cpg / QBN : 1;
      const int grp = (nt * QBN) / cpg + (cpg >= QBN ? 0 : tidE / span);
      const int nslots = tiles_y * tiles_x * tiles_per_group;
      const int slot = trem * tiles_per_group + (cpg >= QBN ? nt % tiles_per_group : 0);
      float* dst = p.gn_partial + ((size_t)(b * p.groups + grp) * nslots + slot) * 2;
      dst[0] = a1;
      dst[1] = a2;
    }
  }
  if (p.stamps && tidE == 0) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    atomicAdd(&p.stamps[0], t1 - t0); atomicAdd(&p.stamps[1], t2 - t1); atomicAdd(&p.stamps[2], t3 - t2);
    atomicAdd(&p.stamps[3], t3 - t0); atomicAdd(&p.stamps[4], 1ull);
    atomicAdd(&p.stamps[5], __builtin_amdgcn_s_memrealtime() - r0);
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
    if (a.Cout % a.groups || cpg % 16) return false;
    if (!(QBN % cpg == 0 || cpg % QBN == 0)) return false;
  }
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.C0, a.C1) >= (1ull << 31)) return false;
  if ((size_t)9 * ((a.C0 + a.C1) / QKC) * (a.Cout / QBN) * QB_BYTES >= (1ull << 31)) return false;
  return true;
}

int conv3x3_mxfp8_stats_slots(const ConvArgs& a) {
  if (a.groups <= 0) return 0;
  const int cpg = a.Cout / a.groups;
  return (a.Hin / QPH) * (a.Win / QPW) * (cpg >= QBN ? cpg / QBN : 1);
}

// OIHW fp32 -> [tap][cc][ntile][16.5 KiB]: 128 rows x 128 B of e4m3 (swizzled LDS image) + 512 E8M0 bytes [wn][r16][blk][J].
// One scale per (output channel, tap, 32 input channels): w = q * 2^(byte - 127).
void pack_conv3x3_mxfp8(const float* src_oihw, int Cin, int Cout, std::vector<unsigned char>& out) {
  const int CC = Cin / QKC, NTL = Cout / QBN;
  out.assign((size_t)9 * CC * NTL * QB_BYTES, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned char* unit = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * QB_BYTES;
        for (int n = 0; n < QBN; ++n) {
          const int o = nt * QBN + n;
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
  static const int want_stamps = env_int("SRGD_MXFP8_STAMPS", 0) ? 1 : 0;
  p.stamps = nullptr;
  if (want_stamps) {
    SRGD_HIP(hipGetSymbolAddress((void**)&p.stamps, HIP_SYMBOL(g_convq_stamps)));
    SRGD_HIP(hipMemsetAsync(p.stamps, 0, sizeof(unsigned long long) * 8, st));
  }
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define SRGD_SETQ(S_, N_)                                                                                 \
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mxfp8_kernel<S_, N_>),              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
    SRGD_SETQ(true, 4) SRGD_SETQ(false, 4) SRGD_SETQ(true, 8) SRGD_SETQ(false, 8)
#undef SRGD_SETQ
    once.done();
  }
  static const int nw = env_int("SRGD_MXFP8_WAVES", QNW_DEFAULT) == 8 ? 8 : 4;    // SRGD_MXFP8_WAVES=8: the 8-wave shape (A/B switch)
  if (nw == 8) {
    if (a.gn_partial) hipLaunchKernelGGL((conv3x3_mxfp8_kernel<true, 8>), dim3(grid), dim3(512), QLDS, st, p);
    else hipLaunchKernelGGL((conv3x3_mxfp8_kernel<false, 8>), dim3(grid), dim3(512), QLDS, st, p);
  } else {
    if (a.gn_partial) hipLaunchKernelGGL((conv3x3_mxfp8_kernel<true, 4>), dim3(grid), dim3(256), QLDS, st, p);
    else hipLaunchKernelGGL((conv3x3_mxfp8_kernel<false, 4>), dim3(grid), dim3(256), QLDS, st, p);
  }
  SRGD_HIP(hipGetLastError());
  if (want_stamps) {                                    // diagnostic mode: synchronous, prints the mean ticks per workgroup and phase
    unsigned long long h[8];
    SRGD_HIP(hipStreamSynchronize(st));
    SRGD_HIP(hipMemcpy(h, p.stamps, sizeof(h), hipMemcpyDeviceToHost));
    const double n = h[4] ? (double)h[4] : 1.0;
    const int steps = 9 * ((a.C0 + a.C1) / QKC);
    fprintf(stderr, "[conv3x3_mxfp8 stamps] C %d+%d -> %d @%dx%d grid %d: prologue %.0f  main %.0f (%.0f per K-step)  epilogue %.0f  total %.0f  "
                    "(s_memtime ticks per workgroup)  in-kernel clock %.3f GHz\n",
            a.C0, a.C1, a.Cout, a.Hin, a.Win, grid, h[0] / n, h[1] / n, h[1] / n / steps, h[2] / n, h[3] / n,
            h[5] ? 0.1 * (double)h[3] / (double)h[5] : 0.0);
  }
  return 0;
}

}  // namespace srgd
