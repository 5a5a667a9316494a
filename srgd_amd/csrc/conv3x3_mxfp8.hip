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
//   * workgroup = 256 threads = 4 waves (2 along M x 2 along N), output tile = 8 x 16 pixel patch (M = 128) x 128 channels,
//     wave tile 64 x 64 = 4 x 4 MFMA blocks; <= 256 VGPRs, 76 KiB LDS -> two workgroups per CU
//   * per chunk: the (8+2) x (16+2) halo patch x 128 B (23 KiB) + its scale bytes (4 B / pixel); per K-step (tap, chunk) one
//     16 KiB weight tile + 512 B of weight scales, 3-deep ring; rows XOR-swizzled (chunk ^= row & 6): every ds_read_b128 of
//     the operand pattern is conflict-free at every tap shift (exhaustive search over the lane groups of ds_read_b128)
//   * epilogue as in the bf16 kernel: + bias, GroupNorm partial sums, LDS transpose, 16-byte stores.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int QPH = 8, QPW = 16;                 // output patch
constexpr int QHP = QPH + 2, QWP = QPW + 2;      // halo patch: 10 x 18 = 180 pixels
constexpr int QKC = 128;                         // channels per chunk = K of one MFMA
constexpr int QBN = 128;
constexpr int QNT = 256;
constexpr int QA_BYTES = 24 * 1024;              // 24 wave-instructions x 1 KiB (180 px * 128 B = 23,040 used)
constexpr int QAS_BYTES = 1024;                  // activation scales: 4 B per halo pixel (720 used)
constexpr int QB_TILE = QBN * QKC;               // 16 KiB of e4m3 weights per K-step
constexpr int QB_BYTES = QB_TILE + 1024;         // + [128 n][4 g] scale bytes (512) + 512 zero pad: one 17 KiB DMA unit
constexpr int QRING = 3;
constexpr int QLDS = QA_BYTES + QAS_BYTES + QRING * QB_BYTES;   // 77,824 B: two workgroups per CU

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds_wave_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds_wave_base, 4, voffset, 0, 0, 0);
}

struct ConvQArgs {
  const unsigned char* q0; const unsigned char* s0; int C0;     // MX-fp8 source 0: [B,H,W,C0] e4m3, [B,H,W,C0/32] E8M0
  const unsigned char* q1; const unsigned char* s1; int C1;     // optional source 1 (channel concat)
  int B, H, W;
  const unsigned char* w;     // packed [tap][cc][ntile][17 KiB]
  const float* bias;
  int Cout;
  bf16* out;
  float* gn_partial; int groups;
};

#define QWAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define QBARRIER()                       \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

template <bool STATS>
__global__ __launch_bounds__(QNT, 2) void conv3x3_mxfp8_kernel(ConvQArgs p) {
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
  const int S = CC * 9;

  // ---- A staging: 24 pieces of 1 KiB per chunk, wave w issues pieces w, w+4, ..., w+20; 16-byte chunk index in the LDS image
  // = piece*64 + lane -> pixel P = idx >> 3, stored position idx & 7 holds logical chunk (idx & 7) ^ (P & 6).
  // Per-lane (pixel offset or -1, logical chunk) of the six pieces as NAMED scalars (indexed arrays would go to scratch).
#define SRGD_QA_DECL(J)                                                       \
  int a_pix##J, a_sub##J;                                                     \
  {                                                                           \
    const int idx = (wave + 4 * J) * 64 + lane;                               \
    const int P = idx >> 3;                                                   \
    const int py = P / QWP, px = P - py * QWP;                                \
    const int y = y0 + py - 1, x = x0 + px - 1;                               \
    const bool ok = P < QHP * QWP && y >= 0 && y < p.H && x >= 0 && x < p.W;  \
    a_pix##J = ok ? y * p.W + x : -1;                                         \
    a_sub##J = (idx & 7) ^ (P & 6);                                           \
  }
  SRGD_QA_DECL(0) SRGD_QA_DECL(1) SRGD_QA_DECL(2) SRGD_QA_DECL(3) SRGD_QA_DECL(4) SRGD_QA_DECL(5)
#undef SRGD_QA_DECL
  int as_pix;                                     // scale piece: wave w stages the scale dwords of halo pixels 64w .. 64w+63
  {
    const int P = wave * 64 + lane;
    const int py = P / QWP, px = P - py * QWP;
    const int y = y0 + py - 1, x = x0 + px - 1;
    const bool ok = P < QHP * QWP && y >= 0 && y < p.H && x >= 0 && x < p.W;
    as_pix = ok ? y * p.W + x : -1;
  }
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

  auto issue_a_piece = [&](int cc, int j, int a_pix, int a_sub) {
    const bool first = cc < CC0;
    const int Cs = first ? p.C0 : p.C1;
    const int coff = (first ? cc : cc - CC0) * QKC;
    const int voff = a_pix >= 0 ? a_pix * Cs + coff + a_sub * 16 : 0x7ffffff0;
    char* dst = sA + (wave + 4 * j) * 1024;
    if (first) dma16(rq0, dst, voff); else dma16(rq1, dst, voff);
  };
  auto issue_a = [&](int cc) {                    // 7 DMA instructions per wave
    issue_a_piece(cc, 0, a_pix0, a_sub0); issue_a_piece(cc, 1, a_pix1, a_sub1); issue_a_piece(cc, 2, a_pix2, a_sub2);
    issue_a_piece(cc, 3, a_pix3, a_sub3); issue_a_piece(cc, 4, a_pix4, a_sub4); issue_a_piece(cc, 5, a_pix5, a_sub5);
    const bool first = cc < CC0;
    const int Cs32 = (first ? p.C0 : p.C1) / 32;
    const int voff = as_pix >= 0 ? as_pix * Cs32 + (first ? cc : cc - CC0) * 4 : 0x7ffffff0;
    if (first) dma4(rs0, sAs + wave * 256, voff); else dma4(rs1, sAs + wave * 256, voff);
  };
  auto issue_b = [&](int s) {                     // K-step s = cc*9 + tap -> weight unit (tap, cc); 5 DMA instructions per wave
    const int cc = s / 9, tap = s - cc * 9;
    const int base = (int)((size_t)(tap * CC + cc) * w_step_stride);
    char* dst = sB0 + (s % QRING) * QB_BYTES;     // (a clamped re-fetch of the last unit lands in the last unit's own slot)
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsw, dst + (wave + 4 * j) * 1024, base + (wave + 4 * j) * 1024 + lane * 16);
    dma4(rsw, dst + QB_TILE + wave * 256, base + QB_TILE + wave * 256 + lane * 4);
  };

  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0,
        c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;

  // operand addresses: row P (pixel of the halo patch / weight row n), logical chunks g and 4+g, swizzle chunk ^= row & 6
  auto row_lo = [&](int row) { return row * 128 + ((g ^ (row & 6)) << 4); };        // the hi chunk is this ^ 64
  int opq = 0;                                    // opaque zero: keeps the per-tap addresses out of long-lived registers
  auto compute = [&](int tap, int s) {
    const char* Bt = sB0 + (s % QRING) * QB_BYTES;
    const int dy = tap / 3, dx = tap - dy * 3;
    v8i a0, a1, a2, a3;
    int sa0, sa1, sa2, sa3;
#define SRGD_QLOAD_A(I)                                                              \
    {                                                                                \
      const int P = (4 * wm + I + dy) * QWP + r16 + dx + opq;                        \
      const int o = row_lo(P);                                                       \
      const v4i lo = *reinterpret_cast<const v4i*>(sA + o);                          \
      const v4i hi = *reinterpret_cast<const v4i*>(sA + (o ^ 64));                   \
      a##I = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};            \
      sa##I = *reinterpret_cast<const unsigned char*>(sAs + P * 4 + g);              \
    }
    SRGD_QLOAD_A(0) SRGD_QLOAD_A(1) SRGD_QLOAD_A(2) SRGD_QLOAD_A(3)
#undef SRGD_QLOAD_A
#define SRGD_QCOL(J, C0_, C1_, C2_, C3_)                                             \
    {                                                                                \
      const int n = wn * 64 + J * 16 + r16;                                          \
      const int o = row_lo(n);                                                       \
      const v4i lo = *reinterpret_cast<const v4i*>(Bt + o);                          \
      const v4i hi = *reinterpret_cast<const v4i*>(Bt + (o ^ 64));                   \
      const v8i bf = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};    \
      const int sb = *reinterpret_cast<const unsigned char*>(Bt + QB_TILE + n * 4 + g); \
      C0_ = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, bf, C0_, 0, 0, 0, sa0, 0, sb); \
      C1_ = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, bf, C1_, 0, 0, 0, sa1, 0, sb); \
      C2_ = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a2, bf, C2_, 0, 0, 0, sa2, 0, sb); \
      C3_ = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a3, bf, C3_, 0, 0, 0, sa3, 0, sb); \
    }
    SRGD_QCOL(0, c00, c10, c20, c30) SRGD_QCOL(1, c01, c11, c21, c31) SRGD_QCOL(2, c02, c12, c22, c32) SRGD_QCOL(3, c03, c13, c23, c33)
#undef SRGD_QCOL
    // Pin the accumulators here: hipcc otherwise SINKS the (register-only) MFMA chains of all nine taps below the chunk's last
    // barrier and carries every tap's operand fragments there through scratch (617 spilled VGPRs); s_setprio brackets
    // (cdna_hip_programming.md T5) did not hold them.  Empty asm, no instruction emitted.
    asm volatile("" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
    asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));
  };

  // ---- prologue: A(0), B[0], B[1]
  issue_a(0);
  issue_b(0);
  issue_b(1);                                    // S >= 9 always
  QWAIT_VM(5);                                   // everything but B[1]
  QBARRIER();

  // ---- main loop.  Per K-step: issue B[s+2]; compute(s); wait for B[s+1]; barrier.  Every step issues exactly one weight
  // unit (the last two steps re-fetch the final unit into a ring slot nobody reads any more) so that the unrolled tap loop is
  // branch-free: with data-dependent branches around the DMAs hipcc tail-merges the MFMA blocks of different taps and
  // passes their operands through scratch.  The halo patch is single-buffered (two of them would not leave room for two
  // workgroups per CU): at a chunk boundary the next patch is fetched after the barrier that retires the last tap's reads,
  // and the co-resident workgroup keeps the matrix pipe busy meanwhile.
  for (int cc = 0; cc < CC; ++cc) {
    asm volatile("" : "+v"(opq));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = cc * 9 + tap;
      issue_b(min(s + 2, S - 1));
      compute(tap, s);
      QWAIT_VM(5);                                 // everything but the unit issued in this step
      QBARRIER();
    }
    if (cc + 1 < CC) {
      issue_a(cc + 1);                             // every wave passed the barrier above: the old patch is dead
      QWAIT_VM(0);
      QBARRIER();
    }
  }
  QWAIT_VM(0);                                     // the dummy re-fetches must not land in the epilogue's staging area

  // ------------------------------- epilogue -------------------------------------------
  // tile transposed through LDS ([128 pixels][128 ch] bf16, rows padded to 272 B), stored as whole 256-byte channel rows
  constexpr int EROW = QBN * 2 + 16;
  float s1[4], s2[4];                                     // (the loop's last barrier retired every operand read)
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    s1[ni] = 0.f;
    s2[ni] = 0.f;
    const int cl = wn * 64 + ni * 16 + r16;               // column inside the tile
    const float bias = p.bias ? p.bias[nt * QBN + cl] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const f32x4 av = mi == 0 ? (ni == 0 ? c00 : ni == 1 ? c01 : ni == 2 ? c02 : c03)
                     : mi == 1 ? (ni == 0 ? c10 : ni == 1 ? c11 : ni == 2 ? c12 : c13)
                     : mi == 2 ? (ni == 0 ? c20 : ni == 1 ? c21 : ni == 2 ? c22 : c23)
                               : (ni == 0 ? c30 : ni == 1 ? c31 : ni == 2 ? c32 : c33);
      // D map: column = lane & 15, row = (lane >> 4) * 4 + reg -> pixel (patch row 4 wm + mi, x = 4 g + reg)
      char* trow = smem + ((4 * wm + mi) * QPW + g * 4) * EROW + cl * 2;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float v = av[reg] + bias;
        if (STATS) {
          s1[ni] += v;
          s2[ni] += v * v;
        }
        *reinterpret_cast<bf16*>(trow + reg * EROW) = (bf16)v;
      }
    }
  }
  __syncthreads();
  {
    bf16* obase = p.out + ((size_t)(b * p.H + y0) * p.W + x0) * p.Cout + nt * QBN;
#pragma unroll
    for (int i = 0; i < (QPH * QPW * 16) / QNT; ++i) {
      const int q = tid + QNT * i;                        // 16-byte chunk: pixel q/16, channels (q%16)*8..+7
      const int pix = q >> 4, c16 = q & 15;
      const int py = pix / QPW, px = pix - py * QPW;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + pix * EROW + c16 * 16);
      *reinterpret_cast<bf16x8*>(obase + ((size_t)py * p.W + px) * p.Cout + c16 * 8) = v;
    }
  }
  if (STATS) {
    __syncthreads();                                      // the staged output tile has been read back
    float* cs = reinterpret_cast<float*>(smem);           // [2 (wm)][128][2]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float t1 = s1[ni], t2 = s2[ni];
      t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 32, 64);
      t1 += __shfl_xor(t1, 16, 64);
      t2 += __shfl_xor(t2, 16, 64);
      if (lane < 16) {
        const int cl = wn * 64 + ni * 16 + r16;
        cs[(wm * QBN + cl) * 2 + 0] = t1;
        cs[(wm * QBN + cl) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    const int cpg = p.Cout / p.groups;                    // multiple of 16, divides or is a multiple of 128
    const int g_in_tile = cpg >= QBN ? 1 : QBN / cpg;
    if (tid < g_in_tile) {
      const int span = cpg >= QBN ? QBN : cpg;
      float a1 = 0.f, a2 = 0.f;
      for (int c = 0; c < span; ++c) {
        const int cl = tid * span + c;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          a1 += cs[(k * QBN + cl) * 2 + 0];
          a2 += cs[(k * QBN + cl) * 2 + 1];
        }
      }
      const int tiles_per_group = cpg >= QBN ? cpg / QBN : 1;
      const int grp = cpg >= QBN ? (nt * QBN) / cpg : (nt * QBN) / cpg + tid;
      const int nslots = tiles_y * tiles_x * tiles_per_group;
      const int slot = trem * tiles_per_group + (cpg >= QBN ? nt % tiles_per_group : 0);
      float* dst = p.gn_partial + ((size_t)(b * p.groups + grp) * nslots + slot) * 2;
      dst[0] = a1;
      dst[1] = a2;
    }
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

// E8M0 exponent of a block with maximum magnitude amax (OCP MX: floor(log2 amax) - emax(e4m3) with emax = 8), clamped
int mx_block_exponent(float amax) {
  if (!(amax > 0.f)) return -127;
  int ex;
  (void)std::frexp(amax, &ex);                    // amax = m * 2^ex, m in [0.5, 1) -> floor(log2 amax) = ex - 1
  int e = ex - 1 - 8;
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

// OIHW fp32 -> [tap][cc][ntile][17 KiB]: 128 rows x 128 B of e4m3 (swizzled LDS image) + [128][4] E8M0 bytes + zero pad.
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
            unit[QB_TILE + n * 4 + blk] = (unsigned char)(ex + 127);
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
  const int grid = a.B * (a.Hin / QPH) * (a.Win / QPW) * (a.Cout / QBN);
  static bool attr_set[64] = {};
  if (first_use_on_device(attr_set)) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mxfp8_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mxfp8_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
  }
  if (a.gn_partial) hipLaunchKernelGGL((conv3x3_mxfp8_kernel<true>), dim3(grid), dim3(QNT), QLDS, st, p);
  else hipLaunchKernelGGL((conv3x3_mxfp8_kernel<false>), dim3(grid), dim3(QNT), QLDS, st, p);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
