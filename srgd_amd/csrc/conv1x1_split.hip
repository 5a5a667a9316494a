// Pointwise (1x1) convolutions in SPLIT-OPERAND precision for gfx950 (MI355X): the layers conv1x1_bf16.hip serves in the bf16 modes -
// ResnetBlock res_conv (reference model.py:271), to_qkv / to_out (:300-303, :338-340), PixelShuffleUpsample's 1x1 + SiLU +
// PixelShuffle (:70-98), Downsample's space-to-depth + 1x1 (:106-110, as a 2x2 / stride-2 gather), and the 7x7 input convolution as
// a 7x1 gather over 64-element windows (:597) - on fp32 tensors, every product
// as three f16 MFMAs on (hi, lo) operand pairs (arithmetic: conv3x3_split.hip).
//
// These layers are HBM-bound (fp32 in and out; <= 192 MFMA-FLOP per input byte at Cout = 128), so the kernel is a streaming GEMM built
// for bytes in flight, like conv1x1_bf16.hip:
//   * workgroup = 512 threads = 8 waves; output tile = 256 consecutive pixels x 128 channels; a wave owns 32 pixels x ALL 128
//     channels (2 pixel blocks x 8 weight-row blocks of v_mfma_f32_16x16x32_f16: 64 accumulators);
//   * every K-step (32 channels) is ONE 48 KB stage brought in by LDS-DMA (6 buffer_load ... lds per wave) into a 3-deep ring - two
//     stages (96 KB: 64 KB of pixels) are always in flight per CU: the fp32 pixel rows (256 x 128 B, 16-byte chunks XOR-swizzled by
//     the row so that the fragment reads are conflict-free; 8 lanes read one full 128-byte line) and the weight unit (hi tile | lo
//     tile, split on the host, pre-swizzled);
//   * a wave DMAs exactly the 32 pixel rows it consumes, reads its fp32 fragments from LDS (8 consecutive channels of a pixel = two
//     ds_read_b128 per lane and block) and splits them in registers: every input element is split once.  (Round 6 first had every
//     lane load its fragments straight from global memory - no LDS for the pixels at all.  Correct, but a quarter-wave then touches
//     16 different 128-byte lines per instruction: 2.3-3.4 TB/s on the 256x256 layers, profiles/r6/conv1x1_split_direct_loads.txt);
//   * weights as srcA with the tile's rows stored in the order (tile row 16 J + 4 g + e <- channel 32 g + 4 J + e), so that a
//     lane's 32 accumulators of a pixel are 32 CONSECUTIVE channels: the epilogue (x inverse weight scale, + bias, + residual or
//     SiLU + PixelShuffle scatter) works on the accumulators directly; the finished values are transposed through a 2 KB per-wave
//     LDS staging area so that every store instruction writes full 128-byte lines (the stores are what bounds this kernel);
//   * LDS 160 KB (144 KB ring + 16 KB store staging): one workgroup per CU, two waves per SIMD - so the workgroup is PERSISTENT (grid = number of CUs) and walks its
//     tiles with one continuous stage stream: the next tile's first two stages are in flight while this tile's stores drain
//     (128 -> 128 @256x256: 3.8 -> 4.4 TB/s, profiles/r6/conv1x1_split_persistent.txt);
//   * further epilogues fold whole passes of the f16x3 mode into this kernel: the GroupNorm2 + SiLU + residual tail of a ResnetBlock
//     (SEPI_GNTAIL), RMSNorm(conv) * g + residual (SEPI_RMS_RESIDUAL) and RMSNorm of the INPUT (RMS_IN: gain folded into the weights).
// Roofline: HBM (algorithmic bytes = input once per n-tile + output once + weights); K-heavy shapes at 32x32 / 64x64 are MFMA-bound
// at a third of the f16 peak like the 3x3 kernel.
#include <cmath>
#include <cstdlib>

#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int BM = 256, BN = 128, KC = 32, NT = 512;
constexpr int A_BYTES = BM * KC * 4;           // 32 KiB: 256 pixel rows x 128 B of fp32
constexpr int B_TILE = BN * KC * 2;            // 8 KiB: one 16-bit weight tile
constexpr int B_SLOT = 2 * B_TILE;             // hi | lo
constexpr int STAGE = A_BYTES + B_SLOT;        // 48 KiB
constexpr int RING = 3;
constexpr int STG_BYTES = 2048;                // per-wave staging area of the store transposition (epilogue)
constexpr int LDS_BYTES = RING * STAGE + (NT / 64) * STG_BYTES;   // 147,456 + 16,384 = 163,840: all of a CU's LDS, one workgroup per CU
enum { SEPI_PLAIN = 0, SEPI_RESIDUAL = 1, SEPI_PS_SILU = 2, SEPI_RMS_RESIDUAL = 3, SEPI_GNTAIL = 4 };

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct Split1Args {
  const float* in0; const float* in1; int C0, C1;
  int B, Hin, Win, Hout, Wout;
  int KH, KW, stride;     // 1x1, or 2x2 / stride 2 (space-to-depth folded into the K walk)
  int ps0, ps1;           // pixel stride of each source in elements
  const void* w;          // pack_conv1x1_split
  const float* bias;
  float w_inv_scale;
  int Cout;
  float* out;
  const float* aux;       // SEPI_RESIDUAL / SEPI_RMS_RESIDUAL: tensor added to the output; SEPI_GNTAIL: tensor the GroupNorm tail is applied to
  const float* gn_a; const float* gn_b;   // SEPI_GNTAIL: [B][Cout] scale / shift of y = silu(a * aux + b)
  int n_wg_tiles;         // m-tiles x n-tiles (the grid is persistent: at most one workgroup per CU)
  const float* rms_g;     // SEPI_RMS_RESIDUAL: [Cout] gain of the RMSNorm applied to the result, already times sqrt(Cout)
};

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BARRIER()                        \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// 8 fp32 (two 16-byte loads) -> the hi and lo f16 fragments (saturating, as conv3x3_split.hip)
__device__ __forceinline__ void split8(const u32x4& r0, const u32x4& r1, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a = __builtin_bit_cast(float, k < 2 ? r0[2 * k] : r1[2 * k - 4]);
    float b = __builtin_bit_cast(float, k < 2 ? r0[2 * k + 1] : r1[2 * k - 3]);
    a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(f32x2{a - hf[0], b - hf[1]}, f16x2);
    hi[k] = __builtin_bit_cast(unsigned, h);
    lo[k] = __builtin_bit_cast(unsigned, l);
  }
}

// RMS_IN: the convolution reads RMSNorm(x) (reference RMSNorm.forward model.py:206-207 ahead of to_qkv, :312 / :349): the gain and
// sqrt(C) are folded into the weights on the host, and the per-pixel 1 / max(||x||, 1e-12) - a wave sees every channel of its 32
// pixels on their way through the K loop - multiplies the accumulators in the epilogue.  SEPI_RMS_RESIDUAL (one n-tile: Cout ==
// 128): out = RMSNorm(acc + bias) * g + aux, the tail of LinearAttention.to_out and the block's residual add (:303, :703).
// Both remove a separate rms_norm pass (4 B read + 4 B written per element, + the residual read).
template <int EPI, bool RMS_IN>
__global__ __launch_bounds__(NT, 2) void conv1x1_split_kernel(Split1Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q16 = lane >> 4;

  const int n_tiles = p.Cout / BN;
  const int HWo = p.Hout * p.Wout;
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / KC;
  const int S = p.KH * p.KW * CC;
  // ---- persistent workgroup: tiles v = blockIdx.x, + gridDim.x, ... of the T = m-tiles x n-tiles (gridDim.x is a multiple of 8 or
  // equals T).  The XCD-aware renumbering (n-tiles of one m-tile back to back inside an XCD's contiguous band) is applied to v: the
  // tiles of one workgroup stay on its XCD's band.
  const int T = p.n_wg_tiles;
  auto tile_of = [&](int v, int& nt_, int& b_, int& p0_) {
    const int q = T >> 3, rem = T & 7, x = v & 7, k = v >> 3;
    const int wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
    nt_ = wg % n_tiles;
    const int mt = wg / n_tiles, tpi = HWo / BM;     // HWo % 256 == 0: a tile lies in one image (32-bit arithmetic throughout)
    b_ = mt / tpi;
    p0_ = (mt - b_ * tpi) * BM;
  };

  // ---- A staging: the wave's 32 pixel rows = 4 KiB = 4 wave-instructions per stage.  Instruction J covers rows 32 w + 8 J .. + 7;
  // lane L fills 16-byte position L & 7 of row 8 J + (L >> 3), which holds SOURCE chunk (L & 7) ^ (row & 7) (4 channels).
  // Per-lane byte offsets RELATIVE to the tile's first pixel (launcher: 256 % Wout == 0 or Wout % 256 == 0, so a tile is whole rows
  // or a piece of one and the offsets are the same for every tile), tap (0,0), channel chunk 0, for each source; everything that
  // changes per tile or per K-step is wave-uniform and rides in the scalar offset.
  const int a_chunk = (lane & 7) ^ (lane >> 3);
#define K_A1_DECL(J)                                                      \
  int a_b0##J, a_b1##J;                                                      \
  {                                                                          \
    const int op = wave * 32 + 8 * J + (lane >> 3);                          \
    const int oy = op / p.Wout, ox = op - oy * p.Wout;                       \
    const int pix = oy * p.stride * p.Win + ox * p.stride;                   \
    a_b0##J = (pix * p.ps0 + a_chunk * 4) * 4;                               \
    a_b1##J = (pix * p.ps1 + a_chunk * 4) * 4;                               \
  }
  K_A1_DECL(0) K_A1_DECL(1) K_A1_DECL(2) K_A1_DECL(3)
#undef K_A1_DECL
  const size_t img0 = (size_t)p.Hin * p.Win * p.ps0, img1 = (size_t)p.Hin * p.Win * p.ps1;
  const size_t w_tile_stride = (size_t)n_tiles * B_SLOT;

  // ---- issue stream: ONE continuous sequence of stages over all tiles of this workgroup, two stages ahead of the consumer - the
  // first two stages of the next tile are in flight while this tile's epilogue stores drain.  Issue-side state: tile (descriptors,
  // scalar base offsets), tap (ty, tx), channel chunk, ring slot, weight offset.
  int i_v = blockIdx.x, i_ty = 0, i_tx = 0, i_cc = 0, i_slot = 0, i_w = 0, i_base0 = 0, i_base1 = 0;
  __amdgpu_buffer_rsrc_t rs0, rs1, rsw;
  auto issue_tile = [&]() __attribute__((always_inline)) {
    int nt_, b_, p0_;
    tile_of(i_v < T ? i_v : 0, nt_, b_, p0_);
    const int oy = p0_ / p.Wout, ox = p0_ - oy * p.Wout;
    const int pix0 = oy * p.stride * p.Win + ox * p.stride;
    i_base0 = pix0 * p.ps0 * 4;
    i_base1 = pix0 * p.ps1 * 4;
    rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in0 + (size_t)b_ * img0), 0, (int)(img0 * 4), 0x00020000);
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? p.in1 + (size_t)b_ * img1 : p.in0), 0, p.in1 ? (int)(img1 * 4) : 0, 0x00020000);
    rsw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + (size_t)nt_ * B_SLOT), 0,
                                            (int)((size_t)(S - 1) * w_tile_stride + B_SLOT), 0x00020000);
    i_ty = i_tx = i_cc = 0;
    i_w = 0;
  };
  issue_tile();
  const int tid16 = tid * 16;
  auto dma = [&](__amdgpu_buffer_rsrc_t rs, char* dst, int voff, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, voff, soff, 0, 0);
  };
  // returns false when the stream is exhausted (nothing issued)
  auto issue = [&]() __attribute__((always_inline)) -> bool {
    if (i_v >= T) return false;
    const int c = i_cc * KC;
    const bool first = c < p.C0;
    const int soff = first ? i_base0 + ((i_ty * p.Win + i_tx) * p.ps0 + c) * 4 : i_base1 + ((i_ty * p.Win + i_tx) * p.ps1 + c - p.C0) * 4;
    char* st = smem + i_slot * STAGE;
    char* sa = st + wave * 4096;
    // (per-lane offsets selected with v_cndmask, the descriptors by two branches: see conv1x1_bf16.hip)
    const int v0 = first ? a_b00 : a_b10, v1 = first ? a_b01 : a_b11, v2 = first ? a_b02 : a_b12, v3 = first ? a_b03 : a_b13;
    if (first) { dma(rs0, sa, v0, soff); dma(rs0, sa + 1024, v1, soff); dma(rs0, sa + 2048, v2, soff); dma(rs0, sa + 3072, v3, soff); }
    else { dma(rs1, sa, v0, soff); dma(rs1, sa + 1024, v1, soff); dma(rs1, sa + 2048, v2, soff); dma(rs1, sa + 3072, v3, soff); }
    dma(rsw, st + A_BYTES + wave * 1024, tid16, i_w);                       // hi tile: 8 waves x 1 KiB
    dma(rsw, st + A_BYTES + B_TILE + wave * 1024, tid16, i_w + B_TILE);     // lo tile
    i_w += (int)w_tile_stride;
    i_slot = i_slot == RING - 1 ? 0 : i_slot + 1;
    if (++i_cc == CC) {
      i_cc = 0;
      if (++i_tx == p.KW) {
        i_tx = 0;
        if (++i_ty == p.KH) {                       // this tile's last stage is out: on to the workgroup's next tile
          i_v += gridDim.x;
          issue_tile();
        }
      }
    }
    return true;
  };

  f32x4 c00, c01, c02, c03, c04, c05, c06, c07, c10, c11, c12, c13, c14, c15, c16, c17;
  // fragment addresses.  Pixel P = 32 w + 16 mi + r16 (P & 7 = r16 & 7): channels 8 q16 .. + 7 are source chunks 2 q16, 2 q16 + 1
  const int sw = r16 & 7;
  const int a_row0 = (wave * 32 + r16) * 128, a_row1 = a_row0 + 16 * 128;
  const int a_c0 = ((2 * q16) ^ sw) << 4, a_c1 = ((2 * q16 + 1) ^ sw) << 4;
  const int b_base = A_BYTES + r16 * 64 + ((q16 ^ ((r16 >> 1) & 3)) << 4);       // weight row 16 J + r16, chunk q16 (the swizzle ignores J)
  int c_slot = 0;
  float ss0 = 0.f, ss1 = 0.f;                       // RMS_IN: sum of squares of the lane's 8 channels of its two pixels
  auto sumsq8 = [&](const u32x4& r0, const u32x4& r1, float acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned u0 = r0[k], u1 = r1[k];
      acc = __builtin_fmaf(__uint_as_float(u0), __uint_as_float(u0), acc);
      acc = __builtin_fmaf(__uint_as_float(u1), __uint_as_float(u1), acc);
    }
    return acc;
  };
  auto mma = [&](f32x4& c, const u32x4& wt, const u32x4& px) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wt), __builtin_bit_cast(f16x8, px), c, 0, 0, 0);
  };
  auto compute = [&]() __attribute__((always_inline)) {
    const char* st = smem + c_slot * STAGE;
    c_slot = c_slot == RING - 1 ? 0 : c_slot + 1;
    u32x4 ah0, al0, ah1, al1;
    const u32x4 x00 = *reinterpret_cast<const u32x4*>(st + a_row0 + a_c0), x01 = *reinterpret_cast<const u32x4*>(st + a_row0 + a_c1);
    const u32x4 x10 = *reinterpret_cast<const u32x4*>(st + a_row1 + a_c0), x11 = *reinterpret_cast<const u32x4*>(st + a_row1 + a_c1);
    if constexpr (RMS_IN) { ss0 = sumsq8(x00, x01, ss0); ss1 = sumsq8(x10, x11, ss1); }
    split8(x00, x01, ah0, al0);
    split8(x10, x11, ah1, al1);
#define K_COL(J, C0_, C1_)                                                                         \
  {                                                                                                \
    const u32x4 bh = *reinterpret_cast<const u32x4*>(st + b_base + J * 1024);                      \
    const u32x4 bl = *reinterpret_cast<const u32x4*>(st + B_TILE + b_base + J * 1024);             \
    mma(C0_, bh, al0); mma(C1_, bh, al1);                                                          \
    mma(C0_, bl, ah0); mma(C1_, bl, ah1);                                                          \
    mma(C0_, bh, ah0); mma(C1_, bh, ah1);                                                          \
  }
    K_COL(0, c00, c10) K_COL(1, c01, c11) K_COL(2, c02, c12) K_COL(3, c03, c13)
    K_COL(4, c04, c14) K_COL(5, c05, c15) K_COL(6, c06, c16) K_COL(7, c07, c17)
#undef K_COL
  };

  // ---- pipeline: stages g+1 and g+2 of the stream in flight while stage g is consumed (6 DMA instructions per wave and stage).
  // The epilogue's stores count in vmcnt as well; the next tile's first two stages are issued before them and are already landing
  // while they drain (see the wait below).
  bool more = issue();
  more = issue() && more;
  if (more) WAIT_VM(6); else WAIT_VM(0);
  BARRIER();
  for (int v = blockIdx.x; v < T; v += gridDim.x) {
  int nt, b, p0;
  tile_of(v, nt, b, p0);
  c00 = 0; c01 = 0; c02 = 0; c03 = 0; c04 = 0; c05 = 0; c06 = 0; c07 = 0;
  c10 = 0; c11 = 0; c12 = 0; c13 = 0; c14 = 0; c15 = 0; c16 = 0; c17 = 0;
  ss0 = 0.f; ss1 = 0.f;
  for (int s = 0; s < S; ++s) {
    const bool issued = issue();
    compute();
    // stage g+1 has landed (this wave's part; the barrier covers the rest) once only the requests younger than it are outstanding:
    // the 6 of stage g+2 - and, in the first K-step of a later tile, the previous tile's 16 epilogue stores, which were issued
    // after stage g+1 (vmcnt retires vector-memory requests of a wave in issue order, loads and stores alike, on gfx9-family
    // parts).  Without the + 16 this wait would drain the stores before the tile's second K-step could start; with it the drain
    // overlaps two K-steps and the loads of the next two stages.
    if (s == 0 && v != (int)blockIdx.x) { if (issued) WAIT_VM(22); else WAIT_VM(16); }
    else if (issued) WAIT_VM(6);
    else WAIT_VM(0);
    BARRIER();
  }

  // ------------------------------- epilogue (register-direct, fp32) --------------------------
  // lane (r16, g = q16), pixel block mi, weight block J, register e  ->  pixel wave * 32 + mi * 16 + r16, channel 32 g + 4 J + e
  const int n0 = nt * BN;
  const float ws = p.w_inv_scale;
  // all bias vectors (and, per pixel block, all residual vectors) are loaded AHEAD of the first store: stores count in vmcnt too, so
  // a load issued behind a store would be waited for together with that store's completion
  f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0, bs2 = bs0, bs3 = bs0, bs4 = bs0, bs5 = bs0, bs6 = bs0, bs7 = bs0;
  if (p.bias) {
    const float* bp = p.bias + n0 + q16 * 32;
    bs0 = *reinterpret_cast<const f32x4*>(bp); bs1 = *reinterpret_cast<const f32x4*>(bp + 4);
    bs2 = *reinterpret_cast<const f32x4*>(bp + 8); bs3 = *reinterpret_cast<const f32x4*>(bp + 12);
    bs4 = *reinterpret_cast<const f32x4*>(bp + 16); bs5 = *reinterpret_cast<const f32x4*>(bp + 20);
    bs6 = *reinterpret_cast<const f32x4*>(bp + 24); bs7 = *reinterpret_cast<const f32x4*>(bp + 28);
  }
  asm volatile("" : "+v"(bs0), "+v"(bs1), "+v"(bs2), "+v"(bs3), "+v"(bs4), "+v"(bs5), "+v"(bs6), "+v"(bs7));
  size_t o0, o1;
#define K_OFF(MI, O_)                                                                                                  \
  {                                                                                                                    \
    const int op = p0 + wave * 32 + MI * 16 + r16;                                                                     \
    if (EPI == SEPI_PS_SILU) {                                                                                         \
      const int CoutPS = p.Cout >> 2, ij = n0 / CoutPS, ch0 = n0 - ij * CoutPS;                                        \
      const int oy = op / p.Wout, ox = op - oy * p.Wout;                                                               \
      O_ = ((size_t)(b * 2 * p.Hout + 2 * oy + (ij >> 1)) * (2 * p.Wout) + 2 * ox + (ij & 1)) * CoutPS + ch0 + q16 * 32; \
    } else {                                                                                                           \
      O_ = ((size_t)b * HWo + op) * p.Cout + n0 + q16 * 32;                                                            \
    }                                                                                                                  \
  }
  K_OFF(0, o0) K_OFF(1, o1)
#undef K_OFF
  float ws0 = ws, ws1 = ws;
  if constexpr (RMS_IN) {
    // the four lanes of a pixel (r16, q16 = 0..3) hold its channels 8 q16 .. of every chunk: sum over them, then F.normalize's
    // x / max(||x||, 1e-12)
    ss0 = xor32_sum(xor16_sum(ss0));
    ss1 = xor32_sum(xor16_sum(ss1));
    ws0 = ws / fmaxf(sqrtf(ss0), 1e-12f);
    ws1 = ws / fmaxf(sqrtf(ss1), 1e-12f);
  }
  f32x4 v0_[8] = {c00 * ws0 + bs0, c01 * ws0 + bs1, c02 * ws0 + bs2, c03 * ws0 + bs3, c04 * ws0 + bs4, c05 * ws0 + bs5, c06 * ws0 + bs6, c07 * ws0 + bs7};
  f32x4 v1_[8] = {c10 * ws1 + bs0, c11 * ws1 + bs1, c12 * ws1 + bs2, c13 * ws1 + bs3, c14 * ws1 + bs4, c15 * ws1 + bs5, c16 * ws1 + bs6, c17 * ws1 + bs7};
  if (EPI == SEPI_RMS_RESIDUAL) {
    // RMSNorm over the pixel's 128 output channels (one n-tile): 32 of them in this lane, the rest in the pixel's other three lanes
    f32x4 r0_[8], r1_[8], g_[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
      r0_[J] = *reinterpret_cast<const f32x4*>(p.aux + o0 + 4 * J);
      r1_[J] = *reinterpret_cast<const f32x4*>(p.aux + o1 + 4 * J);
      g_[J] = *reinterpret_cast<const f32x4*>(p.rms_g + n0 + q16 * 32 + 4 * J);
    }
    asm volatile("" : "+v"(r0_[0]), "+v"(r0_[1]), "+v"(r0_[2]), "+v"(r0_[3]), "+v"(r0_[4]), "+v"(r0_[5]), "+v"(r0_[6]), "+v"(r0_[7]));
    asm volatile("" : "+v"(r1_[0]), "+v"(r1_[1]), "+v"(r1_[2]), "+v"(r1_[3]), "+v"(r1_[4]), "+v"(r1_[5]), "+v"(r1_[6]), "+v"(r1_[7]));
    asm volatile("" : "+v"(g_[0]), "+v"(g_[1]), "+v"(g_[2]), "+v"(g_[3]), "+v"(g_[4]), "+v"(g_[5]), "+v"(g_[6]), "+v"(g_[7]));
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { t0 = __builtin_fmaf(v0_[J][e], v0_[J][e], t0); t1 = __builtin_fmaf(v1_[J][e], v1_[J][e], t1); }
    }
    t0 = xor32_sum(xor16_sum(t0));
    t1 = xor32_sum(xor16_sum(t1));
    const float i0 = 1.0f / fmaxf(sqrtf(t0), 1e-12f), i1 = 1.0f / fmaxf(sqrtf(t1), 1e-12f);
#pragma unroll
    for (int J = 0; J < 8; ++J) { v0_[J] = v0_[J] * i0 * g_[J] + r0_[J]; v1_[J] = v1_[J] * i1 * g_[J] + r1_[J]; }
  }
  if (EPI == SEPI_GNTAIL) {
    // out = conv(x) + silu(a[b][c] * h + b[b][c]): the second GroupNorm + SiLU of a ResnetBlock and its residual add folded into the
    // 1x1 res_conv (reference model.py:250-259, :283-285); h may alias out (a lane reads exactly the addresses it writes)
    // (block 0's tail operand, then block 1's in the same registers: all of it ahead of the first store, which comes at the very end)
    f32x4 r_[8], a_[8], b_[8];
    const float* ga = p.gn_a + (size_t)b * p.Cout + n0 + q16 * 32;
    const float* gb = p.gn_b + (size_t)b * p.Cout + n0 + q16 * 32;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
      r_[J] = *reinterpret_cast<const f32x4*>(p.aux + o0 + 4 * J);
      a_[J] = *reinterpret_cast<const f32x4*>(ga + 4 * J);
      b_[J] = *reinterpret_cast<const f32x4*>(gb + 4 * J);
    }
    asm volatile("" : "+v"(r_[0]), "+v"(r_[1]), "+v"(r_[2]), "+v"(r_[3]), "+v"(r_[4]), "+v"(r_[5]), "+v"(r_[6]), "+v"(r_[7]));
    asm volatile("" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(a_[4]), "+v"(a_[5]), "+v"(a_[6]), "+v"(a_[7]));
    asm volatile("" : "+v"(b_[0]), "+v"(b_[1]), "+v"(b_[2]), "+v"(b_[3]), "+v"(b_[4]), "+v"(b_[5]), "+v"(b_[6]), "+v"(b_[7]));
#pragma unroll
    for (int J = 0; J < 8; ++J) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v0_[J][e] += silu<true>(__builtin_fmaf(a_[J][e], r_[J][e], b_[J][e]));
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) r_[J] = *reinterpret_cast<const f32x4*>(p.aux + o1 + 4 * J);
    asm volatile("" : "+v"(r_[0]), "+v"(r_[1]), "+v"(r_[2]), "+v"(r_[3]), "+v"(r_[4]), "+v"(r_[5]), "+v"(r_[6]), "+v"(r_[7]));
#pragma unroll
    for (int J = 0; J < 8; ++J) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v1_[J][e] += silu<true>(__builtin_fmaf(a_[J][e], r_[J][e], b_[J][e]));
    }
  }
  if (EPI == SEPI_RESIDUAL) {
    f32x4 r0_[8], r1_[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
      r0_[J] = *reinterpret_cast<const f32x4*>(p.aux + o0 + 4 * J);
      r1_[J] = *reinterpret_cast<const f32x4*>(p.aux + o1 + 4 * J);
    }
    asm volatile("" : "+v"(r0_[0]), "+v"(r0_[1]), "+v"(r0_[2]), "+v"(r0_[3]), "+v"(r0_[4]), "+v"(r0_[5]), "+v"(r0_[6]), "+v"(r0_[7]));
    asm volatile("" : "+v"(r1_[0]), "+v"(r1_[1]), "+v"(r1_[2]), "+v"(r1_[3]), "+v"(r1_[4]), "+v"(r1_[5]), "+v"(r1_[6]), "+v"(r1_[7]));
#pragma unroll
    for (int J = 0; J < 8; ++J) { v0_[J] += r0_[J]; v1_[J] += r1_[J]; }
  }
  if (EPI == SEPI_PS_SILU) {
#pragma unroll
    for (int J = 0; J < 8; ++J) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0_[J][e] = silu<true>(v0_[J][e]); v1_[J][e] = silu<true>(v1_[J][e]); }
    }
  }
  {
    // Stores go through a 2 KB per-wave LDS staging area so that every store instruction writes FULL 128-byte lines.  Straight from
    // the accumulators a lane owns one 128-byte segment (32 channels of one pixel) and needs eight 16-byte stores for it: every
    // instruction touches 64 different lines, 16 bytes each - and the stores, not the loads or the MFMAs, were what bounded this
    // kernel (diagnostic builds - knobs in profiles/r6/pricing_and_ab_knobs.patch -, profiles/r6/conv1x1_split_phase_diagnostics.txt: 128 -> 128 @256x256 0.390 ms, without the stores
    // 0.206 ms, without the MFMAs 0.362 ms, without the pixel-row DMAs 0.353 ms).  Per pixel block, in four rounds: the 16 lanes
    // holding channel quarter g' write their 16 pixels x 128 B into the staging area, all 64 lanes read it back 16 bytes at a time
    // in memory order (8 lanes per pixel line) and store.  Wave-private: LDS executes a wave's instructions in order, no barrier.
    char* const stg = smem + RING * STAGE + wave * STG_BYTES;
    const int l8 = lane >> 3, c8 = lane & 7;                      // read-back: pixel row l8 (+ 8 for the second half), 16-byte chunk c8
    auto pixel_out = [&](int mi, int row) -> size_t {             // element offset of channel n0 of pixel `row` of block mi
      const int op = p0 + wave * 32 + mi * 16 + row;
      if (EPI == SEPI_PS_SILU) {
        const int CoutPS = p.Cout >> 2, ij = n0 / CoutPS, ch0 = n0 - ij * CoutPS;
        const int oy = op / p.Wout, ox = op - oy * p.Wout;
        return ((size_t)(b * 2 * p.Hout + 2 * oy + (ij >> 1)) * (2 * p.Wout) + 2 * ox + (ij & 1)) * CoutPS + ch0;
      }
      return ((size_t)b * HWo + op) * p.Cout + n0;
    };
    const size_t pa0 = pixel_out(0, l8), pa1 = pixel_out(0, l8 + 8), pb0 = pixel_out(1, l8), pb1 = pixel_out(1, l8 + 8);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (q16 == gq) {
#pragma unroll
          for (int J = 0; J < 8; ++J) *reinterpret_cast<f32x4*>(stg + r16 * 128 + J * 16) = mi == 0 ? v0_[J] : v1_[J];
        }
        // compiler-level fence: to the optimiser a lane that skipped the branch has not written the staging area, so it may reuse
        // the previous round's read-back (it did: the reads were sunk INTO the branch and three quarters of the channels came out
        // stale - caught by the kernel tests).  The hardware needs nothing here: LDS executes a wave's instructions in order.
        asm volatile("" ::: "memory");
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(stg + lane * 16);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(stg + 1024 + lane * 16);
        asm volatile("" ::: "memory");
        *reinterpret_cast<f32x4*>(p.out + (mi == 0 ? pa0 : pb0) + gq * 32 + c8 * 4) = w0;
        *reinterpret_cast<f32x4*>(p.out + (mi == 0 ? pa1 : pb1) + gq * 32 + c8 * 4) = w1;
      }
    }
  }
  }   // tiles of this workgroup
}

}  // namespace

bool conv1x1_split_eligible(const ConvArgs& a) {
  if (a.pad != 0 || a.stride < 1 || a.KH < 1 || a.KW < 1) return false;
  if ((a.Hout - 1) * a.stride + a.KH > a.Hin || (a.Wout - 1) * a.stride + a.KW > a.Win) return false;   // the gather stays inside
  if (a.ps0 % 4 || (a.C1 && (a.ps1 % 4 || a.ps1 < a.C1))) return false;                              // 16-byte aligned rows
  // a pixel stride below the channel count = the overlapping-window view of the 7x7 input convolution (kernels.hpp): one source,
  // and the last output pixel's window must stay inside its row
  if (a.ps0 != a.C0 && (a.C1 || (long)(a.Win - (a.Wout - 1) * a.stride - a.KW) * a.ps0 + a.ps0 < a.C0)) return false;
  if (a.C1 && (a.KH != 1 || a.KW != 1)) return false;
  if (a.C0 % KC || a.C1 % KC || a.Cout % BN || a.Cout != a.CoutPad) return false;
  if (((long)a.Hout * a.Wout) % BM) return false;
  if (BM % a.Wout != 0 && a.Wout % BM != 0) return false;          // a tile is whole output rows or a piece of one (tile-relative offsets)
  if (a.gn_partial || a.out_q || a.eps4) return false;
  if (a.gn_res_src && (!a.gn_res_a || !a.gn_res_b || a.residual || a.rms_in || a.rms_out_g || a.mode != CONV_PLAIN)) return false;
  if (a.rms_out_g && (a.Cout != BN || !a.residual || a.mode != CONV_PLAIN || a.rms_in)) return false;
  if (a.rms_in && (a.mode != CONV_PLAIN || a.residual)) return false;
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU && ((a.Cout / 4) % BN || a.residual)) return false;
  if (a.mode != CONV_PLAIN && a.mode != CONV_PIXEL_SHUFFLE_SILU) return false;
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.ps0, a.ps1) * 4 >= (1ull << 31)) return false;
  if ((size_t)a.KH * a.KW * ((a.C0 + a.C1) / KC) * (a.Cout / BN) * B_SLOT >= (1ull << 31)) return false;
  return true;
}

// fp32 [tap][Cout][Cin] (pack_conv_weights' order incl. its pixel-shuffle column permutation) -> [tap][cc][ntile][hi tile | lo tile];
// a tile = 128 rows x 64 B, XOR-swizzled, row 16 J + 4 g + e holding output channel 32 g + 4 J + e of the n-tile
void pack_conv1x1_split(const float* src_tap_o_i, int taps, int Cin, int Cout, float scale, std::vector<unsigned short>& out) {
  const int CC = Cin / KC, NTL = Cout / BN;
  out.assign((size_t)taps * CC * NTL * 2 * BN * KC, 0);
  for (int tap = 0; tap < taps; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned short* hi_t = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * 2 * BN * KC;
        unsigned short* lo_t = hi_t + BN * KC;
        for (int n = 0; n < BN; ++n) {
          const int J = n >> 4, g = (n >> 2) & 3, e = n & 3;
          const int o = nt * BN + 32 * g + 4 * J + e;
          for (int c = 0; c < 4; ++c) {
            const int cs = c ^ ((n >> 1) & 3);
            for (int k = 0; k < 8; ++k) {
              const float v = src_tap_o_i[((size_t)tap * Cout + o) * Cin + cc * KC + c * 8 + k] * scale;
              split_halves_host(v, true, &hi_t[n * KC + cs * 8 + k], &lo_t[n * KC + cs * 8 + k]);
            }
          }
        }
      }
}

int conv1x1_split(const ConvArgs& a, const void* packed_w, float w_inv_scale, hipStream_t st) {
  if (!conv1x1_split_eligible(a)) SRGD_FAIL("conv1x1_split: shape not eligible");
  if (a.bias && ((size_t)a.bias & 15)) SRGD_FAIL("conv1x1_split: the bias array must be 16-byte aligned");
  Split1Args p;
  p.in0 = (const float*)a.in0; p.in1 = (const float*)a.in1; p.C0 = a.C0; p.C1 = a.C1;
  p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Hout = a.Hout; p.Wout = a.Wout;
  p.KH = a.KH; p.KW = a.KW; p.stride = a.stride; p.ps0 = a.ps0; p.ps1 = a.C1 ? a.ps1 : 0;
  p.w = packed_w; p.bias = a.bias; p.w_inv_scale = w_inv_scale; p.Cout = a.Cout; p.out = (float*)a.out;
  p.aux = a.gn_res_src ? (const float*)a.gn_res_src : (const float*)a.residual;
  p.gn_a = a.gn_res_a; p.gn_b = a.gn_res_b;
  p.rms_g = a.rms_out_g;
  if (a.rms_out_g && (a.Cout != BN || !a.residual || a.mode != CONV_PLAIN)) SRGD_FAIL("conv1x1_split: the RMSNorm tail needs Cout == 128 and the residual tensor");
  const long tiles = (long)a.B * a.Hout * a.Wout / BM * (a.Cout / BN);
  if (tiles <= 0 || tiles > 0x7fffffffL) SRGD_FAIL("conv1x1_split: bad grid");
  p.n_wg_tiles = (int)tiles;
  // persistent grid: one workgroup per CU (144 KB of LDS each), a multiple of 8 so that a workgroup's tiles stay on its XCD's band
  static int cus[64] = {};
  int dev = 0;
  SRGD_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && cus[dev] == 0) {
    int n = 0;
    SRGD_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    __atomic_store_n(&cus[dev], n > 8 ? n & ~7 : 8, __ATOMIC_RELAXED);
  }
  const int n_cu = (dev >= 0 && dev < 64) ? cus[dev] : 256;
  const long grid = tiles <= n_cu ? tiles : n_cu;
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define K_SET(E_, R_) SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<E_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    K_SET(SEPI_PLAIN, false) K_SET(SEPI_RESIDUAL, false) K_SET(SEPI_PS_SILU, false) K_SET(SEPI_RMS_RESIDUAL, false) K_SET(SEPI_PLAIN, true)
    K_SET(SEPI_GNTAIL, false)
#undef K_SET
    once.done();
  }
#define K_GO(E_, R_) hipLaunchKernelGGL((conv1x1_split_kernel<E_, R_>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, st, p)
  if (a.rms_in) {
    if (a.mode != CONV_PLAIN || a.residual || a.rms_out_g) SRGD_FAIL("conv1x1_split: the RMSNorm-on-input form has the plain epilogue only");
    K_GO(SEPI_PLAIN, true);
  } else if (a.mode == CONV_PIXEL_SHUFFLE_SILU) K_GO(SEPI_PS_SILU, false);
  else if (a.gn_res_src) K_GO(SEPI_GNTAIL, false);
  else if (a.rms_out_g) K_GO(SEPI_RMS_RESIDUAL, false);
  else if (a.residual) K_GO(SEPI_RESIDUAL, false);
  else K_GO(SEPI_PLAIN, false);
#undef K_GO
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
