// Pointwise (1x1) convolutions on the block-scaled MX matrix cores of gfx950 (MI355X), fp8 mode (BASELINE configs[4]:
// "fp8 (CDNA4 MFMA) conv + attention weights"): the ResnetBlock residual projection with the GroupNorm2 + SiLU tail and the
// residual add in its epilogue (reference model.py:271, :250-259, :285), the PixelShuffleUpsample 1x1 conv + SiLU +
// PixelShuffle (:70-98), the Downsample space-to-depth + 1x1 conv (:106-110, as a 2x2 / stride-2 gather) and to_out of the
// softmax attention sites (:341-342) - wherever the layer's input tensor already has an MX-fp8 twin (engine.hip: the 3x3
// convolutions of fp8 mode read the same tensors, so their producers write the twins anyway).
//
//   v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3, one E8M0 scale per 32 K-elements on both operands, fp32 accumulate
//   (operand maps: conv3x3_mxfp8.hip).  What it buys over conv1x1_bf16.hip: the K-heavy layers (1536 -> 1024 @32^2 ...) run on
//   the 2x matrix rate with half the LDS bytes per FLOP (they are LDS-/MFMA-bound at 0.7-0.9 PFLOP/s in bf16), the streaming
//   layers (256 -> 128 @256^2 ...) read 1.03 B instead of 2 B per input element.
//
// Structure = conv1x1_bf16.hip with 128-channel K-steps (tile shapes: QShape below):
//   * 2 threads per output pixel, waves 2 along N x (2 or 4) along M, wave tile 64 x 64 = 4 x 4 MFMA blocks (64 accumulator
//     registers) - the same D layout as the bf16 kernel, so the epilogue (conv1x1_epilogue.hpp) is shared;
//   * every K-step is ONE stage - A: pixel rows x 128 B of e4m3 + 4 scale bytes per pixel, B: one pre-swizzled 16 KiB weight
//     tile + 512 scale bytes - brought in by LDS-DMA into a ring, counted s_waitcnt + one raw barrier per step; what changes
//     from step to step rides in SGPRs (incremental issue stream, conv1x1_bf16.hip);
//   * 128-byte LDS rows XOR-swizzled (chunk ^= row & 6) as in conv3x3_mxfp8.hip: conflict-free ds_read_b128 fragments;
//   * MFMAs as inline asm with the accumulator tied (conv3x3_mxfp8.hip explains why).
#include <cmath>
#include <cstdlib>

#include "conv1x1_epilogue.hpp"

namespace srgd {
namespace {

constexpr int BNQ = 128, KQ = 128;
constexpr int BQ_TILE = BNQ * KQ;                  // 16 KiB of e4m3 weights per K-step
constexpr int BQ_BYTES = BQ_TILE + 512;            // + 512 scale bytes laid out [wn][r16][g][J] (one dword per lane)
// Tile shapes (template parameter BM = pixels per tile; 2 threads per pixel, wave tile 64 x 64 either way):
//   BM = 256: 8 waves, 49.5 KiB stages, 3-deep ring = 148.5 KiB -> ONE workgroup per CU, two stages always in flight
//   BM = 128: 4 waves, 33 KiB stages, 2-deep ring = 66 KiB -> TWO workgroups per CU (a tile's epilogue overlaps the other
//             workgroup's loads - what the streaming layers need), weights re-streamed per 128 pixels instead of 256
template <int BM> struct QShape {
  static constexpr int NT = BM * 2, NW = BM / 32;
  static constexpr int A_BYTES = BM * KQ;                     // e4m3 pixel rows
  static constexpr int AS_BYTES = BM * 4;                     // 4 scale bytes per pixel
  static constexpr int STAGE = A_BYTES + AS_BYTES + BQ_BYTES; // 50,688 / 33,792
  static constexpr int RING = BM == 256 ? 3 : 2;
  static constexpr int LDS = RING * STAGE;                    // 152,064 / 67,584
  static constexpr int A_PER_WAVE = (A_BYTES / 1024) / NW;    // 4 / 4 DMA pieces of 1 KiB per wave and stage
  static constexpr int B_PER_WAVE = 16 / NW;                  // 2 / 4
  static constexpr int DMA_PER_STAGE = A_PER_WAVE + 1 + B_PER_WAVE + 1;   // 8 / 10 (the counted vmcnt waits)
  static_assert(LDS >= BM * EPI_ROW && LDS <= 160 * 1024, "conv1x1_mxfp8: LDS budget");
};

typedef __attribute__((address_space(3))) void* lds_ptrq;
typedef int v8iq __attribute__((ext_vector_type(8)));
typedef int v4iq __attribute__((ext_vector_type(4)));

struct Conv1QArgs {
  const unsigned char* q0; const unsigned char* s0; int C0;     // MX-fp8 source 0: e4m3 [B,Hin,Win,C0], E8M0 [B,Hin,Win,C0/32]
  const unsigned char* q1; const unsigned char* s1; int C1;     // optional source 1 (channel concat; 1x1 taps only)
  int B, Hin, Win, Hout, Wout;
  int KH, KW, stride;     // gather taps: 1x1, or 2x2 / stride 2 (space-to-depth folded into the K walk)
  const unsigned char* w; // packed [tap][cc][ntile][16.5 KiB]
  const float* bias;
  int Cout;
  bf16* out;
  const bf16* aux;        // EPI_RESIDUAL: tensor added to the output; EPI_GNTAIL: tensor the GroupNorm tail is applied to
  const float* gn_a; const float* gn_b;
  unsigned char* oq; unsigned char* os;
  float* eps4; const float* fin_w; const float* fin_b;
};

#define WAIT_VMQ(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BARRIERQ()                       \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

template <int EPI, int BM>
__global__ __launch_bounds__(BM * 2, 2) void conv1x1_mxfp8_kernel(Conv1QArgs p) {
  using Q = QShape<BM>;
  constexpr int BMQ = BM, NW = Q::NW, AQ_BYTES = Q::A_BYTES, ASQ_BYTES = Q::AS_BYTES, STAGEQ = Q::STAGE, RINGQ = Q::RING;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;

  const int n_tiles = p.Cout / BNQ;
  const int HWo = p.Hout * p.Wout;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, x = wg & 7, k = wg >> 3;
    wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
  }
  const int nt = wg % n_tiles;
  const int mt = wg / n_tiles;
  const long m0 = (long)mt * BMQ;                  // first output pixel of the tile (HWo % 256 == 0: one image)
  const int b = (int)(m0 / HWo);
  const int p0 = (int)(m0 - (long)b * HWo);
  const int CC0 = p.C0 / KQ, CC = (p.C0 + p.C1) / KQ;
  const int S = p.KH * p.KW * CC;

  // ---- A staging: 32 pieces of 1 KiB per stage (8 pixel rows of 128 B each), wave w issues pieces w, w+8, w+16, w+24: lane l
  // of piece j fills stored chunk l & 7 of row P = j * 8 + (l >> 3) with logical chunk (l & 7) ^ (P & 6); (j * 8) & 6 == 0, so the
  // source chunk is the same for the four pieces.  Scales: 4 B per pixel, one dword per lane, waves 0..3 cover the 256 pixels
  // (waves 4..7 repeat them: identical bytes, and every wave issues the same eight instructions).
  // Per-lane byte offsets (tap (0,0), channel chunk 0) per source; everything that changes from K-step to K-step (tap offset,
  // channel chunk, ring slot, weight unit) is wave-uniform and rides in SGPRs (scalar buffer offset, M0), advanced incrementally.
  const int a_sub = (lane & 7) ^ ((lane >> 3) & 6);
  auto in_pix = [&](int P) {                        // input pixel offset (tap (0,0)) of output pixel p0 + P
    const int op = p0 + P;
    const int oy = op / p.Wout, ox = op - oy * p.Wout;
    return oy * p.stride * p.Win + ox * p.stride;
  };
  static_assert(Q::A_PER_WAVE == 4, "four 1 KiB pixel pieces per wave and stage");
  const int pix0 = in_pix(wave * 8 + (lane >> 3)), pix1 = in_pix((wave + NW) * 8 + (lane >> 3)),
            pix2 = in_pix((wave + 2 * NW) * 8 + (lane >> 3)), pix3 = in_pix((wave + 3 * NW) * 8 + (lane >> 3));
  constexpr int SW = BM / 64;                        // waves that cover the tile's scale dwords (the others repeat them)
  const int s_pix = in_pix((wave % SW) * 64 + lane);
  const int a00 = pix0 * p.C0 + a_sub * 16, a01 = pix1 * p.C0 + a_sub * 16, a02 = pix2 * p.C0 + a_sub * 16, a03 = pix3 * p.C0 + a_sub * 16;
  // source 1 as a per-lane DIFFERENCE to source 0: offset = a0x + (d1x & mask) with a wave-uniform mask - written as a select
  // between two sets of registers hipcc selects between their ADDRESSES and parks both sets in scratch
  const int d10 = pix0 * (p.C1 - p.C0), d11 = pix1 * (p.C1 - p.C0), d12 = pix2 * (p.C1 - p.C0), d13 = pix3 * (p.C1 - p.C0);
  const int as0 = s_pix * (p.C0 / 32), ds1 = s_pix * ((p.C1 - p.C0) / 32);
  const size_t npix = (size_t)p.Hin * p.Win;
  // per-source base pointers and sizes; the buffer descriptor of a K-step's source is BUILT from a scalar select of these (a
  // select between two ready-made descriptors goes through scratch memory and a waterfall loop in hipcc's hands)
  const unsigned char* const qb0 = p.q0 + (size_t)b * npix * p.C0;
  const unsigned char* const sb0 = p.s0 + (size_t)b * npix * (p.C0 / 32);
  const unsigned char* const qb1 = p.q1 ? p.q1 + (size_t)b * npix * p.C1 : qb0;
  const unsigned char* const sb1 = p.s1 ? p.s1 + (size_t)b * npix * (p.C1 / 32) : sb0;
  const int qn0 = (int)(npix * p.C0), qn1 = p.q1 ? (int)(npix * p.C1) : 0;
  const size_t w_step_stride = (size_t)n_tiles * BQ_BYTES;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.w + (size_t)nt * BQ_BYTES), 0, (int)((size_t)(S - 1) * w_step_stride + BQ_BYTES), 0x00020000);

  int i_ty = 0, i_tx = 0, i_cc = 0, i_slot = 0, i_w = 0;       // issue stream: the next K-step to request
  const int lane16 = lane * 16, lane4 = lane * 4;
  auto issue = [&]() __attribute__((always_inline)) {
    const bool first = i_cc < CC0;
    const int toff = i_ty * p.Win + i_tx;
    const int soff = first ? toff * p.C0 + i_cc * KQ : toff * p.C1 + (i_cc - CC0) * KQ;
    const int ssoff = first ? toff * (p.C0 / 32) + i_cc * 4 : toff * (p.C1 / 32) + (i_cc - CC0) * 4;
    char* st = smem + i_slot * STAGEQ;
    const int m1 = first ? 0 : -1;
    const int v0 = a00 + (d10 & m1), v1 = a01 + (d11 & m1), v2 = a02 + (d12 & m1), v3 = a03 + (d13 & m1), vs = as0 + (ds1 & m1);
#define K_DMAQ(RS_, DST_, VO_, SO_, SZ_) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_, (lds_ptrq)(DST_), SZ_, VO_, SO_, 0, 0)
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? qb0 : qb1), 0, first ? qn0 : qn1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? sb0 : sb1), 0, (first ? qn0 : qn1) >> 5, 0x00020000);
    K_DMAQ(rq, st + wave * 1024, v0, soff, 16);
    K_DMAQ(rq, st + (wave + NW) * 1024, v1, soff, 16);
    K_DMAQ(rq, st + (wave + 2 * NW) * 1024, v2, soff, 16);
    K_DMAQ(rq, st + (wave + 3 * NW) * 1024, v3, soff, 16);
    K_DMAQ(rs, st + AQ_BYTES + (wave % SW) * 256, vs, ssoff, 4);
    char* sb = st + AQ_BYTES + ASQ_BYTES;
#pragma unroll
    for (int j = 0; j < Q::B_PER_WAVE; ++j) K_DMAQ(rsw, sb + (wave + NW * j) * 1024, lane16, i_w + (wave + NW * j) * 1024, 16);
    K_DMAQ(rsw, sb + BQ_TILE + (wave & 1) * 256, lane4, i_w + BQ_TILE + (wave & 1) * 256, 4);
#undef K_DMAQ
    i_w += (int)w_step_stride;
    i_slot = i_slot == RINGQ - 1 ? 0 : i_slot + 1;
    if (++i_cc == CC) {
      i_cc = 0;
      if (++i_tx == p.KW) { i_tx = 0; ++i_ty; }
    }
  };

  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0,
        c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;
  // operand addresses: A row P = wm * 64 + i * 16 + r16 -> (P & 6) == (r16 & 6); B row n = wn * 64 + j * 16 + r16 likewise:
  // ONE per-lane base each, the block index rides in the ds_read offset field.  Logical chunks g and 4 + g (address ^ 64).
  const int sw = (g ^ (r16 & 6)) << 4;
  const int aa = (wm * 64 + r16) * 128 + sw;
  const int asb = AQ_BYTES + (wm * 64 + r16) * 4 + g;
  const int bb = AQ_BYTES + ASQ_BYTES + (wn * 64 + r16) * 128 + sw;
  const int bsb = AQ_BYTES + ASQ_BYTES + BQ_TILE + ((wn * 16 + r16) * 4 + g) * 4;
  int c_slot = 0;                                  // ring slot of the K-step being consumed
  auto compute = [&]() __attribute__((always_inline)) {
    const char* st = smem + c_slot * STAGEQ;
    c_slot = c_slot == RINGQ - 1 ? 0 : c_slot + 1;
    v8iq a0, a1, a2, a3, b0, b1, b2, b3;
    int sa0, sa1, sa2, sa3;
#define K_LOADQ(DST_, BASE_, I_)                                                     \
    {                                                                                   \
      const v4iq lo = *reinterpret_cast<const v4iq*>(st + BASE_ + I_ * 2048);           \
      const v4iq hi = *reinterpret_cast<const v4iq*>(st + (BASE_ ^ 64) + I_ * 2048);    \
      DST_ = v8iq{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};              \
    }
    K_LOADQ(b0, bb, 0) K_LOADQ(b1, bb, 1) K_LOADQ(b2, bb, 2) K_LOADQ(b3, bb, 3)
    const int sbw = *reinterpret_cast<const int*>(st + bsb);
    K_LOADQ(a0, aa, 0) K_LOADQ(a1, aa, 1) K_LOADQ(a2, aa, 2) K_LOADQ(a3, aa, 3)
#undef K_LOADQ
    sa0 = *reinterpret_cast<const unsigned char*>(st + asb);
    sa1 = *reinterpret_cast<const unsigned char*>(st + asb + 64);
    sa2 = *reinterpret_cast<const unsigned char*>(st + asb + 128);
    sa3 = *reinterpret_cast<const unsigned char*>(st + asb + 192);
    // opsel of the weight scale (byte J of sbw): bit 0 -> op_sel[1], bit 1 -> op_sel_hi[1]
#define K_QMM1_OPSEL_0 "op_sel_hi:[0,0,0]"
#define K_QMM1_OPSEL_1 "op_sel:[0,1,0] op_sel_hi:[0,0,0]"
#define K_QMM1_OPSEL_2 "op_sel_hi:[0,1,0]"
#define K_QMM1_OPSEL_3 "op_sel:[0,1,0] op_sel_hi:[0,1,0]"
#define K_QMM1(C_, A_, SA_, B_, J_)                                                                     \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " K_QMM1_OPSEL_##J_           \
                 : "+v"(C_) : "v"(A_), "v"(B_), "v"(SA_), "v"(sbw))
    K_QMM1(c00, a0, sa0, b0, 0); K_QMM1(c01, a0, sa0, b1, 1); K_QMM1(c02, a0, sa0, b2, 2); K_QMM1(c03, a0, sa0, b3, 3);
    K_QMM1(c10, a1, sa1, b0, 0); K_QMM1(c11, a1, sa1, b1, 1); K_QMM1(c12, a1, sa1, b2, 2); K_QMM1(c13, a1, sa1, b3, 3);
    K_QMM1(c20, a2, sa2, b0, 0); K_QMM1(c21, a2, sa2, b1, 1); K_QMM1(c22, a2, sa2, b2, 2); K_QMM1(c23, a2, sa2, b3, 3);
    K_QMM1(c30, a3, sa3, b0, 0); K_QMM1(c31, a3, sa3, b1, 1); K_QMM1(c32, a3, sa3, b2, 2); K_QMM1(c33, a3, sa3, b3, 3);
#undef K_QMM1
#undef K_QMM1_OPSEL_0
#undef K_QMM1_OPSEL_1
#undef K_QMM1_OPSEL_2
#undef K_QMM1_OPSEL_3
  };

  if constexpr (RINGQ == 3) {
    // ---- 3-deep ring: stages s+1 and s+2 in flight while stage s is consumed (8 DMA instructions per wave and stage)
    static_assert(Q::DMA_PER_STAGE == 8 || RINGQ != 3, "counted wait below");
    issue();
    if (S > 1) issue();
    if (S > 1) WAIT_VMQ(8); else WAIT_VMQ(0);
    BARRIERQ();
    for (int s = 0; s < S; ++s) {
      if (s + 2 < S) issue();
      compute();
      if (s + 2 < S) WAIT_VMQ(8); else WAIT_VMQ(0);   // stage s+1 has landed (this wave's part; the barrier covers the rest)
      BARRIERQ();
    }
  } else {
    // ---- 2-deep ring: both slots are requested up front (the streaming layers have S = 2: nothing else ever is), afterwards
    // stage s+2 goes into slot s % 2 once every wave is done with stage s, i.e. it is in flight under the MFMAs of stage s+1
    static_assert(Q::DMA_PER_STAGE == 10 || RINGQ != 2, "counted wait below");
    issue();
    if (S > 1) issue();
    if (S > 1) WAIT_VMQ(10); else WAIT_VMQ(0);
    BARRIERQ();
    for (int s = 0; s < S; ++s) {
      compute();
      WAIT_VMQ(0);                                   // stage s+1 (the only one in flight) has landed
      BARRIERQ();                                    // ... for every wave, and every wave is done reading stage s
      if (s + 2 < S) issue();
    }
  }
  // The MFMAs are inline asm: the compiler inserts none of the wait states a VALU read of a matrix-pipe result needs (<= 18 for
  // a 16-pass MFMA); the accumulators are threaded through this statement, so every epilogue read comes >= 32 states later.
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
  asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));

  conv1x1_epilogue<EPI, BM>(p, smem, tid, nt, m0, b, p0, c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23, c30, c31, c32, c33);
}

}  // namespace

// Which pointwise layers the MX kernel takes (fp8 mode, inputs available as MX-fp8 twins).
bool conv1x1_mxfp8_eligible(const ConvArgs& a) {
  if (a.pad != 0 || a.stride < 1 || a.KH < 1 || a.KW < 1) return false;
  if ((a.Hout - 1) * a.stride + a.KH > a.Hin || (a.Wout - 1) * a.stride + a.KW > a.Win) return false;   // gather stays inside
  if (a.ps0 != a.C0 || (a.C1 && a.ps1 != a.C1)) return false;
  if (a.C1 && (a.KH != 1 || a.KW != 1)) return false;
  if (a.C0 % KQ || a.C1 % KQ || a.Cout % BNQ || a.Cout != a.CoutPad) return false;
  if (((long)a.Hout * a.Wout) % 128) return false;              // tiles of 128 (or, when it divides, 256) pixels of ONE image
  if (a.gn_partial) return false;
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU && ((a.Cout / 4) % BNQ || a.residual || a.gn_res_src)) return false;
  if (a.mode != CONV_PLAIN && a.mode != CONV_PIXEL_SHUFFLE_SILU) return false;
  if (a.residual && a.gn_res_src) return false;
  if (a.eps4 && (!a.gn_res_src || a.Cout != BNQ || !a.fin_w || !a.fin_b || a.out_q)) return false;
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.C0, a.C1) >= (1ull << 31)) return false;
  if ((size_t)a.KH * a.KW * ((a.C0 + a.C1) / KQ) * (a.Cout / BNQ) * BQ_BYTES >= (1ull << 31)) return false;
  return true;
}

// fp32 [tap][Cout][Cin] (k contiguous: the generic path's order incl. its pixel-shuffle column permutation and the
// space-to-depth tap split) -> [tap][cc][ntile][16.5 KiB]: 128 rows x 128 B of e4m3 (swizzled LDS image) + 512 E8M0 bytes
// [wn][r16][blk][J]; one scale per (output channel, tap, 32 input channels): w = q * 2^(byte - 127).
void pack_conv1x1_mxfp8(const float* src_tap_o_i, int taps, int Cin, int Cout, std::vector<unsigned char>& out) {
  const int CC = Cin / KQ, NTL = Cout / BNQ;
  out.assign((size_t)taps * CC * NTL * BQ_BYTES, 0);
  for (int tap = 0; tap < taps; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned char* unit = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * BQ_BYTES;
        for (int n = 0; n < BNQ; ++n) {
          const float* row = src_tap_o_i + ((size_t)tap * Cout + nt * BNQ + n) * Cin + cc * KQ;
          for (int blk = 0; blk < 4; ++blk) {
            float amax = 0.f;
            for (int e = 0; e < 32; ++e) amax = std::max(amax, std::fabs(row[blk * 32 + e]));
            const int ex = mx_block_exponent(amax);
            unit[BQ_TILE + (((n >> 6) * 16 + (n & 15)) * 4 + blk) * 4 + ((n >> 4) & 3)] = (unsigned char)(ex + 127);
            const float inv = std::ldexp(1.0f, -ex);
            for (int e = 0; e < 32; ++e) {
              const int k = blk * 32 + e;
              const int chunk = (k >> 4) ^ (n & 6);
              unit[n * 128 + chunk * 16 + (k & 15)] = e4m3_encode(row[k] * inv);
            }
          }
        }
      }
}

int conv1x1_mxfp8(const ConvArgs& a, const void* q0, const void* s0, const void* q1, const void* s1, const void* packed_w,
                  hipStream_t st) {
  if (!conv1x1_mxfp8_eligible(a)) SRGD_FAIL("conv1x1_mxfp8: shape not eligible");
  if (!q0 || !s0 || (a.C1 && (!q1 || !s1))) SRGD_FAIL("conv1x1_mxfp8: missing quantised operand");
  Conv1QArgs p;
  p.q0 = (const unsigned char*)q0; p.s0 = (const unsigned char*)s0; p.C0 = a.C0;
  p.q1 = a.C1 ? (const unsigned char*)q1 : nullptr; p.s1 = a.C1 ? (const unsigned char*)s1 : nullptr; p.C1 = a.C1;
  p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Hout = a.Hout; p.Wout = a.Wout;
  p.KH = a.KH; p.KW = a.KW; p.stride = a.stride;
  p.w = (const unsigned char*)packed_w; p.bias = a.bias; p.Cout = a.Cout; p.out = (bf16*)a.out;
  p.aux = a.gn_res_src ? (const bf16*)a.gn_res_src : (const bf16*)a.residual;
  p.gn_a = a.gn_res_a; p.gn_b = a.gn_res_b;
  p.oq = (unsigned char*)a.out_q; p.os = (unsigned char*)a.out_s;
  p.eps4 = a.eps4; p.fin_w = a.fin_w; p.fin_b = a.fin_b;
  if ((p.oq != nullptr) != (p.os != nullptr)) SRGD_FAIL("conv1x1_mxfp8: MX-fp8 twin needs both the element and the scale buffer");
  // tile shape: 128 pixels, two workgroups per CU (a 256-pixel one-workgroup-per-CU shape measured slower and was removed)
  constexpr int BM = 128;
  const long m_tiles = (long)a.B * a.Hout * a.Wout / BM;
  const long grid = m_tiles * (a.Cout / BNQ);
  if (grid <= 0 || grid > 0x7fffffffL) SRGD_FAIL("conv1x1_mxfp8: bad grid");
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define K_SETQ1(E_)                                                                                       \
  SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_mxfp8_kernel<E_, 128>),                \
                               hipFuncAttributeMaxDynamicSharedMemorySize, QShape<128>::LDS));
    K_SETQ1(EPI_PLAIN) K_SETQ1(EPI_RESIDUAL) K_SETQ1(EPI_GNTAIL) K_SETQ1(EPI_PS_SILU) K_SETQ1(EPI_GNTAIL_FINAL)
#undef K_SETQ1
    once.done();
  }
#define K_GOQ1(E_)                                                                                                             \
  hipLaunchKernelGGL((conv1x1_mxfp8_kernel<E_, 128>), dim3((unsigned)grid), dim3(256), QShape<128>::LDS, st, p)
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU) K_GOQ1(EPI_PS_SILU);
  else if (a.gn_res_src && a.eps4) K_GOQ1(EPI_GNTAIL_FINAL);
  else if (a.gn_res_src) K_GOQ1(EPI_GNTAIL);
  else if (a.residual) K_GOQ1(EPI_RESIDUAL);
  else K_GOQ1(EPI_PLAIN);
#undef K_GOQ1
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
