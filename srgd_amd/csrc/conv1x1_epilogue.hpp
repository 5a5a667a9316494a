// Epilogue shared by the pointwise streaming GEMMs (conv1x1_bf16.hip and conv1x1_mxfp8.hip): both end with the same 16 x 16
// accumulator blocks per wave (8 waves = 4 along M x 2 along N, wave tile 64 x 64, D layout row = 4 * (lane >> 4) + reg,
// column = lane & 15) of a 256-pixel x 128-channel tile, so the output side - + bias, LDS transpose, 16-byte channel-contiguous
// stores, residual add, the ResnetBlock's GroupNorm2 + SiLU tail (reference model.py:250-259, :285), PixelShuffle scatter + SiLU
// (:70-98), the fused 1x1 output convolution (:776-777), the optional MX-fp8 twin - is written once.
#pragma once
#include "kernels.hpp"

namespace srgd {
namespace {

// Output stores and tail-operand loads are non-temporal (both tensors are streamed once): pointwise share 13.86 -> 13.74 % of a step,
// same box (profiles/r5/nt_policy/r5_nt_c1.json).
enum { EPI_PLAIN = 0, EPI_RESIDUAL = 1, EPI_GNTAIL = 2, EPI_PS_SILU = 3, EPI_GNTAIL_FINAL = 4 };
constexpr int EPI_BM = 256, EPI_BN = 128, EPI_NT = 512;
constexpr int EPI_ROW = EPI_BN * 2 + 16;           // transposed output row (272 B: conflict-free 2-byte column writes)
constexpr int EPI_LDS_BYTES = EPI_BM * EPI_ROW;    // 69,632 B of staging (every caller's operand ring is larger)

// `Args` supplies: Hout, Wout, Cout, bias, out, aux, gn_a, gn_b, oq, os, eps4, fin_w, fin_b (Conv1Args / Conv1QArgs).
// m0: first output pixel of the tile (global), b: its image, p0: pixel offset inside the image, nt: 128-channel tile index.
// Must be entered by all 2 * BM threads AFTER a barrier that retired every read of the operand buffers in `smem`.
// BM: pixels per tile - 256 (8 waves, 512 threads) or 128 (4 waves, 256 threads; conv1x1_mxfp8's two-workgroups-per-CU shape);
// two threads per pixel either way, so the store loop has the same eight iterations.
template <int EPI, int BM, class Args>
__device__ __forceinline__ void conv1x1_epilogue(const Args& p, char* smem, const int tid, const int nt, const long m0, const int b,
                                                 const int p0, const f32x4& c00, const f32x4& c01, const f32x4& c02, const f32x4& c03,
                                                 const f32x4& c10, const f32x4& c11, const f32x4& c12, const f32x4& c13,
                                                 const f32x4& c20, const f32x4& c21, const f32x4& c22, const f32x4& c23,
                                                 const f32x4& c30, const f32x4& c31, const f32x4& c32, const f32x4& c33) {
  constexpr int BM1 = BM, BN1 = EPI_BN, NT1 = BM * 2, EROW1 = EPI_ROW;
  static_assert(BM == 256 || BM == 128, "conv1x1_epilogue: tile of 256 or 128 pixels");
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, q16 = lane >> 4;
  // ---- epilogue: transpose through LDS, then 16-byte channel-contiguous traffic only
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int cl = wn * 64 + ni * 16 + r16;
    const float bias = p.bias ? p.bias[nt * BN1 + cl] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const f32x4 av = mi == 0 ? (ni == 0 ? c00 : ni == 1 ? c01 : ni == 2 ? c02 : c03)
                     : mi == 1 ? (ni == 0 ? c10 : ni == 1 ? c11 : ni == 2 ? c12 : c13)
                     : mi == 2 ? (ni == 0 ? c20 : ni == 1 ? c21 : ni == 2 ? c22 : c23)
                               : (ni == 0 ? c30 : ni == 1 ? c31 : ni == 2 ? c32 : c33);
      char* trow = smem + (wm * 64 + mi * 16 + q16 * 4) * EROW1 + cl * 2;      // C layout: row = (lane >> 4) * 4 + reg
      // staged in bf16 (LDS budget): with a residual / GroupNorm-tail add the conv term is rounded once here and the
      // sum once more at the store
      float v0 = av[0] + bias, v1 = av[1] + bias, v2 = av[2] + bias, v3 = av[3] + bias;
      if (EPI == EPI_PS_SILU) { v0 = silu<false>(v0); v1 = silu<false>(v1); v2 = silu<false>(v2); v3 = silu<false>(v3); }
      // Round 4: the short-K pointwise layers are bound by this phase (a 256 -> 512 @128^2 tile spends 8 K-steps in its loop and
      // emits output at the same ~2 TB/s as every other shape: 64 two-byte LDS writes per lane).  As in conv3x3_bf16.hip: two
      // adjacent lanes hold two adjacent channels of the same four rows; they swap halves (one DPP quad_perm move) so that the
      // even lane owns rows 0-1 and the odd lane rows 2-3 of BOTH channels - two conflict-free ds_write_b32 per block instead of
      // four ds_write_b16 whose lane pairs share a dword.  Same values, same rounding: bit-identical.
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const unsigned own01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2_t));
      const unsigned own23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v2, v3}, bf16x2_t));
      const bool odd_lane = (r16 & 1) != 0;
      const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_lane ? own01 : own23), 0xB1, 0xf, 0xf, true);
      const unsigned lo_ch = odd_lane ? recv : own01, hi_ch = odd_lane ? own23 : recv;     // channel c (even) | c + 1
      char* prow = trow + (odd_lane ? 2 * EROW1 - 2 : 0);   // even lane: rows 0, 1 at its own column; odd lane: rows 2, 3, one column left
      *reinterpret_cast<unsigned*>(prow) = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x05040100u);
      *reinterpret_cast<unsigned*>(prow + EROW1) = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x07060302u);
    }
  }
  __syncthreads();
  const int col0 = nt * BN1;
  size_t obase;                                      // element offset of (tile pixel 0, channel col0) for plain layouts
  int CoutPS = 0, ps_ij = 0, ps_c0 = 0;
  if (EPI == EPI_PS_SILU) {
    CoutPS = p.Cout >> 2;
    ps_ij = col0 / CoutPS;
    ps_c0 = col0 - ps_ij * CoutPS;
  }
  obase = (size_t)m0 * p.Cout + col0;
  // EPI_GNTAIL_FINAL: this lane's 8 channels (c16 = tid & 15 in every iteration) of the three output-convolution rows (the
  // 16 lanes of a DPP row share a pixel; fp32 sums in a different order than out_conv3_coop's: equal to rounding)
  float fw0[8], fw1[8], fw2[8];
  if (EPI == EPI_GNTAIL_FINAL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fw0[e] = p.fin_w[(tid & 15) * 8 + e];
      fw1[e] = p.fin_w[BN1 + (tid & 15) * 8 + e];
      fw2[e] = p.fin_w[2 * BN1 + (tid & 15) * 8 + e];
    }
  }
  // Round 3: the tail operand (`aux`: the tensor the GroupNorm tail is applied to / the residual) of ALL of the thread's chunks is
  // loaded up front.  `aux` and `out` are both bf16 and may alias (the engine runs the tail in place), so written inside the
  // loop the compiler must keep load i+1 behind store i: eight exposed HBM round trips per thread.  Each thread reads exactly the
  // elements it later overwrites, so the hoist is safe in the in-place case; the 64 accumulator registers are dead by now.
  // The GroupNorm coefficients depend on (image, channel chunk) only: once per thread.
  constexpr int NCH = (BM1 * 16) / NT1;               // 16-byte chunks per thread: 8
  bf16x8 auxv[NCH];
  f32x4 a_lo = {0.f, 0.f, 0.f, 0.f}, a_hi = a_lo, b_lo = a_lo, b_hi = a_lo;
  if (EPI == EPI_RESIDUAL || EPI == EPI_GNTAIL || EPI == EPI_GNTAIL_FINAL) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = tid + NT1 * i;
      const bf16x8* ap = reinterpret_cast<const bf16x8*>(p.aux + obase + (size_t)(q >> 4) * p.Cout + (q & 15) * 8);
      auxv[i] = __builtin_nontemporal_load(ap);
    }
    if (EPI != EPI_RESIDUAL) {
      const float* ga = p.gn_a + (size_t)b * p.Cout + col0 + (tid & 15) * 8;
      const float* gb = p.gn_b + (size_t)b * p.Cout + col0 + (tid & 15) * 8;
      a_lo = *reinterpret_cast<const f32x4*>(ga); a_hi = *reinterpret_cast<const f32x4*>(ga + 4);
      b_lo = *reinterpret_cast<const f32x4*>(gb); b_hi = *reinterpret_cast<const f32x4*>(gb + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int q = tid + NT1 * i;
    const int pix = q >> 4, c16 = q & 15;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + pix * EROW1 + c16 * 16);
    if (EPI == EPI_PS_SILU) {
      const int op = p0 + pix;
      const int oy = op / p.Wout, ox = op - oy * p.Wout;
      const size_t o = ((size_t)(b * 2 * p.Hout + 2 * oy + (ps_ij >> 1)) * (2 * p.Wout) + 2 * ox + (ps_ij & 1)) * CoutPS +
                       ps_c0 + c16 * 8;
      __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p.out + o));
      if (p.oq) mx_store_twin(v, p.oq, p.os, o, tid & 3);
    } else {
      const size_t o = obase + (size_t)pix * p.Cout + c16 * 8;
      if (EPI == EPI_RESIDUAL) {
        const bf16x8 rr = auxv[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rr[e]);
      } else if (EPI == EPI_GNTAIL || EPI == EPI_GNTAIL_FINAL) {
        const bf16x8 hh = auxv[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ca = e < 4 ? a_lo[e & 3] : a_hi[e & 3], cb = e < 4 ? b_lo[e & 3] : b_hi[e & 3];
          v[e] = (bf16)(silu<false>(ca * (float)hh[e] + cb) + (float)v[e]);
        }
      }
      if (EPI == EPI_GNTAIL_FINAL) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = (float)v[e];
          s0 += xv * fw0[e];
          s1 += xv * fw1[e];
          s2 += xv * fw2[e];
        }
        s0 = row16_sum(s0);
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if (c16 == 0)
          *reinterpret_cast<f32x4*>(p.eps4 + ((size_t)m0 + pix) * 4) = f32x4{s0 + p.fin_b[0], s1 + p.fin_b[1], s2 + p.fin_b[2], 0.f};
        continue;
      }
      __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p.out + o));
      if (p.oq) mx_store_twin(v, p.oq, p.os, o, tid & 3);
    }
  }
}

}  // namespace
}  // namespace srgd
