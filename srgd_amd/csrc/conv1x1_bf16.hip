// Pointwise (1x1) convolutions in bf16 for gfx950 (MI355X): the ResnetBlock residual projection
// (reference model.py:271, with the GroupNorm2+SiLU tail of Block.forward :250-259 and the residual add :285
// folded into its epilogue), LinearAttention/Attention to_qkv / to_out (:300-303, :338-340), the
// PixelShuffleUpsample 1x1 conv + SiLU + PixelShuffle (:70-98) and the Downsample space-to-depth + 1x1 conv
// (:106-110, as a 2x2 / stride-2 gather).  These layers are HBM-bound (<= 256 FLOP per byte moved), so the
// kernel is a streaming GEMM built for bytes in flight rather than for MFMA occupancy:
//   * workgroup = 512 threads = 8 waves (4 along M x 2 along N); output tile = 256 consecutive pixels x 128
//     channels; K walked in 32-channel steps (64-byte LDS rows, the XOR swizzle and the 16x16x32 MFMA operand
//     pattern of conv3x3_bf16.hip);
//   * every K-step is ONE 24 KB stage (A 256 x 64 B + one pre-swizzled 8 KB weight tile) brought in by LDS-DMA
//     (3 buffer_load...lds per wave) into a 3-deep ring: two stages (48 KB per workgroup, 96 KB per CU at 2
//     workgroups/CU) are always in flight, counted s_waitcnt + one raw barrier per step;
//   * the accumulators are transposed through LDS and leave as 16-byte channel-contiguous stores; the epilogue
//     variants (residual add, GroupNorm tail, pixel-shuffle scatter + SiLU) read/write 16 B per lane as well;
//   * workgroup ids are renumbered so that the n-tiles of one m-tile run back to back on one XCD (A re-read from L2).
#include "conv1x1_epilogue.hpp"

namespace srgd {
namespace {

constexpr int BM1 = 256, BN1 = 128, KC1 = 32, NT1 = 512;
constexpr int A1_BYTES = BM1 * KC1 * 2;            // 16 KiB
constexpr int B1_BYTES = BN1 * KC1 * 2;            // 8 KiB
constexpr int STAGE1 = A1_BYTES + B1_BYTES;        // 24 KiB
constexpr int RING1 = 3;
constexpr int LDS1_BYTES = RING1 * STAGE1;         // 73,728 >= EPI_LDS_BYTES = 69,632 (epilogue staging)
static_assert(LDS1_BYTES >= EPI_LDS_BYTES && BM1 == EPI_BM && BN1 == EPI_BN && NT1 == EPI_NT, "conv1x1_epilogue.hpp tile shape");

typedef __attribute__((address_space(3))) void* lds_ptr1;
// voffset: per-lane byte offset (VGPR); soffset: wave-uniform byte offset (SGPR)
__device__ __forceinline__ void dma16_1(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr1)lds_wave_base, 16, voffset, soffset, 0, 0);
}
#define WAIT_VM1(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BARRIER1()                       \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

struct Conv1Args {
  const bf16* in0; const bf16* in1; int C0, C1;
  int B, Hin, Win;        // input image grid
  int Hout, Wout;         // conv output grid (== input grid for taps 1, half of it for taps 4)
  int KH, KW, stride;     // gather taps: 1x1; 2x2 / stride 2 (space-to-depth folded into the K walk); 7x1 (input conv)
  int ps0, ps1;           // pixel stride of each source in elements (== C0 / C1 except for the input conv's
                          // overlapping 64-element rows over an 8-channel image, see kernels.hpp)
  const bf16* w;          // packed [tap][cc][ntile][128 rows][64 B swizzled]
  const float* bias;
  int Cout;
  bf16* out;
  const bf16* aux;        // EPI_RESIDUAL: tensor added to the output; EPI_GNTAIL: tensor the GroupNorm tail is applied to
  const float* gn_a; const float* gn_b;   // EPI_GNTAIL: [B][Cout] scale / shift
  unsigned char* oq; unsigned char* os;   // optional MX-fp8 twin of the output (ConvArgs::out_q / out_s)
  float* eps4; const float* fin_w; const float* fin_b;   // EPI_GNTAIL_FINAL (ConvArgs::eps4)
};

__device__ __forceinline__ int row_swz1(int row) { return (row >> 1) & 3; }

template <int EPI>
__global__ __launch_bounds__(NT1, 4) void conv1x1_bf16_kernel(Conv1Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, q16 = lane >> 4;

  const int n_tiles = p.Cout / BN1;
  const int HWo = p.Hout * p.Wout;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, x = wg & 7, k = wg >> 3;
    wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
  }
  const int nt = wg % n_tiles;
  const int mt = wg / n_tiles;
  const long m0 = (long)mt * BM1;                  // first output pixel of the tile (HWo % 256 == 0: one image)
  const int b = (int)(m0 / HWo);
  const int p0 = (int)(m0 - (long)b * HWo);        // pixel offset inside the image
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / KC1;
  const int S = p.KH * p.KW * CC;

  // ---- A staging: 16 wave-instructions per stage, wave w issues pieces w and w+8.  Per-lane BYTE offset of its two pieces
  // inside each source at tap (0,0), channel chunk 0: input pixel * pixel stride + swizzled 16-byte chunk.  Everything that
  // changes from K-step to K-step (tap offset, channel chunk, ring slot, weight tile) is wave-uniform and rides in SGPRs - the
  // buffer instruction's scalar offset and M0.  Round 2 recomputed the per-lane offsets every step from a runtime s / CC and
  // tap / KW (66 SALU + 17 VALU instructions, two of them v_mul_lo, per 16 MFMAs).
#define K_A1_DECL(J)                                                      \
  int a_b0##J, a_b1##J;                                                      \
  {                                                                          \
    const int g = (wave + 8 * J) * 64 + lane;                                \
    const int P = g >> 2;                                                    \
    const int op = p0 + P;                                                   \
    const int oy = op / p.Wout, ox = op - oy * p.Wout;                       \
    const int pix = oy * p.stride * p.Win + ox * p.stride;                   \
    const int sub = (g & 3) ^ row_swz1(P);                                   \
    a_b0##J = (pix * p.ps0 + sub * 8) * 2;                                   \
    a_b1##J = (pix * p.ps1 + sub * 8) * 2;                                   \
  }
  K_A1_DECL(0) K_A1_DECL(1)
#undef K_A1_DECL
  const size_t img0 = (size_t)p.Hin * p.Win * p.ps0, img1 = (size_t)p.Hin * p.Win * p.ps1;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.in0 + (size_t)b * img0), 0, (int)(img0 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in1 ? p.in1 + (size_t)b * img1 : p.in0), 0, p.in1 ? (int)(img1 * 2) : 0, 0x00020000);
  const size_t w_tile_stride = (size_t)n_tiles * B1_BYTES;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.w + (size_t)nt * B1_BYTES), 0, (int)((size_t)(S - 1) * w_tile_stride + B1_BYTES), 0x00020000);

  // issue stream: the next K-step to request - tap (ty, tx), channel chunk, ring slot, weight offset - advanced incrementally
  int i_ty = 0, i_tx = 0, i_cc = 0, i_slot = 0, i_w = 0;
  const int tid16 = tid * 16;
  auto issue = [&]() __attribute__((always_inline)) {
    const int c = i_cc * KC1;
    const bool first = c < p.C0;
    const int soff = first ? ((i_ty * p.Win + i_tx) * p.ps0 + c) * 2 : ((i_ty * p.Win + i_tx) * p.ps1 + c - p.C0) * 2;
    char* st = smem + i_slot * STAGE1;
    // (the per-lane offsets are selected with v_cndmask on purpose: written as two branches hipcc merges them into a select of
    // ADDRESSES of the two candidates and parks those in scratch - a flat load behind s_waitcnt vmcnt(0) in every K-step)
    const int vo0 = first ? a_b00 : a_b10, vo1 = first ? a_b01 : a_b11;
    if (first) {
      dma16_1(rs0, st + wave * 1024, vo0, soff);
      dma16_1(rs0, st + (wave + 8) * 1024, vo1, soff);
    } else {
      dma16_1(rs1, st + wave * 1024, vo0, soff);
      dma16_1(rs1, st + (wave + 8) * 1024, vo1, soff);
    }
    dma16_1(rsw, st + A1_BYTES + wave * 1024, tid16, i_w);
    i_w += (int)w_tile_stride;
    i_slot = i_slot == RING1 - 1 ? 0 : i_slot + 1;
    if (++i_cc == CC) {
      i_cc = 0;
      if (++i_tx == p.KW) { i_tx = 0; ++i_ty; }
    }
  };

  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0,
        c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;
  auto a_addr = [&](int i) {
    const int P = wm * 64 + i * 16 + r16;
    return P * 64 + ((q16 ^ row_swz1(P)) << 4);
  };
  auto b_addr = [&](int j) {
    const int n = wn * 64 + j * 16 + r16;
    return A1_BYTES + n * 64 + ((q16 ^ row_swz1(n)) << 4);
  };
  const int aa0 = a_addr(0), aa1 = a_addr(1), aa2 = a_addr(2), aa3 = a_addr(3);
  const int ba0 = b_addr(0), ba1 = b_addr(1), ba2 = b_addr(2), ba3 = b_addr(3);
  int c_slot = 0;                                  // ring slot of the K-step being consumed
  auto compute = [&]() __attribute__((always_inline)) {
    const char* st = smem + c_slot * STAGE1;
    c_slot = c_slot == RING1 - 1 ? 0 : c_slot + 1;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(st + aa0);
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(st + aa1);
    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(st + aa2);
    const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(st + aa3);
    const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(st + ba0);
    const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(st + ba1);
    const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(st + ba2);
    const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(st + ba3);
#define MM(C_, A_, B_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, C_, 0, 0, 0)
    MM(c00, a0, b0); MM(c01, a0, b1); MM(c02, a0, b2); MM(c03, a0, b3);
    MM(c10, a1, b0); MM(c11, a1, b1); MM(c12, a1, b2); MM(c13, a1, b3);
    MM(c20, a2, b0); MM(c21, a2, b1); MM(c22, a2, b2); MM(c23, a2, b3);
    MM(c30, a3, b0); MM(c31, a3, b1); MM(c32, a3, b2); MM(c33, a3, b3);
#undef MM
  };

  // ---- pipeline: stages s+1 and s+2 in flight while stage s is consumed (3 DMA instructions per wave and stage)
  issue();
  if (S > 1) issue();
  if (S > 1) WAIT_VM1(3); else WAIT_VM1(0);
  BARRIER1();
  for (int s = 0; s < S; ++s) {
    if (s + 2 < S) issue();
    compute();
    if (s + 2 < S) WAIT_VM1(3); else WAIT_VM1(0);   // stage s+1 has landed (this wave's part; the barrier covers the rest)
    BARRIER1();
  }

  // ---- epilogue (conv1x1_epilogue.hpp): transpose through LDS, then 16-byte channel-contiguous traffic only
  conv1x1_epilogue<EPI, BM1>(p, smem, tid, nt, m0, b, p0, c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23, c30, c31, c32, c33);
}

}  // namespace

// Which generic-conv calls this kernel takes over (bf16 activations only).
bool conv1x1_bf16_eligible(const ConvArgs& a) {
  if (a.pad != 0 || a.stride < 1 || a.KH < 1 || a.KW < 1) return false;
  if ((a.Hout - 1) * a.stride + a.KH > a.Hin || (a.Wout - 1) * a.stride + a.KW > a.Win) return false;   // gather stays inside
  if (a.ps0 % 8 || (a.C1 && (a.ps1 % 8 || a.ps1 < a.C1))) return false;                              // 16-byte aligned rows
  if (a.ps0 != a.C0 && (a.C1 || (long)(a.Win - (a.Wout - 1) * a.stride - a.KW) * a.ps0 + a.ps0 < a.C0)) return false;
  if (a.C1 && (a.KH != 1 || a.KW != 1)) return false;
  if (a.C0 % KC1 || a.C1 % KC1 || a.Cout % BN1 || a.Cout != a.CoutPad) return false;
  if (((long)a.Hout * a.Wout) % BM1) return false;
  if (a.gn_partial) return false;
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU && ((a.Cout / 4) % BN1 || a.residual || a.gn_res_src)) return false;
  if (a.mode != CONV_PLAIN && a.mode != CONV_PIXEL_SHUFFLE_SILU) return false;
  if (a.residual && a.gn_res_src) return false;
  if (a.eps4 && (!a.gn_res_src || a.Cout != BN1 || !a.fin_w || !a.fin_b || a.out_q)) return false;
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.ps0, a.ps1) * 2 >= (1ull << 31)) return false;
  if ((size_t)a.KH * a.KW * ((a.C0 + a.C1) / KC1) * (a.Cout / BN1) * B1_BYTES >= (1ull << 31)) return false;
  return true;
}

// Host-side packing: fp32 [tap][Cout][Cin] (k contiguous, the generic path's order incl. its pixel-shuffle column
// permutation) -> [tap][cc][ntile][128 rows][64 B swizzled] bf16, the LDS image of each K-step's weight tile.
void pack_conv1x1_bf16(const float* src_tap_o_i, int taps, int Cin, int Cout, std::vector<unsigned short>& out,
                       unsigned short (*to_bf16)(float)) {
  const int CC = Cin / KC1, NTL = Cout / BN1;
  out.assign((size_t)taps * CC * NTL * BN1 * KC1, 0);
  for (int tap = 0; tap < taps; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned short* tile = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * BN1 * KC1;
        for (int n = 0; n < BN1; ++n)
          for (int c = 0; c < 4; ++c) {
            const int cs = c ^ ((n >> 1) & 3);
            for (int e = 0; e < 8; ++e) {
              const int ci = cc * KC1 + c * 8 + e, o = nt * BN1 + n;
              tile[n * KC1 + cs * 8 + e] = to_bf16(src_tap_o_i[((size_t)tap * Cout + o) * Cin + ci]);
            }
          }
      }
}

int conv1x1_bf16(const ConvArgs& a, const void* packed_w, hipStream_t st) {
  if (!conv1x1_bf16_eligible(a)) SRGD_FAIL("conv1x1_bf16: shape not eligible");
  Conv1Args p;
  p.in0 = (const bf16*)a.in0; p.in1 = (const bf16*)a.in1; p.C0 = a.C0; p.C1 = a.C1;
  p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Hout = a.Hout; p.Wout = a.Wout;
  p.KH = a.KH; p.KW = a.KW; p.stride = a.stride; p.ps0 = a.ps0; p.ps1 = a.C1 ? a.ps1 : 0;
  p.w = (const bf16*)packed_w; p.bias = a.bias; p.Cout = a.Cout; p.out = (bf16*)a.out;
  p.aux = a.gn_res_src ? (const bf16*)a.gn_res_src : (const bf16*)a.residual;
  p.gn_a = a.gn_res_a; p.gn_b = a.gn_res_b;
  p.oq = (unsigned char*)a.out_q; p.os = (unsigned char*)a.out_s;
  p.eps4 = a.eps4; p.fin_w = a.fin_w; p.fin_b = a.fin_b;
  if ((p.oq != nullptr) != (p.os != nullptr)) SRGD_FAIL("conv1x1_bf16: MX-fp8 twin needs both the element and the scale buffer");
  const long m_tiles = (long)a.B * a.Hout * a.Wout / BM1;
  const long grid = m_tiles * (a.Cout / BN1);
  if (grid <= 0 || grid > 0x7fffffffL) SRGD_FAIL("conv1x1_bf16: bad grid");
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define K_SET1(E_)                                                                                   \
  SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_bf16_kernel<E_>),                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS1_BYTES));
    K_SET1(EPI_PLAIN) K_SET1(EPI_RESIDUAL) K_SET1(EPI_GNTAIL) K_SET1(EPI_PS_SILU) K_SET1(EPI_GNTAIL_FINAL)
#undef K_SET1
    once.done();
  }
#define K_GO1(E_) hipLaunchKernelGGL((conv1x1_bf16_kernel<E_>), dim3((unsigned)grid), dim3(NT1), LDS1_BYTES, st, p)
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU) K_GO1(EPI_PS_SILU);
  else if (a.gn_res_src && a.eps4) K_GO1(EPI_GNTAIL_FINAL);
  else if (a.gn_res_src) K_GO1(EPI_GNTAIL);
  else if (a.residual) K_GO1(EPI_RESIDUAL);
  else K_GO1(EPI_PLAIN);
#undef K_GO1
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
