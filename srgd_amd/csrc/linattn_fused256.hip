// Fused LinearAttention block at C = 256 (bf16 mode, 4 heads x 32) for gfx950: the two LinearAttention sites of the dim-128
// U-Net that sit on 256-channel tensors (third down stage @64^2, third up stage @128^2; reference model.py:306-324, :703).
// Same two-kernel scheme as linattn_fused.hip (read x, read x, write y instead of the eleven tensor passes of the unfused
// chain RMSNorm -> to_qkv -> k softmax / context -> q softmax / out -> to_out -> RMSNorm -> + x), re-tiled for rows of 512 B:
//   * 32-pixel tiles (16 KiB) in a 3-deep LDS-DMA ring, so a workgroup needs 48-62 KiB and two share a CU;
//   * la1: a wave owns one head; its 64 rows of Wkv' (32 k + 32 v) x 256 channels stay in 128 VGPRs as MFMA B fragments,
//     [k|v] of the tile = 2 x 16 MFMAs (32x32x16), exp2(k - m) and v feed ctx += p^T v straight from the accumulators;
//   * la2: a wave owns head hd for q (Wq' rows in 64 VGPRs) and output channels [64 hd, 64 hd + 64) for to_out (Wout rows in
//     64 VGPRs); att goes through an 8 KiB LDS tile, bias / RMSNorm gain come from LDS, and y = RMSNorm(o) g + x is formed
//     IN the staged x tile (each (pixel, channel) element is owned by one lane), which then leaves as whole 512-byte rows.
// Partials / combine / workspace layout are those of linattn_fused.hip (la_combine_kernel merges the strips).
#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int TM2 = 32;                   // pixels per tile
constexpr int NTH2 = 256;                 // 4 waves
constexpr int RING2 = 3;
constexpr int ATT2 = TM2 * 256;           // att tile [32 px][128 k] bf16
// per channel count C (128 or 256): bytes per pixel row, bytes per tile, 16-byte chunks per row, LDS-DMA pieces per wave and
// tile, k16 steps of the q / kv GEMMs, 32-row output blocks per wave of the to_out GEMM
template <int C> struct LaDims {
  static constexpr int ROW = C * 2, TILE = TM2 * C * 2, NCH = C / 8, PPW = (TM2 * (C / 8)) / NTH2, KS = C / 16, NB = C / 128;
};
constexpr float LOG2E_ = 1.4426950408889634f, LN2_ = 0.6931471805599453f;

typedef __attribute__((address_space(3))) void* lds_ptr2;

template <int C> __device__ __forceinline__ int swz2(int row, int chunk) { return row * (C * 2) + ((chunk ^ (row & 15)) << 4); }    // x tile
__device__ __forceinline__ int swza(int row, int chunk16) { return row * 256 + ((chunk16 ^ (row & 15)) << 4); }  // att tile

#define LB_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define LB_BARRIER()                     \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define LB_SYNC()                                         \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    LB_BARRIER();                                         \
  } while (0)

// one 32-pixel x tile (rows px0..px0+31) -> `buf`, XOR-swizzled: PPW LDS-DMA pieces of 1 KiB per wave
template <int C>
__device__ __forceinline__ void stage_tile2(__amdgpu_buffer_rsrc_t rsrc, char* buf, int wave, int lane, int px0) {
  constexpr int NCH = LaDims<C>::NCH, PPW = LaDims<C>::PPW;
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j;
    const int g = q * 64 + lane;
    const int row = g / NCH, cs = g % NCH;
    const int c = cs ^ (row & 15);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr2)(buf + q * 1024), 16, ((px0 + row) * C + c * 8) * 2, 0, 0, 0);
  }
}

// 1 / max(||x_row||, 1e-12) of the 32 rows of a staged tile (8 threads per row; the swizzle permutes chunks inside a row only)
template <int C>
__device__ __forceinline__ void row_rinv2(const char* tile, float* rinv, int tid) {
  constexpr int CPT = LaDims<C>::NCH / 8;                 // chunks per thread
  const int row = tid >> 3, part = tid & 7;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * (C * 2) + (part * CPT + j) * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
  }
  ss += __shfl_xor(ss, 1, 64);
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 4, 64);
  if (part == 0) rinv[row] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
}

// the DMA pieces of the youngest staged tile may stay in flight (PIECES is a template constant: s_waitcnt takes an immediate)
template <int PIECES> __device__ __forceinline__ void wait_all_but() {
  if constexpr (PIECES == 2) LB_WAIT_VM(2);
  else if constexpr (PIECES == 3) LB_WAIT_VM(3);
  else if constexpr (PIECES == 4) LB_WAIT_VM(4);
  else if constexpr (PIECES == 5) LB_WAIT_VM(5);
  else if constexpr (PIECES == 9) LB_WAIT_VM(9);
  else static_assert(PIECES == 2, "unsupported wait count");
}

__device__ __forceinline__ float ex2_(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ bf16x8 pack8_(const f32x16& a, int s) {
  // four v_cvt_pk_bf16_f32 whose results are the operand tuple's dwords (linattn_fused.hip: element-wise assembly cost ~40
  // register copies per tile and 7 % of la1's time)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w;
  w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 0], a[8 * s + 1]}, bf16x2_t));
  w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 2], a[8 * s + 3]}, bf16x2_t));
  w[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 4], a[8 * s + 5]}, bf16x2_t));
  w[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 6], a[8 * s + 7]}, bf16x2_t));
  return __builtin_bit_cast(bf16x8, w);
}

// ------------------------------------------------------------------------------------------- phase 1
template <int C>
__global__ __launch_bounds__(NTH2, C == 128 ? 3 : 2) void la1_t_kernel(const bf16* __restrict__ x, int N, const bf16* __restrict__ wkv,
                                                           int strip, float* __restrict__ pm, float* __restrict__ pl,
                                                           float* __restrict__ pctx, float* __restrict__ rinv_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  constexpr int ROW2 = LaDims<C>::ROW, TILE2 = LaDims<C>::TILE, KS = LaDims<C>::KS, PPW = LaDims<C>::PPW;
  float* const sR = reinterpret_cast<float*>(smem + RING2 * TILE2);        // [2][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int head = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, sidx = blockIdx.x, nstrips = gridDim.x;
  const int px_begin = sidx * strip;
  const int T = min(strip, N - px_begin) / TM2;

  const __amdgpu_buffer_rsrc_t rsx =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * N * C), 0, N * ROW2, 0x00020000);
  // B fragments of the head's k rows and v rows of Wkv' ([256 rows = k | v][C] bf16 row-major, gains folded)
  bf16x8 fk[KS], fv[KS];
  {
    const bf16* wk = wkv + (size_t)(head * 32 + r) * C + hh * 8;
    const bf16* wv = wkv + (size_t)(128 + head * 32 + r) * C + hh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      fk[s] = *reinterpret_cast<const bf16x8*>(wk + s * 16);
      fv[s] = *reinterpret_cast<const bf16x8*>(wv + s * 16);
    }
  }
  stage_tile2<C>(rsx, sA, head, lane, px_begin);
  if (T > 1) stage_tile2<C>(rsx, sA + TILE2, head, lane, px_begin + TM2);
  float m = -INFINITY, l = 0.f;
  f32x16 ctx = 0;
  if (T > 1) wait_all_but<PPW>(); else LB_WAIT_VM(0);
  LB_BARRIER();

  for (int t = 0; t < T; ++t) {
    const char* A = sA + (t % RING2) * TILE2;
    float* rinv = sR + (t & 1) * TM2;
    if (t + 2 < T) stage_tile2<C>(rsx, sA + ((t + 2) % RING2) * TILE2, head, lane, px_begin + (t + 2) * TM2);
    row_rinv2<C>(A, rinv, tid);
    LB_SYNC();
    if (head == 0 && lane < 8)
      *reinterpret_cast<f32x4*>(rinv_out + (size_t)b * N + px_begin + t * TM2 + lane * 4) = *reinterpret_cast<const f32x4*>(rinv + lane * 4);

    f32x16 k0 = 0, v0 = 0;
    int rr = r;                                      // C = 128 (168-VGPR budget): opaque per iteration - the fragment addresses
    if constexpr (C == 128) asm volatile("" : "+v"(rr));      // are recomputed, not hoisted into 8 long-lived registers
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(A + swz2<C>(rr, 2 * s + hh));
      k0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fk[s], k0, 0, 0, 0);
      v0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fv[s], v0, 0, 0, 0);
    }
    float bm = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 ri = *reinterpret_cast<const f32x4*>(rinv + 8 * g + 4 * hh);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int reg = 4 * g + i;
        k0[reg] *= ri[i] * LOG2E_;
        v0[reg] *= ri[i];
        bm = fmaxf(bm, k0[reg]);
      }
    }
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    // lazy running maximum (linattn_fused.hip, round 4): the reference point moves only when the tile's maximum exceeds it by
    // more than 2^8, so the context rescale almost never runs; pm carries the point actually used
    const float mn = (bm > m + 8.0f) ? bm : m;
    const float f = ex2_(m - mn);
    m = mn;
    l *= f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      k0[i] = ex2_(k0[i] - mn);
      l += k0[i];
    }
    if (!__all(f == 1.0f)) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int d = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        ctx[reg] *= __shfl(f, d, 64);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8_(k0, s), pack8_(v0, s), ctx, 0, 0, 0);
    if (t + 2 < T) wait_all_but<PPW>(); else LB_WAIT_VM(0);
    LB_BARRIER();
  }
  l += __shfl_xor(l, 32, 64);
  const size_t pidx = (size_t)(b * 4 + head) * nstrips + sidx;
  if (hh == 0) {
    pm[pidx * 32 + r] = m * LN2_;
    pl[pidx * 32 + r] = l;
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int d = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
    pctx[(pidx * 32 + d) * 32 + r] = ctx[reg];
  }
}

// ------------------------------------------------------------------------------------------- phase 2
struct La2Args256 {
  const bf16* x; bf16* y; int N;
  const bf16* wq;        // [128 d][C] bf16 row-major, gains folded
  const bf16* wout;      // [C][128 k] bf16 row-major
  const float* bout;     // [C]
  const float* g2;       // [C] = to_out.1.g * sqrt(C)
  const float* ctxn;     // [B*4][32 d][32 e] fp32
  const float* rinv;     // [B][N]
  unsigned char* yq; unsigned char* ys;
  int tiles_per_wg;
};

template <int C>
__global__ __launch_bounds__(NTH2, C == 128 ? 3 : 2) void la2_t_kernel(La2Args256 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROW2 = LaDims<C>::ROW, TILE2 = LaDims<C>::TILE, KS = LaDims<C>::KS, PPW = LaDims<C>::PPW, NB = LaDims<C>::NB;
  constexpr int CW = C / 4;                                           // output channels per wave (32 or 64)
  char* const sA = smem;                                              // RING2 x tiles
  char* const sT = smem + RING2 * TILE2;                              // att tile
  float* const sS = reinterpret_cast<float*>(sT + ATT2);              // [4][32] partial sums of squares of o
  float* const sRv = sS + 4 * TM2;                                    // [RING2][4 waves][64]
  float* const sB = sRv + RING2 * 4 * 64;                             // [C] bias
  float* const sG = sB + C;                                           // [C] gain
  const int tid = threadIdx.x, lane = tid & 63;
  const int hd = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const int tile0 = blockIdx.x * p.tiles_per_wg;
  const int ntiles = p.N / TM2;
  const int T = min(p.tiles_per_wg, ntiles - tile0);
  if (T <= 0) return;

  const __amdgpu_buffer_rsrc_t rsx =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.N * C), 0, p.N * ROW2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.rinv + (size_t)b * p.N), 0, p.N * 4, 0x00020000);
  auto stage = [&](int slot, int px0) {
    stage_tile2<C>(rsx, sA + slot * TILE2, hd, lane, px0);
    // 64 dwords ride along (the tile's 32 + the next 32; out-of-range lanes read 0 through the buffer bounds check)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, (lds_ptr2)(sRv + (slot * 4 + hd) * 64), 4, (px0 + lane) * 4, 0, 0, 0);
  };

  bf16x8 wq[KS], wo0[8], wo1[8];                          // (wo1: the second 32-row block, C = 256 only)
  {
    const bf16* q = p.wq + (size_t)(hd * 32 + r) * C + hh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) wq[s] = *reinterpret_cast<const bf16x8*>(q + s * 16);
    const bf16* o0 = p.wout + (size_t)(hd * CW + r) * 128 + hh * 8;
    const bf16* o1 = p.wout + (size_t)(hd * CW + (NB == 2 ? 32 : 0) + r) * 128 + hh * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      wo0[s] = *reinterpret_cast<const bf16x8*>(o0 + s * 16);
      if (NB == 2) wo1[s] = *reinterpret_cast<const bf16x8*>(o1 + s * 16);
    }
  }
  bf16x8 cx[2];
  {
    const float* c = p.ctxn + (size_t)(b * 4 + hd) * 1024;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) cx[s][j] = (bf16)c[(16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)) * 32 + r];
  }
  if (tid < C) {
    sB[tid] = p.bout[tid];
    sG[tid] = p.g2[tid];
  }
  stage(0, tile0 * TM2);
  if (T > 1) stage(1, (tile0 + 1) * TM2);
  if (T > 1) wait_all_but<PPW + 1>(); else LB_WAIT_VM(0);
  LB_SYNC();

  for (int t = 0; t < T; ++t) {
    char* A = sA + (t % RING2) * TILE2;
    const int px0 = (tile0 + t) * TM2;
    if (t + 2 < T) stage((t + 2) % RING2, px0 + 2 * TM2);
    const float ri = sRv[((t % RING2) * 4 + hd) * 64 + r] * LOG2E_;
    // the LDS operand addresses are recomputed from an opaque copy of the lane's row every iteration: hoisted out of the loop
    // they cost ~40 VGPRs next to the 136 of register-resident weights, and the kernel spills
    int rr = r;
    asm volatile("" : "+v"(rr));

    // q^T: rows d of head hd, columns = the tile's 32 pixels
    f32x16 q0 = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(A + swz2<C>(rr, 2 * s + hh));
      q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s], xv, q0, 0, 0, 0);
    }
    {
      float m0 = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) m0 = fmaxf(m0, q0[i]);
      m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
      const float c0 = -m0 * ri;
      float s0 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        q0[i] = ex2_(fmaf(q0[i], ri, c0));
        s0 += q0[i];
      }
      s0 += __shfl_xor(s0, 32, 64);
      const float i0 = __frcp_rn(s0);
#pragma unroll
      for (int i = 0; i < 16; ++i) q0[i] *= i0;
    }
    f32x16 a0 = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cx[s], pack8_(q0, s), a0, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 w0;
#pragma unroll
      for (int i = 0; i < 4; ++i) w0[i] = (bf16)a0[4 * g + i];
      const int k = hd * 32 + 8 * g + 4 * hh;
      *reinterpret_cast<bf16x4*>(sT + swza(rr, k >> 3) + (k & 7) * 2) = w0;
    }
    LB_SYNC();

    // o^T: rows c in [CW hd, CW hd + CW) as NB 32-row blocks, columns = the 32 pixels
    f32x16 o0 = 0, o1 = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bf16x8 tv = *reinterpret_cast<const bf16x8*>(sT + swza(rr, 2 * s + hh));
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo0[s], tv, o0, 0, 0, 0);
      if (NB == 2) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo1[s], tv, o1, 0, 0, 0);
    }
    float ss = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = hd * CW + 8 * g + 4 * hh;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sB + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o0[4 * g + i] += b0[i];
        ss += o0[4 * g + i] * o0[4 * g + i];
      }
      if (NB == 2) {
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sB + c + 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o1[4 * g + i] += b1[i];
          ss += o1[4 * g + i] * o1[4 * g + i];
        }
      }
    }
    ss += __shfl_xor(ss, 32, 64);
    if (hh == 0) sS[hd * TM2 + r] = ss;
    LB_SYNC();                                              // also: every wave is done reading the att tile
    {
      const float n0 = sS[r] + sS[TM2 + r] + sS[2 * TM2 + r] + sS[3 * TM2 + r];
      const float rn = 1.0f / fmaxf(sqrtf(n0), 1e-12f);
      // y = RMSNorm(o) * g2 + x formed in place: this lane owns pixel r, channels c..c+3 of each block
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = hd * CW + 8 * g + 4 * hh;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sG + c);
        bf16x4* x0p = reinterpret_cast<bf16x4*>(A + swz2<C>(rr, c >> 3) + (c & 7) * 2);
        const bf16x4 x0v = *x0p;
        bf16x4 y0;
#pragma unroll
        for (int i = 0; i < 4; ++i) y0[i] = (bf16)(o0[4 * g + i] * rn * g0[i] + (float)x0v[i]);
        *x0p = y0;
        if (NB == 2) {
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(sG + c + 32);
          bf16x4* x1p = reinterpret_cast<bf16x4*>(A + swz2<C>(rr, (c + 32) >> 3) + (c & 7) * 2);
          const bf16x4 x1v = *x1p;
          bf16x4 y1;
#pragma unroll
          for (int i = 0; i < 4; ++i) y1[i] = (bf16)(o1[4 * g + i] * rn * g1[i] + (float)x1v[i]);
          *x1p = y1;
        }
      }
    }
    LB_SYNC();
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int q = tid + NTH2 * i;
      const int row = q / LaDims<C>::NCH, c16 = q % LaDims<C>::NCH;
      const bf16x8 yv = *reinterpret_cast<const bf16x8*>(A + swz2<C>(row, c16));
      const size_t yo = ((size_t)b * p.N + px0 + row) * C + c16 * 8;
      *reinterpret_cast<bf16x8*>(p.y + yo) = yv;
      if (p.yq) mx_store_twin(yv, p.yq, p.ys, yo, tid & 3);
    }
    // tile t+1 must have landed; this iteration's DMA of tile t+2 (PPW + 1 pieces) and its PPW stores stay in flight
    if (t + 2 < T) wait_all_but<2 * PPW + 1>(); else wait_all_but<PPW>();
    LB_SYNC();
  }
}

}  // namespace

// wkv: [256 rows = k | v][C], wq: [128][C], wout: [C][128]; all bf16 row-major with the RMSNorm gain folded in
void linattn_fused256_pack(const float* to_qkv /*[384][C]*/, const float* norm_g /*[C]*/, const float* to_out /*[C][128]*/, int C,
                           std::vector<unsigned short>& wkv, std::vector<unsigned short>& wq, std::vector<unsigned short>& wout) {
  const float sq = sqrtf((float)C);
  wkv.assign((size_t)256 * C, 0);
  wq.assign((size_t)128 * C, 0);
  wout.assign((size_t)C * 128, 0);
  for (int row = 0; row < 256; ++row)
    for (int c = 0; c < C; ++c) wkv[(size_t)row * C + c] = f32_to_bf16_host(to_qkv[(size_t)(128 + row) * C + c] * (norm_g[c] * sq));
  for (int d = 0; d < 128; ++d)
    for (int c = 0; c < C; ++c) wq[(size_t)d * C + c] = f32_to_bf16_host(to_qkv[(size_t)d * C + c] * (norm_g[c] * sq));
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < 128; ++k) wout[(size_t)c * 128 + k] = f32_to_bf16_host(to_out[(size_t)c * 128 + k]);
}

bool linattn_fused256_eligible(int C, int heads, int dh, int N, bool is_bf16) {
  return is_bf16 && C == 256 && heads == 4 && dh == 32 && N % 64 == 0 && (size_t)N * C * 2 < (1ull << 31);
}

template <int C>
static int launch_la_t(const void* x, void* y, int B, int N, const void* wkv, const void* wq, const void* wout, const float* bout,
                       const float* g2_scaled, float* pm, float* pl, float* pctx, float* ctxn, float* rinv, int strip,
                       hipStream_t st, void* y_q, void* y_s) {
  const int nstrips = cdiv(N, strip);
  const size_t bh = (size_t)B * 4;
  static bool attr[64] = {};
  const int lds1 = RING2 * LaDims<C>::TILE + 2 * TM2 * 4;
  const int lds2 = RING2 * LaDims<C>::TILE + ATT2 + 4 * TM2 * 4 + RING2 * 4 * 64 * 4 + 2 * C * 4;
  if (DeviceSetup once(attr); once.need) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&la1_t_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&la2_t_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    once.done();
  }
  if (strip % TM2 || N % TM2) SRGD_FAIL("linattn_fused256: N and the strip length must be multiples of 32");
  hipLaunchKernelGGL(la1_t_kernel<C>, dim3(nstrips, B), dim3(NTH2), lds1, st, (const bf16*)x, N, (const bf16*)wkv, strip, pm, pl,
                     pctx, rinv);
  SRGD_HIP(hipGetLastError());
  SRGD_TRY(linear_attention_combine(pm, pl, pctx, (int)bh, nstrips, 1.0f / sqrtf(32.0f), ctxn, st));
  La2Args256 a;
  a.x = (const bf16*)x; a.y = (bf16*)y; a.N = N; a.wq = (const bf16*)wq; a.wout = (const bf16*)wout; a.bout = bout;
  a.g2 = g2_scaled; a.ctxn = ctxn; a.rinv = rinv;
  a.yq = (unsigned char*)y_q; a.ys = (unsigned char*)y_s;
  const int ntiles = N / TM2;
  int tpw = 1;
  while (tpw < 32 && (long)B * cdiv(ntiles, tpw * 2) >= 1024) tpw *= 2;
  a.tiles_per_wg = tpw;
  hipLaunchKernelGGL(la2_t_kernel<C>, dim3(cdiv(ntiles, tpw), B), dim3(NTH2), lds2, st, a);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int linattn_fused256(const void* x, void* y, int B, int N, int C, const void* wkv, const void* wq, const void* wout,
                     const float* bout, const float* g2_scaled, float* pm, float* pl, float* pctx, float* ctxn, float* rinv,
                     int strip, hipStream_t st, void* y_q, void* y_s) {
  if (C == 256) return launch_la_t<256>(x, y, B, N, wkv, wq, wout, bout, g2_scaled, pm, pl, pctx, ctxn, rinv, strip, st, y_q, y_s);
  // (the C = 128 instance of these 32-pixel-tile kernels measured 0.4 % slower end to end than linattn_fused.hip's 64-pixel
  // kernels and is no longer instantiated: DESIGN 4.3)
  SRGD_FAIL("linattn_fused256: C must be 256");
}

}  // namespace srgd
