// Implicit-GEMM convolution for gfx950 (MI355X): 3x3 / 1x1 / 2x2-stride-2 / generic KSxKS,
// NHWC activations, up to two channel-concatenated sources (replaces torch.cat on the skip
// path, reference model.py:713,716,722), fused epilogues:
//   + bias, + residual, PixelShuffle(2)+SiLU store (model.py:80-84), and per-(sample,group)
//   partial sum / sum-of-squares for the GroupNorm that follows (model.py:246-247).
//
// GEMM view: M = B*Hout*Wout output pixels, N = Cout, K = KS*KS*Cin.  One workgroup = 256
// threads = 4 waves (2x2), tile 128(M) x 128(N), each wave 64x64 = 2x2 MFMA 32x32 blocks.
// A K-step is one (tap, channel-chunk): the A tile is gathered straight from the NHWC input
// (zero fill outside the image), the B tile from weights packed [tap][CoutPad][Cin] (k
// contiguous), both staged global -> registers -> LDS (rows padded by 16 B: conflict-free
// ds_read_b128) with a 2-deep LDS ring so the next step's global loads overlap the MFMAs.
//   bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate);  fp32: v_mfma_f32_32x32x2_f32 (exact fp32).
#include <cstdlib>

#include "kernels.hpp"

namespace srgd {

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for
// the output stores issued just before the GroupNorm statistics are reduced (thousands of cycles per tile)
#define IGEMM_LDS_BARRIER()                                  \
  do {                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_s_barrier();                            \
    __builtin_amdgcn_sched_barrier(0);                       \
  } while (0)

template <typename T> struct Frag;                 // one lane's 16-byte operand fragment
template <> struct Frag<float> { f32x4 v; };
template <> struct Frag<bf16> { bf16x8 v; };

__device__ __forceinline__ void mma(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
// fp32: the 16 B hold k = 4h..4h+3 of an 8-deep block; MFMA j contracts {j (h=0), 4+j (h=1)} -
// any k pairing is valid as long as A and B use the same one.
__device__ __forceinline__ void mma(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// Direct epilogue (accumulator layout: lane = output channel, register = pixel) for 4-byte outputs: (acc * acc_scale) + bias,
// + residual or SiLU + PixelShuffle(2) store, GroupNorm column sums into s1 / s2.  acc_scale is 1 on the exact-fp32 route (a
// multiplication by one changes no bit) and the inverse power-of-two weight scale on the split-operand route.
template <typename T, bool PRECISE>
__device__ __forceinline__ void epilogue_direct(const ConvArgs& p, const f32x16& acc00, const f32x16& acc01, const f32x16& acc10,
                                                const f32x16& acc11, int n0, int m0, int r0, int HWo, int wm, int wn, int r, int h,
                                                float acc_scale, float (&s1)[2], float (&s2)[2]) {
  const int CoutPS = p.Cout >> 2;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + r;
    const bool cval = col < p.Cout;
    const float bias = (cval && p.bias) ? p.bias[col] : 0.f;
    int ps_c = 0, ps_i = 0, ps_j = 0;
    if (p.mode == CONV_PIXEL_SHUFFLE_SILU && cval) {
      const int ij = col / CoutPS;
      ps_c = col - ij * CoutPS;
      ps_i = ij >> 1;
      ps_j = ij & 1;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const f32x16 accv = mi == 0 ? (ni == 0 ? acc00 : acc01) : (ni == 0 ? acc10 : acc11);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int m = m0 + row;
        float v = accv[reg] * acc_scale + bias;
        if (r0 + row < HWo && cval) {
          s1[ni] += v;
          s2[ni] += v * v;
          if (p.mode == CONV_PIXEL_SHUFFLE_SILU) {
            const int b = m / HWo, rem = m - b * HWo;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const size_t o = ((size_t)(b * 2 * p.Hout + 2 * oy + ps_i) * (2 * p.Wout) + 2 * ox + ps_j) * CoutPS + ps_c;
            reinterpret_cast<T*>(p.out)[o] = from_f32<T>(silu<PRECISE>(v));
          } else {
            const size_t o = (size_t)m * p.Cout + col;
            if (p.residual) v += to_f32<T>(reinterpret_cast<const T*>(p.residual)[o]);
            reinterpret_cast<T*>(p.out)[o] = from_f32<T>(v);
          }
        }
      }
    }
  }
}

// GroupNorm partial sums of a 128 x 128 tile from the per-lane column sums (both implicit-GEMM kernels)
__device__ __forceinline__ void epilogue_gn_partial(const ConvArgs& p, char* smem, const float (&s1)[2], const float (&s2)[2], int tid,
                                                    int lane, int wave, int wm, int wn, int r, int h, int n0, int tb, int tps,
                                                    int tslot) {
  if (p.gn_partial) {
    // column sums -> LDS (fixed order: deterministic), then one thread per group
    float* cs = reinterpret_cast<float*>(smem);          // [2 (wm)][BN][2]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const float t1 = s1[ni] + __shfl_xor(s1[ni], 32, 64);
      const float t2 = s2[ni] + __shfl_xor(s2[ni], 32, 64);
      if (h == 0) {
        const int cl = wn * 64 + ni * 32 + r;
        cs[(wm * BN + cl) * 2 + 0] = t1;
        cs[(wm * BN + cl) * 2 + 1] = t2;
      }
    }
    IGEMM_LDS_BARRIER();
    // columns summed over the 2 row blocks by 128 threads, then a shuffle tree over each group's span of columns (a serial
    // walk by one thread per group cost thousands of cycles per tile: see conv3x3_bf16.hip)
    const int cpg = p.Cout / p.groups;                   // channels per group (<= BN, divides BN: 16, 32, 64 or 128)
    float a1 = 0.f, a2 = 0.f;
    if (tid < BN) {
      a1 = cs[tid * 2 + 0] + cs[(BN + tid) * 2 + 0];
      a2 = cs[tid * 2 + 1] + cs[(BN + tid) * 2 + 1];
      for (int o = 1; o < cpg && o < 64; o <<= 1) {
        a1 += __shfl_xor(a1, o, 64);
        a2 += __shfl_xor(a2, o, 64);
      }
    }
    if (cpg == BN) {                                     // the group spans both waves (uniform branch)
      IGEMM_LDS_BARRIER();
      if (tid < BN && lane == 0) { cs[wave * 2 + 0] = a1; cs[wave * 2 + 1] = a2; }
      IGEMM_LDS_BARRIER();
      if (tid == 0) { a1 = cs[0] + cs[2]; a2 = cs[1] + cs[3]; }
    }
    if (tid < BN && (tid % cpg) == 0) {
      const int g = n0 / cpg + tid / cpg;
      if (g < p.groups) {
        float* dst = p.gn_partial + ((size_t)(tb * p.groups + g) * tps + tslot) * 2;
        dst[0] = a1;
        dst[1] = a2;
      }
    }
  }
}

// fp32 with 16-channel chunks: capped at 128 registers (115 used, no spills) so that four workgroups share a CU - the LDS ring
// (40,960 B) allows exactly four; left alone the compiler takes 93 VGPRs + 64 AGPRs and only three fit
template <typename T, int BKC, bool PRECISE>
__global__ __launch_bounds__(NT, (sizeof(T) == 4 && BKC == 16) ? 4 : 1) void conv_igemm_kernel(ConvArgs p) {
  constexpr int ROWB = BKC * (int)sizeof(T);       // bytes of K per tile row: 32 / 64 / 128
  constexpr int CPR = ROWB / 16;                   // 16-byte chunks per row
  constexpr int STRIDE = ROWB + 16;                // padded LDS row stride
  constexpr int PER = (BM * CPR) / NT;             // chunks per thread per operand
  constexpr int EPC = 16 / (int)sizeof(T);         // elements per chunk
  static_assert(PER >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS ring: [buf 0: A | B][buf 1: A | B]
  auto sA = [&](int buf) -> char* { return smem + buf * (2 * BM * STRIDE); };
  auto sB = [&](int buf) -> char* { return smem + buf * (2 * BM * STRIDE) + BM * STRIDE; };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = p.CoutPad / BN;
  // XCD-aware order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (one L2 each); renumber so
  // that each XCD walks a contiguous run of logical tiles - the n-tiles of one m-tile (same A rows) and the
  // neighbouring m-tiles (shared 3x3 halo rows) then hit the same L2 instead of 8 different ones
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7;
  const int xcd = blockIdx.x & 7, xloc = blockIdx.x >> 3;
  const int bid = xcd * xq + (xcd < xr ? xcd : xr) + xloc;
  const int mt = bid / n_tiles, nt = bid - mt * n_tiles;
  const int n0 = nt * BN;
  const int HWo = p.Hout * p.Wout;
  // M tiles never straddle samples (the GroupNorm partials are per sample): sample tb owns ceil(HWo / BM) tiles and the
  // rows past its last pixel are masked.  For HWo % BM == 0 (every production shape) this is the plain M = B*HWo tiling.
  const int tps = (HWo + BM - 1) / BM;
  const int tb = mt / tps, tslot = mt - tb * tps;
  const int r0 = tslot * BM;                       // first pixel of the tile inside its sample
  const int m0 = tb * HWo + r0;                    // ... and in the flattened [B*HWo] order
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / BKC;
  const int steps = p.KH * p.KW * CC;

  // Per-thread staging state: PER (<= 4) 16-byte chunks of A and of B per K-step.  Everything is a NAMED
  // scalar/vector (token-pasted), never an indexed array: hipcc keeps indexed fragment arrays in scratch,
  // which puts a vmcnt(0) behind every global load and serialises the pipeline.
#define K_DECL(I)                                                        \
  int iy0_##I = 0, ix0_##I = 0, ib_##I = 0, arow_##I = 0, aq_##I = 0;       \
  bool mval_##I = false, ok_##I = false;                                    \
  Frag<T> ra_##I, rb_##I;                                                   \
  if (I < PER) {                                                            \
    const int cid = tid + NT * I;                                           \
    arow_##I = cid / CPR;                                                   \
    aq_##I = cid - arow_##I * CPR;                                          \
    mval_##I = r0 + arow_##I < HWo;                                         \
    const int b = tb, rem = mval_##I ? r0 + arow_##I : 0;                   \
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;                    \
    iy0_##I = oy * p.stride - p.pad;                                        \
    ix0_##I = ox * p.stride - p.pad;                                        \
    ib_##I = b * p.Hin * p.Win;                                             \
  }
  K_DECL(0) K_DECL(1) K_DECL(2) K_DECL(3)
#undef K_DECL

  // branch-free zero fill: out-of-image taps read a valid dummy address and are zeroed when staged to LDS
#define K_LOAD1(I)                                                                                              \
  if (I < PER) {                                                                                                   \
    const int iy_ = iy0_##I + dy_, ix_ = ix0_##I + dx_;                                                            \
    ok_##I = mval_##I && iy_ >= 0 && iy_ < p.Hin && ix_ >= 0 && ix_ < p.Win;                                       \
    const size_t off_ =                                                                                            \
        ok_##I ? ((size_t)(ib_##I + iy_ * p.Win + ix_) * Cs_ + coff_ + aq_##I * EPC) * sizeof(T) : (size_t)0;      \
    ra_##I = *reinterpret_cast<const Frag<T>*>(src_ + off_);                                                       \
    const size_t woff_ = ((size_t)(tap_ * p.CoutPad + n0 + arow_##I) * Cin + c_ + aq_##I * EPC) * sizeof(T);       \
    rb_##I = *reinterpret_cast<const Frag<T>*>((const char*)p.w + woff_);                                          \
  }
#define K_LOAD_STEP(S)                                                    \
  {                                                                          \
    const int tap_ = (S) / CC, cc_ = (S)-tap_ * CC;                          \
    const int dy_ = tap_ / p.KW, dx_ = tap_ - dy_ * p.KW;                    \
    const int c_ = cc_ * BKC;                                                \
    const bool first_ = c_ < p.C0;                                           \
    const char* src_ = first_ ? (const char*)p.in0 : (const char*)p.in1;     \
    const int Cs_ = first_ ? p.ps0 : p.ps1;                                  \
    const int coff_ = first_ ? c_ : c_ - p.C0;                               \
    K_LOAD1(0) K_LOAD1(1) K_LOAD1(2) K_LOAD1(3)                  \
  }
#define K_STORE1(I, BUF)                                                                   \
  if (I < PER) {                                                                              \
    if (!ok_##I) ra_##I.v = 0; /* masked here, not at the load: keeps the loads in flight */  \
    *reinterpret_cast<Frag<T>*>(sA(BUF) + arow_##I * STRIDE + aq_##I * 16) = ra_##I;          \
    *reinterpret_cast<Frag<T>*>(sB(BUF) + arow_##I * STRIDE + aq_##I * 16) = rb_##I;          \
  }
#define K_STORE_STEP(BUF) { K_STORE1(0, BUF) K_STORE1(1, BUF) K_STORE1(2, BUF) K_STORE1(3, BUF) }

  f32x16 acc00 = 0, acc01 = 0, acc10 = 0, acc11 = 0;

  K_LOAD_STEP(0);
  K_STORE_STEP(0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) K_LOAD_STEP(s + 1);
    const char* a_base = sA(buf) + (wm * 64 + r) * STRIDE + h * 16;
    const char* b_base = sB(buf) + (wn * 64 + r) * STRIDE + h * 16;
#pragma unroll
    for (int s2 = 0; s2 < ROWB / 32; ++s2) {
      const Frag<T> fa0 = *reinterpret_cast<const Frag<T>*>(a_base + s2 * 32);
      const Frag<T> fa1 = *reinterpret_cast<const Frag<T>*>(a_base + 32 * STRIDE + s2 * 32);
      const Frag<T> fb0 = *reinterpret_cast<const Frag<T>*>(b_base + s2 * 32);
      const Frag<T> fb1 = *reinterpret_cast<const Frag<T>*>(b_base + 32 * STRIDE + s2 * 32);
      mma(acc00, fa0, fb0);
      mma(acc01, fa0, fb1);
      mma(acc10, fa1, fb0);
      mma(acc11, fa1, fb1);
    }
    if (s + 1 < steps) K_STORE_STEP(buf ^ 1);
    __syncthreads();
  }
#undef K_LOAD1
#undef K_LOAD_STEP
#undef K_STORE1
#undef K_STORE_STEP

  // ------------------------------- epilogue -------------------------------------------
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  const int CoutPS = p.Cout >> 2;
  if constexpr (sizeof(T) == 2) {
    // bf16: the accumulator layout (lane = output channel, register = pixel) would store 2 bytes per lane.
    // Transpose the tile through LDS ([128 pixels][128 ch], rows padded to 272 B) and write whole channel
    // rows, 16 B per lane: 8 store instructions per thread instead of 64 (the K loop of a 1x1 conv is only
    // 2-16 steps long, so the store tail decides its speed).  The final __syncthreads() of the K loop has
    // retired every operand read, so the ring buffers can be reused.
    constexpr int EROW = BN * 2 + 16;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int cl = wn * 64 + ni * 32 + r;
      const int col = n0 + cl;
      const bool cval = col < p.Cout;
      const float bias = (cval && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x16 accv = mi == 0 ? (ni == 0 ? acc00 : acc01) : (ni == 0 ? acc10 : acc11);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          float v = accv[reg] + bias;
          if (r0 + row < HWo && cval) {
            s1[ni] += v;
            s2[ni] += v * v;
          }
          if (p.mode == CONV_PIXEL_SHUFFLE_SILU) v = silu<PRECISE>(v);
          *reinterpret_cast<bf16*>(smem + row * EROW + cl * 2) = (bf16)v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (BM * 16) / NT; ++i) {
      const int q = tid + NT * i;                          // 16-byte chunk: tile row q/16, columns (q%16)*8..+7
      const int row = q >> 4, c16 = q & 15;
      const int m = m0 + row, col = n0 + c16 * 8;
      if (r0 + row < HWo && col < p.Cout) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * EROW + c16 * 16);
        if (p.mode == CONV_PIXEL_SHUFFLE_SILU) {
          const int ij = col / CoutPS, pc = col - ij * CoutPS;
          const int b = m / HWo, rem = m - b * HWo;
          const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
          const size_t o =
              ((size_t)(b * 2 * p.Hout + 2 * oy + (ij >> 1)) * (2 * p.Wout) + 2 * ox + (ij & 1)) * CoutPS + pc;
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + o) = v;
        } else {
          const size_t o = (size_t)m * p.Cout + col;
          if (p.residual) {
            const bf16x8 rr = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.residual) + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rr[e]);
          }
          if (p.gn_res_src) {
            // h + res_conv(x) with h = SiLU(GroupNorm(block2 conv)) evaluated here from the raw conv output
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.gn_res_src) + o);
            const size_t co = (size_t)(m / HWo) * p.Cout + col;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.gn_res_a + co), a1 = *reinterpret_cast<const f32x4*>(p.gn_res_a + co + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.gn_res_b + co), b1 = *reinterpret_cast<const f32x4*>(p.gn_res_b + co + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = (bf16)((float)v[e] + silu<PRECISE>(a0[e] * (float)raw[e] + b0[e]));
              v[4 + e] = (bf16)((float)v[4 + e] + silu<PRECISE>(a1[e] * (float)raw[4 + e] + b1[e]));
            }
          }
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + o) = v;
        }
      }
    }
    if (p.gn_partial) IGEMM_LDS_BARRIER();                 // staged tile consumed before the statistics reuse LDS
  } else {
    epilogue_direct<T, PRECISE>(p, acc00, acc01, acc10, acc11, n0, m0, r0, HWo, wm, wn, r, h, 1.0f, s1, s2);
  }

  epilogue_gn_partial(p, smem, s1, s2, tid, lane, wave, wm, wn, r, h, n0, tb, tps, tslot);
}

// ---- split-operand variant (SRGD_PRECISION_SPLIT) --------------------------------------------------------------------------------
// Same tiling, fp32 tensors in HBM, the contraction on the 16-bit matrix cores with (hi, lo) operand pairs:
// x * w ~= x_hi * w_hi + x_lo * w_hi + x_hi * w_lo, three v_mfma_f32_32x32x16_{f16,bf16} per fragment pair (conv3x3_split.hip has
// the arithmetic and the 3x3 fast path; this kernel serves the pointwise, 2x2 / stride-2 and odd-sized layers of that mode).
// A K-step is one (tap, 32-channel chunk): the fp32 A rows are split in registers on their way to LDS, the weights come pre-split
// from pack_conv_weights_split ([tap][CoutPad][Cin / 32][hi 32 | lo 32] 16-bit).  LDS rows: [hi 64 B | lo 64 B | 16 B pad].
constexpr int SP_KC = 32, SP_STRIDE = 144;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <bool F16>
__device__ __forceinline__ void split4(const f32x4& x, unsigned (&hi)[2], unsigned (&lo)[2]) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float a = x[2 * k], b = x[2 * k + 1];
    if constexpr (F16) {
      a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
      b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
      const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
      const f32x2 hf = __builtin_convertvector(hh, f32x2);
      const f16x2 ll = __builtin_convertvector(f32x2{a - hf[0], b - hf[1]}, f16x2);
      hi[k] = __builtin_bit_cast(unsigned, hh);
      lo[k] = __builtin_bit_cast(unsigned, ll);
    } else {
      const bf16x2 hh = __builtin_convertvector(f32x2{a, b}, bf16x2);
      const unsigned hb = __builtin_bit_cast(unsigned, hh);
      const bf16x2 ll = __builtin_convertvector(f32x2{a - __uint_as_float(hb << 16), b - __uint_as_float(hb & 0xffff0000u)}, bf16x2);
      hi[k] = hb;
      lo[k] = __builtin_bit_cast(unsigned, ll);
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(NT, 2) void conv_igemm_split_kernel(ConvArgs p, float w_inv_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS ring: [buf 0: A rows | B rows][buf 1: A rows | B rows]
  auto sA = [&](int buf) -> char* { return smem + buf * (2 * BM * SP_STRIDE); };
  auto sB = [&](int buf) -> char* { return smem + buf * (2 * BM * SP_STRIDE) + BM * SP_STRIDE; };
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = p.CoutPad / BN;
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7;
  const int xcd = blockIdx.x & 7, xloc = blockIdx.x >> 3;
  const int bid = xcd * xq + (xcd < xr ? xcd : xr) + xloc;
  const int mt = bid / n_tiles, nt = bid - mt * n_tiles;
  const int n0 = nt * BN;
  const int HWo = p.Hout * p.Wout;
  const int tps = (HWo + BM - 1) / BM;
  const int tb = mt / tps, tslot = mt - tb * tps;
  const int r0 = tslot * BM;
  const int m0 = tb * HWo + r0;
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / SP_KC;
  const int steps = p.KH * p.KW * CC;

  // A: 128 rows x 8 four-channel pieces of fp32 (16 B each) = 1024 pieces, four per thread: piece id = tid + 256 I -> row id >> 3,
  // channels 4 (id & 7) ..  B: 128 rows x 8 16-byte pieces (hi 0..3 | lo 4..7) of 16-bit weights, same mapping.
#define SP_DECL(I)                                                        \
  int iy0_##I, ix0_##I, ib_##I;                                             \
  bool mval_##I, ok_##I = false;                                            \
  f32x4 ra_##I;                                                             \
  u32x4 rb_##I;                                                             \
  const int row_##I = (tid + NT * I) >> 3, q_##I = (tid + NT * I) & 7;      \
  {                                                                         \
    mval_##I = r0 + row_##I < HWo;                                          \
    const int rem = mval_##I ? r0 + row_##I : 0;                            \
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;                    \
    iy0_##I = oy * p.stride - p.pad;                                        \
    ix0_##I = ox * p.stride - p.pad;                                        \
    ib_##I = tb * p.Hin * p.Win;                                            \
  }
  SP_DECL(0) SP_DECL(1) SP_DECL(2) SP_DECL(3)
#undef SP_DECL
#define SP_LOAD1(I)                                                                                                  \
  {                                                                                                                  \
    const int iy_ = iy0_##I + dy_, ix_ = ix0_##I + dx_;                                                              \
    ok_##I = mval_##I && iy_ >= 0 && iy_ < p.Hin && ix_ >= 0 && ix_ < p.Win;                                         \
    const size_t off_ = ok_##I ? ((size_t)(ib_##I + iy_ * p.Win + ix_) * Cs_ + coff_ + q_##I * 4) * 4 : (size_t)0;   \
    ra_##I = *reinterpret_cast<const f32x4*>(src_ + off_);                                                           \
    const size_t woff_ = (((size_t)(tap_ * p.CoutPad + n0 + row_##I) * CC + cc_) * 8 + q_##I) * 16;                  \
    rb_##I = *reinterpret_cast<const u32x4*>((const char*)p.w + woff_);                                              \
  }
#define SP_LOAD_STEP(S)                                                   \
  {                                                                          \
    const int tap_ = (S) / CC, cc_ = (S)-tap_ * CC;                          \
    const int dy_ = tap_ / p.KW, dx_ = tap_ - dy_ * p.KW;                    \
    const int c_ = cc_ * SP_KC;                                              \
    const bool first_ = c_ < p.C0;                                           \
    const char* src_ = first_ ? (const char*)p.in0 : (const char*)p.in1;     \
    const int Cs_ = first_ ? p.ps0 : p.ps1;                                  \
    const int coff_ = first_ ? c_ : c_ - p.C0;                               \
    SP_LOAD1(0) SP_LOAD1(1) SP_LOAD1(2) SP_LOAD1(3)                          \
  }
#define SP_STORE1(I, BUF)                                                                       \
  {                                                                                             \
    if (!ok_##I) ra_##I = f32x4{0.f, 0.f, 0.f, 0.f};                                            \
    unsigned hi_[2], lo_[2];                                                                    \
    split4<F16>(ra_##I, hi_, lo_);                                                              \
    char* da_ = sA(BUF) + row_##I * SP_STRIDE + q_##I * 8;                                      \
    *reinterpret_cast<uint2*>(da_) = make_uint2(hi_[0], hi_[1]);                                \
    *reinterpret_cast<uint2*>(da_ + 64) = make_uint2(lo_[0], lo_[1]);                           \
    *reinterpret_cast<u32x4*>(sB(BUF) + row_##I * SP_STRIDE + q_##I * 16) = rb_##I;             \
  }
#define SP_STORE_STEP(BUF) { SP_STORE1(0, BUF) SP_STORE1(1, BUF) SP_STORE1(2, BUF) SP_STORE1(3, BUF) }

  f32x16 acc00 = 0, acc01 = 0, acc10 = 0, acc11 = 0;
  auto mma = [&](f32x16& c, const u32x4& a, const u32x4& b) {
    if constexpr (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };

  SP_LOAD_STEP(0);
  SP_STORE_STEP(0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) SP_LOAD_STEP(s + 1);
    const char* a_base = sA(buf) + (wm * 64 + r) * SP_STRIDE + h * 16;
    const char* b_base = sB(buf) + (wn * 64 + r) * SP_STRIDE + h * 16;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const u32x4 ah0 = *reinterpret_cast<const u32x4*>(a_base + s2 * 32), al0 = *reinterpret_cast<const u32x4*>(a_base + 64 + s2 * 32);
      const u32x4 ah1 = *reinterpret_cast<const u32x4*>(a_base + 32 * SP_STRIDE + s2 * 32),
                  al1 = *reinterpret_cast<const u32x4*>(a_base + 32 * SP_STRIDE + 64 + s2 * 32);
      const u32x4 bh0 = *reinterpret_cast<const u32x4*>(b_base + s2 * 32), bl0 = *reinterpret_cast<const u32x4*>(b_base + 64 + s2 * 32);
      const u32x4 bh1 = *reinterpret_cast<const u32x4*>(b_base + 32 * SP_STRIDE + s2 * 32),
                  bl1 = *reinterpret_cast<const u32x4*>(b_base + 32 * SP_STRIDE + 64 + s2 * 32);
      mma(acc00, al0, bh0); mma(acc01, al0, bh1); mma(acc10, al1, bh0); mma(acc11, al1, bh1);
      mma(acc00, ah0, bl0); mma(acc01, ah0, bl1); mma(acc10, ah1, bl0); mma(acc11, ah1, bl1);
      mma(acc00, ah0, bh0); mma(acc01, ah0, bh1); mma(acc10, ah1, bh0); mma(acc11, ah1, bh1);
    }
    if (s + 1 < steps) SP_STORE_STEP(buf ^ 1);
    __syncthreads();
  }
#undef SP_LOAD1
#undef SP_LOAD_STEP
#undef SP_STORE1
#undef SP_STORE_STEP

  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  epilogue_direct<float, true>(p, acc00, acc01, acc10, acc11, n0, m0, r0, HWo, wm, wn, r, h, w_inv_scale, s1, s2);
  epilogue_gn_partial(p, smem, s1, s2, tid, lane, wave, wm, wn, r, h, n0, tb, tps, tslot);
}

template <typename T, int BKC, bool PRECISE>
int launch(const ConvArgs& a, hipStream_t st) {
  constexpr int ROWB = BKC * (int)sizeof(T);
  constexpr int STRIDE = ROWB + 16;
  size_t lds = (size_t)4 * BM * STRIDE;
  if (sizeof(T) == 2) lds = std::max(lds, (size_t)BM * (BN * 2 + 16));     // bf16 epilogue staging tile
  const int grid = a.B * cdiv(a.Hout * a.Wout, BM) * (a.CoutPad / BN);
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BKC, PRECISE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once.done();
  }
  hipLaunchKernelGGL((conv_igemm_kernel<T, BKC, PRECISE>), dim3(grid), dim3(NT), lds, st, a);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int conv_tile_n() { return BN; }
int conv_tile_m() { return BM; }

// Largest K chunk (in channels) the kernel supports for this layer, 0 if none.
static int pick_bkc(bool is_bf16, int C0, int C1) {
  const int cands_bf16[3] = {64, 32, 16};
  const int cands_f32[2] = {32, 16};
  const int* c = is_bf16 ? cands_bf16 : cands_f32;
  const int n = is_bf16 ? 3 : 2;
  // fp32: 16-channel chunks (64-byte rows, 41 KB of LDS: three workgroups per CU instead of two) measured +4.4 % end to end
  // over 32-channel ones in the parity mode; same k order, bit-identical results.
  for (int i = is_bf16 ? 0 : 1; i < n; ++i)
    if (C0 % c[i] == 0 && (C1 == 0 || C1 % c[i] == 0)) return c[i];
  return 0;
}

bool conv_igemm_split_eligible(const ConvArgs& a) {
  if (a.CoutPad % BN != 0 || a.CoutPad < a.Cout) return false;
  if (a.C0 % SP_KC || a.C1 % SP_KC || a.ps0 % 4 || (a.C1 && a.ps1 % 4)) return false;
  if (a.gn_res_src || a.out_q || a.eps4) return false;
  if (a.gn_partial && (a.Cout % a.groups != 0 || (a.Cout / a.groups) > BN || BN % (a.Cout / a.groups) != 0)) return false;
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU && (a.Cout % 4 != 0 || a.residual || a.gn_partial)) return false;
  return true;
}

// [tap][CoutPad][Cin] fp32 (pack_conv_weights with to_bf16 = false) -> [tap][CoutPad][Cin / 32][hi 32 | lo 32] 16-bit halves of
// w * scale (see conv3x3_split.hip: split_weight_scale, split_halves_host)
void pack_conv_weights_split(const float* src_tap_o_i, int taps, int Cin, int CoutPad, bool f16, float scale,
                             std::vector<unsigned short>& out) {
  const int CC = Cin / SP_KC;
  out.assign((size_t)taps * CoutPad * CC * 64, 0);
  for (size_t row = 0; row < (size_t)taps * CoutPad; ++row)
    for (int cc = 0; cc < CC; ++cc) {
      unsigned short* dst = out.data() + (row * CC + cc) * 64;
      for (int e = 0; e < SP_KC; ++e) split_halves_host(src_tap_o_i[row * Cin + cc * SP_KC + e] * scale, f16, &dst[e], &dst[32 + e]);
    }
}

int conv_igemm_split(const ConvArgs& a, const void* packed_w, float w_inv_scale, bool f16, hipStream_t st) {
  if (!conv_igemm_split_eligible(a)) SRGD_FAIL("conv_igemm_split: shape not eligible");
  if (a.C1 > 0 && a.in1 == nullptr) SRGD_FAIL("conv_igemm_split: second source missing");
  ConvArgs p = a;
  p.w = packed_w;
  const size_t lds = (size_t)4 * BM * SP_STRIDE;
  const int grid = a.B * cdiv(a.Hout * a.Wout, BM) * (a.CoutPad / BN);
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once.done();
  }
  if (f16) hipLaunchKernelGGL((conv_igemm_split_kernel<true>), dim3(grid), dim3(NT), lds, st, p, w_inv_scale);
  else hipLaunchKernelGGL((conv_igemm_split_kernel<false>), dim3(grid), dim3(NT), lds, st, p, w_inv_scale);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int conv_igemm(const ConvArgs& a, bool is_bf16, hipStream_t st) {
  if (a.CoutPad % BN != 0 || a.CoutPad < a.Cout) SRGD_FAIL("conv_igemm: CoutPad must be a multiple of 128 and >= Cout");
  if (a.C1 > 0 && a.in1 == nullptr) SRGD_FAIL("conv_igemm: second source missing");
  if (a.gn_partial) {
    if (a.Cout % a.groups != 0 || (a.Cout / a.groups) > BN || BN % (a.Cout / a.groups) != 0)
      SRGD_FAIL("conv_igemm: unsupported channels-per-group for fused GroupNorm statistics");
  }
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU && (a.Cout % 4 != 0 || a.residual || a.gn_partial))
    SRGD_FAIL("conv_igemm: invalid pixel-shuffle epilogue combination");
  if (is_bf16 && (a.Cout % 8 != 0 || (a.mode == CONV_PIXEL_SHUFFLE_SILU && (a.Cout / 4) % 8 != 0)))
    SRGD_FAIL("conv_igemm: bf16 output channels must be a multiple of 8 (32 for the pixel-shuffle epilogue)");
  if (a.gn_res_src && (!is_bf16 || a.mode != CONV_PLAIN || !a.gn_res_a || !a.gn_res_b))
    SRGD_FAIL("conv_igemm: the fused GroupNorm+residual tail is a bf16 plain-mode epilogue");
  const int bkc = pick_bkc(is_bf16, a.C0, a.C1);
  if (bkc == 0) SRGD_FAIL("conv_igemm: input channels must be a multiple of 16");
  if (is_bf16) {
    if (bkc == 64) return launch<bf16, 64, false>(a, st);
    if (bkc == 32) return launch<bf16, 32, false>(a, st);
    return launch<bf16, 16, false>(a, st);
  }
  if (bkc == 32) return launch<float, 32, true>(a, st);
  return launch<float, 16, true>(a, st);
}

}  // namespace srgd
