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
#include "kernels.hpp"

namespace srgd {

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

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

template <typename T, int BKC, bool PRECISE>
__global__ __launch_bounds__(NT) void conv_igemm_kernel(ConvArgs p) {
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
  const int mt = blockIdx.x / n_tiles, nt = blockIdx.x - mt * n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HWo = p.Hout * p.Wout;
  const int M = p.B * HWo;
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / BKC;
  const int steps = p.KS * p.KS * CC;

  // per-thread staging coordinates
  int iy0[PER], ix0[PER], ib[PER], arow[PER], aq[PER];
  bool mval[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int cid = tid + NT * i;
    arow[i] = cid / CPR;
    aq[i] = cid - arow[i] * CPR;
    const int m = m0 + arow[i];
    mval[i] = m < M;
    const int mm = mval[i] ? m : 0;
    const int b = mm / HWo, rem = mm - b * HWo;
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
    iy0[i] = oy * p.stride - p.pad;
    ix0[i] = ox * p.stride - p.pad;
    ib[i] = b * p.Hin * p.Win;
  }

  Frag<T> ra[PER], rb[PER];
  auto load_step = [&](int s) {
    const int tap = s / CC, cc = s - tap * CC;
    const int dy = tap / p.KS, dx = tap - dy * p.KS;
    const int c = cc * BKC;
    const char* src;
    int Cs, coff;
    if (c < p.C0) { src = (const char*)p.in0; Cs = p.C0; coff = c; }
    else { src = (const char*)p.in1; Cs = p.C1; coff = c - p.C0; }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = mval[i] && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
      Frag<T> z;
      z.v = 0;
      if (ok) {
        const size_t off = ((size_t)(ib[i] + iy * p.Win + ix) * Cs + coff + aq[i] * EPC) * sizeof(T);
        z = *reinterpret_cast<const Frag<T>*>(src + off);
      }
      ra[i] = z;
      const size_t woff = ((size_t)(tap * p.CoutPad + n0 + arow[i]) * Cin + c + aq[i] * EPC) * sizeof(T);
      rb[i] = *reinterpret_cast<const Frag<T>*>((const char*)p.w + woff);
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      *reinterpret_cast<Frag<T>*>(sA(buf) + arow[i] * STRIDE + aq[i] * 16) = ra[i];
      *reinterpret_cast<Frag<T>*>(sB(buf) + arow[i] * STRIDE + aq[i] * 16) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) load_step(s + 1);
    const char* a_base = sA(buf) + (wm * 64 + r) * STRIDE + h * 16;
    const char* b_base = sB(buf) + (wn * 64 + r) * STRIDE + h * 16;
#pragma unroll
    for (int s2 = 0; s2 < ROWB / 32; ++s2) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const Frag<T>*>(a_base + i * 32 * STRIDE + s2 * 32);
        fb[i] = *reinterpret_cast<const Frag<T>*>(b_base + i * 32 * STRIDE + s2 * 32);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) mma(acc[mi][ni], fa[mi], fb[ni]);
    }
    if (s + 1 < steps) store_step(buf ^ 1);
    __syncthreads();
  }

  // ------------------------------- epilogue -------------------------------------------
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
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
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int m = m0 + row;
        float v = acc[mi][ni][reg] + bias;
        if (m < M && cval) {
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
    __syncthreads();
    const int cpg = p.Cout / p.groups;                   // channels per group (<= BN, divides BN)
    const int g_in_tile = BN / cpg;
    if (tid < g_in_tile) {
      const int g = n0 / cpg + tid;
      if (g < p.groups) {
        float a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < cpg; ++c) {
          const int cl = tid * cpg + c;
          a1 += cs[cl * 2 + 0] + cs[(BN + cl) * 2 + 0];
          a2 += cs[cl * 2 + 1] + cs[(BN + cl) * 2 + 1];
        }
        const int b = m0 / HWo;
        const int slot = (m0 - b * HWo) / BM;
        const int nslots = HWo / BM;
        float* dst = p.gn_partial + ((size_t)(b * p.groups + g) * nslots + slot) * 2;
        dst[0] = a1;
        dst[1] = a2;
      }
    }
  }
}

template <typename T, int BKC, bool PRECISE>
int launch(const ConvArgs& a, hipStream_t st) {
  constexpr int ROWB = BKC * (int)sizeof(T);
  constexpr int STRIDE = ROWB + 16;
  const size_t lds = (size_t)4 * BM * STRIDE;
  const int M = a.B * a.Hout * a.Wout;
  const int grid = cdiv(M, BM) * (a.CoutPad / BN);
  static bool attr_set = false;
  if (!attr_set) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BKC, PRECISE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
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
  for (int i = 0; i < n; ++i)
    if (C0 % c[i] == 0 && (C1 == 0 || C1 % c[i] == 0)) return c[i];
  return 0;
}

int conv_igemm(const ConvArgs& a, bool is_bf16, hipStream_t st) {
  if (a.CoutPad % BN != 0 || a.CoutPad < a.Cout) SRGD_FAIL("conv_igemm: CoutPad must be a multiple of 128 and >= Cout");
  if (a.C1 > 0 && a.in1 == nullptr) SRGD_FAIL("conv_igemm: second source missing");
  if (a.gn_partial) {
    if ((a.Hout * a.Wout) % BM != 0) SRGD_FAIL("conv_igemm: GroupNorm statistics need Hout*Wout % 128 == 0");
    if (a.Cout % a.groups != 0 || (a.Cout / a.groups) > BN || BN % (a.Cout / a.groups) != 0)
      SRGD_FAIL("conv_igemm: unsupported channels-per-group for fused GroupNorm statistics");
  }
  if (a.mode == CONV_PIXEL_SHUFFLE_SILU && (a.Cout % 4 != 0 || a.residual || a.gn_partial))
    SRGD_FAIL("conv_igemm: invalid pixel-shuffle epilogue combination");
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
