// Attention kernels for gfx950 (fp32 arithmetic on fp32 or bf16 NHWC tensors).
//
// linear attention (reference model.py:312-323): q softmax over the 32 per-head channels,
//   k softmax over ALL positions n (up to 65,536), ctx[d,e] = sum_n k[d,n] v[e,n],
//   out[e,n] = sum_d ctx[d,e] q[d,n] * dh^-0.5.
//   The position softmax is computed chunk-locally (max, sum, unnormalised ctx per 512
//   positions) and merged by a fixed-order combine - no atomics, deterministic.
// full attention (model.py:344-355 + denoising_diffusion_pytorch Attend, flash=False):
//   softmax(q k^T dh^-0.5) v with n = 1024, flash-style (online softmax, 64-key tiles in LDS).
#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int DH = 32;              // dim_head of every attention site in this model family

// Context partials on the matrix cores.  One wave per (chunk, head): ctx[d][e] += sum_n p[n][d] v[n][e] is a
// 32x32 tile with K = positions, fed to v_mfma_f32_32x32x2_f32 two positions at a time: lane (r, h) supplies
// p[n+h][d=r] as the A operand and v[n+h][e=r] as the B operand, so every load is a coalesced 32-channel row
// segment and nothing is staged through LDS.  Products are exact fp32 (p stays fp32 even in bf16 mode).
// Pass 1 takes the chunk-local max of k (the second read of k comes from L2).
template <typename T>
__global__ __launch_bounds__(256) void la_partial_kernel(const T* __restrict__ qkv, int N, int heads, int chunk_len,
                                                          float* __restrict__ pm, float* __restrict__ pl,
                                                          float* __restrict__ pctx) {
  const int chunk = blockIdx.x, b = blockIdx.z;
  const int nch = gridDim.x;
  const int head = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (head >= heads) return;                                  // wave-uniform
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int hid = heads * DH, C3 = 3 * hid;
  const int n0 = chunk * chunk_len;
  const int cnt = min(chunk_len, N - n0);
  const T* kp = qkv + ((size_t)b * N + n0) * C3 + hid + head * DH + r;
  const T* vp = kp + hid;

  float m = -INFINITY;
  for (int n = h; n < cnt; n += 2) m = fmaxf(m, to_f32<T>(kp[(size_t)n * C3]));
  m = fmaxf(m, __shfl_xor(m, 32, 64));

  f32x16 acc = 0;
  float l = 0.f;
  for (int n = 0; n < cnt; n += 2) {
    const bool ok = n + h < cnt;
    const size_t o = (size_t)(ok ? n + h : 0) * C3;
    const float kk = to_f32<T>(kp[o]);
    const float vv = to_f32<T>(vp[o]);
    const float pe = ok ? expf(kk - m) : 0.f;
    l += pe;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pe, ok ? vv : 0.f, acc, 0, 0, 0);
  }
  l += __shfl_xor(l, 32, 64);
  const size_t pidx = ((size_t)(b * heads + head) * nch + chunk);
  if (h == 0) {
    pm[pidx * DH + r] = m;
    pl[pidx * DH + r] = l;
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int d = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    pctx[(pidx * DH + d) * DH + r] = acc[reg];
  }
}

// one block per (b, head): merge chunk partials -> normalised, pre-scaled context [d][e]
__global__ __launch_bounds__(256) void la_combine_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                          const float* __restrict__ pctx, int nch, float scale,
                                                          float* __restrict__ ctxn) {
  const int bh = blockIdx.x;
  const int tid = threadIdx.x, d = tid >> 3, eg = tid & 7;
  const float* m_ = pm + (size_t)bh * nch * DH;
  const float* l_ = pl + (size_t)bh * nch * DH;
  const float* c_ = pctx + (size_t)bh * nch * DH * DH;
  float m = -INFINITY;
  for (int c = 0; c < nch; ++c) m = fmaxf(m, m_[c * DH + d]);
  float l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; ++c) {
    const float w = expf(m_[c * DH + d] - m);
    l += l_[c * DH + d] * w;
    const f32x4 v = *reinterpret_cast<const f32x4*>(c_ + ((size_t)c * DH + d) * DH + eg * 4);
    acc[0] += v[0] * w;
    acc[1] += v[1] * w;
    acc[2] += v[2] * w;
    acc[3] += v[3] * w;
  }
  const float inv = scale / l;
  float* o = ctxn + ((size_t)bh * DH + d) * DH + eg * 4;
  o[0] = acc[0] * inv;
  o[1] = acc[1] * inv;
  o[2] = acc[2] * inv;
  o[3] = acc[3] * inv;
}

// one thread per (position, head): softmax over the head's 32 q channels, then a 32x32 matvec
template <typename T>
__global__ __launch_bounds__(256) void la_apply_kernel(const T* __restrict__ qkv, const float* __restrict__ ctxn,
                                                        T* __restrict__ out, int N, int heads) {
  extern __shared__ __attribute__((aligned(16))) float sctx[];   // [heads][DH*DH + 4]
  const int b = blockIdx.y;
  const int hid = heads * DH, C3 = 3 * hid;
  const int HS = DH * DH + 4;
  for (int i = threadIdx.x; i < heads * DH * DH; i += 256) {
    const int h = i / (DH * DH), r = i - h * DH * DH;
    sctx[h * HS + r] = ctxn[((size_t)(b * heads + h)) * DH * DH + r];
  }
  __syncthreads();
  const int ppb = 256 / heads;
  const int head = threadIdx.x % heads;
  const int n = blockIdx.x * ppb + threadIdx.x / heads;
  if (n >= N) return;
  const T* qp = qkv + ((size_t)b * N + n) * C3 + head * DH;
  float q[DH];
  constexpr int VN = Vec16<T>::N;
#pragma unroll
  for (int v = 0; v < DH / VN; ++v) {
    Vec16<T> t = reinterpret_cast<const Vec16<T>*>(qp)[v];
#pragma unroll
    for (int j = 0; j < VN; ++j) q[v * VN + j] = t.get(j);
  }
  float mx = q[0];
#pragma unroll
  for (int i = 1; i < DH; ++i) mx = fmaxf(mx, q[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < DH; ++i) {
    q[i] = expf(q[i] - mx);
    sum += q[i];
  }
  const float inv = 1.0f / sum;
  float o[DH];
#pragma unroll
  for (int e = 0; e < DH; ++e) o[e] = 0.f;
  const float* cx = sctx + head * HS;
#pragma unroll 4
  for (int dd = 0; dd < DH; ++dd) {
    const float qs = q[dd] * inv;
#pragma unroll
    for (int e4 = 0; e4 < DH / 4; ++e4) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cx + dd * DH + e4 * 4);
      o[e4 * 4 + 0] += qs * c4[0];
      o[e4 * 4 + 1] += qs * c4[1];
      o[e4 * 4 + 2] += qs * c4[2];
      o[e4 * 4 + 3] += qs * c4[3];
    }
  }
  T* op = out + ((size_t)b * N + n) * hid + head * DH;
#pragma unroll
  for (int v = 0; v < DH / VN; ++v) {
    Vec16<T> t;
#pragma unroll
    for (int j = 0; j < VN; ++j) t.set(j, o[v * VN + j]);
    reinterpret_cast<Vec16<T>*>(op)[v] = t;
  }
}

// flash-style softmax attention: one thread per query row, 64-key tiles of K and V in LDS
constexpr int FA_TILE = 64;
template <typename T>
__global__ __launch_bounds__(256) void full_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N,
                                                         int heads, float scale) {
  __shared__ __attribute__((aligned(16))) float ks[FA_TILE][DH];
  __shared__ __attribute__((aligned(16))) float vs[FA_TILE][DH];
  const int head = blockIdx.y, b = blockIdx.z;
  const int hid = heads * DH, C3 = 3 * hid;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool active = qi < N;
  const T* base = qkv + (size_t)b * N * C3;
  constexpr int VN = Vec16<T>::N;
  float q[DH], o[DH];
  if (active) {
    const T* qp = base + (size_t)qi * C3 + head * DH;
#pragma unroll
    for (int v = 0; v < DH / VN; ++v) {
      Vec16<T> t = reinterpret_cast<const Vec16<T>*>(qp)[v];
#pragma unroll
      for (int j = 0; j < VN; ++j) q[v * VN + j] = t.get(j);
    }
  } else {
#pragma unroll
    for (int i = 0; i < DH; ++i) q[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < DH; ++i) o[i] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < N; k0 += FA_TILE) {
    const int tc = min(FA_TILE, N - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < FA_TILE * DH; i += 256) {
      const int n = i >> 5, c = i & 31;
      float kv = 0.f, vv = 0.f;
      if (n < tc) {
        const size_t off = (size_t)(k0 + n) * C3 + head * DH + c;
        kv = to_f32<T>(base[off + hid]);
        vv = to_f32<T>(base[off + 2 * hid]);
      }
      ks[n][c] = kv;
      vs[n][c] = vv;
    }
    __syncthreads();
    for (int j0 = 0; j0 < tc; j0 += 16) {
      float s[16];
      float bm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a = 0.f;
#pragma unroll
        for (int e4 = 0; e4 < DH / 4; ++e4) {
          const f32x4 k4 = *reinterpret_cast<const f32x4*>(&ks[j0 + j][e4 * 4]);
          a += q[e4 * 4 + 0] * k4[0] + q[e4 * 4 + 1] * k4[1] + q[e4 * 4 + 2] * k4[2] + q[e4 * 4 + 3] * k4[3];
        }
        a *= scale;
        if (j0 + j >= tc) a = -INFINITY;
        s[j] = a;
        bm = fmaxf(bm, a);
      }
      const float mn = fmaxf(m, bm);
      const float corr = expf(m - mn);      // m = -inf on the first block -> 0
      l *= corr;
#pragma unroll
      for (int e = 0; e < DH; ++e) o[e] *= corr;
      m = mn;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = expf(s[j] - m);
        l += p;
#pragma unroll
        for (int e4 = 0; e4 < DH / 4; ++e4) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(&vs[j0 + j][e4 * 4]);
          o[e4 * 4 + 0] += p * v4[0];
          o[e4 * 4 + 1] += p * v4[1];
          o[e4 * 4 + 2] += p * v4[2];
          o[e4 * 4 + 3] += p * v4[3];
        }
      }
    }
  }
  if (!active) return;
  const float inv = 1.0f / l;
  T* op = out + ((size_t)b * N + qi) * hid + head * DH;
#pragma unroll
  for (int v = 0; v < DH / VN; ++v) {
    Vec16<T> t;
#pragma unroll
    for (int j = 0; j < VN; ++j) t.set(j, o[v * VN + j] * inv);
    reinterpret_cast<Vec16<T>*>(op)[v] = t;
  }
}

// Softmax attention on the matrix cores (exact fp32 MFMA, so the same kernel serves both precision modes).
// One wave = 32 query rows of one (sample, head); queries sit on the lanes, keys in the registers:
//   S^T  = K . Q^T        A: K rows (lane r = key, k = d pair), B: Q^T (lane r = query)         16 MFMAs / 32 keys
//   softmax over keys     = per-lane reduction over 16 registers x 2 lane halves (online max / sum)
//   O^T += V^T . P^T      B: P^T straight from the accumulator registers (register t holds keys kappa_t and
//                            kappa_t + 4 of the two lane halves - exactly one k pair of v_mfma_f32_32x32x2_f32),
//                         A: V[key][e] loaded as coalesced 32-channel row segments                 16 MFMAs / 32 keys
// No LDS in the main loop; the output tile is transposed through LDS once for row-contiguous stores.
template <typename T>
__global__ __launch_bounds__(256) void full_attn_mfma_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N,
                                                              int heads, float scale) {
  __shared__ __attribute__((aligned(16))) float so[4][32][36];
  const int head = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int hid = heads * DH, C3 = 3 * hid;
  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= N) return;                                          // wave-uniform (N % 32 == 0)
  const T* base = qkv + (size_t)b * N * C3 + head * DH;
  constexpr int VN = Vec16<T>::N;

  // element 2 t + hh of a row held in registers, as a bit select: written `hh ? row[2t+1] : row[2t]` the compiler turns the pair
  // into ONE dynamically indexed read row[2t + hh] and moves the row into scratch memory (144 B per lane inside the key loop)
  const unsigned hh_mask = 0u - (unsigned)hh;
  auto sel_hh = [&](float even, float odd) {
    const unsigned a = __float_as_uint(even), o = __float_as_uint(odd);
    return __uint_as_float(a ^ ((a ^ o) & hh_mask));
  };
  // this lane's share of its query row: Q[q0 + r][2t + hh] * scale, t = 0..15
  float qv[16];
  {
    const T* qp = base + (size_t)(q0 + r) * C3;
    float row[DH];
#pragma unroll
    for (int v = 0; v < DH / VN; ++v) {
      Vec16<T> t = reinterpret_cast<const Vec16<T>*>(qp)[v];
#pragma unroll
      for (int j = 0; j < VN; ++j) row[v * VN + j] = t.get(j);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) qv[t] = sel_hh(row[2 * t], row[2 * t + 1]) * scale;
  }
  f32x16 o = 0;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < N; k0 += 32) {
    float kk[16];
    {
      const T* kp = base + (size_t)(k0 + r) * C3 + hid;
      float row[DH];
#pragma unroll
      for (int v = 0; v < DH / VN; ++v) {
        Vec16<T> t = reinterpret_cast<const Vec16<T>*>(kp)[v];
#pragma unroll
        for (int j = 0; j < VN; ++j) row[v * VN + j] = t.get(j);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) kk[t] = sel_hh(row[2 * t], row[2 * t + 1]);
    }
    // V[k0 + kappa_t + 4 hh][e = r] for the second product, issued early
    float vv[16];
    {
      const T* vp = base + (size_t)k0 * C3 + 2 * hid + r;
#pragma unroll
      for (int t = 0; t < 16; ++t) vv[t] = to_f32<T>(vp[(size_t)((t & 3) + 8 * (t >> 2) + 4 * hh) * C3]);
    }
    f32x16 st = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[t], qv[t], st, 0, 0, 0);
    float bm = st[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) bm = fmaxf(bm, st[i]);
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m, bm);
    const float corr = expf(m - mn);
    m = mn;
    l *= corr;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      st[i] = expf(st[i] - mn);
      l += st[i];
      o[i] *= corr;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[t], st[t], o, 0, 0, 0);
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  // o: rows e = (reg&3) + 8 (reg>>2) + 4 hh, column = query r  ->  LDS [query][e]  ->  row-contiguous stores
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(&so[wave][r][8 * g + 4 * hh]) = f32x4{o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
  __builtin_amdgcn_s_waitcnt(0xC07F);                           // lgkmcnt(0): this wave's LDS writes are done
  __builtin_amdgcn_wave_barrier();
  {
    // 32 queries x 32 channels per wave: lane -> (query = lane / 2 + 0/16 ..., 16 channels)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int qq = it * 16 + (lane >> 2), c0 = (lane & 3) * 8;
      T* op = out + ((size_t)b * N + q0 + qq) * hid + head * DH + c0;
      Vec16<T> w0;
      if constexpr (VN == 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w0.set(j, so[wave][qq][c0 + j]);
        *reinterpret_cast<Vec16<T>*>(op) = w0;
      } else {
        Vec16<T> w1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          w0.set(j, so[wave][qq][c0 + j]);
          w1.set(j, so[wave][qq][c0 + 4 + j]);
        }
        reinterpret_cast<Vec16<T>*>(op)[0] = w0;
        reinterpret_cast<Vec16<T>*>(op)[1] = w1;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// bf16 softmax attention for the production shape (N = 1024 positions, 4 heads x 32; reference Attention.forward
// model.py:342-355 + Attend): K and V^T of one (sample, head) live in LDS (64 KB + 64.5 KB), every wave owns 32 queries.
//   S^T = K . Q^T         v_mfma_f32_16x16x32_bf16, one instruction per 16 keys x 16 queries (K dim = dim_head = 32)
//   online softmax        per query = per lane column: max over keys via two xor-shuffles, exp2 in fp32
//   O^T += V^T . P^T      P^T goes from the accumulator registers of two S^T tiles straight into the B operand
//                         (lane (query, g) holds keys {4g..4g+3} u {16+4g..16+4g+3} of the 32-key block); V^T is stored
//                         in LDS with the keys of each block permuted the same way, so its A operand is one ds_read_b128.
constexpr int FA_QB = 256;                 // queries per workgroup (8 waves x 32)
constexpr int FA_NT = 512;
__device__ __forceinline__ int fa_vpos(int key) {      // position of `key` inside V^T's permuted 32-key blocks
  return (key & ~31) + (((key & 15) >> 2) << 3) + (((key >> 4) & 1) << 2) + (key & 3);
}
__global__ __launch_bounds__(FA_NT) void full_attn_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int N,
                                                               int heads, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char fa_smem[];
  const int head = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int hid = heads * DH, C3 = 3 * hid;
  const int vstride = N * 2 + 16;                       // bytes per V^T row (+16: ds_read_b128 of 16 rows conflict-free)
  char* const Ks = fa_smem;                             // [N][32] bf16, 64-byte rows
  char* const Vt = fa_smem + (size_t)N * 64;            // [32][N (permuted)] bf16
  const bf16* base = qkv + (size_t)b * N * C3 + head * DH;
  // ---- stage K (16-byte copies) and V^T (transposing 2-byte writes)
  for (int c = tid; c < N * 4; c += FA_NT) {
    const int key = c >> 2, ch = c & 3;
    const bf16x8 kv = *reinterpret_cast<const bf16x8*>(base + (size_t)key * C3 + hid + ch * 8);
    *reinterpret_cast<bf16x8*>(Ks + key * 64 + ch * 16) = kv;
    const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + (size_t)key * C3 + 2 * hid + ch * 8);
    const int pos = fa_vpos(key);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<bf16*>(Vt + (ch * 8 + j) * vstride + pos * 2) = vv[j];
  }
  // ---- this wave's queries: B operand of S^T, lane (query r16, channel chunk g)
  const int q0 = blockIdx.x * FA_QB + wave * 32;
  const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(base + (size_t)(q0 + r16) * C3 + g * 8);
  const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(base + (size_t)(q0 + 16 + r16) * C3 + g * 8);
  __syncthreads();

  f32x4 o00 = 0, o01 = 0, o10 = 0, o11 = 0;             // [query block][channel block]
  float m0 = -INFINITY, m1 = -INFINITY, l0 = 0.f, l1 = 0.f;
  const char* kp = Ks + r16 * 64 + g * 16;
  const char* vp = Vt + r16 * vstride + g * 16;
  for (int kb = 0; kb < N; kb += 32) {
    const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kp + kb * 64);
    const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kp + (kb + 16) * 64);
    const bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(vp + kb * 2);
    const bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(vp + 16 * vstride + kb * 2);
    const f32x4 z = 0;
    f32x4 sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf0, z, 0, 0, 0);
    f32x4 sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf0, z, 0, 0, 0);
    f32x4 sb0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf1, z, 0, 0, 0);
    f32x4 sb1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf1, z, 0, 0, 0);
#define K_FA_SOFTMAX(S0, S1, M, L, OA, OB, PB)                                              \
    {                                                                                          \
      S0 *= scale_log2e;                                                                       \
      S1 *= scale_log2e;                                                                       \
      float bm = fmaxf(fmaxf(fmaxf(S0[0], S0[1]), fmaxf(S0[2], S0[3])),                        \
                       fmaxf(fmaxf(S1[0], S1[1]), fmaxf(S1[2], S1[3])));                       \
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));                                                  \
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));                                                  \
      const float mn = fmaxf(M, bm);                                                           \
      const float corr = exp2f(M - mn);                                                        \
      M = mn;                                                                                  \
      float ps = 0.f;                                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                          \
        const float e0 = exp2f(S0[j] - mn), e1 = exp2f(S1[j] - mn);                            \
        ps += e0 + e1;                                                                         \
        PB[j] = (bf16)e0;                                                                      \
        PB[4 + j] = (bf16)e1;                                                                  \
      }                                                                                        \
      L = L * corr + ps;                                                                       \
      OA *= corr;                                                                              \
      OB *= corr;                                                                              \
    }
    bf16x8 pa, pb;
    K_FA_SOFTMAX(sa0, sa1, m0, l0, o00, o01, pa)
    K_FA_SOFTMAX(sb0, sb1, m1, l1, o10, o11, pb)
#undef K_FA_SOFTMAX
    o00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pa, o00, 0, 0, 0);
    o01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, pa, o01, 0, 0, 0);
    o10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pb, o10, 0, 0, 0);
    o11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, pb, o11, 0, 0, 0);
  }
  // each lane holds a partial denominator over its quarter of the keys
  l0 += __shfl_xor(l0, 16, 64);
  l0 += __shfl_xor(l0, 32, 64);
  l1 += __shfl_xor(l1, 16, 64);
  l1 += __shfl_xor(l1, 32, 64);
  const float i0 = 1.0f / l0, i1 = 1.0f / l1;
  // o[qb][cb][reg]: query q0 + qb*16 + r16, channel cb*16 + g*4 + reg  ->  8-byte stores
  bf16* op0 = out + ((size_t)b * N + q0 + r16) * hid + head * DH + g * 4;
  bf16* op1 = op0 + (size_t)16 * hid;
  *reinterpret_cast<bf16x4*>(op0) = bf16x4{(bf16)(o00[0] * i0), (bf16)(o00[1] * i0), (bf16)(o00[2] * i0), (bf16)(o00[3] * i0)};
  *reinterpret_cast<bf16x4*>(op0 + 16) = bf16x4{(bf16)(o01[0] * i0), (bf16)(o01[1] * i0), (bf16)(o01[2] * i0), (bf16)(o01[3] * i0)};
  *reinterpret_cast<bf16x4*>(op1) = bf16x4{(bf16)(o10[0] * i1), (bf16)(o10[1] * i1), (bf16)(o10[2] * i1), (bf16)(o10[3] * i1)};
  *reinterpret_cast<bf16x4*>(op1 + 16) = bf16x4{(bf16)(o11[0] * i1), (bf16)(o11[1] * i1), (bf16)(o11[2] * i1), (bf16)(o11[3] * i1)};
}

}  // namespace

// positions per partial: long chunks amortise the 4.3 KB partial record; short ones keep small maps parallel
static int la_chunk_len(int N) { return N > 16384 ? 1024 : 512; }
static int la_chunks(int N) { return cdiv(N, la_chunk_len(N)); }

size_t linear_attention_workspace(int B, int N, int heads, int dh) {
  const size_t bh = (size_t)B * heads, nch = la_chunks(N);
  return (bh * nch * (2 * dh + dh * dh) + bh * dh * dh) * sizeof(float);
}

int linear_attention(const void* qkv, void* out, int B, int N, int heads, int dh, float* ws, bool is_bf16,
                     hipStream_t st) {
  if (dh != DH) SRGD_FAIL("linear_attention: dim_head must be 32");
  const int nch = la_chunks(N);
  const size_t bh = (size_t)B * heads;
  float* pm = ws;
  float* pl = pm + bh * nch * DH;
  float* pctx = pl + bh * nch * DH;
  float* ctxn = pctx + bh * nch * DH * DH;
  const float scale = 1.0f / sqrtf((float)dh);
  dim3 g1(nch, cdiv(heads, 4), B);
  const int clen = la_chunk_len(N);
  if (is_bf16)
    hipLaunchKernelGGL((la_partial_kernel<bf16>), g1, dim3(256), 0, st, (const bf16*)qkv, N, heads, clen, pm, pl, pctx);
  else
    hipLaunchKernelGGL((la_partial_kernel<float>), g1, dim3(256), 0, st, (const float*)qkv, N, heads, clen, pm, pl, pctx);
  SRGD_HIP(hipGetLastError());
  hipLaunchKernelGGL(la_combine_kernel, dim3((unsigned)bh), dim3(256), 0, st, pm, pl, pctx, nch, scale, ctxn);
  SRGD_HIP(hipGetLastError());
  if (256 % heads != 0) SRGD_FAIL("linear_attention: heads must divide 256");
  const int ppb = 256 / heads;
  dim3 g3(cdiv(N, ppb), B);
  const size_t lds = (size_t)heads * (DH * DH + 4) * sizeof(float);
  if (is_bf16)
    hipLaunchKernelGGL((la_apply_kernel<bf16>), g3, dim3(256), lds, st, (const bf16*)qkv, ctxn, (bf16*)out, N, heads);
  else
    hipLaunchKernelGGL((la_apply_kernel<float>), g3, dim3(256), lds, st, (const float*)qkv, ctxn, (float*)out, N, heads);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int linear_attention_combine(const float* pm, const float* pl, const float* pctx, int bh, int nch, float scale,
                             float* ctxn, hipStream_t st) {
  hipLaunchKernelGGL(la_combine_kernel, dim3((unsigned)bh), dim3(256), 0, st, pm, pl, pctx, nch, scale, ctxn);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int full_attention(const void* qkv, void* out, int B, int N, int heads, int dh, bool is_bf16, hipStream_t st) {
  if (dh != DH) SRGD_FAIL("full_attention: dim_head must be 32");
  const float scale = 1.0f / sqrtf((float)dh);
  if (is_bf16 && N % FA_QB == 0 && N <= 1024) {            // bf16 MFMA kernel, K / V^T resident in LDS (production: N = 1024)
    const int lds = N * 64 + DH * (N * 2 + 16);
    static bool attr_set[64] = {};
    if (DeviceSetup once(attr_set); once.need) {
      SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_attn_bf16_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 64 + DH * (1024 * 2 + 16)));
      once.done();
    }
    hipLaunchKernelGGL(full_attn_bf16_kernel, dim3(N / FA_QB, heads, B), dim3(FA_NT), lds, st, (const bf16*)qkv, (bf16*)out, N,
                       heads, scale * 1.4426950408889634f);
    SRGD_HIP(hipGetLastError());
    return 0;
  }
  if (N % 32 == 0) {                                       // fp32-MFMA kernel (fp32 mode and other N)
    dim3 g(cdiv(N, 128), heads, B);
    if (is_bf16)
      hipLaunchKernelGGL((full_attn_mfma_kernel<bf16>), g, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, N, heads, scale);
    else
      hipLaunchKernelGGL((full_attn_mfma_kernel<float>), g, dim3(256), 0, st, (const float*)qkv, (float*)out, N, heads, scale);
    SRGD_HIP(hipGetLastError());
    return 0;
  }
  dim3 g(cdiv(N, 256), heads, B);
  if (is_bf16)
    hipLaunchKernelGGL((full_attn_kernel<bf16>), g, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, N, heads, scale);
  else
    hipLaunchKernelGGL((full_attn_kernel<float>), g, dim3(256), 0, st, (const float*)qkv, (float*)out, N, heads, scale);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
