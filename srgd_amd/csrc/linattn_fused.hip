// Fused LinearAttention block for gfx950 (bf16 mode, C = 128, 4 heads x 32; C = 256: linattn_fused256.hip): the whole of reference
// model.py:306-324 -
//     RMSNorm -> to_qkv (1x1) -> q softmax over d / k softmax over ALL positions -> context -> out ->
//     to_out (1x1 + bias) -> RMSNorm -> (+ x, model.py:703)
// in two kernels that touch HBM three times per pixel (read x, read x, write y) instead of the ten tensor
// passes of the unfused chain (q, k, v are never materialised):
//
//   la1_kernel  per 64-pixel tile: x -> LDS (LDS-DMA ring, swizzled), row norms, [k|v] = x . Wkv'^T on MFMA with the
//               wave tiling chosen so that ONE wave owns the k-columns and the v-columns of one head for the tile's
//               rows (its slice of Wkv' stays in registers); exp(k - m) and v then feed the context product  ctx[d][e] += sum_n p[n][d] v[n][e]  straight
//               from the accumulator registers (an accumulator tile is a valid MFMA operand for a product that sums
//               over its row index) - no LDS round trip.  Online max/sum per column, one partial per wave.
//   (la_combine_kernel from attention.hip merges the partials: fixed order, deterministic.)
//   la2_kernel  per 64-pixel tile: q^T = Wq' . x^T (pixels on lanes, so the d-softmax is a per-lane register
//               reduction), att^T = ctx^T . q' again from accumulator registers, att -> LDS, o^T = Wout . att^T
//               + bias, RMSNorm over c (cross-wave sum of squares through LDS), * g2, + x (the x tile is still in
//               LDS), staged and stored as whole 256-byte rows.  Wq', Wout and ctx live in registers for the
//               lifetime of the (persistent) workgroup.
// Both kernels run 256-thread workgroups (one wave per head / channel block) with 48-68 KiB of LDS, so 2 workgroups share a
// CU and their MFMA, exponential and store phases overlap; 1/||x|| is computed once (la1) and handed to la2 (4 B / pixel).
// bf16-mode only, so exponentials are bare v_exp_f32 in the log2 domain (1/||x|| * log2(e) rides in the argument).
// The RMSNorm gains g1*sqrt(C) are folded into Wkv'/Wq' on the host.
#include <cstdlib>

#include "kernels.hpp"

namespace srgd {
namespace {

// x-tile DMAs and y stores are non-temporal (x and y are streamed once per pass): kernel share 6.47 -> 6.40 % of a step, same box
// (profiles/r5/nt_policy/r5_nt_la.json)
#ifndef SRGD_LA_STAMPS
#define SRGD_LA_STAMPS 0  // diagnostic build: per-phase s_memtime ticks of la1's tile loop (wave 0 of every workgroup), printed per launch
#endif
constexpr int TM = 64;                  // pixels per tile
constexpr int TILE_BYTES = TM * 256;    // [64 rows][128 bf16] = 16 KiB
constexpr int NTH = 256;                // 4 waves: one per head (la1) / per 32-channel block (la2)
constexpr int RING = 3;                 // x tiles in flight per workgroup (LDS-DMA ring)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if SRGD_LA_STAMPS
__device__ unsigned long long g_la1_stamps[8];   // [norms + sync, k/v GEMM, scale + max + exp, rescale + context MFMAs, DMA wait + barrier, tiles]
#endif

__device__ __forceinline__ int swz(int row, int chunk16) { return row * 256 + ((chunk16 ^ (row & 15)) << 4); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voffset, 0, 0, 2 /* nt */);
}

#define LA_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define LA_BARRIER()                     \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define LA_SYNC()                                         \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    LA_BARRIER();                                         \
  } while (0)

// stage one 64-pixel x tile (rows px0..px0+63 of the image behind rsrc) into `buf`, XOR-swizzled: 4 LDS-DMA pieces per wave
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* buf, int wave, int lane, int px0) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    const int g = q * 64 + lane;
    const int row = g >> 4, cs = g & 15;
    const int c = cs ^ (row & 15);
    dma16(rsrc, buf + q * 1024, ((px0 + row) * 128 + c * 8) * 2);
  }
}

// 1 / max(||x_row||, 1e-12) for the 64 rows of a staged tile (4 threads per row).
// Measured alternatives, none shipped: v_dot2c_f32_bf16 on the packed pairs (round 4: la1 +8 %); the column sums l from a
// p . ones MFMA (+7 %); tied inline-asm MFMAs; the barrier moved behind the GEMM (+4 %); and round 5: the sums of squares taken
// from the GEMM's own operand fragments by v_dot2 beside the MFMAs, which removes this pass, its LDS re-read and its barrier -
// the GEMM phase then takes 3,369 instead of 1,653 ticks per tile (more than the 1,331 it replaces: la1 +4.5 %,
// profiles/r5/la1_norms_from_gemm_fragments_stamps.txt) and the fdot2 sums cost 8 dB of bf16-mode PSNR against the reference.
__device__ __forceinline__ void row_rinv(const char* tile, float* rinv, int tid) {
  const int row = tid >> 2, part = tid & 3;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * 256 + (part * 4 + j) * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
  }
  ss += __shfl_xor(ss, 1, 64);
  ss += __shfl_xor(ss, 2, 64);
  if (part == 0) rinv[row] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
}

// raw v_exp_f32 (arguments here are <= 0 up to rounding: no denormal-range rescue needed)
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
  // four v_cvt_pk_bf16_f32 whose results ARE the operand tuple's dwords (element-wise assembly of the bf16x8 left the packed
  // pairs in scattered registers and copied them together: 40 v_mov per tile)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w;
  w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 0], a[8 * s + 1]}, bf16x2_t));
  w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 2], a[8 * s + 3]}, bf16x2_t));
  w[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 4], a[8 * s + 5]}, bf16x2_t));
  w[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a[8 * s + 6], a[8 * s + 7]}, bf16x2_t));
  return __builtin_bit_cast(bf16x8, w);
}

// four floats -> four bf16 through two packed converts (element-wise assembly leaves the halves in scattered registers)
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 w;
  w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2_t));
  w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{c, d}, bf16x2_t));
  return __builtin_bit_cast(bf16x4, w);
}

// ------------------------------------------------------------------------------------------- phase 1
// Workgroup = 4 waves = the 4 heads; each wave owns the k-columns and the v-columns of its head for all 64 rows of a tile.
// Its slice of Wkv' (64 rows x 128) lives in registers as MFMA B fragments for the lifetime of the workgroup, so LDS holds
// only the x ring (48 KiB) and several workgroups share a CU: while one is in its exponentials another runs its MFMAs
// (the 8-wave version kept Wkv' in LDS, fitted once per CU and ran its phases in lock-step at 36 % of the HBM rate).
__global__ __launch_bounds__(NTH, 2) void la1_kernel(const bf16* __restrict__ x, int N, const bf16* __restrict__ wkv,
                                                      int strip, float* __restrict__ pm, float* __restrict__ pl,
                                                      float* __restrict__ pctx, float* __restrict__ rinv_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;                                                   // RING x tiles
  float* const sR = reinterpret_cast<float*>(smem + RING * TILE_BYTES);    // [2][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int head = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, sidx = blockIdx.x, nstrips = gridDim.x;
  const int px_begin = sidx * strip;
  const int T = min(strip, N - px_begin) / TM;

  const __amdgpu_buffer_rsrc_t rsx =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * N * 128), 0, N * 256, 0x00020000);
  // B fragments of the head's k rows and v rows of Wkv' (host image: [256 rows][128 c], 16-byte chunks XOR-swizzled by row & 15)
  bf16x8 fk[8], fv[8];
  {
    const int wk = head * 32 + r, wv = 128 + head * 32 + r;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 2 * s + hh;
      fk[s] = *reinterpret_cast<const bf16x8*>(wkv + wk * 128 + ((c ^ (wk & 15)) << 3));
      fv[s] = *reinterpret_cast<const bf16x8*>(wkv + wv * 128 + ((c ^ (wv & 15)) << 3));
    }
  }
  stage_tile(rsx, sA, head, lane, px_begin);
  if (T > 1) stage_tile(rsx, sA + TILE_BYTES, head, lane, px_begin + TM);
  float m = -INFINITY;                              // running max (log2 domain) of this lane's k column d = r
  f32x16 ctx = 0;
  float lvec = 0.f;                                 // running column sum l[d = r] (this lane's half of the rows)
#define LA_MM(C_, A_, B_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
#define LA_MM0(C_, A_, B_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, f32x16(0), 0, 0, 0)
  // tile 0 has landed once at most the second tile's 4 pieces are outstanding (wherever the compiler put the fragment
  // loads relative to the DMAs, "all but the 4 youngest" covers tile 0)
  if (T > 1) LA_WAIT_VM(4); else LA_WAIT_VM(0);
  LA_BARRIER();

#if SRGD_LA_STAMPS
  unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, tq = __builtin_amdgcn_s_memtime(), tn;
#define LA_STAMP(ACC_) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tn = __builtin_amdgcn_s_memtime(); ACC_ += tn - tq; tq = tn; } while (0)
#else
#define LA_STAMP(ACC_) do {} while (0)
#endif
  for (int t = 0; t < T; ++t) {
    const char* A = sA + (t % RING) * TILE_BYTES;
    float* rinv = sR + (t & 1) * TM;
    if (t + 2 < T) stage_tile(rsx, sA + ((t + 2) % RING) * TILE_BYTES, head, lane, px_begin + (t + 2) * TM);
    row_rinv(A, rinv, tid);
    LA_SYNC();
    LA_STAMP(ph0);
    if (head == 0 && lane < 16)                     // la2 re-uses the row norms: 4 B per pixel instead of a second reduction
      *reinterpret_cast<f32x4*>(rinv_out + (size_t)b * N + px_begin + t * TM + lane * 4) = *reinterpret_cast<const f32x4*>(rinv + lane * 4);

    // [k | v] of this wave's head for the tile's 64 rows.  Fragment addresses: swz(row, 2 s + hh) = swz(row, hh) ^ (s << 5) (2 s and
    // hh ^ (row & 15) occupy different bits of the chunk index), and rows r and 32 + r swizzle alike: ONE base register, made
    // opaque per tile so that the sixteen addresses are recomputed (one v_xor each) instead of hoisted out of the tile loop into
    // sixteen long-lived registers - the kernel sat at the 256-register ceiling with two spilled (round-4 review).
    int fa_base = swz(r, hh);
    asm volatile("" : "+v"(fa_base));
    f32x16 k0, k1, v0, v1;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(A + (fa_base ^ (s << 5)));
      const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(A + (fa_base ^ (s << 5)) + 32 * 256);
      if (s == 0) {                                 // C = 0 as the inline constant: no 64-register zero fill per tile
        LA_MM0(k0, fa0, fk[s]);
        LA_MM0(k1, fa1, fk[s]);
        LA_MM0(v0, fa0, fv[s]);
        LA_MM0(v1, fa1, fv[s]);
      } else {
        LA_MM(k0, fa0, fk[s]);
        LA_MM(k1, fa1, fk[s]);
        LA_MM(v0, fa0, fv[s]);
        LA_MM(v1, fa1, fv[s]);
      }
    }
    LA_STAMP(ph1);
    // rows of the accumulator = pixels: k -> k / ||x_n|| in the log2 domain (one multiply), v -> v / ||x_n||
    float bm = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 ri0 = *reinterpret_cast<const f32x4*>(rinv + 8 * g + 4 * hh);
      const f32x4 ri1 = *reinterpret_cast<const f32x4*>(rinv + 32 + 8 * g + 4 * hh);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int reg = 4 * g + i;
        k0[reg] *= ri0[i] * LOG2E;
        k1[reg] *= ri1[i] * LOG2E;
        v0[reg] *= ri0[i];
        v1[reg] *= ri1[i];
        bm = fmaxf(bm, fmaxf(k0[reg], k1[reg]));
      }
    }
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    // lazy running maximum: the reference point of a column only moves when the tile's maximum exceeds it by more than 2^8 (or at
    // the first tile); until then p = exp2(k - m) may be as large as 256 - harmless in fp32 / bf16 - and the rescale of the
    // context (16 cross-lane shuffles + 32 multiplies) almost never runs.  pm carries the reference point actually used, so the
    // combine step is unchanged; results differ from the eager form by bf16 rounding of p only.
    const float mn = (bm > m + 8.0f) ? bm : m;
    const float f = ex2(m - mn);                  // first tile: exp2(-inf) = 0
    m = mn;
    lvec *= f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      k0[i] = ex2(k0[i] - mn);
      k1[i] = ex2(k1[i] - mn);
      lvec += k0[i] + k1[i];
    }
    LA_STAMP(ph2);
    if (!__all(f == 1.0f)) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int d = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        const float fd = __shfl(f, d, 64);        // lane d (< 32) holds the factor of context row d
        ctx[reg] *= fd;
      }
    }
    // ctx[d][e] += sum_n p[n][d] v[n][e]: both operands come from accumulator tiles (same row permutation)
    {
      const bf16x8 p00 = pack8(k0, 0), p01 = pack8(k0, 1), p10 = pack8(k1, 0), p11 = pack8(k1, 1);
      const bf16x8 q00 = pack8(v0, 0), q01 = pack8(v0, 1), q10 = pack8(v1, 0), q11 = pack8(v1, 1);
      LA_MM(ctx, p00, q00);
      LA_MM(ctx, p10, q10);
      LA_MM(ctx, p01, q01);
      LA_MM(ctx, p11, q11);
    }
#if SRGD_LA_STAMPS
    asm volatile("s_nop 0" : "+v"(ctx));
#endif
    LA_STAMP(ph3);
    // tile t+1 must have landed before the next iteration reads it; the DMA of tile t+2 (4 pieces) stays in flight
    if (t + 2 < T) LA_WAIT_VM(4); else LA_WAIT_VM(0);
    LA_BARRIER();
    LA_STAMP(ph4);
  }
#if SRGD_LA_STAMPS
  if (tid == 0) {
    atomicAdd(&g_la1_stamps[0], ph0); atomicAdd(&g_la1_stamps[1], ph1); atomicAdd(&g_la1_stamps[2], ph2);
    atomicAdd(&g_la1_stamps[3], ph3); atomicAdd(&g_la1_stamps[4], ph4); atomicAdd(&g_la1_stamps[5], (unsigned long long)T);
  }
#endif
#undef LA_STAMP
#undef LA_MM
#undef LA_MM0
  const size_t pidx = (size_t)(b * 4 + head) * nstrips + sidx;
  if (hh == 0) pm[pidx * 32 + r] = m * LN2;         // la_combine works in the natural-log domain
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int d = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
    pctx[(pidx * 32 + d) * 32 + r] = ctx[reg];
  }
  lvec += __shfl_xor(lvec, 32, 64);
  if (hh == 0) pl[pidx * 32 + r] = lvec;
}

// ------------------------------------------------------------------------------------------- phase 2
struct La2Args {
  const bf16* x; bf16* y; int N;
  const bf16* wq;        // [128 d][128 c] bf16, row-major, gains folded
  const bf16* wout;      // [128 c][128 k] bf16, row-major
  const float* bout;     // [128]
  const float* g2;       // [128] = to_out.1.g * sqrt(C)
  const float* ctxn;     // [B*4][32 d][32 e] fp32: normalised context * dh^-0.5
  const float* rinv;     // [B][N] 1/||x_n|| written by la1
  unsigned char* yq; unsigned char* ys;   // optional MX-fp8 twin of y (fp8 mode: y feeds a 3x3 convolution)
  int tiles_per_wg;
};

__global__ __launch_bounds__(NTH, 2) void la2_kernel(La2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;                                   // RING x tiles
  char* const sT = smem + RING * TILE_BYTES;               // att tile, later the staged output tile
  float* const sS = reinterpret_cast<float*>(smem + (RING + 1) * TILE_BYTES);   // [4][64] partial sum of squares of o
  float* const sRv = sS + 4 * TM;                          // [RING][4 waves][64] 1/||x_n|| of the staged tiles (one private copy per wave)
  const int tid = threadIdx.x, lane = tid & 63;
  const int hd = __builtin_amdgcn_readfirstlane(tid >> 6);  // head (GEMM-q) / output channel block (GEMM-out)
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const int tile0 = blockIdx.x * p.tiles_per_wg;
  const int ntiles = p.N / TM;
  const int T = min(p.tiles_per_wg, ntiles - tile0);
  if (T <= 0) return;

  const __amdgpu_buffer_rsrc_t rsx =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.N * 128), 0, p.N * 256, 0x00020000);
  // la1 left 1/||x_n|| in the workspace: it rides along with the tile as a fifth LDS-DMA piece (256 B), so the loop has no
  // VGPR-destination loads (hipcc would wait vmcnt(0) for those and drain the tile prefetch with them)
  const __amdgpu_buffer_rsrc_t rsr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.rinv + (size_t)b * p.N), 0, p.N * 4, 0x00020000);
  auto stage = [&](int slot, int px0) {
    stage_tile(rsx, sA + slot * TILE_BYTES, hd, lane, px0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, (lds_ptr_t)(sRv + (slot * 4 + hd) * TM), 4, (px0 + lane) * 4, 0, 0, 0);
  };

  // register-resident operands: rows (hd*32 + r) of Wq' and Wout as MFMA A fragments for the 8 k16 steps
  bf16x8 wq[8], wo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    wq[s] = *reinterpret_cast<const bf16x8*>(p.wq + (size_t)(hd * 32 + r) * 128 + (2 * s + hh) * 8);
    wo[s] = *reinterpret_cast<const bf16x8*>(p.wout + (size_t)(hd * 32 + r) * 128 + (2 * s + hh) * 8);
  }
  // ctx^T of head hd as the A operand of att^T = ctx^T . q': lane (r = e, hh) element j of k-step s must be
  // ctx[d][e] with d in the accumulator's row order: d = 16 s + 8 (j >> 2) + 4 hh + (j & 3)
  bf16x8 cx[2];
  {
    const float* c = p.ctxn + (size_t)(b * 4 + hd) * 1024;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) cx[s][j] = (bf16)c[(16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)) * 32 + r];
  }
  float bo[16], g2v[16];                                   // bias / gain of output rows c = hd*32 + row(reg, hh)
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int c = hd * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
    bo[reg] = p.bout[c];
    g2v[reg] = p.g2[c];
  }
  stage(0, tile0 * TM);
  if (T > 1) stage(1, (tile0 + 1) * TM);
  if (T > 1) LA_WAIT_VM(5); else LA_WAIT_VM(0);            // tile 0 (and every operand load) landed; tile 1's 5 pieces may fly
  LA_BARRIER();

  for (int t = 0; t < T; ++t) {
    const char* A = sA + (t % RING) * TILE_BYTES;
    const int px0 = (tile0 + t) * TM;
    if (t + 2 < T) stage((t + 2) % RING, px0 + 2 * TM);
    const float* rvt = sRv + ((t % RING) * 4 + hd) * TM;
    const float ri0 = rvt[r] * LOG2E, ri1 = rvt[32 + r] * LOG2E;       // this lane's two pixel columns, log2 domain

    // q^T (rows d of head hd, columns = the tile's 64 pixels)
    f32x16 q0 = 0, q1 = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 2 * s + hh;
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(A + swz(r, c));
      const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(A + swz(32 + r, c));
      q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s], x0, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s], x1, q1, 0, 0, 0);
    }
    // softmax over d: registers (16) x lane halves (2) of one pixel column; 1/||x|| rides in the exp2 argument
    {
      float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        m0 = fmaxf(m0, q0[i]);
        m1 = fmaxf(m1, q1[i]);
      }
      m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
      m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
      const float c0 = -m0 * ri0, c1 = -m1 * ri1;          // ri > 0: the max commutes with the scaling
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        q0[i] = ex2(fmaf(q0[i], ri0, c0));
        q1[i] = ex2(fmaf(q1[i], ri1, c1));
        s0 += q0[i];
        s1 += q1[i];
      }
      s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float i0 = __builtin_amdgcn_rcpf(s0), i1 = __builtin_amdgcn_rcpf(s1);     // (1 ulp; __frcp_rn expands to the IEEE division sequence)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        q0[i] *= i0;
        q1[i] *= i1;
      }
    }
    // att^T[e][n] = sum_d ctx[d][e] q'[d][n]  (B operand straight from the q' accumulator registers)
    f32x16 a0 = 0, a1 = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cx[s], pack8(q0, s), a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cx[s], pack8(q1, s), a1, 0, 0, 0);
    }
    // att -> LDS as [pixel][k = head*32 + e] (swizzled rows): 4 consecutive e = 8 bytes per register quad
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 w0 = pack4(a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]);
      const bf16x4 w1 = pack4(a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]);
      const int k = hd * 32 + 8 * g + 4 * hh;              // first of the 4 channels
      *reinterpret_cast<bf16x4*>(sT + swz(r, k >> 3) + (k & 7) * 2) = w0;
      *reinterpret_cast<bf16x4*>(sT + swz(32 + r, k >> 3) + (k & 7) * 2) = w1;
    }
    LA_SYNC();

    // o^T (rows c of block hd, columns = the tile's 64 pixels) = Wout . att^T + bias
    f32x16 o0 = 0, o1 = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 2 * s + hh;
      const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(sT + swz(r, c));
      const bf16x8 t1 = *reinterpret_cast<const bf16x8*>(sT + swz(32 + r, c));
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[s], t0, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[s], t1, o1, 0, 0, 0);
    }
    float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      o0[i] += bo[i];
      o1[i] += bo[i];
      ss0 += o0[i] * o0[i];
      ss1 += o1[i] * o1[i];
    }
    ss0 += __shfl_xor(ss0, 32, 64);
    ss1 += __shfl_xor(ss1, 32, 64);
    if (hh == 0) {
      sS[hd * TM + r] = ss0;
      sS[hd * TM + 32 + r] = ss1;
    }
    LA_SYNC();                                             // also: every wave is done reading the att tile
    {
      const float n0 = sS[r] + sS[TM + r] + sS[2 * TM + r] + sS[3 * TM + r];
      const float n1 = sS[32 + r] + sS[TM + 32 + r] + sS[2 * TM + 32 + r] + sS[3 * TM + 32 + r];
      const float r0 = 1.0f / fmaxf(sqrtf(n0), 1e-12f), r1 = 1.0f / fmaxf(sqrtf(n1), 1e-12f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x4 w0 = pack4(o0[4 * g] * r0 * g2v[4 * g], o0[4 * g + 1] * r0 * g2v[4 * g + 1], o0[4 * g + 2] * r0 * g2v[4 * g + 2],
                                o0[4 * g + 3] * r0 * g2v[4 * g + 3]);
        const bf16x4 w1 = pack4(o1[4 * g] * r1 * g2v[4 * g], o1[4 * g + 1] * r1 * g2v[4 * g + 1], o1[4 * g + 2] * r1 * g2v[4 * g + 2],
                                o1[4 * g + 3] * r1 * g2v[4 * g + 3]);
        const int c = hd * 32 + 8 * g + 4 * hh;
        *reinterpret_cast<bf16x4*>(sT + swz(r, c >> 3) + (c & 7) * 2) = w0;
        *reinterpret_cast<bf16x4*>(sT + swz(32 + r, c >> 3) + (c & 7) * 2) = w1;
      }
    }
    LA_SYNC();
    // y = staged RMSNorm(o) * g2 + x, whole 256-byte rows, 16 B per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + NTH * i;
      const int row = q >> 4, c16 = q & 15;
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(sT + swz(row, c16));
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(A + swz(row, c16));
      bf16x8 yv;
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = (bf16)((float)ov[e] + (float)xv[e]);
      const size_t yo = ((size_t)b * p.N + px0 + row) * 128 + c16 * 8;
      __builtin_nontemporal_store(yv, reinterpret_cast<bf16x8*>(p.y + yo));
      if (p.yq) mx_store_twin(yv, p.yq, p.ys, yo, tid & 3);
    }
    // Tile t+1 (issued one iteration ago) must have landed before the next iteration reads it; this iteration's DMA of
    // tile t+2 (5 pieces) and its 4 stores (younger still) stay in flight.
    if (t + 2 < T) LA_WAIT_VM(9); else LA_WAIT_VM(4);
    LA_SYNC();                                             // every wave has read the x / output tiles from LDS
  }
}

}  // namespace

bool linattn_fused_eligible(int C, int heads, int dh, int N, bool is_bf16) {
  if (C == 256) return linattn_fused256_eligible(C, heads, dh, N, is_bf16);
  return is_bf16 && C == 128 && heads == 4 && dh == 32 && N % TM == 0 && (size_t)N * 256 < (1ull << 31);
}

static int la1_strip(int N) { return N >= 65536 ? 2048 : (N >= 16384 ? 1024 : (N >= 2048 ? 512 : 256)); }

size_t linattn_fused_workspace(int B, int N) {
  const size_t nch = (size_t)cdiv(N, la1_strip(N));
  return ((size_t)B * 4 * nch * (64 + 1024) + (size_t)B * 4 * 1024 + (size_t)B * N) * sizeof(float);
}

// host-side operand preparation
//   wkv_img: [256 rows = k(128) | v(128)][128 c] bf16, rows XOR-swizzled into the LDS image, gains folded
//   wq:      [128][128] bf16 row-major, gains folded;  wout: [128][128] bf16 row-major
void linattn_fused_pack(const float* to_qkv /*[384][C]*/, const float* norm_g /*[C]*/, const float* to_out /*[C][128]*/,
                        int C, std::vector<unsigned short>& wkv_img, std::vector<unsigned short>& wq,
                        std::vector<unsigned short>& wout) {
  if (C == 256) return linattn_fused256_pack(to_qkv, norm_g, to_out, C, wkv_img, wq, wout);
  const float sq = sqrtf((float)C);
  wkv_img.assign(256 * 128, 0);
  wq.assign(128 * 128, 0);
  wout.assign((size_t)C * 128, 0);
  for (int row = 0; row < 256; ++row)
    for (int c = 0; c < 128; ++c) {
      const float v = to_qkv[(size_t)(128 + row) * C + c] * (norm_g[c] * sq);
      const int chunk = (c >> 3) ^ (row & 15);
      wkv_img[row * 128 + chunk * 8 + (c & 7)] = f32_to_bf16_host(v);
    }
  for (int d = 0; d < 128; ++d)
    for (int c = 0; c < 128; ++c) wq[d * 128 + c] = f32_to_bf16_host(to_qkv[(size_t)d * C + c] * (norm_g[c] * sq));
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < 128; ++k) wout[c * 128 + k] = f32_to_bf16_host(to_out[(size_t)c * 128 + k]);
}

int linattn_fused(const void* x, void* y, int B, int N, int C, const void* wkv_img, const void* wq, const void* wout,
                  const float* bout, const float* g2_scaled, float* ws, hipStream_t st, void* y_q, void* y_s) {
  const int strip = la1_strip(N);
  const int nstrips = cdiv(N, strip);
  const int nch = nstrips;
  const size_t bh = (size_t)B * 4;
  float* pm = ws;
  float* pl = pm + bh * nch * 32;
  float* pctx = pl + bh * nch * 32;
  float* ctxn = pctx + bh * nch * 1024;
  float* rinv = ctxn + bh * 1024;
  if (C == 256)
    return linattn_fused256(x, y, B, N, C, wkv_img, wq, wout, bout, g2_scaled, pm, pl, pctx, ctxn, rinv, strip, st, y_q, y_s);
  if (C != 128) SRGD_FAIL("linattn_fused: C must be 128 or 256");
  static bool attr[64] = {};
  const int lds1 = RING * TILE_BYTES + 2 * TM * 4;
  const int lds2 = (RING + 1) * TILE_BYTES + 4 * TM * 4 + RING * 4 * TM * 4;
  if (DeviceSetup once(attr); once.need) {
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&la1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
    SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&la2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    once.done();
  }
#if SRGD_LA_STAMPS
  void* stp = nullptr;
  SRGD_HIP(hipGetSymbolAddress(&stp, HIP_SYMBOL(g_la1_stamps)));
  SRGD_HIP(hipMemsetAsync(stp, 0, 64, st));
#endif
  hipLaunchKernelGGL(la1_kernel, dim3(nstrips, B), dim3(NTH), lds1, st, (const bf16*)x, N, (const bf16*)wkv_img, strip, pm,
                     pl, pctx, rinv);
  SRGD_HIP(hipGetLastError());
#if SRGD_LA_STAMPS
  {
    unsigned long long h[8];
    SRGD_HIP(hipStreamSynchronize(st));
    SRGD_HIP(hipMemcpy(h, stp, sizeof(h), hipMemcpyDeviceToHost));
    const double n = h[5] ? (double)h[5] : 1.0;
    fprintf(stderr, "[la1 stamps] B %d N %d: per 64-pixel tile: norms+sync %.0f  kv GEMM %.0f  scale/max/exp %.0f  rescale+context %.0f  "
                    "DMA wait+barrier %.0f  (s_memtime ticks, wave 0)\n", B, N, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
  }
#endif
  SRGD_TRY(linear_attention_combine(pm, pl, pctx, (int)bh, nch, 1.0f / sqrtf(32.0f), ctxn, st));
  La2Args a;
  a.x = (const bf16*)x; a.y = (bf16*)y; a.N = N; a.wq = (const bf16*)wq; a.wout = (const bf16*)wout; a.bout = bout;
  a.g2 = g2_scaled; a.ctxn = ctxn; a.rinv = rinv;
  a.yq = (unsigned char*)y_q; a.ys = (unsigned char*)y_s;
  const int ntiles = N / TM;
  // persistent-ish: enough tiles per workgroup to amortise the register-resident operands (64 + 40 VGPRs of weights, context,
  // bias and gain), enough workgroups to fill 256 CUs x 2-3 resident workgroups
  int tpw = 1;
  while (tpw < 16 && (long)B * cdiv(ntiles, tpw * 2) >= 1024) tpw *= 2;
  a.tiles_per_wg = tpw;
  hipLaunchKernelGGL(la2_kernel, dim3(cdiv(ntiles, tpw), B), dim3(NTH), lds2, st, a);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
