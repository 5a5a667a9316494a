// 3x3 / stride 1 / pad 1 convolution, bf16 in - fp32 accumulate - bf16 out, for gfx950 (MI355X).
// This is the kernel that carries 89 % of the FLOPs of a ConditionalSRUnet evaluation
// (reference: Block.proj model.py:246, the last-stage 3x3 "resample" convs :647,:668).
//
// Implicit GEMM, M = output pixels, N = Cout, K = 9 * Cin, laid out for CDNA4:
//   * workgroup = 512 threads = 8 waves (4 along M x 2 along N), output tile = an 8 x 32 pixel patch
//     (M = 256) x 128 output channels; each wave owns 2 patch rows x 64 channels = 2x2 MFMA 32x32 blocks.
//   * K is walked channel-chunk-major: for every 32-channel chunk the (8+2) x (32+2) halo patch is
//     staged ONCE in LDS and all 9 taps are served from it by shifting the read address - global->LDS
//     traffic for A drops 9x/1.33 versus gathering a fresh A tile per tap.
//   * all staging is LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip): the A patch
//     (out-of-image halo pixels are zero-filled by the buffer descriptor's range check), and per K-step one
//     8 KB weight tile, pre-swizzled on the host into its LDS image so the copy is linear and coalesced.
//     A is double-buffered, B runs in a 3-deep ring; loads stay in flight across barriers
//     (counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per K-step).  A 4-deep ring measured the same.
//   * 64-byte LDS rows are XOR-swizzled (chunk ^= (row >> 2) & 3): ds_read_b128 is conflict-free for the
//     MFMA operand pattern (16 consecutive rows, same chunk) at every tap shift.
//   * epilogue: + bias, optional per-(sample, group) partial sum / sum of squares for the GroupNorm that
//     follows (fixed-order, deterministic), bf16 store.
//   * blockIdx is remapped so each XCD (private L2) gets a contiguous band of tiles (halo reuse in L2).
//   * GNIN instances (template parameter): the PRODUCER's GroupNorm-apply + SiLU is applied to a chunk's halo patch in LDS right
//     after it lands (reference Block.forward model.py:250-259 between two convolutions), which removes a full HBM pass; these
//     instances issue their MFMAs as inline asm with the accumulator tied (no register migration: 123 VGPRs, no spills).
// What bounds it (round 4, in-kernel stamps): the K loop of the deep layers uses 99 % of the MFMA issue slots at a power-limited
// 1.7 GHz; the 128-channel layers lose ~25 % of a tile's lifetime to prologue + epilogue that the co-resident workgroup only partly
// covers.  The compile-time switches below are the A/B and timing-only builds those statements rest on (DESIGN.md section 4.1,
// profiles/r4/conv3x3_bf16_clock_and_dma_diagnostics.txt); the shipped configuration is their defaults.
#include <cstdlib>

#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int PH = 8, PW = 32;                 // output patch
constexpr int HP = PH + 2, WP = PW + 2;        // halo patch: 10 x 34 = 340 pixels
constexpr int KC = 32;                         // channels per chunk (64 B rows)
constexpr int BN3 = 128;
constexpr int NT3 = 512;
constexpr int A_BYTES = 24 * 1024;             // 24 wave-instructions x 1 KiB (340 px * 64 B = 21,760 used)
constexpr int B_BYTES = BN3 * KC * 2;          // 8 KiB
constexpr int LDS_BYTES = 2 * A_BYTES + 3 * B_BYTES;   // 73,728: two workgroups per CU
constexpr int COEF_BYTES = 2 * 256;            // GNIN: 32 scales | 32 shifts (fp32) per channel chunk, double-buffered
#ifndef SRGD_CONV3_PAIR_WRITES
#define SRGD_CONV3_PAIR_WRITES 1      // epilogue: dword LDS writes after a lane-pair exchange (0: four 2-byte writes per block; A/B builds)
#endif
#ifndef SRGD_GNIN_SCALAR
#define SRGD_GNIN_SCALAR 1            // GNIN transform: single-lane-op fp32 arithmetic (inline asm) instead of what -O3 SLP-packs into
#endif                                // v_pk_fma_f32 / v_pk_mul_f32 - packed f32 VALU beside MFMAs is an anti-lever on gfx950 (A/B builds: 0)
#ifndef SRGD_CONV3_DMA_POS
#define SRGD_CONV3_DMA_POS 0          // where a tap issues its LDS-DMA requests: 0 top of the tap, 1 behind its fragment loads, 2 behind its MFMAs, 3 behind its first 8 MFMAs, 4 top of the tap with the explicit-address tap body (A/B)
#endif
#ifndef SRGD_CONV3_EPI_SCALAR
#define SRGD_CONV3_EPI_SCALAR 0       // epilogue bias / GroupNorm sums with single-lane-op instructions instead of v_pk_* (A/B builds)
#endif
#ifndef SRGD_CONV3_EPI_PRIO
#define SRGD_CONV3_EPI_PRIO 0         // s_setprio level of the epilogue (0 = leave it at the kernel's default)
#endif
#ifndef SRGD_CONV3_TIED
#define SRGD_CONV3_TIED 0             // plain instances: tied inline-asm MFMAs as well (A/B builds)
#endif
#ifndef SRGD_GNIN_LEAN
#define SRGD_GNIN_LEAN 3              // GNIN instances: the plain instances' shared fragment addressing and 8-fragment tap (needs the tied MFMAs)
#endif
constexpr int CONV3_XCD_PIN_KB_DEFAULT = 0;   // (A/B knob SRGD_CONV3_XCD_PIN_KB; see the kernel's tile map)
constexpr int CONV3_M16_DEFAULT = 1;     // 16x16x32 measured +1..2 % over 32x32x16 on the production shapes

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset, int soffset = 0) {
  // LDS destination = wave-uniform base + lane * 16; voffset per lane (VGPR), soffset wave-uniform (SGPR)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds_wave_base, 16, voffset, soffset, 0, 0);
}

struct Conv3Args {
  const bf16* in0; const bf16* in1; int C0, C1;
  int B, H, W;
  const bf16* w;          // packed [tap][cc][ntile][128 rows][4 swizzled chunks][8]
  const float* bias;
  int Cout;
  bf16* out;
  float* gn_partial; int groups;
  const float* gn_in_a;   // GNIN: y = silu(a[b][c] * x + b[b][c]) applied to the input while it is staged ([B][Cin] fp32)
  int gn_in_b_off;        // byte offset of the shift array from the scale array (same allocation)
  int xcd_pin_ntiles;     // blockIdx -> tile map: n-tiles pinned to XCDs (weights stay in the XCD's L2); else contiguous bands of tiles per XCD
  unsigned long long* stamps;   // diagnostics (SRGD_CONV3_STAMPS=1): per-phase s_memtime deltas summed over workgroups; null otherwise
};

// phase accumulators of the diagnostic mode: [prologue, main loop, LDS transpose, stores, statistics, total, workgroups]
__device__ unsigned long long g_conv3_stamps[8];
__device__ unsigned long long g_conv3_stamps2[4];   // the transposition phase split (wave 0 of each workgroup): accumulator -> LDS, statistics, barrier wait

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// Raw barrier (no vmcnt drain: LDS-DMA prefetches stay in flight) fenced for the instruction scheduler:
// s_barrier is IntrNoMem to LLVM, so without sched_barrier(0) the machine scheduler hoists the next step's
// ds_reads above it - a read of a buffer whose DMA other waves have not yet waited for.
#define BARRIER()                        \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// M16 selects the MFMA shape: false = v_mfma_f32_32x32x16_bf16 (2x2 blocks per wave, two k16 steps per chunk),
// true = v_mfma_f32_16x16x32_bf16 (4x4 blocks, one k32 step).  Same LDS bytes read per FLOP and the same
// accumulator count; the 16x16 shape sustains a higher clock on MI355X (MI355X_MICROARCH.md, DVFS item 7).
// Row swizzle: chunk ^= (row >> 2) & 3 for the 32x32 operand pattern (32 rows x 1 chunk per lane half),
//              chunk ^= (row >> 1) & 3 for the 16x16 pattern (16 rows x 4 chunks) - each conflict-free for its
//              ds_read_b128 lane groups at every tap shift (checked exhaustively on the bank model).
template <bool M16> __device__ __forceinline__ int row_swz(int row) { return M16 ? (row >> 1) & 3 : (row >> 2) & 3; }

template <bool STATS, bool GNIN, bool M16>
__global__ __launch_bounds__(NT3, 4) void conv3x3_bf16_kernel(Conv3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;            // 32x32 shape: row / k half
  const int r16 = lane & 15, q16 = lane >> 4;        // 16x16 shape: row / 8-channel chunk

  // ---- tile coordinates (XCD-aware remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous band)
  const int n_tiles = p.Cout / BN3;
  const int tiles_x = p.W / PW, tiles_y = p.H / PH;
  const int m_tiles = p.B * tiles_y * tiles_x;
  const int nwg = m_tiles * n_tiles;
  int wg = blockIdx.x;
  int nt, mt;
  if (p.xcd_pin_ntiles) {
    // Round 4: pin the n-tiles to XCDs.  Blocks b, b + 8, ... share an XCD (round-robin dispatch).  With n-tiles walking fastest
    // inside an XCD's band (below), every XCD streams ALL weights of the layer (18.9 MB at 1024 -> 1024) through its 4 MiB L2
    // once per m-tile: the weight tiles come over the fabric from the Infinity Cache every time (~9.4 GB per launch), and a
    // timing build without the K loop's DMA traffic holds 2.34 GHz where production holds 1.73 GHz at 99 % MFMA issue - the
    // kernel is clock- (power-) limited and the data movement is what the clock pays for.  Here XCD x owns n-tile x % n_tiles
    // for all of its m-tiles: its weight working set is 1 / n_tiles of the layer (2.4 MB) and stays L2-resident; the halo
    // patches are fetched once per XCD that needs them instead (n_tiles x activation bytes over the fabric, 8x less in total).
    const int x = wg & 7, k = wg >> 3, per = 8 / n_tiles;        // host: n_tiles in {2, 4, 8} and m_tiles % per == 0
    nt = x % n_tiles;
    mt = k * per + x / n_tiles;
  } else {
    const int q = nwg >> 3, rem = nwg & 7, x = wg & 7, k = wg >> 3;
    wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
    nt = wg % n_tiles;
    mt = wg / n_tiles;
  }
  const int b = mt / (tiles_y * tiles_x);
  const int trem = mt - b * tiles_y * tiles_x;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * PH, x0 = tx * PW;
  const int Cin = p.C0 + p.C1;
  const int CC = Cin / KC;

  // ---- A staging: 24 wave-instructions per chunk; wave w issues pieces w, w+8, w+16 (pieces >= 22 are all-zero).
  // Per-lane pixel offset (y*W+x) or -1 and source chunk (0..3) of its three pieces, as NAMED scalars (indexed
  // arrays of staging state end up in scratch: see conv_igemm.hip).
  // The source chunk is the same for all three pieces: P = (wave + 8 J) * 16 + (lane >> 2), and (wave + 8 J) * 16 vanishes
  // from row_swz(P) (a multiple of 8 under (P >> 1) & 3, of 4 under (P >> 2) & 3) - ONE register, not three.
#define SRGD_A_DECL(J)                                                        \
  int a_pix##J;                                                               \
  {                                                                           \
    const int g = (wave + 8 * J) * 64 + lane; /* 16-byte chunk in the image */ \
    const int P = g >> 2;                                                     \
    const int py = P / WP, px = P - py * WP;                                  \
    const int y = y0 + py - 1, x = x0 + px - 1;                               \
    const bool ok = P < HP * WP && y >= 0 && y < p.H && x >= 0 && x < p.W;    \
    a_pix##J = ok ? y * p.W + x : -1;                                         \
  }
  SRGD_A_DECL(0) SRGD_A_DECL(1) SRGD_A_DECL(2)
#undef SRGD_A_DECL
  const int a_sub = (lane & 3) ^ row_swz<M16>(lane >> 2);
  // GNIN: which of this lane's three pieces lie inside the image, packed (bit J)
  [[maybe_unused]] const int a_in = GNIN ? (a_pix0 >= 0 ? 1 : 0) | (a_pix1 >= 0 ? 2 : 0) | (a_pix2 >= 0 ? 4 : 0) : 0;
  // GNIN (one source): byte offset of each piece at chunk 0, or the out-of-range sentinel (stays out of range for every chunk)
  const unsigned a_off0 = a_pix0 >= 0 ? (unsigned)(a_pix0 * p.C0 + a_sub * 8) * 2u : 0x7ffffff0u;
  const unsigned a_off1 = a_pix1 >= 0 ? (unsigned)(a_pix1 * p.C0 + a_sub * 8) * 2u : 0x7ffffff0u;
  const unsigned a_off2 = a_pix2 >= 0 ? (unsigned)(a_pix2 * p.C0 + a_sub * 8) * 2u : 0x7ffffff0u;
  const size_t img_elems0 = (size_t)p.H * p.W * p.C0, img_elems1 = (size_t)p.H * p.W * p.C1;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.in0 + (size_t)b * img_elems0), 0, (int)(img_elems0 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in1 ? p.in1 + (size_t)b * img_elems1 : p.in0), 0, p.in1 ? (int)(img_elems1 * 2) : 0, 0x00020000);
  const size_t w_tile_stride = (size_t)n_tiles * B_BYTES;             // bytes between consecutive (tap, cc) tiles
  const char* w_base = (const char*)p.w + (size_t)nt * B_BYTES;
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, (int)((size_t)(9 * CC - 1) * w_tile_stride + B_BYTES), 0x00020000);

  char* const sA0 = smem;
  char* const sB0 = smem + 2 * A_BYTES;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(GNIN ? p.gn_in_a + (size_t)b * Cin : (const float*)p.in0), 0, GNIN ? p.gn_in_b_off + Cin * 4 : 0, 0x00020000);

  // GNIN: GroupNorm-apply + SiLU of the PRODUCER fused into this conv's staging (reference Block.forward
  // model.py:250-259 between two convs): once a wave's own DMA pieces of a chunk have landed it rewrites them in
  // place, y = silu(a*x + b); out-of-image halo chunks stay zero (the conv pads the activated tensor).
  // Round 3: the tap loop of the GNIN instances is branch-free.  Round 2's version had a per-lane `if (a_pix < 0) return` here
  // and a wave-uniform `if (piece == 22)` around the coefficient DMA; either one splits the unrolled tap loop into basic
  // blocks, and the 16x16 instance then spilled 19-25 VGPRs into the K loop (the GroupNorm-in-staging A/B of round 2 was
  // measured on that spilling kernel).  Now: out-of-image chunks are ANDed back to zero, the DMA offsets are precomputed
  // (no multiply / select in the loop: the compiler if-converts those into exec-masked branches too), and the chunk's
  // coefficients (lane l: scale[c + l] or shift[c + l - 32]) come in through ONE 4-byte-per-lane LDS-DMA per wave into a
  // 256-byte slot of their own (every wave issues it: identical bytes, and the counted vmcnt waits stay the same for all
  // waves).  No VGPR load: the compiler would guard its use with s_waitcnt vmcnt(0) - it cannot see the counted waits - and
  // drain every DMA in flight.
  int tid16 = tid * 16;                         // the weight DMA's per-lane offset; GNIN: lane * 16, lane * 4 and (after the loop) tid are derived from it
  int opq = 0;                                  // opaque zero, refreshed once per channel chunk (see the operand addresses below)
  char* const sCoef = smem + LDS_BYTES;
#if SRGD_GNIN_SCALAR
  // the per-lane offset is rebuilt from tid16 at its one use per chunk (3 VALU) rather than carried through the K loop
  auto coef_dma = [&](int cc) {
    int t = tid16;
    asm volatile("" : "+v"(t));
    const int l4 = (t >> 2) & 0xfc;               // lane * 4
    const int voff = l4 < 128 ? l4 : p.gn_in_b_off + l4 - 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (lds_ptr)(sCoef + (cc & 1) * 256), 4, voff, cc * KC * 4, 0, 0);
  };
#else
  const int coef_voff = lane < 32 ? lane * 4 : p.gn_in_b_off + (lane - 32) * 4;
  auto coef_dma = [&](int cc) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (lds_ptr)(sCoef + (cc & 1) * 256), 4, coef_voff, cc * KC * 4, 0, 0);
  };
#endif
#if SRGD_GNIN_SCALAR
  // Round 4.  The transform shares the SIMD's vector-issue port with the MFMAs (an MFMA 16x16x32 holds it for 8 of its 16
  // cycles), so it is written for issue cycles: (1) plain v_fma_f32 / v_mul_f32 through inline asm - the compiler packs the
  // four lanes' affine step and final product into v_pk_fma_f32 / v_pk_mul_f32, which beside MFMAs cost far more than the
  // two single ops they replace (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); (2) two address instructions per
  // half instead of five: per-lane parts (lane * 16, a_sub * 32) live in registers, everything uniform rides in an opaque
  // SGPR (opaque so that lane part + uniform part is not hoisted into one VGPR per (buffer, piece)), the half / shift
  // offsets are ds immediates; (3) out-of-image chunks keep the zeros the DMA wrote because their lanes are switched off
  // for the store (exec = the piece's in-image ballot, two SALU) instead of two v_cndmask.  Same arithmetic as
  // silu<false> in gn_apply: bit-identical results.
  const unsigned asub32 = (unsigned)a_sub * 32u;
#define lane16 ((unsigned)tid16 & 1023u)
  const unsigned long long in_m0 = __builtin_amdgcn_ballot_w64(a_pix0 >= 0), in_m1 = __builtin_amdgcn_ballot_w64(a_pix1 >= 0),
                           in_m2 = __builtin_amdgcn_ballot_w64(a_pix2 >= 0);
  const unsigned smem_lds = (unsigned)(size_t)(lds_ptr)smem;
  auto transform_half = [&](int cc, int j, int hf) {
    int sq = (int)smem_lds + (cc & 1) * A_BYTES + (wave + 8 * j) * 1024 + hf * 8;
    int sc = (int)smem_lds + LDS_BYTES + (cc & 1) * 256 + hf * 16;
    asm volatile("" : "+s"(sq), "+s"(sc));
    const unsigned qa = lane16 + (unsigned)sq, ca_ = asub32 + (unsigned)sc;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const u32x2* lds_u2;
    typedef __attribute__((address_space(3))) const f32x4* lds_f4;
    const u32x2 raw = *(lds_u2)(size_t)qa;
    const f32x4 ca = *(lds_f4)(size_t)ca_, cb = *(lds_f4)(size_t)(ca_ + 128u);
    const unsigned w0 = raw[0], w1 = raw[1];
    const float x0 = __uint_as_float(w0 << 16), x1 = __uint_as_float(w0 & 0xffff0000u), x2 = __uint_as_float(w1 << 16),
                x3 = __uint_as_float(w1 & 0xffff0000u);
    // Two elements per asm block, interleaved: on gfx950 a VALU instruction may not read a transcendental's result in the very
    // next issue slot (one wait state; the compiler inserts it for its own instructions but does not look inside inline asm -
    // a build that scheduled v_rcp directly ahead of a single-instruction asm v_mul computed garbage), so each v_exp / v_rcp
    // is followed by its sibling's before its result is used.
    float y[4];
#define SRGD_SILU2(Y0_, Y1_, E0_, E1_)                                                                                           \
    do {                                                                                                                         \
      float t0_, t1_;                                                                                                            \
      asm("v_fma_f32 %0, %4, %6, %8\n\tv_fma_f32 %1, %5, %7, %9\n\t"                                                           \
          "v_mul_f32 %2, 0xbfb8aa3b, %0\n\tv_mul_f32 %3, 0xbfb8aa3b, %1\n\t"                                                   \
          "v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"                                                                           \
          "v_add_f32 %2, 1.0, %2\n\tv_add_f32 %3, 1.0, %3\n\t"                                                                 \
          "v_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"                                                                           \
          "v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3"                                                                        \
          : "=&v"(Y0_), "=&v"(Y1_), "=&v"(t0_), "=&v"(t1_)                                                                       \
          : "v"(ca[E0_]), "v"(ca[E1_]), "v"(x##E0_), "v"(x##E1_), "v"(cb[E0_]), "v"(cb[E1_]));                                   \
    } while (0)
    SRGD_SILU2(y[0], y[1], 0, 1);
    SRGD_SILU2(y[2], y[3], 2, 3);
#undef SRGD_SILU2
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const unsigned o0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{y[0], y[1]}, bf16x2_t));
    const unsigned o1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{y[2], y[3]}, bf16x2_t));
    const unsigned long long bits64 = (unsigned long long)o0 | ((unsigned long long)o1 << 32);
    const unsigned long long m = j == 0 ? in_m0 : (j == 1 ? in_m1 : in_m2);
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(qa), "v"(bits64), "s"(m) : "memory");
  };
#undef lane16
#else
  auto transform_half = [&](int cc, int j, int hf) {
    // all ones / zero: out-of-image chunks are ANDed back to zero (a select on `inside` is if-converted by the compiler into
    // an exec-masked branch around the arithmetic: a basic-block split inside the unrolled tap loop, see above)
    const int keep = __builtin_amdgcn_sbfe(a_in, j, 1);
    // + opq: recomputed at every use (~3 VALU ops) instead of hoisted out of the K loop into a dozen long-lived VGPRs
    char* q = sA0 + (cc & 1) * A_BYTES + (wave + 8 * j) * 1024 + (lane + opq) * 16;
    const float* sC = reinterpret_cast<const float*>(sCoef + (cc & 1) * 256) + (a_sub + opq) * 8;
    // 8-byte halves: small live temporaries (the kernel sits at the 128-VGPR cap of 2 workgroups/CU) and a unit of
    // VALU work (8 transcendentals per lane) short enough to hide under one tap's MFMAs of the other waves
    bf16x4 v = *reinterpret_cast<const bf16x4*>(q + hf * 8);
    const f32x4 ca = *reinterpret_cast<const f32x4*>(sC + hf * 4);
    const f32x4 cb = *reinterpret_cast<const f32x4*>(sC + 32 + hf * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16)silu<false>(ca[e] * (float)v[e] + cb[e]);
    int2 bits = __builtin_bit_cast(int2, v);
    bits.x &= keep;
    bits.y &= keep;
    // The store goes out as inline asm: a compiler-visible ds_write to LDS that LDS-DMA also writes is guarded with
    // s_waitcnt vmcnt(0) (write-after-write on "LDS" as a whole), which drains the A piece and the weight tile issued a few
    // instructions earlier - one full L2 round trip per tap, the hidden cost of round 2's GNIN build.  The piece rewritten
    // here is this wave's own and landed under the previous tap's counted wait.
    const long long bits64 = __builtin_bit_cast(long long, bits);
    asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)(lds_ptr)(q + hf * 8)), "v"(bits64) : "memory");
  };
#endif
#define transform_a_half(CCV, J, HF) transform_half(CCV, J, HF)
#define transform_a_piece(CCV, J) do { transform_a_half(CCV, J, 0); transform_a_half(CCV, J, 1); } while (0)

  auto issue_a = [&](int cc, int j, int a_pix) {
    const int c = cc * KC;
    const bool first = c < p.C0;
    const int Cs = first ? p.C0 : p.C1;
    const int coff = first ? c : c - p.C0;
    const int voff = a_pix >= 0 ? (a_pix * Cs + coff + a_sub * 8) * 2 : 0x7ffffff0;
    char* dst = sA0 + (cc & 1) * A_BYTES + (wave + 8 * j) * 1024;
    if (GNIN) {                                   // the chunk rides in the scalar offset: no per-lane arithmetic in the loop
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)dst, 16, (int)(j == 0 ? a_off0 : (j == 1 ? a_off1 : a_off2)),
                                               cc * KC * 2, 0, 0);
      return;
    }
    if (first) dma16(rs0, dst, voff);
    else dma16(rs1, dst, voff);
  };
#define issue_a_piece(CCV, J) issue_a(CCV, J, (J) == 0 ? a_pix0 : ((J) == 1 ? a_pix1 : a_pix2))
  // K-step (cc, tap) -> weight tile (tap, cc) into ring slot (cc * 9 + tap) % 3 = tap % 3.  Called with compile-time `tap`
  // (0..10: the unrolled tap loop asks for "two steps ahead"; 9 and 10 mean taps 0 and 1 of the next chunk), so the tile offset is
  // one scalar multiply-add - round 2's issue_b(s) divided the runtime step index by 9 and by 3: ~20 SALU instructions per tap.
  const int w_tap_stride = (int)(CC * w_tile_stride);
  auto issue_b = [&](int cc, int tap) {
    if (tap >= 9) { tap -= 9; cc += 1; }
    dma16(rsw, sB0 + (tap % 3) * B_BYTES + wave * 1024, tid16, tap * w_tap_stride + cc * (int)w_tile_stride);
  };

  // ---- accumulators: 64 fp32 per lane in both shapes
  f32x16 acc00 = 0, acc01 = 0, acc10 = 0, acc11 = 0;                                     // 32x32: [mi][ni]
  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0,          // 16x16: [mi][ni]
        c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;

  // ---- operand read addresses.  `opq` is an opaque zero refreshed once per channel chunk: it stops the compiler
  // from hoisting the per-tap A addresses out of the K loop into ~18 long-lived VGPRs (the kernel lives at the
  // 128-VGPR cap of 2 workgroups per CU); recomputing one costs ~5 VALU ops.
  // 32x32: B row n = wn*64 + j*32 + r, logical chunk 2*s2 + h;  16x16: B row n = wn*64 + j*16 + r16, chunk q16
  // (the swizzle of weight row n = wn*64 + j*16 + r16 (or j*32 + r) does not depend on j or wn: ONE per-lane base, the
  // column block rides in the ds_read offset field)
  const int b_base = (M16 ? wn * 64 + r16 : wn * 64 + r) * 64 + (((M16 ? q16 : h) ^ row_swz<M16>(M16 ? r16 : r)) << 4);
  auto b_addr = [&](int j) { return b_base + j * (M16 ? 16 : 32) * 64; };
  // 16x16 (round 3): halo pixel P = lp + Pc with lp = 2 wm WP + r16 (per lane) and Pc a compile-time constant per (tap, block):
  // P * 64 splits into lp * 64 (one per-lane base) + Pc * 64 (the ds_read's immediate offset), and the swizzle term
  // ((P >> 1) & 3) << 4 = ((P << 3) & 0x30) comes from lp8 = lp << 3: THREE VALU instructions per fragment address (add3 with
  // the opaque zero, and-xor, add) instead of six - the K loop carried ~28 address instructions per 16 MFMAs on the port the
  // MFMAs issue through.
  const int lp = 2 * wm * WP + r16, lp8 = lp << 3, lp64 = lp * 64, q16s = q16 << 4;
  int lp8o = lp8;                               // GNIN instances: refreshed (made opaque) at every tap
  auto a_addr = [&](int tap, int i) {        // 32x32: i = patch row of the wave (0/1); 16x16: i = 16-pixel block (0..3)
    const int dy = tap / 3, dx = tap - dy * 3;
    if constexpr (M16 && (!GNIN || (SRGD_GNIN_LEAN & 1))) {
      const int Pc = ((i >> 1) + dy) * WP + (i & 1) * 16 + dx;
      return lp64 + (((lp8 + Pc * 8 + opq) & 0x30) ^ q16s) + Pc * 64;
    } else if constexpr (M16) {
      // the GNIN instances refresh `opq` every tap so that NO address part survives a tap in a register (they carry ~13 more
      // long-lived registers; with the split form above the eight swizzle terms of a chunk stay live and 33 registers spill into
      // the K loop): P * 8 once, then P * 64 and the swizzle term from it - four VALU instructions per address instead of six
      // (round 4: an opaque COPY of lp8 per tap instead of an opaque zero added to it - the compiler reassociated
      // (lp8 + Pc * 8) + opq, hoisted the loop-invariant first half for every (tap, block) and spilled what did not fit)
      const int Pc = ((i >> 1) + dy) * WP + (i & 1) * 16 + dx;
      const int t = lp8o + Pc * 8;
      return (t << 3) + ((t & 0x30) ^ q16s);
    } else {
      const int P = (2 * wm + i + dy) * WP + r + dx + opq;
      return P * 64 + ((h ^ row_swz<M16>(P)) << 4);
    }
  };
  // after_loads / after_half: hooks run behind the tap's fragment loads / behind its first 8 MFMAs (16x16 shape): where the
  // step's LDS-DMA requests are issued is a tuning knob (SRGD_CONV3_DMA_POS) - a timing-only build without any DMA in the K loop
  // runs 30-38 % faster and one that issues but never waits runs the same as production, i.e. the ISSUE of the 1-2 DMA
  // instructions per wave and tap sits on the tap's critical path, not their latency
  auto compute = [&](int cc, int tap, int s) {
    const char* A = sA0 + (cc & 1) * A_BYTES;
    const char* Bt = sB0 + (tap % 3) * B_BYTES;          // (cc * 9 + tap) % 3
    if constexpr (M16 && GNIN && !(SRGD_GNIN_LEAN & 2)) {
      // GNIN carries ~13 more long-lived registers (piece offsets, coefficient addressing): the pixel fragments come in two
      // pairs here - 24 operand registers at a time instead of 32 - so that nothing spills into the K loop (a scratch reload
      // in this loop is a VMEM load the compiler guards with s_waitcnt vmcnt(0): it drains the DMA pipeline)
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(0));
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(1));
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(2));
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(3));
      bf16x8 a0 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 0));
      bf16x8 a1 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 1));
      // Round 4: the MFMAs of the GNIN instances go out as inline asm with the accumulator TIED (D = C).  Through the builtin the
      // compiler gives every MFMA a fresh destination, the 64 accumulators migrate through the register file, and at the
      // 128-register cap that fragmentation (not the live count: 97 at the loop's fullest point) spilled 3-4 loop invariants
      // into the K loop - scratch reloads behind s_waitcnt vmcnt(0), i.e. behind every DMA in flight.  (Hazards: operands come
      // from ds_read, whose waits the compiler inserts for asm operands too; the accumulators are first read by VALU code
      // after the loop, behind the s_nop block ahead of the epilogue.)
#define MM(C_, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C_) : "v"(A_), "v"(B_))
      MM(c00, a0, b0); MM(c01, a0, b1); MM(c02, a0, b2); MM(c03, a0, b3);
      __builtin_amdgcn_sched_barrier(0);
      a0 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 2));
      MM(c10, a1, b0); MM(c11, a1, b1); MM(c12, a1, b2); MM(c13, a1, b3);
      __builtin_amdgcn_sched_barrier(0);
      a1 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 3));
      MM(c20, a0, b0); MM(c21, a0, b1); MM(c22, a0, b2); MM(c23, a0, b3);
      MM(c30, a1, b0); MM(c31, a1, b1); MM(c32, a1, b2); MM(c33, a1, b3);
#undef MM
    } else if constexpr (M16) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 0));
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 1));
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 2));
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 3));
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(0));
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(1));
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(2));
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(3));
#define MM(C_, A_, B_)                                                                                   \
  do {                                                                                                   \
    if constexpr (GNIN || SRGD_CONV3_TIED) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C_) : "v"(A_), "v"(B_)); \
    else C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, C_, 0, 0, 0);                             \
  } while (0)
      MM(c00, a0, b0); MM(c01, a0, b1); MM(c02, a0, b2); MM(c03, a0, b3);
      MM(c10, a1, b0); MM(c11, a1, b1); MM(c12, a1, b2); MM(c13, a1, b3);
      MM(c20, a2, b0); MM(c21, a2, b1); MM(c22, a2, b2); MM(c23, a2, b3);
      MM(c30, a3, b0); MM(c31, a3, b1); MM(c32, a3, b2); MM(c33, a3, b3);
#undef MM
    } else {
      const int a0 = a_addr(tap, 0), a1 = a_addr(tap, 1), b0 = b_addr(0), b1 = b_addr(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int x = s2 << 5;                 // k16 step toggles bit 1 of the chunk index = byte bit 5
        const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(A + (a0 ^ x));
        const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(A + (a1 ^ x));
        const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(Bt + (b0 ^ x));
        const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(Bt + (b1 ^ x));
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc11, 0, 0, 0);
      }
    }
  };

  // The 8-fragment tap of the 16x16 shape written out at loop level (no closure between the tap loop and issue_a / issue_b: a
  // nested lambda around them made the compiler keep the kernel arguments and captured descriptors in scratch), with the tap's
  // LDS-DMA requests (DMA_) placed behind the fragment loads (SRGD_CONV3_DMA_POS 1) or behind the first 8 MFMAs (3).
  // Fragment addresses here are explicit: halo pixel P = lp + Pc, Pc = (block row + dy) * 34 + 16 (x half) + dx a compile-time
  // constant per (tap, block); 34 = 2 mod 8, so the swizzle term ((P >> 1) & 3) depends on the lane and on Pc & 7 only: eight
  // per-lane row bases (one per value of Pc & 7), the buffer of the chunk rides in an opaque SGPR (one v_add per fragment, and
  // nothing for the compiler to reassociate and hoist: the lambda form's (lp8 + Pc * 8) sums turned into ~16 long-lived
  // registers here and spilled 62), Pc * 64 is the ds_read's immediate offset.
#define SRGD_AB(K) const int ab##K = lp64 + ((((lp8 + (K) * 8) & 0x30)) ^ q16s);
  SRGD_AB(0) SRGD_AB(1) SRGD_AB(2) SRGD_AB(3) SRGD_AB(4) SRGD_AB(5) SRGD_AB(6) SRGD_AB(7)
#undef SRGD_AB
#define SRGD_AFRAG(TAPV, I, SA_)                                                                           \
  ({                                                                                                       \
    const int Pc_ = (((I) >> 1) + (TAPV) / 3) * WP + ((I) & 1) * 16 + (TAPV) % 3;                          \
    const int k7_ = Pc_ & 7;                                                                               \
    const int base_ = k7_ == 0 ? ab0 : k7_ == 1 ? ab1 : k7_ == 2 ? ab2 : k7_ == 3 ? ab3 : k7_ == 4 ? ab4 : k7_ == 5 ? ab5 : k7_ == 6 ? ab6 : ab7; \
    *reinterpret_cast<const bf16x8*>(smem + (base_ + (SA_)) + Pc_ * 64);                                   \
  })
#define SRGD_TAP8(CCV, TAPV, DMA_)                                                                         \
  do {                                                                                                     \
    int sa_ = ((CCV) & 1) * A_BYTES;                                                                       \
    asm volatile("" : "+s"(sa_));                                                                          \
    const char* Bt_ = sB0 + ((TAPV) % 3) * B_BYTES;                                                        \
    const bf16x8 a0 = SRGD_AFRAG(TAPV, 0, sa_);                                                            \
    const bf16x8 a1 = SRGD_AFRAG(TAPV, 1, sa_);                                                            \
    const bf16x8 a2 = SRGD_AFRAG(TAPV, 2, sa_);                                                            \
    const bf16x8 a3 = SRGD_AFRAG(TAPV, 3, sa_);                                                            \
    const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bt_ + b_addr(0));                                   \
    const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bt_ + b_addr(1));                                   \
    const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(Bt_ + b_addr(2));                                   \
    const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(Bt_ + b_addr(3));                                   \
    if (SRGD_CONV3_DMA_POS == 1) { __builtin_amdgcn_sched_barrier(0); DMA_; __builtin_amdgcn_sched_barrier(0); } \
    SRGD_MMT(c00, a0, b0); SRGD_MMT(c01, a0, b1); SRGD_MMT(c02, a0, b2); SRGD_MMT(c03, a0, b3);            \
    SRGD_MMT(c10, a1, b0); SRGD_MMT(c11, a1, b1); SRGD_MMT(c12, a1, b2); SRGD_MMT(c13, a1, b3);            \
    if (SRGD_CONV3_DMA_POS == 3) { __builtin_amdgcn_sched_barrier(0); DMA_; __builtin_amdgcn_sched_barrier(0); } \
    SRGD_MMT(c20, a2, b0); SRGD_MMT(c21, a2, b1); SRGD_MMT(c22, a2, b2); SRGD_MMT(c23, a2, b3);            \
    SRGD_MMT(c30, a3, b0); SRGD_MMT(c31, a3, b1); SRGD_MMT(c32, a3, b2); SRGD_MMT(c33, a3, b3);            \
  } while (0)
#define SRGD_MMT(C_, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C_) : "v"(A_), "v"(B_))

  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, r0 = 0;
  if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // ---- prologue: A(0) and B[0], B[1]
  if (GNIN) coef_dma(0);
  issue_a_piece(0, 0);
  issue_a_piece(0, 1);
  issue_a_piece(0, 2);
  issue_b(0, 0);
  issue_b(0, 1);                                 // S >= 9 always
  WAIT_VM(1);
  if (GNIN) {
    BARRIER();                                   // coefficient slot visible; this wave's A(0) pieces have landed
    transform_a_piece(0, 0);
    transform_a_piece(0, 1);
    transform_a_piece(0, 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  BARRIER();
  if (p.stamps) t1 = __builtin_amdgcn_s_memtime();

  // ---- main loop.  Per K-step: [issue A piece of the next chunk (taps 0..2)] [issue B[s+2]] compute(s)
  //      wait until B[s+1] (and, in order, everything older) has landed, barrier.
  for (int cc = 0; cc < CC - 1; ++cc) {
    if (!GNIN || (SRGD_GNIN_LEAN & 1)) asm volatile("" : "+v"(opq));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = cc * 9 + tap;
      if (GNIN && !(SRGD_GNIN_LEAN & 1)) { lp8o = lp8; asm volatile("" : "+v"(lp8o)); }   // GNIN: per-tap refresh - no operand-address part survives a tap in a register
      // GNIN: the next chunk's coefficients are the OLDEST request of tap 0 (so the tap's counted wait covers them), published
      // by tap 0's barrier, read from tap 2 on
      if (GNIN && tap == 0) coef_dma(cc + 1);
#ifndef SRGD_CONV3_DIAG_NODMA               // timing-only diagnostics (wrong results): 1 = no DMA issue inside the K loop, 2 = no A pieces, 3 = no weight tiles
#define SRGD_CONV3_DIAG_NODMA 0
#endif
#define SRGD_TAP_DMA()                                                                                                  \
      do {                                                                                                             \
        /* 5 = every request re-fetches chunk 0's bytes into the slot it would fill (same DMA traffic, static LDS contents) */ \
        if (SRGD_CONV3_DIAG_NODMA == 5) { if (tap < 3) issue_a_piece((cc + 1) & 1 ? 1 : 0, tap); issue_b(0, (tap + 2) % 3 + 3); } \
        if (SRGD_CONV3_DIAG_NODMA == 0 || SRGD_CONV3_DIAG_NODMA == 3 || SRGD_CONV3_DIAG_NODMA == 4) { if (tap < 3) issue_a_piece(cc + 1, tap); } \
        if (SRGD_CONV3_DIAG_NODMA == 6 && (tap & 1) == 0) issue_b(cc, tap + 2);   /* 6 = weight tiles on even taps only: -38 % DMA bytes */ \
        if (SRGD_CONV3_DIAG_NODMA == 6 && tap < 3) issue_a_piece(cc + 1, tap);                                                                  \
        if (SRGD_CONV3_DIAG_NODMA == 0 || SRGD_CONV3_DIAG_NODMA == 2 || SRGD_CONV3_DIAG_NODMA == 4) issue_b(cc, tap + 2); /* always < S here (cc < CC-1) */ \
      } while (0)
      constexpr int DP = (M16 && (!GNIN || (SRGD_GNIN_LEAN & 2))) ? SRGD_CONV3_DMA_POS : 0;     // (the hooks exist in the 8-fragment tap only)
      if (DP == 0 || DP == 4) SRGD_TAP_DMA();
      // a wave rewrites only the pieces it DMA'd itself: piece issued at tap t has landed after the wait of tap t+1;
      // six half-piece transforms spread over taps 2..7 (j = 0, 0, 1, 1, 2, 2)
      if (GNIN && tap >= 2 && tap < 8) {
        // fenced: the transform's temporaries must not overlap the 32 operand-fragment registers of compute()
        __builtin_amdgcn_sched_barrier(0);
        transform_a_half(cc + 1, (tap - 2) >> 1, (tap - 2) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DP == 1 || DP == 3 || DP == 4) { SRGD_TAP8(cc, tap, SRGD_TAP_DMA()); }
      else compute(cc, tap, s);
      if (DP == 2) SRGD_TAP_DMA();
#undef SRGD_TAP_DMA
      if (SRGD_CONV3_DIAG_NODMA == 4 || SRGD_CONV3_DIAG_NODMA == 6) WAIT_VM(6);         // 4 = DMA issued as usual, but (almost) never waited for
      else if (SRGD_CONV3_DIAG_NODMA == 2) WAIT_VM(1);
      else if (SRGD_CONV3_DIAG_NODMA == 3) { if (tap >= 3) WAIT_VM(0); else WAIT_VM(1); }
      else if (tap < 3) WAIT_VM(2); else WAIT_VM(1);
      BARRIER();
    }
  }
  {
    const int cc = CC - 1;
    asm volatile("" : "+v"(opq));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = cc * 9 + tap;
#define SRGD_TAP_DMA() do { if (SRGD_CONV3_DIAG_NODMA != 1 && SRGD_CONV3_DIAG_NODMA != 3 && tap < 7) issue_b(cc, tap + 2); } while (0)
      constexpr int DP = (M16 && (!GNIN || (SRGD_GNIN_LEAN & 2))) ? SRGD_CONV3_DMA_POS : 0;
      if (DP == 0 || DP == 4) SRGD_TAP_DMA();
      if constexpr (DP == 1 || DP == 3 || DP == 4) { SRGD_TAP8(cc, tap, SRGD_TAP_DMA()); }
      else compute(cc, tap, s);
      if (DP == 2) SRGD_TAP_DMA();
#undef SRGD_TAP_DMA
      if (tap < 7) WAIT_VM(1); else WAIT_VM(0);
      if (tap < 8) BARRIER();
    }
  }
#undef issue_a_piece
#undef transform_a_piece
#undef transform_a_half

  // ------------------------------- epilogue -------------------------------------------
  // The accumulator layout (lane = output channel, register = pixel) would store 2 bytes per lane; instead the
  // tile is transposed through LDS ([256 pixels][128 ch] bf16, rows padded to 272 B) and written out as whole
  // 256-byte channel rows, 16 B per lane - 8 store instructions per thread instead of 64.
  constexpr int EROW = BN3 * 2 + 16;
  BARRIER();                                              // every wave is done reading the operand buffers
#if SRGD_CONV3_EPI_PRIO
  // the epilogue's ~250 VALU / LDS instructions per lane share the SIMD's issue port with the co-resident workgroup's MFMA stream
  // (an MFMA holds the port 8 of its 16 cycles): at equal priority the phase crawls (11-12k ticks for ~2k cycles of work) while
  // this workgroup holds half the CU's LDS and registers; raised priority finishes it and gets the next tile started sooner
  __builtin_amdgcn_s_setprio(SRGD_CONV3_EPI_PRIO);
#endif
  if constexpr (M16 && (GNIN || SRGD_CONV3_TIED || SRGD_CONV3_EPI_SCALAR || SRGD_CONV3_DMA_POS == 1 || SRGD_CONV3_DMA_POS == 3 || SRGD_CONV3_DMA_POS == 4)) {
    // asm MFMAs: the compiler does not know the accumulators were written by the matrix pipe and inserts no wait states ahead of
    // their first VALU read (up to 18 for a 16x16 result); the barrier above does not count as one
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
    asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));
  }
  // the epilogue's per-lane addresses are formed from an opaque copy of the thread id: computed here, not ahead of the K
  // loop where they would be carried through it (in registers the GNIN instances do not have, i.e. through scratch)
  int tidE = tid16;                                       // (tid itself is not kept alive through the K loop)
  asm volatile("" : "+v"(tidE));
  tidE >>= 4;
  const int laneE = tidE & 63, r16E = laneE & 15, q16E = laneE >> 4, rE = laneE & 31, hE = laneE >> 5;
  if (p.stamps) t2 = __builtin_amdgcn_s_memtime();
  constexpr int NI = M16 ? 4 : 2;                         // column blocks per wave (16 or 32 wide)
  float s1[NI], s2[NI];
  f32x2 s1p[NI], s2p[NI];                                 // 16x16 path: the column sums as register pairs (even | odd rows)
  constexpr bool pair_writes = M16 && SRGD_CONV3_PAIR_WRITES;
  const bool odd_lane = (r16E & 1) != 0;
  const int pair_off = odd_lane ? 2 * (BN3 * 2 + 16) - 2 : 0;      // (EROW is declared above; rows 2-3, the even channel's column)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    s1[ni] = 0.f;
    s2[ni] = 0.f;
    s1p[ni] = f32x2{0.f, 0.f};
    s2p[ni] = f32x2{0.f, 0.f};
    const int cl = M16 ? wn * 64 + ni * 16 + r16E : wn * 64 + ni * 32 + rE;       // column inside the tile
    const float bias = p.bias ? p.bias[nt * BN3 + cl] : 0.f;
    if constexpr (M16) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const f32x4 av = mi == 0 ? (ni == 0 ? c00 : ni == 1 ? c01 : ni == 2 ? c02 : c03)
                       : mi == 1 ? (ni == 0 ? c10 : ni == 1 ? c11 : ni == 2 ? c12 : c13)
                       : mi == 2 ? (ni == 0 ? c20 : ni == 1 ? c21 : ni == 2 ? c22 : c23)
                                 : (ni == 0 ? c30 : ni == 1 ? c31 : ni == 2 ? c32 : c33);
        // C layout of 16x16: column = laneE & 15, row = (laneE >> 4) * 4 + reg  ->  pixel (mi & 1) * 16 + row of patch row
        char* trow = smem + ((2 * wm + (mi >> 1)) * PW + (mi & 1) * 16 + q16E * 4) * EROW + cl * 2;
        // round 3: the epilogue is ~40 % of the instruction stream of a 36-step tile, so it is written for instruction count:
        // packed fp32 adds / fmas on register pairs (v_pk_add_f32, v_pk_fma_f32), ONE v_cvt_pk_bf16_f32 per two values whose
        // halves go out as ds_write_b16 / ds_write_b16_d16_hi
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#if SRGD_CONV3_EPI_SCALAR
        // Round 4: the same arithmetic in the same order with single-lane-op instructions (inline asm, so that -O3 does not pack
        // them again): a v_pk_add_f32 / v_pk_fma_f32 issued beside the co-resident workgroup's MFMA stream costs several times
        // the two plain ops it replaces (MI355X_MICROARCH.md: packed f32 VALU is an anti-lever beside MFMAs) - round 3's packed
        // form measured faster in instruction count and slower where it mattered.  Bit-identical sums.
        f32x2 v01, v23;
        {
          float w0, w1, w2, w3, e0, e1, a1 = s1p[ni][0], b1 = s1p[ni][1], a2 = s2p[ni][0], b2s = s2p[ni][1];
          asm("v_add_f32 %0, %4, %8\n\tv_add_f32 %1, %5, %8\n\tv_add_f32 %2, %6, %8\n\tv_add_f32 %3, %7, %8"
              : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(bias));
          if (STATS) {
            asm("v_add_f32 %0, %6, %8\n\tv_add_f32 %1, %7, %9\n\t"
                "v_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\t"
                "v_fma_f32 %4, %6, %6, %4\n\tv_fma_f32 %5, %7, %7, %5\n\t"
                "v_fma_f32 %4, %8, %8, %4\n\tv_fma_f32 %5, %9, %9, %5"
                : "=&v"(e0), "=&v"(e1), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2s) : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
            s1p[ni] = f32x2{a1, b1};
            s2p[ni] = f32x2{a2, b2s};
          }
          v01 = f32x2{w0, w1};
          v23 = f32x2{w2, w3};
        }
#else
        const f32x2 b2 = {bias, bias};
        const f32x2 v01 = f32x2{av[0], av[1]} + b2, v23 = f32x2{av[2], av[3]} + b2;
        if (STATS) {
          s1p[ni] += v01 + v23;
          s2p[ni] = __builtin_elementwise_fma(v01, v01, s2p[ni]);
          s2p[ni] = __builtin_elementwise_fma(v23, v23, s2p[ni]);
        }
#endif
        const bf16x2_t t01 = __builtin_convertvector(v01, bf16x2_t), t23 = __builtin_convertvector(v23, bf16x2_t);
        if (pair_writes) {
          // Two adjacent lanes hold two adjacent channels of the same four rows.  They swap halves (one DPP quad_perm move) so
          // that the even lane owns rows 0-1 and the odd lane rows 2-3 of BOTH channels: two conflict-free ds_write_b32 per
          // block instead of four ds_write_b16 whose lane pairs share a dword.  The transposition phase is LDS-write-bound (64
          // two-byte wave-writes per wave beside the co-resident workgroup's operand reads): 10.0k of a 48k-tick tile.
          const unsigned own01 = __builtin_bit_cast(unsigned, t01), own23 = __builtin_bit_cast(unsigned, t23);
          const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_lane ? own01 : own23), 0xB1, 0xf, 0xf, true);
          const unsigned lo_ch = odd_lane ? recv : own01, hi_ch = odd_lane ? own23 : recv;     // channel c (even) | c + 1
          char* prow = trow + pair_off;                 // even lane: rows 0, 1 at its own column; odd lane: rows 2, 3, one column left
          *reinterpret_cast<unsigned*>(prow) = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x05040100u);
          *reinterpret_cast<unsigned*>(prow + EROW) = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x07060302u);
        } else {
          *reinterpret_cast<bf16*>(trow) = t01[0];
          *reinterpret_cast<bf16*>(trow + EROW) = t01[1];
          *reinterpret_cast<bf16*>(trow + 2 * EROW) = t23[0];
          *reinterpret_cast<bf16*>(trow + 3 * EROW) = t23[1];
        }
      }
      if (STATS) {
        s1[ni] = s1p[ni][0] + s1p[ni][1];
        s2[ni] = s2p[ni][0] + s2p[ni][1];
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x16 accv = mi == 0 ? (ni == 0 ? acc00 : acc01) : (ni == 0 ? acc10 : acc11);
        char* trow = smem + ((2 * wm + mi) * PW) * EROW + cl * 2;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int px = (reg & 3) + 8 * (reg >> 2) + 4 * hE;
          const float v = accv[reg] + bias;
          if (STATS) {
            s1[ni] += v;
            s2[ni] += v * v;
          }
          *reinterpret_cast<bf16*>(trow + px * EROW) = (bf16)v;
        }
      }
    }
  }
  // the waves' column sums go to the 4 KiB behind the staged tile, so ONE barrier publishes both; nothing below waits for
  // the output stores (the first version reduced the statistics after the stores, behind a __syncthreads() whose vmcnt(0)
  // drained them: ~6,000 cycles per tile during which the workgroup held its LDS and registers and issued nothing)
  float* const cs = reinterpret_cast<float*>(smem + PH * PW * EROW);       // [4 (wm)][128][2] = 4,096 B; 69,632 + 4,096 = LDS_BYTES
  unsigned long long t2a = 0, t2b = 0;
  if (p.stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t2a = __builtin_amdgcn_s_memtime(); }
  if (STATS) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float t1 = s1[ni], t2 = s2[ni];
      t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 32, 64);
      if (M16) {
        t1 += __shfl_xor(t1, 16, 64);
        t2 += __shfl_xor(t2, 16, 64);
      }
      if (M16 ? laneE < 16 : hE == 0) {
        const int cl = M16 ? wn * 64 + ni * 16 + r16E : wn * 64 + ni * 32 + rE;
        cs[(wm * BN3 + cl) * 2 + 0] = t1;
        cs[(wm * BN3 + cl) * 2 + 1] = t2;
      }
    }
  }
  if (p.stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t2b = __builtin_amdgcn_s_memtime(); }
  __syncthreads();                                        // (no global store is outstanding yet: this does not wait for HBM)
  if (p.stamps) t3 = __builtin_amdgcn_s_memtime();
  {
    // 16-byte chunk q = tidE + 512 i: pixel q / 16 = patch row i (512 threads = one 32-pixel row x 16 chunks), column tidE / 16,
    // channels (tidE % 16) * 8 .. +7: ONE per-lane address and a uniform row stride instead of eight 64-bit address computations
    static_assert(NT3 == PW * 16 && (PH * PW * 16) / NT3 == PH, "store loop: one patch row per iteration");
    const int pxE = tidE >> 4, c16 = tidE & 15;
    bf16* o = p.out + ((size_t)(b * p.H + y0) * p.W + x0 + pxE) * p.Cout + nt * BN3 + c16 * 8;
    const char* src = smem + pxE * EROW + c16 * 16;
    const size_t row_stride = (size_t)p.W * p.Cout;
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i * PW * EROW);
      *reinterpret_cast<bf16x8*>(o + i * row_stride) = v;
    }
  }
  if (p.stamps) t4 = __builtin_amdgcn_s_memtime();
  if (STATS) {
    // per-(sample, group) sums of this tile: columns summed over the 4 row blocks by 128 threads, then a shuffle tree over the
    // group's span of columns (the first version let one thread per group walk its columns serially)
    const int cpg = p.Cout / p.groups;                    // multiple of 16, divides or is a multiple of 128
    const int span = cpg >= BN3 ? BN3 : cpg;              // columns of this tile that belong to one group: 16, 32, 64 or 128
    float a1 = 0.f, a2 = 0.f;
    if (tidE < BN3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a1 += cs[(k * BN3 + tidE) * 2 + 0];
        a2 += cs[(k * BN3 + tidE) * 2 + 1];
      }
      for (int o = 1; o < span && o < 64; o <<= 1) {
        a1 += __shfl_xor(a1, o, 64);
        a2 += __shfl_xor(a2, o, 64);
      }
    }
    if (span == BN3) {                                    // a group spans both waves: combine through LDS (uniform branch);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // raw barriers: the output stores stay in flight
      BARRIER();
      if (tidE < BN3 && laneE == 0) { cs[wave * 2 + 0] = a1; cs[wave * 2 + 1] = a2; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BARRIER();
      if (tidE == 0) { a1 = cs[0] + cs[2]; a2 = cs[1] + cs[3]; }
    }
    if (tidE < BN3 && (tidE % span) == 0) {
      // slot layout: [b][group][m-tile within image (x n-tiles per group when a group spans several)]
      const int tiles_per_group = cpg >= BN3 ? cpg / BN3 : 1;
      const int g = (nt * BN3) / cpg + (cpg >= BN3 ? 0 : tidE / span);
      const int nslots = tiles_y * tiles_x * tiles_per_group;
      const int slot = trem * tiles_per_group + (cpg >= BN3 ? nt % tiles_per_group : 0);
      float* dst = p.gn_partial + ((size_t)(b * p.groups + g) * nslots + slot) * 2;
      dst[0] = a1;
      dst[1] = a2;
    }
  }
  if (p.stamps && tidE == 0) {
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    atomicAdd(&p.stamps[0], t1 - t0); atomicAdd(&p.stamps[1], t2 - t1); atomicAdd(&p.stamps[2], t3 - t2);
    atomicAdd(&p.stamps[3], t4 - t3); atomicAdd(&p.stamps[4], t5 - t4); atomicAdd(&p.stamps[5], t5 - t0);
    atomicAdd(&p.stamps[6], 1ull);
    atomicAdd(&g_conv3_stamps2[0], t2a - t2); atomicAdd(&g_conv3_stamps2[1], t2b - t2a); atomicAdd(&g_conv3_stamps2[2], t3 - t2b);
    atomicAdd(&p.stamps[7], __builtin_amdgcn_s_memrealtime() - r0);      // 100 MHz ticks: in-kernel clock = total / this * 100 MHz
  }
}

}  // namespace

bool conv3x3_bf16_eligible(const ConvArgs& a) {
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.mode != CONV_PLAIN || a.residual || a.gn_res_src) return false;
  if (a.ps0 != a.C0 || (a.C1 && a.ps1 != a.C1)) return false;
  if (a.C0 % KC || a.C1 % KC || a.Cout % BN3 || a.Cout != a.CoutPad) return false;
  if (a.Hin % PH || a.Win % PW) return false;
  if (a.gn_partial) {
    const int cpg = a.Cout / a.groups;
    if (a.Cout % a.groups || cpg % 16) return false;
    if (!(BN3 % cpg == 0 || cpg % BN3 == 0)) return false;
  }
  // per-image byte offsets must fit the 32-bit buffer offset
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.C0, a.C1) * 2 >= (1ull << 31)) return false;
  if ((size_t)9 * ((a.C0 + a.C1) / KC) * (a.Cout / BN3) * B_BYTES >= (1ull << 31)) return false;
  return true;
}

int conv3x3_bf16_stats_slots(const ConvArgs& a) {
  if (a.groups <= 0) return 0;
  const int cpg = a.Cout / a.groups;
  return (a.Hin / PH) * (a.Win / PW) * (cpg >= BN3 ? cpg / BN3 : 1);
}

// Host-side packing: OIHW fp32 -> [tap][cc][ntile][128 rows][64 B swizzled] bf16 (the LDS image of each K-step tile).
bool conv3x3_bf16_m16() {
  static const int m16 = env_int("SRGD_CONV3_M16", CONV3_M16_DEFAULT) != 0;   // tuning knob; the shipped default is CONV3_M16_DEFAULT
  return m16 != 0;
}

void pack_conv3x3_bf16(const float* src_oihw, int Cin, int Cout, std::vector<unsigned short>& out,
                       unsigned short (*to_bf16)(float)) {
  const bool m16 = conv3x3_bf16_m16();
  const int CC = Cin / KC, NTL = Cout / BN3;
  out.assign((size_t)9 * CC * NTL * BN3 * KC, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned short* tile = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * BN3 * KC;
        for (int n = 0; n < BN3; ++n)
          for (int c = 0; c < 4; ++c) {
            const int cs = c ^ (m16 ? (n >> 1) & 3 : (n >> 2) & 3);  // stored chunk position (row_swz of the kernel)
            for (int e = 0; e < 8; ++e) {
              const int ci = cc * KC + c * 8 + e, o = nt * BN3 + n;
              const float v = src_oihw[(((size_t)o * Cin + ci) * 3 + tap / 3) * 3 + tap % 3];
              tile[n * KC + cs * 8 + e] = to_bf16(v);
            }
          }
      }
}

int conv3x3_bf16(const ConvArgs& a, const void* packed_w, const float* gn_in_a, const float* gn_in_b,
                 hipStream_t st) {
  if (!conv3x3_bf16_eligible(a)) SRGD_FAIL("conv3x3_bf16: shape not eligible");
  const bool gnin = gn_in_a != nullptr;
  if (gnin && (a.C1 != 0 || !gn_in_b || gn_in_b < gn_in_a || (size_t)((const char*)gn_in_b - (const char*)gn_in_a) > (1u << 30)))
    SRGD_FAIL("conv3x3_bf16: fused input GroupNorm needs one source and scale/shift arrays in one allocation");
  Conv3Args p;
  p.in0 = (const bf16*)a.in0; p.in1 = (const bf16*)a.in1; p.C0 = a.C0; p.C1 = a.C1;
  p.B = a.B; p.H = a.Hin; p.W = a.Win; p.w = (const bf16*)packed_w; p.bias = a.bias; p.Cout = a.Cout;
  p.out = (bf16*)a.out; p.gn_partial = a.gn_partial; p.groups = a.groups;
  p.gn_in_a = gn_in_a;
  p.gn_in_b_off = gnin ? (int)((const char*)gn_in_b - (const char*)gn_in_a) : 0;
  const int grid = a.B * (a.Hin / PH) * (a.Win / PW) * (a.Cout / BN3);
  {
    // SRGD_CONV3_XCD_PIN: minimum packed-weight bytes of a layer for the n-tile-per-XCD map (0 = never)
    static const long pin_min = (long)env_int("SRGD_CONV3_XCD_PIN_KB", CONV3_XCD_PIN_KB_DEFAULT) * 1024L;
    const int n_tiles = a.Cout / BN3, m_tiles = grid / n_tiles;
    const long wbytes = 9L * (a.C0 + a.C1) * a.Cout * 2;
    p.xcd_pin_ntiles = pin_min > 0 && wbytes >= pin_min && (n_tiles == 2 || n_tiles == 4 || n_tiles == 8) && m_tiles % (8 / n_tiles) == 0;
  }
  static const int want_stamps = env_int("SRGD_CONV3_STAMPS", 0) ? 1 : 0;
  p.stamps = nullptr;
  if (want_stamps) {
    SRGD_HIP(hipGetSymbolAddress((void**)&p.stamps, HIP_SYMBOL(g_conv3_stamps)));
    SRGD_HIP(hipMemsetAsync(p.stamps, 0, sizeof(unsigned long long) * 8, st));
    void* s2p = nullptr;
    SRGD_HIP(hipGetSymbolAddress(&s2p, HIP_SYMBOL(g_conv3_stamps2)));
    SRGD_HIP(hipMemsetAsync(s2p, 0, sizeof(unsigned long long) * 4, st));
  }
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define SRGD_SET(S_, G_, M_)                                                                              \
  SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<S_, G_, M_>),           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024 + (G_ ? COEF_BYTES : 0)));
    SRGD_SET(true, false, false) SRGD_SET(false, false, false) SRGD_SET(true, true, false) SRGD_SET(false, true, false)
    SRGD_SET(true, false, true) SRGD_SET(false, false, true) SRGD_SET(true, true, true) SRGD_SET(false, true, true)
#undef SRGD_SET
    once.done();
  }
  const bool stats = a.gn_partial != nullptr;
  // diagnostic: SRGD_CONV3_ONE_WG=1 pads the LDS request so that only ONE workgroup fits a CU (what a warp-specialised
  // variant with helper waves would have to live with: at 128 VGPRs the register file holds 16 waves per CU either way)
  static const int lds_req = env_int("SRGD_CONV3_ONE_WG", 0) ? 96 * 1024 : LDS_BYTES;
#define SRGD_GO(S_, G_, M_) \
  hipLaunchKernelGGL((conv3x3_bf16_kernel<S_, G_, M_>), dim3(grid), dim3(NT3), lds_req + (G_ ? COEF_BYTES : 0), st, p)
  if (conv3x3_bf16_m16()) {
    if (stats && gnin) SRGD_GO(true, true, true); else if (stats) SRGD_GO(true, false, true);
    else if (gnin) SRGD_GO(false, true, true); else SRGD_GO(false, false, true);
  } else {
    if (stats && gnin) SRGD_GO(true, true, false); else if (stats) SRGD_GO(true, false, false);
    else if (gnin) SRGD_GO(false, true, false); else SRGD_GO(false, false, false);
  }
#undef SRGD_GO
  SRGD_HIP(hipGetLastError());
  if (want_stamps) {                                    // diagnostic mode: synchronous, prints the mean cycles per workgroup and phase
    unsigned long long h[8];
    SRGD_HIP(hipStreamSynchronize(st));
    SRGD_HIP(hipMemcpy(h, p.stamps, sizeof(h), hipMemcpyDeviceToHost));
    const double n = h[6] ? (double)h[6] : 1.0;
    unsigned long long h2[4];
    void* s2p = nullptr;
    SRGD_HIP(hipGetSymbolAddress(&s2p, HIP_SYMBOL(g_conv3_stamps2)));
    SRGD_HIP(hipMemcpy(h2, s2p, sizeof(h2), hipMemcpyDeviceToHost));
    fprintf(stderr, "[conv3x3_bf16 stamps] transposition phase of the stamping wave: accumulators -> LDS %.0f  statistics %.0f  barrier wait %.0f\n",
            h2[0] / n, h2[1] / n, h2[2] / n);
    fprintf(stderr, "[conv3x3_bf16 stamps] C %d+%d -> %d @%dx%d grid %d: prologue %.0f  main %.0f  transpose %.0f  stores %.0f  "
                    "stats %.0f  total %.0f  (s_memtime ticks per workgroup)  in-kernel clock %.3f GHz\n",
            a.C0, a.C1, a.Cout, a.Hin, a.Win, grid, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n,
            h[7] ? 0.1 * (double)h[5] / (double)h[7] : 0.0);
  }
  return 0;
}

}  // namespace srgd
