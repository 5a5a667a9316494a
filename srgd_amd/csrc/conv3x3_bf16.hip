// 3x3 / stride 1 / pad 1 convolution, bf16 in - fp32 accumulate - bf16 out, for gfx950 (MI355X).
// This is the kernel that carries 89 % of the FLOPs of a ConditionalSRUnet evaluation
// (reference: Block.proj model.py:246, the last-stage 3x3 "resample" convs :647,:668).
//
// Implicit GEMM, M = output pixels, N = Cout, K = 9 * Cin, laid out for CDNA4:
//   * workgroup = 512 threads = 8 waves (4 along M x 2 along N), output tile = an 8 x 32 pixel patch
//     (M = 256) x 128 output channels; each wave owns 2 patch rows x 64 channels = 4 x 4 blocks of
//     v_mfma_f32_16x16x32_bf16 (the 32x32x16 shape measured 1-2 % slower on every production shape: lower sustained clock).
//   * K is walked channel-chunk-major: for every 32-channel chunk the (8+2) x (32+2) halo patch is
//     staged ONCE in LDS and all 9 taps are served from it by shifting the read address - global->LDS
//     traffic for A drops 9x/1.33 versus gathering a fresh A tile per tap.
//   * all staging is LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip): the A patch
//     (out-of-image halo pixels are zero-filled by the buffer descriptor's range check), and per K-step one
//     8 KB weight tile, pre-swizzled on the host into its LDS image so the copy is linear and coalesced.
//     A is double-buffered, B runs in a 3-deep ring; loads stay in flight across barriers
//     (counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per K-step).
//   * 64-byte LDS rows are XOR-swizzled (chunk ^= (row >> 1) & 3): ds_read_b128 is conflict-free for the
//     MFMA operand pattern (16 consecutive rows x 4 chunks) at every tap shift.
//   * the MFMAs take the WEIGHT fragment as their A operand and the PIXEL fragment as B (both fragments have the same register
//     image, so the K loop does not change): D then holds, per lane, four consecutive weight rows of ONE pixel.  The host
//     packs the weight rows of a tile in the order that makes a lane's sixteen values of a 16-pixel block sixteen consecutive
//     output channels (common.hpp, regepi_row_channel) - the epilogue is register-direct: + bias, bf16 pack, GroupNorm partial sums
//     (reference Block.norm, model.py:250-259) by in-lane adds + a DPP row reduction (+ permlane swaps across rows), one slot per
//     wave, no barrier (round 4's workgroup-wide LDS transposition was 23 % of a 128 -> 128 @256^2 tile).  The packed values take
//     one wave-private hop through the idle halo-patch buffer (2 ds_write_b128 + 2 ds_read_b128 per block) so that every store
//     instruction writes 8 full 128-byte lines instead of 64 scattered 16-byte pieces: +0.9 % on the benchmark, same box
//     (profiles/r6/conv3x3_full_line_stores_ab.txt).
//   * blockIdx is remapped so each XCD (private L2) gets a contiguous band of tiles (halo reuse in L2).
//   * GNIN instances (template parameter): the PRODUCER's GroupNorm-apply + SiLU is applied to a chunk's halo patch in LDS right
//     after it lands (reference Block.forward model.py:250-259 between two convolutions), which removes a full HBM pass; these
//     instances issue their MFMAs as inline asm with the accumulator tied (no register migration, no spills).
// What bounds it (in-kernel stamps): the K loop of the deep layers uses 99 % of the MFMA issue slots at a
// power-limited 1.7 GHz.  Diagnostic builds (tools/build_variant.py only): -DSRGD_CONV3_STAMPS=1 adds per-phase s_memtime stamps.
#include <cstdlib>

#include "kernels.hpp"

#ifndef SRGD_CONV3_STAMPS
#define SRGD_CONV3_STAMPS 0
#endif
#if SRGD_CONV3_STAMPS
#include "stamps.hpp"
#endif
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
constexpr bool STAMPS = SRGD_CONV3_STAMPS != 0;
static_assert(8 * 16 * 144 <= A_BYTES, "store staging fits the idle A buffer");

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, int voffset, int soffset = 0) {
  // LDS destination = wave-uniform base + lane * 16; voffset per lane (VGPR), soffset wave-uniform (SGPR)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds_wave_base, 16, voffset, soffset, 0, 0);
}

struct Conv3Args {
  const bf16* in0; const bf16* in1; int C0, C1;
  int B, H, W;
  const bf16* w;          // packed [tap][cc][ntile][128 rows (permuted, see pack_conv3x3_bf16)][4 swizzled chunks][8]
  const float* bias;
  int Cout;
  bf16* out;
  float* gn_partial; int groups;
  const float* gn_in_a;   // GNIN: y = silu(a[b][c] * x + b[b][c]) applied to the input while it is staged ([B][Cin] fp32)
  int gn_in_b_off;        // byte offset of the shift array from the scale array (same allocation)
};

#if SRGD_CONV3_STAMPS
__device__ unsigned long long g_conv3_timeline[(size_t)STAMP_REC * STAMP_MAX_WAVES];       // stamps.hpp
#endif

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// Raw barrier (no vmcnt drain: LDS-DMA prefetches stay in flight) fenced for the instruction scheduler:
// s_barrier is IntrNoMem to LLVM, so without sched_barrier(0) the machine scheduler hoists the next step's
// ds_reads above it - a read of a buffer whose DMA other waves have not yet waited for.
#define BARRIER()                        \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// Row swizzle: chunk ^= (row >> 1) & 3 for the 16x16x32 operand pattern (16 rows x 4 chunks per ds_read_b128) - conflict-free
// for its lane groups at every tap shift (checked exhaustively on the bank model).
__device__ __forceinline__ int row_swz(int row) { return (row >> 1) & 3; }

template <bool STATS, bool GNIN>
__global__ __launch_bounds__(NT3, 4) void conv3x3_bf16_kernel(Conv3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, q16 = lane >> 4;        // fragment row / 8-channel chunk

  // ---- tile coordinates (XCD-aware remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous band)
  const int n_tiles = p.Cout / BN3;
  const int tiles_x = p.W / PW, tiles_y = p.H / PH;
  const int m_tiles = p.B * tiles_y * tiles_x;
  const int nwg = m_tiles * n_tiles;
  int wg = blockIdx.x;
  // Tile map, measured with the L2's own counters (round 5, profiles/r5/conv3x3_bf16_tcc_*.txt; 1024 -> 1024 @32^2, 125 tiles):
  // this map - n-tiles fastest inside an XCD's band, so the 64 workgroups an XCD runs at a time are 8 m-tiles x 8 n-tiles - reads
  // 112.7 M 128-byte lines per launch with an 81 % L2 hit rate (23.4 M misses = 3.0 GB from the Infinity Cache).  Pinning one
  // n-tile per XCD makes the weight stream L2-resident and every XCD read every halo patch: 21.4 M misses, the same clock and
  // throughput.  Blocks of 32 m-tiles x 2 n-tiles cut the misses to 17.1 M (-27 %): clock 1.653 vs 1.648 GHz, +0.4 % (noise).
  // Non-temporal halo DMAs and output stores: 29-36 M misses, 1.55-1.59 GHz, -4 ... -12 %.  The kernel's clock does not follow
  // its traffic beyond L2 within what a tile map can change, so the simplest map stays.
  const int q = nwg >> 3, rem = nwg & 7, x = wg & 7, k = wg >> 3;
  wg = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
  const int nt = wg % n_tiles, mt = wg / n_tiles;
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
  // from row_swz(P) (a multiple of 8 under (P >> 1) & 3) - ONE register, not three.
#define K_A_DECL(J)                                                        \
  int a_pix##J;                                                               \
  {                                                                           \
    const int g = (wave + 8 * J) * 64 + lane; /* 16-byte chunk in the image */ \
    const int P = g >> 2;                                                     \
    const int py = P / WP, px = P - py * WP;                                  \
    const int y = y0 + py - 1, x = x0 + px - 1;                               \
    const bool ok = P < HP * WP && y >= 0 && y < p.H && x >= 0 && x < p.W;    \
    a_pix##J = ok ? y * p.W + x : -1;                                         \
  }
  K_A_DECL(0) K_A_DECL(1) K_A_DECL(2)
#undef K_A_DECL
  const int a_sub = (lane & 3) ^ row_swz(lane >> 2);
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
  // The tap loop of the GNIN instances is branch-free: a per-lane `if (a_pix < 0) return` or a wave-uniform branch around the
  // coefficient DMA splits the unrolled tap loop into basic blocks and the instance then spills into the K loop.  The DMA
  // offsets are precomputed (no multiply / select in the loop: the compiler if-converts those into exec-masked branches too),
  // and the chunk's coefficients (lane l: scale[c + l] or shift[c + l - 32]) come in through ONE 4-byte-per-lane LDS-DMA per
  // wave into a 256-byte slot of their own (every wave issues it: identical bytes, and the counted vmcnt waits stay the same
  // for all waves).  No VGPR load: the compiler would guard its use with s_waitcnt vmcnt(0) - it cannot see the counted
  // waits - and drain every DMA in flight.
  int tid16 = tid * 16;                         // the weight DMA's per-lane offset; GNIN: lane * 16, lane * 4 and (after the loop) tid are derived from it
  int opq = 0;                                  // opaque zero, refreshed once per channel chunk (see the operand addresses below)
  char* const sCoef = smem + LDS_BYTES;
  // the per-lane offset is rebuilt from tid16 at its one use per chunk (3 VALU) rather than carried through the K loop
  auto coef_dma = [&](int cc) {
    int t = tid16;
    asm volatile("" : "+v"(t));
    const int l4 = (t >> 2) & 0xfc;               // lane * 4
    const int voff = l4 < 128 ? l4 : p.gn_in_b_off + l4 - 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (lds_ptr)(sCoef + (cc & 1) * 256), 4, voff, cc * KC * 4, 0, 0);
  };
  // The transform shares the SIMD's vector-issue port with the MFMAs (an MFMA 16x16x32 holds it for 8 of its 16
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
#define K_SILU2(Y0_, Y1_, E0_, E1_)                                                                                           \
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
    K_SILU2(y[0], y[1], 0, 1);
    K_SILU2(y[2], y[3], 2, 3);
#undef K_SILU2
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const unsigned o0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{y[0], y[1]}, bf16x2_t));
    const unsigned o1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{y[2], y[3]}, bf16x2_t));
    const unsigned long long bits64 = (unsigned long long)o0 | ((unsigned long long)o1 << 32);
    const unsigned long long m = j == 0 ? in_m0 : (j == 1 ? in_m1 : in_m2);
    // Precondition: full 64-lane waves in uniform control flow (512-thread workgroups, called from the unrolled tap loop only) -
    // the incoming exec mask is saved and restored around the masked store.
    unsigned long long saved_exec;
    asm volatile("s_and_saveexec_b64 %0, %3\n\tds_write_b64 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved_exec) : "v"(qa), "v"(bits64), "s"(m) : "memory", "scc");
  };
#undef lane16
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
  // one scalar multiply-add.
  const int w_tap_stride = (int)(CC * w_tile_stride);
  auto issue_b = [&](int cc, int tap) {
    if (tap >= 9) { tap -= 9; cc += 1; }
    dma16(rsw, sB0 + (tap % 3) * B_BYTES + wave * 1024, tid16, tap * w_tap_stride + cc * (int)w_tile_stride);
  };

  // ---- accumulators: 64 fp32 per lane, [pixel block mi][weight-row block ni]
  f32x4 c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0,
        c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;

  // ---- operand read addresses.  `opq` is an opaque zero refreshed once per channel chunk: it stops the compiler
  // from hoisting the per-tap addresses out of the K loop into ~18 long-lived VGPRs (the kernel lives at the
  // 128-VGPR cap of 2 workgroups per CU).
  // Weight row n = wn*64 + j*16 + r16, chunk q16: the swizzle of row n does not depend on j or wn - ONE per-lane base, the
  // row block rides in the ds_read offset field.
  const int b_base = (wn * 64 + r16) * 64 + ((q16 ^ row_swz(r16)) << 4);
  auto b_addr = [&](int j) { return b_base + j * 16 * 64; };
  // Halo pixel P = lp + Pc with lp = 2 wm WP + r16 (per lane) and Pc a compile-time constant per (tap, block):
  // P * 64 splits into lp * 64 (one per-lane base) + Pc * 64 (the ds_read's immediate offset), and the swizzle term
  // ((P >> 1) & 3) << 4 = ((P << 3) & 0x30) comes from lp8 = lp << 3: THREE VALU instructions per fragment address (add3 with
  // the opaque zero, and-xor, add), which the compiler shares inside a chunk.
  const int lp = 2 * wm * WP + r16, lp8 = lp << 3, lp64 = lp * 64, q16s = q16 << 4;
  auto a_addr = [&](int tap, int i) {        // i = 16-pixel block of the wave (0..3): patch row i >> 1, x half i & 1
    const int dy = tap / 3, dx = tap - dy * 3;
    const int Pc = ((i >> 1) + dy) * WP + (i & 1) * 16 + dx;
    return lp64 + (((lp8 + Pc * 8 + opq) & 0x30) ^ q16s) + Pc * 64;
  };
  // One tap: 4 pixel fragments x 4 weight fragments -> 16 MFMAs.  Operand order: srcA = weight fragment, srcB = pixel fragment
  // (D[i][j]: i = weight row of the block, j = pixel of the block; lane l holds rows 4 (l >> 4) .. + 3 of pixel l & 15).
  // GNIN instances: inline asm with the accumulator TIED (D = C).  Through the builtin the compiler gives every MFMA a fresh
  // destination, the 64 accumulators migrate through the register file, and at the 128-register cap that fragmentation spilled
  // loop invariants into the K loop - scratch reloads behind s_waitcnt vmcnt(0), i.e. behind every DMA in flight.  (Hazards:
  // operands come from ds_read, whose waits the compiler inserts for asm operands too; the accumulators are first read by VALU
  // code after the loop, behind the s_nop block ahead of the epilogue.)
  auto compute = [&](int cc, int tap) {
    const char* A = sA0 + (cc & 1) * A_BYTES;
    const char* Bt = sB0 + (tap % 3) * B_BYTES;          // (cc * 9 + tap) % 3
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 0));
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 1));
    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 2));
    const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(A + a_addr(tap, 3));
    const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(0));
    const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(1));
    const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(2));
    const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(Bt + b_addr(3));
#define MM(C_, PX_, WT_)                                                                                    \
  do {                                                                                                      \
    if constexpr (GNIN) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C_) : "v"(WT_), "v"(PX_)); \
    else C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WT_, PX_, C_, 0, 0, 0);                              \
  } while (0)
    MM(c00, a0, b0); MM(c01, a0, b1); MM(c02, a0, b2); MM(c03, a0, b3);
    MM(c10, a1, b0); MM(c11, a1, b1); MM(c12, a1, b2); MM(c13, a1, b3);
    MM(c20, a2, b0); MM(c21, a2, b1); MM(c22, a2, b2); MM(c23, a2, b3);
    MM(c30, a3, b0); MM(c31, a3, b1); MM(c32, a3, b2); MM(c33, a3, b3);
#undef MM
  };

  [[maybe_unused]] unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, r0 = 0;
  if constexpr (STAMPS) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
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
  if constexpr (STAMPS) t1 = __builtin_amdgcn_s_memtime();

  // ---- main loop.  Per K-step: [issue A piece of the next chunk (taps 0..2)] [issue B[s+2]] compute(s)
  //      wait until B[s+1] (and, in order, everything older) has landed, barrier.
  for (int cc = 0; cc < CC - 1; ++cc) {
    asm volatile("" : "+v"(opq));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // GNIN: the next chunk's coefficients are the OLDEST request of tap 0 (so the tap's counted wait covers them), published
      // by tap 0's barrier, read from tap 2 on
      if (GNIN && tap == 0) coef_dma(cc + 1);
      if (tap < 3) issue_a_piece(cc + 1, tap);
      issue_b(cc, tap + 2);                      // always a valid step here (cc < CC - 1)
      // a wave rewrites only the pieces it DMA'd itself: piece issued at tap t has landed after the wait of tap t+1;
      // six half-piece transforms spread over taps 2..7 (j = 0, 0, 1, 1, 2, 2)
      if (GNIN && tap >= 2 && tap < 8) {
        // fenced: the transform's temporaries must not overlap the 32 operand-fragment registers of compute()
        __builtin_amdgcn_sched_barrier(0);
        transform_a_half(cc + 1, (tap - 2) >> 1, (tap - 2) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      compute(cc, tap);
      if (tap < 3) WAIT_VM(2); else WAIT_VM(1);
      BARRIER();
    }
  }
  {
    const int cc = CC - 1;
    asm volatile("" : "+v"(opq));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 7) issue_b(cc, tap + 2);
      compute(cc, tap);
      if (tap < 7) WAIT_VM(1); else WAIT_VM(0);
      if (tap < 8) BARRIER();
    }
  }
#undef issue_a_piece
#undef transform_a_piece
#undef transform_a_half

  if constexpr (STAMPS) t2 = __builtin_amdgcn_s_memtime();
  // ------------------------------- epilogue (register-direct) --------------------------
  // Accumulator block (mi, J), register e of lane (r16, g) = pixel (patch row 2 wm + (mi >> 1), x = 16 (mi & 1) + r16), tile row
  // 64 wn + 16 J + 4 g + e = output channel 64 wn + 16 g + 4 J + e (the host's row order, common.hpp: regepi_row_channel): the
  // sixteen registers of a pixel block are 16 consecutive channels - two 16-byte stores, the four lanes of a pixel fill 128
  // contiguous bytes.  No barrier: every wave leaves on its own (the staging rows below are wave-private).
  if constexpr (GNIN) {
    // asm MFMAs: the compiler does not know the accumulators were written by the matrix pipe and inserts no wait states ahead of
    // their first VALU read (up to 18 for a 16x16 result)
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
    asm volatile("" : "+v"(c20), "+v"(c21), "+v"(c22), "+v"(c23), "+v"(c30), "+v"(c31), "+v"(c32), "+v"(c33));
  }
  // the epilogue's per-lane addresses are formed from an opaque copy of the thread id: computed here, not ahead of the K
  // loop where they would be carried through it
  int tidE = tid16;
  asm volatile("" : "+v"(tidE));
  const int laneE = (tidE >> 4) & 63, r16E = laneE & 15, q16E = laneE >> 4;
  const int chw = nt * BN3 + wn * 64;                     // first output channel of the wave (uniform)
  const int chl = q16E * 16;                              // the lane's 16-channel run
  f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0, bs2 = bs0, bs3 = bs0;
  if (p.bias) {
    const float* bp = p.bias + chw + chl;                 // 64-byte aligned (host: the array is 16-byte aligned)
    bs0 = *reinterpret_cast<const f32x4*>(bp);
    bs1 = *reinterpret_cast<const f32x4*>(bp + 4);
    bs2 = *reinterpret_cast<const f32x4*>(bp + 8);
    bs3 = *reinterpret_cast<const f32x4*>(bp + 12);
  }
  // output descriptor: base of the wave's first pixel / channel, two patch rows in range
  const u32x4 rso = make_raw_rsrc(p.out + ((size_t)(b * p.H + y0 + 2 * wm) * p.W + x0) * p.Cout + chw, (unsigned)(2 * p.W * p.Cout * 2));
  const int o_voff = (r16E * p.Cout + chl) * 2;
  // Stores leave as FULL 128-byte lines (the wave's 64 channels of a pixel): each 16-pixel block goes through 16 staging rows of the
  // wave (144-byte pitch) in the A buffer the last chunk does NOT use - every wave is past the barriers that followed its last read,
  // so no barrier here; LDS executes a wave's instructions in order, the fences are for the optimiser (conv1x1_split.hip).  Straight
  // from the accumulators a store instruction writes 64 scattered 16-byte pieces (conv3x3_mxfp8.hip: +1-4 % on the mid-size layers).
  constexpr int STG_ROW = 144;
  char* const stg = smem + (CC & 1) * A_BYTES + wave * (16 * STG_ROW);
  const int stg_w = r16E * STG_ROW + q16E * 32;
  const int stg_r = (laneE >> 3) * STG_ROW + (laneE & 7) * 16;
  const int line_off = (laneE >> 3) * p.Cout * 2 + (laneE & 7) * 16;
  f32x4 s1v = {0.f, 0.f, 0.f, 0.f}, s2v = s1v;            // GroupNorm sums, per register position
  // all four bias vectors are waited for here: behind the first asm store the compiler (which cannot count it) would wait with
  // vmcnt(0) for the remaining ones - and so for that store's completion
  asm volatile("" : "+v"(bs0), "+v"(bs1), "+v"(bs2), "+v"(bs3));
#define K_EMIT(MI, C0_, C1_, C2_, C3_)                                                          \
  do {                                                                                             \
    const int so_ = ((((MI) >> 1) * p.W + ((MI) & 1) * 16) * p.Cout) * 2;                          \
    const f32x4 v0 = C0_ + bs0, v1 = C1_ + bs1, v2 = C2_ + bs2, v3 = C3_ + bs3;                    \
    if (STATS) {                                                                                   \
      s1v += (v0 + v1) + (v2 + v3);                                                                \
      s2v = __builtin_elementwise_fma(v0, v0, s2v);                                                \
      s2v = __builtin_elementwise_fma(v1, v1, s2v);                                                \
      s2v = __builtin_elementwise_fma(v2, v2, s2v);                                                \
      s2v = __builtin_elementwise_fma(v3, v3, s2v);                                                \
    }                                                                                              \
    const u32x4 lo_ = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])}; \
    const u32x4 hi_ = {pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3]), pack_bf16x2(v3[0], v3[1]), pack_bf16x2(v3[2], v3[3])}; \
    {                                                                                              \
      *reinterpret_cast<u32x4*>(stg + stg_w) = lo_;                                                \
      *reinterpret_cast<u32x4*>(stg + stg_w + 16) = hi_;                                           \
      asm volatile("" ::: "memory");                                                               \
      const u32x4 w0_ = *reinterpret_cast<const u32x4*>(stg + stg_r);                              \
      const u32x4 w1_ = *reinterpret_cast<const u32x4*>(stg + stg_r + 8 * STG_ROW);                \
      asm volatile("" ::: "memory");                                                               \
      buffer_store16(w0_, rso, line_off, so_);                                                     \
      buffer_store16(w1_, rso, line_off, so_ + 8 * p.Cout * 2);                                    \
    }                                                                                              \
  } while (0)
  K_EMIT(0, c00, c01, c02, c03);
  K_EMIT(1, c10, c11, c12, c13);
  K_EMIT(2, c20, c21, c22, c23);
  K_EMIT(3, c30, c31, c32, c33);
#undef K_EMIT
  if (STATS) {
    // Per-(sample, group) sums of this wave's 64 pixels x 64 channels: in-lane over the register positions, over the 16 pixels
    // of a row by DPP, then over the rows that share a group (fixed order: deterministic).  cpg = channels per group:
    //   16: one group per lane row (g) -> four groups, written by lanes 0, 16, 32, 48;   32: row pairs -> lanes 0 and 32;
    //   >= 64: the whole wave -> lane 0.
    // Slot layout: [b][group][(m-tile, n-tile of the group) x contributing waves]: 4 waves (wm) of the group's column half, or
    // all 8 when a group spans the whole 128-channel tile; gn_finalize sums the slots in index order (fp64).
    const int cpg = p.Cout / p.groups;                    // 16, 32, 64 or a multiple of 128
    float a1 = row16_sum((s1v[0] + s1v[1]) + (s1v[2] + s1v[3]));
    float a2 = row16_sum((s2v[0] + s2v[1]) + (s2v[2] + s2v[3]));
    if (cpg >= 32) { a1 = xor16_sum(a1); a2 = xor16_sum(a2); }
    if (cpg >= 64) { a1 = xor32_sum(a1); a2 = xor32_sum(a2); }
    const int rows_per_group = cpg >= 64 ? 4 : cpg >> 4;  // lane rows (16 channels each) that share a group: 1, 2 or 4
    if (r16E == 0 && (q16E & (rows_per_group - 1)) == 0) {
      const int tpg = cpg >= BN3 ? cpg / BN3 : 1;         // 128-channel tiles per group
      const int wpt = cpg >= BN3 ? 8 : 4;                 // contributing waves per tile
      const int nslots = tiles_y * tiles_x * tpg * wpt;
      const int slot = (trem * tpg + (cpg >= BN3 ? nt % tpg : 0)) * wpt + (cpg >= BN3 ? wave : wm);
      const int g = cpg >= BN3 ? chw / cpg : (chw + chl) >> __builtin_ctz(cpg);      // cpg < 128: a power of two (eligibility)
      float* dst = p.gn_partial + ((size_t)(b * p.groups + g) * nslots + slot) * 2;
      *reinterpret_cast<f32x2*>(dst) = f32x2{a1, a2};
    }
  }
  if constexpr (STAMPS) {
    t3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t4 = __builtin_amdgcn_s_memtime();
#if SRGD_CONV3_STAMPS
    if (laneE == 0) stamp_record(g_conv3_timeline, blockIdx.x * (NT3 / 64) + wave, r0, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
#endif
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
    if (a.Cout % a.groups) return false;
    if (!(cpg == 16 || cpg == 32 || cpg == 64 || cpg % BN3 == 0)) return false;
  }
  // per-image byte offsets must fit the 32-bit buffer offset
  if ((size_t)a.Hin * a.Win * (size_t)std::max(a.C0, a.C1) * 2 >= (1ull << 31)) return false;
  if ((size_t)a.Hin * a.Win * (size_t)a.Cout * 2 >= (1ull << 31)) return false;
  if ((size_t)9 * ((a.C0 + a.C1) / KC) * (a.Cout / BN3) * B_BYTES >= (1ull << 31)) return false;
  return true;
}

// GroupNorm partial slots per (sample, group): one per contributing wave (see the kernel's epilogue)
int conv3x3_bf16_stats_slots(const ConvArgs& a) {
  if (a.groups <= 0) return 0;
  const int cpg = a.Cout / a.groups;
  return (a.Hin / PH) * (a.Win / PW) * (cpg >= BN3 ? (cpg / BN3) * 8 : 4);
}

// Host-side packing: OIHW fp32 -> [tap][cc][ntile][128 rows][64 B swizzled] bf16 (the LDS image of each K-step tile).
// Row order inside a tile: regepi_row_channel (common.hpp) - the accumulator registers of a lane are 16 consecutive channels.

void pack_conv3x3_bf16(const float* src_oihw, int Cin, int Cout, std::vector<unsigned short>& out,
                       unsigned short (*to_bf16)(float)) {
  const int CC = Cin / KC, NTL = Cout / BN3;
  out.assign((size_t)9 * CC * NTL * BN3 * KC, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int cc = 0; cc < CC; ++cc)
      for (int nt = 0; nt < NTL; ++nt) {
        unsigned short* tile = out.data() + ((size_t)(tap * CC + cc) * NTL + nt) * BN3 * KC;
        for (int n = 0; n < BN3; ++n)
          for (int c = 0; c < 4; ++c) {
            const int cs = c ^ ((n >> 1) & 3);  // stored chunk position (row_swz of the kernel)
            for (int e = 0; e < 8; ++e) {
              const int ci = cc * KC + c * 8 + e, o = nt * BN3 + regepi_row_channel(n);
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
  if (a.bias && ((size_t)a.bias & 15)) SRGD_FAIL("conv3x3_bf16: the bias array must be 16-byte aligned");
  Conv3Args p;
  p.in0 = (const bf16*)a.in0; p.in1 = (const bf16*)a.in1; p.C0 = a.C0; p.C1 = a.C1;
  p.B = a.B; p.H = a.Hin; p.W = a.Win; p.w = (const bf16*)packed_w; p.bias = a.bias; p.Cout = a.Cout;
  p.out = (bf16*)a.out; p.gn_partial = a.gn_partial; p.groups = a.groups;
  p.gn_in_a = gn_in_a;
  p.gn_in_b_off = gnin ? (int)((const char*)gn_in_b - (const char*)gn_in_a) : 0;
  const int grid = a.B * (a.Hin / PH) * (a.Win / PW) * (a.Cout / BN3);
  static bool attr_set[64] = {};
  if (DeviceSetup once(attr_set); once.need) {
#define K_SET(S_, G_)                                                                                  \
  SRGD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<S_, G_>),               \
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + (G_ ? COEF_BYTES : 0)));
    K_SET(true, false) K_SET(false, false) K_SET(true, true) K_SET(false, true)
#undef K_SET
    once.done();
  }
  const bool stats = a.gn_partial != nullptr;
#define K_GO(S_, G_) \
  hipLaunchKernelGGL((conv3x3_bf16_kernel<S_, G_>), dim3(grid), dim3(NT3), LDS_BYTES + (G_ ? COEF_BYTES : 0), st, p)
  if (stats && gnin) K_GO(true, true); else if (stats) K_GO(true, false);
  else if (gnin) K_GO(false, true); else K_GO(false, false);
#undef K_GO
  SRGD_HIP(hipGetLastError());
#if SRGD_CONV3_STAMPS
  {                                                     // stamp build: synchronous, prints the phase means and the slot timeline
    SRGD_HIP(hipStreamSynchronize(st));
    unsigned long long* dtl = nullptr;
    SRGD_HIP(hipGetSymbolAddress((void**)&dtl, HIP_SYMBOL(g_conv3_timeline)));
    const StampSummary r = stamp_summary(dtl, grid, NT3 / 64, 2);
    fprintf(stderr, "[conv3x3_bf16 stamps] C %d+%d -> %d @%dx%d grid %d%s: prologue %.0f  main %.0f  epilogue (wave 0: bias, pack, stores issued, "
                    "statistics) %.0f  store drain %.0f  total %.0f  (s_memtime ticks per workgroup)  in-kernel clock %.3f GHz | launch span %.1f us, "
                    "workgroup %.2f us (wave-exit skew %.2f us), slot gap exit -> next entry %.2f us, slot occupancy %.3f on %zu CUs\n",
            a.C0, a.C1, a.Cout, a.Hin, a.Win, grid, gnin ? " GNIN" : "", r.phase[0], r.phase[1], r.phase[2], r.phase[3],
            r.phase[0] + r.phase[1] + r.phase[2] + r.phase[3], r.clock_ghz, r.span_us, r.wg_us, r.skew_us, r.gap_us, r.occupancy, r.cus);
  }
#endif
  return 0;
}

}  // namespace srgd
