// Decode probe for v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950): which K elements does a lane's scale byte apply to?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPA, int OPB>
__global__ void probe(const uint8_t* a, const uint8_t* b, const uint32_t* sa, const uint32_t* sb, float* d) {
  const int l = threadIdx.x;
  v8i va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = reinterpret_cast<const int*>(a + l * 32)[i]; vb[i] = reinterpret_cast<const int*>(b + l * 32)[i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 0, 0, OPA, (int)sa[l], OPB, (int)sb[l]);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static uint8_t *da, *db; static uint32_t *dsa, *dsb; static float* dd;
static std::vector<float> run(const std::vector<uint8_t>& ha, const std::vector<uint8_t>& hb, const std::vector<uint32_t>& sa,
                              const std::vector<uint32_t>& sb, int opa) {
  hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  if (opa == 0) hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  else if (opa == 1) hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  else if (opa == 2) hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  else hipLaunchKernelGGL((probe<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  hipDeviceSynchronize();
  std::vector<float> hd(256);
  hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
  return hd;   // hd[(lane)*4 + reg]
}
static float at(const std::vector<float>& d, int row, int col) { return d[((row >> 2) * 16 + col) * 4 + (row & 3)]; }

int main() {
  CK(hipMalloc(&da, 2048)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 1024));
  std::vector<uint8_t> ones(2048, 0x38), zeros(2048, 0);
  std::vector<uint32_t> unit(64, 0x7f7f7f7f);
  // 1. all ones, unit scales: every D must be 128
  { auto d = run(ones, ones, unit, unit, 0); float mn = 1e9, mx = -1e9; for (float v : d) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("[1] ones x ones, unit scales: min %.1f max %.1f (expect 128)\n", mn, mx); }
  // 2. data layout: A one-hot at (lane L, byte j) = 1.0, B all ones -> which rows light up; B one-hot -> which cols
  printf("[2] A one-hot (lane, byte 0) x B ones: rows with D != 0 (expect row = lane %% 16)\n");
  for (int L = 0; L < 64; L += 7) {
    std::vector<uint8_t> a = zeros; a[L * 32] = 0x38;
    auto d = run(a, ones, unit, unit, 0);
    printf("    lane %2d:", L);
    for (int row = 0; row < 16; ++row) if (at(d, row, 0) != 0) printf(" row %d (=%.1f)", row, at(d, row, 0));
    printf("\n");
  }
  // 3. K pairing: A one-hot at (lane LA, byte ja); B one-hot at (lane LB, byte jb): product non-zero iff same k
  printf("[3] K pairing A(lane 16g+0, byte j) with B(lane 16g'+0, byte j'): D[0][0] != 0 only for g'=g, j'=j ?\n");
  int bad = 0;
  for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; j += 5) {
    std::vector<uint8_t> a = zeros; a[(16 * g) * 32 + j] = 0x38;
    for (int g2 = 0; g2 < 4; ++g2) for (int j2 = 0; j2 < 32; ++j2) {
      std::vector<uint8_t> b = zeros; b[(16 * g2) * 32 + j2] = 0x38;
      auto d = run(a, b, unit, unit, 0);
      const bool nz = at(d, 0, 0) != 0, want = (g == g2 && j == j2);
      if (nz != want) { if (bad < 10) printf("    A(g%d,j%d) B(g%d,j%d): D=%.2f\n", g, j, g2, j2, at(d, 0, 0)); ++bad; }
    }
  }
  printf("    mismatches: %d\n", bad);
  // 4. scale association: all ones, A-scale of ONE lane (all 4 bytes) = 128 (x2): D[row][0] = 128 + (#K elements scaled)
  printf("[4] A scale of one lane doubled (all bytes), opsel 0: rows changed / by how much (expect row lane%%16: +32)\n");
  for (int L = 0; L < 64; ++L) {
    std::vector<uint32_t> sa = unit; sa[L] = 0x80808080;
    auto d = run(ones, ones, sa, unit, 0);
    printf("    lane %2d:", L);
    for (int row = 0; row < 16; ++row) if (at(d, row, 0) != 128.f) printf(" row %d %+.0f", row, at(d, row, 0) - 128.f);
    printf("\n");
  }
  // 5. which K elements: A one-hot at (lane 16g, byte j), A-scale of lane L2 doubled: does D[0][0] double?
  printf("[5] A element (lane 16g, byte j) is scaled by the scale of which lane? (opsel 0, all 4 bytes set)\n");
  for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; j += 4) {
    std::vector<uint8_t> a = zeros; a[(16 * g) * 32 + j] = 0x38;
    printf("    g%d j%2d:", g, j);
    for (int L2 = 0; L2 < 64; ++L2) {
      std::vector<uint32_t> sa = unit; sa[L2] = 0x80808080;
      auto d = run(a, ones, sa, unit, 0);
      if (at(d, 0, 0) != 1.f) printf(" lane %d (D=%.1f)", L2, at(d, 0, 0));
    }
    printf("\n");
  }
  // 6. opsel: scale word of every lane has byte b = 128, others 127; which opsel value sees it
  printf("[6] byte b of the scale word = 2.0, others 1.0; D[0][0] with all-ones data (256 if selected, 128 if not)\n");
  for (int b = 0; b < 4; ++b) {
    std::vector<uint32_t> sa(64, 0x7f7f7f7f ^ ((0x7fu ^ 0x80u) << (8 * b)));
    printf("    byte %d:", b);
    for (int op = 0; op < 4; ++op) { auto d = run(ones, ones, sa, unit, op); printf(" opsel%d=%.0f", op, at(d, 0, 0)); }
    printf("\n");
  }
  return 0;
}
