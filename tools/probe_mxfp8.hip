// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (operand / scale / result lane maps), run once on the MI355X
// before the MX-fp8 convolution was written:  hipcc --offload-arch=gfx950 -O2 tools/probe_mxfp8.hip -o /tmp/probe && /tmp/probe
// Hypothesis checked with exact small-integer data (every product and sum is exact in fp32):
//   A: lane l holds row (l & 15), K elements 32*(l >> 4) + j, j = 0..31, one e4m3 byte each (8 VGPRs, byte j of the 32)
//   B: lane l holds column (l & 15), same K elements
//   scale: E8M0 byte (2^(s-127)) number `opsel` of the lane's scale VGPR applies to that lane's 32 elements
//   D: lane l, register r = row (l >> 4) * 4 + r, column (l & 15)   (the 16x16 C/D map of every other MFMA)
#include <hip/hip_runtime.h>
#include <cmath>
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
  for (int i = 0; i < 8; ++i) {
    va[i] = reinterpret_cast<const int*>(a + l * 32)[i];
    vb[i] = reinterpret_cast<const int*>(b + l * 32)[i];
  }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 0, 0, OPA, (int)sa[l], OPB, (int)sb[l]);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

// throughput: NI dependent-free MFMAs per wave, 4 accumulators
__global__ void rate_scaled(float* out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x3c3c3c3c ^ (threadIdx.x * 2654435761u); }
  v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void rate_bf16(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + threadIdx.x * 0.01f + i); b[i] = (__bf16)(1.5f - i * 0.1f); }
  v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  if (e == 15 && m == 7) x = NAN;
  return s ? -x : x;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  // exact small values: 0, +-0.5, +-1, +-1.5, +-2, +-3
  const uint8_t vals[] = {0x00, 0x30, 0xb0, 0x38, 0xb8, 0x3c, 0xbc, 0x40, 0xc0, 0x44, 0xc4};
  std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
  std::vector<uint32_t> hsa(64), hsb(64);
  srand(7);
  for (auto& v : ha) v = vals[rand() % 11];
  for (auto& v : hb) v = vals[rand() % 11];
  uint8_t *da, *db; uint32_t *dsa, *dsb; float* dd;
  CK(hipMalloc(&da, ha.size())); CK(hipMalloc(&db, hb.size())); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 4 * 4));
  int fails = 0;
  for (int op = 0; op < 4; ++op) {
    for (int l = 0; l < 64; ++l) {
      uint32_t wa = 0, wb = 0;
      for (int byte = 0; byte < 4; ++byte) {
        wa |= (uint32_t)(124 + (rand() % 7)) << (8 * byte);        // 2^-3 .. 2^3, a different value in every byte
        wb |= (uint32_t)(124 + (rand() % 7)) << (8 * byte);
      }
      hsa[l] = wa; hsb[l] = wb;
    }
    CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    const int opb = (op + 1) & 3;
    switch (op) {
      case 0: hipLaunchKernelGGL((probe<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); break;
      case 1: hipLaunchKernelGGL((probe<1, 2>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); break;
      case 2: hipLaunchKernelGGL((probe<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); break;
      default: hipLaunchKernelGGL((probe<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); break;
    }
    CK(hipDeviceSynchronize());
    std::vector<float> hd(256);
    CK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int row = 0; row < 16; ++row)
      for (int col = 0; col < 16; ++col) {
        double ref = 0;
        for (int kg = 0; kg < 4; ++kg) {
          const int la = kg * 16 + row, lb = kg * 16 + col;
          const double sca = std::ldexp(1.0, (int)((hsa[la] >> (8 * op)) & 255) - 127);
          const double scb = std::ldexp(1.0, (int)((hsb[lb] >> (8 * opb)) & 255) - 127);
          double acc = 0;
          for (int j = 0; j < 32; ++j) acc += (double)e4m3(ha[la * 32 + j]) * (double)e4m3(hb[lb * 32 + j]);
          ref += acc * sca * scb;
        }
        const float got = hd[((row >> 2) * 16 + col) * 4 + (row & 3)];
        maxerr = std::fmax(maxerr, std::fabs(got - ref));
      }
    printf("opsel A=%d B=%d: max |D - ref| = %.3e  %s\n", op, opb, maxerr, maxerr == 0 ? "PASS (layout hypothesis holds exactly)" : "FAIL");
    fails += maxerr != 0;
  }
  // instruction rate: one wave per SIMD on every CU
  float* dout; CK(hipMalloc(&dout, 1024 * 256 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(rate_scaled, dim3(256), dim3(256), 0, 0, dout, iters);
      else hipLaunchKernelGGL(rate_bf16, dim3(256), dim3(256), 0, 0, dout, iters);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 256.0 * 4 * iters * 4 * 2.0 * 16 * 16 * (which == 0 ? 128 : 32);
      if (rep) printf("%s: %.3f ms, %.1f TFLOP/s (register operands, 1 wave/SIMD)\n", which == 0 ? "mfma_scale 16x16x128 e4m3" : "mfma 16x16x32 bf16", ms, flop / ms / 1e9);
    }
  }
  return fails;
}
