// Issue-rate probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 under the operand patterns conv3x3_mxfp8.hip actually uses
// (round 3).  tools/probe_mxfp8.hip measured 3965 TFLOP/s with ONE pair of operand registers, inline-constant scales and one
// wave per SIMD; the convolution's K loop issues, per pixel fragment, four MFMAs with four DIFFERENT weight fragments, scales
// that live in VGPRs (one per fragment + byte select) and 128 different accumulators, from two waves per SIMD.  Timing-only
// diagnostic builds of the kernel (no LDS reads, no barriers, no DMA waits) stay at 2.3-2.6 PFLOP/s, so this probe asks what
// the instruction itself sustains in that pattern - the number the kernel's roofline fraction should be read against.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mxfp8_rate.hip -o /tmp/probe_rate && /tmp/probe_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define MM(C_, A_, B_, SA_, SB_, OPSEL_)                                                                  \
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " OPSEL_                           \
               : "+v"(C_) : "v"(A_), "v"(B_), "v"(SA_), "v"(SB_))

// MODE 0: one A, one B, scales in VGPRs (same registers every time)            -> cost of VGPR scales
// MODE 1: one A, four B fragments, VGPR scales with the four byte selects      -> the kernel's inner pattern, 16 accumulators
// MODE 2: as 1, two A fragments alternating and 32 accumulators (8 MFMAs per loop trip)
template <int MODE>
__global__ __launch_bounds__(512) void rate(const int* src, float* out, int iters) {
  v8i a0, a1, b0, b1, b2, b3;
  const int t = threadIdx.x;
  for (int i = 0; i < 8; ++i) {
    a0[i] = src[(t * 8 + i) & 4095];
    a1[i] = src[(t * 8 + i + 512) & 4095];
    b0[i] = src[(t * 8 + i + 1024) & 4095];
    b1[i] = src[(t * 8 + i + 1536) & 4095];
    b2[i] = src[(t * 8 + i + 2048) & 4095];
    b3[i] = src[(t * 8 + i + 2560) & 4095];
  }
  int sa0 = 0x7f + (t & 1), sa1 = 0x7e + (t & 1), sb = 0x7f7e7f7e + (t & 1);
  v4f c[8] = {};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      MM(c[0], a0, b0, sa0, sb, "op_sel_hi:[0,0,0]"); MM(c[1], a0, b0, sa0, sb, "op_sel_hi:[0,0,0]");
      MM(c[2], a0, b0, sa0, sb, "op_sel_hi:[0,0,0]"); MM(c[3], a0, b0, sa0, sb, "op_sel_hi:[0,0,0]");
    } else {
      MM(c[0], a0, b0, sa0, sb, "op_sel_hi:[0,0,0]"); MM(c[1], a0, b1, sa0, sb, "op_sel:[0,1,0] op_sel_hi:[0,0,0]");
      MM(c[2], a0, b2, sa0, sb, "op_sel_hi:[0,1,0]"); MM(c[3], a0, b3, sa0, sb, "op_sel:[0,1,0] op_sel_hi:[0,1,0]");
      if (MODE == 2) {
        MM(c[4], a1, b0, sa1, sb, "op_sel_hi:[0,0,0]"); MM(c[5], a1, b1, sa1, sb, "op_sel:[0,1,0] op_sel_hi:[0,0,0]");
        MM(c[6], a1, b2, sa1, sb, "op_sel_hi:[0,1,0]"); MM(c[7], a1, b3, sa1, sb, "op_sel:[0,1,0] op_sel_hi:[0,1,0]");
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
  out[blockIdx.x * blockDim.x + t] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  int* dsrc; float* dout;
  CK(hipMalloc(&dsrc, 4096 * 4)); CK(hipMalloc(&dout, 2048 * 512 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int data = 0; data < 2; ++data) {
    std::vector<int> h(4096);
    srand(5);
    // data 0: every byte 0x38 (= 1.0 in e4m3): no operand toggling; data 1: random e4m3 bytes without NaN / large exponents
    for (auto& v : h) {
      if (!data) { v = 0x38383838; continue; }
      unsigned w = 0;
      for (int k = 0; k < 4; ++k) w |= (unsigned)((rand() & 0x80) | (0x20 + rand() % 0x28)) << (8 * k);
      v = (int)w;
    }
    CK(hipMemcpy(dsrc, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; ++mode)
      for (int wps = 1; wps <= 2; ++wps) {              // waves per SIMD: 256 or 512 threads per workgroup, one workgroup per CU
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256 * wps), 0, 0, dsrc, dout, iters);
          if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256 * wps), 0, 0, dsrc, dout, iters);
          if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256 * wps), 0, 0, dsrc, dout, iters);
          hipEventRecord(e1); CK(hipEventSynchronize(e1));
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep && ms < best) best = ms;
        }
        const double n_mfma = 256.0 * 4 * wps * (double)iters * (mode == 2 ? 8 : 4);
        printf("data %s  mode %d (%s)  %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s\n", data ? "random  " : "constant", mode,
               mode == 0 ? "1 A, 1 B, VGPR scales      " : mode == 1 ? "1 A, 4 B, VGPR scales+opsel" : "2 A, 4 B, 32 accumulators  ",
               wps, best, n_mfma * 2.0 * 16 * 16 * 128 / best / 1e9);
      }
  }
  return 0;
}
