// Probe of DPP row operations on gfx950 (which lane feeds which), run once before the fragment-shift experiment in conv3x3_bf16:
//   hipcc --offload-arch=gfx950 -O2 tools/probe_dpp.hip -o tools/probe_dpp.bin && tools/probe_dpp.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL, bool BOUND>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  const int src = 100 + l, old = 900 + l;
  out[l] = __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, BOUND);
}

template <int CTRL, bool BOUND>
static void run(const char* name) {
  int* d; hipMalloc(&d, 64 * 4);
  hipLaunchKernelGGL((k<CTRL, BOUND>), dim3(1), dim3(64), 0, 0, d);
  int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  printf("%-28s", name);
  for (int i = 0; i < 20; ++i) printf(" %d", h[i]);
  printf(" ... lane31=%d lane32=%d\n", h[31], h[32]);
  hipFree(d);
}

int main() {
  run<0x101, false>("row_shl:1 bound_ctrl=0");
  run<0x101, true>("row_shl:1 bound_ctrl=1");
  run<0x111, false>("row_shr:1 bound_ctrl=0");
  run<0x121, false>("row_ror:1");
  run<0x12F, false>("row_ror:15");
  run<0x140, false>("row_mirror");
  run<0x141, false>("row_half_mirror");
  return 0;
}
