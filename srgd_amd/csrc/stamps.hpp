// Diagnostic (stamp) builds only - tools/build_variant.py -DSRGD_CONV3_STAMPS=1 / -DSRGD_MXFP8_STAMPS=1; never compiled into the
// shipped library's kernels.  Every wave records {hardware id (XCC | HW_ID), s_memrealtime at entry, at exit, four phase lengths in
// s_memtime ticks} with plain stores into its own slots of a device array; the host derives the phase means, the in-kernel
// clock, the wave-exit skew of a workgroup and - per CU - how long a workgroup slot stays empty between one workgroup's exit
// and its successor's entry.  (Rounds 3-5 summed the phases with atomicAdd on eight shared counters: with 32,000 workgroups
// per launch those atomics serialised in one L2 channel, stretched the launch to twice its length and inflated every phase
// of the short-K shapes; the numbers from those builds are superseded by the ones from this header.)
#pragma once
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#include "common.hpp"

namespace srgd {

constexpr int STAMP_REC = 7;                       // u64 per wave
constexpr int STAMP_MAX_WAVES = 1 << 19;

__device__ __forceinline__ void stamp_record(unsigned long long* tl, unsigned wave_index, unsigned long long r0, unsigned long long d0,
                                             unsigned long long d1, unsigned long long d2, unsigned long long d3) {
  if (wave_index >= (unsigned)STAMP_MAX_WAVES) return;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long* t = tl + (size_t)wave_index * STAMP_REC;
  t[0] = ((unsigned long long)(xcc & 15) << 32) | hw; t[1] = r0; t[3] = d0; t[4] = d1; t[5] = d2; t[6] = d3;
  t[2] = __builtin_amdgcn_s_memrealtime();
}

struct StampSummary {
  double phase[4] = {0, 0, 0, 0};                  // mean s_memtime ticks per workgroup (wave 0)
  double clock_ghz = 0, wg_us = 0, skew_us = 0, gap_us = 0, occupancy = 0, span_us = 0;
  size_t cus = 0;
};

// device_tl: the device array; nwg workgroups of `waves` waves, `slots` workgroups resident per CU
inline StampSummary stamp_summary(const unsigned long long* device_tl, int nwg, int waves, int slots) {
  StampSummary r;
  nwg = std::min(nwg, STAMP_MAX_WAVES / waves);
  std::vector<unsigned long long> tl((size_t)nwg * waves * STAMP_REC);
  if (hipMemcpy(tl.data(), device_tl, tl.size() * 8, hipMemcpyDeviceToHost) != hipSuccess || nwg <= 0) return r;
  struct Iv { unsigned long long s, e; };
  std::map<unsigned long long, std::vector<Iv>> cu;             // key: XCC, SE, SH, CU
  double ticks = 0, real = 0;
  unsigned long long lo = ~0ull, hi = 0;
  for (int w = 0; w < nwg; ++w) {
    unsigned long long s0 = ~0ull, e0 = 0, e1 = ~0ull;
    for (int k = 0; k < waves; ++k) {
      const unsigned long long* t = &tl[((size_t)w * waves + k) * STAMP_REC];
      s0 = std::min(s0, t[1]); e0 = std::max(e0, t[2]); e1 = std::min(e1, t[2]);
      if (k == 0) {
        for (int i = 0; i < 4; ++i) { r.phase[i] += (double)t[3 + i]; ticks += (double)t[3 + i]; }
        real += (double)(t[2] - t[1]);
      }
    }
    const unsigned long long id = tl[(size_t)w * waves * STAMP_REC];
    cu[((id >> 32) << 16) | ((unsigned)id & 0xff00)].push_back({s0, e0});       // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    r.skew_us += (double)(e0 - e1); r.wg_us += (double)(e0 - s0);
    lo = std::min(lo, s0); hi = std::max(hi, e0);
  }
  double gap = 0; long ngap = 0;
  for (auto& kv : cu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [](const Iv& a, const Iv& b) { return a.s < b.s; });
    std::vector<unsigned long long> slot_end((size_t)slots, 0ull);
    double busy = 0;
    for (auto& iv : v) {
      size_t sl = 0;
      for (size_t i = 1; i < slot_end.size(); ++i) if (slot_end[i] < slot_end[sl]) sl = i;      // the slot that became free first
      if (slot_end[sl]) { gap += (double)iv.s - (double)slot_end[sl]; ++ngap; }
      slot_end[sl] = iv.e; busy += (double)(iv.e - iv.s);
    }
    r.occupancy += busy / ((double)slots * (double)(hi - lo));
  }
  for (double& p : r.phase) p /= nwg;
  r.clock_ghz = real > 0 ? 0.1 * ticks / real : 0.0;             // s_memrealtime: 100 MHz
  r.wg_us = r.wg_us / nwg * 0.01; r.skew_us = r.skew_us / nwg * 0.01; r.gap_us = ngap ? gap / ngap * 0.01 : 0.0;
  r.occupancy /= (double)cu.size(); r.span_us = (double)(hi - lo) * 0.01; r.cus = cu.size();
  return r;
}

}  // namespace srgd
