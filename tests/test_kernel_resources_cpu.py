"""No kernel of the bf16 / fp8 hot path may spill: a spilled VGPR in an LDS-DMA kernel is a scratch reload behind
`s_waitcnt vmcnt(0)`, i.e. behind every DMA in flight.  hipcc cross-compiles without a GPU, so the
register / spill / scratch table of every hot-path source is checked here, in the CPU suite, and DESIGN.md cannot drift from
it again (round-4 review: la1_kernel was documented as spill-free while it carried 2 spilled registers)."""
import os
import shutil
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.kernel_resources import kernel_table  # noqa: E402

CSRC = os.path.join(ROOT, "srgd_amd", "csrc")
# source -> kernels of the throughput modes (bf16, fp8, fp8_mixed) and their VGPR ceilings (registers per lane that still give
# the occupancy the kernel is designed for: 128 -> 4 waves per SIMD, 256 -> 2)
HOT = {
    "conv3x3_bf16.hip": {"conv3x3_bf16_kernel": 128},
    "conv3x3_mxfp8.hip": {"conv3x3_mxfp8_kernel": 256},
    "conv3x3_split.hip": {"conv3x3_split_kernel": 256},          # f16x3 mode: one 512-thread workgroup per CU, two waves per SIMD
    "conv3x3_mx2.hip": {"conv3x3_mx2_kernel": 256},              # f16mx2 prototype mode: the same occupancy
    "conv_igemm.hip": {"conv_igemm_split_kernel": 256, "conv_igemm_kernel<float, 16": 128},      # fp32 mode's instance: four workgroups per CU
    "conv1x1_bf16.hip": {"conv1x1_bf16_kernel": 128},
    "conv1x1_mxfp8.hip": {"conv1x1_mxfp8_kernel": 256},
    "linattn_fused.hip": {"la1_kernel": 256, "la2_kernel": 256},
    "linattn_fused256.hip": {"la1_t_kernel": 256, "la2_t_kernel": 256},
    "norm_act.hip": {"gn_apply_kernel": 128, "gn_finalize_kernel": 128, "rms_norm_kernel": 128},
    "quant_mxfp8.hip": {"quant_mxfp8_kernel": 128},
    "attention.hip": {"full_attn_bf16_kernel": 128, "la_combine_kernel": 128},
    "sampler.hip": {"final_step_kernel": 128, "init_gather_kernel": 128},
}


@pytest.fixture(scope="module")
def tables():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda f: kernel_table(os.path.join(CSRC, f)), HOT))
    return dict(zip(HOT, res))


def test_hot_path_kernels_do_not_spill(tables):
    seen = 0
    for src, kernels in HOT.items():
        for key, cap in kernels.items():
            rows = [r for r in tables[src] if key in r["name"]]
            assert rows, f"{src}: no kernel named *{key}* in the resource table"
            for r in rows:
                seen += 1
                assert r["spill"] == 0 and r["scratch"] == 0, f"{src}: {r['name']} spills ({r['spill']} VGPRs, {r['scratch']} B scratch)"
                assert 0 < r["vgpr"] + max(r["agpr"], 0) <= cap, f"{src}: {r['name']} uses {r['vgpr']} + {r['agpr']} registers (> {cap})"
    assert seen >= 20
