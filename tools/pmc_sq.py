"""Per-kernel SQ / GRBM counter summary of one rocprofv3 PMC pass (rocpd sqlite output):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_sq -o s -- python bench.py ...
    python tools/pmc_sq.py gpurun_out/pmc_sq/s_results.db profiles/r1_pmc_sq.json
Derived per kernel family (MI355X_MICROARCH.md, rocprofv3 PMC slots / DVFS give-back):
  clock_ghz      = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
  wave time split: parked (SQ_WAIT_ANY) / issue-stalled (SQ_WAIT_INST_ANY) / issuing (SQ_ACTIVE_INST_ANY) over SQ_WAVE_CYCLES."""
import json
import sqlite3
import sys

FAMILIES = ["conv3x3_mx2_kernel", "conv3x3_mxfp8_kernel", "conv1x1_mxfp8_kernel", "quant_mxfp8_kernel", "conv3x3_split_kernel", "conv1x1_split_kernel", "conv_igemm_split_kernel", "conv3x3_bf16_kernel", "conv1x1_bf16_kernel", "conv_igemm_kernel", "gn_apply_kernel", "la1_t_kernel", "la2_t_kernel", "la1_kernel", "la2_kernel",
            "full_attn_bf16_kernel", "rms_norm_kernel", "final_step"]


def main(db, out_path):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection "
                     "group by kernel_name, counter_name").fetchall()
    fam = {}
    for name, counter, total, n, dur in rows:
        f = next((x for x in FAMILIES if x in name), None)
        if f is None:
            continue
        d = fam.setdefault(f, {})
        e = d.setdefault(counter, {"sum": 0.0, "launches": 0, "ns": 0})
        e["sum"] += total
        e["launches"] += n
        e["ns"] += dur
    res = {}
    for f, d in fam.items():
        g = d.get("GRBM_GUI_ACTIVE")
        if not g:
            continue
        cycles = g["sum"] / 8.0
        r = {"launches": g["launches"], "total_ms": g["ns"] / 1e6, "clock_ghz": cycles / g["ns"]}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            r["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (1024.0 * cycles)
        wc = d.get("SQ_WAVE_CYCLES", {}).get("sum")
        if wc:
            for k, label in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_issue_stall_frac"),
                             ("SQ_ACTIVE_INST_ANY", "wave_issuing_frac")):
                if k in d:
                    r[label] = d[k]["sum"] / wc
        r["raw_sums"] = {k: v["sum"] for k, v in d.items()}
        res[f] = r
        print(f, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "raw_sums"})
    res["_note"] = ("one rocprofv3 --kernel-trace --pmc pass (SQ + GRBM counters only) of 'python bench.py --steps 5 --warmup 0 "
                    "--no_cpu_baseline --no_profile --ddpm_steps 2'; profiled passes run ~3 % slower clocks than un-profiled ones")
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
