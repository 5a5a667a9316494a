"""HBM traffic per launch from rocprofv3 PMC passes (one counter per pass, as MI355X_MICROARCH.md prescribes):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 5 --warmup 0 \
              --no_cpu_baseline --no_profile --ddpm_steps 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python bench.py ... (same)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db profiles/r1_pmc_traffic.json
Writes, per kernel family, launches and the mean counter value (KB) per launch.  bench.py turns that into bytes:
HBM bytes/launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (FETCH_SIZE counts a wide coalesced read as half)."""
import json
import re
import sqlite3
import sys

FAMILIES = ["conv3x3_mxfp8_kernel", "conv1x1_mxfp8_kernel", "quant_mxfp8_kernel", "conv3x3_split_kernel", "conv1x1_split_kernel", "conv_igemm_split_kernel", "conv3x3_bf16_kernel", "conv1x1_bf16_kernel", "conv_igemm_kernel", "gn_apply_kernel", "la1_t_kernel", "la2_t_kernel", "la1_kernel", "la2_kernel",
            "full_attn_bf16_kernel", "rms_norm_kernel", "final_step_kernel"]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                     "group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, n, avg in rows:
        fam = next((f for f in FAMILIES if f in name or re.search(r"\d+" + f, name)), None)
        if fam is None:
            continue
        cur = out.setdefault(fam, {"launches": 0, "sum": 0.0})
        cur["launches"] += n
        cur["sum"] += avg * n
    return {k: {"launches": v["launches"], "avg_kb": v["sum"] / v["launches"]} for k, v in out.items()}


def main(fetch_db, write_db, out_path, note):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {k: {"FETCH_SIZE": f[k], "WRITE_SIZE": w.get(k)} for k in f}
    res["_note"] = note
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res.items():
        if k != "_note":
            hb = (2 * v["FETCH_SIZE"]["avg_kb"] + (v["WRITE_SIZE"] or {"avg_kb": 0})["avg_kb"]) * 1024
            print(f"{k:26s} launches {v['FETCH_SIZE']['launches']:5d}  fetch {v['FETCH_SIZE']['avg_kb']:12.0f} KB  "
                  f"write {(v['WRITE_SIZE'] or {'avg_kb': 0})['avg_kb']:12.0f} KB  -> {hb / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
