"""Board power, core clock and temperature while one kernel family runs back to back: `rocm-smi` is sampled from this (GPU-free)
process around a child that launches one shape of tools/bench_conv.py / bench_conv_fp8.py a few thousand times.
Evidence for "the 3x3 convolutions run at the board's power limit" (DESIGN 4.1 / 4.3).
    python tools/power_trace.py [out.json]"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        row = {}
        for k, v in card.items():
            if "Power" in k:
                row["power_w"] = float(v)
            elif k.startswith("sclk clock speed"):
                row["sclk_mhz"] = float(re.sub(r"[^0-9.]", "", v))
            elif "junction" in k:
                row["junction_c"] = float(v)
        return row
    except Exception as e:                      # noqa: BLE001
        return {"error": str(e)[:80]}


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            self.rows.append(dict(t=time.perf_counter(), **smi_sample()))
            time.sleep(0.1)


CASES = [   # (label, tool, shape filter, extra args, iterations): ~6-8 s of kernel time each
    ("conv3x3_bf16 1024->1024 @32^2", "bench_conv.py", "3x3 1024->1024 @32", ["--impls", "2", "--stats", "1"], 4000),
    ("conv3x3_bf16 128->128 @256^2", "bench_conv.py", "3x3 128->128 @256", ["--impls", "2", "--stats", "1"], 3000),
    ("conv3x3_mxfp8 1024->1024 @32^2 (+ its bf16 twin run)", "bench_conv_fp8.py", "3x3 1024->1024 @32", [], 3000),
    ("conv1x1_bf16 128+128->128 @256^2 (HBM-bound)", "bench_conv.py", "1x1 128+128->128 @256", ["--impls", "3", "--stats", "0"], 5000),
]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    report = {"idle": smi_sample(), "cases": []}
    cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
    m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", cap)
    report["power_cap_w"] = float(m.group(1)) if m else None
    for label, tool, shape, extra, iters in CASES:
        s = Sampler()
        s.start()
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--only", shape, "--batch", "125", "--iters", str(iters), *extra],
                           capture_output=True, text=True)
        t1 = time.perf_counter()
        s.stop = True
        s.join()
        line = [ln for ln in r.stdout.splitlines() if shape in ln]
        # steady state: samples of the last 60 % of the child's life with the core clock up (the first seconds are import + data set-up)
        rows = [x for x in s.rows if "power_w" in x and x["t"] > t0 + 0.4 * (t1 - t0)]
        busy = [x for x in rows if x.get("sclk_mhz", 0) > 500] or rows
        avg = lambda k: round(sum(x[k] for x in busy if k in x) / max(1, sum(1 for x in busy if k in x)), 1)
        report["cases"].append({"case": label, "result": line[-1].strip() if line else r.stdout[-200:], "samples": len(busy),
                                "power_w": avg("power_w"), "power_w_max": max((x["power_w"] for x in busy), default=None),
                                "sclk_mhz": avg("sclk_mhz"), "junction_c": avg("junction_c"), "wall_s": round(t1 - t0, 1)})
        print(json.dumps(report["cases"][-1]), flush=True)
    if out_path:
        json.dump(report, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
