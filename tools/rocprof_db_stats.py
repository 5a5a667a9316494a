"""Per-kernel summary (calls, total/avg/min/max duration, share) of a rocprofv3 --kernel-trace run whose
output is the rocpd sqlite database (the default output format of this ROCm build).
    python tools/rocprof_db_stats.py gpurun_out/prof/x_results.db profiles/name.csv"""
import re
import sqlite3
import subprocess
import sys


def short(name: str) -> str:
    if name.startswith("_Z"):
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    m = re.match(r"_ZN4srgd12_GLOBAL__N_1(\d+)", name)
    if m:   # cxxfilt does not know the bf16 mangling (DF16b): keep the identifier, tag the element type
        n = int(m.group(1))
        ident = name[m.end():m.end() + n]
        name = "srgd::" + ident + ("<bf16>" if "DF16b" in name else "<f32>" if "IfL" in name or "IfE" in name else "")
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+)(<[^(]*>)?", name)
    base, targs = (m.group(1), m.group(2) or "") if m else (name, "")
    return (base + targs)[:100]


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["name,calls,total_ms,avg_us,min_us,max_us,percent"]
    for r in rows:
        lines.append(f'"{short(r[0])}",{r[1]},{r[2]:.3f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{100 * r[2] / tot:.2f}')
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))
    print("total kernel ms", round(tot, 1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
