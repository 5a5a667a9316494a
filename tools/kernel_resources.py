"""Per-kernel register / spill / scratch / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), gfx950.
    python tools/kernel_resources.py srgd_amd/csrc/conv3x3_bf16.hip [filter-substring] [-DNAME=VALUE ...]
Used for the spill checks DESIGN.md quotes (a kernel with `spill > 0` or `scratch > 0` runs its inner loop through memory) and by
tests/test_kernel_resources_cpu.py, which asserts that no kernel of the bf16 / fp8 hot path spills."""
import re
import subprocess
import sys


def kernel_table(src, defs=()):
    """[{name, vgpr, agpr, spill, scratch, sgpr, occupancy, lds}] for every kernel in `src` (the build's flags)."""
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "-DNDEBUG", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *defs]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        raise RuntimeError(res.stderr[-4000:])
    rows, cur = [], None
    for ln in res.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", ln) or re.search(r"remark:\s+(.*?) \[-Rpass", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    out = []
    for r, name in zip(rows, names):
        name = re.sub(r"srgd::\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*\)$", "", name)

        def num(*keys):
            for k in keys:
                if k in r:
                    return int(r[k])
            return -1
        out.append(dict(name=name, vgpr=num("VGPRs"), agpr=num("AGPRs"), spill=num("VGPRs Spill", "VGPR Spill"),
                        scratch=num("ScratchSize [bytes/lane]"), sgpr=num("TotalSGPRs"), occupancy=num("Occupancy [waves/SIMD]"),
                        lds=num("LDS Size [bytes/block]")))
    return out


def main():
    src = sys.argv[1]
    flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>6} {'scratch':>8} {'SGPR':>5} {'occ':>4} {'LDS':>7}  kernel")
    for r in kernel_table(src, defs):
        if flt and flt not in r["name"]:
            continue
        print(f"{r['vgpr']:>5} {r['agpr']:>5} {r['spill']:>6} {r['scratch']:>8} {r['sgpr']:>5} {r['occupancy']:>4} {r['lds']:>7}  {r['name'][:110]}")


if __name__ == "__main__":
    main()
