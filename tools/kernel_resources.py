"""Per-kernel register / spill / scratch / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), gfx950.
    python tools/kernel_resources.py srgd_amd/csrc/conv3x3_bf16.hip [filter-substring] [-DNAME=VALUE ...]
Used for the spill checks DESIGN.md quotes (a kernel with `spill > 0` or `scratch > 0` runs its inner loop through memory)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
defs = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "-DNDEBUG", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", *defs]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", ln) or re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                          text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>6} {'scratch':>8} {'SGPR':>5} {'occ':>4} {'LDS':>7}  kernel")
for r, name in zip(rows, demangle):
    if flt and flt not in name:
        continue
    name = re.sub(r"srgd::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', r.get('VGPR Spill', '?')):>6} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8} {r.get('TotalSGPRs', '?'):>5} {r.get('Occupancy [waves/SIMD]', '?'):>4} "
          f"{r.get('LDS Size [bytes/block]', '?'):>7}  {name[:110]}")
