"""Build an A/B variant of the engine library with extra compile-time knobs, next to the shipped one:

    python tools/build_variant.py ws2 -DSRGD_CONV3_WSTAG=2      ->  srgd_amd/variants/libsrgd_hip_ws2.so

and select it on the GPU box with `SRGD_HIP_LIB=$PWD/srgd_amd/variants/libsrgd_hip_ws2.so python tools/bench_conv.py ...`
(srgd_amd/_lib.py).  Same sources, same flags as srgd_amd/build.py plus the -D options given; objects go to a scratch
directory, so the shipped library and its stamp are untouched.  The .so files are git-ignored and travel with gpurun."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srgd_amd.build import CSRC, FLAGS, SOURCES, _hipcc  # noqa: E402


def main():
    global CSRC
    name, defs = sys.argv[1], sys.argv[2:]
    CSRC = os.environ.get("SRGD_CSRC", CSRC)          # A/B against an older source tree (e.g. `git worktree add /tmp/base HEAD~1`)
    out_dir = os.path.join(ROOT, "srgd_amd", "variants")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, f"libsrgd_hip_{name}.so")
    hipcc = _hipcc()
    with tempfile.TemporaryDirectory() as tmp:
        def one(src):
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            r = subprocess.run([hipcc, *FLAGS, *defs, "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"{src}:\n{r.stderr}")
            return obj
        with ThreadPoolExecutor(max_workers=6) as ex:
            objs = list(ex.map(one, SOURCES))
        vers = os.path.join(tmp, "exports.map")
        with open(vers, "w") as f:
            f.write("{ global: srgd_*; local: *; };\n")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={vers}", "-o", lib, *objs],
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
    print(lib)


if __name__ == "__main__":
    main()
