"""Hazard lint for the kernels that issue MFMAs (and a SiLU) through inline asm.  The compiler inserts the wait states gfx950 needs
between dependent instructions only for instructions it can see; it does not look inside inline asm.  Round 4 hit both cases on the
hardware: a v_rcp_f32 scheduled directly ahead of a one-instruction asm v_mul_f32 (transcendental results may not be read in the
next issue slot) and VALU-written operands (v_cvt_pk) read by an asm MFMA without the two wait states a VALU -> MFMA dependency needs.
Round 5 added a third: a 16-byte buffer store whose data registers the next packed VALU instruction overwrote (LLVM's hazard rule
exempts stores with an SGPR soffset; on gfx950 lanes 12-15 of every row then stored the NEW register contents).
This test disassembles the MFMA sources (hipcc -S, device only; no GPU needed) and checks every MFMA kernel for
  (1) an MFMA whose A / B / C source registers were written by a VALU instruction less than two wait states earlier
      (`s_nop N` counts N + 1 wait states, any other instruction one),
  (2) a transcendental whose result is read by the very next instruction,
  (3) scratch (spill) traffic inside the innermost loop that contains MFMAs,
  (4) an inline-asm MFMA whose result is read by a non-MFMA instruction less than 18 wait states later (the compiler pads
      only the MFMAs it can see; an MFMA that accumulates into the same registers is interlocked by the hardware),
  (5) a 12- / 16-byte store whose data registers a VALU instruction writes less than two wait states later.
It guards against a future compiler (or edit) re-introducing a hazard silently - the GPU tests would catch wrong numbers, this says why."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "srgd_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "-DNDEBUG", "-S", "--cuda-device-only", "-I", CSRC]
TRANS = re.compile(r"v_(exp|rcp|rsq|log|sqrt|sin|cos)_f(32|16)")


def _regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
    return set(out)


def _kernels(asm):
    """{kernel name: [instruction strings]} for every kernel of the file"""
    out, cur, name = {}, None, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if ln.strip().startswith(".amdhsa_kernel") or ln.strip().startswith(".section"):
            cur = None
            continue
        if "#ASMSTART" in ln:
            cur.append("#ASMSTART")
            continue
        if "#ASMEND" in ln:
            cur.append("#ASMEND")
            continue
        t = ln.split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":"):
            cur.append(t)
        elif t.endswith(":"):
            cur.append(t)                      # labels kept: loop detection
    return out


def _dst_src(ins):
    op, _, rest = ins.partition(" ")
    parts = [a.strip() for a in rest.split(",")] if rest else []
    if op.startswith(("ds_write", "scratch_store", "buffer_store", "global_store", "s_", "v_cmp", ";")) or "lds" in ins:
        return op, set(), set().union(*[_regs(a) for a in parts]) if parts else set()
    return op, (_regs(parts[0]) if parts else set()), (set().union(*[_regs(a) for a in parts[1:]]) if len(parts) > 1 else set())


def _is_valu_write(op):
    return op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp"))


def _wait_states(ins):
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def _lint(ins):
    """findings (strings) for one kernel's instruction list (labels as 'name:', inline-asm regions as #ASMSTART / #ASMEND)"""
    bad = []
    real, in_asm, asm_flag = [], False, []
    for i in ins:
        if i == "#ASMSTART":
            in_asm = True
        elif i == "#ASMEND":
            in_asm = False
        elif not i.endswith(":"):
            real.append(i)
            asm_flag.append(in_asm)
    for k, cur in enumerate(real):
        op, dst, srcs = _dst_src(cur)
        if op.startswith("v_mfma"):
            ws, j = 0, k - 1
            while j >= 0 and ws < 2:                             # two wait states between a VALU write and the MFMA that reads it
                pop, pdst, _ = _dst_src(real[j])
                if _is_valu_write(pop) and pdst & (srcs | dst):
                    bad.append(f"VALU -> MFMA with {ws} wait state(s): {real[j]} | {cur}")
                ws += _wait_states(real[j])
                j -= 1
            if asm_flag[k]:                                      # (4) asm MFMA result -> first non-MFMA reader
                ws = 0
                for j in range(k + 1, min(k + 40, len(real))):
                    nop, ndst, nsrc = _dst_src(real[j])
                    if nop.startswith("v_mfma"):
                        if ndst & dst:
                            break                                # accumulates into the same registers: interlocked
                    elif (nsrc | (ndst if nop.startswith(("ds_write", "buffer_store", "global_store")) else set())) & dst:
                        if ws < 18:
                            bad.append(f"asm MFMA result read after {ws} wait states: {cur} | {real[j]}")
                        break
                    elif ndst & dst:
                        break                                    # overwritten
                    ws += _wait_states(real[j])
                    if ws >= 18:
                        break
        if re.match(r"(buffer|global)_store_dwordx[34]", op):    # (5) store data overwritten too early
            data = _regs(cur.split(",")[0].split(" ", 1)[1]) if op.startswith("buffer") else _regs(cur.split(",")[1])
            ws, j = 0, k + 1
            while j < len(real) and ws < 2:
                nop, ndst, _ = _dst_src(real[j])
                if _is_valu_write(nop) and ndst & data:
                    bad.append(f"store data overwritten after {ws} wait state(s): {cur} | {real[j]}")
                ws += _wait_states(real[j])
                j += 1
        if TRANS.match(op) and k + 1 < len(real):
            nop, ndst, nsrc = _dst_src(real[k + 1])
            if not nop.startswith("s_") and not TRANS.match(nop) and dst & nsrc:
                bad.append(f"transcendental result read in the next slot: {cur} | {real[k + 1]}")
    # no scratch traffic inside an innermost loop that carries MFMAs: between a label and the backward branch to it
    ins = [i for i in ins if not i.startswith("#ASM")]
    labels = {i[:-1]: n for n, i in enumerate(ins) if i.endswith(":")}
    for n, i in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\S+)", i)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            body = ins[labels[m.group(1)]:n]
            if sum(b.startswith("v_mfma") for b in body) >= 16 and any(b.startswith(("scratch_load", "scratch_store")) for b in body):
                bad.append(f"scratch traffic in the MFMA loop at {m.group(1)}")
    return bad


def test_the_lint_sees_the_hazards():
    ok = ["ds_read_b128 v[8:11], v1", "s_waitcnt lgkmcnt(0)", "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"]
    assert _lint(ok) == []
    assert _lint(["v_cvt_pk_bf16_f32 v8, v20, v21", "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"])
    assert _lint(["v_cvt_pk_bf16_f32 v8, v20, v21", "s_nop 1", "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"]) == []
    assert _lint(["v_cvt_pk_bf16_f32 v8, v20, v21", "s_nop 0", "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"])      # one wait state: not enough
    mm = "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"
    assert _lint(["#ASMSTART", mm, "#ASMEND", "v_add_f32 v20, v0, v1"])                                   # asm MFMA result read at once
    assert _lint(["#ASMSTART", mm, "#ASMEND", "s_nop 15", "s_nop 7", "v_add_f32 v20, v0, v1"]) == []
    assert _lint(["#ASMSTART", mm, "#ASMEND", "#ASMSTART", mm, "#ASMEND"]) == []                          # same accumulator: interlocked
    assert _lint([mm, "v_add_f32 v20, v0, v1"]) == []                                                     # builtin MFMA: the compiler pads
    st = "buffer_store_dwordx4 v[20:23], v92, s[0:3], s9 offen"
    assert _lint([st, "v_pk_add_f32 v[20:21], v[14:15], v[10:11]"])                                       # the round-5 failure
    assert _lint([st, "s_nop 1", "v_pk_add_f32 v[20:21], v[14:15], v[10:11]"]) == []
    assert _lint(["global_store_dwordx4 v[0:1], v[20:23], off", "v_mov_b32 v21, v3"])
    assert _lint(["v_rcp_f32_e32 v5, v5", "v_mul_f32 v6, v7, v5"])
    assert _lint(["v_rcp_f32_e32 v5, v5", "v_rcp_f32_e32 v4, v4", "v_mul_f32 v6, v7, v5"]) == []
    loop = [".LBB0_1:"] + ["v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"] * 16 + ["scratch_load_dword v9, off, off", "s_cbranch_scc1 .LBB0_1"]
    assert _lint(loop)


@pytest.mark.parametrize("src", ["conv3x3_bf16.hip", "conv3x3_mxfp8.hip", "conv1x1_mxfp8.hip", "conv1x1_bf16.hip", "linattn_fused.hip",
                                 "linattn_fused256.hip", "attention.hip", "conv_igemm.hip"])
def test_inline_asm_kernels_have_no_unseen_hazards(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "k.s"
    r = subprocess.run([hipcc, *FLAGS, os.path.join(CSRC, src), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = _kernels(out.read_text())
    checked = 0
    for name, ins in kernels.items():
        if not any(i.startswith("v_mfma") for i in ins):
            continue
        checked += 1
        assert _lint(ins) == [], (name[:70], _lint(ins)[:3])
    assert checked >= 2
