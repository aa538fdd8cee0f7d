#!/usr/bin/env python3
"""Audit of the hand-issued late-argument loads of render_fwd_fused_small<KG, false>
(doodle_amd/csrc/splat_fwd.hip; cdna_hip_programming.md §5.7 item 1): compile the file to gfx950
assembly and check, for every forward-only instantiation, that between the statement that issues the
three s_load instructions and the statement that waits for them

  * no instruction reads or writes a destination register of those loads (hipcc counts an asm
    load's destination as written when the statement ends and may copy or reuse it);
  * hipcc issues no scalar-memory or LDS instruction and no lgkmcnt wait of its own (its counted
    waits do not know about the loads in flight);
  * no destination of the three loads overlaps the base register pair they share;

and that the kernel spills nothing.  Exit status 0 = clean.  Run by the test suite AND by the product build (__graft_entry__.build,
`python -m doodle_amd.build --audit`): another hipcc must not silently produce a bad library."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from doodle_amd import build as hb


def regs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bs(\d+)\b", text))
    return out


def audit(asm: str):
    problems, checked = [], 0
    kernels = re.split(r"\n(?=_ZN5helio22render_fwd_fused_small)", asm)
    for k in kernels:
        m = re.match(r"(_ZN5helio22render_fwd_fused_smallILi(\d)ELb0E\S*):", k)
        if not m:
            continue
        name = f"render_fwd_fused_small<{m.group(2)}, false>"
        body = k[:k.index("s_endpgm")]
        lines = body.split("\n")
        issue = next((i for i, l in enumerate(lines) if "s_load_dwordx16" in l and "0x38" in l), None)
        if issue is None:
            problems.append(f"{name}: the late-argument loads were not found")
            continue
        end_issue = next(i for i in range(issue, len(lines)) if "ASMEND" in lines[i])
        dest, base = set(), set()
        for l in lines[issue:end_issue]:
            ops = l.split(";")[0].split(",")
            dest |= regs(ops[0])
            if len(ops) > 1:
                base |= regs(ops[1])
        if dest & base:
            # the three loads share one base pair: a destination on top of it would be overwritten by the first
            # load's data before the later loads issue ("=&s" in late_issue keeps the allocator from doing that)
            problems.append(f"{name}: a late-argument destination overlaps the kernarg base pair s{sorted(base)}")
        wait = next((i for i in range(end_issue, len(lines)) if "s_waitcnt lgkmcnt(0)" in lines[i] and "ASMSTART" in lines[i - 1]), None)
        if wait is None:
            problems.append(f"{name}: the wait statement was not found")
            continue
        for l in lines[end_issue + 1:wait - 1]:
            code = l.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            if regs(code) & dest:
                problems.append(f"{name}: `{code}` touches a late-argument register before the wait")
            if re.match(r"(s_load|s_buffer_load|s_memtime|s_memrealtime|ds_|s_waitcnt.*lgkmcnt)", code):
                problems.append(f"{name}: `{code}` between issue and wait")
        checked += 1
    meta = re.findall(r"\.name:\s+(_ZN5helio22render_fwd_fused_smallILi\dELb0E\S*)(.*?)\.wavefront_size", asm, re.S)
    for n, block in meta:
        if ".sgpr_spill_count: 0" not in block or ".vgpr_spill_count: 0" not in block or ".private_segment_fixed_size: 0" not in block:
            problems.append(f"{n[:60]}: spills")
    return checked, problems


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "splat_fwd.s")
        flags = [f for f in hb.FLAGS if f not in ("-fPIC", "-shared")]
        subprocess.check_call([hb.hipcc(), *flags, "-I", os.path.join(ROOT, "include"), "-I", hb.CSRC, "--offload-device-only",
                               "-S", "-o", out, os.path.join(hb.CSRC, "splat_fwd.hip")], stderr=subprocess.DEVNULL)
        checked, problems = audit(open(out).read())
    for p in problems:
        print("AUDIT:", p)
    print(f"audited {checked} kernels: {'FAILED' if problems or checked != 3 else 'clean'}")
    return 1 if problems or checked != 3 else 0


if __name__ == "__main__":
    sys.exit(main())
