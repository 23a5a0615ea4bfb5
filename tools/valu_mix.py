#!/usr/bin/env python3
"""Static vector-ALU instruction mix of one kernel against the issue costs tools/micro/valu_issue.hip measured on gfx950
(profiles/r03_valu_issue.txt): 2 cycles per wave-instruction per SIMD for the plain fp32 / integer add / logic /
move class, 4 for compares, selects on an SGPR mask, min/max, three-operand integer, packed and 64-bit forms, 8 for
the transcendentals.  usage: tools/valu_mix.py <kernel-name-substring> [source.hip]   (compiles to ISA with hipcc)

The mix is static (every instruction of the kernel counted once), so it says what the compiler emitted, not what
the waves execute; it gives the band a kernel's mean cost per counted instruction (SQ_INSTS_VALU) lies in."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
        "v_cndmask_b32_e32"}       # the e32 select reads vcc: 2.8 per instruction in a select+fma pair, so not 4
SLOW8 = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op in FAST or base in FAST:
        return 2
    if base in SLOW8:
        return 8
    return 4


_ISA = {}


def isa_text(src):
    if src not in _ISA:
        with tempfile.TemporaryDirectory() as td:
            s = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math",
                                   "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
                                   "-o", s, src], stderr=subprocess.DEVNULL)
            _ISA[src] = open(s).read()
    return _ISA[src]


def static_mix(pat, src=None):
    """[{name, n, by_cost {cycles: count}, mean_cycles, top}] for every kernel whose demangled name contains `pat`."""
    text = isa_text(src or os.path.join(ROOT, "gigl_amd/csrc/gigl_rt.hip"))
    out = []
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", text, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pat not in name:
            continue
        ops = collections.Counter(re.findall(r"^\s+(v_[a-z0-9_]+)", m.group(0), re.M))
        ops = {k: v for k, v in ops.items() if not re.match(r"v_(accvgpr|mfma|nop)", k)}
        n = sum(ops.values())
        by = collections.Counter()
        for k, v in ops.items():
            by[cost(k)] += v
        out.append({"name": name.replace("(anonymous namespace)::", ""), "n": n, "by_cost": dict(by),
                    "mean_cycles": sum(c * v for c, v in by.items()) / n,
                    "top": sorted(ops.items(), key=lambda kv: -kv[1])[:12]})
    return out


def main():
    for k in static_mix(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None):
        print(k["name"])
        print("   %d vector-ALU instructions: %s; static mean %.2f cycles per instruction per SIMD"
              % (k["n"], ", ".join("%d at %d cycles" % (k["by_cost"][c], c) for c in sorted(k["by_cost"])), k["mean_cycles"]))
        print("   most frequent: " + ", ".join("%s x%d" % kv for kv in k["top"]))


if __name__ == "__main__":
    main()
