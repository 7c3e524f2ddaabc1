"""Static instruction mix of a kernel from the gfx950 assembly hipcc emits: by class, for the whole kernel and for
its hottest loop (the innermost backward-branch region with the most vector instructions).
python scripts/isa_mix.py [source.hip] [kernel-name-substring]
default: render_kernels.hip raster_scan_kernelILi0ELb0ELi1ELi4ELb0E (OBB, no trace, mode 1 = the dense frames' mid-round exit, 4 samples, no depth: the
headline frame's instantiation). Compiled with the Makefile's flags; instruction classes priced as measured on the chip
(profiles/r4_micro/valu_issue.txt): f32 fma / mul / add / mov / logic / integer add 2 clocks, min / max / compare / select /
convert / shift / packed 4, transcendental 8."""
import collections
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bevy_gaussian_splatting_amd", "csrc")
src = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(CSRC, "render_kernels.hip")
want = sys.argv[2] if len(sys.argv) > 2 else "raster_scan_kernelILi0ELb0ELi1ELi4ELb0E"
asm = "/tmp/isa_mix.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                "-fPIC", "--offload-device-only", "-S", src, "-o", asm], check=True, capture_output=True, cwd=CSRC)
lines = open(asm).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", l))
# (a kernel may hold several s_endpgm — early returns —: its end is the function-end label)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]

TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


SLOW = ("v_max", "v_min", "v_med3", "v_cndmask", "v_cvt", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_readlane", "v_writelane",
        "v_readfirstlane", "v_mbcnt", "v_bfi", "v_perm")


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(SLOW):
        return "valu_slow"
    if op.startswith(TRANS):
        return "valu_transcendental"
    if op.startswith("v_") and ("_f64" in op):
        return "valu_fp64"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "valu_compare"
    if op.startswith("v_"):
        return "valu_plain"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def mix(seq):
    c = collections.Counter()
    for l in seq:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c[classify(t.split()[0])] += 1
    return c


# loops: a backward branch to a label defines [label, branch]; take the innermost ones
labels = {l.strip()[:-1]: i for i, l in enumerate(body) if re.match(r"^\.?\w+:\s*(;.*)?$", l.strip()) and l.strip().split(":")[0]}
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\w+):", l.strip())
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def nvec(lp):
    return sum(v for k, v in mix(body[lp[0]:lp[1] + 1]).items() if k.startswith("valu"))


# "hot loop" candidates: loops whose nested loops hold less than a quarter of their vector instructions (a skip loop
# inside the record loop does not make the record loop an outer loop)
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] and 4 * nvec(o) >= nvec(lp) for o in loops)]
whole = mix(body)
valu_keys = [k for k in whole if k.startswith("valu")]
print(f"kernel {want}: {sum(whole.values())} instructions, {sum(whole[k] for k in valu_keys)} vector")
for k, v in sorted(whole.items(), key=lambda kv: -kv[1]):
    print(f"  {k:22s} {v}")
best = max(inner, key=lambda lp: sum(v for k, v in mix(body[lp[0]:lp[1] + 1]).items() if k.startswith("valu")), default=None)
if best:
    m = mix(body[best[0]:best[1] + 1])
    nv = sum(v for k, v in m.items() if k.startswith("valu"))
    clocks = 2 * m["valu_plain"] + 4 * (m["valu_compare"] + m["valu_slow"] + m["valu_packed"] + m["valu_fp64"]) + 8 * m["valu_transcendental"]
    print(f"hottest innermost loop: lines {best[0]}..{best[1]} of the kernel, {sum(m.values())} instructions, {nv} vector")
    for k, v in sorted(m.items(), key=lambda kv: -kv[1]):
        print(f"  {k:22s} {v}")
    print(f"  issue clocks per iteration at 2 (plain), 4 (compare / min / max / select / convert / packed / fp64), 8 (transcendental): {clocks} "
          f"= {clocks / max(nv, 1):.2f} per vector instruction")
