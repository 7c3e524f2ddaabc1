"""The gfx950 assembly of ONE kernel, directives stripped: python scripts/kernel_asm.py [mangled-name-regex] [out.s] [-Dflag ...]
default: the headline's rasteriser instantiation (OBB, mid-round exit, 4 samples). Prints instruction-class counts per
loop depth marker so that a change in the record loop shows at a glance."""
import collections, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(R, "bevy_gaussian_splatting_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("-D")]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
want = args[0] if args and args[0] else r"_ZN3bgs18raster_scan_kernelILi0ELb0ELi1ELi4ELb0ELb0E"
out = args[1] if len(args) > 1 else "/tmp/kernel.s"
src = "sort_kernels.hip" if "sort" in want or "keygen" in want or "onesweep" in want else "render_kernels.hip"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                "--offload-device-only", *extra, "-S", src, "-o", "/tmp/_all.s"], check=True, cwd=CSRC, capture_output=True)
lines = open("/tmp/_all.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^" + want + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = [l for l in lines[start:end] if not re.match(r"^\s*(\.loc|\.cfi|\.file|$)", l)]
open(out, "w").write("\n".join(body) + "\n")
c = collections.Counter()
for l in body:
    t = l.split()
    if not t or t[0].endswith(":") or t[0].startswith(";"):
        continue
    op = t[0]
    c["v_mov" if op.startswith("v_mov") else "valu" if op.startswith("v_") else "branch" if "branch" in op else
      "scratch" if op.startswith("scratch") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"] += 1
print(out, len(body), dict(c))
