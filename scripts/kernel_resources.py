"""Per-kernel register / LDS / occupancy table of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage).
python scripts/kernel_resources.py [-Dflag ...] [file.hip ...]   (default: every csrc/*.hip; the Makefile's flags)"""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bevy_gaussian_splatting_amd", "csrc")
extra = [a for a in sys.argv[1:] if a.startswith("-")]
files = [os.path.abspath(a) for a in sys.argv[1:] if not a.startswith("-")] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
for f in files:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", *extra, "-c", f,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=CSRC)
    cur = None
    rows = []
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print(os.path.basename(f))
    for r in rows:
        print("  %-90s vgpr %3d  sgpr %3d  occ %d  lds %6d  scratch %4d  spill v%d s%d" % (
            r["name"][:90], r.get("vgpr", -1), r.get("sgpr", -1), r.get("occ", -1), r.get("lds", -1), r.get("scratch", -1),
            r.get("vspill", -1), r.get("sspill", -1)))
