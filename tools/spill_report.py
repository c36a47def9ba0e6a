"""Where the register spills of a kernel sit: compiles csrc/scg_kernels.hip to ISA and counts spill / reload
instructions (scratch_* "Folded Spill/Reload", v_writelane / v_readlane SGPR spills) per basic block, separately for the basic blocks
that contain MFMAs (the hot loops of E, U1, U2). No GPU needed.   Usage: python tools/spill_report.py [kernel-symbol-substring ...]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "skill-chaining-with-graphs_amd", "csrc", "scg_kernels.hip")
out = os.path.join(tempfile.gettempdir(), "scg_spill_report.s")
flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp".split()
subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
want = sys.argv[1:] or ["td_kernelILi0E", "td_kernelILi1E", "reduce_kernel"]
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not any(w in name for w in want):
        continue
    blocks, cur = [], []
    for l in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur); cur = []
        else:
            cur.append(l)
    blocks.append(cur)
    is_spill = lambda x: ("Folded Spill" in x or "Folded Reload" in x or re.search(r"v_(writelane|readlane)_b32", x) is not None)
    hot = [b for b in blocks if any("v_mfma" in x for x in b)]
    n_all = sum(sum(1 for x in b if is_spill(x)) for b in blocks)
    n_hot = sum(sum(1 for x in b if is_spill(x)) for b in hot)
    print(f"{name[:40]:40s} basic blocks {len(blocks):4d} (with MFMAs: {len(hot):2d})   spill/reload/lane-move instructions: {n_all:4d} in all, {n_hot} inside blocks with MFMAs")
