#!/usr/bin/env python
"""Development aid: compile lh_kernels.hip with -DLH_MARK and report, for one function of the gfx950
assembly, the number of instructions between consecutive `; LQMARK <name>` comments (first
occurrence of each name; static counts, not executed counts), plus a histogram by class.
usage: tools/isa_sections.py [function-substring] [--keep] [--asm file.s]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deprecated-lame-mirror_amd", "csrc")
OUT = "/tmp/isa_marks"


def main():
    fn = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "lq_outer_loop_stage4"
    os.makedirs(OUT, exist_ok=True)
    asm = None
    if "--asm" in sys.argv:     # an assembly file made elsewhere (hipcc -S --cuda-device-only -DLH_MARK ...)
        asm = sys.argv[sys.argv.index("--asm") + 1]
    cmd = ["true"] if asm else ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fno-slp-vectorize", "-falign-functions=256", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
           "-fPIC", "-I.", "-I../../include", "-DLH_MARK", "-c", "lh_kernels.hip", "-o", OUT + "/k.o", "-save-temps=obj",
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        print(r.stdout[-3000:])
        sys.exit(1)
    for l in r.stdout.splitlines():
        if "lh_encode_kernel" in l or ("VGPRs" in l and "523" in l) or "Spill" in l and "52" in l:
            pass
    s = open(asm or OUT + "/lh_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read().splitlines()
    start = None
    for i, l in enumerate(s):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(fn), l):
            start = i
            break
    if start is None:
        print("function not found")
        sys.exit(1)
    end = start
    while not s[end].startswith(".Lfunc_end"):
        end += 1
    body = s[start:end]
    stats = [l for l in s[end:end + 40] if re.search(r"; (NumVgprs|NumSgprs|ScratchSize|SGPRs Spill|VGPRs Spill|codeLenInByte)", l)]
    print("\n".join(x.strip() for x in stats[:8]))
    seen = {}
    cur = "<entry>"
    counts = collections.OrderedDict()
    hist = collections.defaultdict(collections.Counter)
    for l in body:
        m = re.search(r"; LQMARK (\S+)", l)
        if m:
            name = m.group(1)
            n = seen.get(name, 0)
            seen[name] = n + 1
            cur = name if n == 0 else "%s#%d" % (name, n + 1)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        counts[cur] = counts.get(cur, 0) + 1
        cls = ("spill" if op in ("v_readlane_b32", "v_writelane_b32") else "nop" if op == "s_nop" else
               "wait" if op == "s_waitcnt" else "branch" if op.startswith(("s_cbranch", "s_branch")) else
               "lds" if op.startswith("ds_") else "mem" if op.startswith(("global_", "flat_", "scratch_", "s_load", "buffer_")) else
               "salu" if op.startswith("s_") else "valu")
        hist[cur][cls] += 1
    print("%-28s %6s   %s" % ("section", "instrs", "valu salu lds mem branch wait nop spill"))
    tot = 0
    for k, v in counts.items():
        h = hist[k]
        print("%-28s %6d   %4d %4d %3d %3d %4d %4d %3d %4d" % (k, v, h["valu"], h["salu"], h["lds"], h["mem"], h["branch"],
                                                             h["wait"], h["nop"], h["spill"]))
        tot += v
    print("total", tot)


if __name__ == "__main__":
    main()
