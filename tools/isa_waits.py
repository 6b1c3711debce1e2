#!/usr/bin/env python
"""Loads that are waited for right behind their issue, kernel by kernel, in the ISA of one HIP source file.

    python tools/isa_waits.py mdapy_amd/csrc/csp.hip [min_hits]

hipcc does not hoist a load over control flow: a loop body "gather neighbour j; fold it into the minimum image" becomes K
dependent memory latencies (global_load ... s_waitcnt vmcnt(1) ... 120 instructions ... global_load ...), and a load behind
`if (n > 0)` with an else branch that fills in constants is followed by register copies behind a wait.  The counters call such a
kernel latency-bound; the listing shows why.  For every kernel: instruction count, global loads, and the `s_waitcnt vmcnt(N)`
that follow a load within three instructions as (instruction index, N) — regular distances between them are the signature.
Compiles with the flags of mdapy_amd/csrc/Makefile (gfx950, -ffp-contract=off); nothing is run."""
import os
import re
import subprocess
import sys
import tempfile


def kernels(asm_path):
    name, buf = None, []
    for line in open(asm_path):
        if line.startswith("_Z") and "@" in line and ":" in line:
            name, buf = line.split(":")[0], []
        elif name and line.startswith("\t") and not line.startswith("\t."):
            buf.append(line)
        if name and "s_endpgm" in line:
            yield name, buf
            name, buf = None, []


def main():
    src = sys.argv[1]
    min_hits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S",
                        "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        for name, buf in kernels(out):
            hits, last = [], -99
            for i, line in enumerate(buf):
                if "global_load" in line or "buffer_load" in line:
                    last = i
                m = re.search(r"s_waitcnt vmcnt\((\d+)\)", line)
                if m and i - last <= 3:
                    hits.append((i, int(m.group(1))))
            if len(hits) >= min_hits:
                loads = sum("global_load" in line for line in buf)
                print(f"{name[:80]:80s} instr {len(buf):6d} loads {loads:4d} early waits {len(hits):3d} {hits[:10]}")


if __name__ == "__main__":
    main()
