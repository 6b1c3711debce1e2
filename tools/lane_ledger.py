#!/usr/bin/env python
"""Instruction ledger of the headline instance of the tile neighbour kernel (k_neighbor_lane<COUNT=0, TRI=0, LOOP=0, FCNA=0, TK8=1, NW=4>):
the gfx950 ISA of mdapy_amd/csrc/neighbor_lane.hip with line tables, every instruction attributed to a phase of the kernel by the
source line it came from (inlined helpers by their own lines), counted by unit.  Static counts; the kernel is straight-line per tile and
per chunk of 64 centres, so a wave that takes one chunk of a tile executes the front phases once and the chunk phases once — the
trip factors that are not 1 are listed with the table.  Nothing is run on a GPU.

    python tools/lane_ledger.py [path/to/lane.s]     (without a path: compiles the file first, ~40 s)"""
import os, re, subprocess, sys, tempfile, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mdapy_amd", "csrc", "neighbor_lane.hip")
KERNEL = "_ZN3mdh4lane15k_neighbor_laneILb0ELb0ELb0ELb0ELb1ELi4EE"

# phases by source line of neighbor_lane.hip (first line of the range); helpers that are inlined carry their own lines
def build_phase_table():
    text = open(SRC).read().splitlines()
    def line_of(needle, start=0):
        for k in range(start, len(text)):
            if needle in text[k]:
                return k + 1
        raise SystemExit(f"marker not found: {needle}")
    kern = line_of("void k_neighbor_lane(")
    marks = [
        (line_of("struct FastDiv {"), "front: tile / halo-cell coordinates, first loads"),
        (line_of("__device__ __forceinline__ int excl_scan_block"), "front: workgroup scan, run table"),
        (line_of("__device__ __forceinline__ int combine_codes"), "front: staging into LDS"),
        (line_of("__device__ __forceinline__ double exact_d2"), "write-out: f64 distance + sqrt"),
        (line_of("#define MDH_CAND2"), "scan: trip loop, 4 slots per trip (COLD on the lattice: runs > 12)"),
        (line_of("__device__ __forceinline__ void scan_run12_asm"), "scan: 9 runs x 12 candidate slots (hand-written)"),
        (line_of("__device__ __forceinline__ int run_slots"), "misc helpers"),
        (line_of("__device__ __forceinline__ bool sqrt_fast_ok"), "write-out: f64 distance + sqrt"),
        (line_of("template <bool SELF, bool TRI>"), "band: f64 re-decision (rare)"),
        (line_of("template <bool TRI, int NN, class Index>"), "fused CNA (not in this instance)"),
        (line_of("__device__ __forceinline__ int bit_select"), "write-out: quad transpose + stores"),
        (kern, "front: tile / halo-cell coordinates, first loads"),
        (line_of("hc(tid) = (unsigned)cnt;", kern), "front: workgroup scan, run table"),
        (line_of("// ---- stage this cell's atoms", kern), "front: staging into LDS"),
        (line_of("__syncthreads(); // publishes the run table", kern), "front: barrier, flags, chunk setup"),
        (line_of("for (int cbase = ", kern), "chunk: centre + run-table reads"),
        (line_of("if (TK8 && short_runs) {", kern), "scan: mask clean-up around the runs"),
        (line_of("{   // the centre itself sits in run 4", kern), "chunk: self bit, centre wrap, band test"),
        (line_of("int hits = 0;", kern), "tickets: popcounts, nn store, masks -> tickets"),
        (line_of("if (TK8 && !COUNT) {", kern), "write-out: ticket decode, LDS position reads"),
        (line_of("bool slow = false;", kern), "write-out: f64 distance + sqrt"),
        (line_of("// ---- the rows go out.", kern), "write-out: quad transpose + stores"),
        (line_of("} else if (!COUNT) {", kern), "wide instance (not in this instance)"),
    ]
    marks.sort()
    return marks

def phase_of(marks, line):
    name = "before the kernel (other helpers)"
    for first, label in marks:
        if line >= first:
            name = label
        else:
            break
    return name

def unit_of(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    return "other"

def main():
    if len(sys.argv) > 1:
        asm = sys.argv[1]
    else:
        tmp = tempfile.mkdtemp()
        asm = os.path.join(tmp, "lane.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-gline-tables-only",
                        "-S", "--cuda-device-only", "-o", asm, SRC], check=True, stderr=subprocess.DEVNULL)
    marks = build_phase_table()
    files = {}
    counts = collections.defaultdict(collections.Counter)
    inside, cur_file, cur_line = False, 0, 0
    this = None
    for raw in open(asm):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', raw)
        if m:
            files[int(m.group(1))] = m.group(3)
            if m.group(3).endswith("neighbor_lane.hip") or m.group(2).endswith("neighbor_lane.hip"):
                this = int(m.group(1)) if this is None else this
            continue
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*neighbor_lane\.hip)"', raw)
        if m and this is None:
            this = int(m.group(1))
            continue
        if raw.startswith(KERNEL) and ":" in raw:
            inside = True
            continue
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", raw)
        if m:
            cur_file, cur_line = int(m.group(1)), int(m.group(2))
            continue
        s = raw.strip()
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        if cur_file == this or files.get(cur_file, "").endswith("neighbor_lane.hip"):
            ph = phase_of(marks, cur_line)
        else:
            ph = "inlined from " + os.path.basename(files.get(cur_file, "?"))
        counts[ph][unit_of(op)] += 1
        if op == "s_endpgm":
            break
    units = ["VALU", "SALU", "LDS", "VMEM", "SMEM", "wait", "branch", "other"]
    tot = collections.Counter()
    print(f"{'phase':62s} " + " ".join(f"{u:>6s}" for u in units))
    order = [label for _, label in sorted(set(marks))]
    seen = []
    for label in order + sorted(k for k in counts if k not in order):
        if label in seen or label not in counts:
            continue
        seen.append(label)
        c = counts[label]
        tot.update(c)
        print(f"{label:62s} " + " ".join(f"{c[u]:6d}" for u in units))
    print(f"{'static total':62s} " + " ".join(f"{tot[u]:6d}" for u in units))
    # the hot path of the headline lattice: what a wave that takes one chunk of a tile executes
    cold = ("COLD", "band:", "not in this instance", "__clang_hip_math", "inlined from grid.hpp", "before the kernel", "misc helpers")
    factor = {"write-out: ticket decode, LDS position reads": 0.75, "write-out: f64 distance + sqrt": 0.75}  # 3 of the 4 unrolled groups of four slots run (12 neighbours)
    hot = 0.0
    print()
    print("hot path of the headline lattice (one chunk of 64 centres per wave and tile; VALU instructions per wave):")
    for label in seen:
        if any(c in label for c in cold):
            continue
        v = counts[label]["VALU"] * factor.get(label, 1.0)
        hot += v
        print(f"  {label:60s} {v:7.0f}" + ("   (x 3/4: three of the four unrolled slot groups)" if label in factor else ""))
    print(f"  {'sum':60s} {hot:7.0f}   (SQ_INSTS_VALU / SQ_WAVES of the same kernel: profiles/r0*_neighbor_sq_counters.json)")

if __name__ == "__main__":
    main()
