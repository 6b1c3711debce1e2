#!/usr/bin/env python
"""Assemble profiles/r05_* from what tools/measure_r05.sh left under gpurun_out/ (run in the build container after the gpurun
call).  Copies and per-kernel reductions of rocprofv3 / bench.py / tool output only; nothing is typed in by hand."""
import csv, glob, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", "")).replace("mdh::", "")


def stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = {}
    if f:
        for r in csv.DictReader(open(f[0])):
            rows[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]))
    return rows, (f[0] if f else None)


def lines(path):
    return [l for l in open(path) if not re.match(r"^[WE]\d{8}|.*amdgpu\.ids", l)] if os.path.exists(path) else []


# ---- bench line + kernel stats of the same command
bj = os.path.join(G, "r05_bench.json")
if os.path.exists(bj):
    bench = json.loads(open(bj).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(P, "r05_bench_line.json"), "w"), indent=1)
    rf = bench["roofline"]
    json.dump({"source": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) run by bench.py itself over the benchmarked step, this round's measurement pass",
               "traffic_bytes_per_launch_raw": rf["traffic_raw_fetch_plus_write"], "traffic_bytes_per_launch_fetch_x2": rf["traffic"],
               "traffic_source": rf["traffic_source"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
               "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; kernel = k_neighbor_lane (the instance that also labels: lists + CNA) incl. its slice pass"},
              open(os.path.join(P, "r05_traffic.json"), "w"), indent=1)
    bs, src = stats(os.path.join(G, "r05_bench"))
    if src:
        shutil.copy(src, os.path.join(P, "r05_bench_kernel_stats.csv"))
        with open(os.path.join(P, "r05_bench_kernel_stats.md"), "w") as f:
            f.write("# bench.py under `rocprofv3 --kernel-trace --stats` (round 5, one MI355X, 10 061 824-atom FCC Cu, M = 16)\n\n")
            f.write(f"`tools/measure_r05.sh bench`: the default bench line first ({bench['ms_per_step']:.3f} ms/step = {bench['value'] / 1e9:.2f} G atoms/s; `k_neighbor` range by HIP "
                    f"events inside the library {rf['avg_kernel_ms']:.4f} ms -> {rf['achieved']:.0f} GB/s algorithmic = {rf['frac']:.4f} of 8 TB/s; PMC traffic "
                    f"{(rf['traffic'] or 0) / 1e9:.3f} GB per launch, FETCH doubled, vs {rf['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic), then the same command "
                    "(`--no-extra --no-pmc --no-cpu-baseline`) under the kernel trace, whose table follows.\n\n"
                    "The tile kernel (the instance with the fused CNA: it writes the lists and the labels) is launched twice per build (all tiles, then the one-cell slices of the tiles whose halo overflowed LDS: an empty stand-by on this input).\n\n"
                    "| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|\n")
            for k, (c, a, mn, mx) in sorted(bs.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:16]:
                f.write(f"| `{k}` | {c} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |\n")

# ---- atom order: the step on five orders, per kernel
rows = []
for o in ("lattice", "blocks", "shuffled", "poly", "poly_shuffled"):
    d = os.path.join(G, f"r05_order_{o}")
    log = [l for l in lines(os.path.join(d, "run.log")) if l.startswith("order=")]
    st, _ = stats(d)
    if log:
        rows.append((o, log[-1].strip(), st))
if rows:
    with open(os.path.join(P, "r05_order_probe.txt"), "w") as f:
        f.write("# the headline step (ONE C-ABI call, mdh_build_neighbor_fcna: lists and labels in one pass over the tiles; the two-call form of the start of the round: r05a_order_probe.txt) on the same atoms in five orders; tools/measure_r05.sh order = tools/order_probe.py under rocprofv3 --kernel-trace --stats\n"
                "# lattice: the builder's order; blocks: 4096-atom blocks dealt out at random; shuffled: one random permutation; poly: 9.87 M-atom polycrystal in its builder's\n"
                "# order (grain by grain); poly_shuffled: the same permuted.  Kernel averages include the three warm-up steps (the first call with new positions still takes\n"
                "# the spatial-order paths: the hints of csrc/runtime.hip order_hint are sampled on it)\n\n")
        for o, line, st in rows:
            f.write(line + "\n")
            for k in ("k_assign<false, 4>", "k_assign<false, 1>", "k_scan_onepass<true, 32>", "k_scatter", "k_sort_cells", "k_gather", "k_gather_records",
                      "lane::k_neighbor_lane<false, false, false, true, true, 4>", "lane::k_neighbor_lane<false, false, false, false, true, 4>",
                      "k_fcna_f32<false, false>", "k_fcna_f32<false, true>", "k_fcna<false, true>", "k_pack_positions"):
                if k in st:
                    f.write(f"    {k:64s} calls {st[k][0]:4d}  avg {st[k][1] / 1e3:9.1f} us\n")
            f.write("\n")

# ---- the same step as two calls (lattice, shuffled): kernel split
rows2 = []
for o in ("lattice", "shuffled"):
    d = os.path.join(G, f"r05_two_{o}")
    log = [l for l in lines(os.path.join(d, "run.log")) if l.startswith("order=")]
    st, _ = stats(d)
    if log:
        rows2.append((o, log[-1].strip(), st))
if rows2:
    with open(os.path.join(P, "r05_two_calls_probe.txt"), "w") as f:
        f.write("# the headline step as the TWO calls it was until the labels moved into the tile kernel (PROBE_TWO_CALLS=1 tools/order_probe.py), same box and pass as r05_order_probe.txt\n\n")
        for o, line, st in rows2:
            f.write(line + "\n")
            for k in ("k_assign<false, 4>", "k_assign<false, 1>", "k_scan_onepass<true, 32>", "k_scatter", "k_sort_cells", "k_gather", "k_gather_records",
                      "lane::k_neighbor_lane<false, false, false, false, true, 4>", "k_fcna_f32<false, false>", "k_fcna_f32<false, true>", "k_fcna<false, true>", "k_pack_positions"):
                if k in st:
                    f.write(f"    {k:64s} calls {st[k][0]:4d}  avg {st[k][1] / 1e3:9.1f} us\n")
            f.write("\n")

# ---- plain copies
for src, dst in (("r05_order_sweep.txt", "r05_order_sweep.txt"), ("r05_strong.txt", "r05_strong.txt"), ("r05_halo_cost.txt", "r05_halo_cost.txt"),
                 ("r05_weak.txt", "r05_weak_timeline.txt"), ("r05_lane_tiles.txt", "r05_lane_tiles.txt"), ("r05_fuzz_parity.txt", "r05_fuzz_parity.txt"),
                 ("r05_fuzz_system.txt", "r05_fuzz_system.txt"), ("r05_fuzz_twin.txt", "r05_fuzz_twin.txt"), ("r05_twin_probe.txt", "r05_twin_probe.txt"), ("r05_fused_ab.txt", "r05_fused_ab.txt")):
    if os.path.exists(os.path.join(G, src)):
        open(os.path.join(P, dst), "w").writelines(lines(os.path.join(G, src)))

# ---- the analyses of configs 2 and 4 under the kernel trace
an, src = stats(os.path.join(G, "r05_analyses"))
if src:
    shutil.copy(src, os.path.join(P, "r05_analyses_kernel_stats.csv"))
    open(os.path.join(P, "r05_analyses_run.log"), "w").writelines(lines(os.path.join(G, "r05_analyses", "run.log")))
# ---- SQ / traffic counters of the headline step's kernels
merged = {}
for tag in ("r05_sq1", "r05_sq2", "r05_fetch", "r05_write"):
    pth = os.path.join(G, f"pmc_{tag}.json")
    if os.path.exists(pth):
        for k, v in json.load(open(pth)).items():
            merged.setdefault(short(k), {}).update({c: val for c, val in v.items()})
merged = {k: v for k, v in merged.items() if not k.startswith("k_warm_")}
kernel_ns = {k: a for k, (c, a, mn, mx) in stats(os.path.join(G, "r05_order_lattice"))[0].items()}
if merged:
    bl = os.path.join(P, "r05_bench_line.json")
    kms = json.load(open(bl))["kernels_ms"] if os.path.exists(bl) else {}
    for k, v in merged.items():
        d = {}
        if "SQ_INSTS_VALU" in v and v.get("SQ_WAVES"):
            d["valu_instr_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
            d["salu_instr_per_wave"] = v.get("SQ_INSTS_SALU", 0.0) / v["SQ_WAVES"]
            d["lds_instr_per_wave"] = v.get("SQ_INSTS_LDS", 0.0) / v["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in v and k in kernel_ns:
            # VALU-busy fraction of a SIMD: SQ_ACTIVE_INST_VALU (units of 4 cycles, summed over the chip) over 1024 SIMDs x the kernel's
            # duration (the kernel-trace average of the bench command, 2.4 GHz)
            d["kernel_us"] = kernel_ns[k] / 1e3
            d["valu_busy_fraction_per_simd"] = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (kernel_ns[k] * 2.4)
        if v.get("SQ_WAVE_CYCLES"):
            d["wave_cycles_waiting_fraction"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            d["hbm_bytes_fetch_x2_plus_write"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            d["hbm_bytes_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        v["_derived"] = d
    merged["_note"] = ("per dispatch means, tools/measure_r05.sh counters = tools/order_probe.py lattice 136 5 (the headline step, 10 061 824 atoms) under four separate rocprofv3 "
                       "--pmc passes (with --kernel-trace only); SQ_* summed over the chip; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU in units of 4 cycles per SIMD; FETCH_SIZE / WRITE_SIZE in KB "
                       "(FETCH doubled in hbm_bytes_fetch_x2_plus_write as MI355X_MICROARCH.md prescribes for gfx950)")
    json.dump(merged, open(os.path.join(P, "r05_step_counters.json"), "w"), indent=1)
print("profiles/r05_*:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(P, "r05_*"))))
