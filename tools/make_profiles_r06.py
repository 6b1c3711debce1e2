#!/usr/bin/env python
"""Assemble profiles/r06_* from the raw outputs of tools/measure_r06.sh under gpurun_out/ (run in the build container after the gpurun
call).  Everything written here is a copy or a per-kernel reduction of rocprofv3 / bench.py output; nothing is typed in by hand."""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
HBM = 8000.0  # GB/s


def short(name):
    name = name.replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    return name.replace("mdh::", "")


def stats(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        if "mdh::" in r["Name"]:
            rows[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]))
    return rows


def pmc(tag):
    p = os.path.join(G, f"pmc_{tag}.json")
    return {short(k)[:62]: v for k, v in json.load(open(p)).items()} if os.path.exists(p) else {}


what = sys.argv[1:] or ["analyses", "bench", "texts"]
if "analyses" in what:
    src = os.path.join(G, "r06_analyses")
    shutil.copy(os.path.join(src, "s_kernel_stats.csv"), os.path.join(P, "r06_analyses_kernel_stats.csv"))
    log = [l for l in open(os.path.join(src, "run.log")) if not re.match(r"^[WE]\d{8}", l) and "amdgpu.ids" not in l]
    open(os.path.join(P, "r06_analyses_run.log"), "w").writelines(log)
    an = stats(os.path.join(src, "s_kernel_stats.csv"))
    fe, wr, s1, s2 = pmc("r06_an_fetch"), pmc("r06_an_write"), pmc("r06_an_sq1"), pmc("r06_an_sq2")
    N3, N5 = 10061824, 9841500
    # (kernel, atoms, algorithmic bytes per atom, what the bytes are)
    ALG = [
        ("lane::k_neighbor_lane<false, false, false, false, false, 4, false>", N3, 24 + 4 + 4 * 24, "the cutoff build behind the k-nearest searches (and the rc = 0.85 a / rc = 3.6 lists): positions in, count + ids out (ids only for the searches; 24-32 slots)"),
        ("k_knn_rows<false, 18, 32, false>", N3, 24 + 4 + 4 * 32 + 12 * 18, "k = 18: wrapped positions, count, row of 32 ids in; ids 4k + distances 8k out"),
        ("k_knn_rows<false, 12, 24, false>", N3, 24 + 4 + 4 * 24 + 12 * 12, "k = 12"),
        ("k_knn_rows<false, 14, 28, false>", N3, 24 + 4 + 4 * 28 + 12 * 14, "k = 14"),
        ("k_knn_near<false, 18>", N3, 24 + 12 * 18, "(the cell walk: only what k_knn_rows leaves)"),
        ("ptms::k_ptm_order_faces<false, 8, false, 2, true>", N3, 24 + 72 + 18 + 72, "positions, row 4*18 in; order 18 B + ordered ids 4*18 out"),
        ("ptms::k_ptm_order_faces<false, 10, false, 2, true>", N3, 24 + 72 + 18 + 72, "the same, ten-vertex polygons (the first call of a shape)"),
        ("ptms::k_ptm_hull<false>", N3, 24 + 72 + 2 * (56 + 1), "positions + ordered ids in; 2 hulls x (28 facets x 2 B + status) out (fcc-hcp-bcc)"),
        ("ptms::k_ptm_canon<12, false>", N3, 57 + 8 + 17 + 1, "facets in; hash, labelling, flag out"),
        ("ptms::k_ptm_canon<14, false>", N3, 57 + 8 + 17 + 1, "same, 15-point cluster"),
        ("ptms::k_ptm_match<false, false>", N3, 24 + 72 + 2 * 26 + 64 + 72, "positions, ids, 2 x (hash, labelling, flag) in; (N,8) f64 + (N,18) i32 out"),
        ("k_sq_stage1_pair<false, 4, 6>", N3, 24 + 4 + 12 * 12 + 2 * 16 * (9 + 13), "positions, count, 12 ids + distances in; q_4m and q_6m (9 + 13 x re,im) written"),
        ("k_sq_final<true>", N3, 16 * 2 * 13 + 16, "q_lm rows in (416 B); q4, q6 out"),
        ("k_csp<false, 12>", N3, 24 + 4 * 12 + 8, "positions, 12 ids in; csp out"),
        ("k_acna_f32<false>", N3, 24 + 4 * 14 + 4, "positions, 14 ids in; label out (single-precision pair tests; the double-precision kernel finishes its to-do list)"),
        ("ptms::k_ptm_shell<false, 4, 3>", N3, 24 + 72 + 18 + 17 * 28 + 1, '"all": positions, ordered ids + ranks in; 17-point cluster (ids, points) out'),
        ("ptms::k_ptm_hull_shell", N3, 17 * 24 + 1 + 57, '"all": cluster points in; 28 facets + status out'),
        ("ptms::k_ptm_canon<16, true>", N3, 57 + 8 + 17 + 1, '"all": facets in; hash, labelling, flag out'),
        ("ptms::k_ptm_match<false, true>", N3, 24 + 72 + 3 * 26 + 17 * 28 + 10 * 28 + 64 + 72, '"all": + both clusters in'),
        ("k_rdf_tile<false>", N5, 28, "positions 24 + type 4 (the histogram is 6.4 kB)"),
        ("k_wcp_count", N5, 8 + 4 * 27, "count, type, row of the rc = 3.6 list (width 27)"),
    ]
    with open(os.path.join(P, "r06_analyses_roofline.md"), "w") as f:
        f.write("# Kernels of BASELINE configs 2 and 4 at full size (round 6, one MI355X)\n\n"
                "`tools/profile_analyses.py c3 c5` under `rocprofv3 --kernel-trace --stats` (`r06_analyses_kernel_stats.csv`, wall times in "
                "`r06_analyses_run.log`); FETCH_SIZE / WRITE_SIZE and the SQ counters from four separate `--pmc` passes over the same command "
                "(`tools/measure_r06.sh analyses analyses_counters`, `tools/pmc_any.sh`).  c3 = 136^3 fcc Cu, N(0, 0.05) seed 0, 10 061 824 atoms; "
                "c5 = Cu64Zr36 glass, 9 841 500 atoms.\n"
                "Algorithmic bytes per atom: compulsory unique traffic of that kernel (SURVEY 8d convention: gathers and LDS reuse not counted). "
                "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per dispatch (FETCH doubled as the guide prescribes for wide streaming reads; gathers are "
                "over-counted by that, so read the column as an upper bound). VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / kernel time "
                "(double-precision instructions issue in 5-6 cycles, `profiles/r02_ubench_valu_rate.txt`: a kernel of mostly f64 arithmetic is "
                "issue-bound near 70 % of this scale); waves waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; instr = VALU wave-instructions per wavefront.\n\n"
                "| kernel | avg ms | algorithmic B/atom | algorithmic GB/s | % of 8 TB/s | HBM bytes / algorithmic | VALU busy | waves waiting | VALU instr / wave | bound by |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, n, bpa, txt in ALG:
            if k not in an:
                f.write(f"| `{k}` | (not in this run) | {bpa} | | | | | | | |\n")
                continue
            calls, avg, mn, mx = an[k]
            kk = k[:62]
            ms = avg / 1e6
            gbs = bpa * n / (avg * 1e-9) / 1e9
            hbm = (2 * fe.get(kk, {}).get("FETCH_SIZE", 0) + wr.get(kk, {}).get("WRITE_SIZE", 0)) * 1024
            valu = s1.get(kk, {}).get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (avg * 1e-9 * 2.4e9)
            ipw = s1.get(kk, {}).get("SQ_INSTS_VALU", 0) / max(1.0, s1.get(kk, {}).get("SQ_WAVES", 1))
            wait = s2.get(kk, {}).get("SQ_WAIT_ANY", 0) / max(1.0, s2.get(kk, {}).get("SQ_WAVE_CYCLES", 1))
            bound = "VALU issue" if valu > 0.55 else ("latency (waves parked)" if wait > 0.5 else "mixed: issue + latency")
            f.write(f"| `{k}` | {ms:.2f} | {bpa} ({txt}) | {gbs:.0f} | {100 * gbs / HBM:.1f} | {hbm / (bpa * n):.2f} | {100 * valu:.0f} % | {100 * wait:.0f} % | {ipw:.0f} | {bound} |\n")
        f.write("\nRound 5 -> 6, same input (`r05_analyses_kernel_stats.csv`): `k_ptm_order_faces` 41.0 -> 33.3 ms (polygon cached in registers), "
                "`k_ptm_hull` 34.6 -> 21.8 (duplicate check over the insertion's facets, reference point from registers, pipelined walk), "
                "`k_ptm_match` 25.8 -> 13.7 (correspondence gathered at once, graph hashes bisected), `k_ptm_canon<14>` 16.3 -> 15.7, "
                "`k_ptm_hull_shell` 55.0 -> 34.0, `k_ptm_match<false, true>` 37.7 -> 21.8; the 18-nearest search 9.9 ms of `k_knn_near` -> "
                "`k_neighbor_lane` (ids only) + `k_knn_rows`.  What was tried and lost is in `r06_ptm_experiments.txt`.\n"
                "None of these kernels is near the HBM roofline, and none should be: per atom they do 10^3 - 10^5 operations on a few hundred bytes.\n")
    print(open(os.path.join(P, "r06_analyses_roofline.md")).read()[:6000])


def stats_dir(d):
    import glob
    hits = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    return (stats(hits[0]), hits[0]) if hits else ({}, None)


if "bench" in what:
    bj = os.path.join(G, "r06_bench.json")
    if os.path.exists(bj):
        bench = json.loads(open(bj).read().strip().splitlines()[-1])
        json.dump(bench, open(os.path.join(P, "r06_bench_line.json"), "w"), indent=1)
        rf = bench["roofline"]
        json.dump({"source": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) run by bench.py itself over the benchmarked step, this round's measurement pass",
                   "traffic_bytes_per_launch_raw": rf["traffic_raw_fetch_plus_write"], "traffic_bytes_per_launch_fetch_x2": rf["traffic"],
                   "traffic_source": rf["traffic_source"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                   "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; kernel = k_neighbor_lane (the instance that also labels: lists + CNA) incl. its slice pass"},
                  open(os.path.join(P, "r06_traffic.json"), "w"), indent=1)
        bs, src = stats_dir(os.path.join(G, "r06_bench"))
        if src:
            shutil.copy(src, os.path.join(P, "r06_bench_kernel_stats.csv"))
            with open(os.path.join(P, "r06_bench_kernel_stats.md"), "w") as f:
                f.write("# bench.py under `rocprofv3 --kernel-trace --stats` (round 6, one MI355X, 10 061 824-atom FCC Cu, M = 16)\n\n")
                f.write(f"`tools/measure_r06.sh bench`: the default bench line first ({bench['ms_per_step']:.3f} ms/step = {bench['value'] / 1e9:.2f} G atoms/s; `k_neighbor` range by HIP "
                        f"events inside the library {rf['avg_kernel_ms']:.4f} ms -> {rf['achieved']:.0f} GB/s algorithmic = {rf['frac']:.4f} of 8 TB/s; PMC traffic "
                        f"{(rf['traffic'] or 0) / 1e9:.3f} GB per launch, FETCH doubled, vs {rf['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic), then the same command "
                        "(`--no-extra --no-pmc --no-cpu-baseline`) under the kernel trace, whose table follows.\n\n"
                        "The tile kernel (the instance with the fused CNA: it writes the lists and the labels) is launched twice per build (all tiles, then the one-cell slices of the "
                        "tiles whose halo overflowed LDS: an empty stand-by on this input).  Round 6: `k_gather` is gone from the cell grid (a build of spatially ordered input keeps no cell-sorted copy of the atoms); "
                        "the tile kernel — the `..., true>` instance — reads them through the cell-sorted id list instead and is 0.06-0.09 ms slower for it (`r06_cell_grid_ab.txt`).\n\n| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|\n")
                for k, (c, a, mn, mx) in sorted(bs.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:16]:
                    f.write(f"| `{k}` | {c} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |\n")
    merged = {}
    for tag in ("r06_sq1", "r06_sq2", "r06_fetch", "r06_write"):
        pth = os.path.join(G, f"pmc_{tag}.json")
        if os.path.exists(pth):
            for k, v in json.load(open(pth)).items():
                merged.setdefault(short(k), {}).update({c: val for c, val in v.items()})
    merged = {k: v for k, v in merged.items() if not k.startswith("k_warm_")}
    bs, _ = stats_dir(os.path.join(G, "r06_bench"))
    for k, v in merged.items():
        d = {}
        if "SQ_INSTS_VALU" in v and v.get("SQ_WAVES"):
            d["valu_instr_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            d["wave_cycles_waiting_fraction"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
        if k in bs and "SQ_ACTIVE_INST_VALU" in v:
            d["kernel_us_under_the_bench_trace"] = bs[k][1] / 1e3
            d["valu_busy_fraction_per_simd"] = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (bs[k][1] * 1e-9 * 2.4e9)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            d["hbm_bytes_fetch_x2_plus_write"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        v["_derived"] = d
    if merged:
        merged["_note"] = ("per dispatch of every kernel of the headline step (tools/order_probe.py lattice 136 5 under four separate rocprofv3 --pmc passes, "
                           "tools/measure_r06.sh counters); SQ_* summed over the chip; FETCH_SIZE / WRITE_SIZE in KB (FETCH undoubled)")
        json.dump(merged, open(os.path.join(P, "r06_step_counters.json"), "w"), indent=1)

if "texts" in what:
    for src, dst in (("r06_knn_split.txt", "r06_knn_split.txt"), ("r06_knn_rows.txt", "r06_knn_rows.txt"), ("r06_notebook.txt", "r06_notebook.txt"),
                     ("r06_ptm_bench.txt", "r06_ptm_bench.txt"), ("r06_rc_sweep.txt", "r06_rc_sweep.txt")):
        pth = os.path.join(G, src)
        if os.path.exists(pth):
            keep = [l for l in open(pth) if not re.match(r"^[WE]\d{8}", l) and "amdgpu.ids" not in l]
            open(os.path.join(P, dst), "w").writelines(keep)
