#!/usr/bin/env python
"""Assemble profiles/r04_* from the raw outputs of tools/measure_r04.sh under gpurun_out/ (run in the build container after
the gpurun call).  Everything written here is a copy or a per-kernel reduction of rocprofv3 / bench.py output; nothing is
typed in by hand."""
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
    return {short(k): v for k, v in json.load(open(p)).items()} if os.path.exists(p) else {}


# ---- bench line + its kernel stats
bench = json.load(open(os.path.join(G, "r04_bench.json")))
json.dump(bench, open(os.path.join(P, "r04_bench_line.json"), "w"), indent=1)
shutil.copy(os.path.join(G, "r04_prof_bench", "b_kernel_stats.csv"), os.path.join(P, "r04_bench_kernel_stats.csv"))
rf = bench["roofline"]
json.dump({"source": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) run by bench.py itself over the benchmarked step, gpurun of this round",
           "traffic_bytes_per_launch_raw": rf["traffic_raw_fetch_plus_write"], "traffic_bytes_per_launch_fetch_x2": rf["traffic"],
           "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
           "note": "FETCH_SIZE doubled for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); kernel = k_neighbor_lane incl. its second pass"},
          open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)
bs = stats(os.path.join(G, "r04_prof_bench", "b_kernel_stats.csv"))
with open(os.path.join(P, "r04_bench_kernel_stats.md"), "w") as f:
    f.write("# bench.py under `rocprofv3 --kernel-trace --stats` (round 4, one MI355X, 10 061 824-atom FCC Cu, M = 16)\n\n")
    f.write(f"bench line of the same build: {bench['ms_per_step']:.3f} ms/step = {bench['value'] / 1e9:.2f} G atoms/s; `k_neighbor` range "
            f"(HIP events inside the library) {rf['avg_kernel_ms']:.4f} ms -> {rf['achieved']:.0f} GB/s algorithmic = {rf['frac']:.4f} of 8 TB/s; "
            f"PMC traffic {rf['traffic'] / 1e9:.3f} GB per launch (FETCH doubled) vs {rf['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic.\n\n")
    f.write("The lane kernel is launched twice per build (all tiles, then the one-cell slices of the tiles whose halo overflowed LDS): "
            "the `Max` column is the first launch, the average mixes both.\n\n| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|\n")
    for k, (c, a, mn, mx) in sorted(bs.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        f.write(f"| `{k}` | {c} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |\n")

# ---- neighbour kernel counters
# every build = the main launch (instance <COUNT=false, TRI=false, LOOP=false, FCNA=false>: one tile per workgroup) + the slice
# launch (<false, false, true, false>: the listed tiles, ~2 % of the work); pmc() gives the mean per dispatch of each instance
nbm = {}
for tag in ("nb_sq1", "nb_sq2", "nb_fetch", "nb_write"):
    for k, v in pmc(tag).items():
        if "k_neighbor_lane<false, false" in k:
            for c, val in v.items():
                if c != "dispatches":
                    nbm[c] = nbm.get(c, 0.0) + val
nbm["_note"] = ("per BUILD (main launch + slice launch) of k_neighbor_lane<false,false,*,false>, tools/nb_probe.py 136 16 0.854; SQ_* summed over the chip. "
                "SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU are in units of 4 cycles per SIMD; FETCH_SIZE / WRITE_SIZE in KB (FETCH undoubled)")
kernel_ms = rf["avg_kernel_ms"]
cyc = kernel_ms * 1e-3 * 2.4e9
nbm["_derived"] = {"kernel_ms": kernel_ms,
                   "valu_busy_fraction_per_simd": nbm["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                   "valu_wave_instructions": nbm["SQ_INSTS_VALU"], "waves": nbm["SQ_WAVES"],
                   "valu_instr_per_wave": nbm["SQ_INSTS_VALU"] / nbm["SQ_WAVES"], "lds_instr_per_wave": nbm["SQ_INSTS_LDS"] / nbm["SQ_WAVES"],
                   "salu_instr_per_wave": nbm["SQ_INSTS_SALU"] / nbm["SQ_WAVES"],
                   "wave_cycles_waiting_fraction": nbm["SQ_WAIT_ANY"] / nbm["SQ_WAVE_CYCLES"],
                   "wave_cycles_issuing_fraction": nbm["SQ_ACTIVE_INST_ANY"] / nbm["SQ_WAVE_CYCLES"],
                   "wave_cycles_issue_stalled_fraction": nbm["SQ_WAIT_INST_ANY"] / nbm["SQ_WAVE_CYCLES"],
                   "hbm_bytes_fetch_x2_plus_write": (2 * nbm["FETCH_SIZE"] + nbm["WRITE_SIZE"]) * 1024}
json.dump(nbm, open(os.path.join(P, "r04_neighbor_sq_counters.json"), "w"), indent=1)

# ---- the analyses of configs 2 and 4
shutil.copy(os.path.join(G, "r04_prof_an", "an_kernel_stats.csv"), os.path.join(P, "r04_analyses_kernel_stats.csv"))
log = [l for l in open(os.path.join(G, "r04_prof_an", "run.log")) if not re.match(r"^[WE]\d{8}", l)]
open(os.path.join(P, "r04_analyses_run.log"), "w").writelines(log)
an = stats(os.path.join(G, "r04_prof_an", "an_kernel_stats.csv"))
fe, wr, s1, s2 = pmc("an_fetch"), pmc("an_write"), pmc("an_sq1"), pmc("an_sq2")
N3, N5 = 10061824, 9841500
# (kernel, atoms, algorithmic bytes per atom, what the bytes are)
ALG = [
    ("k_knn_near<false, 18>", N3, 24 + 12 * 18, "k = 18: positions 24 + ids 4k + distances 8k"),
    ("k_knn_near<false, 12>", N3, 24 + 12 * 12, "k = 12"),
    ("k_knn_near<false, 14>", N3, 24 + 12 * 14, "k = 14"),
    ("k_knn<false>", N3, 24 + 12 * 18, "the general kernel on the to-do list of the near kernel (empty here: it leaves at once)"),
    ("ptms::k_ptm_order_faces<false, 10, false, 2>", N3, 24 + 72 + 18 + 72, "positions, row 4*18 in; order 18 B + ordered ids 4*18 out"),
    ("ptms::k_ptm_hull<false>", N3, 24 + 72 + 2 * (56 + 1), "positions + ordered ids in; 2 hulls x (28 facets x 2 B + status) out (fcc-hcp-bcc)"),
    ("ptms::k_ptm_canon<12, false>", N3, 57 + 8 + 17 + 1, "facets in; hash, labelling, flag out"),
    ("ptms::k_ptm_canon<14, false>", N3, 57 + 8 + 17 + 1, "same, 15-point cluster"),
    ("ptms::k_ptm_match<false, false>", N3, 24 + 72 + 2 * 26 + 64 + 72, "positions, ids, 2 x (hash, labelling, flag) in; (N,8) f64 + (N,18) i32 out"),
    ("k_sq_stage1_l<false, 4>", N3, 24 + 4 + 12 * 12 + 2 * 16 * 9, "positions, count, 12 ids + distances in; q_4m (9 x re,im) read and written"),
    ("k_sq_stage1_l<false, 6>", N3, 24 + 4 + 12 * 12 + 2 * 16 * 13, "same, q_6m (13 x re,im)"),
    ("k_sq_final<true>", N3, 16 * 2 * 13 + 16, "q_lm rows in (416 B); q4, q6 out"),
    ("k_csp<false, 12>", N3, 24 + 4 * 12 + 8, "positions, 12 ids in; csp out"),
    ("k_acna_f32", N3, 24 + 4 * 14 + 4, "positions, 14 ids in; label out (single-precision pair tests; the double-precision kernel finishes its to-do list)"),
    ("ptms::k_ptm_shell<false, 4, 3>", N3, 24 + 72 + 18 + 17 * 28 + 1, '"all": positions, ordered ids + ranks in; 17-point cluster (ids, points) out'),
    ("ptms::k_ptm_hull_shell", N3, 17 * 24 + 1 + 57, '"all": cluster points in; 28 facets + status out'),
    ("ptms::k_ptm_canon<16, true>", N3, 57 + 8 + 17 + 1, '"all": facets in; hash, labelling, flag out'),
    ("ptms::k_ptm_match<false, true>", N3, 24 + 72 + 3 * 26 + 17 * 28 + 10 * 28 + 64 + 72, '"all": + both clusters in'),
    ("k_rdf_tile<false>", N5, 28, "positions 24 + type 4 (the histogram is 6.4 kB)"),
    ("k_wcp_count", N5, 8 + 4 * 27, "count, type, row of the rc = 3.6 list (width 27)"),
]
with open(os.path.join(P, "r04_analyses_roofline.md"), "w") as f:
    f.write("# Kernels of BASELINE configs 2 and 4 at full size (round 4, one MI355X)\n\n"
            "`tools/profile_analyses.py c3 c5` under `rocprofv3 --kernel-trace --stats` (`r04_analyses_kernel_stats.csv`, wall times in "
            "`r04_analyses_run.log`); FETCH_SIZE / WRITE_SIZE and the SQ counters from four separate `--pmc` passes over the same command "
            "(`tools/pmc_any.sh`).  c3 = 136^3 fcc Cu, N(0, 0.05) seed 0, 10 061 824 atoms; c5 = Cu64Zr36 glass, 9 841 500 atoms.\n"
            "Algorithmic bytes per atom: compulsory unique traffic of that kernel (SURVEY 8d convention: gathers and LDS reuse not counted). "
            "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per dispatch (FETCH doubled as the guide prescribes for wide streaming reads; gathers are "
            "over-counted by that, so read the column as an upper bound). VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / kernel time.\n\n"
            "| kernel | avg ms | algorithmic B/atom | algorithmic GB/s | % of 8 TB/s | HBM bytes / algorithmic | VALU busy | waves waiting | bound by |\n|---|---|---|---|---|---|---|---|---|\n")
    for k, n, bpa, what in ALG:
        if k not in an:
            f.write(f"| `{k}` | (not in this run) | {bpa} | | | | | | |\n")
            continue
        calls, avg, mn, mx = an[k]
        ms = avg / 1e6
        gbs = bpa * n / (avg * 1e-9) / 1e9
        hbm = (2 * fe.get(k, {}).get("FETCH_SIZE", 0) + wr.get(k, {}).get("WRITE_SIZE", 0)) * 1024
        valu = s1.get(k, {}).get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (avg * 1e-9 * 2.4e9)
        wait = s2.get(k, {}).get("SQ_WAIT_ANY", 0) / max(1.0, s2.get(k, {}).get("SQ_WAVE_CYCLES", 1))
        bound = "VALU issue" if valu > 0.55 else ("latency (waves parked)" if wait > 0.5 else "mixed: issue + latency")
        f.write(f"| `{k}` | {ms:.2f} | {bpa} ({what}) | {gbs:.0f} | {100 * gbs / HBM:.1f} | {hbm / (bpa * n):.2f} | {100 * valu:.0f} % | {100 * wait:.0f} % | {bound} |\n")
    f.write("\nNone of these kernels is near the HBM roofline, and none should be: per atom they do 10^3 - 10^5 operations on a few hundred bytes "
            "(PTM: convex hulls, graph canonical forms, quaternion superpositions; Steinhardt: spherical harmonics up to l = 6 for 12 bonds; RDF: "
            "~460 distance tests per atom).  What the counters say is whether the lanes are busy: the staged PTM kernels and the RDF tile kernel "
            "issue VALU work most of the time, the gather-heavy ones wait on the L2.\n")

# ---- texts copied as measured
for name in ("r04_ptm_bench.txt", "r04_tri_lane.txt", "r04_lane_phase_stamps.txt", "r04_rc5_m50.txt", "r04_consumer_times.txt", "r04_halo_cost.txt",
             "r04_host_path.txt"):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))
# the rc 5 / 50-slot call against its roofline: (28 + 12 * 50) B per atom
rows = []
cur = None
for line in open(os.path.join(G, "r04_rc5_m50.txt")):
    if line.startswith("=="):
        cur = line.strip("= \n")
    m = re.match(r"k_neighbor (\d+) ([0-9.]+) N (\d+)", line)
    if m and cur:
        reps, tot, n = int(m.group(1)), float(m.group(2)), int(m.group(3))
        ms = tot / reps
        rows.append((cur, n, ms, 628.0 * n / (ms * 1e-3) / 1e9))
with open(os.path.join(P, "r04_rc5_m50.md"), "w") as f:
    f.write("# build_neighbor(5.0, max_neigh=50) on fcc Cu — the reference's own benchmark call (doc/gettingstarted/benchmark.ipynb)\n\n"
            "`tools/nb_probe.py <cells> 50 1.38313` (rc = 5.0 A), `k_neighbor` range = every kernel of the neighbour pass (HIP events inside the "
            "library); algorithmic bytes 28 + 12 x 50 = 628 per atom.  The reference publishes 0.4 s per 10^6 atoms for this call (BASELINE.md).\n\n"
            "| run | atoms | ms per build | algorithmic GB/s | fraction of 8 TB/s | M atoms/s |\n|---|---|---|---|---|---|\n")
    for cur, n, ms, gbs in rows:
        f.write(f"| {cur} | {n} | {ms:.2f} | {gbs:.0f} | {gbs / HBM:.3f} | {n / ms / 1e3:.0f} |\n")

# ---- round 4: fixed cost per step
for name in ("r04_launch_gap.txt", "r04_atomic_scope.txt", "r04_fast_path_holes.txt", "r04_step_vs_n.txt", "r04_unwrapped_sweep.txt", "r04_triclinic_sweep.txt",
             "r04_open_box_sweep.txt", "r04_disorder_sweep.txt", "r04_host_enqueue.txt", "r04_cold_rdf.txt", "r04_fcna_hot.txt"):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))
svn = os.path.join(G, "r04_step_vs_n.txt")
if os.path.exists(svn):
    recs = [json.loads(l) for l in open(svn) if l.startswith("{")]
    und = [r for r in recs if r.get("kind") == "undivided"]
    slabs = [r for r in recs if str(r.get("kind", "")).startswith("slab")]
    fit = next((r for r in recs if any(k.startswith("fit") for k in r)), None)
    ratios = [r for r in recs if "slab" in r and "step / (undivided / 8)" in r]
    with open(os.path.join(P, "r04_step_vs_n.md"), "w") as f:
        f.write("# Step time against system size (round 4, one MI355X; `tools/step_vs_n.py`)\n\n"
                "The bench step — `build_neighbor(rc = 0.854 a, max_neigh = 16)` + fixed-cutoff CNA through the C ABI, positions resident in HBM — on fcc Cu "
                "cubes of n^3 cells; `ranges` are the library's own HIP-event ranges (each costs ~8 us of stream time, so they are measured in a run of their own).\n\n"
                "| cells | atoms | us per step | cell grid | k_neighbor | k_fcna | plan (tile, cap) |\n|---|---|---|---|---|---|---|\n")
        for r in und:
            g = r["ranges_us"]
            f.write(f"| {r['cells']}^3 | {r['atoms']} | {r['us']} | {g.get('cell_grid')} | {g.get('k_neighbor')} | {g.get('k_fcna')} | {r['plan'][0]}x{r['plan'][0]}x{r['plan'][1]}, {r['plan'][2]} |\n")
        if fit:
            v = list(fit.values())[0]
            f.write(f"\nLeast-squares fit t = t0 + N / rate over these rows: **t0 = {v['t0_us']} us**, rate = {v['atoms_per_us']} atoms/us.\n")
        f.write("\nSlabs of a box split 8 ways along x, rank 1, loop-back transport (the product's exchange code, a device copy instead of RCCL — the wire is NOT measured):\n\n"
                "| slab (cells) | owned atoms | ghosts | us per step | us with the next frame's halo prefetched | cell grid | k_neighbor | k_fcna |\n|---|---|---|---|---|---|---|---|\n")
        for r in slabs:
            g = r["ranges_us"]
            f.write(f"| {r['cells']} | {r['atoms']} | {r['ghosts']} | {r['us']} | {r['us_prefetched']} | {g.get('cell_grid')} | {g.get('k_neighbor')} | {g.get('k_fcna')} |\n")
        f.write("\n| slab | step / (undivided step of the whole box / 8) | with prefetch |\n|---|---|---|\n")
        for r in ratios:
            f.write(f"| {r['slab']} | {r['step / (undivided / 8)']} | {r['prefetched']} |\n")
        # the device timeline of one small step
        tr = os.path.join(G, "r04_prof_small", "s_kernel_trace.csv")
        if os.path.exists(tr):
            rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
            steps, cur = [], []
            for r in rows:
                if "FillFunctor<int>" in r["Kernel_Name"] and cur:
                    steps.append(cur); cur = []
                cur.append(r)
            steps = [s_ for s_ in steps[10:-1] if len(s_) == 11]
            if steps:
                s_ = steps[len(steps) // 2]
                t0 = int(s_[0]["Start_Timestamp"]); prev = None
                f.write("\nDevice timeline of one 4 000-atom step (rocprofv3 --kernel-trace of `tools/step_vs_n.py --only 10`; the profiler slows the host side "
                        "down, the kernel durations are what counts):\n\n| kernel | start us | duration us | gap before us |\n|---|---|---|---|\n")
                for r in s_:
                    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
                    f.write(f"| `{short(r['Kernel_Name'])[:60]}` | {st / 1e3:.1f} | {(en - st) / 1e3:.1f} | {((st - prev) / 1e3 if prev is not None else 0):.1f} |\n")
                    prev = en
                f.write(f"\nSum of the eleven kernel durations: {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in s_) / 1e3:.1f} us.\n")
        lg = os.path.join(G, "r04_launch_gap.txt")
        if os.path.exists(lg):
            f.write("\nWhat a dependent launch costs on this box (`tools/ubench/launch_gap.hip`: chains of K empty kernels on one stream, as plain launches and as one hipGraph):\n\n```\n" + open(lg).read() + "```\n")

# ---- round 4: cold path
with open(os.path.join(P, "r04_cold_path.md"), "w") as f:
    f.write("# The first calls of a process, and an NPT-like trajectory (round 4, one MI355X; `tools/cold_path.py`)\n\n"
            "Fresh process, rattled fcc Cu as host numpy arrays, every API called once on a new `System` (first), then on a second and a third `System` of the "
            "same atoms.  With `mdh_warm` (the default: every code object of the library and the queue's scratch area are loaded when the package loads the library, "
            "i.e. at the first `System`) and with `MDAPY_HIP_WARM=0` (the round-3 behaviour: each translation unit's code object is loaded by the first launch that needs it).\n")
    for tag, title in (("63", "1 000 188 atoms"), ("63_nowarm", "1 000 188 atoms, MDAPY_HIP_WARM=0"), ("136", "10 061 824 atoms"), ("136_nowarm", "10 061 824 atoms, MDAPY_HIP_WARM=0")):
        pth = os.path.join(G, f"r04_cold_path_{tag}.txt")
        if os.path.exists(pth):
            f.write(f"\n## {title}\n\n```\n" + open(pth).read() + "```\n")
# ---- per-kernel counters of the headline step
cnt = {}
for tag in ("step_sq1", "step_sq2", "step_sq3", "step_ta", "step_tcp", "step_tcc"):
    for k, v in pmc(tag).items():
        if "k_warm" in k:
            continue
        cnt.setdefault(k, {}).update({c: val for c, val in v.items() if c != "dispatches"})
if cnt:
    cnt["_note"] = ("per dispatch, bench.py --steps 5 (10 061 824-atom headline step), six separate rocprofv3 --pmc passes (tools/measure_r04_h.sh); SQ_* summed over the chip, "
                    "SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU in units of 4 cycles per SIMD; TA_BUSY_avr = cycles the address unit of an average CU was busy")
    json.dump(cnt, open(os.path.join(P, "r04_step_counters.json"), "w"), indent=1)
print("written:", sorted(x for x in os.listdir(P) if x.startswith("r04_")))
