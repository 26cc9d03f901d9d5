"""profiles/rNN/rows.md: per-kernel algorithmic bytes, microseconds, GB/s and fraction of the 8 TB/s HBM peak for the kernels
of one outer iteration (from `rocprofv3 --kernel-trace --stats` of the default bench command: kernel_stats.csv + the bench line
printed under the profiler) and for the secondary rows (scripts/bench_rows.py under the profiler: rows_kernel_stats.csv +
rows.json).  SURVEY.md §8(d) gives the per-unit figures (fp32 storage model; this build stores fp64: 12 -> 24, 4-byte weight -> 8)."""
import csv
import json
import re
import sys

out = sys.argv[1]
b = json.load(open(f"{out}/bench_under_rocprof.json"))
P, K, V = b["config"]["points"], b["config"]["nodes"], b["config"]["vertices"]
E = 8 * V                                               # stored adjacency entries (ELL-8)
rl = b["roofline"]
sweep_bytes = rl["bytes_per_launch"]
alg = {   # kernel -> (bytes per launch, formula)
    "k_assoc_prep": (56 * K, "bounded passes: per node 24 B position + 24 B previous position + 4 B previous distance read, 4 B bound written (+ the lists of the few nodes a wave or a workgroup takes)"),
    "k_assoc_all": (24 * P + 40 * K, "bounded passes, association as a whole (SURVEY 8(d): 24 P + 24 K read + 16 K written — the bounded search visits a few dozen points per node, not all P) + node grid build + 9-NN graph + cotangent weights (24 V + 20 E read, 8 E + 8 V written) in the same launch"),
    "k_assoc_local": (24 * P + 40 * K, "the first two (unbounded) associations of a fit"),
    "k_assoc_heavy_knn": (None, "the far nodes of the unbounded associations + 9-NN graph + cotangent weights"),
    "k_ng_build1": (24 * K + 16 * K, "24 K node positions read, 16 K sorted records written (unbounded passes; bounded passes build the node grid inside k_assoc_all)"),
    "k_smooth": (144 * K, "one Jacobi sweep: 9 x (4 + 12) B gathered per node (SURVEY: 288 K for two)"),
    "k_ras_prepare": (None, "patch matrix: (8 W) B weights gathered + (8 W + 8) B written per patch-local row, 72 V rotations + 24 V start written"),
    "k_arap_rhs": ((72 + 24 + 24) * V + 12 * E + 48 * V, "(72 rotation + 24 rest + 24 x) V + 12 E (col, w) read, b and bpure (48 V) written; neighbour gathers are re-reads"),
    "k_ras_sweep<6, 0>": (sweep_bytes, "bench.py roofline: patch tables (10 W + 12) B per patch-local row + 72 V (x, b read; x written)"),
    "k_ras_sweep<6, 2>": ((24 + 24 + 72) * V + 12 * E, "deciding launch + ARAP local step on the owned rows: (24 rest + 24 x) V + 12 E read, 72 V rotations written (SURVEY: 108 V in fp32)"),
    "k_arap_finalize": (48 * V + 40 * K, "x -> geometry (24 V read, 24 V written), node positions / normals refreshed"),
}
rows = {}
for r in csv.DictReader(open(f"{out}/kernel_stats.csv")):
    rows[r["Name"].strip()] = r
steps = None
for name, r in rows.items():
    if name.endswith("k_arap_finalize"):
        steps = int(r["Calls"])
print(f"## Hot path, config 3 (P = {P}, K = {K}, V = {V}), one MI355X — rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-alt-solver --no-single-solve`\n")
print(f"bench line under the profiler: {b['ms_per_step']} ms per step; {steps} outer iterations in the trace (warm-up, timed region, event-timing and reference-schedule runs; the first iterations of every fresh handle are in the averages)\n")
print("| kernel | launches / outer iteration | avg us | algorithmic bytes / launch | GB/s | of 8 TB/s | what the bytes are |\n|---|---|---|---|---|---|---|")
for key, (nbytes, note) in alg.items():
    hit = [r for n, r in rows.items() if n.endswith(key) or n == key]
    if not hit:
        continue
    r = hit[0]
    avg = float(r["AverageNs"]) / 1e3
    per = int(r["Calls"]) / steps if steps else float("nan")
    if nbytes:
        gb = nbytes / avg / 1e3
        print(f"| `{key}` | {per:.1f} | {avg:.1f} | {nbytes / 1e6:.2f} MB | {gb:.0f} | {gb / 8000:.3f} | {note} |")
    else:
        print(f"| `{key}` | {per:.1f} | {avg:.1f} | — | — | — | {note} |")
print(f"\n`k_ras_sweep<6, 0>`: the average is over ALL its dispatches (idle ones return after one scalar load); per ACTIVE launch the bench line's `roofline.frac_active` = {rl.get('frac_active')} "
      f"(HIP events inside bench.py: {rl['avg_launch_us']} us per planned sweep launch, active fraction {rl['active_fraction']}).\n")
try:
    stats2, rows2 = f"{out}/rows_kernel_stats.csv", json.load(open(f"{out}/rows.json"))
    print("## Secondary rows (a2, a9, a14, f2, f3) — `rocprofv3 --kernel-trace --stats -- python3 scripts/bench_rows.py`\n")
    print("| kernel | calls | avg us | algorithmic bytes / call | GB/s | of 8 TB/s | note |\n|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats2)):
        m = re.search(r"\b(k_[a-z0-9_]+)\(", r["Name"]) or re.search(r"\b(k_[a-z0-9_]+)$", r["Name"].split("(")[0])
        if not m:
            continue
        short = m.group(1)
        avg = float(r["AverageNs"]) / 1e3
        if short in rows2:
            nb = rows2[short]["bytes"]
            print(f"| `{short}` | {r['Calls']} | {avg:.1f} | {nb / 1e6:.1f} MB | {nb / avg / 1e3:.0f} | {nb / avg / 1e3 / 8000:.3f} | {rows2[short]['note']} |")
        elif short in ("k_label_far",):
            print(f"| `{short}` | {r['Calls']} | {avg:.1f} | | | | queries the first pass left open (a wave each) |")
except FileNotFoundError:
    pass
