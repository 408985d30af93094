"""Per-kernel floors of the configs[2] step from a committed counter summary ("close it with a per-kernel floor table").

usage: python tools/floor_table.py profiles/r5/pmc_summary.csv profiles/r5/kernel_stats_bench_steps3.csv > profiles/r5/floor_table.md

For every kernel above 0.2 ms of the step: the time its instruction streams and bytes would take on their own --
  valu   SQ_INSTS_VALU wave-instructions x 4 cycles (a wave64 instruction, fp64 or 32-bit, occupies its SIMD's 16 lanes for four
         cycles) / 1,024 SIMDs / the clock the kernel ran at (GRBM_GUI_ACTIVE / 8 XCDs / its duration under the counters);
  lds    SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over the CUs) / 256 CUs / the same clock;
  hbm    (FETCH_SIZE + WRITE_SIZE) KiB x 1024 / 6.3 TB/s (what a streaming kernel reaches on this part, MI355X_MICROARCH.md);
the floor is the largest of the three, `x floor` = measured time (rocprofv3 --kernel-trace average, no counters) / floor.  A kernel at
1.0 would overlap its other streams perfectly behind the binding one."""
import collections
import csv
import sys


def main():
    pmc, stats = sys.argv[1], sys.argv[2]
    per = collections.defaultdict(dict)
    with open(pmc) as fh:
        for row in csv.reader(l for l in fh if not l.startswith("#")):
            if len(row) == 4 and row[0] != "kernel":
                per[row[0].replace("void ", "").replace("natac::", "")][row[1]] = float(row[2]) / max(1.0, float(row[3]))
    avg = {}
    with open(stats) as fh:
        for r in csv.DictReader(fh):
            name = r["Name"].split("(")[0].replace("void ", "").replace("natac::", "").strip()
            avg[name] = float(r["AverageNs"])
    rows = []
    for k, c in per.items():
        if k not in avg or avg[k] < 2e5 or "GRBM_GUI_ACTIVE" not in c or "kernel_ns_under_pmc" not in c or k.startswith("natac_clock"):
            continue
        if any(x in k for x in ("tile_ranges", "frag_centres", "tile_heavy", "fft_template", "fft_edge_table", "lr_table")):
            continue            # indexes over the (immutable) inputs and model tables: formed once per batch / model, not per step
        ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / c["kernel_ns_under_pmc"]
        t_valu = c.get("SQ_INSTS_VALU", 0.0) * 4.0 / 1024.0 / ghz / 1e6
        t_lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256.0 / ghz / 1e6
        t_hbm = (c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 / 6.3e12 * 1e3
        floor = max(t_valu, t_lds, t_hbm)
        bound = "valu" if floor == t_valu else ("lds" if floor == t_lds else "hbm")
        rows.append((avg[k] / 1e6, k, ghz, t_valu, t_lds, t_hbm, bound, avg[k] / 1e6 / floor if floor > 0 else float("nan")))
    rows.sort(reverse=True)
    print("| kernel | ms per step | clock GHz | valu ms | lds ms | hbm ms | floor | x floor |")
    print("|---|---|---|---|---|---|---|---|")
    tot = tot_floor = 0.0
    for ms, k, ghz, tv, tl, th, bound, x in rows:
        print("| `%s` | %.2f | %.2f | %.2f | %.2f | %.2f | %s | %.2f |" % (k, ms, ghz, tv, tl, th, bound, x))
        tot += ms
        tot_floor += max(tv, tl, th)
    print("| sum | %.2f | | | | | %.2f | %.2f |" % (tot, tot_floor, tot / tot_floor))


if __name__ == "__main__":
    main()
