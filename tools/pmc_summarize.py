"""Condense rocprofv3 --pmc counter_collection CSVs (one directory per pass) into profiles/<round>/pmc_summary.csv.

usage: python tools/pmc_summarize.py OUT.csv PASS_DIR [PASS_DIR ...]
Per kernel and counter the value of the LAST dispatch is kept (the timed step of `bench.py --steps 1 --warmup 1`);
`kernel_ns_under_pmc` is that dispatch's duration in the pass that collected GRBM_GUI_ACTIVE.
"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    return name.split("(")[0].strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    last = collections.OrderedDict()
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    last[(k, r["Counter_Name"])] = float(r["Counter_Value"])
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
                        last[(k, "kernel_ns_under_pmc")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    with open(out, "w") as fh:
        fh.write("# rocprofv3 --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE), command: python bench.py "
                 "--steps 1 --warmup 1 --no-cpu-baseline\n")
        fh.write("# values of the LAST dispatch of each kernel (the timed step). FETCH_SIZE / WRITE_SIZE in KiB as reported; on gfx950 "
                 "FETCH_SIZE\n# under-counts wide streaming reads 2x (MI355X_MICROARCH.md, HBM section). GRBM_GUI_ACTIVE is summed over "
                 "the 8 XCDs:\n# effective clock = value / 8 / kernel_ns_under_pmc\n")
        from nucleoatac_amd._lib import profile_sha16
        fh.write("# source_sha16=%s (sha256 of nucleoatac_amd/csrc/* + include/natac.h + nucleoatac_amd/synth.py + bench.py's workload builder at "
                 "collection time; bench.py compares it with the current ones)\n" % profile_sha16())
        fh.write("kernel,counter,sum_over_dispatches,dispatches\n")
        for (k, c), v in last.items():
            fh.write('"%s",%s,%.1f,1\n' % (k, c, v))


if __name__ == "__main__":
    main()
