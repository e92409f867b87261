#!/usr/bin/env python3
"""Counter bytes per launch of the rows of bench.py's `data_movement` table, from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
separate runs, --kernel-trace only) of `tools/profile_data_movement.py --pmc-manifest M.json`:

    pmc_data_movement.py FETCH_counter_collection.csv WRITE_counter_collection.csv M.json OUT.json

The manifest gives, per row, the positions of its three eager calls in the library's launch sequence (sige_hip_launch_count);
every library launch is one kernel of namespace sige::, so the k-th sige:: dispatch (by dispatch id) IS launch k.  FETCH_SIZE is
in KB and reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): doubled.  WRITE_SIZE in KB.  Infinity-Cache hits
are included (fabric-side bytes: an upper bound on HBM bytes).  The output carries the hash of the kernel sources it was measured
on; bench.py prints `counter_MB` only when that hash matches."""
import csv
import json
import sys


def sige_dispatches(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "sige::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main():
    fetch, write, manifest, out_path = sys.argv[1:5]
    man = json.load(open(manifest))
    f, w = sige_dispatches(fetch, "FETCH_SIZE"), sige_dispatches(write, "WRITE_SIZE")
    total = man["total_library_launches"]
    if len(f) != total or len(w) != total:
        raise SystemExit("sige:: dispatches (%d fetch, %d write) != library launches %d: the positions cannot be trusted" % (len(f), len(w), total))
    out = []
    for r in man["rows"]:
        i0, n, calls = r["first_launch"], r["launches"], r["calls"]
        rd = 2.0 * sum(float(x["Counter_Value"]) for x in f[i0:i0 + n]) / calls / 1e3
        wr = sum(float(x["Counter_Value"]) for x in w[i0:i0 + n]) / calls / 1e3
        kern = sorted({x["Kernel_Name"].split("(")[0].replace("void ", "")[:80] for x in f[i0:i0 + n]})
        out.append({"op": r["op"], "layout": r["layout"], "edit_ratio": r["edit_ratio"], "alg_MB": r["alg_MB"],
                    "read_MB_corrected_x2": round(rd, 3), "write_MB": round(wr, 3), "counter_MB": round(rd + wr, 3),
                    "kernels_per_call": n // calls, "kernels": kern})
    json.dump({"source_hash": man["source_hash"],
               "provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python "
                             "tools/profile_data_movement.py --pmc-manifest ...; three eager launches per row, matched by launch position",
               "correction": "FETCH_SIZE in KB, doubled (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE in KB; Infinity-Cache hits included",
               "rows": out}, open(out_path, "w"), indent=1)
    for r in out:
        print("%-58s %-5s %5.3f  alg %8.2f MB  counter %8.2f MB (read %8.2f + write %8.2f)" % (r["op"], r["layout"], r["edit_ratio"], r["alg_MB"], r["counter_MB"], r["read_MB_corrected_x2"], r["write_MB"]))


if __name__ == "__main__":
    main()
