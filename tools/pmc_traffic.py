#!/usr/bin/env python3
"""Fabric-side bytes per launch per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
--kernel-trace only) of `tools/profile_forward.py --mode eager --replays R --manifest M.json`.

    pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv M.json OUT.json

The SIGE (sparse) and the dense-remainder convs run the SAME kernel symbols, so a kernel name cannot tell them apart; the
manifest is the family of every conv launch of one forward in launch order (written by the profiled script itself from
bench.py's op accounting), and the rows of `conv_mfma_kernel` dispatches, sorted by dispatch id, are matched to it
cyclically -- the same family definition as bench.py's `roofline` (block_conv_mfma = every conv over a SIGE tile list).
FETCH_SIZE is in KB and reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): doubled.  WRITE_SIZE in KB,
uncalibrated.  Infinity-Cache hits are included: an upper bound on HBM bytes.  The output records the hash of the kernel
sources it was measured on; bench.py prints `roofline.traffic` only when that hash matches the sources it runs."""
import collections
import csv
import json
import sys


def conv_rows(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    conv = [r for r in rows if "conv_mfma_kernel" in r["Kernel_Name"]]
    other = collections.defaultdict(list)
    for r in rows:
        if "sige::" in r["Kernel_Name"] and "conv_mfma_kernel" not in r["Kernel_Name"] and "pack_weights" not in r["Kernel_Name"]:
            other[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return conv, other


def main():
    fetch, write, manifest, out_path = sys.argv[1:5]
    man = json.load(open(manifest))
    seq = man["conv_families_in_launch_order"]
    fam = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    others = {}
    skipped = {}
    for path, counter in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        conv, other = conv_rows(path, counter)
        if len(conv) % len(seq):
            # the cache-producing full pass in front of the sparse forwards also runs tile-kernel launches when it is on the
            # library's kernels (f16 / f16x3 compute): they come first -- keep the trailing whole forwards, and make sure that what
            # is kept really is a whole number of identical launch sequences
            skipped[counter] = len(conv) % len(seq)
            conv = conv[skipped[counter]:]
        # the measured (steady-state) forwards must be repetitions of one launch sequence; the warm-up forwards in front of them
        # run the same calls through other instantiations (before the activated twins exist the consumers activate for themselves)
        names = [(r["Kernel_Name"], r.get("Grid_Size", "")) for r in conv]
        tail = max(0, len(names) - max(2, man.get("measured_forwards", 2)) * len(seq))
        if any(names[i] != names[i + len(seq)] for i in range(tail, len(names) - len(seq))):
            raise SystemExit("%s: the last %d conv dispatches are not repetitions of one %d-launch sequence" % (path, len(names) - tail, len(seq)))
        for i, r in enumerate(conv):
            fam[seq[i % len(seq)]][counter].append(float(r["Counter_Value"]))
        for k, v in other.items():
            others.setdefault(k, {})[counter] = sum(v) / len(v)
    out = {}
    for k, d in fam.items():
        rd = 2 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) / 1e3
        wr = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) / 1e3
        out[k] = {"dispatches_profiled": len(d["FETCH_SIZE"]), "launches_per_forward": seq.count(k),
                  "hbm_read_MB_per_launch_corrected_x2": round(rd, 3), "hbm_write_MB_per_launch": round(wr, 3),
                  "traffic_MB_per_launch": round(rd + wr, 3)}
    for k, d in others.items():
        out[k] = {"hbm_read_MB_per_launch_corrected_x2": round(2 * d.get("FETCH_SIZE", 0) / 1e3, 3),
                  "hbm_write_MB_per_launch": round(d.get("WRITE_SIZE", 0) / 1e3, 3),
                  "traffic_MB_per_launch": round((2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) / 1e3, 3)}
    meta = {"source_hash": man["source_hash"],
            "provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python "
                          "tools/profile_forward.py --mode eager --manifest ... (NHWC, in-place scatter, 1.2% edit); conv rows matched "
                          "to bench.py's families by launch order",
            "correction": "FETCH_SIZE in KB, doubled (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE in KB, uncalibrated; "
                          "Infinity-Cache hits included (upper bound on HBM bytes)",
            "leading_conv_dispatches_skipped": skipped,
            "families": out}
    json.dump(meta, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
