#!/usr/bin/env python3
"""DESIGN.md's result tables of the current round (TAG), generated from the committed evidence (profiles/<TAG>_bench*.json = the bench lines and detail
files of the final GPU session) -- so that the document quotes the final evidence and nothing else.

    python tools/design_tables.py            # print the block
    python tools/design_tables.py --write    # replace the block between the markers in DESIGN.md

tests/test_profiles.py regenerates the block and compares it with what DESIGN.md holds."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(REPO, "profiles")
TAG = "r6"
BEGIN, END = "<!-- %s-tables:begin (tools/design_tables.py) -->" % TAG, "<!-- %s-tables:end -->" % TAG


def load(name):
    path = os.path.join(PROF, name)
    if not os.path.isfile(path):
        return None
    text = open(path).read().strip()
    try:
        return json.loads(text)
    except ValueError:
        return json.loads(text.splitlines()[-1])


def g(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


def fmt(v, nd=3):
    if v is None:
        return "—"
    if isinstance(v, float):
        return ("%%.%df" % nd) % v
    return str(v)


def block():
    line, det = load(TAG + "_bench.json"), load(TAG + "_bench_detail.json")
    if line is None or det is None:
        return None
    f16, sd, gl = load(TAG + "_bench_f16_detail.json"), load(TAG + "_bench_sd_detail.json"), load(TAG + "_bench_2ranks_gloo_detail.json")
    out = []
    out.append("Source: `profiles/%s_bench.json` (the <= 4 KB contract line as the driver reads it), `profiles/%s_bench_detail.json` (everything), "
               "`%s_bench_f16*`, `%s_bench_sd*`, `%s_bench_2ranks_gloo*`; kernel sources `%s` (= `sige_amd.build.source_hash()`, the hash `profiles/pmc_traffic.json` "
               "was measured on)." % (TAG, TAG, TAG, TAG, TAG, g(load("pmc_traffic.json"), "source_hash", default="?")))
    out.append("")
    out.append("| DDPM-256 sparse forward, 1×MI355X, exact fp32 | value |")
    out.append("|---|---|")
    r = det.get("roofline") or {}
    out.append("| forward at %.1f %% edit (hipGraph replay) | **%s ms** = %s forwards/s, %s launches |" % (
        100 * g(det, "config", "edit_ratio", default=0.012), fmt(det.get("forward_ms"), 4), fmt(det.get("value"), 1), det.get("launches_per_forward")))
    sd_ = det.get("speedup_vs_dense_detail") or {}
    out.append("| dense forward of the same U-Net (MIOpen, best layout) | %s ms: **%s×**; vs the library's exact-fp32 full pass %s×, vs its split-operand one %s× |" % (
        fmt(det.get("dense_forward_ms")), fmt(sd_.get("vs_miopen"), 2), fmt(sd_.get("vs_library_full_pass_f32_exact"), 2), fmt(sd_.get("vs_library_full_pass_f16x3"), 2)))
    out.append("| `roofline` (block conv, all %s SIGE launches of a forward) | %s TFLOP/s of %s = **%s**; %s µs per launch; counter traffic %s B per launch |" % (
        r.get("launches_per_forward"), fmt(r.get("achieved"), 1), fmt(r.get("peak"), 1), fmt(r.get("frac"), 3), fmt(r.get("avg_launch_us"), 2), r.get("traffic")))
    dr = g(det, "kernels", "dense_conv_mfma") or {}
    out.append("| dense remainder (%s launches) | %s µs, %s TFLOP/s (%s of the fp32 MFMA peak) |" % (
        dr.get("launches"), fmt(dr.get("us_total"), 1), fmt(dr.get("TFLOPs"), 1), fmt((dr.get("TFLOPs") or 0) / 157.3, 2)))
    for key, label in (("roofline_hbm", "scatter (out of place)"), ("roofline_gather", "gather + SiLU"), ("roofline_scatter_gather", "scatter_gather + SiLU")):
        h = det.get(key) or {}
        out.append("| `%s`: %s, %s | counter bytes: **%s** of 8 TB/s; algorithmic bytes: %s (%s µs) |" % (
            key, label, h.get("shape", ""), fmt(h.get("frac_on_counter_bytes"), 3), fmt(h.get("frac"), 3), fmt(h.get("us"), 2)))
    par = det.get("parity_max_abs") or {}
    out.append("| parity vs the reference's CPU path (same weights / inputs / masks) | %s (tolerance %s): %s |" % (
        ", ".join("%s: %.1e" % (k, v) for k, v in par.items()), det.get("parity_tolerance"), "ok" if det.get("parity_ok") else "FAILED"))
    c = det.get("cpu_baseline") or {}
    out.append("| `cpu_baseline` (%s, %s of %s host CPUs) | %s ms per forward = %s forwards/s |" % (
        c.get("kind"), c.get("cores"), c.get("host_cpus"), fmt(c.get("ms_per_forward"), 1), fmt(c.get("value"), 2)))
    out.append("| eager forward (Python, one ctypes call per launch) / issued from C by a launch plan | %s ms / %s ms |" % (
        fmt(det.get("forward_ms_eager")), fmt(det.get("forward_ms_eager_launch_plan"), 4)))
    p = g(det, "dynamic", "mask_change_plan") or {}
    out.append("| a new mask → first output through a launch plan | bind_mask %s + run %s = **%s ms** |" % (
        fmt(p.get("bind_mask")), fmt(p.get("run_from_c")), fmt(p.get("to_first_output_ms"))))
    out.append("")
    out.append("| edit ratio | forward ms | vs dense | block conv TFLOP/s | of the fp32 MFMA peak |")
    out.append("|---|---|---|---|---|")
    for row in det.get("sweep") or []:
        out.append("| %.1f %% | %s | %s× | %s | %s |" % (100 * row["edit_ratio"], fmt(row.get("forward_ms")), fmt(row.get("speedup_vs_dense"), 2),
                                                     fmt(row.get("block_conv_TFLOPs"), 1), fmt(row.get("block_conv_frac_of_mfma_peak"), 3)))
    t3 = det.get("tile_conv3") or {}
    if t3.get("rows"):
        out.append("")
        out.append("| tile conv v3 (exact fp32; routed from %s v3 workgroups on; docs/history.md §3.14) | router off | default | max \\|Δ\\| |" % t3.get("min_blocks"))
        out.append("|---|---|---|---|")
        for row in t3["rows"]:
            out.append("| forward at %.0f %% | %s ms | **%s ms** | %s |" % (100 * row["edit_ratio"], fmt(row.get("router_off_ms")), fmt(row.get("default_ms")),
                                                                     row.get("max_abs_vs_router_off")))
    b = det.get("batched_edits") or {}
    if b.get("rows"):
        out.append("")
        out.append("| stacked edits at %.1f %% (best routing per E) | ms per edit | forwards/s | block conv of peak |" % (100 * b.get("edit_ratio", 0.012)))
        out.append("|---|---|---|---|")
        best = {}
        for row in b["rows"]:
            if row["edits"] not in best or row["forwards_per_s"] > best[row["edits"]]["forwards_per_s"]:
                best[row["edits"]] = row
        for e in sorted(best):
            row = best[e]
            out.append("| E = %d (%s) | %s | %s | %s |" % (e, row.get("dense_route"), fmt(row.get("ms_per_edit"), 4), fmt(row.get("forwards_per_s"), 1),
                                                       fmt(row.get("block_conv_frac_of_mfma_peak"), 3)))
        for row in b.get("rows_at_other_edit_ratios") or []:
            out.append("| E = %d at %.0f %% | %s | %s | %s |" % (row["edits"], 100 * row["edit_ratio"], fmt(row.get("ms_per_edit"), 4),
                                                              fmt(row.get("forwards_per_s"), 1), fmt(row.get("block_conv_frac_of_mfma_peak"), 3)))
    gg = det.get("gaugan") or {}
    if gg and "error" not in gg:
        out.append("")
        out.append("| GauGAN SPADE generator, 256×512, %.1f %% relabelled | value |" % (100 * gg.get("edit_ratio", 0.05)))
        out.append("|---|---|")
        out.append("| sparse forward (hipGraph replay), all-library form / module chain | **%s ms** (%s launches) / %s ms; dense %s ms = %s× |" % (
            fmt(g(gg, "fused_spade_modulation", "forward_ms")), g(gg, "fused_spade_modulation", "hip_kernel_launches"),
            fmt(g(gg, "module_chain", "forward_ms")), fmt(gg.get("dense_forward_ms")), fmt(g(gg, "fused_spade_modulation", "speedup_vs_dense"), 2)))
        pe, pp = gg.get("per_edit_latency_ms") or {}, gg.get("per_edit_latency_plan_ms") or {}
        out.append("| a NEW edit → first output, module path | difference mask + set_masks %s + first eager forward %s = **%s ms** |" % (
            fmt(pe.get("difference_mask_and_set_masks")), fmt(pe.get("first_forward_eager")), fmt(pe.get("to_first_output"))))
        out.append("| a NEW edit → first output, launch plan | input + difference mask %s + bind_mask %s + run %s = **%s ms** (%s calls; max \\|Δ\\| to the module forward %s) |" % (
            fmt(pp.get("input_and_difference_mask")), fmt(pp.get("bind_mask")), fmt(pp.get("run")), fmt(pp.get("to_first_output")),
            g(pp, "calls", "forward"), pp.get("max_abs_vs_module_forward")))
        out.append("| parity vs the same generator on the CPU (reference natives) | %s |" % gg.get("parity_max_abs"))
        be = (gg.get("batched_edits") or {}).get("E") or {}
        if be:
            out.append("| stacked edits (E label maps of one original in one forward), ms per edit | %s (one edit per forward: %s ms) |" % (
                ", ".join("E = %s: **%s** (%s×, max \\|Δ\\| %s)" % (e, fmt(r.get("ms_per_edit")), fmt(r.get("speedup_vs_one_edit_per_forward"), 2),
                                                                   r.get("max_abs_vs_single_edit_forward")) for e, r in sorted(be.items(), key=lambda kv: int(kv[0]))),
                fmt(g(gg, "batched_edits", "one_edit_forward_ms"))))
    if f16:
        fr = f16.get("roofline") or {}
        out.append("")
        out.append("| configs[4] (`--dtype f16`: fp16 MFMA operands, fp16-stored cache) | value |")
        out.append("|---|---|")
        out.append("| forward at 1.2 %% | %s ms; block conv %s TFLOP/s = %s of the fp16 peak; traffic %s B per launch; parity (f16 criterion) %s |" % (
            fmt(f16.get("forward_ms"), 4), fmt(fr.get("achieved"), 1), fmt(fr.get("frac"), 4), fr.get("traffic"), "ok" if f16.get("parity_ok") else f16.get("parity_ok")))
        for row in f16.get("sweep") or []:
            out.append("| %.0f %% | %s ms, %s× |" % (100 * row["edit_ratio"], fmt(row.get("forward_ms")), fmt(row.get("speedup_vs_dense"), 2)))
    if sd:
        out.append("")
        out.append("| configs[3]: SD v1 U-Net (860 M parameters), latent [2,4,64,64], 15 % edit | value |")
        out.append("|---|---|")
        out.append("| sparse forward | **%s ms**, dense %s ms = %s×; %s library launches; parity %s (tolerance %s) |" % (
            fmt(sd.get("forward_ms")), fmt(sd.get("dense_forward_ms")), fmt(sd.get("speedup_vs_dense"), 2), sd.get("hip_kernel_launches_per_forward"),
            sd.get("parity_max_abs"), sd.get("parity_tolerance")))
    if gl:
        m = gl.get("multi_gpu") or {}
        out.append("")
        out.append("| 2 ranks on one GPU over gloo (the multi-rank code path; RCCL needs the driver's 8-GPU node) | value |")
        out.append("|---|---|")
        out.append("| start-up choice | `method_chosen` = **%s** of %s ms (watchdog %s s)%s |" % (
            m.get("method_chosen"), json.dumps(m.get("methods_ms")), m.get("watchdog_s"), "; fallback: " + str(m["fallback"]) if m.get("fallback") else ""))
        out.append("| job | value %s forwards/s, efficiency %s, cache identical on all ranks: %s |" % (
            fmt(gl.get("value"), 1), fmt(m.get("efficiency"), 3), m.get("cache_identical_on_all_ranks")))
    return "\n".join(out)


def main():
    b = block()
    if b is None:
        print("no profiles/%s_bench.json + %s_bench_detail.json yet" % (TAG, TAG), file=sys.stderr)
        return 1
    if "--write" in sys.argv:
        path = os.path.join(REPO, "DESIGN.md")
        text = open(path).read()
        i, j = text.index(BEGIN), text.index(END)
        open(path, "w").write(text[:i + len(BEGIN)] + "\n" + b + "\n" + text[j:])
    else:
        print(b)
    return 0


if __name__ == "__main__":
    sys.exit(main())
