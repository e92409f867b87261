"""bench.py section: the GauGAN SPADE generator (BASELINE.json configs[2])."""
import time

import torch

from .common import _hip, _replay_ms


def gaugan_section(dev, cpu_parity=True):
    """BASELINE.json configs[2]: GauGAN SPADE generator (ngf 64, 93 M parameters, random init), 256 x 512 label map
    (crop 512, aspect 2 -- gaugan/test.py:53-54), ~5 % relabelled rectangle; fp32, channels-last."""
    import numpy as np

    from sige_amd import runtime
    from sige_amd.utils import compute_difference_mask, dilate_mask, downsample_mask
    from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator

    def labels(dy=0, dx=0):
        rs = np.random.RandomState(3)
        coarse = rs.randint(0, 36, size=(32, 64))
        lab0 = np.kron(coarse, np.ones((8, 8), dtype=np.int64))
        lab1 = lab0.copy()
        lab1[85 + dy:136 + dy, 128 + dx:256 + dx] = (lab0[85 + dy:136 + dy, 128 + dx:256 + dx] + 5) % 36
        oh = lambda l: torch.nn.functional.one_hot(torch.from_numpy(l), 36).permute(2, 0, 1)[None].float().contiguous()  # noqa: E731
        return oh(lab0), oh(lab1)

    def build():
        torch.manual_seed(0)
        m = SpadeGenerator(SPADEConfig()).eval()
        g = torch.Generator().manual_seed(7)
        for n_, b_ in m.named_buffers():  # running statistics away from (0, 1): the cached affine matters
            if n_.endswith("running_mean"):
                b_.copy_(torch.randn(b_.shape, generator=g) * 0.3)
            elif n_.endswith("running_var"):
                b_.copy_(torch.rand(b_.shape, generator=g) + 0.5)
        return m

    x0c, x1c = labels()
    cl = lambda t_: t_.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    model = build().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, x1 = cl(x0c), cl(x1c)
    res = {}
    with torch.no_grad():
        model.set_mode("full")
        dense_ms, _, gd = _replay_ms(lambda: model(x1))
        del gd
        model(x0)
        diff = compute_difference_mask(x0, x1)
        model.set_masks(downsample_mask(dilate_mask(diff, 1), (model.sh, model.sw), dilation=2))
        model.set_mode("sparse")
        outs = {}
        for name, fused in (("fused_spade_modulation", True), ("module_chain", False)):
            model.cfg.fused = fused
            model(x1)  # (the first forward of a form packs weights / builds tables: not part of a steady-state forward)
            n0 = _hip().launch_count()
            model(x1)
            launches = _hip().launch_count() - n0
            ms, out, g = _replay_ms(lambda: model(x1))
            outs[name] = out.float().cpu()
            res[name] = {"forward_ms": round(ms, 3), "speedup_vs_dense": round(dense_ms / ms, 2), "hip_kernel_launches": launches}
            del g
        model.cfg.fused = True
        # the generator's REAL per-edit latency: the reference runs ONE sparse forward per edit (gaugan/runner.py:150-195), so what
        # a user waits for is difference mask + set_masks + the first (eager) forward under the new mask -- not a graph replay
        import statistics

        lat = {"difference_mask_and_set_masks": [], "first_forward_eager": []}
        # the edited label maps are resident on the GPU before anything is timed: a label map built with numpy and uploaded from
        # pageable memory inside the loop is registered as a userptr by the driver, and when Python frees it the process's queues are
        # evicted and restored ~100 ms later (amdgpu KFD) -- the "40 ms one-off" of round 4's per-edit numbers was this artefact
        # of the bench itself (tools/gaugan_latency.py --host-inputs reproduces it; profiles/r5*_gaugan_latency*.json)
        edit_places = ((20, 40), (-40, -60), (60, 120), (0, -100), (35, 10), (10, -30), (-20, 90))
        edit_inputs = [cl(labels(dy, dx)[1]) for dy, dx in edit_places]
        torch.cuda.synchronize()
        import gc

        gc.collect()       # (the host copies of those uploads are freed NOW, and the eviction they cause has passed before anything
        time.sleep(0.3)    #  below is timed)
        torch.cuda.synchronize()
        for i, xi in enumerate(edit_inputs[:5]):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_i = compute_difference_mask(x0, xi)
            model.set_masks(downsample_mask(dilate_mask(d_i, 1), (model.sh, model.sw), dilation=2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model(xi)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if i:  # (the first one warms the allocator up)
                lat["difference_mask_and_set_masks"].append((t1 - t0) * 1e3)
                lat["first_forward_eager"].append((t2 - t1) * 1e3)
        med = {k: round(statistics.median(v), 3) for k, v in lat.items()}
        res["per_edit_latency_ms"] = dict(med, to_first_output=round(sum(med.values()), 3),
                                          statistic="median of 4 edits (the first of 5 warms the allocator up)",
                                          all_edits={k: [round(x_, 3) for x_ in v] for k, v in lat.items()},
                                          note="a NEW edit of the same original through the MODULE path: difference mask + set_masks + "
                                               "the first eager forward (what gaugan/runner.py:150-195 does per edit); forward_ms above "
                                               "is the hipGraph replay of an unchanged mask")
        # ... and through a launch plan (sige_amd/plan.py): since round 5 the sparse forward reaches the GPU through the library only
        # (csrc/spade_ops.hip: nearest resizes, ReLU + split of the label features, dense SPADE modulation, conv_img with its leaky
        # ReLU / tanh), so ONE recording serves every later edit: copy the label map, difference mask, bind_mask, run
        def build_masks(m_):
            return downsample_mask(dilate_mask(m_, 1), (model.sh, model.sw), dilation=2)

        try:
            from sige_amd.plan import LaunchPlan

            seg = x1.clone()

            t0 = time.perf_counter()
            plan = LaunchPlan(model)
            plan.record(compute_difference_mask(x0, seg), build_masks, lambda: model(seg))
            torch.cuda.synchronize()
            rec_ms = (time.perf_counter() - t0) * 1e3
            rows = {"input_and_difference_mask": [], "bind_mask": [], "run": []}
            worst = 0.0
            for i, xi in enumerate(edit_inputs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                seg.copy_(xi)
                d_i = compute_difference_mask(x0, seg)
                t1 = time.perf_counter()
                plan.bind_mask(d_i)
                t2 = time.perf_counter()
                o_i = plan.run()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                if i:
                    rows["input_and_difference_mask"].append((t1 - t0) * 1e3)
                    rows["bind_mask"].append((t2 - t1) * 1e3)
                    rows["run"].append((t3 - t2) * 1e3)
                worst = max(worst, float((o_i - model(xi)).abs().max()))
            pm = {k: round(statistics.median(v), 3) for k, v in rows.items()}
            res["per_edit_latency_plan_ms"] = dict(pm, to_first_output=round(sum(pm.values()), 3), record_once_ms=round(rec_ms, 1),
                                                   calls={"masks": plan.calls(0), "forward": plan.calls(1)}, shape_bound=plan.shape_bound,
                                                   unbound_counts=plan.unbound_counts, tile_counts_last=[int(c_) for c_ in plan.counts][:6],
                                                   max_abs_vs_module_forward=round(worst, 9),
                                                   all_edits={k: [round(x_, 3) for x_ in v] for k, v in rows.items()},
                                                   statistic="median of 6 edits")
            del plan
        except Exception as e:  # (never the reason the section dies)
            res["per_edit_latency_plan_ms"] = {"error": repr(e)[:300]}
        # ---- stacked edits (sige_amd/stacked.py): E edited label maps of ONE original, each with its own mask,
        # through one set of launches -- every tensor of the sparse forward is the tall image [1,C,E*h,w]
        try:
            from sige_amd import stacked

            edits8 = ([x1] + edit_inputs)[:8]
            pyrs = [build_masks(compute_difference_mask(x0, xi)) for xi in edits8]
            singles = []
            for xi, p in zip(edits8, pyrs):
                model.set_masks(p)
                singles.append(model(xi).clone())
            one_ms = res["fused_spade_modulation"]["forward_ms"]
            rows = {}
            for E in (2, 4, 8):
                xs = torch.cat(edits8[:E], 0).contiguous(memory_format=torch.channels_last)
                stacked.stack_caches(model, E)
                try:
                    stacked.set_masks(model, pyrs[:E])
                    with stacked.edit_batch(model, E):
                        model(xs)
                        n0 = _hip().launch_count()
                        out = model(xs)
                        launches = _hip().launch_count() - n0
                        worst = max(float((out[e] - singles[e][0]).abs().max()) for e in range(E))
                        ms, _, g = _replay_ms(lambda: model(xs))
                        del g
                finally:
                    stacked.unstack_caches(model)
                rows[str(E)] = {"forward_ms": round(ms, 3), "ms_per_edit": round(ms / E, 3), "edits_per_s": round(1e3 * E / ms, 1),
                                "speedup_vs_one_edit_per_forward": round(one_ms * E / ms, 2), "hip_kernel_launches": launches,
                                "max_abs_vs_single_edit_forward": round(worst, 9)}
            res["batched_edits"] = {"one_edit_forward_ms": one_ms, "E": rows,
                                    "note": "E edited label maps of one original (different rectangles), each with its own mask, stacked "
                                            "along H (sige_amd/stacked.py + sige_hip_set_edit_batch): one launch per layer sees the tiles of "
                                            "all E edits; hipGraph replay; parity against each edit's own single-edit sparse forward"}
        except Exception as e:  # (never the reason the section dies)
            res["batched_edits"] = {"error": repr(e)[:300]}
        finally:
            model.set_masks(build_masks(diff))
    res["dense_forward_ms"] = round(dense_ms, 3)
    res["edit_ratio"] = round(float(diff.float().mean()), 4)
    res["fused_vs_chain_max_abs"] = round(float((outs["fused_spade_modulation"] - outs["module_chain"]).abs().max()), 8)
    if cpu_parity:
        from oracle import oracle

        ref = None
        try:
            from oracle import build_ref

            ref = build_ref.load()
        except Exception:
            ref = None
        runtime.register_backend("cpu", oracle.as_backend(ref) if ref is not None else oracle)
        try:
            cm = build()
            with torch.no_grad():
                cm.set_mode("full")
                cm(x0c)
                d = compute_difference_mask(x0c, x1c)
                cm.set_masks(downsample_mask(dilate_mask(d, 1), (cm.sh, cm.sw), dilation=2))
                cm.set_mode("sparse")
                want = cm(x1c)
        finally:
            runtime.unregister_backend("cpu")
        res["parity_max_abs"] = round(float((outs["fused_spade_modulation"] - want).abs().max()), 7)
        res["parity_against"] = "the same generator on the CPU, native ops = %s" % ("oracle/_ref (reference sige/cpu)" if ref is not None else "oracle C restatement")
    res["workload"] = "GauGAN SPADE generator ngf 64 (%.1fM params, random init), one-hot label map [1,36,256,512], %.1f%% relabelled, fp32 NHWC, hipGraph replay" % (
        sum(p_.numel() for p_ in model.parameters()) / 1e6, 100 * res["edit_ratio"])
    return res
