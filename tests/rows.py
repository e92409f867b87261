"""SURVEY.md 8 rows -> the GPU tests that DEFINE them: each compares the HIP path with the CPU oracle (oracle/, the reference's
sige/cpu restated and -- as oracle/_ref -- compiled), with the golden vectors the real reference produced (tests/golden/), or,
for the rows whose arithmetic in the reference IS a torch op (a9 GroupNorm, f1 the dense remainder), with that torch op.
tests/conftest.py collects these first (tier 0) and the HIP-vs-HIP self-checks last (tier 2), so that one brittle comparison
cannot keep `pytest -x -m gpu` from reaching a row's evidence (GPUTEST_r05: test #281 of 462 failed, rows f2 / f3 / f4 / +g never
ran).  tests/test_collection_order.py (CPU) asserts both the order and that every name below exists, is gpu-marked and tier 0."""

ROW_TESTS = {
    "a1 reduce_mask": ["test_gpu_parity.py::test_reduce_mask_fixture_masks", "test_gpu_parity.py::test_golden_cases"],
    "a2 mask pyramid": ["test_gpu_round2.py::test_device_mask_helpers_bit_exact"],
    "a3 gather": ["test_gpu_parity.py::test_golden_cases", "test_gpu_parity.py::test_against_oracle_mid_size",
                  "test_gpu_round2.py::test_grouped_nchw_gather_bit_exact"],
    "a4 block conv": ["test_gpu_parity.py::test_block_conv_vs_torch_and_oracle", "test_gpu_parity.py::test_golden_cases"],
    "a5 scatter": ["test_gpu_parity.py::test_golden_cases", "test_gpu_parity.py::test_against_oracle_mid_size"],
    "a6 scatter with block residual": ["test_gpu_parity.py::test_golden_cases", "test_gpu_parity.py::test_resblock_gpu_vs_oracle_backend"],
    "a7 scatter_gather + map": ["test_gpu_parity.py::test_golden_cases", "test_gpu_round3.py::test_scatter_gather_row_form_bit_exact"],
    "a8 module / model wrappers": ["test_gpu_parity.py::test_ddpm_unet_gpu_vs_oracle_backend", "test_gpu_parity.py::test_example_py_on_gpu",
                                   "test_gpu_parity.py::test_example_golden_output_on_gpu",
                                   "test_gpu_round2.py::test_benchmarked_forward_vs_oracle_at_full_size"],
    "a9 cached affine producer": ["test_gpu_parity.py::test_group_norm_affine_vs_torch", "test_gpu_channels_last.py::test_group_norm_affine_cl"],
    "f1 dense remainder": ["test_gpu_parity.py::test_dense_fused_conv_vs_torch", "test_gpu_parity.py::test_attention_vs_torch",
                           "test_gpu_round2.py::test_benchmarked_forward_vs_oracle_at_full_size"],
    "f2 SPADE modulation": ["test_models_golden.py::test_gaugan_generator_on_the_gpu_matches_the_reference_fixture",
                            "test_models_golden.py::test_spade_modulate_kernel_equals_the_module_chain"],
    "f3 sparse-query attention glue": ["test_models_golden.py::test_sd_spatial_transformer_on_the_gpu_matches_the_reference_fixture",
                                       "test_models_golden.py::test_sd_unet_on_the_gpu_matches_the_reference_fixture",
                                       "test_gpu_round3.py::test_sd_unet_at_its_own_size_vs_cpu_oracle"],
    "f4 cache management + device mask pipeline": ["test_gpu_round3.py::test_multi_step_caches_cache_id",
                                                   "test_gpu_round3.py::test_f16_cache_ddpm_forward_whole_sweep",
                                                   "test_gpu_round2.py::test_set_masks_builds_every_index_list_with_one_sync",
                                                   "test_gpu_round5.py::test_stacked_edits_vs_cpu_oracle"],
    "+g fp16 block conv sweep (configs[4])": ["test_gpu_round2.py::test_f16_compute_block_conv_exact_products",
                                              "test_gpu_round2.py::test_f16_compute_ddpm_forward_vs_fp32_oracle",
                                              "test_gpu_round3.py::test_f16_policy_ddpm_forward_whole_sweep"],
}
