#!/bin/bash
mkdir -p gpurun_out
python tests/probes/trained_diag.py > gpurun_out/r04_trained_diag.txt 2>&1
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_render.py::test_render_bf16_on_reference_trained_networks --deselect tests/test_gpu_render.py::test_render_bf16_parameter_gradients_on_reference_trained_networks --deselect tests/test_gpu_train_step.py::test_bf16_trainer_step_matches_the_oracle_with_bf16_rounding_emulation 2>&1 | tail -25 > gpurun_out/r04_fullsuite.txt
cat gpurun_out/r04_trained_diag.txt; cat gpurun_out/r04_fullsuite.txt
