cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r3g/gpu_tests.txt 2>&1
tail -15 gpurun_out/r3g/gpu_tests.txt
