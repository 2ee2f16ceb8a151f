cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/full_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/full_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/full_smoke.txt
python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
tail -3 gpurun_out/full_gpu_tests.txt; tail -2 gpurun_out/full_smoke.txt; cut -c1-400 gpurun_out/full_bench.json
bash tools/profile.sh r03 > gpurun_out/profile_log.txt 2>&1; tail -5 gpurun_out/profile_log.txt
