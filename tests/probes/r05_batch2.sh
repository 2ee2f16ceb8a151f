#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "=== full GPU suite"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "=== trained_diag"; timeout 600 python tests/probes/trained_diag.py 2>&1 | grep -v amdgpu
echo "=== dist A/B"; timeout 900 python tests/probes/r05_dist_ab.py 2>&1 | grep -v amdgpu
