cd "${GRAFT_REPO_ROOT:-/root/repo}/tests"
for rep in 1 2 3; do python probes/pair_debug.py 2>&1 | grep "weight.*rel\|bias.*rel" | grep -v "rel [0-9.]*e-0[23]" | head -8; done
echo "debug runs done"
