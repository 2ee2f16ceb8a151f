import sys, os, random, tempfile, pathlib, importlib, traceback
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
m = importlib.import_module("test_gpu_spin_iter")
bad = 0
for s in range(40):
    random.seed(s)
    try:
        m.test_colmap_depth_render_and_prepare_export_and_lpips_hookup(pathlib.Path(tempfile.mkdtemp()))
    except Exception as e:
        bad += 1
        print("seed", s, "FAILED:", repr(e)[:300])
        traceback.print_exc(limit=2)
print("failures", bad, "of 40")
