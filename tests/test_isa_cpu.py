"""CPU: the fused kernels read LDS through inline asm with hand-counted s_waitcnt lgkmcnt(N) (the compiler
would otherwise drain the LDS-DMA queue in front of every read).  tools/check_lds_asm.py proves on the
gfx950 listing that no instruction touches a destination register before a wait has covered it."""
import importlib.util
import os
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_lds_asm", os.path.join(ROOT, "tools", "check_lds_asm.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def _listing(tmp_path, body):
    p = tmp_path / "k.s"
    # hipcc writes the kernel label with a trailing comment
    p.write_text("_Z6kernelv:                             ; @_Z6kernelv\n" + textwrap.dedent(body) + "\ts_endpgm\n")
    return str(p)


def test_checker_accepts_counted_waits(tmp_path):
    ok = """
        ;;#ASMSTART
        ds_read_b128 v[0:3], v9 offset:0
        ;;#ASMEND
        ;;#ASMSTART
        ds_read_b128 v[4:7], v9 offset:1024
        ;;#ASMEND
        ;;#ASMSTART
        s_waitcnt lgkmcnt(1)
        ;;#ASMEND
        v_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[40:43], v[16:31]
        ;;#ASMSTART
        s_waitcnt lgkmcnt(0)
        ;;#ASMEND
        v_mfma_f32_32x32x16_bf16 v[16:31], v[4:7], v[40:43], v[16:31]
    """
    assert chk.check(_listing(tmp_path, ok)) == 0


def test_checker_flags_early_use(tmp_path):
    bad = """
        ;;#ASMSTART
        ds_read_b128 v[0:3], v9 offset:0
        ;;#ASMEND
        ;;#ASMSTART
        ds_read_b128 v[4:7], v9 offset:1024
        ;;#ASMEND
        ;;#ASMSTART
        s_waitcnt lgkmcnt(1)
        ;;#ASMEND
        v_mov_b32_e32 v20, v5
    """
    assert chk.check(_listing(tmp_path, bad)) == 1


def test_built_listings_are_clean():
    lib = os.path.join(ROOT, "spin-nerf_amd", "lib")
    listings = [os.path.join(lib, f) for f in sorted(os.listdir(lib))] if os.path.isdir(lib) else []
    listings = [p for p in listings if p.endswith(".gfx950.s")]
    if not listings:
        pytest.skip("no device listing (run __graft_entry__.build() first)")
    unverified_before = chk.unverified[0]
    for p in listings:
        seen = chk.n_kernels[0]
        assert chk.check(p) == 0, p
        # the kernels were recognised (for a round the label pattern missed hipcc's `name: ; @name` lines and the check
        # passed on nothing)
        assert chk.n_kernels[0] - seen >= 5, p
        txt = open(p).read()
        assert "ds_read_b128" in txt   # the listing really contains the asm reads
    # no asm global load whose counted wait sits behind a branch (the checker cannot judge those): the kernels are straight-line
    # code between a flag load and its use
    assert chk.unverified[0] == unverified_before, "an asm global load is used behind a branch: not checkable"


def test_checker_covers_asm_global_loads(tmp_path):
    load = """
        ;;#ASMSTART
        s_nop 4
        global_load_dwordx4 v[0:3], v8, s[2:3]
        ;;#ASMEND
    """
    dma = "        global_load_lds_dwordx4 v9, s[4:5]\n"
    use = "        v_and_b32_e32 v20, 0x10001, v1\n"
    wait = "        ;;#ASMSTART\n        s_waitcnt vmcnt(6)\n        ;;#ASMEND\n"
    assert chk.check(_listing(tmp_path, load + use)) == 1                       # consumed right away
    assert chk.check(_listing(tmp_path, load + dma * 3 + wait + use)) == 1      # only 3 loads behind it: vmcnt(6) proves nothing
    assert chk.check(_listing(tmp_path, load + dma * 7 + wait + use)) == 0      # 7 younger loads, at most 6 outstanding


def test_checker_flags_a_valu_write_right_in_front_of_an_asm_mfma(tmp_path):
    mfma = "        ;;#ASMSTART\n        v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n        ;;#ASMEND\n"
    mov = "        v_mov_b32_e32 v2, v30\n"
    assert chk.check(_listing(tmp_path, mov + mfma)) == 1                       # no wait state in between
    assert chk.check(_listing(tmp_path, mov + "        s_add_u32 s4, s4, 1\n" + mfma)) == 1     # one
    assert chk.check(_listing(tmp_path, mov + "        s_nop 1\n" + mfma)) == 0                 # two
    assert chk.check(_listing(tmp_path, "        v_mov_b32_e32 v9, v30\n" + mfma)) == 0         # another register


def test_checker_does_not_vouch_for_a_global_load_used_behind_a_branch(tmp_path):
    body = """
        ;;#ASMSTART
        global_load_dwordx4 v[0:3], v8, s[2:3]
        ;;#ASMEND
        s_cbranch_scc1 .LBB0_2
        v_and_b32_e32 v20, 0x10001, v1
    """
    before = chk.unverified[0]
    assert chk.check(_listing(tmp_path, body)) == 0
    assert chk.unverified[0] == before + 1


def test_spill_checker_tells_a_tile_loop_from_the_prologue(tmp_path):
    """tools/check_spills.py (round 5): a spill inside an innermost loop that contains MFMAs is a per-tile
    cost and fails the named kernel; the same spill in the prologue does not.  The built library's listing: the pair kernel's
    51 scratch operations and 215 SGPR-spill lane moves all sit outside its 14 tile loops."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_spills.py")
    head = "_Z6kernelv:                             ; @_Z6kernelv\n"
    loop = ".LBB0_1:\n\tv_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[40:43], v[16:31]\n%s\ts_cbranch_scc1 .LBB0_1\n"
    tail = "\ts_endpgm\n.Lfunc_end0:\n"
    spill = "\tscratch_store_dword off, v2, off offset:4\n"
    good = tmp_path / "good.s"; good.write_text(head + spill + loop % "" + tail)
    bad = tmp_path / "bad.s"; bad.write_text(head + loop % spill + tail)
    assert subprocess.run([sys.executable, tool, str(good), "kernel"]).returncode == 0
    assert subprocess.run([sys.executable, tool, str(bad), "kernel"]).returncode == 1
    # a gate name that matches no kernel of the listings fails instead of passing on nothing; a stated budget of
    # in-loop spill operations passes at the budget and fails below it
    assert subprocess.run([sys.executable, tool, str(good), "renamed_kernel"]).returncode == 1
    assert subprocess.run([sys.executable, tool, str(good), str(bad), "--no-scratch", "renamed"]).returncode == 1
    assert subprocess.run([sys.executable, tool, str(bad), "--max-inner", "kernel=1"]).returncode == 0
    assert subprocess.run([sys.executable, tool, str(bad), "--max-inner", "kernel=0"]).returncode == 1
    lst = os.path.join(ROOT, "spin-nerf_amd", "lib", "mlp_bwd.gfx950.s")
    if os.path.exists(lst):
        r = subprocess.run([sys.executable, tool, lst, "mlp_wgrad_pair_kernel", "mlp_wgrad_kernelILi0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        assert "mlp_wgrad_pair_kernel" in r.stdout and "(in an innermost MFMA loop: 0)" in r.stdout

