"""The build guard of the hand-managed AGPR stash (scripts/check_agpr.py, run by csrc/build.sh on the generated assembly): it must
see EVERY instruction of a kernel -- also those behind an early `s_endpgm` (round 3: the shared-element kernels return early when
an exchange times out; scanning up to the first `s_endpgm` had hidden the whole reverse pass) -- and tell compiler-allocated
registers (`aN`, `a[N:M]`) from the hand-managed ones (printed by the inline asm as `a[0x..]` / `a[N]`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = """\t.text
_Z6k_testILi3EEv8MfmaArgs:
\tv_accvgpr_write_b32 a[200], v1
\tv_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[4:5], a[0:7]
\ts_cbranch_scc1 .LBB0_2
\ts_endpgm
.LBB0_2:
\tv_accvgpr_read_b32 v9, a{hi}
\tv_mfma_f64_16x16x4_f64 a[16:23], v[2:3], v[4:5], a[16:23]
\ts_endpgm
.Lfunc_end0:
_Z7k_otherv:
\tv_accvgpr_read_b32 v9, a250
\ts_endpgm
.Lfunc_end1:
"""


def _run(tmp_path, hi, base):
    f = tmp_path / "k.s"
    f.write_text(ASM.format(hi=hi))
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_agpr.py"), str(f), "k_testILi3", str(base)],
                          capture_output=True, text=True)


def test_guard_sees_code_behind_an_early_return(tmp_path):
    ok = _run(tmp_path, 90, 106)
    assert ok.returncode == 0 and "a90" in ok.stdout, ok.stdout + ok.stderr
    bad = _run(tmp_path, 120, 106)          # the offending register sits BEHIND the first s_endpgm
    assert bad.returncode != 0 and "a120" in (bad.stdout + bad.stderr)


def test_guard_ignores_other_kernels_and_hand_managed_operands(tmp_path):
    ok = _run(tmp_path, 23, 24)             # a[16:23] is the highest compiler register; a[200] (hand-managed) and k_other's a250 do not count
    assert ok.returncode == 0 and "a23" in ok.stdout, ok.stdout + ok.stderr
    assert _run(tmp_path, 23, 23).returncode != 0


def test_guard_tells_a_trip_from_a_check_that_could_not_run(tmp_path):
    """exit 1 = overlap (build.sh builds without the kernel), exit 2 = symbol / file missing (build.sh must FAIL: advisor, round 3)."""
    assert _run(tmp_path, 120, 106).returncode == 1
    f = tmp_path / "k.s"
    f.write_text(ASM.format(hi=90))
    script = os.path.join(ROOT, "scripts", "check_agpr.py")
    renamed = subprocess.run([sys.executable, script, str(f), "k_renamedILi3", "106"], capture_output=True, text=True)
    assert renamed.returncode == 2 and "not found" in renamed.stderr
    nofile = subprocess.run([sys.executable, script, str(tmp_path / "missing.s"), "k_testILi3", "106"], capture_output=True, text=True)
    assert nofile.returncode == 2 and "cannot read" in nofile.stderr


def test_margins_of_the_built_kernels():
    """The hand-managed AGPR ranges of the library as BUILT here: every guarded instantiation keeps at least 6 registers between the
    compiler's high-water mark and the stash (verdict round 4, item 5: a two-register margin would let a compiler point release
    move the headline kernel to its fallback plan silently).  Reads the assembly csrc/build.sh leaves beside the objects; skipped
    when the library was built elsewhere."""
    import re
    import pytest
    csrc = os.path.join(ROOT, "hp_vpinns_amd", "csrc")
    S20, S16, S12, T = "ELi20ELi20ELi10ELi10E", "ELi16ELi16ELi8ELi8E", "ELi12ELi12ELi6ELi6E", "ELi80ELi80ELi5ELi5"
    guarded = {"kernels_fused.s": [("k_iter_fusedILi3ELb0ELb1ELb0" + S20, 106), ("k_iter_fusedILi3ELb0ELb0ELb0" + S20, 106),
                                   ("k_iter_fusedILi3ELb1ELb0ELb0" + S20, 106), ("k_iter_fusedILi2ELb0ELb1ELb0" + S20, 156),
                                   ("k_iter_fusedILi3ELb0ELb1ELb0" + S16, 166), ("k_iter_fusedILi3ELb0ELb1ELb0" + S12, 226)],
               # round 6, the general forms (template tail <.., MULTI = false, NT2, GEN = true>): three channels on the one-hot kernels' stash,
               # four channels with one more tile per wave in LDS (the stash starts 2 L x 5 registers higher)
               "kernels_fused_gen.s": [("k_iter_fusedILi3ELb0ELb1ELb0" + S20 + "Lb0ELi0ELb1E", 106), ("k_iter_fusedILi3ELb0ELb0ELb0" + S20 + "Lb0ELi0ELb1E", 106),
                                       ("k_iter_fusedILi3ELb1ELb0ELb0" + S20 + "Lb0ELi0ELb1E", 106), ("k_iter_fusedILi3ELb0ELb1ELb0" + S16 + "Lb0ELi0ELb1E", 166),
                                       ("k_iter_fusedILi3ELb0ELb1ELb0" + S16 + "Lb0ELi1ELb1E", 196), ("k_iter_fusedILi3ELb1ELb0ELb0" + S16 + "Lb0ELi1ELb1E", 196),
                                       ("k_iter_fusedILi2ELb0ELb0ELb0" + S20 + "Lb0ELi1ELb1E", 176),
                                       # the tight plan (FzPlan): three of the first stash place's fifteen doubles in registers, twelve in LDS
                                       ("k_iter_fusedILi3ELb0ELb0ELb0" + S20 + "Lb0ELi1ELb1E", 160)],
               "kernels_tall.s": [("k_iter_tallILi2ELi1ELi3" + T + "ELb0", 136), ("k_iter_tallILi2ELi1ELi3" + T + "ELb1", 166),
                                  ("k_iter_tallILi2ELi0ELi3" + T + "ELb1", 166)]}
    if not all(os.path.exists(os.path.join(csrc, f)) for f in guarded):
        pytest.skip("no assembly beside the objects (library built elsewhere)")
    script = os.path.join(ROOT, "scripts", "check_agpr.py")
    for f, ks in guarded.items():
        for key, base in ks:
            r = subprocess.run([sys.executable, script, os.path.join(csrc, f), key, str(base)], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            hi = int(re.search(r"AGPR a(\d+),", r.stdout).group(1))
            assert base - hi >= 6, (key, hi, base)


def test_no_whole_iteration_kernel_spills_to_scratch():
    """Round 6: one more pointer kept alive across the phases of k_iter_fused<.., NT2 = 1> sent the register allocator to scratch
    memory -- 420 - 540 scratch accesses per instantiation, correct results, 148 instead of 58 us per iteration, and nothing said so.
    check_agpr.py now counts them (exit 3, build.sh fails); here: NO instantiation of the hand-scheduled whole-iteration kernels in
    the assembly csrc/build.sh left beside the objects touches scratch memory."""
    import re
    import pytest
    csrc = os.path.join(ROOT, "hp_vpinns_amd", "csrc")
    files = ["kernels_fused.s", "kernels_fused_gen.s", "kernels_tall.s"]
    if not all(os.path.exists(os.path.join(csrc, f)) for f in files):
        pytest.skip("no assembly beside the objects (library built elsewhere)")
    n_kernels, spills = 0, {}
    for f in files:
        lines = open(os.path.join(csrc, f)).read().split("\n")
        name = None
        for l in lines:
            m = re.match(r"^(_Z\d+k_iter_(fused|tall|small)\w+):", l)
            if m:
                name, n_kernels = m.group(1), n_kernels + 1
            elif l.startswith(".Lfunc_end"):
                name = None
            elif name and re.search(r"\bscratch_(load|store)_", l.split(";")[0]):
                spills[name] = spills.get(name, 0) + 1
    assert n_kernels >= 40, n_kernels
    # (k_iter_small<3> -- eight waves of 256 registers, config 3 -- has carried seven spilled quad-words since round 3: 14 accesses
    #  outside its tile loops; anything beyond that, or any other kernel, is a regression)
    known = {k: v for k, v in spills.items() if "k_iter_smallILi3E" in k and v <= 16}
    assert spills == known, {k: v for k, v in spills.items() if k not in known}


def test_guard_reports_spills(tmp_path):
    f = tmp_path / "k.s"
    f.write_text(ASM.format(hi=90).replace("\ts_cbranch_scc1 .LBB0_2", "\tscratch_store_dwordx2 off, v[2:3], off offset:8\n\ts_cbranch_scc1 .LBB0_2"))
    script = os.path.join(ROOT, "scripts", "check_agpr.py")
    r = subprocess.run([sys.executable, script, str(f), "k_testILi3", "106"], capture_output=True, text=True)
    assert r.returncode == 3 and "spills" in r.stderr
    r = subprocess.run([sys.executable, script, str(f), "k_testILi3", "106", "--spills-ok"], capture_output=True, text=True)
    assert r.returncode == 0 and "SCRATCH" in r.stdout
