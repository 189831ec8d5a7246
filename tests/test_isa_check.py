"""tools/check_isa.py: the static check of the column-packed band kernels' hand-managed memory operations.  Those kernels
left the library in round 6 (tools/band/: a second implementation built on demand; `make -C tools/band check` runs the
check), so nothing in libta_hip.so depends on it any more; the tool and its failure modes stay tested.  Here: it passes on the code the build generated, and
it fails on the two failure modes it exists for — an inline-assembly load without its `s_nop 4`, and an
instruction touching the destination register of an inline-assembly load that is still in flight."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

CSRC = os.path.join(REPO, "tools", "band")
TOOL = os.path.join(REPO, "tools", "check_isa.py")


def run_check(*paths):
    return subprocess.run([sys.executable, TOOL, *paths], capture_output=True, text=True, timeout=600)


@pytest.fixture(scope="module")
def isa_files():
    """the device assembly of tools/band/band.hip / band32.hip (make builds it when missing)"""
    r = subprocess.run(["make", "-s", "-C", CSRC, "isa/band.s", "isa/band32.s"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    return [os.path.join(CSRC, "isa", f) for f in ("band.s", "band32.s")]


def test_library_build_passes(isa_files):
    r = run_check(*isa_files)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if "k_band" in ln and "gather" not in ln]
    assert len(lines) == 4 and all(ln.rstrip().endswith("ok") for ln in lines), r.stdout  # two float64 forms + two float32 kernels
    assert all(" 0 inline" not in ln for ln in lines)  # ... and it did look at inline loads


def test_a_removed_nop_is_caught(isa_files, tmp_path):
    for src in isa_files:
        text = open(src).read().split("\n")
        # drop the s_nop 4 of one inline load in the middle of the file
        idx = [i for i, ln in enumerate(text) if ln.strip() == "s_nop 4"]
        assert len(idx) > 4
        del text[idx[len(idx) // 2]]
        bad = tmp_path / ("bad_" + os.path.basename(src))
        bad.write_text("\n".join(text))
        r = run_check(str(bad))
        assert r.returncode == 1 and "is not preceded by its `s_nop 4`" in r.stdout, r.stdout


KERNEL = """
\t.text
_ZN2ta11k_band_demoEv:
\ts_load_dwordx4 s[0:3], s[4:5], 0x0
.LBB0_1:
\t;;#ASMSTART
\ts_nop 4
\tbuffer_load_dwordx4 v[8:11], v1, s[0:3], 0 offen
\t;;#ASMEND
\t;;#ASMSTART
\ts_nop 4
\tbuffer_load_dwordx4 v[12:15], v2, s[0:3], 0 offen
\t;;#ASMEND
{between}
\t;;#ASMSTART
\ts_waitcnt vmcnt({n})
\t;;#ASMEND
\tv_add_f64 v[20:21], v[8:9], v[10:11]
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt vmcnt(0)
\tv_add_f64 v[22:23], v[12:13], v[14:15]
\ts_endpgm
\t.section\t.rodata
"""


@pytest.mark.parametrize("between,n,ok", [
    ("\tv_mov_b32_e32 v30, v3", 1, True),             # waits for the older load only, then uses only that one
    ("\tv_mov_b32_e32 v30, v3", 2, False),            # vmcnt(2) covers nothing: v[8:11] used in flight
    ("\tv_mov_b32_e32 v30, v9", 1, False),            # a copy of a destination before the wait
    ("\tv_accvgpr_write_b32 a5, v12", 0, False),      # the "spill to an AGPR" the compiler likes
    ("\tscratch_store_dword off, v40, off", 2, True),   # a younger VMEM operation counts: vmcnt(2) now covers v[8:11]
    ("\tscratch_store_dword off, v40, off", 3, False),
])
def test_registers_in_flight(tmp_path, between, n, ok):
    f = tmp_path / "demo.s"
    f.write_text(KERNEL.format(between=between, n=n))
    r = run_check(str(f))
    assert (r.returncode == 0) == ok, r.stdout
    if not ok:
        assert "may still be in flight" in r.stdout


def test_loop_carried_load_needs_a_wait_on_the_back_edge(tmp_path):
    """v[12:15] is requested in every iteration and only waited for after the loop: the second iteration's
    request overwrites a destination whose load may still be in flight -> caught through the back edge"""
    f = tmp_path / "demo.s"
    f.write_text(KERNEL.format(between="\tv_mov_b32_e32 v30, v3", n=1))
    assert run_check(str(f)).returncode == 0  # (re-requesting is fine: loads complete in order)
    text = KERNEL.format(between="\tv_mov_b32_e32 v30, v3", n=1).replace("\ts_cbranch_scc1 .LBB0_1", "\tv_mov_b32_e32 v31, v13\n\ts_cbranch_scc1 .LBB0_1")
    f.write_text(text)
    r = run_check(str(f))
    assert r.returncode == 1 and "v13" in r.stdout
