"""tools/check_asm_loads.py (run by the build on the generated ISA) must flag any instruction that touches
the destination of an inline-asm prefetch load before the matching s_waitcnt - on every control-flow path"""
import os
import subprocess
import sys

from conftest import ROOT

TOOL = os.path.join(ROOT, "tools", "check_asm_loads.py")

GOOD = """
_Zkernel_good:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:
	;;#ASMSTART
	global_load_dwordx2 v[10:11], v2, s[0:1]
	;;#ASMEND
	v_add_f64 v[4:5], v[6:7], v[8:9]
	s_cbranch_scc1 .LBB0_3
	v_add_f64 v[4:5], v[4:5], v[8:9]
.LBB0_3:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f64 v[12:13], v[10:11], v[4:5]
	s_cbranch_scc0 .LBB0_1
	s_endpgm
.Lfunc_end0:
"""

# the load's destination is read on the taken path of a branch, which skips the wait
BAD_BRANCH = GOOD.replace("s_cbranch_scc1 .LBB0_3\n\tv_add_f64 v[4:5], v[4:5], v[8:9]\n.LBB0_3:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND",
                          "s_cbranch_scc1 .LBB0_3\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n.LBB0_3:")
# the wait sits at the top of the loop: the registers are live across the back edge and a copy reads them
BAD_BACKEDGE = """
_Zkernel_bad:
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f64 v[12:13], v[10:11], v[4:5]
	;;#ASMSTART
	global_load_dwordx2 v[10:11], v2, s[0:1]
	;;#ASMEND
	v_mov_b32_e32 v20, v10
	s_cbranch_scc0 .LBB0_1
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""
NEVER_WAITED = GOOD.replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_add_f64 v[12:13], v[10:11], v[4:5]\n", "")


def run(text, tmp_path, name):
    p = tmp_path / name
    p.write_text(text)
    r = subprocess.run([sys.executable, TOOL, str(p)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_checker_accepts_and_rejects(tmp_path):
    rc, out = run(GOOD, tmp_path, "good.s")
    assert rc == 0 and "OK" in out, out
    rc, out = run(BAD_BRANCH, tmp_path, "bad_branch.s")
    assert rc == 1 and "touches in-flight" in out, out
    rc, out = run(BAD_BACKEDGE, tmp_path, "bad_backedge.s")
    assert rc == 1 and "v_mov_b32_e32 v20, v10" in out, out
    rc, out = run(NEVER_WAITED, tmp_path, "never.s")
    assert rc == 1 and "never waited" in out, out
