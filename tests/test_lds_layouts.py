"""The LDS layouts of k_kpt.hip against the bank model of gfx950's LDS instructions (scripts/probes/lds_bank_sim.py: lane groups
and bank functions from MI355X_MICROARCH.md applied to every address the kernel issues).  Pins the two layouts the kernel ships
with as conflict-free for their fragment reads, and records what the first layouts cost."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("lds_bank_sim", os.path.join(os.path.dirname(__file__), "..", "scripts", "probes", "lds_bank_sim.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


def test_kpt3_fragment_reads_are_conflict_free_in_the_bank_model():
    assert sim.kpt3_stage1_blocks() == 1.0
    assert sim.kpt3_stage2(True, 32) == 1.0


def test_the_layouts_that_were_replaced_are_not():
    assert sim.kpt3_stage1_linear(144) >= 2.0 and sim.kpt3_stage1_linear(160) >= 1.5
    assert sim.kpt3_stage2(False, 32) > 1.5


def test_a_broadcast_is_not_a_conflict_and_a_stride_is():
    assert sim.read_b128([0] * 64) == 1.0                       # identical addresses broadcast
    assert sim.read_b128([lane * 256 for lane in range(64)]) == 16.0   # every lane of a group on the same four banks
    b128, b64 = sim.c2f2_stores()
    assert b128 == 2.0 and b64 == 4.0                           # the epilogue stores DESIGN.md 4g puts at <= 9 % of c2f2's time
