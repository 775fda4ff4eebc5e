"""The compiled kernels around the generated assembly (round 5): hipcc may not carry a per-lane value across an
asm block through a copy that executes under a partial EXEC mask -- the root cause of round 4's "LDS anomaly" (the sweep's bound
parked in an LDS word read back other values: its flat address was saved into v122 with EXEC = lanes 8..11, lane 0 -- the lane
v_readfirstlane reads -- kept whatever the register held before).  tools/lint_exec_copies.py finds such copies in the ISA."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("unit", ["prot_olasave.hip", "prot_olasave8k.hip", "prot_olasave16k.hip"])
def test_shipped_kernels_carry_nothing_across_the_generated_code_under_a_partial_exec_mask(unit):
    import lint_exec_copies as L
    hits = L.lint(L.compile_asm(unit))
    assert hits == [], hits


def test_the_lint_flags_round_4s_lds_parked_bound():
    """-DPROT_SW_THR_LDS=1 restores the round-4 form (a volatile generic read of the bound from an LDS word): the compiler's save of
    its address in front of the staged block's asm sits inside the divergent gather region; =2 (the address made from a scalar at
    the point of use) is clean."""
    import lint_exec_copies as L
    bad = L.lint(L.compile_asm("prot_olasave.hip", ["-DPROT_SW_THR_LDS=1"]))
    assert len(bad) >= 1 and all("olasave_kernel" in k and "ELi3E" in k for k, _, _ in bad), bad   # (the MODE_SWEEP instantiations)
    assert any("s_or_b64 exec" in t for _, _, t in bad)
    assert L.lint(L.compile_asm("prot_olasave.hip", ["-DPROT_SW_THR_LDS=2"])) == []
