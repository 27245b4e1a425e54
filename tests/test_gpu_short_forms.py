"""The sphere-list kernels' short forms -- quotients over a shared reciprocal, square roots and unit vectors without
the rescaling wrappers, the integer test of the unit-sphere sampler, draws in (-1, 1) by one conversion -- against the
operations they stand for, evaluated on the device by tools/ubench/short_roots (built by __graft_entry__.build() from
the kernels' own headers): 2^24 operands each over the whole allowed exponent range, the ends of the ranges, triples
that land inside the band where the integer test hands over to the doubles, and all 2^31 draws.  Bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "ubench", "short_roots")


@pytest.mark.gpu
def test_short_forms_agree_with_the_operations_they_stand_for():
    if not os.path.exists(EXE):
        pytest.skip("tools/ubench/short_roots is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "all short forms agree bit for bit", r.stdout
    assert len(lines) == 6, r.stdout
