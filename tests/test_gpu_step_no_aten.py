"""no torch (aten) operator may launch anything inside a captured training step -- every kernel of the
step is one of the library's own (tools/aten_in_step.py is the same check as a tool, with the source line of each offender)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,extra", [("cfg1", ["--batch", "30"]), ("cfg3", ["--graphs", "32"]), ("cfg4", ["--graphs", "2048", "--batch", "256"]),
                                          ("cfg5", ["--graphs", "512"])])
def test_captured_step_launches_no_torch_operator(config, extra):
    import aten_in_step
    wl, step = aten_in_step.build(config, extra, graphed=True)
    seen = aten_in_step.log_step(step)
    assert not seen, "torch operators inside the captured step of %s: %s" % (
        config, ["%d x %s %s <- %s" % (n, k[0], k[2], k[1]) for k, n in seen.items()])
