"""A slice of the differential fuzzing of tools/fuzz_parity.py inside the GPU suite: random inputs (k, ploidy, -z, variant density,
insertion lengths, repeat-rich genomes, integral and fractional scores, tight and loose cut-offs) through the product's CLI and
through the pinned oracle (oracle/_build/pf_oracle_cli; reference src/CDBG.cpp, src/CCDBG.cpp, src/SeqAlign.cpp) -- all twelve
files byte-identical.  Ten seeds: five single-sample, three colored, two with chromosome-long traversals (graphs of the
reference's own `Bifrost build`, which travels with the snapshot under oracle/_ref)."""
import os
import sys
import tempfile

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

CASES = [(101, False, False), (102, False, False), (103, False, False), (104, False, False), (105, False, False),
         (201, True, False), (202, True, False), (203, True, False), (301, False, True), (302, False, True)]


@pytest.mark.parametrize("seed,colored,giant", CASES)
def test_random_inputs_give_the_oracles_files(seed, colored, giant):
    import fuzz_parity
    import pyoracle
    if giant and not os.path.exists(pyoracle.REF_BIFROST):
        pytest.skip("oracle/_ref/Bifrost (the reference's graph builder) is not on this box")
    if colored and not os.path.exists(pyoracle.REF_COLORS_DUMP):
        pytest.skip("oracle/_ref/colors_dump is not on this box")
    pyoracle.build()
    for attempt in range(4):   # (a seed whose repeats this repository's graph builder cannot compact is replaced by the next one)
        with tempfile.TemporaryDirectory() as tmp:
            msg = fuzz_parity.one_case(seed + 1000 * attempt, tmp, "cuda", force_colored=colored, force_giant=giant)
        if not msg.startswith("skipped"):
            break
    assert ("colored" in msg) == colored or "skipped" in msg, msg
    assert msg.endswith("identical"), msg
    if giant:   # a traversal that outgrew the 128-entry LDS tier of K-BFS: walked by the third tier
        import re
        mm = re.search(r"largest=(\d+)", msg)
        assert mm and int(mm.group(1)) > 128, msg
