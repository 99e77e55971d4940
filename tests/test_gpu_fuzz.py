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

CASES = [(101, False, False, False), (102, False, False, False), (103, False, False, False), (104, False, False, False), (105, False, False, False),
         (201, True, False, False), (202, True, False, False), (203, True, False, False), (301, False, True, False), (302, False, True, False)]
# scores drawn from the whole region the reference accepts (fuzz_parity.draw_scores): gap-friendly, zero, D = M, 1e5, negative match
CASES += [(s, False, False, True) for s in (401, 402, 403, 404, 405, 406, 407, 408, 409, 410, 411, 412)] + [(s, True, False, True) for s in (501, 502, 503, 504)]
# seed 4028 (colored, scores 0.5 / 0 / -4.5, cutoffs 15 / 70): a site whose first strings fail their range test and whose LATER string lies on
# no unitig of the bubble -- the reference never looks that string up (its walk over a site's strings ends at the first failure,
# src/CCDBG.cpp:3262-3276), so it is no error; K-SITES looks all strings up side by side and must judge in the walk's order
CASES += [(4028, True, False, True)]


@pytest.mark.parametrize("seed,colored,giant,wide", CASES)
def test_random_inputs_give_the_oracles_files(seed, colored, giant, wide):
    import fuzz_parity
    import pyoracle
    if giant and not os.path.exists(pyoracle.REF_BIFROST):
        pytest.skip("oracle/_ref/Bifrost (the reference's graph builder) is not on this box")
    if colored and not os.path.exists(pyoracle.REF_COLORS_DUMP):
        pytest.skip("oracle/_ref/colors_dump is not on this box")
    pyoracle.build()
    for attempt in range(4):   # (a seed whose repeats this repository's graph builder cannot compact is replaced by the next one)
        with tempfile.TemporaryDirectory() as tmp:
            msg = fuzz_parity.one_case(seed + 1000 * attempt, tmp, "cuda", force_colored=colored, force_giant=giant, wide_scores=wide)
        if not msg.startswith("skipped"):
            break
    assert ("colored" in msg) == colored or "skipped" in msg, msg
    if wide and "oracle rc" in msg:
        # both sides ended the run themselves (a k-mer of a site string that is in no database, src/CDBG.cpp:52-56): agreement
        import re
        mm = re.search(r"oracle rc (-?\d+), product rc (-?\d+)", msg)
        assert mm and int(mm.group(1)) != 0 and int(mm.group(2)) != 0 and "MISMATCH" not in msg, msg
        return
    assert msg.endswith("identical"), msg
    if giant:   # a traversal that outgrew the 128-entry LDS tier of K-BFS: walked by the third tier
        import re
        mm = re.search(r"largest=(\d+)", msg)
        assert mm and int(mm.group(1)) > 128, msg
