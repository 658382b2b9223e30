"""A reduced, fixed-seed selection of the randomised differential cases of scripts/fuzz_pinned.py and
scripts/fuzz_contacts.py inside `pytest -m gpu` (the long runs stay in scripts/): random scene kind, body count,
substeps, dt, schedule, block size / narrowphase, pre-test schedule, pad, pile width, joints -- GPU == oracle in every
bit.  The contact pipeline is the EXTENSION (parity unpinned: the oracle is the build's own)."""
import numpy as np
import pytest

from fuzz_cases import contacts_case, pinned_case

pytestmark = pytest.mark.gpu


def test_fuzz_pinned_path_against_the_oracle():
    rng = np.random.default_rng(20260402)
    with_contacts = 0
    for case in range(15):
        ok, what = pinned_case(rng, max_bodies=4000)
        assert ok, "case %d: %s" % (case, what)
        with_contacts += "contacts 0" not in what
    assert with_contacts >= 10                       # the selection does exercise ground contacts


def test_fuzz_contact_pipeline_against_its_oracle():
    rng = np.random.default_rng(20260403)
    touching = 0
    for case in range(10):
        ok, what = contacts_case(rng, max_bodies=1200)
        assert ok, "case %d: %s" % (case, what)
        touching += "touching 0 " not in what
    assert touching >= 8                             # ... and body-body contacts


def test_fuzz_contact_pipeline_with_narrow_groups(monkeypatch):
    """The same kind of cases with XPBD_SAT_WIDE_PAIRS=1: every SAT launch takes the narrow groups that only large launches get
    otherwise -- eight lanes per box pair, and four in dense scenes (16 pairs per wave, the clipper two polygon vertices per
    lane): all group widths must give the bits of the oracle."""
    monkeypatch.setenv("XPBD_SAT_WIDE_PAIRS", "1")
    rng = np.random.default_rng(20260405)
    touching = 0
    for case in range(12):
        ok, what = contacts_case(rng, max_bodies=1500)
        assert ok, "case %d: %s" % (case, what)
        touching += "touching 0 " not in what
    assert touching >= 9
