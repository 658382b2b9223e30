"""State history on the device (xpbd_world_history_*): the reference app keeps every simulated frame in
`states: Vec<(World, DebugLines)>` and scrubs through it with `current_state` (src/app.rs:48, 206-212).
A restored state must be the state that was pushed, bit for bit, and stepping on from it must reproduce the
original run -- in the reference path and in the contact pipeline."""
import os
import subprocess

import numpy as np
import pytest

from constraint_solver_amd import capi

pytestmark = pytest.mark.gpu
DT = 1.0 / 60.0


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


def make_world(mode, kind, n):
    bodies, sid = capi.scene_generate(kind, 4, n)
    w = capi.World(mode=mode)
    if mode == capi.MODE_CONTACTS:
        w.set_polytopes(capi.scene_polytopes(kind))
    else:
        w.set_shapes(*capi.scene_shapes(kind))
    w.upload(bodies, sid)
    return w, bodies, sid


@pytest.mark.parametrize("mode,kind,n", [(capi.MODE_FUSED, capi.SCENE_BOXES, 3000), (capi.MODE_PER_SUBSTEP, capi.SCENE_MIXED, 700),
                                         (capi.MODE_CONTACTS, capi.SCENE_BOX_STACKS, 1024)])
def test_restore_gives_back_the_pushed_state_and_replays_bit_for_bit(mode, kind, n):
    w, bodies, sid = make_world(mode, kind, n)
    with w:
        assert w.history_length() == 0
        assert w.history_push() == 0                                   # states = vec![World::new(..)]
        seen, contacts = [w.download()], [None]
        for frame in range(1, 6):
            w.step(DT, 10)
            assert w.history_push() == frame
            seen.append(w.download())
            contacts.append(w.contacts())
        assert w.history_length() == 6 and bits_equal(seen[0], bodies)
        for index in (2, 0, 5, 3):                                     # scrubbing back and forth
            w.history_restore(index)
            assert bits_equal(w.download(), seen[index])
            if index:
                assert np.array_equal(w.contacts(), contacts[index])   # the contact list belongs to the state
        # branch off state 2: drop the later states, step on, and get the original frames 3..5 again
        w.history_restore(2)
        w.history_truncate(3)
        assert w.history_length() == 3
        for frame in range(3, 6):
            w.step(DT, 10)
            assert w.history_push() == frame
            assert bits_equal(w.download(), seen[frame])
            assert np.array_equal(w.contacts(), contacts[frame])


def test_history_errors_and_growth():
    w, bodies, sid = make_world(capi.MODE_FUSED, capi.SCENE_BOXES, 100)
    with w:
        with pytest.raises(capi.XpbdError):
            w.history_restore(0)                                       # nothing pushed yet
        states = []
        for k in range(40):                                            # beyond the first block of 8: the history grows
            assert w.history_push() == k
            states.append(w.download())
            w.step(DT, 3)
        for k in (0, 7, 8, 23, 39):
            w.history_restore(k)
            assert bits_equal(w.download(), states[k])
        with pytest.raises(capi.XpbdError):
            w.history_restore(40)
        with pytest.raises(capi.XpbdError):
            w.history_truncate(41)
        w.upload(bodies, sid)                                          # a new upload starts a new history
        assert w.history_length() == 0


def test_headless_timeline_rewind_equals_a_shorter_run(tmp_path):
    exe = os.path.join(capi.LIB_DIR, "xpbd_headless")
    common = ["--bodies", "2048", "--substeps", "10", "--scene", "boxes-drop"]
    a, b = tmp_path / "rewound.bin", tmp_path / "short.bin"
    p = subprocess.run([exe] + common + ["--frames", "5", "--history", "--rewind", "2", "--dump", str(a)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    p = subprocess.run([exe] + common + ["--frames", "2", "--warmup", "0", "--dump", str(b)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert a.read_bytes() == b.read_bytes() and a.stat().st_size == 2048 * 304
