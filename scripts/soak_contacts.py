#!/usr/bin/env python3
"""Long runs of the contact pipeline on the GPU (no oracle: sanity only): nothing goes NaN, nothing sinks through the
ground, piles settle, stacks stand.  usage: soak_contacts.py [frames=600]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from constraint_solver_amd import capi  # noqa: E402


def run(name, kind, bodies, sid, narrowphase, frames):
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_narrowphase(narrowphase)
        w.upload(bodies, sid)
        for f in range(frames):
            w.step(1.0 / 60.0, 20)
            if (f + 1) % (frames // 4) == 0:
                s = w.download()
                speed = np.linalg.norm(s[:, 22:25], axis=1)                  # Rigid::velocity
                stats = w.contact_stats()
                print("%-28s frame %4d: NaN %d  z min %.4f max %.2f  |v| median %.3g p99 %.3g max %.3g  touching/substep %.0f points %.0f"
                      % (name, f + 1, int(np.isnan(s).any(axis=1).sum()), s[:, 33].min(), s[:, 33].max(), np.median(speed),
                         np.percentile(speed, 99), speed.max(), stats[1] / (20.0 * frames / 4), stats[2] / (20.0 * frames / 4)), flush=True)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    for np_name, nph in (("sat", capi.NARROWPHASE_SAT), ("gjk_epa", capi.NARROWPHASE_GJK_EPA)):
        b, sid = capi.scene_generate(capi.SCENE_BOX_STACKS, 1, 65536)
        run("stacks_65536_" + np_name, capi.SCENE_BOX_STACKS, b, sid, nph, frames)
        b, sid = capi.scene_pile(capi.SCENE_MIXED_DROP, 1, 65536, 1.4, 4)
        run("mixed_pile_65536_" + np_name, capi.SCENE_MIXED_DROP, b, sid, nph, frames)
        b, sid = capi.scene_pile(capi.SCENE_BOXES_DROP, 1, 65536, 1.8, 4)
        run("boxes_pile_65536_" + np_name, capi.SCENE_BOXES_DROP, b, sid, nph, frames)


if __name__ == "__main__":
    main()
