"""Round 6: the root cause of round 5's once-in-eight-runs failure of test_concurrent_resident_selections, reproduced.

h16_select_kernel (csrc/sbq_select_win.hip) hands its remaining workgroups over to the ticket sweeps
(win_resident_rounds) when a workgroup has resigned from a resident round.  Round 5 restarted the round numbering at 2
there; the verdict words are tagged (epoch, serial, round) and never cleared, so a hand-over AFTER round 2 made every
waiter of the new "round 2" take the old round 2's verdict for its own: wrong tickets, a gather racing the flushes.

  python tools/lab/build_variant.py -DSBQ_R05_ROUND_RESTART=1      # round 5's numbering -> tools/lab/libsbq_variant.so
  SBQ_LIB=tools/lab/libsbq_variant.so python tools/lab/r06_roundtag_repro.py   # mismatches with knob 2 = 32
  python tools/lab/r06_roundtag_repro.py                            # the product library: none

knob 2 = 32: every waiting workgroup resigns half a microsecond into its wait from round 2 on, never in round 1.
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from sparsebit_amd import lib as L  # noqa: E402

if os.environ.get("SBQ_LIB"):
    L.LIB_PATH = os.environ["SBQ_LIB"]
from sparsebit_amd import ops  # noqa: E402
import test_gpu_r06 as T  # noqa: E402

print("library:", L.LIB_PATH, flush=True)
for knob in (0, 31, 32, 33):
    total = bad = 0
    first = None
    for n in (4096 * 4096, 3 * 1024 * 1024 + 5):
        for name, x in T._many_round_data(n, 72).items():
            script = T._selection_script(x)
            want = T._script_reference(x, script)
            xd = x.cuda()
            L.set_tuning(2, knob)
            for rep in range(3):
                for j in range(len(script)):
                    got = T._run_script(ops, xd, script[j:j + 1])[0].reshape(-1).tolist()
                    total += 1
                    if got != want[j]:
                        bad += 1
                        if first is None:
                            first = (name, n, script[j], got, want[j], T._select_state_words(xd.device))
            L.set_tuning(2, 0)
    print("knob 2 = %2d: %d selections, %d mismatches%s" % (knob, total, bad, "" if first is None else "  first: %r" % (first,)), flush=True)
