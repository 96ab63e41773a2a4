"""CPU oracle for the chart inside-outside hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cliora_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / timed baseline.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference (``/root/reference/cliora``) in the build container with
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks every
function here against them.  The one exception is the TreeLSTM composition
(BASELINE config 5): the reference ships it only as commented-out text
(cliora/net/vg.py:28-76), so that part is "parity unpinned".
"""
