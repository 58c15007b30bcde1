"""CPU oracle for the mclSTExp contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mclstexp_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the timed CPU baseline.
"""
from . import ref_cpu  # noqa: F401
