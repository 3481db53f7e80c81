"""CPU oracle for the MLPG / DTW hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``nnmnkwii_amd/`` imports this package.  It is imported by
``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s
``cpu_baseline`` leg, and only as the checker / reported baseline.

* ``oracle.mlpg``   : C restatement of the reference's numpy/bandmat/Cython
  forward path (``mlpg_oracle.c``) plus numpy restatements of ``mlpg_grad``,
  ``unit_variance_mlpg_matrix`` and ``reshape_means``.  Parity PINNED against
  golden vectors generated from the reference itself (``tests/golden``).
* ``oracle.dtw``    : restatement of ``DTWAligner.transform`` + fastdtw
  (third-party, absent from /root/reference).  Parity UNPINNED -- see the
  module header.
"""
from . import mlpg  # noqa: F401
