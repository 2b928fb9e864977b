"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the Wadaboa/titanet hot path (encoder / decoder / loss heads), used
as the parity checker by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``titanet_amd/`` may import this package: the product
path is the HIP library only and fails loudly when that library is missing.
"""
