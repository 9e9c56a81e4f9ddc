"""CPU oracle for the detection hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference detector's
arithmetic (NumPy / torch-CPU / plain C).  It exists to check the HIP path and
to time a CPU baseline.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``facedetection_amd`` never does and fails loudly without its HIP library.

Parity status per module is stated in each module's header.
"""
