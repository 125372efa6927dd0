"""CPU oracle for the Subtask-2C dual-encoder fine-tune step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / reported CPU baseline.
The product package never imports this module and has no CPU fallback.
"""
