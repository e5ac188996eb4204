"""CPU checkers for the hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``hevcasm_amd``) never does.

* ``oracle.pyoracle``  ctypes binding of ``liboracle.so`` (our C restatement,
  ``hevc_oracle.c``).
* ``oracle.pyref``     client for ``_ref/refdrv`` (the reference's own C files,
  built by ``oracle/Makefile`` from /root/reference where it exists).
"""
