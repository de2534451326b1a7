"""CPU oracle for the Faster-OreFSDet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker / the timed CPU baseline -- never as the
thing shipped.  The product path (``faster-orefsdet_amd/``) must not import this
package; it fails loudly when the HIP library is missing.

Contents
--------
``ref_model.py``   plain-PyTorch fp32 restatement of the floating-point part of the
                   path (VoVNet-eSE + FPN, SM_Block, query<->support depthwise
                   correlation, CenterNet head), each function citing the reference
                   file:line it follows.
``ref_decode.c``   plain-C restatement of the integer/index part (sigmoid -> threshold
                   -> per-level top-k -> box decode -> NMS -> post-NMS top-k), built
                   into ``oracle/_build/liboracle_decode.so`` by ``oracle/Makefile``.
``decode.py``      ctypes loader for the above + a pure-numpy twin for small cases.
``refrun/``        (this container only) shim loader that executes the reference's own
                   Python files from /root/reference to pin the restatement and to
                   generate ``tests/golden/*.npz``.  Never runs on the GPU box.

Parity pinning: see the header of ``ref_model.py`` and DESIGN.md ("Oracle").
"""
