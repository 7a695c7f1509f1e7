"""CPU oracle for the torch-cfd spectral hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (plain torch-CPU / torch.fft ops, the same
third-party FFT the reference itself calls) of the algorithm behind

  * the batched 2-D pseudo-spectral RK4-CN vorticity step
    (reference: torch_cfd/spectral.py:29-115, torch_cfd/equations.py:249-463,
     fno/data_gen/solvers.py:191-265), and
  * the mode-truncated spectral convolution of the FNO/SFNO layer
    (reference: fno/base.py:114-237, fno/sfno.py:331-457, fno/fno3d.py:19-116,
     fno/losses.py:263-315).

It is the *checker* the HIP path is compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``torch-cfd_amd/`` imports it: the product path fails
loudly when the HIP library is missing instead of falling back to this code.

Parity status: PINNED.  Every function here is checked against outputs of the
reference itself (imported from /root/reference inside the build container by
``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/*.npz``) by ``tests/test_oracle_golden.py``.  The reference's own
test-suite holds no golden vectors for this path.
"""
