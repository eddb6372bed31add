"""CPU oracle for the PAOS wavefront-propagation hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm
(arielmission-space/PAOS v1.2.12: paos/core/run.py, paos/classes/wfo.py,
paos/classes/zernike.py, paos/classes/abcd.py, paos/core/coordinateBreak.py).
Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker*, never as the thing
that is measured or shipped.  Nothing under ``paos_amd/`` imports it; the
product path fails loudly if the HIP library is missing.

Pinning (see tests/test_oracle_golden.py and DESIGN.md section 3):
  * everything except aperture-mask VALUES is pinned to golden vectors generated
    by importing the reference itself in the build container
    (tests/golden_tools/make_golden.py -> tests/golden/*.npz) plus the reference's own
    notebook known-answers;
  * aperture-mask values come from photutils 1.11.0 (poetry.lock:2360), which is
    absent from /root/reference and from this image: that boundary is
    "parity unpinned" and is anchored on analytic properties instead
    (oracle/aperture_np.py header).
"""
