"""CPU oracle for the SGTAPose inference hot path  --  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (torch-CPU fp32 / numpy fp64), the algorithm of the
reference's hot path.  It exists so that tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg can *check* the HIP path; nothing in ``sgtapose_amd/``
(the product) imports it, and the product fails loudly when the HIP library is missing.

Parity status
-------------
* Everything except the deformable convolution is PINNED: ``tests/golden/make_golden.py``
  imports the reference's own Python files in the build container (with import stubs) and
  stores input/output vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
  this package against them.
* ``oracle/prepost.py`` (prior rendering, post-processing) is pinned the same way; only its
  3-point affine solve stands in for ``cv2.getAffineTransform`` (OpenCV absent: unpinned, see that file).
* The deformable convolution (DCNv2) lives in an un-vendored third-party dependency of the
  reference (github.com/lbin/DCNv2, branch ``pytorch_1.9``; reference README.md:21-28, import
  site sgtapose/lib/model/networks/dla.py:21-25, only call site dla.py:545-548).  Its source
  is not under /root/reference and the reference holds no test for it, so for that one
  operator the oracle restates the published algorithm (``oracle/dcnv2.py``, and a plain-C twin
  ``oracle/dcnv2_ref.c``): **parity unpinned** for DCNv2 itself; all end-to-end golden vectors
  were produced with the reference network calling this restatement as its ``DCN``.
"""
from .dcnv2 import dcn_v2_forward, dcn_v2_im2col          # noqa: F401
from .dla34 import dla34_backbone, dla_up_forward, ida_up_forward  # noqa: F401
from .temporal import (topk_index, window_coords, window_gather, mhca_forward,   # noqa: F401
                       encoder_layer_forward, encoder_forward, substitute_tokens)
from .decode import (gaussian_filter_f32, peaks_from_maps, peaks_info, dream_decode,  # noqa: F401
                     soft_argmax_pavlo)
from .forward import (plan_a_window_forward, plan_a_window_three_forward, plan_a_window_seq_forward,   # noqa: F401
                      seq_keys_from_three, process_reference)
from . import prepost                                                   # noqa: F401  (prior rendering, post-processing)
