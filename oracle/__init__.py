"""CPU oracle for the MARS alignment-and-ranking path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 / numpy CPU
restatement of the reference algorithm, used as the checker by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  The
shipped GPU path never imports it.

Pinning status
--------------
* ViT towers (DINOv2, CLIP, AlphaCLIP), GradCAM and the evaluator are pinned
  against the reference modules themselves, imported from /root/reference in the
  build container by ``tests/golden/make_golden.py``; the resulting vectors are
  committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
* Pieces whose arithmetic lives in third-party wheels that are absent here
  (POT 0.9.4 ``ot.emd2``, opencv-python 4.10.0.84 contours, torchvision
  transforms) have no reference-run vectors: the optimum of the transport LP is
  pinned against scipy's HiGHS solver, the box extraction against
  ``scipy.ndimage.label``, the resampling against ``torch.nn.functional.interpolate``
  and PIL.  Those three are "parity unpinned" with respect to the reference
  wheels and say so where they are defined.
"""
