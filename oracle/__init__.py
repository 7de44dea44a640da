"""CPU oracle for the MVD hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch (CPU, fp32) restatement of the arithmetic on
the reference's denoising hot path:

  * ``sd21_unet``  -- the diffusers-0.32.2 ``UNet2DConditionModel`` forward in
    its SD-2.1 configuration (third-party code the reference calls at
    ``src/models/mvd_unet.py:318-326`` and ``src/models/image_encoder.py:105-110``;
    diffusers itself is NOT vendored in /root/reference and not installed here).
  * ``mvd``        -- ``CameraEncoder`` (``src/models/camera_encoder.py``),
    ``ImageCrossAttentionProcessor`` (``src/models/attention.py``) and the
    ``MultiViewUNet.forward`` orchestration (``src/models/mvd_unet.py:179-338``)
    including quirks Q1-Q9 of SURVEY.md section 8a.

Pinning status
--------------
* ``mvd.camera_*`` and ``mvd.image_cross_attention`` are pinned against golden
  vectors produced by importing the reference's own ``attention.py`` /
  ``camera_encoder.py`` in the authoring container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* ``sd21_unet`` is **parity unpinned** at the diffusers boundary: the reference
  holds no tests / golden vectors for it and diffusers cannot be imported here.
  It is guarded by the exact SD-2.1 parameter-count identity (865,910,724), a
  state-dict key/shape audit and shape checks only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product path
(``mvd_amd``) never imports it and has no CPU fallback.
"""
