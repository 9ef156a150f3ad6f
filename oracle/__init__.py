"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's PointNet hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package, and only as the checker / reported baseline.  The product
(``pointcloudprocessing_amd``) never imports it and has no CPU fallback.

PARITY UNPINNED for the numerical model path: the reference (TensorFlow 2.20 / Keras 3.10,
``/root/reference/point_cloud_analysis``) cannot be imported in this image (``ModuleNotFoundError:
tensorflow`` -- an ordinary missing module, not a permission denial), ships no trained weights and has
no tests or golden vectors for the model.  What IS pinned against reference-held
data: the Aftr frame parser + normalisation on the two labelled clouds the reference ships, the
parameter census (4,210,476 / 14,208) and the trainability-name list logged by the reference's own run.
"""
