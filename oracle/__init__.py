"""CPU oracle for the FaceAna hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in ``peppa_pig_face_landmark_amd/`` imports this package.  The only
legitimate importers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; everywhere it is the *checker*, never the
thing being shipped or measured as the product.

Contents (each function cites the reference file:line it restates):

* ``landmark_net``   torch-CPU restatement of the Student landmark regressor
                     (``TRAIN/face_landmark/lib/core/base_trainer/model.py``)
                     including the un-vendored timm MobileNetV3-large encoder.
* ``detector_net``   torch-CPU restatement of yolov5n-0.5 (deepcam-cn/yolov5-face,
                     not in /root/reference) + the in-graph Detect decode.
* ``prepost``        numpy restatement of ``Skps/core/api/face_detector.py`` and
                     ``face_landmark.py`` pre/post processing, with an OpenCV-style
                     fixed-point bilinear resize.
* ``synth_weights``  deterministic synthetic weights (the real ``.onnx`` blobs are
                     absent from the checkout, see ``.MISSING_LARGE_BLOBS``).
* ``ref_import``     imports the reference's own ``model.py`` (only possible in the
                     build container, where /root/reference exists) to pin the
                     restatement; used by ``tests/golden/make_golden.py``.

Parity status (see DESIGN.md section "Oracle"):
  decoder / heads / postp ......... pinned against the reference's executable source
  numpy pre/post (NMS, box math) .. pinned against the reference's executable source
  MobileNetV3 encoder (timm) ...... PARITY UNPINNED (third-party, not vendored)
  yolov5n-0.5 network ............. PARITY UNPINNED (third-party blob, absent)
  cv2.resize / copyMakeBorder ..... PARITY UNPINNED (OpenCV absent)
"""
