"""peppa_pig_face_landmark_amd -- MI355X-native FaceAna hot path (HIP/CDNA4 kernels behind a C ABI).

    from peppa_pig_face_landmark_amd import FaceAna     # or: from Skps import FaceAna
    result = FaceAna().run(image_bgr)
"""
__all__ = ["FaceAna"]


def __getattr__(name):
    if name == "FaceAna":
        from .core.api.facer import FaceAna
        return FaceAna
    raise AttributeError(name)
