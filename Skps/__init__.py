"""Drop-in alias: ``from Skps import FaceAna`` keeps working (reference Skps/__init__.py:7-9)."""
from peppa_pig_face_landmark_amd.core.api.facer import FaceAna

__all__ = ["FaceAna"]
