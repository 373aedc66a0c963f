"""pathtracer_amd — MI355X-native wavefront path-tracing core behind the reference's host API.

Wavefront            driver of the HIP stages (libpt_hip.so, include/pt_hip.h)
types                the host's blittable structs as numpy dtypes
"""
from .types import (CAMERA_STATE, DIELECTRIC, DIFFUSE, FRAME_STATE, MATERIAL, METAL, MIRROR, NODE, RENDER_STATE,
                    SKY_COLOR, TRIANGLE, VERTEX, material, render_state)
from .wavefront import PtError, Wavefront

__all__ = ["Wavefront", "PtError", "NODE", "TRIANGLE", "VERTEX", "MATERIAL", "RENDER_STATE", "FRAME_STATE",
           "CAMERA_STATE", "DIFFUSE", "MIRROR", "DIELECTRIC", "METAL", "SKY_COLOR", "material", "render_state"]
