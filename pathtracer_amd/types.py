"""Blittable host structs of the reference as numpy dtypes (byte-for-byte the C# layouts).

Node      PathTracer/Node.cs:5-10         Triangle  PathTracer/Triangle.cs:7-13
Vertex    PathTracer/Vertex.cs:5-11       Material  PathTracer/Material.cs:7-28
RenderState / FrameState / CameraState    PathTracer/State.cs:9-41
"""
import numpy as np

NODE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left_first", "<i4"), ("count", "<i4")])
VERTEX = np.dtype([("position", "<f4", 3), ("pad0", "<f4"), ("normal", "<f4", 3), ("pad1", "<f4")])
TRIANGLE = np.dtype([("v1", VERTEX), ("v2", VERTEX), ("v3", VERTEX), ("material", "<i4"), ("pad", "<i4", 3)])
MATERIAL = np.dtype([("color", "<f4", 4), ("emissive", "<f4", 4), ("type", "<i4"), ("index", "<f4"),
                     ("roughness", "<f4"), ("pad", "<i4")])
CAMERA_STATE = np.dtype([("p1", "<f4", 3), ("_0", "<f4"), ("p2", "<f4", 3), ("_1", "<f4"), ("p3", "<f4", 3), ("_2", "<f4"),
                         ("position", "<f4", 3), ("_3", "<f4"), ("right", "<f4", 3), ("_4", "<f4"),
                         ("up", "<f4", 3), ("_5", "<f4")])
RENDER_STATE = np.dtype([("width", "<i4"), ("height", "<i4"), ("pad", "<i4", 2), ("sky_color", "<f4", 4)])
FRAME_STATE = np.dtype([("frames", "<i4"), ("samples", "<i4"), ("focal_distance", "<f4"), ("pad", "<i4"),
                        ("camera", CAMERA_STATE)])

# pt_instance of pt_hip.h: a mesh placed by a System.Numerics Matrix4x4 (row-vector convention), its triangles' start in the flattened array
INSTANCE = np.dtype([("transform", "<f4", 16), ("mesh", "<i4"), ("first_triangle", "<i4")])

assert INSTANCE.itemsize == 72
assert NODE.itemsize == 32 and VERTEX.itemsize == 32 and TRIANGLE.itemsize == 112 and MATERIAL.itemsize == 48
assert CAMERA_STATE.itemsize == 96 and RENDER_STATE.itemsize == 32 and FRAME_STATE.itemsize == 112

# MaterialType, PathTracer/Material.cs:22-28
DIFFUSE, MIRROR, DIELECTRIC, METAL = 0, 1, 2, 3

# Game.SkyColor, PathTracer/Game.cs:13
SKY_COLOR = (142.0 / 255.0, 178.0 / 255.0, 237.0 / 255.0, 1.0)


def render_state(width, height, sky=SKY_COLOR):
    rs = np.zeros(1, RENDER_STATE)
    rs["width"], rs["height"], rs["sky_color"] = width, height, sky
    return rs


def material(color=(1, 1, 1, 0), emissive=(0, 0, 0, 0), type=DIFFUSE, index=0.0, roughness=0.0):
    m = np.zeros(1, MATERIAL)
    m["color"], m["emissive"], m["type"], m["index"], m["roughness"] = color, emissive, type, index, roughness
    return m
