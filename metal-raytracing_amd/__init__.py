"""metal-raytracing_amd — MI355X-native path tracer behind the reference's Scene / Model / Mesh /
Submesh / Renderer surface.  The render path is hand-written HIP for gfx950 reached through the C
ABI of include/mrt_abi.h (libmrt_hip.so); importing this package fails if that library is missing.

The directory name carries a hyphen (it follows the reference repository's name); import it as
`metal_raytracing_amd` (the sibling alias package re-exports this one).
"""
from ._ffi import (Camera, Float3, Intersection, Light, LightType, Material, MRTError, Ray, RenderStats, SceneStats, Uniforms, lib, LIB_PATH)
from .scene import (SCENES, CornellScene, DragonScene, GardenScene, InstancedDragonScene, IrregularDragonScene, HostileDragonScene, dragon_proxy_irregular, dragon_proxy_hostile, Mesh, Model, Scene, Submesh, flatten_scene, make_transform, load_obj, dragon_proxy, bunny_proxy)
from .renderer import Context, DeviceScene, GroupRenderer, Renderer, save_png, save_pfm, INTERSECTION_DTYPE

__all__ = ["Camera", "Float3", "Intersection", "Light", "LightType", "Material", "MRTError", "Ray", "RenderStats", "SceneStats",
           "Uniforms", "lib", "LIB_PATH", "SCENES", "CornellScene", "DragonScene", "GardenScene", "InstancedDragonScene", "IrregularDragonScene", "HostileDragonScene", "dragon_proxy_irregular", "dragon_proxy_hostile", "Mesh",
           "Model", "Scene", "Submesh", "flatten_scene", "make_transform", "load_obj", "dragon_proxy", "bunny_proxy", "Context",
           "DeviceScene", "GroupRenderer", "Renderer", "save_png", "save_pfm", "INTERSECTION_DTYPE"]
