// abi_check.h — compile-time pin of the data contract of include/mrt_abi.h against the reference's bridging header
// (ShaderTypes.h:60-107; `vector_float3` is 16 bytes, 16-byte aligned).  Included by api.cpp, so a struct edit that moves a
// field breaks the build of libmrt_hip.so; tests/test_abi_and_host.py checks the same offsets through ctypes.
#pragma once
#include <cstddef>
#include "../../include/mrt_abi.h"

#define MRT_AT(T, field, off) static_assert(offsetof(T, field) == (off), #T "." #field " must sit at byte " #off)

static_assert(sizeof(MRTFloat3) == 16, "vector_float3 is 16 bytes");
// Camera, ShaderTypes.h:60-65
static_assert(sizeof(MRTCamera) == 64, "Camera is 64 bytes");
MRT_AT(MRTCamera, position, 0); MRT_AT(MRTCamera, right, 16); MRT_AT(MRTCamera, up, 32); MRT_AT(MRTCamera, forward, 48);
// LightType, ShaderTypes.h:67-74 (the enum quirk of :16-21: 0 is unused)
static_assert(MRTLightTypeUnused == 0 && MRTLightTypeSunlight == 1 && MRTLightTypeSpotlight == 2 && MRTLightTypePointlight == 3 && MRTLightTypeAreaLight == 4, "LightType values");
// Light, ShaderTypes.h:76-87
static_assert(sizeof(MRTLight) == 128, "Light is 128 bytes");
MRT_AT(MRTLight, type, 0); MRT_AT(MRTLight, position, 16); MRT_AT(MRTLight, color, 32); MRT_AT(MRTLight, forward, 48);
MRT_AT(MRTLight, right, 64); MRT_AT(MRTLight, up, 80); MRT_AT(MRTLight, coneAngle, 96); MRT_AT(MRTLight, direction, 112);
// Uniforms, ShaderTypes.h:89-97
static_assert(sizeof(MRTUniforms) == 96, "Uniforms is 96 bytes");
MRT_AT(MRTUniforms, width, 0); MRT_AT(MRTUniforms, height, 4); MRT_AT(MRTUniforms, blocksWide, 8); MRT_AT(MRTUniforms, frameIndex, 12);
MRT_AT(MRTUniforms, lightCount, 16); MRT_AT(MRTUniforms, camera, 32);
// Material, ShaderTypes.h:99-107
static_assert(sizeof(MRTMaterial) == 64, "Material is 64 bytes");
MRT_AT(MRTMaterial, baseColor, 0); MRT_AT(MRTMaterial, specular, 16); MRT_AT(MRTMaterial, emission, 32);
MRT_AT(MRTMaterial, specularExponent, 48); MRT_AT(MRTMaterial, refractionIndex, 52); MRT_AT(MRTMaterial, dissolve, 56);
// query records
static_assert(sizeof(MRTRay) == 32 && sizeof(MRTIntersection) == 32, "ray / intersection records are 32 bytes");
MRT_AT(MRTRay, min_distance, 12); MRT_AT(MRTRay, direction, 16); MRT_AT(MRTRay, max_distance, 28);
#undef MRT_AT
