// host_geometry.h — host-side mesh container shared by the OBJ reader, the procedural meshes
// and the C ABI (MRTMeshData).
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "../../include/mrt_abi.h"

namespace mrt {

struct Submesh {                       // SubMesh.swift:10-33
    std::string name;                  // usemtl name
    std::vector<uint32_t> indices;     // 3 per triangle, into the mesh's vertex arrays
    MRTMaterial material;
};

struct MeshData {                      // what MDLMesh/MTKMesh hold for the hot path (Mesh.swift:10-33)
    std::vector<float> positions;      // packed xyz
    std::vector<float> normals;        // packed xyz
    std::vector<Submesh> submeshes;
    bool generated_normals = false;
};

bool load_obj(const std::string &path, MeshData &out);
void make_dragon_proxy(MeshData &out);
void make_dragon_proxy_irregular(MeshData &out);   // same count, extents and material; irregular connectivity, shuffled order
void make_dragon_proxy_hostile(MeshData &out);     // same count / extents / material; 100 : 1 triangle sizes and 1 % slivers up to 50 x their edge
void make_bunny_proxy(MeshData &out);
void make_transform(const float position[3], const float rotation[3], float scale, float out16[16]);
void default_camera(int width, int height, MRTCamera *out);

}  // namespace mrt
