// Scene graph mirror over the C ABI.  Names and initialiser arguments follow the reference's classes
// (Scene.swift, DragonScene.swift, Model.swift, Mesh.swift, SubMesh.swift); geometry ingest is done by the
// library (mrt_obj_load / mrt_make_transform) instead of ModelIO.
import CMRT
import Foundation

public struct MRTError: Error { public let code: Int32; public let message: String }
@inline(__always) func check(_ rc: Int32) throws {
    if rc != 0 { throw MRTError(code: rc, message: String(cString: mrt_last_error())) }
}
func f3(_ v: SIMD3<Float>) -> MRTFloat3 { MRTFloat3(x: v.x, y: v.y, z: v.z, _pad: 0) }

public typealias Camera = MRTCamera
public typealias Material = MRTMaterial
public typealias Light = MRTLight

public extension MRTLight {
    static func areaLight(position: SIMD3<Float>, forward: SIMD3<Float>, right: SIMD3<Float>, up: SIMD3<Float>, color: SIMD3<Float>) -> MRTLight {
        var l = MRTLight(); l.type = Int32(MRTLightTypeAreaLight.rawValue)
        l.position = f3(position); l.forward = f3(forward); l.right = f3(right); l.up = f3(up); l.color = f3(color); return l
    }
    static func sunLight(direction: SIMD3<Float>, color: SIMD3<Float>) -> MRTLight {
        var l = MRTLight(); l.type = Int32(MRTLightTypeSunlight.rawValue); l.direction = f3(direction); l.color = f3(color); return l
    }
    static func pointLight(position: SIMD3<Float>, color: SIMD3<Float>) -> MRTLight {
        var l = MRTLight(); l.type = Int32(MRTLightTypePointlight.rawValue); l.position = f3(position); l.color = f3(color); return l
    }
    static func spotLight(position: SIMD3<Float>, direction: SIMD3<Float>, coneAngle: Float, color: SIMD3<Float>) -> MRTLight {
        var l = MRTLight(); l.type = Int32(MRTLightTypeSpotlight.rawValue)
        l.position = f3(position); l.direction = f3(direction); l.coneAngle = coneAngle; l.color = f3(color); return l
    }
}

public struct Submesh { public let name: String; public let indices: [UInt32]; public var material: Material }

public struct Mesh {
    public let positions: [Float], normals: [Float]        // packed xyz
    public let transform: [Float]                          // 16 floats, column-major T*R*S
    public let submeshes: [Submesh]
}

public final class Model {
    public static var resourceDirectory = "assets/Resources"
    public var meshes: [Mesh] = []
    public let name: String                                // the resource it was loaded from (Renderer(instancing:) shares meshes by it)
    public init(name: String, position: SIMD3<Float>, rotation: SIMD3<Float> = [0, 0, 0], scale: Float) throws {
        self.name = name
        var md: MRTMeshData?
        let rc = mrt_obj_load("\(Model.resourceDirectory)/\(name).obj", &md)
        if rc == MRT_ERR_IO.rawValue && name == "dragon" { try check(mrt_dragon_proxy(&md)) }
        else if rc == MRT_ERR_IO.rawValue && name == "bunny" { try check(mrt_bunny_proxy(&md)) }
        else { try check(rc) }
        defer { mrt_meshdata_free(md) }
        var nv = 0; var ns: Int32 = 0
        try check(mrt_meshdata_counts(md, &nv, &ns))
        var pos = [Float](repeating: 0, count: nv * 3), nrm = pos
        try check(mrt_meshdata_vertices(md, &pos, &nrm))
        var subs: [Submesh] = []
        for s in 0..<ns {
            var nt = 0
            try check(mrt_meshdata_submesh(md, s, &nt, nil, nil, nil, 0))
            var idx = [UInt32](repeating: 0, count: nt * 3); var mat = MRTMaterial()
            var buf = [CChar](repeating: 0, count: 256)
            try check(mrt_meshdata_submesh(md, s, &nt, &idx, &mat, &buf, 256))
            subs.append(Submesh(name: String(cString: buf), indices: idx, material: mat))
        }
        var xf = [Float](repeating: 0, count: 16)
        var p = [position.x, position.y, position.z], r = [rotation.x, rotation.y, rotation.z]
        try check(mrt_make_transform(&p, &r, scale, &xf))
        meshes = [Mesh(positions: pos, normals: nrm, transform: xf, submeshes: subs)]
    }
}

open class Scene {
    public var models: [Model] = []
    public var camera: Camera
    public var lights: [Light]
    public init(width: Int, height: Int) throws {
        camera = try Scene.setupCamera(width: width, height: height)
        lights = [Scene.setupLight(),
                  .spotLight(position: [2, 1, 4], direction: [-1.5, -0.5, -1.5], coneAngle: 25 / 180 * .pi, color: [4, 4, 4])]
    }
    public func updateUniforms(width: Int, height: Int) throws { camera = try Scene.setupCamera(width: width, height: height) }
    public static func setupCamera(width: Int, height: Int) throws -> Camera {
        var c = MRTCamera(); try check(mrt_default_camera(Int32(width), Int32(height), &c)); return c
    }
    public static func setupLight() -> Light {
        .areaLight(position: [0, 1.98, 0], forward: [0, -1, 0], right: [0.25, 0, 0], up: [0, 0, 0.25], color: [4, 4, 4])
    }
}

public final class DragonScene: Scene {
    public override init(width: Int, height: Int) throws {
        try super.init(width: width, height: height)
        models = [
            try Model(name: "train", position: [-0.3, 0, 0.4], scale: 0.5),
            try Model(name: "dragon", position: [0.3, 0.38, 2.5], rotation: [0, .pi / 2 * 1.2, 0], scale: 1.2),
            try Model(name: "treefir", position: [0.5, 0, -0.2], scale: 0.7),
            try Model(name: "plane", position: [0, 0, 0], scale: 10),
            try Model(name: "sphere", position: [-1.9, 0.0, 0.3], scale: 1),
            try Model(name: "sphere", position: [2.9, 0.0, -0.5], scale: 2),
            try Model(name: "plane-back", position: [0, 0, -1.5], scale: 10),
        ]
    }
}
