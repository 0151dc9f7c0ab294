// Renderer mirror: init = device + scene upload + acceleration-structure build + targets; draw() = one frame.
import CMRT

/// createBuffers / geometry descriptors (Renderer.swift:107-182): hand a Scene's models and lights to an MRTScene (not committed).
func upload(_ s: Scene, to scene: MRTScene?, instancing: Bool) throws {
    if instancing { try check(mrt_scene_set_option(scene, "instancing", 1)) }
    var loaded: [String: Int32] = [:]                 // resource name -> mesh id of its first use
    for model in s.models {
        for mesh in model.meshes {
            var id: Int32 = -1
            if instancing, model.meshes.count == 1, let source = loaded[model.name] {
                try check(mrt_scene_add_instance(scene, source, mesh.transform, &id)); continue
            }
            try check(mrt_scene_add_mesh(scene, mesh.positions, 12, mesh.normals, 12, mesh.positions.count / 3, mesh.transform, &id))
            for sub in mesh.submeshes {
                var mat = sub.material
                try check(mrt_mesh_add_submesh(scene, id, sub.indices, sub.indices.count / 3, &mat, nil))
            }
            if loaded[model.name] == nil { loaded[model.name] = id }
        }
    }
    try check(mrt_scene_set_lights(scene, s.lights, Int32(s.lights.count)))
}

public final class Renderer {
    public let maxFramesInFlight = 3
    public private(set) var width: Int, height: Int
    private var ctx: MRTContext?, scene: MRTScene?, renderer: MRTRenderer?

    /// instancing = true: models loaded from the same resource become instances of one mesh (mrt_scene_add_instance) and the scene is committed
    /// two-level — one BLAS per distinct mesh + a TLAS, the reference's instance acceleration structure (Renderer.swift:193-213).
    public init(width: Int, height: Int, scene s: Scene, device: Int32 = 0, seed: UInt32 = 1, maxBounces: Int32 = 3, instancing: Bool = false) throws {
        self.width = width; self.height = height
        try check(mrt_context_create(device, &ctx))
        try check(mrt_scene_create(ctx, &scene))
        try upload(s, to: scene, instancing: instancing)
        try check(mrt_scene_commit(scene))
        try check(mrt_renderer_create(ctx, scene, Int32(width), Int32(height), seed, maxBounces, &renderer))
        var cam = s.camera
        try check(mrt_renderer_set_camera(renderer, &cam))
    }
    deinit { mrt_renderer_destroy(renderer); mrt_scene_destroy(scene); mrt_context_destroy(ctx) }

    public var frameIndex: UInt32 { var f: UInt32 = 0; mrt_renderer_frame_index(renderer, &f); return f }
    public func draw(frames: Int32 = 1) throws { try check(mrt_renderer_render(renderer, frames)) }
    public func wait() throws { try check(mrt_renderer_wait(renderer)) }
    /// the completion handler of Renderer.swift:285-287 as a poll: frames whose accumulation has finished on the device; never blocks
    public var framesCompleted: UInt64 { var f: UInt64 = 0; mrt_renderer_frames_completed(renderer, &f); return f }
    /// updateUniforms (Renderer.swift:216-229): size, frameIndex, lightCount and camera as the one block the reference binds at buffer index 0
    public var uniforms: MRTUniforms {
        get { var u = MRTUniforms(); mrt_renderer_get_uniforms(renderer, &u); return u }
        set { var u = newValue; if mrt_renderer_set_uniforms(renderer, &u) == 0 { width = Int(u.width); height = Int(u.height) } }
    }
    /// animated transforms: a new object->world matrix (column-major 4x4) for one mesh / instance, then commit(); a two-level scene rebuilds only its TLAS
    public func setInstanceTransform(meshId: Int32, transform: [Float]) throws { try check(mrt_scene_set_instance_transform(scene, meshId, transform)) }
    /// Deforming geometry: new object-space positions / normals (packed xyz) for one mesh's vertices, same count; `commit()` then refits the tree of a flattened scene.
    public func updateMesh(meshId: Int32, positions: [Float], normals: [Float]) throws {
        // the C side reads positions.count / 3 normals: a shorter array would be read past its end
        guard positions.count % 3 == 0, normals.count == positions.count else { throw MRTError(code: 1, message: "updateMesh: \(normals.count / 3) normals for \(positions.count / 3) positions (one normal per vertex)") }
        try check(mrt_scene_update_mesh(scene, meshId, positions, 12, normals, 12, positions.count / 3))
    }
    public func commit() throws { try check(mrt_scene_commit(scene)) }
    /// implementation knobs of include/mrt_abi.h: "frames_in_flight", "frame_batch", "materials", "megakernel", ...
    public func setOption(_ key: String, _ value: Double) throws { try check(mrt_renderer_set_option(renderer, key, value)) }
    public func drawableSizeWillChange(width: Int, height: Int) throws {
        self.width = width; self.height = height
        try check(mrt_renderer_resize(renderer, Int32(width), Int32(height)))
    }
    public func accumulation() throws -> [Float] {
        var a = [Float](repeating: 0, count: width * height * 4)
        try check(mrt_renderer_read_accum(renderer, &a, a.count * 4)); return a
    }
    public func tonemapped() throws -> [UInt8] {
        var a = [UInt8](repeating: 0, count: width * height * 4)
        try check(mrt_renderer_read_tonemapped_rgba8(renderer, &a, a.count)); return a
    }
}

/// The same Renderer over the n GPUs of one node in one process (mrt_group_*): the scene is replicated, the image sharded by 8x8 screen tile
/// (tile_id % n == rank), gather() runs the ONE reduce(sum) per output image (RCCL over xGMI).  The reference creates a single MTLDevice
/// (Renderer.swift:46-59); this widens that seam.
public final class GroupRenderer {
    public let width: Int, height: Int
    private var group: MRTGroup?, template: MRTScene?, gr: MRTGroupRenderer?

    public init(width: Int, height: Int, scene s: Scene, devices: [Int32], seed: UInt32 = 1, maxBounces: Int32 = 3, instancing: Bool = false) throws {
        self.width = width; self.height = height
        try check(mrt_group_create(devices, Int32(devices.count), &group))
        var c0: MRTContext?
        try check(mrt_group_context(group, 0, &c0))
        try check(mrt_scene_create(c0, &template))
        try upload(s, to: template, instancing: instancing)                 // replicated and committed on every device by the next call
        try check(mrt_group_renderer_create(group, template, Int32(width), Int32(height), seed, maxBounces, &gr))
        var cam = s.camera
        try check(mrt_group_set_camera(gr, &cam))
    }
    deinit { mrt_group_renderer_destroy(gr); mrt_scene_destroy(template); mrt_group_destroy(group) }

    public func draw(frames: Int32 = 1) throws { try check(mrt_group_render(gr, frames)) }      // every device; returns at once
    public func wait() throws { try check(mrt_group_wait(gr)) }
    public var framesCompleted: UInt64 { var f: UInt64 = 0; mrt_group_frames_completed(gr, &f); return f }
    public func setOption(_ key: String, _ value: Double) throws { try check(mrt_group_set_option(gr, key, value)) }
    /// the assembled image (w * h RGBA32F, row 0 = bottom): one ncclReduce(sum) into device 0, then the copy to the host
    public func gather() throws -> [Float] {
        var a = [Float](repeating: 0, count: width * height * 4)
        try check(mrt_group_gather(gr, &a, a.count * 4)); return a
    }
}
