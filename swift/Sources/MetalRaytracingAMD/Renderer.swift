// Renderer mirror: init = device + scene upload + acceleration-structure build + targets; draw() = one frame.
import CMRT

public final class Renderer {
    public let maxFramesInFlight = 3
    public private(set) var width: Int, height: Int
    private var ctx: MRTContext?, scene: MRTScene?, renderer: MRTRenderer?

    public init(width: Int, height: Int, scene s: Scene, device: Int32 = 0, seed: UInt32 = 1, maxBounces: Int32 = 3) throws {
        self.width = width; self.height = height
        try check(mrt_context_create(device, &ctx))
        try check(mrt_scene_create(ctx, &scene))
        for mesh in s.models.flatMap(\.meshes) {
            var id: Int32 = -1
            try check(mrt_scene_add_mesh(scene, mesh.positions, 12, mesh.normals, 12, mesh.positions.count / 3, mesh.transform, &id))
            for sub in mesh.submeshes {
                var mat = sub.material
                try check(mrt_mesh_add_submesh(scene, id, sub.indices, sub.indices.count / 3, &mat, nil))
            }
        }
        try check(mrt_scene_set_lights(scene, s.lights, Int32(s.lights.count)))
        try check(mrt_scene_commit(scene))
        try check(mrt_renderer_create(ctx, scene, Int32(width), Int32(height), seed, maxBounces, &renderer))
        var cam = s.camera
        try check(mrt_renderer_set_camera(renderer, &cam))
    }
    deinit { mrt_renderer_destroy(renderer); mrt_scene_destroy(scene); mrt_context_destroy(ctx) }

    public var frameIndex: UInt32 { var f: UInt32 = 0; mrt_renderer_frame_index(renderer, &f); return f }
    public func draw(frames: Int32 = 1) throws { try check(mrt_renderer_render(renderer, frames)) }
    public func wait() throws { try check(mrt_renderer_wait(renderer)) }
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
