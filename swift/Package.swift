// swift-tools-version:5.7
import PackageDescription

let package = Package(
    name: "MetalRaytracingAMD",
    products: [.library(name: "MetalRaytracingAMD", targets: ["MetalRaytracingAMD"])],
    targets: [
        .systemLibrary(name: "CMRT", path: "Sources/CMRT"),
        .target(name: "MetalRaytracingAMD", dependencies: ["CMRT"]),
    ]
)
