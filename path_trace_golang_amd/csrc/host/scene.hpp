// scene.hpp -- C++ mirror of the reference's internal/scene package
// (/root/reference/internal/scene/scene.go:9-158, io.go:10-38).
#pragma once

#include <memory>
#include <string>
#include <vector>

namespace pthost {
namespace scene {

struct Vec3 { double X = 0, Y = 0, Z = 0; };   // scene.go:9-13
struct Color { double R = 0, G = 0, B = 0; };  // scene.go:16-20

struct Camera {  // scene.go:24-32
    Vec3 Position, Target, Up;
    double FOV = 0, Aperture = 0, FocusDist = 0, AspectRatio = 0;
};

// scene.go:33-40
constexpr const char *MaterialLambert = "lambert";
constexpr const char *MaterialMetal = "metal";
constexpr const char *MaterialDielectric = "dielectric";
constexpr const char *MaterialEmissive = "emissive";
constexpr const char *MaterialMirror = "mirror";

struct Material {  // scene.go:41-63
    std::string ID, Type;
    Color Albedo;
    double Rough = 0, IOR = 0;
    Color Emit;
    double Power = 0;
    Color Absorption;
    double Smoothness = 0, Reflectivity = 0;
    Color Tint;
    double AbsorptionScale = 0;
};

// scene.go:66-73
constexpr const char *ObjectSphere = "sphere";
constexpr const char *ObjectPlane = "plane";
constexpr const char *ObjectBox = "box";
constexpr const char *ObjectSphereLight = "sphere_light";

struct Object {  // scene.go:76-84
    std::string ID, Type;
    Vec3 Position, Size;
    std::string MaterialID;
};

struct RenderSettings {  // scene.go:87-92
    int Width = 0, Height = 0, SamplesPerPx = 0, MaxDepth = 0;
};

struct Fog {  // scene.go:96-131 (carried for Save round trips; the CPU engine ignores it)
    double Density = 0;
    Color Col;
    double Scatter = 0, SigmaS = 0, SigmaA = 0, G = 0, HeteroStrength = 0, NoiseScale = 0;
    int NoiseOctaves = 0;
    bool AffectSky = false, GPUVolumetric = false;
};

struct Sky {  // scene.go:135-140
    std::string Type;
    Color Col, Horizon, Zenith;
};

struct Scene {  // scene.go:143-158
    std::string Name;
    Camera Cam;
    std::vector<Object> Objects;
    std::vector<Material> Materials;
    // Go distinguishes a nil slice (no "objects" key, or null: Save writes null) from an empty one (Save writes [])
    bool ObjectsNil = true, MaterialsNil = true;
    RenderSettings Settings;
    Color Background;
    std::unique_ptr<Sky> SkyPtr;  // nil when absent or null
    std::unique_ptr<Fog> FogPtr;
};

// Load reads a Scene from a JSON file (io.go:10-22). Throws std::runtime_error with the
// reference's message prefixes ("open scene: ...", "decode scene: ...").
std::unique_ptr<Scene> Load(const std::string &path);
std::unique_ptr<Scene> Decode(const std::string &json_text);

// Save writes a Scene as indented JSON (io.go:25-38). Throws "create scene: ..." / "encode scene: ...".
void Save(const std::string &path, const Scene &sc);
std::string Encode(const Scene &sc);

}  // namespace scene
}  // namespace pthost
