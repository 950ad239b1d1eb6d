// rt_host.hpp — C++ host side above the C ABI (include/rt_abi.h).
//
// The reference's host is compiled code (C#, Assets/Scripts/Tracer/RayComputeManager.cs, "RCM");
// the image has no .NET, so the host that is actually built and run here is this C++ one.  It
// mirrors the reference's dispatcher — same field names, same method names, same call sequence —
// with the Unity objects replaced by plain data:
//     UnityEngine.Transform / Camera  -> Transform, Camera
//     Model (Types/Model.cs)          -> Model (mesh + RayTracingMaterial + Transform)
//     ComputeShader + ComputeBuffers  -> RtContext* (libraytrace_hip.so)
// host/dotnet/ holds the C# P/Invoke version of the same thing (uncompiled).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rthost {

struct Vec3 { double x = 0, y = 0, z = 0; };

struct Mat4 { // row-major doubles, m[r][c]
    double m[4][4];
    static Mat4 identity();
    Mat4 operator*(const Mat4& o) const;
    Mat4 inverse() const;                 // general 4x4 inverse (Gauss-Jordan, partial pivoting)
    void toUnity(float out[16]) const;    // Unity Matrix4x4 memory order: column-major floats
};

// UnityEngine.Transform: position / rotation (Euler degrees, Unity z-x-y order, left-handed) / scale
struct Transform {
    Vec3 position{0, 0, 0}, euler{0, 0, 0}, scale{1, 1, 1};
    Transform() = default;
    Transform(Vec3 p, Vec3 e, Vec3 s) : position(p), euler(e), scale(s) {}
    Mat4 localToWorldMatrix() const;
    Mat4 worldToLocalMatrix() const { return localToWorldMatrix().inverse(); }
    Vec3 forward() const;
};

struct Camera { // the UnityEngine.Camera fields RCM:183-190 reads
    Transform transform;
    float fieldOfView = 60.0f;
    float aspect = 16.0f / 9.0f;
};

struct Mesh { // Mesh.vertices / .normals / .triangles (RCM:218)
    std::string name;
    std::vector<float> vertices, normals; // 3 floats per vertex
    std::vector<int32_t> triangles;       // 3 indices per triangle
    int triangleCount() const { return (int)triangles.size() / 3; }
};
std::shared_ptr<Mesh> MakeQuad();                              // Unity built-in Quad
std::shared_ptr<Mesh> MakeCube();                              // Unity built-in Cube
std::shared_ptr<Mesh> MakeRoundedCube(int k = 12, double radius = 0.18);

// Types/RayTracingMaterial.cs:4-38
struct RayTracingMaterial {
    int flag = RT_MATERIAL_DEFAULT;
    float diffuseCol[4] = {1, 1, 1, 1}, emissionCol[4] = {0, 0, 0, 0}, specularCol[4] = {1, 1, 1, 1}, absorption[4] = {0, 0, 0, 0};
    float absorptionMultiplier = 0, emissionStrength = 0, smoothness = 0, specularProbability = 1, ior = 1;
    RtMaterial pack() const;
};

struct Model { // Types/Model.cs:5,14
    std::shared_ptr<Mesh> mesh;
    RayTracingMaterial material;
    Transform transform;
    std::string name;
};

struct Sphere { // extension buffer (rt_abi.h RtSphere)
    Vec3 centre;
    float radius = 1;
    RayTracingMaterial material;
};

class RtError : public std::runtime_error {
  public:
    int status;
    RtError(int s, const std::string& m) : std::runtime_error("rt status " + std::to_string(s) + ": " + m), status(s) {}
};

// RCM:7-264 without the MonoBehaviour
class RayComputeManager {
  public:
    // Main settings — RCM:9-19
    bool rayTracingEnabled = true;
    bool accumulate = true;
    int bvhQuality = RT_BVH_QUALITY_HIGH;
    bool bvhOnGpu = false;                      // build every distinct mesh's BVH in one rt_build_bvh_gpu_batch call (same bytes)
    int bvhDevice = 0;
    int maxBounceCount = 4;
    int numRaysPerPixel = 1;
    float defocusStrength = 0;
    float divergeStrength = 0.3f;
    float focusDistance = 1;
    // Sky settings — RCM:21-27
    bool useSky = false;
    float sunFocus = 500, sunIntensity = 10;
    float sunColor[3] = {1, 1, 1};
    const Transform* sunTransform = nullptr;
    // Info — RCM:36-42
    int numAccumulatedFrames = 0;
    int renderSeed = 0;
    int screenWidth = 0, screenHeight = 0;

    Camera camera;
    std::vector<Model> models;
    std::vector<Sphere> spheres;

    // what InitBVH uploaded (kept for inspection / dumps)
    std::vector<RtModel> meshInfo;
    std::vector<RtTriangle> triangles;
    std::vector<RtBVHNode> nodes;
    std::vector<RtSphere> sphereBuffer;

    explicit RayComputeManager(int width, int height, int device = 0, bool createContext = true);
    ~RayComputeManager();                       // RCM:238-247 OnDestroy -> Release
    RayComputeManager(const RayComputeManager&) = delete;

    void OnEnable(int seed);                    // RCM:61-67 (seed passed in: the reference draws a random one)
    void ResetAccumulatedRender();              // RCM:69-76
    void RenderFrame();                         // RCM:84-95
    void RenderFrames(int n);                   // n x RenderFrame without re-uploading unchanged state
    void InitFrame();                           // RCM:115-124
    void InitTexturesAndBuffers();              // RCM:126-141
    void InitBVH();                             // RCM:143-161
    void UpdateModels();                        // RCM:192-204
    RtParams ShaderParams() const;              // RCM:163-190 (SetShaderParams + UpdateCameraParams) as one POD
    void SetShaderParams();
    void CreateAllMeshData();                   // RCM:206-236 -> meshInfo / triangles / nodes

    std::vector<float> ReadAccumulated();
    RtCounters Counters();
    RtContext* context() const { return ctx_; }

  private:
    RtContext* ctx_ = nullptr;
    bool hasBVH_ = false, sized_ = false;
    void check(int status) const;
};

// BASELINE.json configurations built in C++ (same construction as ray_tracing_amd/scenes.py)
void BuildConfig2(RayComputeManager& m); // 16 spheres + checkered ground quad
void BuildConfig3(RayComputeManager& m); // Cornell room + glass / opaque rounded cubes

} // namespace rthost
