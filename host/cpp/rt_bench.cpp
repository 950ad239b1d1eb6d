// rt_bench — C++ console host over libraytrace_hip.so (the role the north_star gives to a
// .NET console host; see host/dotnet/ for the C# source).  Builds a BASELINE scene with the
// C++ RayComputeManager mirror, renders N frames, prints one JSON line, optionally dumps the
// uploaded buffers + the accumulation image so tests can replay them through the CPU oracle.
//
//   rt_bench --config 2|3 [--width W --height H] [--frames N] [--warmup K] [--seed S] [--dump prefix] [--scene-only]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rt_host.hpp"

using namespace rthost;

template <typename T>
static void dump(const std::string& path, const std::vector<T>& v)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

int main(int argc, char** argv)
{
    int config = 2, width = 0, height = 0, frames = 10, warmup = 1, seed = 1;
    std::string dumpPrefix;
    bool sceneOnly = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--config") config = atoi(val());
        else if (a == "--width") width = atoi(val());
        else if (a == "--height") height = atoi(val());
        else if (a == "--frames") frames = atoi(val());
        else if (a == "--warmup") warmup = atoi(val());
        else if (a == "--seed") seed = atoi(val());
        else if (a == "--dump") dumpPrefix = val();
        else if (a == "--scene-only") sceneOnly = true;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (!width) width = 1920;
    if (!height) height = 1080;
    try {
        RayComputeManager mgr(width, height, 0, /*createContext=*/!sceneOnly);
        if (config == 2) BuildConfig2(mgr);
        else if (config == 3) BuildConfig3(mgr);
        else { fprintf(stderr, "config must be 2 or 3\n"); return 2; }

        if (sceneOnly) { // host logic only (no GPU): build the buffers the dispatcher would upload
            mgr.renderSeed = seed;
            mgr.numAccumulatedFrames = 1;
            mgr.CreateAllMeshData();
        } else {
            mgr.OnEnable(seed);
            if (warmup > 0) {
                mgr.RenderFrames(warmup);
                mgr.ResetAccumulatedRender(); // measured / dumped frames start again at Frame 1
            }
            rt_reset_counters(mgr.context());
            rt_timer_begin(mgr.context());
            mgr.RenderFrames(frames);
            rt_timer_end(mgr.context());
            RtCounters c = mgr.Counters();
            printf("{\"host\": \"c++\", \"config\": %d, \"width\": %d, \"height\": %d, \"frames\": %d, \"segments\": %llu, \"gpu_ms\": %.4f, "
                   "\"Mrays_per_s\": %.1f, \"models\": %zu, \"spheres\": %zu, \"triangles\": %zu}\n",
                   config, width, height, frames, (unsigned long long)c.segments, c.gpuMs, c.gpuMs > 0 ? c.segments / c.gpuMs / 1e3 : 0.0,
                   mgr.models.size(), mgr.spheres.size(), mgr.triangles.size());
        }
        if (!dumpPrefix.empty()) {
            dump(dumpPrefix + ".models.bin", mgr.meshInfo);
            dump(dumpPrefix + ".triangles.bin", mgr.triangles);
            dump(dumpPrefix + ".nodes.bin", mgr.nodes);
            dump(dumpPrefix + ".spheres.bin", mgr.sphereBuffer);
            RtParams p = mgr.ShaderParams();
            p.frame = 1;
            dump(dumpPrefix + ".params.bin", std::vector<RtParams>(1, p));
            if (!sceneOnly) dump(dumpPrefix + ".accumulated.bin", mgr.ReadAccumulated());
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "rt_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
