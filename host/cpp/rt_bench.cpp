// rt_bench — C++ console host over libraytrace_hip.so (the role the north_star gives to a
// .NET console host; see host/dotnet/ for the C# source).  Builds a BASELINE scene with the
// C++ RayComputeManager mirror, renders N frames, prints one JSON line, optionally dumps the
// uploaded buffers + the accumulation image so tests can replay them through the CPU oracle.
//
//   rt_bench --config 2|3 [--width W --height H] [--frames N] [--warmup K] [--seed S] [--dump prefix] [--scene-only] [--bvh-gpu]
//            [--devices 0,1,...]   several GPUs from this one process: rt_create_multi (cyclic 8-row strips per
//                                  device, gather at readback); a device id may repeat (virtual shards on one GPU)
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
    std::vector<int> devices;
    bool sceneOnly = false, bvhGpu = false;
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
        else if (a == "--bvh-gpu") bvhGpu = true; // every distinct mesh's BVH in one rt_build_bvh_gpu_batch call
        else if (a == "--devices") {
            std::string list = val();
            for (size_t p = 0; p <= list.size();) {
                size_t q = list.find(',', p);
                if (q == std::string::npos) q = list.size();
                if (q > p) devices.push_back(atoi(list.substr(p, q - p).c_str()));
                p = q + 1;
            }
        }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (!width) width = 1920;
    if (!height) height = 1080;
    try {
        const bool multi = !devices.empty();
        RayComputeManager mgr(width, height, 0, /*createContext=*/!sceneOnly && !multi);
        if (config == 2) BuildConfig2(mgr);
        else if (config == 3) BuildConfig3(mgr);
        else { fprintf(stderr, "config must be 2 or 3\n"); return 2; }
        mgr.bvhOnGpu = bvhGpu;

        std::vector<float> multiImage;
        if (sceneOnly) { // host logic only (no GPU): build the buffers the dispatcher would upload
            mgr.renderSeed = seed;
            mgr.numAccumulatedFrames = 1;
            mgr.CreateAllMeshData();
        } else if (multi) { // the same dispatcher calls, forwarded to one context per device (rt_abi.h, rt_create_multi)
            mgr.renderSeed = seed;
            mgr.numAccumulatedFrames = 1;
            mgr.CreateAllMeshData();
            RtMulti* m = nullptr;
            auto ok = [&](int rc, const char* what) {
                if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(m ? rt_multi_context(m, 0) : nullptr));
            };
            ok(rt_create_multi(devices.data(), (int)devices.size(), &m), "rt_create_multi");
            ok(rt_multi_resize(m, width, height), "rt_multi_resize");
            auto u0 = std::chrono::steady_clock::now();
            ok(rt_multi_upload_scene(m, mgr.meshInfo.data(), (int)mgr.meshInfo.size(), mgr.triangles.data(), (int)mgr.triangles.size(),
                                     mgr.nodes.data(), (int)mgr.nodes.size(), mgr.sphereBuffer.data(), (int)mgr.sphereBuffer.size()), "rt_multi_upload_scene");
            double uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u0).count();
            RtParams p = mgr.ShaderParams();
            p.frame = 1;
            ok(rt_multi_set_params(m, &p), "rt_multi_set_params");
            ok(rt_multi_reset_accumulation(m), "rt_multi_reset_accumulation");
            auto t0 = std::chrono::steady_clock::now();
            for (int f = 0; f < frames; f++) { // per frame: UpdateModels + Dispatch on every device, like RenderFrame (RCM:84-95)
                ok(rt_multi_update_models(m, mgr.meshInfo.data(), (int)mgr.meshInfo.size()), "rt_multi_update_models");
                ok(rt_multi_render_frame(m), "rt_multi_render_frame");
            }
            ok(rt_multi_synchronize(m), "rt_multi_synchronize");
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            multiImage.resize((size_t)width * height * 4);
            ok(rt_gather_accumulated(m, multiImage.data(), multiImage.size() * sizeof(float)), "rt_gather_accumulated");
            RtCounters c;
            ok(rt_multi_get_counters(m, &c), "rt_multi_get_counters");
            printf("{\"host\": \"c++\", \"config\": %d, \"width\": %d, \"height\": %d, \"frames\": %d, \"segments\": %llu, \"devices\": %d, "
                   "\"wall_ms\": %.4f, \"Mrays_per_s\": %.1f, \"upload_ms\": %.3f, \"gather_ms\": %.3f}\n",
                   config, width, height, frames, (unsigned long long)c.segments, rt_multi_count(m), ms, ms > 0 ? c.segments / ms / 1e3 : 0.0,
                   uploadMs, rt_multi_last_gather_ms(m));
            rt_destroy_multi(m);
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
            if (multi) dump(dumpPrefix + ".accumulated.bin", multiImage);
            else if (!sceneOnly) dump(dumpPrefix + ".accumulated.bin", mgr.ReadAccumulated());
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "rt_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
