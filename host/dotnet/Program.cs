// .NET 8 console host replacing RayComputeManager.cs (Unity MonoBehaviour) for the compute path:
// builds BASELINE config 2 (16 spheres + ground quad), renders N frames, reports Mrays/s.
// UNCOMPILED IN THIS ENVIRONMENT — see INTEGRATION.md §2.
using System;
using System.Runtime.InteropServices;

namespace RayTraceHost
{
    static class Program
    {
        static Float4 Col(float r, float g, float b, float a = 1) => new Float4(r, g, b, a);

        static RtMaterial Material(Float4 diffuse, int flag = 0)
        {
            // RayTracingMaterial.SetDefaultValues (RayTracingMaterial.cs:29-38)
            return new RtMaterial {
                diffuseCol = diffuse, emissionCol = Col(0, 0, 0, 0), specularCol = Col(1, 1, 1), absorption = Col(0, 0, 0, 0),
                absorptionStrength = 0, emissionStrength = 0, smoothness = 0, specularProbability = 1, ior = 1, flag = flag };
        }

        static int Main(string[] args)
        {
            int frames = args.Length > 0 ? int.Parse(args[0]) : 20;
            const int W = 1920, H = 1080;
            RayTraceNative.Check(IntPtr.Zero, RayTraceNative.rt_create(0, out IntPtr ctx));
            RayTraceNative.Check(ctx, RayTraceNative.rt_resize(ctx, W, H));

            // ground quad: Unity Quad rotated to XZ, scale 40 (RayComputeManager.CreateAllMeshData, RCM:206-236)
            float[] verts = { -0.5f, -0.5f, 0, 0.5f, -0.5f, 0, -0.5f, 0.5f, 0, 0.5f, 0.5f, 0 };
            float[] normals = { 0, 0, -1, 0, 0, -1, 0, 0, -1, 0, 0, -1 };
            int[] indices = { 0, 3, 1, 3, 0, 2 };
            var nodes = new RtBVHNode[4];
            var tris = new RtTriangle[2];
            RayTraceNative.Check(ctx, RayTraceNative.rt_build_bvh(verts, normals, 4, indices, 6, 1, nodes, out int nNodes, tris, IntPtr.Zero));
            var ground = new RtModel {
                nodeOffset = 0, triOffset = 0,
                // localToWorld = T(0) * Rx(90deg) * S(40,40,1), column-major; worldToLocal = inverse
                localToWorld = new Float4x4(new Float4(40, 0, 0, 0), new Float4(0, 0, 40, 0), new Float4(0, -1, 0, 0), new Float4(0, 0, 0, 1)),
                worldToLocal = new Float4x4(new Float4(0.025f, 0, 0, 0), new Float4(0, 0, -1, 0), new Float4(0, 0.025f, 0, 0), new Float4(0, 0, 0, 1)),
                material = Material(Col(0.82f, 0.82f, 0.82f), flag: 1) };
            ground.material.emissionCol = Col(0.28f, 0.28f, 0.33f);
            ground.material.specularProbability = 0;

            var spheres = new RtSphere[16];
            var rng = new Random(2);
            for (int i = 0; i < 16; i++)
            {
                float r = 0.3f + 0.6f * (float)rng.NextDouble();
                spheres[i] = new RtSphere {
                    centre = new Float3((i % 4 - 1.5f) * 2.2f, r, (i / 4 - 1.5f) * 2.2f), radius = r,
                    material = Material(Col((float)rng.NextDouble(), (float)rng.NextDouble(), (float)rng.NextDouble())) };
            }
            RayTraceNative.Check(ctx, RayTraceNative.rt_upload_scene(ctx, new[] { ground }, 1, tris, 2, nodes, nNodes, spheres, 16));

            RayTraceNative.rt_camera_view_params(60f, (float)W / H, 1f, out Float3 vp); // RCM:185-188
            var p = new RtParams {
                abi_version = 1, struct_size = (uint)Marshal.SizeOf<RtParams>(),
                maxBounceCount = 8, numRaysPerPixel = 8, frame = 1, renderSeed = 1, useSky = 1, accumulate = 1,
                defocusStrength = 0, divergeStrength = 1.5f, sunFocus = 500, sunIntensity = 10,
                sunColour = new Float3(1, 1, 1), dirToSun = new Float3(0, -1, 0), viewParams = vp,
                camLocalToWorld = new Float4x4(new Float4(1, 0, 0, 0), new Float4(0, 1, 0, 0), new Float4(0, 0, 1, 0), new Float4(0, 2.6f, -8.8f, 1)) };
            RayTraceNative.Check(ctx, RayTraceNative.rt_set_params(ctx, ref p));
            RayTraceNative.Check(ctx, RayTraceNative.rt_reset_accumulation(ctx));   // RCM:69-76

            RayTraceNative.Check(ctx, RayTraceNative.rt_render_frame(ctx));          // warm-up
            RayTraceNative.rt_reset_counters(ctx);
            RayTraceNative.rt_timer_begin(ctx);
            RayTraceNative.Check(ctx, RayTraceNative.rt_render_frames(ctx, frames)); // RCM:84-95, N times
            RayTraceNative.rt_timer_end(ctx);
            RayTraceNative.Check(ctx, RayTraceNative.rt_get_counters(ctx, out RtCounters c));
            Console.WriteLine($"{frames} frames, {c.segments} segments, {c.gpuMs:F2} ms -> {c.segments / c.gpuMs / 1e3:F1} Mrays/s");

            var sum = new float[W * H * 4];
            RayTraceNative.Check(ctx, RayTraceNative.rt_read_accumulated(ctx, sum, (UIntPtr)(sum.Length * 4)));
            RayTraceNative.rt_destroy(ctx);
            return 0;
        }
    }
}
