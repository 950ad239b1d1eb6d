// P/Invoke binding of include/rt_abi.h for a .NET 8 host.
// UNCOMPILED IN THIS ENVIRONMENT (no dotnet/mono in the image); kept in sync with the header by
// tests/test_dotnet_layout.py, which parses both.  Every struct is BLITTABLE (plain float/int fields only),
// so RtModel[] / RtTriangle[] / ... are pinned and handed to the library without per-element marshalling.
using System;
using System.Runtime.InteropServices;

namespace RayTraceHost
{
    // Blittable value types (no marshalling: arrays of these structs are pinned and passed as they are), the
    // counterparts of the Unity types the reference's own buffers are made of: Color = 4 floats
    // (RayTracingMaterial.cs:15-27), Vector3 = 3 floats (BVH.cs:579-598), Matrix4x4 = 16 floats in COLUMN-major
    // memory order m00,m10,m20,m30, m01,... (RayComputeManager.cs:256-263).
    [StructLayout(LayoutKind.Sequential)]
    public struct Float3 { public float x, y, z; public Float3(float x, float y, float z) { this.x = x; this.y = y; this.z = z; } }
    [StructLayout(LayoutKind.Sequential)]
    public struct Float4 { public float x, y, z, w; public Float4(float x, float y, float z, float w) { this.x = x; this.y = y; this.z = z; this.w = w; } }
    [StructLayout(LayoutKind.Sequential)]
    public struct Float4x4
    {
        public Float4 c0, c1, c2, c3; // columns
        public Float4x4(Float4 c0, Float4 c1, Float4 c2, Float4 c3) { this.c0 = c0; this.c1 = c1; this.c2 = c2; this.c3 = c3; }
    }

    // RayTracingMaterial.cs:15-27 == RtMaterial (88 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtMaterial
    {
        public Float4 diffuseCol;
        public Float4 emissionCol;
        public Float4 specularCol;
        public Float4 absorption;
        public float absorptionStrength;
        public float emissionStrength;
        public float smoothness;
        public float specularProbability;
        public float ior;
        public int flag;
    }

    // RayComputeManager.cs:256-263 == RtModel (224 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtModel
    {
        public int nodeOffset;
        public int triOffset;
        public Float4x4 worldToLocal;
        public Float4x4 localToWorld;
        public RtMaterial material;
    }

    // BVH.cs:579-598 == RtTriangle (72 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtTriangle
    {
        public Float3 posA;
        public Float3 posB;
        public Float3 posC;
        public Float3 normA;
        public Float3 normB;
        public Float3 normC;
    }

    // BVH.cs:432-457 == RtBVHNode (32 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtBVHNode
    {
        public Float3 boundsMin;
        public Float3 boundsMax;
        public int startIndex;
        public int triangleCount;
    }

    // extension buffer (104 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtSphere
    {
        public Float3 centre;
        public float radius;
        public RtMaterial material;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtParams
    {
        public uint abi_version;
        public uint struct_size;
        public int maxBounceCount;
        public int numRaysPerPixel;
        public int frame;
        public int renderSeed;
        public int useSky;
        public int accumulate;
        public float defocusStrength;
        public float divergeStrength;
        public float sunFocus;
        public float sunIntensity;
        public Float3 sunColour;
        public Float3 dirToSun;
        public Float3 viewParams;
        public Float4x4 camLocalToWorld;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtCounters
    {
        public ulong segments;
        public ulong innerSteps;
        public ulong leafSteps;
        public ulong triTests;
        public ulong sphereTests;
        public ulong modelVisits;
        public ulong pixelFrames;
        public double gpuMs;
    }

    public static class RayTraceNative
    {
        const string Lib = "raytrace_hip"; // libraytrace_hip.so

        [DllImport(Lib)] public static extern int rt_create(int device_id, out IntPtr ctx);
        [DllImport(Lib)] public static extern void rt_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr rt_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_resize(IntPtr ctx, int width, int height);
        [DllImport(Lib)] public static extern int rt_set_partition(IntPtr ctx, int strip_rows, int part_index, int part_count);
        [DllImport(Lib)] public static extern int rt_local_rows(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_upload_scene(IntPtr ctx,
            [In] RtModel[] models, int n_models, [In] RtTriangle[] triangles, int n_triangles,
            [In] RtBVHNode[] nodes, int n_nodes, [In] RtSphere[] spheres, int n_spheres);
        [StructLayout(LayoutKind.Sequential)] public struct RtSceneInfo { public int n_pairs, max_height, flat, n_filtered; public float prepare_ms; }
        [DllImport(Lib)] public static extern int rt_validate_scene(
            [In] RtModel[] models, int n_models, [In] RtTriangle[] triangles, int n_triangles,
            [In] RtBVHNode[] nodes, int n_nodes, [In] RtSphere[] spheres, int n_spheres, out RtSceneInfo info);
        [DllImport(Lib)] public static extern int rt_update_models(IntPtr ctx, [In] RtModel[] models, int n_models);
        [DllImport(Lib)] public static extern int rt_update_spheres(IntPtr ctx, [In] RtSphere[] spheres, int n_spheres);
        [DllImport(Lib)] public static extern int rt_set_params(IntPtr ctx, ref RtParams p);
        [DllImport(Lib)] public static extern int rt_reset_accumulation(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_render_frame(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_render_frames(IntPtr ctx, int n);
        [DllImport(Lib)] public static extern int rt_synchronize(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_get_frame(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_read_frame(IntPtr ctx, [Out] float[] rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_read_accumulated(IntPtr ctx, [Out] float[] rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_timer_begin(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_timer_end(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_reset_counters(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_get_counters(IntPtr ctx, out RtCounters c);
        [DllImport(Lib)] public static extern int rt_build_bvh([In] float[] verts, [In] float[] normals, int n_verts,
            [In] int[] indices, int n_indices, int quality, [Out] RtBVHNode[] out_nodes, out int out_n_nodes,
            [Out] RtTriangle[] out_tris, IntPtr out_stats);
        [DllImport(Lib)] public static extern int rt_camera_view_params(float fov_deg, float aspect, float focus_distance, out Float3 view_params);

        // several GPUs from this one process (rt_abi.h: rt_create_multi): cyclic 8-row strips per device, gather at readback
        [DllImport(Lib)] public static extern int rt_create_multi([In] int[] device_ids, int n_devices, out IntPtr multi);
        [DllImport(Lib)] public static extern void rt_destroy_multi(IntPtr multi);
        [DllImport(Lib)] public static extern IntPtr rt_multi_context(IntPtr multi, int i);
        [DllImport(Lib)] public static extern int rt_multi_resize(IntPtr multi, int width, int height);
        [DllImport(Lib)] public static extern int rt_multi_upload_scene(IntPtr multi,
            [In] RtModel[] models, int n_models, [In] RtTriangle[] triangles, int n_triangles,
            [In] RtBVHNode[] nodes, int n_nodes, [In] RtSphere[] spheres, int n_spheres);
        [DllImport(Lib)] public static extern int rt_multi_update_models(IntPtr multi, [In] RtModel[] models, int n_models);
        [DllImport(Lib)] public static extern int rt_multi_set_params(IntPtr multi, ref RtParams p);
        [DllImport(Lib)] public static extern int rt_multi_reset_accumulation(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_multi_render_frame(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_multi_render_frames(IntPtr multi, int n);
        [DllImport(Lib)] public static extern int rt_multi_synchronize(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_gather_accumulated(IntPtr multi, [Out] float[] rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_multi_get_counters(IntPtr multi, out RtCounters c);
        [DllImport(Lib)] public static extern double rt_multi_last_gather_ms(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_multi_peer_access(IntPtr multi, out int pairs, out int enabled);
        [DllImport(Lib)] public static extern int rt_gather_accumulated_to_device(IntPtr multi, int root, IntPtr d_rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_gather_frame_to_device(IntPtr multi, int root, IntPtr d_rgba, UIntPtr bytes);
        // one process per GPU: the per-tile buffers gathered over a caller-owned ncclComm_t (RCCL over xGMI), de-interleaved on root
        [DllImport(Lib)] public static extern int rt_gather_rccl(IntPtr ctx, IntPtr nccl_comm, int root, int use_accumulated, IntPtr d_rgba, UIntPtr bytes);
        // the rest of include/rt_abi.h (display blit, checkpoint, caller stream / caller-owned targets, statistics, the other builders)
        [DllImport(Lib)] public static extern IntPtr rt_version();
        [DllImport(Lib)] public static extern int rt_set_stream(IntPtr ctx, IntPtr hip_stream);
        [DllImport(Lib)] public static extern int rt_local_to_global_row(IntPtr ctx, int local_row);
        [DllImport(Lib)] public static extern int rt_bind_render_targets(IntPtr ctx, IntPtr d_frame_render, IntPtr d_accumulated);
        [DllImport(Lib)] public static extern int rt_get_render_targets(IntPtr ctx, out IntPtr d_frame_render, out IntPtr d_accumulated);
        [DllImport(Lib)] public static extern int rt_flush(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_display(IntPtr ctx, int frame, int use_accumulated, [Out] float[] rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_display_srgb8(IntPtr ctx, int frame, int use_accumulated, int flip_y, [Out] byte[] rgba8, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_write_accumulated(IntPtr ctx, [In] float[] rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_enable_stats(IntPtr ctx, int enabled);
        [DllImport(Lib)] public static extern int rt_multi_count(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_multi_update_spheres(IntPtr multi, [In] RtSphere[] spheres, int n_spheres);
        [DllImport(Lib)] public static extern int rt_gather_frame(IntPtr multi, [Out] float[] rgba, UIntPtr bytes);
        [DllImport(Lib)] public static extern int rt_build_bvh_mt([In] float[] verts, [In] float[] normals, int n_verts,
            [In] int[] indices, int n_indices, int quality, int n_threads, [Out] RtBVHNode[] out_nodes, out int out_n_nodes,
            [Out] RtTriangle[] out_tris, IntPtr out_stats);
        [DllImport(Lib)] public static extern int rt_build_bvh_gpu(int device_id, [In] float[] verts, [In] float[] normals, int n_verts,
            [In] int[] indices, int n_indices, int quality, [Out] RtBVHNode[] out_nodes, out int out_n_nodes,
            [Out] RtTriangle[] out_tris, IntPtr out_stats);
        [DllImport(Lib)] public static extern void rt_build_bvh_gpu_release();
        [DllImport(Lib)] public static extern int rt_build_bvh_gpu_batch(int device_id, int n_meshes, IntPtr[] verts, IntPtr[] normals, int[] n_verts, IntPtr[] indices, int[] n_indices, int quality, IntPtr out_nodes, int[] out_n_nodes, int[] out_node_offset, IntPtr out_tris, int[] out_tri_offset, IntPtr out_stats);

        public static void Check(IntPtr ctx, int status)
        {
            if (status != 0)
                throw new InvalidOperationException($"rt status {status}: {Marshal.PtrToStringAnsi(rt_last_error(ctx))}");
        }
    }
}
