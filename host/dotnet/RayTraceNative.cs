// P/Invoke binding of include/rt_abi.h for a .NET 8 host.
// UNCOMPILED IN THIS ENVIRONMENT (no dotnet/mono in the image); kept in sync with the header by
// tests/test_dotnet_layout.py, which parses both.
using System;
using System.Runtime.InteropServices;

namespace RayTraceHost
{
    // RayTracingMaterial.cs:15-27 == RtMaterial (88 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtMaterial
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 4)] public float[] diffuseCol;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 4)] public float[] emissionCol;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 4)] public float[] specularCol;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 4)] public float[] absorption;
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
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 16)] public float[] worldToLocal;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 16)] public float[] localToWorld;
        public RtMaterial material;
    }

    // BVH.cs:579-598 == RtTriangle (72 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtTriangle
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] posA;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] posB;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] posC;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] normA;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] normB;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] normC;
    }

    // BVH.cs:432-457 == RtBVHNode (32 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtBVHNode
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] boundsMin;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] boundsMax;
        public int startIndex;
        public int triangleCount;
    }

    // extension buffer (104 bytes)
    [StructLayout(LayoutKind.Sequential)]
    public struct RtSphere
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] centre;
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
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] sunColour;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] dirToSun;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] viewParams;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 16)] public float[] camLocalToWorld;
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
        [DllImport(Lib)] public static extern int rt_camera_view_params(float fov_deg, float aspect, float focus_distance, [Out] float[] out3);

        public static void Check(IntPtr ctx, int status)
        {
            if (status != 0)
                throw new InvalidOperationException($"rt status {status}: {Marshal.PtrToStringAnsi(rt_last_error(ctx))}");
        }
    }
}
