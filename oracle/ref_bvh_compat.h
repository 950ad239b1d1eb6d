/*
 * ref_bvh_compat.h — the few C# / UnityEngine names that the reference's BVH builder (Assets/Scripts/Types/BVH.cs:26-318 and
 * its nested types :437-599) uses, as C++.  TEST INFRASTRUCTURE ONLY: oracle/make_ref.py compiles the reference's own C# text
 * (after the syntactic rewrites listed there) between this header and oracle/ref_bvh_driver.h into oracle/_ref/libref_bvh.so;
 * nothing here is linked into the product.
 *
 * Semantics that matter for the bits:
 *   - C# arrays are references to zero-initialised storage with a Length; Array.Resize allocates a NEW array and rebinds the
 *     variable, while a `ref` local taken before still names the OLD element (BVH.cs:94 + :161-167 write the parent back for
 *     exactly that reason).  CsArray keeps superseded storage alive until cs_collect() so that the same program is defined here;
 *   - float arithmetic is IEEE single per operation (this file is compiled without contraction or fast-math);
 *   - UnityEngine.Mathf: Max/Min are `a > b ? a : b` / `a < b ? a : b` (params overload: a running maximum from values[0]),
 *     CeilToInt(f) = (int)Math.Ceiling(f) with the x64 conversion of NaN / out-of-range values to int.MinValue,
 *     Math.Clamp(v, lo, hi) = v < lo ? lo : v > hi ? hi : v.
 */
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

#define CS_FLOAT_MAX 3.40282347e+38f
#define CS_FLOAT_MIN (-3.40282347e+38f) /* float.MinValue is the most negative float */
#define CS_FLOAT_POSITIVE_INFINITY (__builtin_inff())
#define CS_INT_MAX 2147483647

struct Vector3 {
    float x, y, z;
};

inline std::vector<void*>& cs_graveyard()
{
    static thread_local std::vector<void*> g;
    return g;
}
inline void cs_collect() /* the garbage collector: call when no CsArray of the build is in use any more */
{
    for (void* p : cs_graveyard()) free(p);
    cs_graveyard().clear();
}

template <typename T>
struct CsSpan {
    T* p;
    int n;
    struct Copy { T* p; int Length; T& operator[](int i) const { return p[i]; } };
    Copy ToArray() const
    {
        T* q = static_cast<T*>(calloc(n > 0 ? n : 1, sizeof(T)));
        if (n > 0) memcpy(static_cast<void*>(q), static_cast<const void*>(p), sizeof(T) * (size_t)n);
        cs_graveyard().push_back(q);
        return Copy{q, n};
    }
};

template <typename T>
struct CsArray { /* reference semantics: copies share the storage */
    T* p = nullptr;
    int Length = 0;
    CsArray() = default;
    explicit CsArray(int n) : p(static_cast<T*>(calloc(n > 0 ? n : 1, sizeof(T)))), Length(n) { cs_graveyard().push_back(p); }
    CsArray(T* q, int n) : p(q), Length(n) {}          /* a view of caller memory (the driver's inputs) */
    CsArray(const typename CsSpan<T>::Copy& c) : p(c.p), Length(c.Length) {}
    T& operator[](int i) const { return p[i]; }
    CsSpan<T> AsSpan(int start, int length) const { return CsSpan<T>{p + start, length}; }
};

struct Array {
    template <typename T>
    static void Resize(CsArray<T>& a, int n)
    {
        CsArray<T> b(n);
        memcpy(static_cast<void*>(b.p), static_cast<const void*>(a.p), sizeof(T) * (size_t)(a.Length < n ? a.Length : n));
        a = b; /* the old storage stays in the graveyard: `ref` locals may still point into it */
    }
};

struct Mathf {
    static float Max(float a, float b) { return a > b ? a : b; }
    static float Min(float a, float b) { return a < b ? a : b; }
    static int Max(int a, int b) { return a > b ? a : b; }
    static int Min(int a, int b) { return a < b ? a : b; }
    static float Max(float a, float b, float c) /* Max(params float[] values) */
    {
        float m = a;
        if (b > m) m = b;
        if (c > m) m = c;
        return m;
    }
    static int CeilToInt(float f)
    {
        const double c = std::ceil((double)f);
        if (!(c >= -2147483648.0 && c <= 2147483647.0)) return INT32_MIN; /* NaN and overflow: cvttsd2si's indefinite integer */
        return (int)c;
    }
};
struct Math {
    static int Clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
};

struct Stopwatch {
    std::chrono::steady_clock::time_point t0;
    long long ElapsedMilliseconds = 0;
    static Stopwatch StartNew()
    {
        Stopwatch s;
        s.t0 = std::chrono::steady_clock::now();
        return s;
    }
    void Stop() { ElapsedMilliseconds = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
