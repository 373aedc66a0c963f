// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// System.Numerics / System.MathF / System.Random of the .NET Core 3.1 BCL, as
// used by the reference.  The BCL is not part of /root/reference
// (PathTracer/PathTracer.csproj:5 pins netcoreapp3.1); what is restated here is
// its published behaviour:
//   Vector3.Dot      (x*x' + y*y') + z*z'      (scalar and dpps agree on this order)
//   Vector3.Cross    plain scalar products, unfused
//   Vector3.Normalize v / sqrt(Dot(v,v))
//   Vector3.Min/Max  (a < b) ? a : b  /  (a > b) ? a : b
//   MathF.Min/Max    IEEE 754:2019 minimum/maximum (NaN-propagating, -0 < +0)
//   Matrix4x4 *      SSE form ((a1*b1 + a2*b2) + (a3*b3 + a4*b4)) per element
//   Random(seed)     Knuth subtractive generator, MBIG = int.MaxValue, MSEED = 161803398
#include "oracle.hpp"
#include <cmath>
#include <climits>
#include <cstring>

namespace oracle {

float Dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

Vec3 Cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

Vec3 Normalize(Vec3 a) {
    float length = sqrtf(Dot(a, a));
    return {a.x / length, a.y / length, a.z / length};
}

Vec3 Vec3Min(Vec3 a, Vec3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
Vec3 Vec3Max(Vec3 a, Vec3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }

float MathFMax(float x, float y) {
    if ((x > y) || std::isnan(x)) return x;
    if (x == y) return std::signbit(x) ? y : x;
    return y;
}

float MathFMin(float x, float y) {
    if ((x < y) || std::isnan(x)) return x;
    if (x == y) return std::signbit(x) ? x : y;
    return y;
}

Matrix4x4 MatIdentity() {
    Matrix4x4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return r;
}

// Matrix4x4.CreateRotationX: M22 = c, M23 = s, M32 = -s, M33 = c (row-vector convention)
Matrix4x4 MatRotationX(float a) {
    Matrix4x4 r = MatIdentity();
    float c = cosf(a), s = sinf(a);
    r.m[1][1] = c; r.m[1][2] = s; r.m[2][1] = -s; r.m[2][2] = c;
    return r;
}

// M11 = c, M13 = -s, M31 = s, M33 = c
Matrix4x4 MatRotationY(float a) {
    Matrix4x4 r = MatIdentity();
    float c = cosf(a), s = sinf(a);
    r.m[0][0] = c; r.m[0][2] = -s; r.m[2][0] = s; r.m[2][2] = c;
    return r;
}

// M11 = c, M12 = s, M21 = -s, M22 = c
Matrix4x4 MatRotationZ(float a) {
    Matrix4x4 r = MatIdentity();
    float c = cosf(a), s = sinf(a);
    r.m[0][0] = c; r.m[0][1] = s; r.m[1][0] = -s; r.m[1][1] = c;
    return r;
}

Matrix4x4 MatTranslation(float x, float y, float z) {
    Matrix4x4 r = MatIdentity();
    r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z;
    return r;
}

Matrix4x4 MatScale(float s) {
    Matrix4x4 r = MatIdentity();
    r.m[0][0] = s; r.m[1][1] = s; r.m[2][2] = s;
    return r;
}

Matrix4x4 MatMul(const Matrix4x4& a, const Matrix4x4& b) {
    Matrix4x4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + (a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j]);
    return r;
}

Vec3 Transform(Vec3 p, const Matrix4x4& m) {
    return {p.x * m.m[0][0] + p.y * m.m[1][0] + p.z * m.m[2][0] + m.m[3][0],
            p.x * m.m[0][1] + p.y * m.m[1][1] + p.z * m.m[2][1] + m.m[3][1],
            p.x * m.m[0][2] + p.y * m.m[1][2] + p.z * m.m[2][2] + m.m[3][2]};
}

Vec3 TransformNormal(Vec3 n, const Matrix4x4& m) {
    return {n.x * m.m[0][0] + n.y * m.m[1][0] + n.z * m.m[2][0],
            n.x * m.m[0][1] + n.y * m.m[1][1] + n.z * m.m[2][1],
            n.x * m.m[0][2] + n.y * m.m[1][2] + n.z * m.m[2][2]};
}

// ------------------------------------------------------------ System.Random
DotNetRandom::DotNetRandom(int32_t seed) {
    const int32_t MBIG = INT_MAX, MSEED = 161803398;
    int32_t subtraction = (seed == INT_MIN) ? INT_MAX : (seed < 0 ? -seed : seed);
    int32_t mj = MSEED - subtraction;
    memset(sa, 0, sizeof(sa));
    sa[55] = mj;
    int32_t mk = 1;
    int ii = 0;
    for (int i = 1; i < 55; i++) {
        if ((ii += 21) >= 55) ii -= 55;
        sa[ii] = mk;
        mk = mj - mk;
        if (mk < 0) mk += MBIG;
        mj = sa[ii];
    }
    for (int k = 1; k < 5; k++) {
        for (int i = 1; i < 56; i++) {
            int n = i + 30;
            if (n >= 55) n -= 55;
            sa[i] = (int32_t)((uint32_t)sa[i] - (uint32_t)sa[1 + n]);
            if (sa[i] < 0) sa[i] += MBIG;
        }
    }
    inext = 0;
    inextp = 21;
}

int32_t DotNetRandom::InternalSample() {
    const int32_t MBIG = INT_MAX;
    int locINext = inext, locINextp = inextp;
    if (++locINext >= 56) locINext = 1;
    if (++locINextp >= 56) locINextp = 1;
    int32_t retVal = sa[locINext] - sa[locINextp];
    if (retVal == MBIG) retVal--;
    if (retVal < 0) retVal += MBIG;
    sa[locINext] = retVal;
    inext = locINext;
    inextp = locINextp;
    return retVal;
}

int32_t DotNetRandom::Next() { return InternalSample(); }
double DotNetRandom::NextDouble() { return InternalSample() * (1.0 / INT_MAX); }
int32_t DotNetRandom::Next(int32_t maxValue) { return (int32_t)(NextDouble() * maxValue); }

}  // namespace oracle
