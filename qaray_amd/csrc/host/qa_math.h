// qa_math.h — host-side vector/matrix types of the scene graph.
//
// The host layer rebuilds the reference's transforms (src/core/transform.h:36-79) and camera
// frame (src/renderers/renderer.cpp:71-113).  The reference does that arithmetic through GLM
// 0.9.8.4; parity of the flattened scene is checked bit-for-bit against the reference's own
// matrices, so the operators below are written in the evaluation order GLM's scalar code uses
// (mat*mat: type_mat3x3.inl:446-479, inverse: func_matrix.inl:272-294, rotate/scale:
// gtc/matrix_transform.inl:19-47,79-87) and the library is compiled with -ffp-contract=off.
#pragma once
#include <cmath>

namespace qaray_hip {

struct Vec3 {
  float x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  float &operator[](int i) { return (&x)[i]; }
  const float &operator[](int i) const { return (&x)[i]; }
};
using Point3 = Vec3;
using Color3f = Vec3;

inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(const Vec3 &a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(const Vec3 &a, const Vec3 &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(const Vec3 &a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, const Vec3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(const Vec3 &a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 &operator+=(Vec3 &a, const Vec3 &b) { a = a + b; return a; }
inline Vec3 &operator-=(Vec3 &a, const Vec3 &b) { a = a - b; return a; }
inline Vec3 &operator*=(Vec3 &a, float s) { a = a * s; return a; }
inline float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b)
{
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline Vec3 normalize(const Vec3 &a) { return a * (1.f / std::sqrt(dot(a, a))); }
inline float length(const Vec3 &a) { return std::sqrt(dot(a, a)); }

// Column-major 3x3: c[col][row], same storage as glm::mat3.
struct Mat3 {
  float c[3][3];
  Mat3() : Mat3(1.f) {}
  explicit Mat3(float d)
  {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i][j] = (i == j) ? d : 0.f;
  }
  const float *data() const { return &c[0][0]; }
};

inline Vec3 operator*(const Mat3 &m, const Vec3 &v)
{
  return {m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z,
          m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
          m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z};
}

inline Mat3 operator*(const Mat3 &a, const Mat3 &b)
{
  Mat3 r;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i)
      r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
  return r;
}

inline Mat3 inverse(const Mat3 &m)
{
  const float ood = 1.f / (+m.c[0][0] * (m.c[1][1] * m.c[2][2] - m.c[2][1] * m.c[1][2])
                           - m.c[1][0] * (m.c[0][1] * m.c[2][2] - m.c[2][1] * m.c[0][2])
                           + m.c[2][0] * (m.c[0][1] * m.c[1][2] - m.c[1][1] * m.c[0][2]));
  Mat3 r;
  r.c[0][0] = +(m.c[1][1] * m.c[2][2] - m.c[2][1] * m.c[1][2]) * ood;
  r.c[1][0] = -(m.c[1][0] * m.c[2][2] - m.c[2][0] * m.c[1][2]) * ood;
  r.c[2][0] = +(m.c[1][0] * m.c[2][1] - m.c[2][0] * m.c[1][1]) * ood;
  r.c[0][1] = -(m.c[0][1] * m.c[2][2] - m.c[2][1] * m.c[0][2]) * ood;
  r.c[1][1] = +(m.c[0][0] * m.c[2][2] - m.c[2][0] * m.c[0][2]) * ood;
  r.c[2][1] = -(m.c[0][0] * m.c[2][1] - m.c[2][0] * m.c[0][1]) * ood;
  r.c[0][2] = +(m.c[0][1] * m.c[1][2] - m.c[1][1] * m.c[0][2]) * ood;
  r.c[1][2] = -(m.c[0][0] * m.c[1][2] - m.c[1][0] * m.c[0][2]) * ood;
  r.c[2][2] = +(m.c[0][0] * m.c[1][1] - m.c[1][0] * m.c[0][1]) * ood;
  return r;
}

// Upper-left 3x3 of glm::rotate(mat4(1), angle, axis): the identity columns are still multiplied
// and summed (1*R + 0*R' + 0*R''), which is exact, so only the Rotate entries matter.
inline Mat3 rotation(float angleRad, const Vec3 &v)
{
  const float c = std::cos(angleRad);
  const float s = std::sin(angleRad);
  const Vec3 axis = normalize(v);
  const Vec3 temp = (1.f - c) * axis;
  float R[3][3];
  R[0][0] = c + temp[0] * axis[0];
  R[0][1] = temp[0] * axis[1] + s * axis[2];
  R[0][2] = temp[0] * axis[2] - s * axis[1];
  R[1][0] = temp[1] * axis[0] - s * axis[2];
  R[1][1] = c + temp[1] * axis[1];
  R[1][2] = temp[1] * axis[2] + s * axis[0];
  R[2][0] = temp[2] * axis[0] + s * axis[1];
  R[2][1] = temp[2] * axis[1] - s * axis[0];
  R[2][2] = c + temp[2] * axis[2];
  const Mat3 I(1.f);
  Mat3 r;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i)
      r.c[j][i] = I.c[0][i] * R[j][0] + I.c[1][i] * R[j][1] + I.c[2][i] * R[j][2];
  return r;
}

inline Mat3 scaling(float sx, float sy, float sz)
{
  Mat3 r(1.f);
  const float s[3] = {sx, sy, sz};
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) r.c[j][i] = r.c[j][i] * s[j];
  return r;
}

}  // namespace qaray_hip
