// vec3.h -- host-side Vec3 / point3 / color with the reference's surface (vec3.h:11-336 there).
// Arithmetic keeps the reference's operation order (e.g. v / t multiplies by 1/t), because host
// results such as the camera basis feed the GPU path and are compared bit-for-bit.
#ifndef PT_HOST_VEC3_H
#define PT_HOST_VEC3_H
#include <cmath>
#include <ostream>
#include <string>

#include "str_fun.h"

class Vec3 {
public:
    float e[3] = {0.0f, 0.0f, 0.0f};

    Vec3() {}
    Vec3(float a, float b, float c) { e[0] = a; e[1] = b; e[2] = c; }

    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float r() const { return e[0]; }
    float g() const { return e[1]; }
    float b() const { return e[2]; }

    Vec3 operator-() const { return Vec3(-e[0], -e[1], -e[2]); }
    float operator[](int i) const { return e[i]; }
    float &operator[](int i) { return e[i]; }

    Vec3 &operator+=(const Vec3 &o) { for (int i = 0; i < 3; ++i) e[i] += o.e[i]; return *this; }
    Vec3 &operator*=(float t) { for (int i = 0; i < 3; ++i) e[i] *= t; return *this; }
    Vec3 &operator*=(const Vec3 &o) { for (int i = 0; i < 3; ++i) e[i] *= o.e[i]; return *this; }
    Vec3 &operator/=(float t) { return *this *= 1 / t; }

    float length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    float length() const { return sqrtf(length_squared()); }
    bool near_zero() const
    {
        const float s = 1e-8f;
        return (fabsf(e[0]) < s) && (fabsf(e[1]) < s) && (fabsf(e[2]) < s);
    }
    // "x,y,z"; throws std::invalid_argument / std::out_of_range on malformed text
    void load_from_string(std::string text)
    {
        auto parts = explode(text, ',');
        if (parts.size() < 3) throw std::invalid_argument("vector needs three comma-separated numbers: '" + text + "'");
        for (int i = 0; i < 3; ++i) e[i] = std::stof(parts[i]);
    }
};

using point3 = Vec3;
using color = Vec3;
using vec3 = Vec3;

inline std::ostream &operator<<(std::ostream &os, const vec3 &v) { return os << v.e[0] << ' ' << v.e[1] << ' ' << v.e[2]; }
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline vec3 operator*(float t, const vec3 &v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3 &v, float t) { return t * v; }
inline vec3 operator/(vec3 v, float t) { return (1 / t) * v; }
inline float dot(const vec3 &a, const vec3 &b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
inline vec3 cross(const vec3 &a, const vec3 &b)
{
    return vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline vec3 unit_vector(vec3 v) { return v / v.length(); }
inline vec3 reflect(const vec3 &v, const vec3 &n) { return v - 2 * dot(v, n) * n; }
inline vec3 refract(const vec3 &uv, const vec3 &n, float etai_over_etat)
{
    float cos_theta = fminf(dot(-uv, n), 1.0f);
    vec3 perp = etai_over_etat * (uv + cos_theta * n);
    vec3 parallel = -sqrtf((float)fabs(1.0 - (double)perp.length_squared())) * n;
    return perp + parallel;
}
#endif
