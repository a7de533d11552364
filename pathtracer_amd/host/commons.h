// commons.h -- shared constants and the umbrella include (commons.h in the reference).
// Errors: the reference prints and exit(99)s on any CUDA error (commons.h:68-77); here failures of
// the GPU library surface as pt_host::scene_error exceptions carrying pt_last_error().
#ifndef PT_HOST_COMMONS_H
#define PT_HOST_COMMONS_H
#include <cfloat>
#include <iostream>
#include <string>
#include <vector>

const float pi = 3.1415926535897932385f;
#ifndef PI
#define PI 3.1415926535897932385f // cuda_functions.h:6
#endif
// the reference redefines INFINITY as FLT_MAX (commons.h:11); the value the hit queries use
constexpr float PT_T_MAX = FLT_MAX;

inline float degrees_to_radians(float degrees) { return degrees * pi / 180.0f; }
inline float clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

#include "str_fun.h"
#include "vec3.h"
#include "ray.h"
#include "pt_host.h"
#include "hittable.h"
#endif
