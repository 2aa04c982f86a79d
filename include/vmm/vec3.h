/* vmm/vec3.h -- float vector type of the reference's host API (reference: include/vmm/vec3.h). */
#ifndef VRT_VMM_VEC3_H
#define VRT_VMM_VEC3_H

#ifdef __cplusplus
extern "C" {
#endif

typedef union _vec3 {
    struct { float x, y, z; };
    struct { float r, g, b; };
} Vector3;

Vector3 vec3_scalar_mul(Vector3 in, float scalar);

#ifdef __cplusplus
}
#endif
#endif
