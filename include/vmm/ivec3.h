/* vmm/ivec3.h -- the integer-vector subset of the reference's vector-math
 * library that its octree API exposes (reference: include/vmm/ivec3.h; the
 * library itself ships only as a Windows binary, lib/libvmm.a).
 * Only the operations the kept host API needs are provided; semantics are the
 * SHIPPED library's, including its ivec3_equal_vec behaviour (see vmm5.c). */
#ifndef VRT_VMM_IVEC3_H
#define VRT_VMM_IVEC3_H
#include <stdbool.h>
#include <stdint.h>
#include "vec3.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef union _ivec3 {
    struct { int32_t x, y, z; };
    struct { int32_t r, g, b; };
} IVector3;

IVector3 ivec3_add(IVector3 a, IVector3 b);
IVector3 ivec3_sub(IVector3 a, IVector3 b);
IVector3 ivec3_scalar_div(IVector3 in, int scalar); /* truncating; /0 -> (0,0,0) */
IVector3 ivec3_vec3(Vector3 vec);                   /* truncates toward zero */
/* NOT component equality: a.x==b.x && a.y!=0 && b.y!=0 && a.z==b.z, as shipped */
bool ivec3_equal_vec(IVector3 a, IVector3 b);

#ifdef __cplusplus
}
#endif
#endif
