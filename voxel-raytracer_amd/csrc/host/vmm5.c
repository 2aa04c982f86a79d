/* vmm5.c -- the six vector-math routines the host API needs from the
 * reference's vmm library (lib/libvmm.a: binary only, no source). Semantics are
 * those of the shipped binary (from its disassembly); the
 * flattened-octree bytes depend on them, ivec3_equal_vec in particular. */
#include <vmm/ivec3.h>
#include <vmm/vec3.h>

IVector3 ivec3_add(IVector3 a, IVector3 b) {
    IVector3 r;
    r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z;
    return r;
}

IVector3 ivec3_sub(IVector3 a, IVector3 b) {
    IVector3 r;
    r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z;
    return r;
}

IVector3 ivec3_scalar_div(IVector3 in, int scalar) {
    IVector3 r;
    r.x = r.y = r.z = 0;
    if (scalar == 0) return r;
    r.x = in.x / scalar; r.y = in.y / scalar; r.z = in.z / scalar;
    return r;
}

IVector3 ivec3_vec3(Vector3 vec) {
    IVector3 r;
    r.x = (int32_t)vec.x; r.y = (int32_t)vec.y; r.z = (int32_t)vec.z;
    return r;
}

/* The shipped routine compares x and z but only tests the two y values for
 * being non-zero. Split/merge decisions in the octree builder observe this. */
bool ivec3_equal_vec(IVector3 a, IVector3 b) {
    return a.x == b.x && a.y != 0 && b.y != 0 && a.z == b.z;
}

Vector3 vec3_scalar_mul(Vector3 in, float scalar) {
    Vector3 r;
    r.x = in.x * scalar; r.y = in.y * scalar; r.z = in.z * scalar;
    return r;
}
