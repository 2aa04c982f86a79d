/* vmm/ray.h -- ray type taken by octree_ray_cast (reference: include/vmm/ray.h). */
#ifndef VRT_VMM_RAY_H
#define VRT_VMM_RAY_H
#include "vec3.h"

typedef struct _ray {
    Vector3 origin, direction;
} Ray;

#endif
