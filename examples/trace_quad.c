/* Minimal C client of the ABI (include/raycore_mi355x.h): one quad, two instances, four rays.
 *   gcc -Iinclude examples/trace_quad.c -L raycore.jl_amd -lraycore_mi355x -Wl,-rpath,$PWD/raycore.jl_amd -o trace_quad */
#include <stdio.h>

#include "raycore_mi355x.h"

#define CHECK(x)                                                      \
    do {                                                              \
        if ((x) != RC_OK) {                                           \
            fprintf(stderr, "%s failed: %s\n", #x, rc_last_error()); \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main(void) {
    const float quad[2 * 9] = {0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0, 0, 1, 0};
    const float xforms[2 * 12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0,  /* identity */
                                  1, 0, 0, 5, 0, 1, 0, 0, 0, 0, 1, 0}; /* translate x by 5 (Vulkan row-major 3x4) */
    const uint32_t ids[2] = {11, 22};
    const rc_ray rays[4] = {{0.25f, 0.25f, 1, 0, 0, 0, -1, 1e30f}, {5.25f, 0.25f, 2, 0, 0, 0, -1, 1e30f},
                            {9, 9, 1, 0, 0, 0, -1, 1e30f}, {0.75f, 0.5f, 3, 0, 0, 0, -1, 1e30f}};
    rc_hit hits[4];
    rc_scene* scene = NULL;
    uint32_t blas = 0, handle = 0;
    if (rc_device_count() == 0) {
        fprintf(stderr, "no GPU visible (the library has no CPU fallback)\n");
        return 2;
    }
    CHECK(rc_scene_create(0, &scene));
    CHECK(rc_add_blas(scene, quad, NULL, 2, &blas));
    CHECK(rc_add_instances(scene, blas, xforms, ids, 2, &handle));
    CHECK(rc_sync(scene, NULL));
    CHECK(rc_trace_closest(scene, rays, hits, 4));
    for (int i = 0; i < 4; ++i)
        printf("ray %d: hit=%u t=%g prim=%u instance=%u custom=%u\n", i, hits[i].hit, hits[i].t, hits[i].primitive_id, hits[i].instance_id,
               hits[i].instance_custom_index);
    CHECK(rc_scene_destroy(scene));
    return 0;
}
