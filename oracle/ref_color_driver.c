// ref_color_driver.c -- TEST INFRASTRUCTURE ONLY.
// Harness around the REFERENCE's own src/color.c (compiled in place from /root/reference; nothing is copied): the colour
// packing the flattener's leaf texels are made from (src/octree.cpp:587-590 reads red/green/blue through these) and
// include/color.h:33-46's other integer entry points. Emits JSON -> tests/golden/color.json: for every input the value of
// each of the eleven integer functions (the two float-vector ones are not on the path and not exported by the product).
#include <color.h>
#include <stdint.h>
#include <stdio.h>

int main(void) {
    uint32_t in[96];
    int n = 0;
    const uint32_t fixed[] = {0x00000000u, 0xffffffffu, 0x000000ffu, 0xff000000u, 0x00ff0000u, 0x0000ff00u, 0x12345678u,
                              0xa0a0a0ffu, 0x50b43cffu, 0xffd2d2ffu, 0x3c64dc96u, 0x80ff80c0u, 0x01020304u, 0xfffefdfcu,
                              0x7f808182u, 0x00ffffffu};
    for (unsigned i = 0; i < sizeof fixed / sizeof fixed[0]; ++i) in[n++] = fixed[i];
    uint32_t s = 0x9e3779b9u;
    while (n < 96) { s = s * 1664525u + 1013904223u; in[n++] = s; }
    printf("{\n \"generator\": \"oracle/ref_color_driver.c linked with the reference's src/color.c\",\n \"cases\": [\n");
    for (int i = 0; i < n; ++i) {
        const uint32_t c = in[i], rgb = c & 0xffffffu;
        const uint8_t b0 = (uint8_t)(c >> 24), b1 = (uint8_t)(c >> 16), b2 = (uint8_t)(c >> 8), b3 = (uint8_t)c;
        printf("  {\"in\": %u, \"make_color_rgb\": %u, \"make_color_rgba\": %u, \"get_color_rgba\": %u, \"get_color_rgb\": %u, "
               "\"get_red_rgb\": %u, \"get_red_rgba\": %u, \"get_green_rgb\": %u, \"get_green_rgba\": %u, "
               "\"get_blue_rgb\": %u, \"get_blue_rgba\": %u, \"get_alpha_rgba\": %u}%s\n",
               c, make_color_rgb(b0, b1, b2), make_color_rgba(b0, b1, b2, b3), get_color_rgba(c), get_color_rgb(rgb),
               get_red_rgb(rgb), get_red_rgba(c), get_green_rgb(rgb), get_green_rgba(c), get_blue_rgb(rgb), get_blue_rgba(c),
               get_alpha_rgba(c), i + 1 < n ? "," : "");
    }
    printf(" ]\n}\n");
    return 0;
}
