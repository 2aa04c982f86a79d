/* color.c -- packed colour helpers (API of the reference's include/color.h / src/color.c). */
#include <color.h>

#define BYTE_AT(v, shift) ((uint8_t)(((v) >> (shift)) & 0xffu))

ColorRGB make_color_rgb(uint8_t red, uint8_t green, uint8_t blue) {
    return ((uint32_t)red << 16) | ((uint32_t)green << 8) | (uint32_t)blue;
}
ColorRGBA make_color_rgba(uint8_t red, uint8_t green, uint8_t blue, uint8_t alpha) {
    return ((uint32_t)red << 24) | ((uint32_t)green << 16) | ((uint32_t)blue << 8) | (uint32_t)alpha;
}
ColorRGB get_color_rgba(ColorRGBA color) { return color >> 8; }
ColorRGBA get_color_rgb(ColorRGB color) { return (color << 8) | 0xffu; }
uint8_t get_red_rgb(ColorRGB color) { return BYTE_AT(color, 16); }
uint8_t get_green_rgb(ColorRGB color) { return BYTE_AT(color, 8); }
uint8_t get_blue_rgb(ColorRGB color) { return BYTE_AT(color, 0); }
uint8_t get_red_rgba(ColorRGBA color) { return BYTE_AT(color, 24); }
uint8_t get_green_rgba(ColorRGBA color) { return BYTE_AT(color, 16); }
uint8_t get_blue_rgba(ColorRGBA color) { return BYTE_AT(color, 8); }
uint8_t get_alpha_rgba(ColorRGBA color) { return BYTE_AT(color, 0); }
