/* color.h -- packed colour helpers of the host API (reference: include/color.h,
 * src/color.c). ColorRGBA = R<<24 | G<<16 | B<<8 | A; the flatten step unpacks
 * it with these getters (reference src/octree.cpp:587-596). */
#ifndef VRT_COLOR_H
#define VRT_COLOR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t ColorRGBA;
typedef uint32_t ColorRGB;

ColorRGB make_color_rgb(uint8_t red, uint8_t green, uint8_t blue);
ColorRGBA make_color_rgba(uint8_t red, uint8_t green, uint8_t blue, uint8_t alpha);
ColorRGB get_color_rgba(ColorRGBA color);
ColorRGBA get_color_rgb(ColorRGB color);
uint8_t get_red_rgb(ColorRGB color);
uint8_t get_red_rgba(ColorRGBA color);
uint8_t get_green_rgb(ColorRGB color);
uint8_t get_green_rgba(ColorRGBA color);
uint8_t get_blue_rgb(ColorRGB color);
uint8_t get_blue_rgba(ColorRGBA color);
uint8_t get_alpha_rgba(ColorRGBA color);

#ifdef __cplusplus
}
#endif
#endif
